tag=r5o; mkdir -p gpurun_out/$tag
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
rm -f gpurun_out/tolerances_measured.json
python -m pytest tests -m gpu -q > gpurun_out/$tag/pytest_gpu.log 2>&1; tail -n 3 gpurun_out/$tag/pytest_gpu.log
cp gpurun_out/tolerances_measured.json gpurun_out/$tag/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/$tag/bench_c2_driver_cmd.json 2> gpurun_out/$tag/bench_c2_driver_cmd.err
tail -c 200 gpurun_out/$tag/bench_c2_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5o/bench_c2_driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'us', r['avg_launch_us'], 'cpu', d['cpu_baseline']['value'])
PY
python bench.py --config c3 --no-cpu --no-f32 --no-pmc 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 ms/step', d['ms_per_step'])"
python bench.py --batch 10 --no-cpu --no-f32 --no-pmc --steps 100 --warmup 20 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2 B=10 ms/step', d['ms_per_step'])"
