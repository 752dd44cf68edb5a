mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vjp or recon or autograd or chain_vs_reference or baseline_shape or keep_path or reference_callers" 2>&1 | tail -6 > gpurun_out/r4t/pytest_c3.txt
cat gpurun_out/r4t/pytest_c3.txt
for i in 1 2; do
python bench.py --config c3 --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/c3_after.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/tolerances_measured.json'))
for k,v in sorted(d.items()):
    if 'vjp' in k or 'autograd' in k or 'recon' in k or 'edit' in k or 'c3' in k: print(k, v)
PY
