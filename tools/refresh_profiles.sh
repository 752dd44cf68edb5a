mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r2/bench_c2_driver_cmd.json 2> gpurun_out/r2/bench_c2_driver_cmd.err
python bench.py --no-cpu > gpurun_out/r2/bench_c2.json 2> gpurun_out/r2/bench_c2.err
for c in c3 c4 c5 unet; do python bench.py --config $c --no-cpu > gpurun_out/r2/bench_$c.json 2> gpurun_out/r2/bench_$c.err; done
CMDI_GROUPS=1 CMDI_PIPELINES=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_c2 -- python bench.py --steps 40 --warmup 5 --no-cpu --no-pmc --no-f32 --no-roofline > gpurun_out/r2/prof_c2.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/r2/prof_c2 -name "*.db" | head -1)" gpurun_out/r2/c2_kernel_stats_single_stream.md "round 2 (final), LN folded, CMDI_GROUPS=1 CMDI_PIPELINES=0: bench.py --steps 40 --warmup 5 (c2)" > /dev/null 2>&1
rm -rf gpurun_out/r2/prof_c2
tail -c 600 gpurun_out/r2/bench_c2_driver_cmd.err
for f in gpurun_out/r2/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
head -12 gpurun_out/r2/c2_kernel_stats_single_stream.md
