# round-end evidence from ONE box: bench lines of every config + single-stream kernel statistics (c2, c3, c4, unet)
# usage (on the GPU box): bash tools/refresh_profiles.sh [tag]   -> gpurun_out/<tag>/
tag=${1:-r4}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/$tag/bench_c2_driver_cmd.json 2> gpurun_out/$tag/bench_c2_driver_cmd.err
for c in c3 c4 c5 unet; do python bench.py --config $c --no-cpu > gpurun_out/$tag/bench_$c.json 2> gpurun_out/$tag/bench_$c.err; done
bash tools/prof_config.sh c2 $tag 40
for c in c3 c4 unet; do bash tools/prof_config.sh $c $tag 20; done
tail -c 300 gpurun_out/$tag/bench_c2_driver_cmd.err
for f in gpurun_out/$tag/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
head -12 gpurun_out/$tag/c2_kernel_stats_single_stream.md | cut -c1-150
