mkdir -p gpurun_out/r4z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --config unet --no-cpu > gpurun_out/r4z/bench_unet.json 2> gpurun_out/r4z/bench_unet.err
bash tools/prof_config.sh unet r4z 20
python bench.py --config c3 --no-cpu --no-pmc > gpurun_out/r4z/bench_c3.json 2> gpurun_out/r4z/bench_c3.err
for f in gpurun_out/r4z/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"))
PY
done
cut -c1-150 gpurun_out/r4z/unet_kernel_stats_single_stream.md | head -14
