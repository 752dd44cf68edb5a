# round-end evidence from ONE box: bench lines of every config + single-stream kernel statistics (c2, c3, c4, unet, unet_recon)
# usage (on the GPU box): bash tools/refresh_profiles.sh [tag]   -> gpurun_out/<tag>/
tag=${1:-r5}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/$tag/bench_c2_driver_cmd.json 2> gpurun_out/$tag/bench_c2_driver_cmd.err
for c in c3 c4 c5 unet unet_recon; do python bench.py --config $c --no-cpu > gpurun_out/$tag/bench_$c.json 2> gpurun_out/$tag/bench_$c.err; done
for b in 2 10; do python bench.py --batch $b --no-cpu --no-f32 --no-pmc --steps 100 --warmup 20 > gpurun_out/$tag/bench_c2_batch$b.json 2> gpurun_out/$tag/bench_c2_batch$b.err; done
python bench.py --config unet_recon --batch 10 --no-cpu --no-f32 --no-pmc > gpurun_out/$tag/bench_unet_recon_batch10.json 2> gpurun_out/$tag/bench_unet_recon_batch10.err
bash tools/prof_config.sh c2 $tag 40
for c in c3 c4 unet unet_recon; do bash tools/prof_config.sh $c $tag 20; done
python tools/inproj_l2_pmc.py > gpurun_out/$tag/inproj_l2_counters.txt 2> gpurun_out/$tag/inproj_l2_counters.err
tail -c 300 gpurun_out/$tag/bench_c2_driver_cmd.err
for f in gpurun_out/$tag/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("kernel","")[:40])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
head -12 gpurun_out/$tag/c2_kernel_stats_single_stream.md | cut -c1-150
cat gpurun_out/$tag/inproj_l2_counters.txt
