# round 5, GPU call 2: guided-error attribution at the operating point (+ the fp32-stash schedule), chunk-major conv K order A/B
tag=r5b; mkdir -p gpurun_out/$tag
python tools/recon_chain_error.py --stages 1b > gpurun_out/$tag/recon_opoint.txt 2> gpurun_out/$tag/recon_opoint.err
CMDI_LN_FOLD_KEEP=0 python tools/recon_chain_error.py --stages 1b,2 --modes f16x3 >> gpurun_out/$tag/recon_opoint.txt 2>> gpurun_out/$tag/recon_opoint.err
cat gpurun_out/$tag/recon_opoint.txt; tail -n 3 gpurun_out/$tag/recon_opoint.err
python -m pytest tests -m gpu -x -q -k "conv_rows or unet" > gpurun_out/$tag/pytest_unet.log 2>&1; tail -n 6 gpurun_out/$tag/pytest_unet.log
for v in none tapmajor none tapmajor; do
if [ $v != none ]; then export CMDI_LIB_VARIANT=$v; else unset CMDI_LIB_VARIANT; fi
python bench.py --config unet --steps 20 --warmup 5 --no-cpu --no-f32 > gpurun_out/$tag/bench_unet_$v.json 2> gpurun_out/$tag/bench_unet_$v.err
python - gpurun_out/$tag/bench_unet_$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("unet variant=%s ms/step %.4f conv_us %.1f frac %.3f traffic %s mfma_busy %s clock %s" % (sys.argv[2], d["ms_per_step"], r["avg_launch_us"], r["frac"], r.get("traffic"), r.get("mfma_busy"), r.get("effective_clock_ghz")))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
unset CMDI_LIB_VARIANT
