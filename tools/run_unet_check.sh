mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unet" 2>&1 | tail -6 > gpurun_out/r4t/pytest_unet.txt
cat gpurun_out/r4t/pytest_unet.txt
for G in 0 1 0 1; do
  CMDI_UNET_GN1=$G python bench.py --config unet --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet GN1=$G ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/unet_gn1.txt
