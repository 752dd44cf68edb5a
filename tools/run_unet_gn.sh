mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for G in 1 2 1 2 1 2; do
  CMDI_UNET_GN1=$G python bench.py --config unet --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet GN1=$G ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/unet_gn2.txt
