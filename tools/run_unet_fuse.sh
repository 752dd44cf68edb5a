mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for F in 0 1 0 1 3 0 1; do
  CMDI_UNET_FUSE_GN=$F python bench.py --config unet --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet FUSE_GN=$F ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/unet_fuse.txt
