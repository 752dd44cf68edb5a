mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for G in 2 3 4 2 3 4; do
  CMDI_GROUPS=$G python bench.py --config c4 --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 CMDI_GROUPS=$G ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/parts_c4.txt
