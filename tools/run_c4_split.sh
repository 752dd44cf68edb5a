mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for SP in 0 1; do for B in 64 256; do
  CMDI_ATTN_SPLIT=$SP python bench.py --config c4 --batch $B --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 B=$B split=$SP ms/step', round(d['ms_per_step'],4))"
done; done | tee gpurun_out/r4t/c4_split.txt
