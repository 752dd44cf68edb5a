mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in 0 -1 0 -1; do
  CMDI_H3_PERSIST=$P python bench.py --config c4 --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 B=256 two pipelines, CMDI_H3_PERSIST=$P ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/c4_persist.txt
