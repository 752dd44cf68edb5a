mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in old new old new old new; do
  if [ $v = old ]; then export CMDI_LIB_VARIANT=old; else unset CMDI_LIB_VARIANT; fi
  python bench.py --config c3 --steps 20 --warmup 3 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 $v ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/c3_ab.txt
