cd $GRAFT_REPO_ROOT
for rep in 1 2; do for b in 10 2; do for g in 1 2 3; do
CMDI_GROUPS=$g python bench.py --batch $b --no-cpu --no-f32 --no-pmc --no-roofline --no-graph-leg --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep B=$b CMDI_GROUPS=$g ms_per_step', round(d['ms_per_step'],4))"
done; done; done
