# usage (GPU box): bash tools/h3w_step_ab.sh [config] — same-box A/B of the C2 step with / without the weight-stationary GEMM
# (CMDI_H3W) and with one / two sequence groups (CMDI_GROUPS); prints ms per step of each combination, twice.
cfg=${1:-c2}
for rep in 1 2; do
for h3w in 0 1; do for grp in 0 1; do
  ms=$(CMDI_H3W=$h3w CMDI_GROUPS=$grp python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu --no-roofline --no-pmc --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "rep $rep  CMDI_H3W=$h3w CMDI_GROUPS=$grp  ms_per_step $ms"
done; done; done
