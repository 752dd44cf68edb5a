# usage (GPU box): bash tools/ab_variants.sh <config> "<variant list: none p1 ...>"  — same-box A/B of library variants
# built by tools/variant_build.sh (CMDI_LIB_VARIANT); prints ms per step and the in-run time of the dominant GEMM
cfg=$1
for v in $2; do
if [ $v != none ]; then export CMDI_LIB_VARIANT=$v; else unset CMDI_LIB_VARIANT; fi
python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu --no-f32 --no-pmc 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('roofline_attention') or {}
print('$cfg variant=$v ms/step %.4f gemm_us %.1f attn_us %.1f' % (d['ms_per_step'], r.get('avg_launch_us',0), a.get('avg_launch_us',0)))"
done
