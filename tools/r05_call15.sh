tag=r5n; mkdir -p gpurun_out/$tag
for v in none h3psc1 none h3psc1; do
if [ $v != none ]; then export CMDI_LIB_VARIANT=$v; else unset CMDI_LIB_VARIANT; fi
echo "variant $v"; python tools/h3_small_ab.py 8,50 100864 2>> gpurun_out/$tag/err.txt | head -n 1
done | tee gpurun_out/$tag/h3p_sc1_time.txt
export CMDI_LIB_VARIANT=h3psc1
python tools/inproj_l2_pmc.py 2>> gpurun_out/$tag/err.txt | tee gpurun_out/$tag/h3p_sc1_counters.txt
unset CMDI_LIB_VARIANT
for v in none h3psc1 none h3psc1; do
if [ $v != none ]; then export CMDI_LIB_VARIANT=$v; else unset CMDI_LIB_VARIANT; fi
python bench.py --config c4 --no-cpu --no-f32 --no-pmc --no-graph-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 variant $v ms/step', d['ms_per_step'], 'in_proj us', d['roofline']['avg_launch_us'])"
done | tee -a gpurun_out/$tag/h3p_sc1_time.txt
tail -n 2 gpurun_out/$tag/err.txt
