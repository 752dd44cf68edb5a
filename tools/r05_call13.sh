tag=r5m; mkdir -p gpurun_out/$tag
for c in 0 10000 20000 30000 40000 60000; do
echo "CMDI_DEPHASE_CYCLES=$c"; CMDI_DEPHASE_CYCLES=$c python tools/h3_small_ab.py 8,29 12608 2>> gpurun_out/$tag/err.txt
done | tee gpurun_out/$tag/dephase.txt
tail -n 3 gpurun_out/$tag/err.txt
