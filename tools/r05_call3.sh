# round 5, GPU call 3: which tensor of the folded schedule's split stash costs the guided VJP its accuracy
tag=r5c; mkdir -p gpurun_out/$tag
out=gpurun_out/$tag/recon_stash_switches.txt; : > $out
for bits in 0 1 2 4 6 7; do
CMDI_STASH_F32=$bits python tools/recon_chain_error.py --stages 1b --modes f16x3 2>> gpurun_out/$tag/err.txt | grep -v "^# stage\|numpy" >> $out
done
for bits in 1 6 7; do
CMDI_STASH_F32=$bits python tools/recon_chain_error.py --stages 2 --modes f16x3 2>> gpurun_out/$tag/err.txt | grep -v "^# stage" >> $out
done
cat $out; tail -n 3 gpurun_out/$tag/err.txt
