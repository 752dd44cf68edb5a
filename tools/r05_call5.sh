tag=r5e; mkdir -p gpurun_out/$tag
export CMDI_PROBES_LIB=1
CMDI_STASH_F32=6 python tools/recon_chain_error.py --stages audit 2> gpurun_out/$tag/err.txt | tee gpurun_out/$tag/stash_audit.txt
CMDI_H3_TILE=8 RECON_DUMP=gpurun_out/$tag/tile8 python tools/recon_chain_error.py --stages 1b --modes f16x3 2>> gpurun_out/$tag/err.txt | tail -n 1
CMDI_H3_TILE=21 RECON_DUMP=gpurun_out/$tag/tile21 python tools/recon_chain_error.py --stages 1b --modes f16x3 2>> gpurun_out/$tag/err.txt | tail -n 1
tail -n 3 gpurun_out/$tag/err.txt
