tag=r5f; mkdir -p gpurun_out/$tag
export CMDI_PROBES_LIB=1
CMDI_STASH_F32=6 python tools/recon_chain_error.py --stages audit2 2> gpurun_out/$tag/err.txt | tee gpurun_out/$tag/stash_audit2.txt
tail -n 5 gpurun_out/$tag/err.txt
