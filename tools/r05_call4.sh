tag=r5d; mkdir -p gpurun_out/$tag
RECON_DUMP=gpurun_out/$tag/fold python tools/recon_chain_error.py --stages 1b --modes f16x3,bf16x6 2> gpurun_out/$tag/err.txt | tail -n 3
CMDI_LN_FOLD_KEEP=0 RECON_DUMP=gpurun_out/$tag/nofold python tools/recon_chain_error.py --stages 1b --modes f16x3 2>> gpurun_out/$tag/err.txt | tail -n 1
CMDI_LN_FOLD=0 RECON_DUMP=gpurun_out/$tag/nofold_all python tools/recon_chain_error.py --stages 1b --modes f16x3 2>> gpurun_out/$tag/err.txt | tail -n 1
