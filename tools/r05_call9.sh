tag=r5i; mkdir -p gpurun_out/$tag
python tools/attn_bwd_replay.py 2> gpurun_out/$tag/err.txt | grep -v "^ layer . u\|unscaled" | head -n 12 | cut -c1-330
python tools/recon_chain_error.py --stages 1,1b,2 > gpurun_out/$tag/recon_chain_error_after_fix.txt 2>> gpurun_out/$tag/err.txt
cat gpurun_out/$tag/recon_chain_error_after_fix.txt
python -m pytest tests -m gpu -x -q -k "attention or vjp or conv_rows_x6 or unet" > gpurun_out/$tag/pytest_subset.log 2>&1; tail -n 15 gpurun_out/$tag/pytest_subset.log
tail -n 3 gpurun_out/$tag/err.txt
