tag=r5h; mkdir -p gpurun_out/$tag
python tools/attn_bwd_replay2.py gpurun_out/$tag/attn_bwd_raw.npz 2> gpurun_out/$tag/err.txt | tail -n 2
tail -n 3 gpurun_out/$tag/err.txt
