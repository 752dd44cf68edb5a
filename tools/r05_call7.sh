tag=r5g; mkdir -p gpurun_out/$tag
python tools/attn_bwd_replay.py 2> gpurun_out/$tag/err.txt | tee gpurun_out/$tag/attn_bwd_replay.txt | cut -c1-400
tail -n 3 gpurun_out/$tag/err.txt
