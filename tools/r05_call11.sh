tag=r5k; mkdir -p gpurun_out/$tag
python tools/h3_small_ab.py 21,8,24,25,26,28 788,1970,3940 2> gpurun_out/$tag/err.txt | tee gpurun_out/$tag/h3_small_ab.txt
python -m pytest tests -m gpu -x -q -k "graph_cache_hit" 2>&1 | tail -n 3
tail -n 3 gpurun_out/$tag/err.txt
