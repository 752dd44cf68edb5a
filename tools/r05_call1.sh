# round 5, GPU call 1: guided-error attribution per precision, the changed tests, unet_recon refreshed
tag=r5a; mkdir -p gpurun_out/$tag
python tools/recon_chain_error.py > gpurun_out/$tag/recon_chain_error.txt 2> gpurun_out/$tag/recon_chain_error.err
cat gpurun_out/$tag/recon_chain_error.txt; tail -n 3 gpurun_out/$tag/recon_chain_error.err
python -m pytest tests -m gpu -x -q -k "not config3 and not long_c3" > gpurun_out/$tag/pytest_gpu.log 2>&1; tail -n 6 gpurun_out/$tag/pytest_gpu.log
python bench.py --config unet_recon --no-cpu > gpurun_out/$tag/bench_unet_recon.json 2> gpurun_out/$tag/bench_unet_recon.err; tail -c 600 gpurun_out/$tag/bench_unet_recon.json
python bench.py --config unet_recon --batch 10 --no-cpu --no-pmc > gpurun_out/$tag/bench_unet_recon_b10.json 2> gpurun_out/$tag/bench_unet_recon_b10.err; tail -c 300 gpurun_out/$tag/bench_unet_recon_b10.json
bash tools/prof_config.sh unet_recon $tag 10
head -n 30 gpurun_out/$tag/unet_recon_kernel_stats_single_stream.md | cut -c1-160
