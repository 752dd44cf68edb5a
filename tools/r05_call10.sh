tag=r5j; mkdir -p gpurun_out/$tag
rm -f gpurun_out/tolerances_measured.json
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/$tag/pytest_gpu.log 2>&1; tail -n 40 gpurun_out/$tag/pytest_gpu.log
cp gpurun_out/tolerances_measured.json gpurun_out/$tag/ 2>/dev/null
