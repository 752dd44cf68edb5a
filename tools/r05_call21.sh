rm -f gpurun_out/tolerances_measured.json
python -m pytest tests -m gpu -q > gpurun_out/r5_final_pytest_gpu.log 2>&1; tail -n 3 gpurun_out/r5_final_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
