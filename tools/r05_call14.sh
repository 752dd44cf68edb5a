rm -f gpurun_out/tolerances_measured.json
python -m pytest tests -m gpu -q > gpurun_out/r5_pytest_gpu.log 2>&1; tail -n 5 gpurun_out/r5_pytest_gpu.log
cp gpurun_out/tolerances_measured.json gpurun_out/r5_tolerances_measured.json 2>/dev/null
bash tools/refresh_profiles.sh r5
python tools/eval_client.py --batches 2 --mode edit > gpurun_out/r5/eval_client_edit.json 2> gpurun_out/r5/eval_client.err; tail -n 1 gpurun_out/r5/eval_client_edit.json | cut -c1-400
python tools/eval_client.py --batches 2 --mode plain > gpurun_out/r5/eval_client_plain.json 2>> gpurun_out/r5/eval_client.err; tail -n 1 gpurun_out/r5/eval_client_plain.json | cut -c1-400
python tools/eval_client.py --batches 1 --mode edit --model unet > gpurun_out/r5/eval_client_unet_edit.json 2>> gpurun_out/r5/eval_client.err; tail -n 1 gpurun_out/r5/eval_client_unet_edit.json | cut -c1-400
