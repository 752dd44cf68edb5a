mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vjp or recon or autograd or chain_vs_reference or baseline_shape or keep_path or reference_callers or forward_uncond or forward_text" 2>&1 | tail -6 > gpurun_out/r4t/pytest_c3.txt
cat gpurun_out/r4t/pytest_c3.txt
bash tools/run_c3_ab.sh
python - <<'PY'
import json
d=json.load(open('gpurun_out/tolerances_measured.json'))
for k,v in sorted(d.items()):
    if ('vjp' in k or 'autograd' in k or 'recon' in k or 'edit' in k) and 'unet' not in k: print(k, v['max'])
PY
