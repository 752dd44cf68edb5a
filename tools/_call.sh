cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "unet" 2>&1 | tail -4
for rep in 1 2; do for b in 32 10 2; do for k in 4 1; do
CMDI_UNET_KSPLIT=$k python bench.py --config unet_recon --batch $b --no-cpu --no-f32 --no-pmc --no-roofline --no-graph-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep unet_recon B=$b CMDI_UNET_KSPLIT=$k ms_per_step', round(d['ms_per_step'],4))"
done; done; done
