tag=r5q; mkdir -p gpurun_out/$tag
CMDI_UNET_COARSE_FAT=1 python -m pytest tests -m gpu -x -q -k "unet_forward_vs_reference or unet_vjp_vs_reference or unet_xl" 2>&1 | tail -n 2
for v in 0 1 2 0 1 2; do
CMDI_UNET_COARSE_FAT=$v python bench.py --config unet --steps 20 --warmup 5 --no-cpu --no-f32 --no-pmc 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet coarse_fat=$v ms/step %.4f' % d['ms_per_step'])"
done | tee gpurun_out/$tag/unet_coarse_fat.txt
for v in 0 1 0 1; do
CMDI_UNET_COARSE_FAT=$v python bench.py --config unet_recon --steps 20 --warmup 5 --no-cpu --no-f32 --no-pmc 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet_recon coarse_fat=$v ms/step %.4f' % d['ms_per_step'])"
done | tee -a gpurun_out/$tag/unet_coarse_fat.txt
for v in 0 1; do
CMDI_UNET_COARSE_FAT=$v python bench.py --config unet_recon --batch 10 --steps 20 --warmup 5 --no-cpu --no-f32 --no-pmc 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('unet_recon B=10 coarse_fat=$v ms/step %.4f' % d['ms_per_step'])"
done | tee -a gpurun_out/$tag/unet_coarse_fat.txt
