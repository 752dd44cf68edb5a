for v in 0 1; do
echo "--- trasm=$v"; CMDI_ATTN_TRASM=$v CMDI_ATTN_DBG=16 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-110
done
CMDI_ATTN_TRASM=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "attention_core" 2>&1 | tail -3
for v in 0 1 0 1; do
CMDI_ATTN_TRASM=$v python bench.py --config c2 --steps 40 --warmup 5 --no-cpu --no-pmc --no-f32 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trasm=$v', d['ms_per_step'])"
done
