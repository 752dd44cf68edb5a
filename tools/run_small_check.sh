mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_split or attention_core or attention_h3_ignores or full_size_batch" 2>&1 | tail -3
for B in 2 10 2 10; do
  python bench.py --config c2 --batch $B --steps 200 --warmup 20 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B ring4 ms/step', round(d['ms_per_step'],4))"
  CMDI_ATTN_SPLIT=1 python bench.py --config c2 --batch $B --steps 200 --warmup 20 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B ring2 ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/small_ring.txt
