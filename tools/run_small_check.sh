mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "attention_split or attention_core or forward_ or test_chain_vs_reference or full_size_batch or sharded_p_sample" 2>&1 | tail -5 > gpurun_out/r4t/pytest_small.txt
cat gpurun_out/r4t/pytest_small.txt
for B in 2 10 32; do
  python bench.py --config c2 --batch $B --steps 200 --warmup 20 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B ms/step', round(d['ms_per_step'],4))"
done | tee gpurun_out/r4t/small_bench.txt
