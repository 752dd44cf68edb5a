tag=r5l; mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -x -q -k "batch_independence or gemm_h3 or forward_other_shapes or forward_uncond or chain_vs_reference or vjp_vs_reference or graph_replay" > gpurun_out/$tag/pytest_subset.log 2>&1; tail -n 4 gpurun_out/$tag/pytest_subset.log
for b in 2 10; do
python bench.py --batch $b --no-cpu --no-f32 --no-pmc --steps 100 --warmup 20 > gpurun_out/$tag/bench_c2_b$b.json 2> gpurun_out/$tag/bench_c2_b$b.err
python - gpurun_out/$tag/bench_c2_b$b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "ms/step", d["ms_per_step"], "graph legs", d.get("hip_graph_legs"))
PY
done
python bench.py --config c3 --batch 10 --no-cpu --no-f32 --no-pmc --steps 50 --warmup 10 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 B=10 ms/step', d['ms_per_step'])"
