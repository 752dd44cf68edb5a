#!/bin/bash
# Samples rocm-smi clocks / power while a long chain runs (is the f16x3 GEMM clock a power limit?).
mkdir -p gpurun_out
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/power_samples.txt &
SP=$!
python bench.py --config c5 --steps 300 --warmup 20 --no-cpu --no-pmc --no-f32 --no-roofline > gpurun_out/power_bench.json 2>/dev/null
python bench.py --config c5 --precision f32 --steps 150 --warmup 20 --no-cpu --no-pmc --no-f32 --no-roofline > gpurun_out/power_bench_f32.json 2>/dev/null
kill $SP
python - <<'PY'
import re
rows=[]
for ln in open("gpurun_out/power_samples.txt"):
    m=re.findall(r"\((\d+)Mhz\)", ln); p=re.findall(r"(\d+\.\d+)\s*$", ln.strip())
    if m and p: rows.append((int(m[0]), float(p[0])))
busy=[r for r in rows if r[1] > 400]
print(len(rows), "samples;", len(busy), "under load")
import statistics as st
if busy:
    print("sclk MHz  min/median/max:", min(r[0] for r in busy), st.median(r[0] for r in busy), max(r[0] for r in busy))
    print("power W   min/median/max:", min(r[1] for r in busy), st.median(r[1] for r in busy), max(r[1] for r in busy))
    half=len(busy)//2
print(busy[:40]); print(busy[-40:])
PY
