mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 2 8; do python bench.py --config c2 --batch $B --steps 100 --warmup 10 --no-cpu --no-pmc --no-roofline --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B ms/step', round(d['ms_per_step'],4))"; done
rocprofv3 --kernel-trace --stats -d gpurun_out/r3f/prof_b2 -- python bench.py --config c2 --batch 2 --steps 50 --warmup 5 --no-cpu --no-pmc --no-f32 --no-roofline --precision f16x3 > gpurun_out/r3f/prof_b2.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/r3f/prof_b2 -name "*.db" | head -1)" gpurun_out/r3f/b2_kernel_stats.md "B=2 (T=196, CFG) bench.py --batch 2 --steps 50" > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/r3f/prof_b2/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select start, end, name from kernels order by start"))
# the last full step: gaps between kernels
n = len(rows)
seg = rows[n - 70: n - 5]
tot = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print("last 65 kernels: span %.1f us, kernel time %.1f us, gaps %.1f us" % (tot / 1e3, busy / 1e3, (tot - busy) / 1e3))
PY
rm -rf gpurun_out/r3f/prof_b2
cut -c1-140 gpurun_out/r3f/b2_kernel_stats.md | head -22
