# Small-batch anatomy (round 4): one step's kernel timeline at B=2 and B=10 (the reference scripts' default batch),
# eager, with the attention kernel as one 8-wave block per (sequence, head) and as two 4-wave blocks.
mkdir -p gpurun_out/r4s
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 2 10; do for SP in auto; do
  python bench.py --config c2 --batch $B --steps 200 --warmup 20 --no-cpu --no-pmc --no-roofline --no-f32 --no-graph-leg --precision f16x3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$B split=$SP ms/step', round(d['ms_per_step'],4))"
done; done
for B in 2 10; do
rocprofv3 --kernel-trace -d gpurun_out/r4s/prof_b$B -- python bench.py --config c2 --batch $B --steps 30 --warmup 5 --no-cpu --no-pmc --no-f32 --no-roofline --no-graph-leg --precision f16x3 > gpurun_out/r4s/prof_b$B.log 2>&1
python - $B <<'PY'
import sqlite3, glob, sys
B = sys.argv[1]
db = glob.glob(f"gpurun_out/r4s/prof_b{B}/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = list(cur.execute("select start, end, name, grid_x from kernels order by start")) if "kernels" in tabs else []
# one step = from one sampler_step kernel to the next
idx = [i for i, r in enumerate(rows) if "sampler_step" in r[2]]
a, b = idx[-3], idx[-2]
seg = rows[a + 1: b + 1]
out = open(f"gpurun_out/r4s/b{B}_step_timeline_after.md", "w")
out.write(f"# one step at B={B} (T=196, CFG): {len(seg)} kernels, span {(seg[-1][1]-seg[0][0])/1e3:.1f} us, kernel time {sum(e-s for s,e,_,_ in seg)/1e3:.1f} us\n\n| # | kernel | grid | us | gap before us |\n|---|---|---|---|---|\n")
prev = rows[a][1]
for i, (s, e, n, g) in enumerate(seg):
    out.write(f"| {i} | `{n[:70]}` | {g} | {(e-s)/1e3:.1f} | {(s-prev)/1e3:.1f} |\n")
    prev = e
out.close()
print(open(f"gpurun_out/r4s/b{B}_step_timeline_after.md").read())
PY
rm -rf gpurun_out/r4s/prof_b$B
done
