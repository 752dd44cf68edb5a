"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a markdown table for profiles/."""
import sqlite3
import sys


def main(db, out, title):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
        "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nTotal kernel time {tot:.1f} ms\n\n"
                "| kernel | calls | total ms | % | avg us | min us | max us |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:16]:
            f.write(f"| `{r[0][:120]}` | {r[1]} | {r[2]:.2f} | {100 * r[2] / tot:.1f} | {r[3]:.1f} "
                    f"| {r[4]:.1f} | {r[5]:.1f} |\n")
    print(open(out).read())


def by_grid(db, out):
    """Per (kernel, grid size) durations — separates the GEMM shapes hiding behind one kernel name."""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gcol = next((c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols), None)
    if gcol is None:
        print("no grid column among", cols)
        return
    rows = list(cur.execute(
        f"select name, {gcol}, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels "
        f"group by name, {gcol} order by 4 desc"))
    tot = sum(r[3] for r in rows)
    with open(out, "w") as f:
        f.write("| kernel | grid | calls | total ms | % | avg us |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write(f"| `{r[0][:90]}` | {r[1]} | {r[2]} | {r[3]:.2f} | {100 * r[3] / tot:.1f} | {r[4]:.1f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "--by-grid":
        by_grid(sys.argv[2], sys.argv[3])
    else:
        main(sys.argv[1], sys.argv[2], sys.argv[3])
