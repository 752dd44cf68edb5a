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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
