"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel calls / total / avg / min / max.

    python scripts/rocpd_stats.py gpurun_out/prof22/r01_results.db > profiles/rNN_kernel_stats.txt

(rocprofv3 in this image writes results.db by default; this is the `--stats` table re-derived from
the `kernels` view so the summary can be committed as text.)
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        f"from kernels group by {name_col} order by sum(end - start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'min_us':>11} {'max_us':>11} {'pct':>6}  kernel")
    for name, n, s, a, mn, mx in rows:
        short = name if len(name) < 150 else name[:147] + "..."
        print(f"{n:7d} {s / 1e6:11.3f} {a / 1e3:11.1f} {mn / 1e3:11.1f} {mx / 1e3:11.1f} {100.0 * s / tot:6.2f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1])
