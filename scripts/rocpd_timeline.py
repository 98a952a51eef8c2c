"""Print a per-kernel timeline (start/end in ms relative to the first kernel of the last proof) from a
rocprofv3 rocpd database: shows which kernels of different streams actually overlap.
    python scripts/rocpd_timeline.py results.db [n_last_kernels]
"""
import sqlite3
import sys


def main(path, last=140):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
    sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    rows = rows[-last:]
    t0 = rows[0][1]
    for name, st, en, q, s in rows:
        short = name.replace("void g16::(anonymous namespace)::", "").replace("g16::", "").replace("(anonymous namespace)::", "")
        short = short.split("(")[0][:60]
        print(f"{(st - t0) / 1e6:9.3f} {(en - t0) / 1e6:9.3f} {(en - st) / 1e6:8.3f}  q={q} s={s}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 140)
