"""Summarise rocprofv3 --pmc results (rocpd sqlite) for kernels matching a pattern: mean counter value per dispatch."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%OpRound%"
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
q = "select counter_name, count(*), avg(value), sum(value) from counters_collection where kernel_name like ? group by counter_name" \
    if "counter_name" in cols and "kernel_name" in cols else None
if q is None:
    print("columns:", cols)
    sys.exit(0)
for r in cur.execute(q, (pat,)):
    print(f"{r[0]:28s} dispatches={r[1]:6d} mean={r[2]:.4g} sum={r[3]:.6g}")
