"""rocprofv3 --kernel-trace --stats writes a rocpd sqlite database in this image; turn it into the text
summary committed under profiles/ (top kernels: calls, total, average, share) plus per-kernel resources."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc limit 24"))
grand = list(cur.execute("select sum(duration) from kernels"))[0][0]
rows = [(n, c, t, a, 100.0 * t / grand) for n, c, t, a in rows]
print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'%':>7}  kernel   (durations from the kernel-dispatch table, ns)")
for name, calls, tot, avg, pct in rows:
    short = name if len(name) < 150 else name[:110] + " ... " + name[-30:]
    print(f"{calls:7d} {tot / 1e6:11.3f} {avg / 1e3:11.3f} {pct:7.3f}  {short}")
print()
print("engine kernels (k_game<N, GAME, Op>): launch geometry and resources")
for r in cur.execute("select name, count(*), avg(duration), min(duration), max(duration), grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, "
                     "sgpr_count from kernels where name like '%k_game%' group by name"):
    print(f"  {r[0]}: calls={r[1]} avg={r[2] / 1e3:.1f}us min={r[3] / 1e3:.1f}us max={r[4] / 1e3:.1f}us grid={r[5]} wg={r[6]} lds={r[7]}B "
          f"scratch={r[8]}B vgpr={r[9]} sgpr={r[10]}")
