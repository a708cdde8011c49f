"""Print per-kernel PMC sums from a rocprofv3 rocpd database (run on the GPU box; the .db files are too big to copy back)."""
import sqlite3
import sys

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    rows = c.execute("select counter_name, kernel_name, sum(value), count(*), avg(duration) from counters_collection "
                     "group by 1, 2 order by 3 desc limit 80").fetchall()
    for n, k, v, cnt, dur in rows:
        print("%s | %s | sum %.6g | dispatches %d | avg dur %.3f ms" % (n, k.replace("(anonymous namespace)::", "")[:60], v, cnt, dur / 1e6))
