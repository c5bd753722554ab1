"""Per-kernel PMC counter sums from rocprofv3 rocpd databases.
Usage: python tools/rocpd_pmc.py results1.db [results2.db ...]
For every (kernel, counter): sum over dispatches, and the value of the longest-named dominant dispatch."""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cc = [t for t in tables if t.startswith("counters_collection")]
    if not cc:
        print(path, "no counters_collection view; tables:", tables[:40])
        continue
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % cc[0])]
    print("#", path, "columns:", cols)
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, counter_name, count(*), sum(value), max(value) from %s group by %s, counter_name" % (name_col, cc[0], name_col)
    rows = list(cur.execute(q))
    rows.sort(key=lambda r: (-r[3], r[0]))
    for k, c, n, s, m in rows:
        if s and s > 0:
            print("%-28s n=%-6d sum=%-14.6g max=%-14.6g %s" % (c, n, s, m, k[:70]))
