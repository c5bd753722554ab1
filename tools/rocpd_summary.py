"""Summarise a rocprofv3 rocpd database (kernel trace) as text: per-kernel stats + the dispatches of
the dominant kernel with their launch resources.  Usage: python tools/rocpd_summary.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# per-kernel statistics (durations in us, as the top_kernels view of rocpd reports them)")
print("%-12s %-14s %-14s %-8s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-12d %-14.0f %-14.0f %-8.3f  %s" % (calls, total, avg, pct, name[:110]))
top = cur.execute("select name from top_kernels limit 1").fetchone()[0]
print("\n# dispatches of the dominant kernel:", top[:90])
print("%-6s %-14s %-10s %-10s %-8s %-8s %-8s %-10s" % ("#", "duration_ns", "grid_x", "wg_x", "vgpr", "sgpr", "lds", "scratch"))
for i, r in enumerate(cur.execute("select duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels where name=? order by start", (top,))):
    print("%-6d %-14d %-10d %-10d %-8d %-8d %-8d %-10d" % ((i,) + tuple(r)))
