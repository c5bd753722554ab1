"""Summarise a rocprofv3 rocpd database (kernel trace) as text: per-kernel stats; the dominant kernel of the headline step (k_fim_bundle) BY
LAUNCH CONFIGURATION -- grid, workgroup, dynamic LDS: the coarse launches of a step (768 bundles at 256 threads and the wide tail at 768 threads on
a second stream, both with the coarse grid's tile bitmap in LDS) and the refined-box launches are different rows of the same kernel name --; the
dispatches of the coarse launches with their start / end times; and the COARSE SPAN PER STEP: from the first start to the last end of the
coarse launches that overlap or follow one another within a millisecond -- the time bench.py's HIP events bracket (kernel_ms_per_step.fim_coarse,
roofline.achieved).  Usage: python tools/rocpd_summary.py results.db [bench line (JSON file or trace.log) to compare with]"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# per-kernel statistics (durations in us, as the top_kernels view of rocpd reports them)")
print("%-12s %-14s %-14s %-8s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-12d %-14.0f %-14.0f %-8.3f  %s" % (calls, total, avg, pct, name[:110]))

rows = cur.execute("select name,start,end,grid_x,workgroup_x,lds_size - static_lds_size,vgpr_count,sgpr_count,scratch_size from kernels where name like '%k_fim_bundle%' order by start").fetchall()
if not rows:
    top = cur.execute("select name from top_kernels limit 1").fetchone()[0]
    print("\n# no k_fim_bundle dispatch in this trace; dispatches of the dominant kernel:", top[:90])
    for i, r in enumerate(cur.execute("select duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels where name=? order by start", (top,))):
        print("%-6d %-14d %-10d %-10d %-8d %-8d %-8d %-10d" % ((i,) + tuple(r)))
    sys.exit(0)

t0 = rows[0][1]
cfg = {}
for name, s, e, gx, wx, lds, vg, sg, scr in rows:
    short = name[name.find("k_fim_bundle"):][:40]
    cfg.setdefault((short, gx // max(wx, 1), wx, lds, vg, scr), []).append((s, e))
print("\n# k_fim_bundle by launch configuration (workgroups = bundles of the launch; dyn_lds = the grid's tile bitmap: 40 B = a 129^2 refined box, ~2 KB = the 1025^2 coarse grid)")
print("%-42s %-10s %-6s %-8s %-6s %-8s %-7s %-12s %-12s" % ("kernel", "workgroups", "wg_x", "dyn_lds", "vgpr", "scratch", "calls", "avg_ms", "total_ms"))
for k in sorted(cfg, key=lambda k: -sum(e - s for s, e in cfg[k])):
    d = [e - s for s, e in cfg[k]]
    print("%-42s %-10d %-6d %-8d %-6d %-8d %-7d %-12.3f %-12.3f" % (k[0], k[1], k[2], k[3], k[4], k[5], len(d), sum(d) / len(d) / 1e6, sum(d) / 1e6))

tot = {}
for k in cfg:
    tot[k[3]] = tot.get(k[3], 0) + sum(e - s for s, e in cfg[k])
lds_coarse = max(tot, key=tot.get)          # (the grid whose launches take the most time: the headline's coarse grid)
coarse = sorted((s, e, k[1], k[2]) for k in cfg if k[3] == lds_coarse for s, e in cfg[k])
print("\n# dispatches of the coarse launches (dynamic LDS %d B), times in ms from the first k_fim_bundle dispatch" % lds_coarse)
print("%-5s %-12s %-12s %-12s %-10s %-6s" % ("#", "start", "end", "duration", "workgroups", "wg_x"))
for i, (s, e, nwg, wx) in enumerate(coarse):
    print("%-5d %-12.3f %-12.3f %-12.3f %-10d %-6d" % (i, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, nwg, wx))
spans = []
for s, e, nwg, wx in coarse:
    if spans and s - spans[-1][1] < 1_000_000:
        spans[-1][1] = max(spans[-1][1], e); spans[-1][2] += 1; spans[-1][3] += nwg
    else:
        spans.append([s, e, 1, nwg])
print("\n# coarse spans (launches that overlap or follow within 1 ms = the coarse solves of one step / one chunk)")
print("%-5s %-12s %-12s %-10s %-10s" % ("#", "start_ms", "span_ms", "launches", "bundles"))
for i, (s, e, nl, nb) in enumerate(spans):
    print("%-5d %-12.3f %-12.3f %-10d %-10d" % (i, (s - t0) / 1e6, (e - s) / 1e6, nl, nb))
# the headline's steps: the spans with the most bundles (secondary legs and the settle passes of other sizes have fewer or other grids)
nb_head = max(sp[3] for sp in spans)
head = [(e - s) / 1e6 for s, e, nl, nb in spans if nb == nb_head]
bench_ms, bench_steps = None, 0
if len(sys.argv) > 2:
    try:
        for line in open(sys.argv[2]):
            if '"metric"' in line:
                rec = json.loads(line[line.find("{"):])
                bench_ms, bench_steps = rec.get("kernel_ms_per_step", {}).get("fim_coarse"), int(rec.get("steps", 0))
    except (OSError, ValueError):
        pass
# the timed steps: the last `steps` headline spans of a --no-secondary run (settle and warm-up passes come first); without a bench line the later half
steps = head[-bench_steps:] if 0 < bench_steps <= len(head) else (head[-(len(head) // 2):] if len(head) >= 4 else head)
mean = sum(steps) / len(steps)
line = "headline coarse span per step: %.2f ms (mean of the last %d of %d spans of %d bundles; min %.2f, max %.2f)" % (mean, len(steps), len(head), nb_head, min(steps), max(steps))
if bench_ms:
    line += "; bench.py kernel_ms_per_step.fim_coarse (HIP events, same run) %.2f ms: %+.1f %%" % (bench_ms, 100.0 * (mean - bench_ms) / bench_ms)
print("\n" + line)
