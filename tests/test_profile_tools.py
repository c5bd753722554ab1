"""tools/rocpd_summary.py on a hand-made rocpd database: the coarse launches of a step -- 768 bundles at 256 threads and the wide tail on a second
stream, overlapping -- are told apart from the refined-box launches of the same kernel name by their dynamic LDS, grouped into one span per step, and
the span of the timed steps is set beside the bench line's HIP-event figure (the roofline reproducible from profiles/ alone)."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rocpd_summary_spans(tmp_path):
    db = sqlite3.connect(tmp_path / "r.db")
    db.execute("create table top_kernels (name, total_calls, total_duration, average, percentage)")
    db.execute("create table kernels (name, start, end, duration, grid_x, workgroup_x, lds_size, static_lds_size, vgpr_count, sgpr_count, scratch_size)")
    name = "void dsa::k_fim_bundle<16, 256, 4, true>(dsa::FimBundle const*)"
    wide = "void dsa::k_fim_bundle<16, 768, 4, true>(dsa::FimBundle const*)"
    t, rows = 0, []
    for step in range(5):
        # refined boxes (dynamic LDS 40 B), then the coarse pair (2084 B): 250 ms and, from the same start, 340 ms
        rows.append((name, t, t + 7_000_000, 7_000_000, 768 * 256, 256, 49504 + 40, 49504, 84, 100, 48)); t += 8_000_000
        rows.append((name, t, t + 250_000_000, 250_000_000, 768 * 256, 256, 49464 + 2084, 49464, 84, 100, 48))
        rows.append((wide, t + 10_000, t + 340_000_000, 339_990_000, 232 * 768, 768, 84280 + 2084, 84280, 84, 100, 52)); t += 360_000_000
    db.executemany("insert into kernels values (?,?,?,?,?,?,?,?,?,?,?)", rows)
    db.execute("insert into top_kernels values (?,?,?,?,?)", (wide, 5, 1.7e6, 3.4e5, 57.0))
    db.commit(); db.close()
    line = tmp_path / "trace.log"
    line.write_text('{"metric": "m", "steps": 2, "kernel_ms_per_step": {"fim_coarse": 340.5}}\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), str(tmp_path / "r.db"), str(line)], capture_output=True, text=True, check=True).stdout
    assert "dispatches of the coarse launches (dynamic LDS 2084 B)" in out
    last = [l for l in out.splitlines() if l.startswith("headline coarse span per step")][0]
    assert "340.00 ms (mean of the last 2 of 5 spans of 1000 bundles" in last and "340.50 ms: -0.1 %" in last
    # the refined-box launches are rows of their own
    assert any("768" in l and " 40 " in l for l in out.splitlines() if l.startswith("k_fim_bundle<16, 256"))
