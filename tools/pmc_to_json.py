"""profiles/pmc_latest.json from the rocprofv3 --pmc passes of tools/profile_bench.sh.

    python3 tools/pmc_to_json.py <dir with fetch/ write/ busy/ insts/ sub-directories> <solves per launch> <out.json>

What bench.py reads from it (only when `kernel_source_hash` matches the sources in the tree):
  fabric_bytes_per_solve  (FETCH_SIZE * cf + WRITE_SIZE * cw) * 1024 / solves of the dominant kernel's launch.  FETCH_SIZE /
                          WRITE_SIZE count requests on the L2's memory side (Infinity-Cache hits included, MI355X_MICROARCH.md);
                          cf, cw are the calibration factors of profiles/fetch_calibration.json (tools/micro/fetch_calib.hip: the
                          same counters on a kernel that gathers a known number of 8-byte records out of 512-B tiles), 1.0 and
                          flagged "uncalibrated" when that file is absent.
  valu_issue              SQ_ACTIVE_INST_VALU / (8 * SQ_BUSY_CYCLES): SQ_ACTIVE_INST_VALU counts quad-cycles per wave, summed
                          over waves; SQ_BUSY_CYCLES counts cycles per shader engine (32 of them, 32 SIMDs each): the share of
                          the chip's SIMD issue cycles that carried a VALU instruction.
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counters(dbdir, kernel_substr):
    out = {}
    for dirpath, _, files in os.walk(dbdir):
        for f in files:
            if not f.endswith(".db"):
                continue
            db = sqlite3.connect(os.path.join(dirpath, f))
            cur = db.cursor()
            tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
            cc = [t for t in tables if t.startswith("counters_collection")]
            if not cc:
                continue
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % cc[0])]
            name_col = "kernel_name" if "kernel_name" in cols else "name"
            # per dispatch of the dominant kernel: the largest one is the coarse solve (the refined launch is ~3 % of it); `sum` is for a
            # step that takes several launches of it (the bundle kernel: whole bundles, then the halves of the last ones)
            q = "select counter_name, max(value), count(*), sum(value) from %s where %s like ? group by counter_name" % (cc[0], name_col)
            for c, v, n, t in cur.execute(q, ("%" + kernel_substr + "%",)):
                out[c] = {"max_per_dispatch": v, "dispatches": n, "sum": t}
    return out


def dispersion(d, roots, dst):
    """profiles/pmc_dispersion.json: FP64 work of k_dispersion per root.  SQ_INSTS_VALU_FLOPS_FP64 counts flops per wave
    instruction (2 per FMA, 1 per ADD / MUL / transcendental); x 64 lanes x lane fill, the lane fill being
    SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) of the same pass."""
    import bench
    c = counters(d, "k_dispersion")
    need = ("SQ_INSTS_VALU_FLOPS_FP64", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")
    if not all(k in c for k in need):
        raise SystemExit("missing counters: have %s" % sorted(c))
    fill = c["SQ_THREAD_CYCLES_VALU"]["max_per_dispatch"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]["max_per_dispatch"])
    flops = c["SQ_INSTS_VALU_FLOPS_FP64"]["max_per_dispatch"] * 64.0 * fill
    rec = {"kernel_source_hash": bench.kernel_source_hash(bench.DISP_SOURCES), "kernel": "k_dispersion", "roots": roots, "counters": c, "lane_fill": round(fill, 4),
           "fp64_flops_per_root": flops / roots,
           "fp64_transcendentals_per_root": c.get("SQ_INSTS_VALU_FLOPS_FP64_TRANS", {}).get("max_per_dispatch", 0) * 64.0 * fill / roots,
           "command": "bash tools/collect_pmc.sh <tag> flops64 - -- python3 tools/disp_roofline.py 1"}
    if "SQ_BUSY_CYCLES" in c:
        rec["valu_issue"] = round(c["SQ_ACTIVE_INST_VALU"]["max_per_dispatch"] / (8.0 * c["SQ_BUSY_CYCLES"]["max_per_dispatch"]), 4)
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "counters"}, indent=1))


def main():
    if sys.argv[1] == "--dispersion":
        return dispersion(sys.argv[2], float(sys.argv[3]), sys.argv[4])
    d, solves, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "k_fim"
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # > 0: the counters of ALL launches of `kernel`, over `steps` steps
    import bench
    c = counters(d, kernel)
    if steps > 0:
        for k in c:
            c[k]["max_per_dispatch"] = c[k]["sum"] / steps          # (per step: what the per-solve figures below divide)
            c[k]["per"] = "step (all launches of the kernel in it)"
    cal_path = os.path.join(ROOT, "profiles", "fetch_calibration.json")
    cf = cw = 1.0
    cal = "uncalibrated (profiles/fetch_calibration.json absent)"
    if os.path.exists(cal_path):
        with open(cal_path) as f:
            k = json.load(f)
        cf, cw = k["fetch_factor"], k["write_factor"]
        cal = "profiles/fetch_calibration.json: fetch x%.3f, write x%.3f" % (cf, cw)
    rec = {"kernel_source_hash": bench.kernel_source_hash(), "kernel": kernel, "solves_per_launch": solves, "counters": c, "calibration": cal,
           "command": "tools/profile_bench.sh: rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0, one pass per group"}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["fabric_bytes_per_solve"] = (c["FETCH_SIZE"]["max_per_dispatch"] * cf + c["WRITE_SIZE"]["max_per_dispatch"] * cw) * 1024.0 / solves
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
        rec["valu_issue"] = round(c["SQ_ACTIVE_INST_VALU"]["max_per_dispatch"] / (8.0 * c["SQ_BUSY_CYCLES"]["max_per_dispatch"]), 4)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"):
        if k in c:
            rec[k.lower() + "_per_solve"] = c[k]["max_per_dispatch"] / solves
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "counters"}, indent=1))


if __name__ == "__main__":
    main()
