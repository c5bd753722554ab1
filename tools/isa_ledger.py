"""ISA ledger of the dominant kernel k_fim_sorted<256, compact>: per phase, the static instruction mix of the
gfx950 code and -- weighted with the wave-trip counters of a -DDSA_LEDGER build run on the GPU (tools/ledger_probe.py) -- where the
instructions of a solve go.

    python3 tools/isa_ledger.py [counters.json] > profiles/r03_isa_ledger.txt

Method.  fim_kernel.hip is compiled with -DDSA_LEDGER: every phase starts with a named marker (an assembly comment) and a counter of
wave trips through it.  The assembly of the kernel is cut at the markers, in text order, and the instructions of each piece are
classified.  dynamic = static instructions of the piece x its wave trips per solve: exact for straight-line pieces, an upper bound
where a piece holds wave-uniform skips (the unrolled 64-lane groups of pass A, the rare exception-table lookups); the sum is compared
with the SQ_INSTS_* counters of profiles/pmc_latest.json where that file matches the tree."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-DDSA_LEDGER_MARKS", "-S", "--cuda-device-only"]
KERNEL = "_ZN3dsa12k_fim_sortedILi256ELb1ELb0EEE"
FP32 = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_div_scale_f32", "v_div_fmas_f32",
        "v_div_fixup_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_min_f32", "v_max_f32", "v_mad_f32")


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep")):
            return "sync"
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch"
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    counters = None
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        with open(sys.argv[1]) as f:
            counters = json.load(f)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "fim.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(ROOT, "dsurftomo_amd", "csrc", "fim_kernel.hip")], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    meta = {}
    for i, l in enumerate(lines):
        if ".name:" in l and KERNEL in l and "symbol" not in l:
            for l2 in lines[max(0, i - 60):i + 60]:
                m = re.match(r"\s*\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s*(\d+)", l2)
                if m:
                    meta[m.group(1)] = int(m.group(2))
    seg_order, segs, copies = [], {}, {}
    cur = ("-", "prologue (before the first round)")
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r";\s*LEDGER (\d+) (\S+)", t)
        if m:
            cur = (m.group(1), m.group(2))
            copies[cur] = copies.get(cur, 0) + 1
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if cur not in segs:
            segs[cur] = {}
            seg_order.append(cur)
        d = segs[cur]
        c = classify(op)
        d[c] = d.get(c, 0) + 1
        if op.startswith("v_cndmask"):
            d["cndmask"] = d.get("cndmask", 0) + 1
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            d["lane_moves"] = d.get("lane_moves", 0) + 1
        if op.startswith(FP32):
            d["fp32_arith"] = d.get("fp32_arith", 0) + 1
        if op.startswith("scratch_"):
            d["scratch"] = d.get("scratch", 0) + 1
    cols = ["valu", "salu", "vmem", "lds", "smem", "branch", "sync", "cndmask", "lane_moves", "fp32_arith", "scratch"]
    print("ISA ledger of k_fim_sorted<256, compact> (gfx950, the tree's flags + -DDSA_LEDGER_MARKS: assembly comments only); static = instructions in the piece of code that follows the marker, in text order")
    print("kernel metadata: " + ", ".join("%s %s" % kv for kv in sorted(meta.items())))
    trips = counters["wave_trips_per_solve"] if counters else {}
    hdr = "%-4s %-22s %6s " % ("id", "piece", "copies") + " ".join("%8s" % c for c in cols) + "   %12s %12s %12s" % ("trips/solve", "VALU/solve", "SALU/solve")
    print(hdr)
    tot = {c: 0 for c in cols}
    dyn_v = dyn_s = dyn_m = dyn_l = 0.0
    rows = []
    for k in seg_order:
        d = segs[k]
        for c in cols:
            tot[c] += d.get(c, 0)
        tr = trips.get(k[0]) if k[0] != "-" else None
        nc = max(copies.get(k, 1), 1)          # an unrolled loop holds several copies of the piece: a trip runs one of them
        dv = d.get("valu", 0) / nc * tr if tr is not None else None
        ds_ = d.get("salu", 0) / nc * tr if tr is not None else None
        if tr is not None:
            dyn_v += dv; dyn_s += ds_; dyn_m += d.get("vmem", 0) / nc * tr; dyn_l += d.get("lds", 0) / nc * tr
        rows.append((k, d, tr, dv, ds_))
    for k, d, tr, dv, ds_ in rows:
        print("%-4s %-22s %6d " % (k[0], k[1], copies.get(k, 1)) + " ".join("%8d" % d.get(c, 0) for c in cols) + ("   %12.0f %12.0f %12.0f" % (tr, dv, ds_) if tr is not None else "   %12s %12s %12s" % ("-", "-", "-")))
    print("%-4s %-22s %6s " % ("", "total (static)", "") + " ".join("%8d" % tot[c] for c in cols))
    if counters:
        print()
        print("dynamic, per solve (%d units at N = %d, %s medium; %.0f evaluations, rounds <= %.0f): wave instructions = static x wave trips" %
              (counters["units"], counters["grid"], counters["medium"], counters["evals_per_solve"], counters["rounds_max"]))
        print("  VALU %.2f M   SALU %.2f M   VMEM %.2f M   LDS %.2f M" % (dyn_v / 1e6, dyn_s / 1e6, dyn_m / 1e6, dyn_l / 1e6))
        print("  share of the VALU wave instructions by piece:")
        for k, d, tr, dv, ds_ in sorted([r for r in rows if r[3]], key=lambda r: -r[3]):
            nc = max(copies.get(k, 1), 1)
            print("    %-22s %5.1f %%   (%.0f VALU per trip x %.0f trips; of the static %d: fp32 arithmetic %d, selects %d, lane moves %d)" %
                  (k[1], 100.0 * dv / dyn_v, d.get("valu", 0) / nc, tr, d.get("valu", 0), d.get("fp32_arith", 0), d.get("cndmask", 0), d.get("lane_moves", 0)))
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            if pmc.get("sq_insts_valu_per_solve"):
                print("  measured by the counters (profiles/pmc_latest.json: rocprofv3 --pmc of the bench command, production build, kernel source hash %s):" % pmc.get("kernel_source_hash"))
                print("    SQ_INSTS_VALU %.2f M   SQ_INSTS_SALU %.2f M per solve -> the straight-line estimate above is %.2f x / %.2f x the measured counts" %
                      (pmc["sq_insts_valu_per_solve"] / 1e6, pmc.get("sq_insts_salu_per_solve", 0) / 1e6, dyn_v / pmc["sq_insts_valu_per_solve"],
                       dyn_s / max(pmc.get("sq_insts_salu_per_solve", 1), 1)))
                print("    (pieces with wave-uniform skips inside -- the exception-table lookups of nodes_group64 / passB_loads / store_activate, the unrolled 64-lane")
                print("     groups of routing_window -- count their whole static body per trip: upper bounds)")


if __name__ == "__main__":
    main()
