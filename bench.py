#!/usr/bin/env python3
"""Benchmark of the CalSurfG hot path on MI355X: (period, source) eikonal solves per second.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2] / configs[3]): 1025x1025 propagation grid
(nx = ny = 131 velocity vertices), 16 periods, 1000 sources, 32 receivers per source, smooth
2-D phase-velocity maps pv_p = (2.8 + 0.05 p)(1 + 0.10 sin(4 pi i/nx) cos(4 pi j/ny)).  A "step"
is one pass of the hot path over the whole set: 16 000 solves (refine + two fixed-point solves)
plus 512 000 receiver times.  Velocity maps and source/receiver descriptors are resident in HBM
before the timed region; the receiver times come back to the host inside it.

With N > 1 the 16 000 units are split by SOURCES: rank r takes sources r, r + N, ... with all their
periods (the units of the loop nest of CalSurfG.f90:1144-1145 are independent; a source's periods
stay on one rank so that ranks keep whole bundles), every rank solves its share, and the
receiver-time vector is completed on every rank by an RCCL all-gather and put into the reference's
order: total work is fixed ("strong" scaling, as configs[3] states).  `python bench.py --gpus N` without a launcher starts the N ranks itself (one
child process per GPU, before anything in the parent touches the GPU).

Prints ONE JSON line on rank 0.
  roofline      dominant kernel (the coarse fixed-point solve): algorithmic bytes / launch time from HIP
                events recorded on the engine's own stream; `traffic` (fabric bytes per launch) and
                `valu_issue` come from the rocprofv3 --pmc passes of tools/profile_bench.sh, which
                records the hash of the kernel sources: a file made from other sources is ignored (null)
  cpu_baseline  the reference's own Fortran (oracle/_ref, single core) -- or the C oracle if that library
                is absent -- on a bounded sample of the same workload
  max_abs_err   the parity half of the metric: largest |t_gpu - t_ref| over the receiver times of the
                sampled units (the reference times are the ones cpu_baseline computes), with the count
                beyond the 1e-4 s bar
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NX = 131
NPER = 16
NSRC = 1000
NREC = 32
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
TOL = 1e-4                   # north_star: travel times within 1e-4 s of the reference FMM
KERNEL_SOURCES = ["fim_kernel.hip", "bundle_kernel.hip", "wave_ops.h", "eikonal_core.h", "kernels.h", "engine.hip"]
DISP_SOURCES = ["disp_kernels.hip", "dispersion_core.h"]
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X vector FP64 = half the 157.3 TFLOP/s vector FP32 peak of MI355X_MICROARCH.md


def bytes_per_solve(n):
    """Algorithmic HBM bytes of one solve: fp32 velocity read + fp32 travel-time
    write over the coarse grid, plus the 129x129 refined box (slowness + time)."""
    return 8.0 * n * n + 129 * 129 * 8.0


def kernel_source_hash(names=None):
    h = hashlib.sha256()
    for n in (names or KERNEL_SOURCES):
        with open(os.path.join(ROOT, "dsurftomo_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def reference_times(units, pick, kind):
    """Receiver times of the units `pick` by the reference CPU path -- the reference's own Fortran (oracle/_ref) when that library is there,
    else the C oracle: per unit dicing (the reference re-dices per source, CalSurfG.f90:1186), refined + coarse Fast Marching and the unit's
    receiver times; one core.  Returns (times [len(pick), NREC], seconds, "reference" | "port")."""
    import numpy as np
    import _libs as L
    import synth
    times = np.zeros((len(pick), NREC), np.float32)
    ref = L.ref()
    if ref is not None:
        wb = L.RefWB(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
        t0 = time.perf_counter()
        for k, u in enumerate(pick):
            p = int(units["map_index"][u])
            wb.L.ref_wb_gridder(L.ptr(np.ascontiguousarray(synth.medium(NX, kind, p))))
            wb.L.ref_wb_solve(float(units["scx"][u]), float(units["scz"][u]))
            for r in range(NREC):
                times[k, r] = wb.L.ref_wb_srtimes(float(units["scx"][u]), float(units["scz"][u]),
                                                  float(units["rcx"][u * NREC + r]), float(units["rcz"][u * NREC + r]))
        dt = time.perf_counter() - t0
        wb.close()
        return times, dt, "reference"
    g = L.grid(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
    t0 = time.perf_counter()
    for k, u in enumerate(pick):
        pv = synth.medium(NX, kind, int(units["map_index"][u]))
        veln = L.o_gridder(g, pv)
        sol = L.o_solve(g, pv, veln, units["scx"][u], units["scz"][u])
        for r in range(NREC):
            times[k, r] = L.o_srtimes(g, veln, sol["T"], units["scx"][u], units["scz"][u], units["rcx"][u * NREC + r], units["rcz"][u * NREC + r])
    return times, time.perf_counter() - t0, "port"


def cpu_baseline(budget_units=96):
    """Reference CPU path on a bounded sample of the headline workload.  Returns (record, unit indices, reference receiver times [units, NREC])."""
    import numpy as np
    import synth
    units = synth.units(NX, NSRC, NPER, NREC)
    pick = np.linspace(0, NSRC * NPER - 1, budget_units).astype(int)
    times, dt, kind = reference_times(units, pick, "smooth")
    rec = {"value": round(len(pick) / dt, 4), "unit": "solves/s", "cores": 1, "kind": kind,
           "sample": "%d of the %d (period, source) units, evenly spaced; dicing + refined/coarse FMM + %d receiver times each" % (len(pick), NSRC * NPER, NREC),
           "seconds": round(dt, 2)}
    return rec, pick, times


def exact_secondary(eng, with_reference=True):
    """The tie-prone medium at the headline size: configs[2]'s grid and unit count with configs[4]'s checkerboard, 1000 sources x 16 periods =
    16 000 units x 32 receivers.  exact_ties = 2 -- the reference's Fast Marching replayed on the device (csrc/exact_kernel.hip), all units
    marching at once (the march is bound by the memory system's latency: its rate grows with the units in flight) -- timed with the engine's
    HIP events over the whole call and checked bit for bit against the reference on 16 of the units; the DEFAULT mode, exact_ties = 1 -- fixed
    point, census of its exact ties, the march for the flagged units -- with the flagged fraction and the worst receiver of the units it left
    alone (against the exact_ties = 2 times, which ARE the reference's); and the fixed point alone (exact_ties = 0).  A unit the default leaves
    alone and that ends beyond 1e-4 s is reported in `note` and raises `tolerance_violation` at the top of the line."""
    import numpy as np
    import synth
    nsrc = NSRC
    units = synth.units(NX, nsrc, NPER, NREC, seed=synth.SEED + 41)
    n = nsrc * NPER
    pv = np.stack([synth.medium(NX, "checker", p) for p in range(NPER)])
    out = {"workload": "1025x1025 grid, checkerboard +-8 %% (configs[4]'s medium), %d sources x %d periods = %d units, %d receivers each" % (nsrc, NPER, n, NREC)}
    try:
        eng.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        eng.set_option("exact_ties", 2)
        eng.plan(**units)
        eng.solve()                                   # (first use allocates the marching pool)
        tx = eng.solve().reshape(n, NREC)
        st = eng.stats()
        rec = {"mode": "exact_ties=2: every unit by the literal march, four units per wavefront (k_xmarch)",
               "solves_per_s": round(n / (st["ms_total"] / 1e3), 1), "ms": round(st["ms_total"], 1), "timed_with": "HIP events on the engine's stream, whole call",
               "march_accepts_per_s": round(st["exact_pops"] / (st["ms_exact"] / 1e3), 0)}
        if with_reference:
            pick = np.linspace(0, n - 1, 16).astype(int)
            ref, dt, kind = reference_times(units, pick, "checker")
            rec["not_bit_identical"] = int((tx[pick].view(np.uint32) != ref.view(np.uint32)).sum())
            rec["checked_receiver_times"] = int(ref.size)
            rec["against"] = kind
            rec["vs_one_reference_core"] = round(rec["solves_per_s"] / (len(pick) / dt), 1)
        out["exact"] = rec
        eng.set_option("exact_ties", 1)
        eng.plan(**units)
        eng.solve()                                   # (as for the other modes: the first call allocates the flagged units' marching pool, and follows
        first_ms = eng.stats()["ms_total"]            #  seconds of host work -- the reference sample above -- during which the chip clocks down)
        t1 = eng.solve().reshape(n, NREC)
        st1 = eng.stats()
        flags, infl = eng.unit_ties()
        marched = (flags & 2) != 0
        d = np.abs(t1.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
        left_beyond = int((d[~marched] > TOL).sum())
        out["exact_ties1"] = {"mode": "exact_ties=1 (the default): fixed point + census of its exact ties, literal march for the flagged units (a tie above tie_threshold 2e-5 s; on a map where some unit holds one, every unit holding a tie with any influence)",
                              "tie_prone_maps": int(st1.get("tie_prone_maps", 0)), "flagged_by_their_map": int(st1.get("tie_units_strict", 0)), "flagged_for_the_size_of_their_times": int(st1.get("tie_units_by_scale", 0)),
                              "units_left_alone_holding_a_tie_with_an_influence": int(st1.get("tie_units_tied", 0)),
                              "solves_per_s": round(n / (st1["ms_total"] / 1e3), 1), "ms": round(st1["ms_total"], 1), "ms_march": round(st1["ms_exact"], 1), "ms_first_call": round(first_ms, 1),
                              "flagged_fraction": round(float(marched.mean()), 4), "flagged_units": int(marched.sum()),
                              "flagged_not_bit_identical_to_exact": int((t1[marched].view(np.uint32) != tx[marched].view(np.uint32)).sum()),
                              "unflagged_worst_abs_dt_s": float(d[~marched].max()) if (~marched).any() else 0.0,
                              "unflagged_units_beyond_1e-4_s": left_beyond}
        if left_beyond:
            out["exact_ties1"]["note"] = "the census is a heuristic: %d unit(s) it left to the fixed point end beyond 1e-4 s; exact_ties=2 is the guarantee" % left_beyond
        eng.set_option("exact_ties", 0)
        eng.plan(**units)
        eng.solve()
        t0 = eng.solve().reshape(n, NREC)
        st0 = eng.stats()
        d0 = np.abs(t0.astype(np.float64) - tx.astype(np.float64)).max(axis=1)
        out["exact_ties0"] = {"mode": "exact_ties=0: the fixed point alone (its census reports, nothing is marched)",
                              "solves_per_s": round(n / (st0["ms_total"] / 1e3), 1), "ms": round(st0["ms_total"], 1),
                              "census_flagged_units": int(st0["tie_units"]), "units_beyond_1e-4_s": int((d0 > TOL).sum()), "worst_abs_dt_s": float(d0.max())}
    finally:
        eng.set_option("exact_ties", 1)
    return out


def tie_modes_headline(eng, units, pv):
    """what the default's tie handling costs on the headline medium, where nothing is flagged -- the headline call with the
    fixed point alone and no census (round 4's configuration), with the census reporting only (exact_ties = 0), and the default (exact_ties = 1:
    census + march of the flagged units); HIP events over the whole call, best of two."""
    import synth
    n = len(units["map_index"])
    out = {"workload": "the headline call: %d units" % n}
    try:
        eng.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        for tag, opts in (("fixed_point_no_census", (("exact_ties", 0), ("tie_detect", 0))), ("fixed_point_census_reports", (("exact_ties", 0), ("tie_detect", 1))),
                          ("default_exact_ties1", (("exact_ties", 1), ("tie_detect", 1)))):
            for k, v in opts:
                eng.set_option(k, v)
            eng.plan(**units)
            eng.solve()
            best = None
            for _ in range(2):
                eng.solve()
                st = eng.stats()
                if best is None or st["ms_total"] < best["ms_total"]:
                    best = st
            out[tag] = {"solves_per_s": round(n / (best["ms_total"] / 1e3), 1), "ms": round(best["ms_total"], 2), "ms_coarse_kernels": round(best["ms_fim_coarse"], 2),
                        "census_flagged_units": int(best["tie_units"]), "marched_units": int(best["exact_units"]), "largest_tie_influence_s": float(best.get("tie_influence_max", 0.0))}
        a, b = out["fixed_point_no_census"]["ms"], out["default_exact_ties1"]["ms"]
        out["default_costs_percent"] = round(100.0 * (b - a) / a, 2)
    finally:
        eng.set_option("exact_ties", 1); eng.set_option("tie_detect", 1)
    return out


def headline_all_receivers(eng, units, pv, default_times):
    """The parity half of the metric over the WHOLE headline call, not a sample (round 6): every receiver time of the timed steps' last pass (default
    mode) against exact_ties = 2 on the same call -- the reference's Fast Marching replayed on the device, itself checked bit for bit against the
    reference's Fortran on the sampled units (`parity`, secondary.exact_mode) and by the GPU tests."""
    import numpy as np
    import synth
    n = len(units["map_index"])
    out = {"workload": "the headline call: %d units x %d receivers, default mode against exact_ties = 2" % (n, NREC)}
    try:
        eng.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        eng.set_option("exact_ties", 2)
        eng.plan(**units)
        tx = eng.solve().reshape(n, NREC)
        st = eng.stats()
        t1 = np.asarray(default_times, np.float32).reshape(n, NREC)
        d = np.abs(t1.astype(np.float64) - tx.astype(np.float64))
        out.update({"receiver_times": int(d.size), "beyond_1e-4_s": int((d > TOL).sum()), "units_beyond_1e-4_s": int((d.max(axis=1) > TOL).sum()), "worst_abs_dt_s": float(d.max()),
                    "beyond_5e-5_s": int((d > 0.5 * TOL).sum()), "not_bit_identical": int((t1.view(np.uint32) != tx.view(np.uint32)).sum()),
                    "exact_ties2_solves_per_s": round(n / (st["ms_total"] / 1e3), 1)})
    finally:
        eng.set_option("exact_ties", 1)
    return out


def tolerance_summary(line):
    """ every place this line holds default-mode times checked against the reference -- or against exact_ties = 2, which is the
    reference bit for bit -- and how many of them lie beyond 1e-4 s; `tolerance_violation` at the top of the line is true when any does."""
    checks = {}
    par = line.get("parity")
    if par:
        checks["headline_sample_vs_reference"] = par.get("beyond_1e-4_s")
    sec = line.get("secondary", {})
    h = sec.get("headline_all_receivers", {})
    if "beyond_1e-4_s" in h:
        checks["headline_all_receivers_vs_exact_ties2"] = h["beyond_1e-4_s"]
    x = sec.get("exact_mode", {}).get("exact_ties1", {})
    if "unflagged_units_beyond_1e-4_s" in x:
        checks["checkerboard_1025_default_units_left_alone"] = x["unflagged_units_beyond_1e-4_s"]
        checks["checkerboard_1025_default_marched_units_not_bit_identical"] = x.get("flagged_not_bit_identical_to_exact")
    c4 = sec.get("config4_share", {}).get("exact_ties1_default", {})
    if "vs_reference_sample" in c4:
        checks["config4_share_default_sample_vs_reference"] = c4["vs_reference_sample"].get("beyond_1e-4_s")
    if "units_left_to_the_fixed_point" in c4:
        checks["config4_share_default_units_left_alone"] = c4["units_left_to_the_fixed_point"].get("receiver_times_beyond_1e-4_s")
    return bool(any(v for v in checks.values() if v)), checks


def config4_share(device_index, with_reference=True):
    """one GPU's share of configs[4] -- 4097x4097 grid (nx = ny = 515), checkerboard +-8 %, 128 sources x 24 periods = 3072
    units x 32 receivers -- HIP events over the whole call (engine of its own: the headline's buffers are released first).  A march at this size
    takes a minute whatever the number of units (16.8 M sequential accepts per unit), so the leg runs ONE: the default mode, which flags nearly every
    unit here -- its march IS exact_ties = 2 on those units (same kernels; tests/test_gpu_fullsize.py::test_receivers_at_scale_config4 checks both
    bit for bit against the oracle), and `exact_ties2` reports that march's own rate.  DSA_BENCH_FULL=1 adds the separate exact_ties = 2 call.
    Receiver times against the reference's arithmetic (the C restatement, pinned bit for bit to the reference's Fortran by
    tests/test_oracle_vs_ref.py, one unit per host thread) on an 8-unit sample and on the units the census left to the fixed point."""
    import numpy as np
    import synth
    from dsurftomo_amd.engine import Engine
    nx, nsrc, nper = 515, 128, 24
    n = nsrc * nper
    units = synth.units(nx, nsrc, nper, NREC, seed=synth.SEED + 47)
    pv = np.stack([synth.medium(nx, "checker", p) for p in range(nper)])
    out = {"workload": "configs[4] share of one GPU: 4097x4097 grid, checkerboard +-8 %% (16-vertex squares), %d sources x %d periods = %d units, %d receivers each" % (nsrc, nper, n, NREC)}
    e = Engine(device_index)
    try:
        e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        times, marched = {}, None
        modes = [("exact_ties0", 0), ("exact_ties1_default", 1)] + ([("exact_ties2", 2)] if os.environ.get("DSA_BENCH_FULL") else [])
        for tag, mode in modes:
            e.set_option("exact_ties", mode)
            e.plan(**units)
            if mode == 0:
                e.solve()                              # (allocations; the march's legs are too long to repeat)
            times[tag] = e.solve().reshape(n, NREC)
            st = e.stats()
            rec = {"solves_per_s": round(n / (st["ms_total"] / 1e3), 2), "ms": round(st["ms_total"], 1), "timed_with": "HIP events on the engine's stream, whole call"}
            if mode != 2:
                rec.update({"ms_coarse_kernels": round(st["ms_fim_coarse"], 1), "bundles": "%d x %d" % (int(st["bundles"]), int(st["bundle_size"])), "census_flagged_units": int(st["tie_units"])})
            if mode != 0:
                rec.update({"marched_units": int(st["exact_units"]), "ms_march": round(st["ms_exact"], 1), "units_marching_side_by_side": int(st.get("exact_pool", 0)),
                            "march_accepts_per_s": round(st["exact_pops"] / max(st["ms_exact"], 1e-9) * 1e3, 0)})
            if mode == 1:
                fl, _ = e.unit_ties()
                marched = (fl & 2) != 0
                out["exact_ties2"] = {"solves_per_s": round(float(marched.sum()) / (st["ms_exact"] / 1e3), 2), "marched_units": int(marched.sum()),
                                      "note": "the march of the default-mode call above (exact_ties = 2 runs these kernels on every unit): its units over its own time; "
                                              "DSA_BENCH_FULL=1 times the separate exact_ties = 2 call"}
            out[tag] = {**out.get(tag, {}), **rec} if tag == "exact_ties2" else rec
        d = np.abs(times["exact_ties0"].astype(np.float64) - times["exact_ties1_default"].astype(np.float64))[marched]
        out["exact_ties0"]["vs_the_march_on_the_marched_units"] = {"receiver_times_beyond_1e-4_s": int((d > TOL).sum()), "of": int(d.size), "worst_abs_dt_s": float(d.max()) if d.size else 0.0}
        if with_reference:
            import _libs as L
            from concurrent.futures import ThreadPoolExecutor
            left = np.nonzero(~marched)[0][:8]
            pick = np.unique(np.concatenate([np.linspace(0, n - 1, 8).astype(int), left]))
            g = L.grid(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, 8)
            veln = {p: L.o_gridder(g, pv[p]) for p in sorted(set(int(units["map_index"][k]) for k in pick))}

            def one(k):
                p = int(units["map_index"][k])
                o = L.o_solve(g, pv[p], veln[p], units["scx"][k], units["scz"][k])
                return np.array([L.o_srtimes(g, veln[p], o["T"], units["scx"][k], units["scz"][k], units["rcx"][k * NREC + r], units["rcz"][k * NREC + r]) for r in range(NREC)], np.float32)

            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
                ref = np.stack(list(ex.map(one, pick)))
            out["reference_sample"] = {"units": int(len(pick)), "of_which_left_to_the_fixed_point_by_the_default": int(len(left)), "receiver_times": int(ref.size),
                                       "against": "port (C restatement pinned to the reference's Fortran)", "seconds": round(time.perf_counter() - t0, 1)}
            for tag in times:
                dd = np.abs(times[tag][pick].astype(np.float64) - ref.astype(np.float64))
                out[tag]["vs_reference_sample"] = {"beyond_1e-4_s": int((dd > TOL).sum()), "not_bit_identical": int((times[tag][pick].view(np.uint32) != ref.view(np.uint32)).sum()), "worst_abs_dt_s": float(dd.max())}
            if len(left):
                sel = np.isin(pick, left)
                dl = np.abs(times["exact_ties1_default"][pick][sel].astype(np.float64) - ref[sel].astype(np.float64))
                out["exact_ties1_default"]["units_left_to_the_fixed_point"] = {"units": int((~marched).sum()), "checked": int(len(left)), "receiver_times_beyond_1e-4_s": int((dl > TOL).sum()), "worst_abs_dt_s": float(dl.max())}
                if (dl > TOL).any():
                    out["exact_ties1_default"]["note"] = "the census is a heuristic: a unit it left to the fixed point ends beyond 1e-4 s; exact_ties=2 is the guarantee"
    finally:
        e.close()
    return out


def bundling_secondary(eng):
    """What the headline's bundles are worth when the periods' maps have nothing in common: the same grid and sizes with
    +-10 % random vertices, a different draw per period, the headline's 1 000 sources x 16 periods (round 5; 256 sources before: a call that
    small runs 256 lone wide bundles and gives 20 500 solves/s), bundled as the engine chooses and unit by unit."""
    import numpy as np
    import synth
    nsrc = NSRC
    units = synth.units(NX, nsrc, NPER, NREC, seed=synth.SEED + 43)
    n = nsrc * NPER
    pv = np.stack([synth.medium(NX, "rough", p) for p in range(NPER)])
    out = {"workload": "1025x1025 grid, +-10 %% random vertices, an unrelated draw per period; %d sources x %d periods = %d units; exact_ties = 0 (the fixed-point kernels are the subject)" % (nsrc, NPER, n)}
    try:
        eng.set_option("exact_ties", 0)
        eng.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        for tag, opt in (("bundled", 1), ("unit_by_unit", 0)):
            eng.set_option("bundle", opt)
            eng.plan(**units)
            eng.solve()
            eng.solve()
            st = eng.stats()
            out[tag] = {"solves_per_s": round(n / (st["ms_total"] / 1e3), 1), "bundle_size": int(st.get("bundle_size", 0)),
                        "evals_per_node": round(st["evals_total"] / n / (eng.nnx * eng.nnz), 3), "marched_units": int(st["exact_units"])}
            if tag == "bundled" and not st.get("bundle_size", 0):
                out[tag]["note"] = "the engine did not bundle this call (memory budget / bundles that did not converge): this figure is a unit-by-unit solve"
    finally:
        eng.set_option("bundle", 1)
        eng.set_option("exact_ties", 1)
    return out


def rays_secondary(eng):
    """Rays and Frechet rows at the headline grid (SURVEY.md 8d secondary metric; reference rpaths CalSurfG.f90:1771-2318 and the row assembly
    :1383-1432): 256 sources x 32 receivers on the smooth map of period 0, synthetic depth kernels (nz = 9), rows left on the device."""
    import numpy as np
    import synth
    nsrc, nz = 256, 9
    u = synth.units(NX, nsrc, 1, NREC)
    pv = synth.medium(NX, "smooth", 0)
    ncol = NX * NX
    rng = synth.LCG(5)
    vel = (2.5 + 0.2 * np.arange(nz)[:, None, None] + 0.0 * np.zeros((nz, NX, NX))).astype(np.float32)
    depz = (np.arange(nz) * 5.0).astype(np.float32)
    sen = [0.02 + 0.05 * rng.uniform(nz * ncol).reshape(nz, 1, ncol) for _ in range(3)]
    try:
        eng.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
        eng.set_depth_kernels(vel, depz, *sen)
        eng.set_option("rows_on_device", 1)
        eng.plan(u["map_index"], u["scx"], u["scz"], u["nrec"], u["rcx"], u["rcz"])
        eng.solve_rows_device()
        eng.solve_rows_device()
        st = eng.stats()
    finally:
        eng.set_option("rows_on_device", 0)
    rays, nar = st["rays"], st["nar"]
    return {"kernel": "k_rays (four lanes per ray: launches of up to 81 920 rays; one lane per ray beyond) + k_row_list / k_row_emit / k_scan",
            "workload": "%d sources x %d receivers at 1025^2, smooth map, nz = %d depth layers; rows left on the device" % (nsrc, NREC, nz),
            "rays": int(rays), "matrix_entries": int(nar), "entries_per_ray": round(nar / max(rays, 1), 1), "steps_per_ray": round(st["ray_steps"] / max(rays, 1), 1),
            "ms_rays": round(st["ms_rays"], 2), "ms_rows": round(st["ms_rows"], 2),
            "rays_per_s": round(rays / (st["ms_rays"] / 1e3), 0), "rays_per_s_incl_rows": round(rays / ((st["ms_rays"] + st["ms_rows"]) / 1e3), 0),
            "entries_per_s": round(nar / ((st["ms_rays"] + st["ms_rows"]) / 1e3), 0)}


def pmc_record():
    """Counters of the dominant kernel from tools/profile_bench.sh (separate rocprofv3 --pmc passes of this
    very command); ignored unless they were taken from the kernel sources that are in the tree now."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    return rec if rec.get("kernel_source_hash") == kernel_source_hash() else None


def dispersion_secondary(eng):
    """Secondary kernel: the dispersion stage at the headline model size, bounded by FP64 vector / transcendental
    throughput, not by HBM.  Timed with the engine's HIP events; flops per root come from the FP64 instruction counters of
    tools/collect_pmc.sh (profiles/pmc_dispersion.json), ignored when taken from other kernel sources."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import disp_roofline
    rec = disp_roofline.run(eng, reps=2)
    out = {"kernel": "k_dispersion + k_depth_kernels (surfdisp96.f:223-305, 807-843 under CalSurfG.f90:1-169)",
           "workload": "nx=ny=131, nz=9, 16 Rayleigh phase periods with depth kernels: 17161 columns x 55 models x 16 roots",
           "roots": rec["roots"], "ms": rec["ms"], "roots_per_s": rec["roots_per_s"],
           "roofline": {"bound": "fp64 vector", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "achieved": None, "frac": None, "flops_per_root": None}}
    path = os.path.join(ROOT, "profiles", "pmc_dispersion.json")
    if os.path.exists(path):
        with open(path) as f:
            pmc = json.load(f)
        if pmc.get("kernel_source_hash") == kernel_source_hash(DISP_SOURCES) and pmc.get("fp64_flops_per_root"):
            ach = pmc["fp64_flops_per_root"] * rec["roots_per_s"] / 1e12
            out["roofline"].update({"flops_per_root": round(pmc["fp64_flops_per_root"], 1), "transcendentals_per_root": pmc.get("fp64_transcendentals_per_root"),
                                    "achieved": round(ach, 3), "frac": round(ach / FP64_VECTOR_PEAK_TFLOPS, 4)})
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` typed as is: start the N ranks as child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, as torch.distributed.run would), pass rank 0's line through, and exit with
    the worst exit code.  The parent never touches the GPU and never replaces itself with another program."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        if args.no_secondary:
            cmd.append("--no-secondary")
        procs.append(subprocess.Popen(cmd, env=env))
    # poll: when one rank dies (build error, bad device, out of memory) the others would sit in the rendezvous or the
    # collective until the backend's timeout -- end them and fail at once; an overall limit covers a silent hang
    rc, t0, limit = 0, time.time(), float(os.environ.get("DSA_BENCH_TIMEOUT", "3600"))
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            rc = max(rc, abs(r))
        if live and (rc != 0 or time.time() - t0 > limit):
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            sys.exit(rc or 124)
        if live:
            time.sleep(0.2)
    sys.exit(rc)


SETTLE_STEPS = int(os.environ.get("DSA_BENCH_SETTLE", "6"))       # (tools/profile_bench.sh: 0 for the counter passes, which count one launch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only (the counter passes of tools/profile_bench.sh)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import numpy as np
    import torch
    import synth
    from dsurftomo_amd import build, sharding
    from dsurftomo_amd.engine import Engine

    # one rank per GPU over RCCL; with fewer devices than ranks (a 1-GPU box rehearsing the N-rank path) the
    # ranks share devices and the exchange runs over gloo with host tensors -- same partition, same collective
    ndev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % ndev
    shared = world > ndev
    dist = None
    # DSA_BENCH_FORCE_DIST=1: run the N-rank code path (RCCL process group, dsa_solve_device, device-to-device all-gather) with ONE rank --
    # the only way to put that path on hardware where a single GPU is offered (tools/rccl_check.py, profiles/r03_bench_rccl_one_rank.log)
    collective = world > 1 or bool(os.environ.get("DSA_BENCH_FORCE_DIST"))
    if collective:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()

    # ---- workload, sharded by SOURCES: every period of a source on the same rank, so that the engine can solve them side by side
    # (bundles); a rank's units are a strided subset of the reference's (period, source) order (sharding.source_shard)
    units = synth.units(NX, NSRC, NPER, NREC)
    total_units = NSRC * NPER
    sl = sharding.source_shard(NSRC, NPER, world, rank)
    rsl = sharding.ray_positions(units["nrec"], sl)
    lo, hi = 0, int(sl.size)                                      # (units of this rank)
    pv = np.stack([synth.medium(NX, "smooth", p) for p in range(NPER)])

    eng = Engine(device_index)
    if os.environ.get("DSA_MAX_CHUNK"):
        eng.set_option("max_chunk", int(os.environ["DSA_MAX_CHUNK"]))
    if os.environ.get("DSA_MEM_BUDGET_GB"):
        eng.set_memory_budget(int(float(os.environ["DSA_MEM_BUDGET_GB"]) * 1e9))
    elif shared:
        eng.set_memory_budget(int(200e9 / ((world + ndev - 1) // ndev)))
    if os.environ.get("DSA_BUNDLE"):                      # 0 = unit by unit (the dominant kernel of rounds 1-2), 4 / 8 / 16 = that bundle size
        eng.set_option("bundle", int(os.environ["DSA_BUNDLE"]))
    if os.environ.get("DSA_FIM_SORTED"):
        eng.set_option("fim_sorted", int(os.environ["DSA_FIM_SORTED"]))
    t_setup = time.perf_counter()
    eng.set_maps(NX, NX, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    eng.plan(units["map_index"][sl], units["scx"][sl], units["scz"][sl], units["nrec"][sl], units["rcx"][rsl], units["rcz"][rsl])
    setup_ms = 1000.0 * (time.perf_counter() - t_setup)      # host buffers -> HBM: maps, dicing, descriptors
    n = eng.nnx

    dev = torch.device("cpu") if shared else torch.device("cuda", device_index)
    counts, order_np = sharding.gather_order(units["nrec"], NSRC, NPER, world)
    order = torch.from_numpy(order_np).to(dev)

    mine = torch.empty(eng.ndata, dtype=torch.float32, device=dev) if (collective and not shared) else None

    def step():
        if collective and not shared:
            # the path's one exchange step, device to device: the rank's receiver times stay in HBM (dsa_solve_device) and the RCCL
            # all-gather completes the vector on every rank; nothing visits the host before the collective
            eng.solve_device(mine.data_ptr())
            return sharding.all_gather_ordered(dist, mine, counts, order)
        t = eng.solve()
        if collective:                                 # ranks sharing a device (1-GPU rehearsal): gloo, host tensors
            return sharding.all_gather_ordered(dist, torch.from_numpy(t).to(dev), counts, order)
        return t

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # settling passes, untimed and on top of the W warm-up steps: the first seconds of load after the idle set-up phase (imports, synthetic
    # data, plan) ran the same kernels up to 8 % slower on some boxes (profiles/README_r05.md: r05_bench_final.json against its own later legs)
    for _ in range(SETTLE_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    acc = {"ms_fim_coarse": 0.0, "launches_fim_coarse": 0.0, "ms_total": 0.0, "ms_fim_refined": 0.0, "ms_stages": 0.0,
           "evals_total": 0.0, "rounds_max": 0.0}
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
        st = eng.stats()
        for k in acc:
            acc[k] = max(acc[k], st[k]) if k == "rounds_max" else acc[k] + st[k]
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the other ranks are done: their engines' memory goes back before rank 0 runs the secondary legs (ranks sharing a device -- the 1-GPU rehearsal --
    # had a memory budget each; rank 0's is lifted)
    if dist is not None:
        if rank != 0:
            eng.close()
        dist.barrier()
        if rank == 0 and shared and not os.environ.get("DSA_MEM_BUDGET_GB"):
            eng.set_memory_budget(0)
    if rank == 0:
        solves = total_units * args.steps
        bps = bytes_per_solve(n)
        my_units = (hi - lo) * args.steps
        launches = max(acc["launches_fim_coarse"], 1)
        kernel_s = acc["ms_fim_coarse"] / 1000.0
        achieved = my_units * bps / kernel_s / 1e9 if kernel_s > 0 else 0.0
        pmc = pmc_record()
        traffic = valu = None
        traffic_source = None
        if pmc:
            traffic_source = {"file": "profiles/pmc_latest.json", "kernel_source_hash": pmc.get("kernel_source_hash"), "tree_hash": kernel_source_hash(),
                              "collected_by": "tools/profile_bench.sh (rocprofv3 --pmc passes of this command)"}
            if pmc.get("fabric_bytes_per_solve"):
                traffic = round(pmc["fabric_bytes_per_solve"] * my_units / launches)
            valu = pmc.get("valu_issue")
        line = {
            "metric": "source-period FMM solves/sec on NxN grid; travel-time max-abs-err vs ref",
            "value": round(solves / dt, 2), "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS,
            "ms_per_step": round(1000.0 * dt / args.steps, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: 1025x1025 grid (nx=ny=131, dicing 8), 16 periods x 1000 sources, 32 receivers each, smooth +-10% velocity",
                       "medium_note": "the 16 maps are one pattern times a scale per period (BASELINE configs[2]): the best case for the bundle kernel, which solves the periods of a "
                                      "source under one shared schedule; secondary.bundling_on_unrelated_maps holds the figure for maps that have nothing in common",
                       "grid": n, "units_per_step": total_units, "receivers_per_step": total_units * NREC,
                       "parallelism": "sources (all their periods) sharded over %d rank(s) on %d GPU(s), %s all-gather of receiver times" % (world, min(world, ndev), "gloo (shared devices)" if shared else "RCCL")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": (("k_fim_bundle<%d,...> (coarse fixed-point solve, the %d periods of a source per workgroup" % (int(st.get("bundle_size", 0)), int(st.get("bundle_size", 0)))
                                     + ("; the last bundles of the launch are cut in halves and run as k_fim_bundle<%d,...> on a second stream beside the whole ones: "
                                        "avg_launch_ms spans the pair, as the counters of profiles/pmc_latest.json do)" % (int(st.get("bundle_size", 0)) // 2)
                                        if st.get("bundles", 0) * st.get("bundle_size", 0) > st.get("bundled_units", 0) + 0.5 * st.get("bundle_size", 0) else ")"))
                                    if st.get("bundle_size", 0) else "k_fim_sorted<256, compact> (coarse fixed-point solve)"), "bytes_per_solve": bps,
                         "launches": int(acc["launches_fim_coarse"]),
                         "avg_launch_ms": round(acc["ms_fim_coarse"] / launches, 3),
                         "solves_per_launch": round(my_units / launches, 1),
                         "valu_issue": valu,
                         "note": "achieved = algorithmic bytes / kernel launch time (HIP events on the engine's stream). `traffic` = bytes at the L2's memory side per "
                                 "launch (FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits included): an order of magnitude above the algorithmic bytes -- every "
                                 "evaluation of a node re-fetches its neighbourhood, and 768 fronts in flight leave no cache level a band to hold -- and close to what "
                                 "the fabric delivers while the kernel runs (DESIGN.md 'What binds the bundle kernel': the two-point experiment); valu_issue (share of "
                                 "the SIMDs' issue cycles that carried a VALU instruction) is the other co-limiter. traffic / valu_issue are null when "
                                 "profiles/pmc_latest.json was not taken from the kernel sources in the tree"},
            "kernel_ms_per_step": {"fim_coarse": round(acc["ms_fim_coarse"] / args.steps, 2), "fim_refined": round(acc["ms_fim_refined"] / args.steps, 2),
                                   "stages": round(acc["ms_stages"] / args.steps, 2)},
            "evals_per_node": round(acc["evals_total"] / max(my_units, 1) / (n * n), 3),
            "setup_ms": round(setup_ms, 1),
            "bundles": {"members": int(st.get("bundle_size", 0)), "per_step": int(st.get("bundles", 0)), "field_slots": int(st.get("bundle_slots", 0))},
            "field_slots": int(st.get("field_slots", 0)), "footprint_mb": round(st.get("footprint_mb", 0.0), 1),     # coarse field slots of the launch (recycled when fewer than the units), HBM held by the solve
            "value_incl_setup": round(solves / (dt + setup_ms / 1000.0), 2),
            "max_abs_err": None,
            "tie_handling": {"mode": "exact_ties=1 (default): fixed point + census of its exact ties + the reference's march for the flagged units -- a unit holding a tie whose "
                                     "influence exceeds tie_threshold (2e-5 s), and (tie_map_strict) every unit holding a tie with any influence on a map where some unit holds such a tie",
                             "census_flagged_units_last_step": int(st.get("tie_units", 0)), "marched_units_last_step": int(st.get("exact_units", 0)),
                             "tie_prone_maps_last_step": int(st.get("tie_prone_maps", 0)),
                             "marched_for_the_size_of_their_times_last_step": int(st.get("tie_units_by_scale", 0)),
                             "units_left_to_the_fixed_point_holding_a_tie_with_an_influence": int(st.get("tie_units_tied", 0)),
                             "largest_tie_influence_s": float(st.get("tie_influence_max", 0.0)),
                             "note": "no tie on these maps reaches tie_threshold (largest: one or two ulps of the travel time), so nothing is marched; the units that hold such "
                                     "small ties stay with the fixed point, whose times are the reference's to 1e-4 s by measurement, not by construction: "
                                     "secondary.headline_all_receivers checks every receiver time of the call against exact_ties = 2.  The measurement behind it is an envelope -- at "
                                     "most 26 ulps of the travel time at a receiver on grids up to 1025^2 in all but about one unit in 400 000 (36 ulps seen), i.e. within 1e-4 s while the times stay below 64 s, as this call's do; a unit "
                                     "that holds a tie and lies outside (longer paths, larger grids) is marched (option tie_scale_guard; DESIGN.md 'Ties')"},
        }
        if not args.no_cpu_baseline:
            # N = 1: the bounded CPU baseline (its times are the parity reference).  N > 1: no baseline line (contract), but the
            # gathered vector is still checked against the reference on a smaller sample of units
            rec, pick, ref_times = cpu_baseline(64 if world == 1 else 16)
            if world == 1:
                line["cpu_baseline"] = rec
                line["vs_baseline"] = round(line["value"] / rec["value"], 1)
                line["vs_baseline_note"] = "value / cpu_baseline.value: the %s on ONE host core of this box (%d cores present); BASELINE.md holds no published number" % (
                    "reference's own Fortran" if rec["kind"] == "reference" else "C restatement of the reference", os.cpu_count() or 0)
            last_host = last.detach().cpu().numpy() if hasattr(last, "detach") else np.asarray(last, np.float32)
            got = np.asarray(last_host, np.float32).reshape(total_units, NREC)[pick]
            d = np.abs(got.astype(np.float64) - ref_times.astype(np.float64))
            line["max_abs_err"] = float(d.max())
            line["parity"] = {"checked_receiver_times": int(d.size), "units": int(len(pick)), "beyond_1e-4_s": int((d > TOL).sum()),
                              "not_bit_identical": int((got.view(np.uint32) != ref_times.view(np.uint32)).sum()),
                              "against": rec["kind"], "tolerance_s": TOL}
        # secondary legs (rank 0, after the timed region; none of them may take the headline line down with it)
        line["secondary"] = {}
        full_units = dict(map_index=units["map_index"], scx=units["scx"], scz=units["scz"], nrec=units["nrec"], rcx=units["rcx"], rcz=units["rcz"])
        last_host_all = last.detach().cpu().numpy() if hasattr(last, "detach") else np.asarray(last, np.float32)
        legs = [("headline_all_receivers", lambda: headline_all_receivers(eng, full_units, pv, last_host_all)),
                ("tie_modes_headline", lambda: tie_modes_headline(eng, full_units, pv)),
                ("exact_mode", lambda: exact_secondary(eng, with_reference=not args.no_cpu_baseline)), ("bundling_on_unrelated_maps", lambda: bundling_secondary(eng)),
                ("rays", lambda: rays_secondary(eng)), ("dispersion", lambda: dispersion_secondary(eng)),
                ("config4_share", lambda: (eng.close(), config4_share(device_index, with_reference=not args.no_cpu_baseline))[1])]
        if world > 1:          # (the other ranks wait in the closing barrier: the two long legs -- 16 000 and 3 072 marching units -- belong to the N = 1 line)
            legs = [l for l in legs if l[0] not in ("exact_mode", "config4_share", "headline_all_receivers")]
            line["secondary"]["note"] = "N > 1: headline_all_receivers, exact_mode and config4_share are legs of the N = 1 line"
        for name, leg in legs:
            if args.no_secondary:
                break
            try:
                line["secondary"][name] = leg()
            except Exception as ex:
                line["secondary"][name] = {"error": str(ex)[:300]}
        viol, checks = tolerance_summary(line)
        line = {**{k: line[k] for k in ("metric", "value", "unit")}, "tolerance_violation": viol, "tolerance_checks": checks, **{k: v for k, v in line.items() if k not in ("metric", "value", "unit")}}
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
