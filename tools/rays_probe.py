#!/usr/bin/env python3
"""Secondary metrics at the headline grid: rays/s and Frechet rows/s.

    python tools/rays_probe.py [nx] [nsrc] [nrec] [nz]          (DSA_RAY_LANES=1|4: lanes per ray, default the engine's choice)
Smooth map of bench.py, `nsrc` sources with `nrec` receivers each, synthetic depth kernels."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth           # noqa: E402
from dsurftomo_amd.engine import Engine   # noqa: E402


def main():
    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 131
    nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    nrec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    nz = int(sys.argv[4]) if len(sys.argv) > 4 else 9
    u = synth.units(nx, nsrc, 1, nrec)
    pv = synth.medium(nx, "smooth", 0)
    ncol = nx * nx
    rng = synth.LCG(5)
    vel = (2.5 + 0.2 * np.arange(nz)[:, None, None] + 0.0 * np.zeros((nz, nx, nx))).astype(np.float32)
    depz = (np.arange(nz) * 5.0).astype(np.float32)
    sen = [0.02 + 0.05 * rng.uniform(nz * ncol).reshape(nz, 1, ncol) for _ in range(3)]
    e = Engine(0)
    if os.environ.get("DSA_RAY_LANES"):
        e.set_option("ray_lanes", float(os.environ["DSA_RAY_LANES"]))
    e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, pv)
    e.set_depth_kernels(vel, depz, *sen)
    e.plan(u["map_index"], u["scx"], u["scz"], u["nrec"], u["rcx"], u["rcz"])
    cap = int(nsrc * nrec * (nx * 6) * (nz - 1))
    for k in range(2):
        t0 = time.perf_counter()
        t, rw, iw, col = e.solve_rows(cap)
        dt = time.perf_counter() - t0
        st = e.stats()
        import zlib
        print("pass %d: rows crc %08x %08x; %.3f s wall; rays %d, steps/ray %.0f, ms_rays %.1f, ms_rows %.1f (incl. copy out), nar %d (%.0f per row), ms_fim %.1f" %
              (k, zlib.crc32(np.ascontiguousarray(rw).tobytes()), zlib.crc32(np.ascontiguousarray(col).tobytes()), dt, st["rays"], st["ray_steps"] / max(st["rays"], 1), st["ms_rays"], st["ms_rows"], st["nar"], st["nar"] / max(st["rays"], 1), st["ms_fim_coarse"]))
        print("        rays/s (tracing kernel) %.0f; rays/s incl. row assembly and copy-out %.0f" %
              (st["rays"] / (st["ms_rays"] / 1e3), st["rays"] / ((st["ms_rays"] + st["ms_rows"]) / 1e3)))
    e.close()


if __name__ == "__main__":
    main()
