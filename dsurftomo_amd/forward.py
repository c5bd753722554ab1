"""Forward-model a DSurfTomo example directory on the GPU.

    python -m dsurftomo_amd.forward <directory with DSurfTomo.in, the data file and MOD> [--out PREFIX]

Reads the reference's input files (dsurftomo_amd/io.py), makes the CalSurfG call through the drop-in
entry (dispersion + depth kernels, one eikonal solve per (period, source), receiver times, rays and
Frechet rows) and writes
    PREFIX.residual.dat   per datum: distance, synthetic time, observed time   (the first three
                          columns of the reference's residualFirst.dat, main.f90:397-403)
    PREFIX.G.npz          the sensitivity matrix as COO (rw, row, col; 1-based) with its shape
There is no CPU path: without a usable GPU this fails with the engine's error text.
"""
import argparse
import sys
import time

import numpy as np

from . import io


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("directory")
    ap.add_argument("--out", default="forward")
    ap.add_argument("--model", default="MOD")
    args = ap.parse_args(argv)
    c = io.load(args.directory, args.model)
    print("model %d x %d x %d, %d period slots, %d data, %d parameters" % (c["nx"], c["ny"], c["nz"], c["kmax"], c["ndata"], c["nparpi"]))
    t0 = time.perf_counter()
    dsyn, rw, row, col = io.call_calsurfg(c)
    dt = time.perf_counter() - t0
    res = c["obst"] - dsyn
    print("CalSurfG on the device: %.3f s; %d matrix entries; residual mean %.1f ms, std %.1f ms" % (dt, rw.size, 1e3 * res.mean(), 1e3 * res.std()))
    np.savetxt(args.out + ".residual.dat", np.column_stack([c["dist"], dsyn, c["obst"]]), fmt="%14.6f")
    np.savez_compressed(args.out + ".G.npz", rw=rw, row=row, col=col, shape=np.array([c["ndata"], c["nparpi"]]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
