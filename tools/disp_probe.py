"""Dispersion stage throughput: python tools/disp_probe.py nx nz nper"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth
from dsurftomo_amd.engine import Engine
nx = int(sys.argv[1]); nz = int(sys.argv[2]); nper = int(sys.argv[3])
c = synth.boundary_case(nx=nx, ny=nx, nz=nz, kRc=1, kRg=0, kLc=0, kLg=0, nsrc=1, nrcf=1)
vel = np.ascontiguousarray(c["vels"].T)
t = np.linspace(2.0, 17.0, nper)
e = Engine(0)
if os.environ.get('DSA_DISP_GROUP'): e.set_option('disp_group_shift', int(os.environ['DSA_DISP_GROUP']))
if os.environ.get('DSA_DISP_LDS'): e.set_option('disp_layers_lds', int(os.environ['DSA_DISP_LDS']))
for iwave, igr, kern in ((2, 0, True), (2, 0, False), (2, 1, True), (1, 0, True)):
    e.dispersion_begin(vel, c["depz"], float(c["minthk"]), nper, nper)
    t0 = time.perf_counter(); e.dispersion_run(iwave, igr, t, kern, 0, 0); dt = time.perf_counter() - t0
    st = e.stats()
    roots = st["curves"] * nper * (2 if igr else 1)
    if os.environ.get('DSA_DISP_SAVE') or os.environ.get('DSA_DISP_COMPARE'):      # bit comparison between library builds
        got = e.dispersion_fetch(0, nper, kern, 0)
        blob = np.concatenate([np.ravel(a) for a in (got if kern else (got,))])
        tag = "%d_%d_%d" % (iwave, igr, kern)
        if os.environ.get('DSA_DISP_SAVE'): np.save(os.environ['DSA_DISP_SAVE'] + tag + ".npy", blob)
        else:
            ref = np.load(os.environ['DSA_DISP_COMPARE'] + tag + ".npy")
            print("      against the saved run: identical=%s (%d of %d values differ)" % (np.array_equal(ref.view(np.uint64), blob.view(np.uint64)), int((ref.view(np.uint64) != blob.view(np.uint64)).sum()), blob.size), flush=True)
    print("nx %d nz %d nper %d iwave %d igr %d kernels %d: %.1f ms, %d curves, %.2f M roots/s" % (nx, nz, nper, iwave, igr, kern, 1e3 * dt, st["curves"], roots / dt / 1e6), flush=True)
