"""Device LSMR (csrc/lsmr.hip) against the oracle's restatement of the reference's LSMR (pinned bit-exactly to the
reference's own objects in tests/test_oracle_lsmr.py): bit-identical solution, iteration count and estimates."""
import ctypes as C
import time

import numpy as np
import pytest

import _libs as L
import inversion as inv
import synth
from dsurftomo_amd import io as taipei      # the reference's Taipei example (tests/golden/taipei/) through the package's format readers
from dsurftomo_amd.engine import Engine, load_library

pytestmark = pytest.mark.gpu


def system(c, seed=0, threshold0=3.0, weight0=2.0, fwd=None):
    fwd = fwd or L.call_boundary(L.oracle().dso_calsurfg, c)
    r = synth.LCG(77 + seed)
    obst = (fwd["dsurf"] * (1.0 + 0.04 * (r.uniform(c["ndata"]) - 0.5))).astype(np.float32)
    return inv.build_system(c, fwd, obst, threshold0, weight0)


def device_lsmr(S, damp, device_vectors=0, **kw):
    e = Engine(0)
    try:
        e.set_option("lsmr_device_vectors", device_vectors)
        nar = S["nar"]
        e.spmv_load(S["m"], S["n"], S["rw"], S["iw"][1:nar + 1], S["iw"][nar + 1:])
        return e.lsmr(S["b"], damp, **kw)
    finally:
        e.close()


@pytest.mark.parametrize("device_vectors", [0, 1])
@pytest.mark.parametrize("damp,local_size,itnlim", [(1.0, 10, 400), (0.0, 10, 60), (0.5, 0, 100), (1.0, 3, 7)])
def test_lsmr_boundary_case(damp, local_size, itnlim, device_vectors):
    """both placements of the vectors (ordered reductions on the host / on the device) give the reference's bits"""
    S = system(synth.boundary_case())
    want = inv.call_lsmr(L.oracle().dso_lsmr, S, damp, itnlim=itnlim, local_size=local_size)
    got = device_lsmr(S, damp, device_vectors, itnlim=itnlim, local_size=local_size)
    assert want["itn"] > 3
    assert inv.same(got, want) == []


def test_lsmr_dropin_entry_and_golden():
    """dsa_lsmr_dropin (the reference's argument list, what fortran/lsmr_shim.f90 forwards to) against the vectors the
    reference's LSMR produced (tests/golden/b_lsmr.npz)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "b_lsmr.npz"))
    S = dict(m=int(z["m"]), n=int(z["n"]), iw=z["iw"], rw=z["rw"], b=z["b"])
    lib = load_library()
    got = inv.call_lsmr(lib.dsa_lsmr_dropin, S, 1.0, nout=0)
    assert got["itn"] == int(z["itn"]) and got["istop"] == int(z["istop"])
    assert (got["x"].view(np.uint32) == z["x"].view(np.uint32)).all()
    for k in ("normA", "condA", "normr", "normAr", "normx"):
        assert np.float32(got[k]).view(np.uint32) == np.float32(z[k]).view(np.uint32)


def test_lsmr_degenerate_inputs():
    """b = 0 (the reference leaves at once, x = 0) and a matrix with empty rows / columns"""
    S = system(synth.boundary_case())
    S0 = dict(S); S0["b"] = np.zeros_like(S["b"])
    for dvec in (0, 1):
        got = device_lsmr(S0, 1.0, dvec)
        assert got["itn"] == 0 and got["istop"] == 0 and not got["x"].any()
    keep = (np.arange(S["nar"]) % 3) != 0
    rows, cols = S["iw"][1:S["nar"] + 1][keep], S["iw"][S["nar"] + 1:][keep]
    S1 = dict(S); S1["rw"] = S["rw"][keep]; S1["nar"] = int(keep.sum())
    S1["iw"] = np.concatenate([[S1["nar"]], rows, cols]).astype(np.int32)
    want = inv.call_lsmr(L.oracle().dso_lsmr, S1, 0.3)
    assert inv.same(device_lsmr(S1, 0.3), want) == [] and inv.same(device_lsmr(S1, 0.3, 1), want) == []


def test_lsmr_taipei_iteration():
    """first outer iteration of the reference's Taipei example (main.f90:355-489 with the example's weight / damp /
    threshold): forward call on the device, the system of main.f90:361-466, LSMR on the device == the oracle"""
    c = taipei.load()
    fwd = L.call_boundary(load_library().dsa_calsurfg, c)
    obst = c["obst"]
    S = inv.build_system(c, fwd, obst, 3.0, 4.0)
    t0 = time.time(); want = inv.call_lsmr(L.oracle().dso_lsmr, S, 1.0); t_cpu = time.time() - t0
    e = Engine(0)
    try:
        nar = S["nar"]
        e.spmv_load(S["m"], S["n"], S["rw"], S["iw"][1:nar + 1], S["iw"][nar + 1:])
        e.lsmr(S["b"], 1.0, itnlim=2)                                  # warm-up (allocation, code load)
        t0 = time.time(); got = e.lsmr(S["b"], 1.0); t_gpu = time.time() - t0
        e.set_option("lsmr_device_vectors", 1)
        e.lsmr(S["b"], 1.0, itnlim=2)
        t0 = time.time(); got_d = e.lsmr(S["b"], 1.0); t_dev = time.time() - t0
    finally:
        e.close()
    print("taipei LSMR: m %d n %d nar %d, %d iterations, istop %d | products on the device + ordered sums on the host %.1f ms, "
          "everything on the device %.1f ms, C restatement on one core %.1f ms" %
          (S["m"], S["n"], S["nar"], got["itn"], got["istop"], 1e3 * t_gpu, 1e3 * t_dev, 1e3 * t_cpu))
    assert inv.same(got_d, want) == []
    assert want["itn"] > 10 and np.abs(want["x"]).max() > 0.01
    assert inv.same(got, want) == []


def test_taipei_inversion_driver_known_answer(tmp_path):
    """two outer iterations of the reference's Taipei example through dsurftomo_amd.invert.  Known answer of the
    reference's executable for the first iteration (measured on the reference built with patched
    argument handling): 'mean,std_devs and rms of residual after weighting: -442.8ms 1224.32ms 1.302' and
    'min and max velocity variation -0.1868 0.4053'."""
    import ctypes as C
    from dsurftomo_amd import invert
    lines = []
    vsf, hist = invert.run(taipei.HERE, maxiter=2, out_dir=str(tmp_path), log=lines.append)
    h = hist[0]
    print("\n".join(lines))
    assert abs(h["mean_ms"] - (-442.8)) < 0.06 and abs(h["std_ms"] - 1224.32) < 0.02 and abs(h["rms"] - 1.302) < 6e-4
    assert abs(h["dv_min"] - (-0.1868)) < 6e-5 and abs(h["dv_max"] - 0.4053) < 6e-5
    assert hist[1]["rms"] < 0.8 * h["rms"]                              # the update reduces the misfit
    # the same first iteration assembled from the oracle's pieces gives the same model, bit for bit
    c = taipei.load()
    fwd = L.call_boundary(load_library().dsa_calsurfg, c)
    S = inv.build_system(c, fwd, c["obst"], float(c["threshold0"]), float(c["weight0"]))
    want = inv.call_lsmr(L.oracle().dso_lsmr, S, float(c["damp"]))
    assert want["itn"] == h["itn"] and want["istop"] == h["istop"]
    vs = np.asfortranarray(c["vels"].copy()); dv = want["x"].copy()
    O = L.oracle()
    O.dso_model_update.argtypes = [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_float] * 2
    O.dso_model_update(c["nx"], c["ny"], c["nz"], L.ptr(dv), L.ptr(vs), float(c["minvel"]), float(c["maxvel"]))
    first = np.loadtxt(str(tmp_path / "DSurfTomo.inMeasure.dat.iter001"))
    assert first.shape == (16 * 16 * 8, 4)
    got = first[:, 3].reshape(8, 16, 16).transpose(2, 1, 0)             # [i, j, k]
    assert np.abs(got - vs[1:-1, 1:-1, :-1]).max() < 6e-6                 # f10.5
    assert np.allclose(first[0, :3], [121.35, 25.2, c["depz"][0]], atol=1e-5)
    for name in ("residualFirst.dat", "residualLast.dat", "DSurfTomo.inMeasure.dat", "DSurfTomo.inMeasure.dat.iter002"):
        assert (tmp_path / name).exists()
    assert np.loadtxt(str(tmp_path / "residualFirst.dat")).shape == (c["ndata"], 6)


def test_device_resident_rows_match_the_host_path():
    """One outer iteration of the Taipei example twice: the matrix handed through host arrays like the reference
    (dsa_calsurfg -> dsa_iteration_system -> dsa_lsmr_dropin) and resident on the device from CalSurfG to LSMR
    (dsa_calsurfg with null rw / iw / col -> dsa_iteration_system_device -> dsa_lsmr): every output bit for bit --
    travel times, weights, DWS column sums, right-hand side, LSMR solution and counters, updated model"""
    from dsurftomo_amd import invert
    lib = invert.bind(load_library())
    c = taipei.load()
    obst = np.ascontiguousarray(c["obst"])
    out = []
    for fn in (invert.iteration, invert.iteration_device):
        vsf = np.asfortranarray(c["vels"].copy())
        st = fn(lib, c, vsf, obst, lambda *_: None)
        out.append((st, vsf))
    (a, va), (b, vb) = out
    print("taipei iteration, host rows: forward %.3f s system %.3f s LSMR %.3f s | device rows: forward %.3f s system %.3f s LSMR %.3f s | %d entries" %
          (a["seconds"]["forward"], a["seconds"]["glue"], a["seconds"]["lsmr"], b["seconds"]["forward"], b["seconds"]["glue"], b["seconds"]["lsmr"], b["nar"]))
    assert a["nar"] == b["nar"] and a["m"] == b["m"] and a["itn"] == b["itn"] and a["istop"] == b["istop"]
    for k in ("dsyn", "datweight", "norm", "cbst", "dv"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert a["dws"] == b["dws"]
    assert np.array_equal(va.view(np.uint32), vb.view(np.uint32))
