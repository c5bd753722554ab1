"""Error behaviour of the C ABI on a GPU box: the reference prints and STOPs (CalSurfG.f90:1214-1220,
:1686-1692); the engine returns a status and a message with the reference's words, and refuses calls in
the wrong order instead of computing something."""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import synth
from dsurftomo_amd.engine import Engine, EngineError, load_library

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    e = Engine(0)
    yield e
    e.close()


def small_maps(e, nx=18):
    e.set_maps(nx, nx, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, synth.medium(nx, "smooth"))
    gox, goz, dnx, dnz = synth.grid_origin(nx)
    N = synth.nprop(nx)
    return gox, goz, dnx, dnz, N


def test_source_outside_the_model(eng):
    gox, goz, dnx, dnz, N = small_maps(eng)
    with pytest.raises(EngineError) as ex:
        eng.plan([0], [np.float32(gox - 3 * dnx)], [np.float32(goz + 5 * dnz)], [0], [], [])
    assert ex.value.code == -3 and "Source lies outside bounds of model" in str(ex.value)


def test_receiver_outside_the_model(eng):
    gox, goz, dnx, dnz, N = small_maps(eng)
    with pytest.raises(EngineError) as ex:
        eng.plan([0], [np.float32(gox + 5 * dnx)], [np.float32(goz + 5 * dnz)], [1], [np.float32(gox + (N + 2) * dnx)], [np.float32(goz + dnz)])
    assert ex.value.code == -3 and "Receiver lies outside model" in str(ex.value)


def test_call_order_is_enforced(eng):
    with pytest.raises(EngineError) as ex:
        eng.plan([0], [0.0], [0.0], [0], [], [])
    assert ex.value.code == -5                                  # no maps yet
    gox, goz, dnx, dnz, N = small_maps(eng)
    with pytest.raises(EngineError) as ex:
        eng.solve()
    assert ex.value.code == -5                                  # no plan yet
    eng.plan([0], [np.float32(gox + 5 * dnx)], [np.float32(goz + 5 * dnz)], [1], [np.float32(gox + 9 * dnx)], [np.float32(goz + 3 * dnz)])
    with pytest.raises(EngineError) as ex:
        eng.solve_rows(1000)
    assert ex.value.code == -5 and "depth kernels" in str(ex.value)


def test_bad_arguments(eng):
    with pytest.raises(EngineError) as ex:
        eng.set_option("no_such_option", 1)
    assert ex.value.code == -2
    with pytest.raises(EngineError) as ex:
        eng.set_maps(2, 2, synth.GOXD, synth.GOZD, synth.DVD, synth.DVD, np.ones(4))
    assert ex.value.code == -2
    gox, goz, dnx, dnz, N = small_maps(eng)
    with pytest.raises(EngineError) as ex:
        eng.plan([3], [np.float32(gox + 5 * dnx)], [np.float32(goz + 5 * dnz)], [0], [], [])      # map 3 of 1
    assert ex.value.code == -2


def test_row_capacity_is_checked(eng):
    c = synth.boundary_case(kRc=2, kRg=0, kLc=0, kLg=0)
    vel = np.ascontiguousarray(c["vels"].T)
    eng.dispersion_begin(vel, c["depz"], float(c["minthk"]), 2, 2)
    eng.dispersion_run(2, 0, c["tRc"], True, 0, 0)
    eng.maps_from_dispersion(c["goxd"], c["gozd"], c["dvxd"], c["dvzd"], 8)
    eng.kernels_from_dispersion()
    eng.plan([0], [c["scxf"][0, 0]], [c["sczf"][0, 0]], [2], c["rcxf"][:2, 0, 0], c["rczf"][:2, 0, 0], sen_slot=[0])
    with pytest.raises(EngineError) as ex:
        eng.solve_rows(3)
    assert ex.value.code == -6 and "entries" in str(ex.value)          # DSA_ERR_CAPACITY
    t, rw, iw, col = eng.solve_rows(10000)                       # and the same plan still works with room
    assert rw.size > 3 and set(iw.tolist()) == {1, 2}


def test_dropin_reports_instead_of_stopping():
    lib = load_library()
    c = synth.boundary_case()
    c["kmax"] = c["kmax"] + 1                                    # inconsistent period counts
    with pytest.raises(RuntimeError):
        L.call_boundary(lib.dsa_calsurfg, c)
    assert b"kmax" in lib.dsa_dropin_error()
    c = synth.boundary_case()
    c["scxf"] = c["scxf"].copy(order="F")
    c["scxf"][0, 0] = np.float32(0.1)                            # a source far outside
    with pytest.raises(RuntimeError):
        L.call_boundary(lib.dsa_calsurfg, c)
    assert b"Source lies outside bounds of model" in lib.dsa_dropin_error()
