"""The C-ABI library loads and exports every function include/dsurftomo_amd.h declares.
No compute calls (no GPU here); creating an engine without a GPU must fail loudly, not fall back."""
import ctypes as C
import os
import re

import pytest

import _libs as L

HEADER = os.path.join(L.ROOT, "include", "dsurftomo_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsa_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from dsurftomo_amd import build
    return C.CDLL(build.build())


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("dsa_create", "dsa_destroy", "dsa_set_maps", "dsa_plan", "dsa_solve", "dsa_calsurfg", "dsa_synthetic"):
        assert must in names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in include/dsurftomo_amd.h but not exported: %s" % missing


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    lib.dsa_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    rc = lib.dsa_create(C.byref(h), 0)
    assert rc != 0 and not h.value
    lib.dsa_error_string.restype = C.c_char_p
    lib.dsa_error_string.argtypes = [C.c_void_p]
    msg = lib.dsa_error_string(None).decode()
    assert "no CPU path" in msg or "HIP" in msg


def test_product_does_not_reference_the_oracle():
    """the product sources must not include, link or load anything under oracle/"""
    root = os.path.join(L.ROOT, "dsurftomo_amd")
    for dirpath, _, files in os.walk(root):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".h", ".hip", ".py", ".f90", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "dsurf_oracle" not in text and "oracle/" not in text.replace("the oracle (tests/hostcheck.cpp)", ""), f
