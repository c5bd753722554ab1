"""Build the HIP shared library in-tree (dsurftomo_amd/libdsurftomo_amd.so) for gfx950.

    python -m dsurftomo_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract
(csrc/eikonal_core.h): the travel-time arithmetic must not be fused.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdsurftomo_amd.so")
SOURCES = ["fim_kernel.hip", "bundle_kernel.hip", "exact_kernel.hip", "stage_kernels.hip", "ray_kernels.hip", "disp_kernels.hip", "spmv.hip", "lsmr.hip", "iteration.hip", "engine.hip", "dropin.hip", "selfcheck.hip"]
HEADERS = ["eikonal_core.h", "source_stage.h", "host_geometry.h", "kernels.h", "engine.h", "ray_core.h", "dispersion_core.h", "spmv_state.h", "exact_march.h", "wave_ops.h", "receiver_core.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, n) for n in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "dsurftomo_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        extra = ["-D" + d for d in os.environ.get("DSA_DEFINES", "").split() if d]     # experiments only
        extra += os.environ.get("DSA_EXTRA_FLAGS", "").split()                          # experiments only (compiler switches)
        cmd = [hipcc()] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s ==\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
