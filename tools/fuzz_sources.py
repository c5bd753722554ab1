"""Where the fuzzers' sources go (tools/tie_fuzz.py, tools/tie_diagnose.py): environment switches over tests/synth.py's source generator.
   DSA_FUZZ_INNER=f   sources over the fraction f of the grid instead of its inner 90 % (1.0 puts some on the very edge)
   DSA_FUZZ_SNAP=1    a third of the sources exactly on a node line in x, a third in z, some on a node: symmetric fronts, exact ties by construction"""
import os, numpy as np
import synth


def install():
    if os.environ.get("DSA_FUZZ_INNER"):
        _inner = float(os.environ["DSA_FUZZ_INNER"]); _src = synth.sources
        synth.sources = lambda nx, nsrc, gd=8, inner=0.90, seed=synth.SEED: _src(nx, nsrc, gd, _inner, seed)
    if os.environ.get("DSA_FUZZ_SNAP"):
        _src2 = synth.sources
        def _snapped(nx, nsrc, gd=8, inner=0.90, seed=synth.SEED):
            sx, sz = _src2(nx, nsrc, gd, inner, seed)
            gox, goz, dnx, dnz = synth.grid_origin(nx, gd)
            fx = (sx - gox) / dnx; fz = (sz - goz) / dnz
            k = np.arange(nsrc)
            fx = np.where(k % 3 == 0, np.round(fx), fx); fz = np.where(k % 3 != 2, fz, np.round(fz)); fz = np.where(k % 9 == 0, np.round(fz), fz)
            return (gox + fx.astype(np.float32) * dnx).astype(np.float32), (goz + fz.astype(np.float32) * dnz).astype(np.float32)
        synth.sources = _snapped
