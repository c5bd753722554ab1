"""dsurftomo_amd: MI355X (gfx950) forward modelling for DSurfTomo's CalSurfG path.

The product is the HIP library `libdsurftomo_amd.so` (csrc/, built by `python -m dsurftomo_amd.build`) behind the C ABI of
include/dsurftomo_amd.h; this package holds its ctypes binding (engine.py), the readers of the reference's input files (io.py),
the drivers built on them (forward.py, invert.py) and the unit sharding of the multi-GPU path (sharding.py)."""
