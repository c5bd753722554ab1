import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """The measured worst differences behind the parity assertions, whatever the verbosity."""
    import parity_log
    if not parity_log.LINES:
        return
    terminalreporter.section("parity: HIP path vs oracle, measured (bar: 1e-4 s)")
    for line in parity_log.LINES:
        terminalreporter.write_line(line)
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.txt"), "w") as f:
            f.write("\n".join(parity_log.LINES) + "\n")
    except OSError:
        pass


# the product's defaults of the options tests change (csrc/engine.h: kDefaultExactTies, kDefaultTieThreshold); every GPU test starts from them
PRODUCT_DEFAULTS = (("exact_ties", 1), ("tie_threshold", 2e-5), ("tie_detect", 1), ("bundle", 1), ("bundle_pool", 0), ("field_pool", 0), ("max_chunk", 0),
                    ("exact_lds_slots", 0), ("exact_pool", 0), ("bundle_members_per_lane", 0), ("bundle_threads", 0), ("bundle_far_all", 0), ("bundle_tail", 1), ("tie_list", 1), ("exact_tiles", 0), ("exact_tile_cap", 0), ("bundle_refined", 1), ("ray_lanes", 0), ("exact_heap_blocked", 1),
                    ("tie_map_strict", 1), ("tie_scale_guard", 1), ("handoff_replay", 1), ("tie_tolerance", 1e-4), ("tie_sum_threshold", 0), ("tie_count_threshold", 0), ("tie_frozen_bundles", 0))


@pytest.fixture(autouse=True)
def _product_defaults(request):
    """a test that takes the session's engine finds it with the product's defaults, whatever the test before it left behind"""
    if "engine" in request.fixturenames:
        e = request.getfixturevalue("engine")
        for k, v in PRODUCT_DEFAULTS:
            e.set_option(k, v)
    yield


@pytest.fixture(scope="session")
def engine():
    """One engine per session; fails loudly when the HIP library or the GPU is missing."""
    from dsurftomo_amd import build
    from dsurftomo_amd.engine import Engine
    build.build()
    e = Engine(0)
    yield e
    e.close()
