"""The reference's Taipei example (tests/golden/taipei/) through the package's format readers."""
from dsurftomo_amd.io import HERE, load  # noqa: F401
