"""moby_amd -- MI355X-native many-worlds contact-dynamics core for Moby's hot path.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of
``include/moby_hip.h``), ``cpp/`` (the C++ adapter a Moby maintainer links) and
thin Python mirrors of the reference's interfaces used by the tests and bench.
"""
from . import _lib  # noqa: F401
from ._lib import MobyHipError, load  # noqa: F401
