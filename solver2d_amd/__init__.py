"""solver2d_amd -- MI355X-native constraint-solve hot path of erincatto/solver2d.

`solver2d_amd.hip.Solver` is the Python host side of the C-ABI (include/solver2d_amd.h);
`solver2d_amd.wire` holds the wire-format dtypes; `solver2d_amd.synthetic` builds benchmark
snapshots.  Importing the package does not load the HIP library; constructing a Solver does and
raises if the extension or a GPU is missing (no CPU fallback by design).
"""
from . import wire  # noqa: F401

__all__ = ["wire", "hip", "synthetic"]
