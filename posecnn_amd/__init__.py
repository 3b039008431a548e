"""posecnn_amd — MI355X (gfx950) native PoseCNN single-frame inference hot path.

`ops` is the drop-in operator surface (same names/arguments as the reference's
lib/networks/network.py layer wrappers) over the C-ABI of libposecnn_hip.so
(include/posecnn_hip.h). See DESIGN.md.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the .so until first use)
