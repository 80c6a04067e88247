"""guetzli_amd -- MI355X (gfx950) implementation of Guetzli's block-parallel hot path
(block DCT / quantise / IDCT round trip + butteraugli distance map) behind the C ABI of
include/guetzli_amd.h.  See DESIGN.md."""
from .capi import Context, GuetzliAmdError, Library, load  # noqa: F401
from .encoder import HostLibrary, load_host, process, process_png, read_png  # noqa: F401

__all__ = ["Context", "GuetzliAmdError", "Library", "load", "HostLibrary", "load_host", "process", "process_png", "read_png"]
