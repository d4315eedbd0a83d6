"""augcl_amd -- MI355X-native (gfx950) implementation of AugCL's data-parallel hot path:
pixel augmentations + shared conv encoder + SAC/SVEA/DrQ/AugCL update(), behind the
reference's own Python call surface (`augmentations.aug_to_func`, `algorithms.make_agent`).
All device work goes through the C-ABI library libaugcl_hip.so (include/augcl.h)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
