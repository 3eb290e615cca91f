"""libavif_amd -- MI355X-native execution of libavif's pixel-reformat path.

The product is the C-ABI shared library ``libavif_amd/csrc/libavifhip.so`` (see ``include/avifhip.h``);
this package is the thin Python host side used by the tests, ``bench.py`` and the tile farm:
ctypes mirrors of libavif's boundary structs (``abi``), the library binding (``native``),
device-resident images (``device``), synthetic inputs (``synth``) and the multi-GPU tile farm (``farm``).
"""
from . import abi  # noqa: F401

__all__ = ["abi", "native", "device", "synth", "farm"]
