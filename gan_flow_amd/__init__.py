"""Import shim: the product package lives in `gan-flow_amd/` (not an importable name); this makes it
importable as `gan_flow_amd` by putting that directory on the package path."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gan-flow_amd"))

from ._native import CnfError, library_path, load_library  # noqa: E402,F401
