"""custrings_amd -- MI355X-native columnar string engine behind the custrings API.

`nvstrings`, `nvcategory`, `nvtext` mirror the reference's Python modules for the
hot path; all compute happens in libcustrings_amd.so (HIP, gfx950).  Importing
this package fails if the library has not been built -- there is no fallback.
"""
import os as _os

if not _os.environ.get("CUSTRINGS_AMD_NO_TORCH"):
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  A process must use ONE HIP
    # runtime: when torch is installed load it first, so that this library's
    # NEEDED libamdhip64.so.7 resolves to the copy torch already mapped (loading
    # torch afterwards would map a second runtime next to /opt/rocm's).
    try:
        import torch as _torch  # noqa: F401
    except ImportError:
        pass

from . import _lib  # noqa: F401,E402  (loads the shared library or raises)
from . import nvstrings, nvcategory, nvtext  # noqa: F401,E402

__version__ = "0.1.0"
