"""custrings_amd -- MI355X-native columnar string engine behind the custrings API.

`nvstrings`, `nvcategory`, `nvtext` mirror the reference's Python modules for the
hot path; all compute happens in libcustrings_amd.so (HIP, gfx950).  Importing
this package fails if the library has not been built -- there is no fallback.
"""
from . import _lib  # noqa: F401  (loads the shared library or raises)
from . import nvstrings, nvcategory, nvtext  # noqa: F401

__version__ = "0.1.0"
