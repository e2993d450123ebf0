"""`import nvstrings` -- the reference's module name (python/nvstrings.py) for this back-end: the
module lives in the custrings_amd package (custrings_amd/nvstrings.py, one C-ABI call per method); the
CPython glue module pyniNVStrings built from custrings_amd/host/pyni_strings.cpp is its n_* counterpart."""
from custrings_amd.nvstrings import *  # noqa: F401,F403
from custrings_amd.nvstrings import nvstrings, to_device, from_strings, from_offsets, from_offsets64, free, bind_cpointer  # noqa: F401
