"""`import nvcategory` -- the reference's module name (python/nvcategory.py) for this back-end."""
from custrings_amd.nvcategory import *  # noqa: F401,F403
from custrings_amd.nvcategory import nvcategory, to_device, from_offsets, from_strings, from_strings_list, from_categories, bind_cpointer  # noqa: F401
