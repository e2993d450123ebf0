import ctypes as C, os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from custrings_amd import _lib, nvcategory, nvstrings
_lib.ensure_init(0)
bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
print("rccl", bundled, os.path.exists(bundled), flush=True)
rccl = C.CDLL(bundled, mode=C.RTLD_GLOBAL)
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
print("uid", rccl.ncclGetUniqueId(C.byref(uid)), flush=True)
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
print("init", rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0), comm.value, flush=True)
out = C.c_void_p()
_lib.check(_lib.lib.cs_synth_column(4, 0, 50000, 20240607, 3000, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
got = C.c_void_p()
print("calling", flush=True)
rc = _lib.lib.cs_category_build_distributed(col.m_cptr, comm, 1, 0, None, C.byref(got))
print("rc", rc, _lib.last_error(), flush=True)
cat = nvcategory.nvcategory(got.value)
want = nvcategory.from_strings(col)
print("same keys", cat.keys().to_host() == want.keys().to_host(), "same values", list(cat.values()) == list(want.values()), flush=True)
