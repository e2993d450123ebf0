"""ctypes binding of libcustrings_amd.so (the C ABI in include/custrings_amd.h).

The library is the product: there is no Python or CPU fallback.  Importing this
module only loads the shared object; the first compute call binds the process to
a GPU (`cs_init`) and raises if none is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("CS_LIB_PATH") or os.path.join(_HERE, "libcustrings_amd.so")  # CS_LIB_PATH: instrumented dev builds

if not os.path.exists(_PATH):
    raise ImportError(
        "custrings_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C custrings_amd/csrc`); there is no fallback implementation" % _PATH
    )

lib = C.CDLL(_PATH)

vp, i32, i64, u64, cp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_char_p
P = C.POINTER


class ColumnView(C.Structure):
    _fields_ = [
        ("chars", vp),
        ("offsets", vp),
        ("validity", vp),
        ("rows", i64),
        ("nbytes", i64),
        ("null_count", i64),
    ]


_PROTOS = {
    "cs_version": (i32, []),
    "cs_has_experiments": (i32, []),
    "cs_last_error": (cp, []),
    "cs_device_count": (i32, []),
    "cs_init": (i32, [i32]),
    "cs_current_device": (i32, []),
    "cs_fallback_count": (i64, []),
    "cs_debug_last_route": (cp, []),
    "cs_config_set": (i32, [cp, cp]),
    "cs_debug_malloc_count": (i64, []),
    "cs_box_rates": (i32, [i64, i32, vp, P(C.c_double)]),
    "cs_pool_cached_bytes": (i64, []),
    "cs_pool_trim": (i32, [i64]),
    "cs_device_bytes_in_use": (i64, []),
    "cs_free": (None, [vp]),
    "cs_column_from_host_strings": (i32, [P(cp), i64, vp, P(vp)]),
    "cs_column_from_offsets32": (i32, [vp, i64, vp, vp, i32, vp, P(vp)]),
    "cs_column_from_offsets64": (i32, [vp, i64, vp, vp, i32, i32, vp, P(vp)]),
    "cs_column_concat": (i32, [P(vp), i32, vp, P(vp)]),
    "cs_column_ipc_export": (i32, [vp, vp]),
    "cs_column_ipc_import": (i32, [vp, P(vp)]),
    "cs_category_ipc_export": (i32, [vp, vp]),
    "cs_category_ipc_import": (i32, [vp, P(vp)]),
    "cs_column_slice": (i32, [vp, i64, i64, vp, P(vp)]),
    "cs_column_destroy": (i32, [vp]),
    "cs_column_rows": (i64, [vp]),
    "cs_column_nbytes": (i64, [vp]),
    "cs_column_offset_width": (i32, [vp]),
    "cs_column_null_count": (i64, [vp]),
    "cs_column_cached_meta": (i32, [vp, vp]),
    "cs_column_get_view": (i32, [vp, P(ColumnView)]),
    "cs_column_export_offsets32": (i32, [vp, vp, vp, vp, i32, vp]),
    "cs_column_export_offsets64": (i32, [vp, vp, vp, vp, i32, vp]),
    "cs_column_byte_count": (i32, [vp, vp, i32, vp, P(i64)]),
    "cs_column_null_bitarray": (i32, [vp, vp, i32, i32, vp, P(i64)]),
    "cs_column_from_index": (i32, [vp, i64, i32, i32, vp, P(vp)]),
    "cs_column_create_index": (i32, [vp, vp, i32, vp]),
    "cs_len": (i32, [vp, vp, i32, vp, P(i64)]),
    "cs_gather": (i32, [vp, vp, i64, i32, vp, P(vp)]),
    "cs_gather_mask": (i32, [vp, vp, i32, vp, P(vp)]),
    "cs_sublist": (i32, [vp, i64, i64, i64, vp, P(vp)]),
    "cs_scatter": (i32, [vp, vp, vp, i32, vp, P(vp)]),
    "cs_scatter_scalar": (i32, [vp, cp, vp, i64, i32, vp, P(vp)]),
    "cs_sort": (i32, [vp, i32, i32, i32, vp, P(vp)]),
    "cs_order": (i32, [vp, i32, i32, i32, vp, i32, vp]),
    "cs_cat": (i32, [vp, P(vp), i32, cp, cp, vp, P(vp)]),
    "cs_join": (i32, [vp, cp, cp, vp, P(vp)]),
    "cs_replace_re_multi": (i32, [vp, P(vp), i32, vp, vp, P(vp)]),
    "cs_split_record": (i32, [vp, cp, i32, vp, i32, vp, P(vp)]),
    "cs_rsplit_record": (i32, [vp, cp, i32, vp, i32, vp, P(vp)]),
    "cs_partition": (i32, [vp, cp, i32, vp, P(vp)]),
    "cs_category_to_strings": (i32, [vp, vp, P(vp)]),
    "cs_category_gather_strings": (i32, [vp, vp, i64, i32, vp, P(vp)]),
    "cs_category_gather": (i32, [vp, vp, i64, i32, vp, P(vp)]),
    "cs_category_gather_and_remap": (i32, [vp, vp, i64, i32, vp, P(vp)]),
    "cs_category_add_strings": (i32, [vp, vp, vp, P(vp)]),
    "cs_category_remove_strings": (i32, [vp, vp, vp, P(vp)]),
    "cs_category_merge_category": (i32, [vp, vp, vp, P(vp)]),
    "cs_category_add_keys": (i32, [vp, vp, vp, P(vp)]),
    "cs_category_remove_keys": (i32, [vp, vp, vp, P(vp)]),
    "cs_category_remove_unused_keys": (i32, [vp, vp, P(vp)]),
    "cs_category_set_keys": (i32, [vp, vp, vp, P(vp)]),
    "cs_token_count": (i32, [vp, cp, vp, i32, vp]),
    "cs_unique_tokens": (i32, [vp, cp, vp, P(vp)]),
    "cs_tokens_counts": (i32, [vp, vp, cp, vp, i32, vp]),
    "cs_replace_tokens": (i32, [vp, vp, vp, cp, vp, P(vp)]),
    "cs_normalize_spaces": (i32, [vp, vp, P(vp)]),
    "cs_lower": (i32, [vp, vp, P(vp)]),
    "cs_upper": (i32, [vp, vp, P(vp)]),
    "cs_strip": (i32, [vp, cp, i32, vp, P(vp)]),
    "cs_find": (i32, [vp, cp, i32, i32, vp, i32, vp, P(i64)]),
    "cs_contains": (i32, [vp, cp, vp, i32, vp, P(i64)]),
    "cs_rfind": (i32, [vp, cp, i32, i32, vp, i32, vp, P(i64)]),
    "cs_find_from": (i32, [vp, cp, vp, vp, i32, vp, i32, vp, P(i64)]),
    "cs_find_multiple": (i32, [vp, vp, vp, i32, vp, P(i64)]),
    "cs_compare": (i32, [vp, cp, vp, i32, vp, P(i64)]),
    "cs_match_strings": (i32, [vp, vp, vp, i32, vp, P(i64)]),
    "cs_startswith": (i32, [vp, cp, vp, i32, vp, P(i64)]),
    "cs_endswith": (i32, [vp, cp, vp, i32, vp, P(i64)]),
    "cs_replace": (i32, [vp, cp, cp, i32, vp, P(vp)]),
    "cs_split": (i32, [vp, cp, i32, vp, P(P(vp)), P(i32)]),
    "cs_rsplit": (i32, [vp, cp, i32, vp, P(P(vp)), P(i32)]),
    "cs_regex_compile": (i32, [cp, P(vp)]),
    "cs_regex_destroy": (i32, [vp]),
    "cs_regex_inst_count": (i32, [vp]),
    "cs_regex_engine": (i32, [vp]),
    "cs_regex_blob": (i32, [vp, P(P(C.c_int32)), P(i32)]),
    "cs_contains_re": (i32, [vp, vp, vp, i32, vp, P(i64)]),
    "cs_match_re": (i32, [vp, vp, vp, i32, vp, P(i64)]),
    "cs_count_re": (i32, [vp, vp, vp, i32, vp, P(i64)]),
    "cs_replace_re": (i32, [vp, vp, cp, i32, vp, P(vp)]),
    "cs_replace_with_backrefs": (i32, [vp, vp, cp, vp, P(vp)]),
    "cs_extract": (i32, [vp, vp, vp, P(P(vp)), P(i32)]),
    "cs_findall": (i32, [vp, vp, vp, P(P(vp)), P(i32)]),
    "cs_records_from_columns": (i32, [P(vp), i32, i32, vp, i32, vp, P(vp)]),
    "cs_category_build": (i32, [vp, vp, P(vp)]),
    "cs_category_merge": (i32, [P(vp), i32, vp, P(vp)]),
    "cs_category_merge_gathered": (i32, [vp, P(vp), i32, i32, vp, P(vp), vp]),
    "cs_category_build_distributed": (i32, [vp, vp, i32, i32, vp, P(vp)]),
    "cs_category_build_distributed_with": (i32, [vp, vp, vp, i32, i32, vp, P(vp)]),
    "cs_category_build_distributed_with2": (i32, [vp, vp, vp, vp, i32, i32, vp, P(vp)]),
    "cs_category_destroy": (i32, [vp]),
    "cs_category_size": (i64, [vp]),
    "cs_category_keys_size": (i64, [vp]),
    "cs_category_keys": (i32, [vp, P(vp)]),
    "cs_category_values_ptr": (vp, [vp]),
    "cs_category_get_values": (i32, [vp, vp, i32, vp]),
    "cs_remap_codes": (i32, [vp, i64, vp, vp, vp]),
    "cs_tokenize": (i32, [vp, cp, vp, P(vp)]),
    "cs_tokenize_multi": (i32, [vp, vp, vp, P(vp)]),
    "cs_ngrams": (i32, [vp, C.c_uint, cp, vp, P(vp)]),
    "cs_synth_column": (i32, [i32, i64, i64, u64, i64, vp, P(vp)]),
    "cs_column_digest": (i32, [vp, vp, P(u64)]),
    "cs_prof_reset": (i32, []),
    "cs_prof_enable": (i32, [i32]),
    "cs_prof_get": (i32, [cp, P(C.c_double), P(i64)]),
    "cs_debug_spin": (i32, [i32, i32, i32, vp]),
    "cs_stream_forget": (i32, [vp]),
}
for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

CS_OK = 0
CS_ERR_INVALID_ARG, CS_ERR_ALLOC, CS_ERR_HIP, CS_ERR_NO_DEVICE, CS_ERR_RANGE, CS_ERR_INTERNAL = 1, 2, 3, 4, 5, 6

_initialised = False


def last_error():
    m = lib.cs_last_error()
    return m.decode("utf8", "replace") if m else ""


def check(status):
    """The reference's Python glue turns every C++ exception into ValueError
    (python/cpp/pystrings.cpp:1912-1931); allocation / device failures surface as
    RuntimeError like std::runtime_error does for direct C++ callers."""
    if status == CS_OK:
        return
    msg = last_error()
    if status in (CS_ERR_INVALID_ARG, CS_ERR_RANGE):
        raise ValueError(msg)
    raise RuntimeError("custrings_amd: %s (status %d)" % (msg, status))


def ensure_init(device=None):
    global _initialised
    if _initialised and device is None:
        return
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("CS_DEVICE") is None else int(os.environ["CS_DEVICE"])
        if lib.cs_device_count() == 1:
            device = 0
    check(lib.cs_init(device))
    _initialised = True


def loaded_hip():
    """The HIP runtime THIS process already has mapped (the one libcustrings_amd.so is bound to -- torch's bundled copy when
    torch was imported first): opened by the path /proc/self/maps shows, never by bare name (a bare `libamdhip64.so` can map
    a second runtime next to it, and two runtimes in one process crash)."""
    path = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    if path is None:
        raise RuntimeError("custrings_amd: no HIP runtime is mapped in this process")
    return C.CDLL(path)


def b(s):
    if s is None:
        return None
    return s.encode("utf8") if isinstance(s, str) else bytes(s)


def addr(x):
    """int address | numpy/buffer object -> (address, keepalive)"""
    if x is None:
        return None, None
    if isinstance(x, int):
        return (x or None), None
    if hasattr(x, "ctypes"):
        return x.ctypes.data, x
    if hasattr(x, "data_ptr"):
        return x.data_ptr(), x
    mv = memoryview(x)
    buf = (C.c_char * mv.nbytes).from_buffer(mv) if not mv.readonly else (C.c_char * mv.nbytes).from_buffer_copy(mv)
    return C.addressof(buf), (buf, mv)
