"""Row-range sharding across the GPUs of one node (one process per GPU,
`torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm).

Every hot-path op except the category key set is row-local: a rank runs it on
its own row range and results stay sharded -- no collective on the data path.
`split` needs one 4-byte all-reduce(max) so that every shard emits the same
number of columns.  The category build is the one real exchange step
(SURVEY.md section 8e): each rank dictionary-encodes its shard, the ranks
all-gather their (small) sorted key sets, every rank merges them into the
global key set -- the semantics of NVCategory::create_from_categories
(NVCategory.cu:430-514) -- and remaps its local codes.  Output: identical keys on
every rank, values for the rank's own rows, equal to a single-GPU build of the
whole column.

The local work goes through an `ops` object so the communication logic can be
exercised on CPU (gloo) by the tests with a stand-in; the default `GpuOps` is the
HIP library through the C ABI and is the only implementation shipped.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard_range(rows, rank, world):
    """Contiguous row range [lo, hi) of `rank`: equal counts, remainder spread over the first ranks."""
    base, rem = divmod(rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GpuOps:
    """Local operations on the MI355X through libcustrings_amd.so."""

    device = "cuda"

    def __init__(self):
        from . import _lib, nvstrings, nvcategory

        self._lib, self._nvs, self._nvc = _lib, nvstrings, nvcategory
        _lib.ensure_init()

    def category(self, col):
        """col: nvstrings -> (keys as (chars u8, offsets i64, has_null) device tensors, values cat handle)"""
        cat = self._nvc.from_strings(col)
        keys = cat.keys()
        return cat, self.export(keys)

    def export(self, col):
        L = self._lib
        rows = col.size()
        nbytes = int(L.lib.cs_column_nbytes(col.m_cptr))
        chars = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
        offs = torch.zeros(rows + 1, dtype=torch.int64, device="cuda")
        valid = torch.zeros((rows + 7) // 8 + 8, dtype=torch.uint8, device="cuda")
        if rows:
            L.check(L.lib.cs_column_export_offsets64(col.m_cptr, chars.data_ptr(), offs.data_ptr(), valid.data_ptr(), 1, None))
        torch.cuda.synchronize()
        has_null = bool(rows) and not bool(valid[0] & 1)  # the null key, when present, is key 0
        return chars[:nbytes], offs, has_null

    def column(self, chars, offsets, null_first):
        """device tensors -> nvstrings (copy); null_first marks row 0 as null"""
        rows = offsets.numel() - 1
        valid = None
        if null_first:
            bits = np.ones(rows, dtype=np.uint8)
            bits[0] = 0
            valid = torch.from_numpy(np.packbits(bits, bitorder="little")).cuda()
        out = C.c_void_p()
        L = self._lib
        L.check(L.lib.cs_column_from_offsets64(chars.data_ptr() if chars.numel() else None, rows, offsets.data_ptr(),
                                               valid.data_ptr() if valid is not None else None, 1, 1, None, C.byref(out)))
        return self._nvs.nvstrings(out.value)

    def concat_category(self, cols):
        """category of the row-wise concatenation of key columns: (merged keys nvstrings, codes i32 device tensor)"""
        cat = self._nvc.from_strings_list(cols)
        codes = torch.empty(cat.size(), dtype=torch.int32, device="cuda")
        if cat.size():
            self._lib.check(self._lib.lib.cs_category_get_values(cat.m_cptr, codes.data_ptr(), 1, None))
        return cat.keys(), codes

    def remap(self, cat, table):
        """values of `cat` mapped through `table` (i32 device tensor) -> i32 device tensor"""
        n = cat.size()
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        if n:
            L = self._lib
            L.check(L.lib.cs_remap_codes(cat.values_cpointer(), n, table.data_ptr(), out.data_ptr(), None))
        torch.cuda.synchronize()
        return out


def _all_gather_ragged(t, group):
    """all-gather of 1-D tensors of different lengths (padded to the max; two collectives)."""
    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return [p[:s] for p, s in zip(parts, sizes)]


def global_category(local_col, ops=None, group=None):
    """Distributed NVCategory build.  `local_col` holds this rank's row range.
    Returns (keys column -- identical on all ranks, values i32 tensor for the local rows)."""
    ops = ops or GpuOps()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    cat, (chars, offs, has_null) = ops.category(local_col)
    if world == 1:
        return cat.keys(), ops.remap(cat, torch.arange(cat.keys_size(), dtype=torch.int32, device=chars.device))
    rank = dist.get_rank(group)
    flag = torch.tensor([1 if has_null else 0], dtype=torch.int64, device=chars.device)
    all_chars = _all_gather_ragged(chars, group)
    all_offs = _all_gather_ragged(offs, group)
    flags = [torch.zeros_like(flag) for _ in range(world)]
    dist.all_gather(flags, flag, group=group)
    key_cols = [ops.column(c, o, bool(f.item())) for c, o, f in zip(all_chars, all_offs, flags)]
    merged_keys, codes = ops.concat_category(key_cols)
    start = sum(k.size() for k in key_cols[:rank])
    table = codes[start : start + key_cols[rank].size()].contiguous()
    return merged_keys, ops.remap(cat, table)


def agree_on_columns(ncols, device="cuda", group=None):
    """split(): every shard must emit max-over-ranks columns (the one scalar exchanged)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ncols
    t = torch.tensor([ncols], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())
