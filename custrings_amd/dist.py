"""Row-range sharding across the GPUs of one node (one process per GPU,
`torch.distributed`; backend "nccl" is RCCL over xGMI on ROCm).

Every hot-path op except the category key set is row-local: a rank runs it on
its own row range and results stay sharded -- no collective on the data path.
`split` needs one 4-byte all-reduce(max) so that every shard emits the same
number of columns.  `ngrams` runs over the token column of ALL rows (ngram.cu:32-110), so an
n-gram may begin in one shard and end in the next: every rank receives the first n-1 tokens of
the ranks behind it (`sharded_ngrams`, a few hundred bytes).  The category build is the one real exchange step
(SURVEY.md section 8e): each rank dictionary-encodes its shard, the ranks
all-gather their (small) sorted key sets, every rank merges them into the
global key set -- the semantics of NVCategory::create_from_categories
(NVCategory.cu:430-514) -- and remaps its local codes.  Output: identical keys on
every rank, values for the rank's own rows, equal to a single-GPU build of the
whole column.  When the ranks hold millions of keys in all (K close to N) the merge is partitioned by key range
instead -- splitters from a sample, an all-to-all of every key to its range's owner, an all-gather of the merged
ranges (`_merge_partitioned`): every rank merges 1/G of the keys instead of all of them.

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

    # ---- token columns (sharded_ngrams) ----
    def drop_empty(self, col):
        """the rows create_ngrams keeps (ngram.cu:47-58): not null and not empty; `col` itself when nothing is dropped"""
        L = self._lib
        rows = col.size()
        if rows == 0:
            return col
        lens = torch.empty(rows, dtype=torch.int32, device="cuda")
        total = C.c_int64()
        L.check(L.lib.cs_len(col.m_cptr, lens.data_ptr(), 1, None, C.byref(total)))  # (characters; -1 for null rows)
        mask = lens > 0
        if bool(mask.all()):
            return col
        out = C.c_void_p()
        m8 = mask.to(torch.uint8).contiguous()
        L.check(L.lib.cs_gather_mask(col.m_cptr, m8.data_ptr(), 1, None, C.byref(out)))
        return self._nvs.nvstrings(out.value)

    def head(self, col, k):
        return col.sublist(0, min(k, col.size()))

    def slice(self, col, start, end, step=1):
        """rows start, start + step, ... before end (NVStrings::sublist)"""
        end = min(end, col.size())
        if start >= end:
            return col.sublist(0, 0)
        return col.sublist(start, end, step)

    def concat(self, cols):
        cols = [c for c in cols if c.size()]
        if len(cols) == 1:
            return cols[0]
        arr = (C.c_void_p * max(len(cols), 1))(*[c.m_cptr for c in cols])
        out = C.c_void_p()
        self._lib.check(self._lib.lib.cs_column_concat(arr, len(cols), None, C.byref(out)))
        return self._nvs.nvstrings(out.value)

    def ngrams(self, col, n, sep):
        from . import nvtext

        return nvtext.ngrams(col, n, sep)

    def values(self, cat):
        """the category's int32 codes as a device tensor (a copy: the tensor outlives the handle)"""
        n = cat.size()
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        if n:
            self._lib.check(self._lib.lib.cs_category_get_values(cat.m_cptr, out.data_ptr(), 1, None))
        return out

    def merge_gathered(self, cat, key_cols, rank):
        """cs_category_merge_gathered: the ranks' key sets (columns, in rank order) merged, the local codes remapped"""
        L = self._lib
        arr = (C.c_void_p * len(key_cols))(*[k.m_cptr for k in key_cols])
        out = C.c_void_p()
        n = cat.size()
        values = torch.empty(n, dtype=torch.int32, device="cuda")
        L.check(L.lib.cs_category_merge_gathered(cat.m_cptr, arr, len(key_cols), rank, None, C.byref(out), values.data_ptr()))
        return self._nvs.nvstrings(out.value), values

    def remap(self, cat, table):
        """values of `cat` mapped through `table` (i32 device tensor) -> i32 device tensor"""
        n = cat.size()
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        if n:
            L = self._lib
            L.check(L.lib.cs_remap_codes(cat.values_cpointer(), n, table.data_ptr(), out.data_ptr(), None))
        torch.cuda.synchronize()
        return out


def _all_gather_ragged(tensors, group, scalars=()):
    """All-gather of several 1-D tensors whose lengths differ between ranks, plus a few integers per rank.  One
    collective carries every length and the integers (ONE host read for all of them), then one padded all-gather
    per tensor (RCCL's all-gather wants equal counts; the padding is max - own, small for key sets of similar
    shards).  Returns (per tensor the list of the ranks' parts, per rank the list of its integers)."""
    world = dist.get_world_size(group)
    dev = tensors[0].device
    nt = len(tensors)
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # (several ranks on one GPU, as the two-process GPU test runs: gloo moves host memory)
        parts, scal = _all_gather_ragged([t.cpu() for t in tensors], group, scalars)
        return [[p.to(dev) for p in ps] for ps in parts], scal
    mine = torch.tensor([t.numel() for t in tensors] + [int(v) for v in scalars], dtype=torch.int64, device=dev)
    table = torch.empty(world * mine.numel(), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(table, mine, group=group)
    table = table.view(world, mine.numel()).tolist()  # the one device-to-host read
    out = []
    for j, t in enumerate(tensors):
        m = max(max(row[j] for row in table), 1)
        pad = torch.zeros(m, dtype=t.dtype, device=dev)
        pad[: t.numel()] = t
        parts = torch.empty(world * m, dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(parts, pad, group=group)
        out.append([parts[r * m : r * m + table[r][j]] for r in range(world)])
    return out, [row[nt:] for row in table]


def _all_to_all_ragged(parts, group):
    """parts[d]: the 1-D tensor this rank sends to rank d (one dtype; any lengths).  Returns the list of the tensors
    received, by source rank.  One all-to-all of the counts (one host read), one of the data."""
    world = dist.get_world_size(group)
    dev = parts[0].device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":  # (several ranks on one GPU: gloo moves host memory)
        return [t.to(dev) for t in _all_to_all_ragged([p.cpu() for p in parts], group)]
    send = [int(p.numel()) for p in parts]
    counts = torch.tensor(send, dtype=torch.int64, device=dev)
    got = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(got, counts, group=group)
    recv = got.tolist()
    data = torch.cat([p.reshape(-1) for p in parts]) if sum(send) else torch.empty(0, dtype=parts[0].dtype, device=dev)
    out = torch.empty(sum(recv), dtype=parts[0].dtype, device=dev)
    dist.all_to_all_single(out, data, recv, send, group=group)
    return list(torch.split(out, recv))


# The key-set all-gather makes every rank merge ALL ranks' key sets: fine for the K = 1K / 1M configurations, G times
# redundant when the column's keys are mostly distinct (K close to N).  From this many keys (sum of the ranks' local
# key counts) the merge is PARTITIONED by key range instead (`_merge_partitioned`).
PARTITION_MIN_KEYS = 1 << 21
SPLITTER_SAMPLES_PER_RANK = 64  # per destination rank


def _merge_partitioned(ops, cat, keys_col, chars, offs, has_null, group):
    """Global key set by key RANGES: splitters from a sample of every rank's (sorted) keys cut the key space into one
    range per rank; an all-to-all sends every local key to its range's owner (the local keys are sorted: contiguous
    slices), the owner merges what it receives -- 1/G of the keys instead of all of them -- and returns to every sender
    the position of its keys in the merged range; an all-gather of the merged ranges, in rank order, is the global
    sorted key set on every rank (the same result as NVCategory::create_from_categories on the gathered key sets,
    NVCategory.cu:430-514), and a local key's global code is its range's base plus the returned position."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    K = keys_col.size()
    dev = chars.device
    first = 1 if has_null else 0  # (the null key, key 0 where present, never becomes a splitter: it belongs to range 0)
    want = SPLITTER_SAMPLES_PER_RANK * world
    step = max((K - first) // want, 1)
    sample = ops.slice(keys_col, first, K, step)
    sc, so, _ = ops.export(sample)
    (all_sc, all_so), flags = _all_gather_ragged([sc, so], group, scalars=[1 if has_null else 0])
    null_on = [bool(f[0]) for f in flags]
    sample_cols = [ops.column(c, o, False) for c, o in zip(all_sc, all_so) if o.numel() > 1]
    splitters = None
    if sample_cols:
        pool, _ = ops.concat_category(sample_cols)  # sorted, unique
        M = pool.size()
        stride = max(M // world, 1)
        splitters = ops.head(ops.slice(pool, stride, M, stride), world - 1)
        if splitters.size() == 0:
            splitters = None
    # the range of every local key: the number of splitters <= key, from the codes of (local keys ++ splitters)
    if splitters is not None and K:
        _, codes = ops.concat_category([keys_col, splitters])
        kc, sp = codes[:K].long(), codes[K:].long()
        dest = torch.searchsorted(sp.contiguous(), kc.contiguous(), right=True)
        if has_null:
            dest[0] = 0
    else:
        dest = torch.zeros(K, dtype=torch.int64, device=dev)
    cut = torch.searchsorted(dest.contiguous(), torch.arange(world + 1, dtype=torch.int64, device=dev)).tolist()
    lens = (offs[1:] - offs[:-1]).contiguous()
    byte_at = offs[torch.tensor(cut, dtype=torch.int64, device=dev)].tolist() if K else [0] * (world + 1)
    got_lens = _all_to_all_ragged([lens[cut[d] : cut[d + 1]] for d in range(world)], group)
    got_chars = _all_to_all_ragged([chars[byte_at[d] : byte_at[d + 1]] for d in range(world)], group)
    # this rank's range: merge what the ranks sent (each part sorted and unique)
    cols, counts = [], []
    for r in range(world):
        n = int(got_lens[r].numel())
        counts.append(n)
        if n:
            o = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            o[1:] = torch.cumsum(got_lens[r], 0)
            cols.append(ops.column(got_chars[r].contiguous(), o, rank == 0 and null_on[r]))
    if cols:
        range_keys, range_codes = ops.concat_category(cols)
        rk_chars, rk_offs, rk_null = ops.export(range_keys)
        nrange = range_keys.size()
    else:
        range_codes = torch.empty(0, dtype=torch.int32, device=dev)
        rk_chars, rk_offs, rk_null, nrange = torch.empty(0, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), False, 0
    # positions back to the senders, merged ranges to everybody
    back = _all_to_all_ragged(list(torch.split(range_codes.to(torch.int32), counts)), group)
    (all_c, all_o), sizes = _all_gather_ragged([rk_chars, rk_offs], group, scalars=[nrange, 1 if rk_null else 0])
    nk = [int(z[0]) for z in sizes]
    base, run, byte_run = [], 0, 0
    goffs = [torch.zeros(1, dtype=torch.int64, device=dev)]
    for r in range(world):
        base.append(run)
        run += nk[r]
        if nk[r]:
            goffs.append(all_o[r][1 : nk[r] + 1] + byte_run)
            byte_run += int(all_c[r].numel())
    gchars = torch.cat([c for c, z in zip(all_c, nk) if z]) if run else torch.empty(0, dtype=torch.uint8, device=dev)
    keys = ops.column(gchars.contiguous(), torch.cat(goffs).contiguous(), bool(sizes[0][1]) and nk[0] > 0)
    table = torch.cat([back[d].to(torch.int32) + base[d] for d in range(world)]) if K else torch.empty(0, dtype=torch.int32, device=dev)
    local_rows = cat.size()
    exchanged = int(chars.numel()) + 4 * K + sum(int(c.numel()) for c in all_c) + 8 * sum(int(o.numel()) for o in all_o)
    last_category_exchange.update(partitioned=True, range_keys=nrange, keys_sent=K, keys_received=sum(counts), global_keys=run,
                                  key_bytes_gathered=exchanged, local_keys=K, local_rows=local_rows, keys_per_row=K / max(local_rows, 1),
                                  range_share=nrange * world / max(run, 1))
    import warnings

    if run and nrange * world > 2 * run:  # (the splitters come from a fixed-size sample of every rank's keys)
        warnings.warn("global_category: this rank's key range holds %d of %d keys (%.1f times its share): the sampled splitters "
                      "do not balance this key distribution" % (nrange, run, nrange * world / run))
    if local_rows and K > local_rows // 2:
        warnings.warn("global_category: %d distinct keys in %d local rows -- the exchange of the key sets (%d bytes on this rank) is as "
                      "large as the data; a hash-partitioned exchange of the rows would move less (SURVEY.md section 8e)" % (K, local_rows, exchanged))
    return keys, ops.remap(cat, table.contiguous())


# Reported when the distributed category build stops paying (with K close to N every rank
# ends up holding, and merging, almost the whole column): the exchanged key bytes against the shard's own bytes.
last_category_exchange = {}


def global_category(local_col, ops=None, group=None, partitioned=None):
    """Distributed NVCategory build.  `local_col` holds this rank's row range.
    (`ops` objects other than GpuOps need `slice` / `head` / `concat_category` as well as the basics: the partitioned merge uses them.)
    Returns (keys column -- identical on all ranks, values i32 tensor for the local rows).
    `partitioned`: merge by key ranges (True), by all-gathered key sets (False), or by the ranks' key counts (None:
    partitioned from PARTITION_MIN_KEYS keys in all)."""
    ops = ops or GpuOps()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    cat, (chars, offs, has_null) = ops.category(local_col)
    if world == 1:  # the local codes are the global ones
        if hasattr(ops, "values"):
            return cat.keys(), ops.values(cat)
        return cat.keys(), ops.remap(cat, torch.arange(cat.keys_size(), dtype=torch.int32, device=chars.device))
    rank = dist.get_rank(group)
    if partitioned is None:  # (the same answer on every rank: from the sum of the key counts)
        dev = chars.device if not (chars.device.type == "cuda" and dist.get_backend(group) == "gloo") else torch.device("cpu")
        total = torch.tensor([cat.keys_size()], dtype=torch.int64, device=dev)
        dist.all_reduce(total, group=group)
        partitioned = int(total.item()) >= PARTITION_MIN_KEYS
    last_category_exchange.clear()
    if partitioned:
        return _merge_partitioned(ops, cat, cat.keys(), chars, offs, has_null, group)
    (all_chars, all_offs), flags = _all_gather_ragged([chars, offs], group, scalars=[1 if has_null else 0])
    key_cols = [ops.column(c, o, bool(f[0])) for c, o, f in zip(all_chars, all_offs, flags)]
    gathered = sum(int(c.numel()) + 8 * int(o.numel()) for c, o in zip(all_chars, all_offs))
    local_rows = cat.size()
    last_category_exchange.update(key_bytes_gathered=gathered, local_keys=cat.keys_size(), local_rows=local_rows,
                                  keys_per_row=cat.keys_size() / max(local_rows, 1))
    if local_rows and cat.keys_size() > local_rows // 2:
        import warnings

        warnings.warn("global_category: %d distinct keys in %d local rows -- the key-set all-gather (%d bytes per rank) is as "
                      "large as the data; a hash-partitioned exchange of the rows would move less (SURVEY.md section 8e)"
                      % (cat.keys_size(), local_rows, gathered))
    if hasattr(ops, "merge_gathered"):  # the C ABI's merge step (cs_category_merge_gathered): what a C++ host would call
        return ops.merge_gathered(cat, key_cols, rank)
    merged_keys, codes = ops.concat_category(key_cols)
    start = sum(k.size() for k in key_cols[:rank])
    table = codes[start : start + key_cols[rank].size()].contiguous()
    return merged_keys, ops.remap(cat, table)


def global_category_c_abi(local_col, group=None):
    """The distributed build through the C ABI's own entry point (cs_category_build_distributed_with: local build, key
    sizes / offsets / bytes all-gathered, merge, remap -- all in the library), with torch.distributed as the transport:
    the callback below is what ncclAllGather is to cs_category_build_distributed, which a C++ host calls with its
    ncclComm_t.  Returns an nvcategory whose keys are identical on every rank."""
    import ctypes as C

    from . import _lib, nvcategory

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    hip = _lib.loaded_hip()  # (the runtime already mapped: a bare name may load a second one)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    staged = dist.is_initialized() and dist.get_backend(group) == "gloo"  # (several ranks on one GPU: through host memory)

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    def allgather(_ctx, send, recv, nbytes, stream):
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            dev = torch.device("cpu") if staged else torch.device("cuda", torch.cuda.current_device())
            mine = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if hip.hipMemcpy(mine.data_ptr(), send, nbytes, 2 if staged else 3) != 0:  # (device -> host / device -> device)
                return 1
            parts = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(world)]
            if world > 1:
                dist.all_gather(parts, mine, group=group)
            else:
                parts[0].copy_(mine)
            if not staged:
                torch.cuda.synchronize()
            for r, p in enumerate(parts):
                if hip.hipMemcpy(recv + r * nbytes, p.data_ptr(), nbytes, 1 if staged else 3) != 0:
                    return 1
            return 0
        except Exception:  # (no exception may cross the C frames)
            return 1

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                 C.c_int, C.c_void_p)
    def alltoallv(_ctx, send, send_bytes, send_off, recv, recv_bytes, recv_off, n, stream):
        # (what grouped ncclSend / ncclRecv are to cs_category_build_distributed: the key-range partitioned merge's transport)
        try:
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            dev = torch.device("cpu") if staged else torch.device("cuda", torch.cuda.current_device())
            outs, ins = [], []
            for d in range(n):
                t = torch.empty(send_bytes[d], dtype=torch.uint8, device=dev)
                if send_bytes[d] and hip.hipMemcpy(t.data_ptr(), (send or 0) + send_off[d], send_bytes[d], 2 if staged else 3) != 0:
                    return 1
                outs.append(t)
                ins.append(torch.empty(recv_bytes[d], dtype=torch.uint8, device=dev))
            if world > 1:
                dist.all_to_all(ins, outs, group=group) if not staged else _all_to_all_by_gather(ins, outs, group)
            else:
                ins[0].copy_(outs[0])
            if not staged:
                torch.cuda.synchronize()
            for d in range(n):
                if recv_bytes[d] and hip.hipMemcpy((recv or 0) + recv_off[d], ins[d].data_ptr(), recv_bytes[d], 1 if staged else 3) != 0:
                    return 1
            return 0
        except Exception:  # (no exception may cross the C frames)
            return 1

    out = C.c_void_p()
    ctx = C.c_void_p(1)  # (non-null: run the exchange with one rank too)
    _lib.check(_lib.lib.cs_category_build_distributed_with2(local_col.m_cptr, C.cast(allgather, C.c_void_p), C.cast(alltoallv, C.c_void_p), ctx, world, rank,
                                                            None, C.byref(out)))
    return nvcategory.nvcategory(out.value)


def _all_to_all_by_gather(ins, outs, group):
    """all_to_all for the gloo backend (which has none for tensors of unequal sizes on every build): every rank gathers
    every rank's pieces for it -- test transport only."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    for src in range(world):
        # rank `src` scatters its pieces: piece d goes to rank d
        sizes = torch.tensor([int(t.numel()) for t in outs], dtype=torch.int64) if rank == src else torch.zeros(world, dtype=torch.int64)
        dist.broadcast(sizes, src=src, group=group)
        total = int(sizes.sum())
        flat = torch.cat(outs) if rank == src else torch.empty(total, dtype=torch.uint8)
        if total:
            dist.broadcast(flat, src=src, group=group)
        lo = int(sizes[:rank].sum())
        ins[src].copy_(flat[lo : lo + int(sizes[rank])])


def agree_on_columns(ncols, device="cuda", group=None):
    """split(): every shard must emit max-over-ranks columns (the one scalar exchanged)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ncols
    t = torch.tensor([ncols], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def sharded_ngrams(local_tokens, n=2, sep="_", ops=None, group=None):
    """NVText::create_ngrams (ngram.cu:32-110) over the token columns of all ranks taken in rank order: rank r
    returns the n-grams that BEGIN in its shard, so the ranks' results in rank order are the single-GPU result
    on the whole token column.  An n-gram that begins in the last n-1 tokens of a shard ends in the shards behind
    it: the ranks all-gather their first n kept tokens (a few hundred bytes) and each appends what follows its
    own.  `local_tokens`: this rank's tokens (e.g. tokenize() of its row range); n >= 2 (n = 1 is a row-local copy:
    call nvtext.ngrams on the shard)."""
    ops = ops or GpuOps()
    n = 2 if n == 0 else int(n)
    if n < 2:
        raise ValueError("sharded_ngrams: n must be at least 2")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ops.ngrams(local_tokens, n, sep)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = ops.drop_empty(local_tokens)  # the rows create_ngrams keeps (ngram.cu:47-58)
    chars, offs, _ = ops.export(ops.head(mine, n))
    (all_chars, all_offs), counts = _all_gather_ragged([chars, offs], group, scalars=[mine.size()])
    counts = [c[0] for c in counts]
    heads = [ops.column(c, o, False) for c, o in zip(all_chars, all_offs)]
    return _ngrams_of_shard(ops, rank, world, local_tokens, mine, heads, counts, n, sep)


def _ngrams_of_shard(ops, rank, world, local_tokens, mine, heads, counts, n, sep):
    """what rank `rank` returns once every rank's first n kept tokens (`heads`) and kept-token count are known"""
    total = sum(counts)
    if total <= n:
        # the reference then returns ONE row, all tokens joined (ngram.cu:60-66); it lands on rank 0.  No shard
        # holds more than n tokens here, so the gathered heads are the whole token column.  (No kept token at
        # all: rank 0 answers for its own rows.)
        if rank != 0:
            return ops.head(mine, 0)
        return ops.ngrams(ops.concat(heads), n, sep) if total else ops.ngrams(local_tokens, n, sep)
    # tokens that follow this shard, n - 1 of them at most
    tail, need = [], n - 1
    for r in range(rank + 1, world):
        take = min(need, counts[r])
        if take:
            tail.append(ops.head(heads[r], take))
            need -= take
    have = mine.size() + (n - 1 - need)
    if mine.size() == 0 or have < n:
        return ops.head(mine, 0)  # no n-gram begins here
    # (have == n: create_ngrams joins all n tokens into one row -- exactly the one n-gram that begins here)
    return ops.ngrams(ops.concat([mine] + tail), n, sep)
