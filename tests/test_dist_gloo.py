"""world_size-2 gloo runs (CPU) of the multi-GPU logic in custrings_amd/dist.py:
the key-set all-gather + merge + remap of the distributed category build, the
shard ranges and the split column-count agreement.  The local string work is done
by a stand-in built on the ORACLE (test infrastructure); on a GPU box the same
code runs with GpuOps (see test_gpu_parity.py::test_gpu_global_category_single_rank)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cpulibs

HERE = os.path.dirname(os.path.abspath(__file__))


class _Cat:
    def __init__(self, keys_col, values):
        self._k, self.values = keys_col, values

    def keys(self):
        return _ColWrap(self._k)

    def keys_size(self):
        return self._k.rows

    def size(self):
        return len(self.values)


class _ColWrap:
    def __init__(self, col):
        self.col = col

    def size(self):
        return self.col.rows


class OracleOps:
    """CPU stand-in for custrings_amd.dist.GpuOps (same interface)."""

    def __init__(self):
        self.o = cpulibs.Oracle()

    def category(self, colw):
        k, v = self.o.category(colw.col)
        return _Cat(k, v), self.export(_ColWrap(k))

    def export(self, colw):
        c = colw.col
        has_null = c.rows > 0 and not c.valid_bits()[0]
        return torch.from_numpy(c.chars.copy()), torch.from_numpy(c.offsets.copy()), bool(has_null)

    def column(self, chars, offsets, null_first):
        rows = offsets.numel() - 1
        valid = None
        if null_first:
            bits = np.ones(rows, dtype=np.uint8)
            bits[0] = 0
            valid = np.packbits(bits, bitorder="little")
        return _ColWrap(cpulibs.Col(chars.numpy(), offsets.numpy(), valid))

    def concat_category(self, cols):
        items = []
        for c in cols:
            items.extend(c.col.to_bytes_list())
        k, v = self.o.category(cpulibs.Col.from_list(items))
        return _ColWrap(k), torch.from_numpy(v.copy())

    def remap(self, cat, table):
        v = torch.from_numpy(np.asarray(cat.values).copy()).long()
        return table[v].to(torch.int32)

    # ---- token columns (sharded_ngrams) ----
    def drop_empty(self, colw):
        items = [b for b in colw.col.to_bytes_list() if b]
        return _ColWrap(cpulibs.Col.from_list(items))

    def head(self, colw, k):
        return _ColWrap(cpulibs.Col.from_list(colw.col.to_bytes_list()[:k]))

    def slice(self, colw, start, end, step=1):
        return _ColWrap(cpulibs.Col.from_list(colw.col.to_bytes_list()[start:end:step]))

    def concat(self, cols):
        items = []
        for c in cols:
            items.extend(c.col.to_bytes_list())
        return _ColWrap(cpulibs.Col.from_list(items))

    def ngrams(self, colw, n, sep):
        return _ColWrap(self.o.ngrams(colw.col, n, sep))


def _worker(rank, world, port, rows, K, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from custrings_amd import dist as csd

    ops = OracleOps()
    lo, hi = csd.shard_range(rows, rank, world)
    local = ops.o.synth(4, lo, hi - lo, param=K)
    keys, values = csd.global_category(_ColWrap(local), ops=ops)
    ncols = csd.agree_on_columns(3 + rank, device="cpu")
    q.put((rank, lo, hi, keys.col.to_list(), values.tolist(), ncols))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows,K", [(5000, 50), (4001, 100000)])
def test_global_category_two_ranks(rows, K):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = cpulibs.Oracle()
    full = o.synth(4, 0, rows, param=K)
    ek, ev = o.category(full)
    for rank, lo, hi, keys, values, ncols in got:
        assert keys == ek.to_list()  # every rank holds the same, global key set
        assert values == ev[lo:hi].tolist()  # and the codes of its own rows
        assert ncols == 4  # max over ranks of (3 + rank)
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == rows


# ---- the merge partitioned by key ranges (K close to N): splitters, all-to-all, all-gather of the merged ranges ----
def _partition_rows(kind, rows):
    import random

    rnd = random.Random(len(kind) * 1000 + rows)
    if kind == "dense":  # nearly every row its own key, a few repeated across the shards
        items = [("k%06d" % rnd.randrange(rows * 4)).encode() for _ in range(rows)]
    elif kind == "few":  # fewer keys than ranks * samples, some ranges stay empty
        items = [rnd.choice([b"a", b"b", b"zz", b""]) for _ in range(rows)]
    elif kind == "nulls":
        items = [None if rnd.random() < 0.1 else ("%x" % rnd.randrange(rows)).encode() for _ in range(rows)]
    elif kind == "skewed":  # the first shard holds one key only, the last all the others (sorted input)
        items = [b"same"] * (rows // 2) + [("t%05d" % i).encode() for i in range(rows - rows // 2)]
    elif kind == "prefixes":  # keys that share long prefixes, multi-byte characters
        items = [("préfixe-commun-" + "x" * rnd.randrange(4) + "%d" % rnd.randrange(rows // 3 + 1)).encode() for _ in range(rows)]
    else:
        items = []
    return items


def _worker_part(rank, world, port, kind, rows, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from custrings_amd import dist as csd

    ops = OracleOps()
    items = _partition_rows(kind, rows)
    lo, hi = csd.shard_range(rows, rank, world)
    keys, values = csd.global_category(_ColWrap(cpulibs.Col.from_list(items[lo:hi])), ops=ops, partitioned=True)
    report = dict(csd.last_category_exchange)
    # the automatic choice: the ranks agree on it from the sum of their key counts
    csd.PARTITION_MIN_KEYS = 10 if kind == "dense" else 1 << 40
    k2, v2 = csd.global_category(_ColWrap(cpulibs.Col.from_list(items[lo:hi])), ops=ops)
    auto = bool(csd.last_category_exchange.get("partitioned"))
    q.put((rank, lo, hi, keys.col.to_list(), values.tolist(), report, k2.col.to_list() == keys.col.to_list() and v2.tolist() == values.tolist(), auto))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,rows,world", [("dense", 6000, 2), ("dense", 5001, 3), ("few", 900, 3), ("nulls", 4000, 2), ("skewed", 3000, 3),
                                             ("prefixes", 2500, 2), ("empty", 0, 2)])
def test_global_category_partitioned_by_key_ranges(kind, rows, world):
    port = 33500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_part, args=(r, world, port, kind, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = cpulibs.Oracle()
    ek, ev = o.category(cpulibs.Col.from_list(_partition_rows(kind, rows)))
    for rank, lo, hi, keys, values, report, same, auto in got:
        assert keys == ek.to_list(), (kind, rank)  # every rank: the global key set of a single build
        assert values == ev[lo:hi].tolist(), (kind, rank)
        assert report.get("partitioned") and report["global_keys"] == ek.rows
        assert same  # the all-gather form agrees
        assert auto == (kind == "dense")
    if kind == "dense":  # every rank merged about its share, not everything
        assert all(g[5]["range_keys"] < 0.8 * ek.rows for g in got), [g[5] for g in got]
        assert sum(g[5]["range_keys"] for g in got) == ek.rows


def test_shard_range_covers_rows():
    from custrings_amd import dist as csd

    for rows in (0, 1, 7, 100_000_003):
        for world in (1, 2, 8):
            edges = [csd.shard_range(rows, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == rows
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


# ---- n-grams across the shard boundary, split's column agreement on a real column, the K ~ N report ----
def _worker2(rank, world, port, case, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import warnings

    from custrings_amd import dist as csd

    ops = OracleOps()
    out = {}
    if case[0] == "ngrams":
        _, rows, n, sep, cuts = case
        lo, hi = cuts[rank], cuts[rank + 1]
        local = cpulibs.Col.from_list(rows[lo:hi])
        toks = ops.o.tokenize(local)
        out["ngrams"] = csd.sharded_ngrams(_ColWrap(toks), n, sep, ops=ops).col.to_list()
    elif case[0] == "split":
        _, total_rows = case
        lo, hi = csd.shard_range(total_rows, rank, world)
        local = ops.o.synth(3, lo, hi - lo)
        cols = ops.o.split(local, " ", -1)
        ncols = csd.agree_on_columns(len(cols), device="cpu")
        # a shard with fewer columns pads with all-null columns (what split emits for rows without such a token)
        out["split"] = [c.to_list() for c in cols] + [[None] * (hi - lo)] * (ncols - len(cols))
        out["range"] = (lo, hi)
    elif case[0] == "dense_keys":
        _, total_rows = case
        lo, hi = csd.shard_range(total_rows, rank, world)
        local = cpulibs.Col.from_list([("key%07d" % i).encode() for i in range(lo, hi)])  # every row its own key
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            keys, values = csd.global_category(_ColWrap(local), ops=ops)
        out["warned"] = [str(x.message) for x in w]
        out["report"] = dict(csd.last_category_exchange)
        out["keys"] = keys.col.rows
        out["values"] = values.tolist()
        out["range"] = (lo, hi)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _run2(case, world=2):
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker2, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [got[r] for r in range(world)]


TWEETS = [b"the quick brown fox", b"jumped", None, b"", b"over the lazy dog and", b"a", b"b c", b"d e f g h", b"  ", b"tail end here"]


@pytest.mark.parametrize("n,sep,cuts", [(2, "_", [0, 5, 10]), (3, "-", [0, 2, 10]), (2, "", [0, 0, 10]), (4, "_", [0, 9, 10]), (3, "_", [0, 3, 4]),
                                        (5, "+", [0, 1, 2])])
def test_sharded_ngrams_equal_the_unsharded_result(n, sep, cuts):
    """ngram.cu:32-110 runs over the token column of all rows: the n-grams that begin in the last n-1 tokens of a
    shard need the next shard's first tokens (custrings_amd/dist.py: sharded_ngrams).  Cases: an ordinary cut, a cut
    that leaves one shard with fewer than n tokens, an empty shard, a last shard of one row, and totals at or below n
    (the reference then returns one joined row)."""
    rows = TWEETS[: cuts[-1]]
    got = _run2(("ngrams", rows, n, sep, cuts))
    o = cpulibs.Oracle()
    want = o.ngrams(o.tokenize(cpulibs.Col.from_list(rows)), n, sep).to_list()
    assert got[0]["ngrams"] + got[1]["ngrams"] == want


def test_sharded_split_agrees_on_the_column_count():
    rows = 3000
    got = _run2(("split", rows))
    o = cpulibs.Oracle()
    want = [c.to_list() for c in o.split(o.synth(3, 0, rows), " ", -1)]
    assert len(got[0]["split"]) == len(got[1]["split"]) == len(want)
    for k in range(len(want)):
        assert got[0]["split"][k] + got[1]["split"][k] == want[k]


def test_global_category_reports_dense_key_sets():
    """with K close to N the key-set all-gather moves as much as the data; the build still
    gives the right answer and says so (a warning and dist.last_category_exchange)."""
    rows = 600
    got = _run2(("dense_keys", rows))
    for r in got:
        assert r["keys"] == rows and r["values"] == list(range(*r["range"]))
        assert r["warned"] and "hash-partitioned" in r["warned"][0]
        assert r["report"]["keys_per_row"] == 1.0 and r["report"]["key_bytes_gathered"] > 0
