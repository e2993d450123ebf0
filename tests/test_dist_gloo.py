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


def test_shard_range_covers_rows():
    from custrings_amd import dist as csd

    for rows in (0, 1, 7, 100_000_003):
        for world in (1, 2, 8):
            edges = [csd.shard_range(rows, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == rows
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
