"""-m gpu: the exchange paths of custrings_amd/dist.py with the real GPU ops in TWO processes.  Both ranks share the one
GPU of the test box, so the process group is gloo (RCCL refuses two ranks on one device); dist.py moves the small
exchanged tensors through host memory for that backend.  What is checked is the product code on both sides of the
collective: local category build -> key-set all-gather -> merge -> remap, and tokenize -> first-tokens all-gather ->
n-grams, each against the single-process result on the whole column."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, rows, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import ctypes as C

        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from custrings_amd import _lib, nvstrings, nvtext
        from custrings_amd import dist as csd

        _lib.ensure_init(0)
        lo, hi = csd.shard_range(rows, rank, world)

        def synth(kind, first, n, param=0):
            out = C.c_void_p()
            _lib.check(_lib.lib.cs_synth_column(kind, first, n, 20240607, param, None, C.byref(out)))
            return nvstrings.nvstrings(out.value)

        keys, values = csd.global_category(synth(4, lo, hi - lo, 3000))
        # the merge partitioned by key ranges (what K close to N takes): mostly distinct keys, a null row on rank 1
        dense = synth(4, lo, hi - lo, 1 << 30)
        pk, pv = csd.global_category(dense, partitioned=True)
        part = (pk.to_host(), pv.cpu().tolist(), dict(csd.last_category_exchange))
        grams = csd.sharded_ngrams(nvtext.tokenize(synth(5, lo, hi - lo)), 2, "_")
        ncols = csd.agree_on_columns(len(synth(3, lo, hi - lo).split(" ")), device="cpu")
        # the C ABI's own distributed entry point (the exchange inside the library, torch.distributed as its transport)
        ccat = csd.global_category_c_abi(synth(4, lo, hi - lo, 3000))
        cabi = (ccat.keys().to_host(), ccat.values())
        # ... and its key-range partitioned merge (forced: the test's key sets are far below 2^21 keys), on the mostly
        # distinct keys and on a skewed split of the small key set (rank 0 gets a tenth of the rows)
        _lib.check(_lib.lib.cs_config_set(b"CS_DIST_PARTITIONED", b"1"))
        pcat = csd.global_category_c_abi(dense)
        slo, shi = (0, rows // 10) if rank == 0 else (rows // 10 + (rank - 1) * ((rows - rows // 10) // (world - 1)), rows // 10 + rank * ((rows - rows // 10) // (world - 1)) if rank < world - 1 else rows)
        scat = csd.global_category_c_abi(synth(4, slo, shi - slo, 3000))
        _lib.check(_lib.lib.cs_config_set(b"CS_DIST_PARTITIONED", None))
        cabi = cabi + (pcat.keys().to_host(), pcat.values(), scat.keys().to_host(), scat.values())
        q.put((rank, "ok", keys.to_host(), values.cpu().tolist(), grams.to_host(), ncols, part, cabi))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # the parent reports it
        import traceback

        q.put((rank, "error", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_gpu_two_rank_exchanges_with_the_gpu_ops(gpu_engine, world):
    import ctypes as C

    import torch.multiprocessing as mp

    from custrings_amd import _lib, nvcategory, nvstrings, nvtext

    rows = 40_000
    port = 33500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] == "ok" for g in got), [g[2] for g in got if g[1] != "ok"]

    def synth(kind, param=0):
        out = C.c_void_p()
        _lib.check(_lib.lib.cs_synth_column(kind, 0, rows, 20240607, param, None, C.byref(out)))
        return nvstrings.nvstrings(out.value)

    cat = nvcategory.from_strings(synth(4, 3000))
    want_keys, want_values = cat.keys().to_host(), cat.values()
    want_grams = nvtext.ngrams(nvtext.tokenize(synth(5)), 2, "_").to_host()
    want_cols = len(synth(3).split(" "))
    assert all(g[2] == want_keys for g in got)                              # the same, global key set on every rank
    assert sum((g[3] for g in got), []) == list(want_values)                # each rank the codes of its own rows
    assert sum((g[4] for g in got), []) == want_grams                       # the n-grams across the shard boundaries included
    assert all(g[5] == want_cols for g in got)
    dcat = nvcategory.from_strings(synth(4, 1 << 30))
    dkeys, dvalues = dcat.keys().to_host(), list(dcat.values())
    assert all(g[6][0] == dkeys for g in got)
    assert sum((g[6][1] for g in got), []) == dvalues
    infos = [g[6][2] for g in got]
    assert infos[0]["partitioned"] and infos[0]["global_keys"] == len(dkeys) and sum(i["range_keys"] for i in infos) == len(dkeys)
    assert all(0.4 / world * len(dkeys) < i["range_keys"] < 1.6 / world * len(dkeys) for i in infos)  # (each rank merged about its share)
    # cs_category_build_distributed_with2: the same global key set and codes out of the library's own exchange
    assert all(g[7][0] == want_keys for g in got)
    assert sum((list(g[7][1]) for g in got), []) == list(want_values)
    # ... by key ranges (mostly distinct keys), and with a skewed split of the rows
    assert all(g[7][2] == dkeys for g in got)
    assert sum((list(g[7][3]) for g in got), []) == dvalues
    assert all(g[7][4] == want_keys for g in got)
    assert sum((list(g[7][5]) for g in got), []) == list(want_values)


def _failing_worker(rank, world, port, rows, q, fail_at=None):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import ctypes as C

        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from custrings_amd import _lib, nvstrings
        from custrings_amd import dist as csd

        _lib.ensure_init(0)
        lo, hi = csd.shard_range(rows, rank, world)
        out = C.c_void_p()
        _lib.check(_lib.lib.cs_synth_column(4, lo, hi - lo, 20240607, 3000, None, C.byref(out)))
        col = nvstrings.nvstrings(out.value)
        if fail_at is None:
            _lib.check(_lib.lib.cs_config_set(b"CS_DIST_TEST_FAIL", b"1"))  # rank 1's local build "fails"
        else:  # a failure INSIDE the partitioned merge: "<rank>:<stage>" (cs_dist.hip: merge_partitioned)
            _lib.check(_lib.lib.cs_config_set(b"CS_DIST_PARTITIONED", b"1"))
            _lib.check(_lib.lib.cs_config_set(b"CS_DIST_TEST_FAIL_AT", fail_at.encode()))
        try:
            csd.global_category_c_abi(col)
            res = "no error"
        except RuntimeError as e:
            res = str(e)
        _lib.check(_lib.lib.cs_config_set(b"CS_DIST_TEST_FAIL", None))
        _lib.check(_lib.lib.cs_config_set(b"CS_DIST_TEST_FAIL_AT", None))
        ok = csd.global_category_c_abi(col).keys_size()  # the ranks are still in step: the next build works
        q.put((rank, "ok", res, ok))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback

        q.put((rank, "error", traceback.format_exc() + repr(e)))


def test_gpu_distributed_build_fails_on_every_rank_together():
    """a rank whose local build throws must not leave the others waiting in a collective.  It takes part in the
    first exchange with a status word; every rank returns an error (naming the rank), none hangs, and the next build works."""
    import torch.multiprocessing as mp

    rows, world = 20_000, 2
    port = 35500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] == "ok" for g in got), [g[2] for g in got if g[1] != "ok"]
    assert all("rank 1 failed before the exchange" in g[2] for g in got), [g[2] for g in got]
    assert got[0][3] == got[1][3] > 0


@pytest.mark.parametrize("fail_at", ["1:1", "1:2", "0:3", "1:4"])
def test_gpu_partitioned_merge_fails_on_every_rank_together(fail_at):
    """the key-range partitioned merge runs about ten collectives with multi-GB allocations and consistency checks
    between them.  A rank that fails in any stage (sample, range cuts, receive buffers, merge of its range) reports it at the
    next agreement point and EVERY rank returns an error there -- none is left waiting in RCCL -- and the next build works."""
    import torch.multiprocessing as mp

    rows, world = 20_000, 2
    port = 37500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, rows, q, fail_at)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(g[1] == "ok" for g in got), [g[2] for g in got if g[1] != "ok"]
    bad_rank, stage = fail_at.split(":")
    assert "simulated failure in stage %s" % stage in got[int(bad_rank)][2], got[int(bad_rank)][2]
    assert ("rank %s failed" % bad_rank) in got[1 - int(bad_rank)][2], got[1 - int(bad_rank)][2]
    assert got[0][3] == got[1][3] > 0


def test_gpu_category_build_distributed_over_rccl():
    """cs_category_build_distributed with a real ncclComm_t: one rank (RCCL refuses two ranks on the test box's one GPU),
    so the all-gathers run a rank with itself -- RCCL resolved in the process, the communicator passed through, the
    padded key sets back through the merge.  Same keys and codes as the local build; a null row and an empty key set too."""
    import ctypes as C

    from custrings_amd import _lib, nvcategory, nvstrings

    _lib.ensure_init(0)
    # the RCCL of the HIP runtime this process already uses (PyTorch bundles both: a second runtime next to it crashes)
    import torch

    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = C.CDLL(bundled if os.path.exists(bundled) else "librccl.so.1", mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        for kind, rows, param in ((4, 50_000, 3000), (4, 2_000, 1 << 30), (2, 5_000, 0)):
            out = C.c_void_p()
            _lib.check(_lib.lib.cs_synth_column(kind, 0, rows, 20240607, param, None, C.byref(out)))
            col = nvstrings.nvstrings(out.value)
            want = nvcategory.from_strings(col)
            got = C.c_void_p()
            _lib.check(_lib.lib.cs_category_build_distributed(col.m_cptr, comm, 1, 0, None, C.byref(got)))
            cat = nvcategory.nvcategory(got.value)
            assert cat.keys().to_host() == want.keys().to_host()
            assert list(cat.values()) == list(want.values())
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("config", ["c3", "c5"])
def test_gpu_bench_two_ranks_on_one_gpu(config):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), here with two ranks on
    the one GPU over gloo (--backend gloo): the barrier / max-over-ranks / sum-over-ranks plumbing and rank 0's single
    JSON line."""
    import json
    import subprocess

    root = os.path.dirname(HERE)
    port = 34500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", config, "--rows", "300000", "--steps", "2", "--warmup", "1", "--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    if config == "c5":
        assert d["rank0"]["ngrams"] == d["rank0"]["tokens"]  # (rank 0's last token pairs with rank 1's first)
    else:
        assert d["config"]["rows_per_gpu"] == 300000 and abs(d["mstrings_per_s"] * d["ms_per_step"] / 1e3 - 0.6) < 0.01
