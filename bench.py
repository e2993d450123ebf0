#!/usr/bin/env python3
"""Headline benchmark: split(' ') + replace_re(IPv4 -> "<IP>") over the C3 log-line
column (BASELINE.json: "GB/s input chars + Mstrings/s, split+replace_re on 100M
rows"), one process per GPU, rows sharded by range, no data-path collective.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (split + replace_re) over this rank's
100M-row shard, inputs already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402  (first: the process must use ONE HIP runtime, see custrings_amd/__init__.py)
import torch.distributed as dist  # noqa: E402

IPV4 = r"\d+\.\d+\.\d+\.\d+"
REPL = "<IP>"
SEED = 20240607
RANKS_INFO = {"n_gpus": 1, "backend": None, "rccl_ranks": 0}
KERNELS = ["k_split_measure", "k_split_emit", "k_split_write", "k_replace_re", "k_replace_re_size", "k_replace_re_write",
           "k_split_count", "k_split_sizes", "k_write_offsets", "k_scan_lookback"]


def source_hash():
    """sha256 over the kernel sources: profiles/<round>/traffic.json is stamped with it when the
    PMC passes are taken, and bench.py reports `traffic` only while the sources are unchanged."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "custrings_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def effective_cpus():
    """CPUs this process may actually use: the affinity mask cut by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (weak scaling)")
    ap.add_argument("--cpu-rows", type=int, default=3_000_000, help="rows of the same workload timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cold-steps", type=int, default=4, help="steps on a freshly generated column each (reported as `cold`; 0 = skip)")
    ap.add_argument("--pmc-traffic", type=float, default=None, help="HBM bytes/launch of the dominant kernel from a separate rocprofv3 --pmc run")
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "c5"],
                    help="c3 = the headline (default); c2 = lower + strip + split(' ') on 10M rows x 64 chars; "
                         "c4 = distributed NVCategory build (key-set all-gather inside the timed region); "
                         "c5 = tokenize + n-grams(2) with the shard-boundary exchange inside the timed region")
    ap.add_argument("--keys", type=int, default=1_000_000, help="c4: distinct tokens K")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo lets several ranks share one GPU for a plumbing check)")
    ap.add_argument("--no-box", action="store_true", help="skip the box's streaming-rate calibration (`box` block)")
    ap.add_argument("--concurrent-steps", type=int, default=10, help="steps with split and replace_re issued on two streams (reported as `concurrent`; 0 = skip)")
    args = ap.parse_args()

    # ---- one process per GPU.  The driver launches N > 1 under torch.distributed.run (WORLD_SIZE in the environment);
    # a plain `python bench.py --gpus N` re-executes itself that way, so that it cannot silently measure one rank.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = torch.cuda.device_count()
        if args.backend == "nccl" and have < args.gpus:
            sys.exit("bench.py: --gpus %d asked for, %d GPU(s) visible: refusing to measure fewer ranks than asked" % (args.gpus, have))
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the process group has %d rank(s) (WORLD_SIZE): launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    local = local % max(torch.cuda.device_count(), 1)  # (more ranks than GPUs only with --backend gloo)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus, "process group size differs from --gpus"
    global RANKS_INFO
    RANKS_INFO = {"n_gpus": dist.get_world_size() if world > 1 else 1, "backend": (args.backend if world > 1 else None),
                  "rccl_ranks": (dist.get_world_size() if world > 1 and args.backend == "nccl" else 0)}

    from custrings_amd import _lib, nvstrings

    L = _lib.lib
    _lib.ensure_init(local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == "c2":
        run_c2(args, rank, world, barrier)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.config != "c3":
        run_other_config(args, rank, world, barrier)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- what this box's memory delivers to hand-written streaming kernels (custrings_amd/csrc/box_rates.h), same process
    box = None
    if not args.no_box and rank == 0:
        rates = (C.c_double * 6)()
        _lib.check(L.cs_box_rates(2048, 3, None, rates))
        box = {"copy_TBps": round(rates[0], 2), "read_TBps": round(rates[1], 2), "write_TBps": round(rates[2], 2),
               "scatter21_TBps": round(rates[3], 2), "scatter21_nt_TBps": round(rates[4], 2), "shader_GHz_under_load": round(rates[5], 3),
               "what": "own streaming kernels, 2 GiB buffers, median of 3 launches, bytes read + written over the launch time: a 16 B/lane "
                       "copy, a read-only and a write-only stream, and the split emit kernel's shape (one read stream into 20 x (256 + 192 + 8) B "
                       "pieces per 64-row sub-tile, runs of 24 sub-tiles a wave) with plain / non-temporal stores; the shader clock (s_memtime over "
                       "s_memrealtime) one wave saw while every CU ran integer + LDS work"}
        L.cs_pool_trim(0)  # (the calibration's buffers go back to the driver: the pool is empty again, as at the process's start)
    barrier()

    # ---- this rank's shard: rows [rank*rows, (rank+1)*rows) of the C3 column
    out = C.c_void_p()
    _lib.check(L.cs_synth_column(3, rank * args.rows, args.rows, SEED, 0, None, C.byref(out)))
    col = nvstrings.nvstrings(out.value)
    in_bytes = int(L.cs_column_nbytes(col.m_cptr))
    re = nvstrings._compile(IPV4)
    stats = {}

    def do_split(stream=None, record=False):
        arr = C.POINTER(C.c_void_p)()
        ncols = C.c_int()
        _lib.check(L.cs_split(col.m_cptr, b" ", -1, stream, C.byref(arr), C.byref(ncols)))
        if record:
            stats["split_cols"] = ncols.value
            stats["split_out_bytes"] = sum(int(L.cs_column_nbytes(arr[i])) for i in range(ncols.value))
            stats["split_off_bytes"] = 4 if all(int(L.cs_column_offset_width(arr[i])) == 4 for i in range(ncols.value)) else 8
        for i in range(ncols.value):
            L.cs_column_destroy(arr[i])
        L.cs_free(arr)

    def do_replace(stream=None, record=False):
        o = C.c_void_p()
        _lib.check(L.cs_replace_re(col.m_cptr, re, REPL.encode(), -1, stream, C.byref(o)))
        if record:
            stats["replace_out_bytes"] = int(L.cs_column_nbytes(o))
        L.cs_column_destroy(o)

    def step(record=False):
        do_split(None, record)
        do_replace(None, record)

    step(record=True)
    for _ in range(max(args.warmup - 1, 0)):
        step()
    fallbacks0 = int(L.cs_fallback_count())
    L.cs_prof_reset()
    L.cs_prof_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    L.cs_prof_enable(0)

    # ---- the same step on a FRESH column each iteration: nothing cached on the input from an earlier call (the timed
    # loop above re-uses one column, whose metadata -- longest row, largest 64-row span, the non-ASCII sample -- the
    # warm-up step paid for).  Generating the column is not timed; the step is, call to synchronised return.
    barrier()
    cold_ms = []
    mallocs0 = int(L.cs_debug_malloc_count())
    cold_fallbacks0 = int(L.cs_fallback_count())
    warm_prof = {}
    for k in KERNELS:  # (the timed loop's kernel times are read here: the cold steps get a profile of their own)
        ms, n = C.c_double(), C.c_int64()
        L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
        if n.value:
            warm_prof[k] = {"avg_ms": ms.value / n.value, "launches": n.value}
    L.cs_prof_reset()
    L.cs_prof_enable(1)
    for i in range(args.cold_steps):
        fresh = C.c_void_p()
        col = None  # (the old column's buffers go back to the pool first: the new one takes them)
        # (ANOTHER seed every step: the column and all 21 outputs differ in size by a few KB from the last ones; the pool's
        # size classes -- cs_core.hip: size_class -- still find blocks for them, `cold.mallocs` counts the hipMalloc calls)
        _lib.check(L.cs_synth_column(3, rank * args.rows, args.rows, SEED + 1 + i, 0, None, C.byref(fresh)))
        col = nvstrings.nvstrings(fresh.value)
        barrier()
        tc0 = time.perf_counter()
        step()
        barrier()
        cold_ms.append((time.perf_counter() - tc0) * 1e3)
    cold_fallbacks = int(L.cs_fallback_count()) - cold_fallbacks0
    cold_mallocs = int(L.cs_debug_malloc_count()) - mallocs0
    L.cs_prof_enable(0)
    cold_prof = {}
    for k in KERNELS:
        ms, n = C.c_double(), C.c_int64()
        L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
        if n.value:
            cold_prof[k] = round(ms.value / n.value, 3)

    # ---- the same step with the two ops issued on two streams by two host threads (split's emit kernel is bound by its
    # store drain and its LDS round trips, replace_re by its instruction stream: they overlap).  A SEPARATE entry: `value`
    # and `ms_per_step` above are the ops one after the other on one stream.
    conc_elapsed = 0.0
    if args.concurrent_steps > 0:
        import threading

        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        pa, pb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
        errs = []

        def guarded(fn, st):
            try:
                fn(st)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        def step_concurrent():
            ta = threading.Thread(target=guarded, args=(do_split, pa))
            tb = threading.Thread(target=guarded, args=(do_replace, pb))
            ta.start()
            tb.start()
            ta.join()
            tb.join()
            if errs:
                raise errs[0]

        for _ in range(2):
            step_concurrent()
        conc_fallbacks0 = int(L.cs_fallback_count())
        barrier()
        tq = time.perf_counter()
        for _ in range(args.concurrent_steps):
            step_concurrent()
        barrier()
        conc_elapsed = time.perf_counter() - tq
        conc_fallbacks = int(L.cs_fallback_count()) - conc_fallbacks0
        L.cs_stream_forget(pa)
        L.cs_stream_forget(pb)

    t = torch.tensor([elapsed, conc_elapsed] + cold_ms, dtype=torch.float64, device="cuda")
    tot = torch.tensor([float(in_bytes), float(args.rows)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t[0].item())
    conc_elapsed = float(t[1].item())
    cold_ms = [float(x) for x in t[2:].tolist()]
    total_bytes, total_rows = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        # ---- per-kernel device time from HIP events recorded on the launch stream
        prof = warm_prof
        rows = args.rows
        Lb = in_bytes / rows
        Ccols = stats["split_cols"]
        split_out = stats["split_out_bytes"] / rows
        repl_out = stats["replace_out_bytes"] / rows
        ob = stats["split_off_bytes"] + 0.125  # offset + validity bytes per row of a split output column
        # algorithmic bytes per row (SURVEY.md section 8d / BASELINE.md section 2)
        alg_replace = (Lb + 8.125) + (repl_out + 8.125)
        alg_split = (Lb + 8.125) + split_out + Ccols * ob
        # per-kernel share: what that kernel must read and write given its role
        alg_kernel = {
            "k_replace_re": alg_replace,
            "k_split_measure": Lb + 8.125,
            "k_split_emit": Lb + 8.125 + split_out + Ccols * ob,
            "k_split_write": Lb + 8.125 + split_out + Ccols * 8,
            "k_replace_re_size": Lb + 8.125 + 4,
            "k_replace_re_write": Lb + 16 + repl_out,
            "k_split_count": Lb + 8.125 + 4,
            "k_split_sizes": Lb + 8.125 + 4 + 4 * Ccols,
            "k_write_offsets": 12,
        }
        dom = max(prof, key=lambda k: prof[k]["avg_ms"] * prof[k]["launches"]) if prof else None
        # HBM bytes per launch measured in separate rocprofv3 --pmc passes (profiles/<round>/traffic.json);
        # only quoted while the kernel sources are the ones those passes ran on
        tj = None
        try:
            rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "traffic.json")))
            with open(os.path.join(ROOT, "profiles", rounds[-1], "traffic.json")) as f:
                tj = json.load(f)
            if tj.get("rows") != rows or tj.get("source_hash") != source_hash():
                tj = None
        except Exception:
            tj = None

        def roof(k):
            a = alg_kernel.get(k, 0) * rows / (prof[k]["avg_ms"] * 1e-3) / 1e9
            tr = args.pmc_traffic if (args.pmc_traffic is not None and k == dom) else None
            if tr is None and tj and k in tj["kernels"]:
                tr = float(tj["kernels"][k]["hbm_bytes"])
            return {"bound": "hbm", "kernel": k, "achieved": round(a, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(a / 8000.0, 4), "frac_of_achievable_6300": round(a / 6300.0, 4), "traffic": tr,
                    "avg_ms": round(prof[k]["avg_ms"], 3), "alg_bytes_per_row": round(alg_kernel.get(k, 0), 2)}

        roofline_kernels = [roof(k) for k in sorted(prof, key=lambda k: -prof[k]["avg_ms"] * prof[k]["launches"])]
        # ---- op level: an op's algorithmic bytes over the SUMMED device time of every kernel it launched in a step (the
        # measure pass re-reads what emit reads again: that read is the op's cost, not extra algorithmic bytes).
        # `roofline` is the slowest op's line; the per-kernel list above keeps each kernel's own bytes.
        op_kernels = {"split": ["k_split_measure", "k_split_emit", "k_split_write", "k_split_count", "k_split_sizes", "k_write_offsets", "k_scan_lookback"],
                      "replace_re": ["k_replace_re", "k_replace_re_size", "k_replace_re_write"]}
        op_alg = {"split": alg_split, "replace_re": alg_replace}

        def roof_op(op):
            ks = [k for k in op_kernels[op] if k in prof]
            ms = sum(prof[k]["avg_ms"] * prof[k]["launches"] for k in ks) / args.steps
            if ms <= 0:
                return None
            a = op_alg[op] * rows / (ms * 1e-3) / 1e9
            tr = None
            if tj and all(k in tj["kernels"] for k in ks):
                tr = float(sum(tj["kernels"][k]["hbm_bytes"] * prof[k]["launches"] / args.steps for k in ks))
            domk = max(ks, key=lambda k: prof[k]["avg_ms"] * prof[k]["launches"])
            return {"bound": "hbm", "op": op, "kernel": domk, "kernels": {k: round(prof[k]["avg_ms"] * prof[k]["launches"] / args.steps, 3) for k in ks},
                    "achieved": round(a, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(a / 8000.0, 4), "traffic": tr,
                    "ms": round(ms, 3), "alg_bytes_per_row": round(op_alg[op], 2)}

        roofline_ops = [r for r in (roof_op(op) for op in op_kernels) if r]
        roofline = max(roofline_ops, key=lambda r: r["ms"]) if roofline_ops else None
        ms_step = elapsed / args.steps * 1e3
        pipeline = (alg_replace + alg_split) * total_rows / (elapsed / args.steps) / 1e9
        result = {
            "metric": "GB/s input chars, split(' ') + replace_re(IPv4) on 100M log-line rows per GPU",
            "value": round(total_bytes / (elapsed / args.steps) / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": RANKS_INFO["n_gpus"],
            "rccl_ranks": RANKS_INFO["rccl_ranks"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "C3 headline: 100M log-line rows (48-80 B, mean %.1f) per GPU, split(' ') -> %d columns"
                                   " + replace_re('%s','%s')" % (Lb, Ccols, IPV4, REPL),
                       "rows_per_gpu": rows, "seed": SEED, "sharding": "row ranges, no data-path collective"},
            "mstrings_per_s": round(total_rows / (elapsed / args.steps) / 1e6, 1),
            "roofline": roofline,
            "roofline_ops": roofline_ops,
            "roofline_kernels": roofline_kernels,
            "fallbacks_in_timed_region": int(L.cs_fallback_count()) - fallbacks0,
            "roofline_pipeline": {"alg_bytes_per_row": round(alg_replace + alg_split, 1), "achieved": round(pipeline, 1),
                                  "peak": 8000.0 * world, "unit": "GB/s", "frac": round(pipeline / (8000.0 * world), 4)},
            "kernels": {k: {"avg_ms": round(v["avg_ms"], 3), "launches": v["launches"]} for k, v in prof.items()},
        }
        if cold_ms:
            # (the first cold step also grows the buffer pool by a column: reported, not averaged in)
            rest = cold_ms[1:] if len(cold_ms) > 1 else cold_ms
            result["cold"] = {"ms_per_step": round(sum(rest) / len(rest), 3), "steps": len(rest), "first_ms": round(cold_ms[0], 3),
                              "all_ms": [round(x, 3) for x in cold_ms], "fallbacks": cold_fallbacks, "kernels_avg_ms": cold_prof,
                              "mallocs": cold_mallocs,
                              "what": "the same step (split + replace_re, call to synchronised return) on a column of ANOTHER SEED generated just "
                                      "before it (other sizes, nothing cached on it from an earlier call); generation untimed; `mallocs` = hipMalloc "
                                      "calls during all cold steps incl. the generation (the pool's size classes serve the rest)"}
        if box:
            result["box"] = box
        if args.concurrent_steps > 0:
            cms = conc_elapsed / args.concurrent_steps * 1e3
            result["concurrent"] = {"ms_per_step": round(cms, 3), "steps": args.concurrent_steps, "fallbacks": conc_fallbacks,
                                    "value": round(total_bytes / (cms * 1e-3) / 1e9, 2), "unit": "GB/s",
                                    "what": "the same two ops on the same column issued on TWO streams by two host threads (cs_split on one, "
                                            "cs_replace_re on the other), K steps between barriers; not the headline: `ms_per_step` above is "
                                            "the ops one after the other"}
        if not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(args.cpu_rows)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_c2(args, rank, world, barrier):
    """BASELINE.json configs[1]: 10M synthetic UTF-8 rows x 64 chars per GPU, lower() + strip() + split(' ') -- each op on
    the result of the one before, as a caller would chain them.  One JSON line: whole-step throughput, the three ops'
    device times (HIP events on the launch stream, recorded around each call) and their shares of the HBM roofline over
    the ops' algorithmic bytes (output offsets counted at the width actually written)."""
    from custrings_amd import _lib, nvstrings

    L = _lib.lib
    rows = args.rows if args.rows != 100_000_000 else 10_000_000
    out = C.c_void_p()
    _lib.check(L.cs_synth_column(2, rank * rows, rows, SEED, 0, None, C.byref(out)))
    col = nvstrings.nvstrings(out.value)
    in_bytes = int(L.cs_column_nbytes(col.m_cptr))
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    info = {}

    def step(marks=None):
        if marks:
            marks[0].record()
        low = col.lower()
        if marks:
            marks[1].record()
        st = low.strip()
        if marks:
            marks[2].record()
        cols = st.split(" ")
        if marks:
            marks[3].record()
        if "cols" not in info:
            info.update(cols=len(cols), lower_bytes=int(L.cs_column_nbytes(low.m_cptr)), strip_bytes=int(L.cs_column_nbytes(st.m_cptr)),
                        split_bytes=sum(int(L.cs_column_nbytes(c.m_cptr)) for c in cols),
                        off_bytes=4 if all(int(L.cs_column_offset_width(c.m_cptr)) == 4 for c in cols) else 8)

    for _ in range(max(args.warmup, 1)):
        step()
    fallbacks0 = int(L.cs_fallback_count())
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    tot = torch.tensor([float(in_bytes), float(rows)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    if rank != 0:
        return
    per = elapsed / args.steps
    ms = [sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(args.steps)) / args.steps for k in range(3)]
    ov = 8.125
    alg = {"lower": in_bytes + info["lower_bytes"] + 2 * ov * rows,
           "strip": info["lower_bytes"] + info["strip_bytes"] + 2 * ov * rows,
           "split": info["strip_bytes"] + ov * rows + info["split_bytes"] + info["cols"] * (info["off_bytes"] + 0.125) * rows}
    ops = [{"op": name, "ms": round(ms[k], 3), "alg_bytes_per_row": round(alg[name] / rows, 1),
            "achieved_GBps": round(alg[name] / (ms[k] * 1e-3) / 1e9, 1), "frac": round(alg[name] / (ms[k] * 1e-3) / 8e12, 4)}
           for k, name in enumerate(("lower", "strip", "split"))]
    dom = max(ops, key=lambda o: o["ms"])
    total_alg = sum(alg.values())
    result = {
        "metric": "GB/s input chars, lower() + strip() + split(' ') on 10M rows x 64 chars per GPU (C2)",
        "value": round(float(tot[0].item()) / per / 1e9, 2), "unit": "GB/s", "n_gpus": RANKS_INFO["n_gpus"], "rccl_ranks": RANKS_INFO["rccl_ranks"], "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(per * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "C2: %d rows x 64 chars per GPU (5 %% rows with two-byte characters, 1 %% null, 0.5 %% empty), lower() -> strip() -> split(' ') into %d columns"
                               % (rows, info["cols"]), "rows_per_gpu": rows, "seed": SEED, "sharding": "row ranges, no data-path collective"},
        "mstrings_per_s": round(float(tot[1].item()) / per / 1e6, 1),
        "roofline": {"bound": "hbm", "kernel": dom["op"], "achieved": dom["achieved_GBps"], "peak": 8000.0, "unit": "GB/s", "frac": dom["frac"], "traffic": None},
        "roofline_ops": ops,
        "roofline_pipeline": {"alg_bytes_per_row": round(total_alg / rows, 1), "achieved": round(total_alg * world / per / 1e9, 1), "peak": 8000.0 * world,
                              "unit": "GB/s", "frac": round(total_alg / per / 8e12, 4)},
        "fallbacks_in_timed_region": int(L.cs_fallback_count()) - fallbacks0,
    }
    print(json.dumps(result), flush=True)


def run_other_config(args, rank, world, barrier):
    """The two BASELINE.json configs with an exchange step, one process per GPU, the collective inside the timed
    region: c4 = NVCategory build of a 16-character token column (1B rows over 8 GPUs = 125M per GPU, Zipf over K
    distinct tokens): local build, RCCL all-gather of the sorted key sets, merge, remap (custrings_amd/dist.py);
    c5 = tokenize + n-grams(2) of tweet-like rows (500M over 8 GPUs = 62.5M per GPU) with the first-tokens exchange
    across the shard boundaries.  Rank 0 prints ONE JSON line (no CPU baseline: that belongs to the headline)."""
    from custrings_amd import _lib, nvstrings, nvtext
    from custrings_amd import dist as csd

    L = _lib.lib
    c4 = args.config == "c4"
    rows = args.rows if args.rows != 100_000_000 else (125_000_000 if c4 else 62_500_000)
    out = C.c_void_p()
    _lib.check(L.cs_synth_column(4 if c4 else 5, rank * rows, rows, SEED, args.keys if c4 else 0, None, C.byref(out)))
    col = nvstrings.nvstrings(out.value)
    in_bytes = int(L.cs_column_nbytes(col.m_cptr))
    ops = csd.GpuOps()
    info = {}

    def step():
        if c4:
            keys, values = csd.global_category(col, ops=ops)
            info["keys"] = keys.size()
            info["exchange"] = dict(csd.last_category_exchange)
        else:
            toks = nvtext.tokenize(col)
            grams = csd.sharded_ngrams(toks, 2, "_", ops=ops)
            info["tokens"], info["ngrams"] = toks.size(), grams.size()

    for _ in range(max(args.warmup, 1)):
        step()
    fallbacks0 = int(L.cs_fallback_count())
    L.cs_prof_reset()
    L.cs_prof_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    L.cs_prof_enable(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    tot = torch.tensor([float(in_bytes), float(rows)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    if rank != 0:
        return
    per = elapsed / args.steps
    result = {
        "metric": ("Mstrings/s, NVCategory key build incl. the all-gather of the key sets (C4)" if c4 else
                   "GB/s input chars, tokenize + ngrams(2) incl. the shard-boundary exchange (C5)"),
        "value": round((float(tot[1].item()) / per / 1e6) if c4 else (float(tot[0].item()) / per / 1e9), 2),
        "unit": "Mstrings/s" if c4 else "GB/s",
        "n_gpus": RANKS_INFO["n_gpus"], "rccl_ranks": RANKS_INFO["rccl_ranks"], "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(per * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": ("C4: %d rows x 16-char tokens per GPU, Zipf(1.1) over K = %d, category build + key-set all-gather + merge + remap" % (rows, args.keys))
                   if c4 else ("C5: %d tweet-like rows per GPU, tokenize() + ngrams(2, '_') with the first-token exchange" % rows),
                   "rows_per_gpu": rows, "seed": SEED, "sharding": "row ranges; " + ("RCCL all-gather of the key sets" if c4 else "all-gather of each rank's first 2 tokens")},
        "gb_per_s_input": round(float(tot[0].item()) / per / 1e9, 2),
        "mstrings_per_s": round(float(tot[1].item()) / per / 1e6, 1),
        "fallbacks_in_timed_region": int(L.cs_fallback_count()) - fallbacks0,
        "rank0": info,
    }
    print(json.dumps(result), flush=True)


def _pandas_chunk(args):
    first, n = args
    import time as _t

    import cpulibs
    import pandas as pd

    s = pd.Series(cpulibs.Oracle().synth(3, first, n).to_list())
    t0 = _t.perf_counter()
    s.str.split(" ", expand=True)
    s.str.replace(IPV4, REPL, regex=True)
    return int(s.str.len().sum()), _t.perf_counter() - t0


def cpu_baseline(rows):
    """The oracle (scalar C++ port of the reference algorithm) on a bounded sample of the same
    workload: one thread, then one worker process per USABLE core (affinity mask cut by the cgroup
    CPU quota -- on a box whose container is limited to a few CPUs, starting a worker per visible
    core only measures the quota); plus pandas.Series.str, one core and a Pool over the usable
    cores, as BASELINE.json's north_star asks."""
    import numpy as np

    import cpulibs
    import engines

    orc = cpulibs.Oracle()
    c = orc.synth(3, 0, rows)
    blob = np.ascontiguousarray(engines.reference_blob(IPV4) if engines.reference_blob(IPV4) is not None else engines.product_blob(IPV4), dtype=np.int32)
    t0 = time.perf_counter()
    orc.split(c, " ")
    orc.replace_re(c, blob, REPL)
    dt = time.perf_counter() - t0
    visible, cores = len(os.sched_getaffinity(0)), effective_cpus()
    res = {"value": round(c.chars.size / dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": "first %d rows of the same C3 column, split(' ') + replace_re, oracle/liboracle.so, %.1f s" % (rows, dt),
           "host": {"cpus_visible": visible, "cpus_usable": cores, "os_cpu_count": os.cpu_count()}}
    # the same port on all usable cores: one worker process per core (tests/cpu_worker.py), each on
    # its own row range of the same column (about as long as the one-thread leg); all start together
    try:
        import subprocess
        import tempfile

        per = max(50_000, min(rows, 2_000_000))
        worker = os.path.join(ROOT, "tests", "cpu_worker.py")
        prog = os.path.join(tempfile.mkdtemp(prefix="cs_bench_"), "ipv4_program.npy")
        np.save(prog, blob)
        env = dict(os.environ, CS_CPULIBS_PREBUILT="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, worker, str(i * per), str(per), prog], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  text=True, env=env) for i in range(cores)]
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu worker failed to start")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        total = 0
        for p in procs:
            total += int(p.stdout.readline().split()[1])
        dta = time.perf_counter() - t0
        for p in procs:
            p.wait()
        res["all_cores"] = {"value": round(total / dta / 1e9, 4), "unit": "GB/s", "cores": cores, "rows": per * cores,
                            "seconds": round(dta, 1), "speedup_over_one_core": round(total / dta / (c.chars.size / dt), 1)}
    except Exception as e:
        res["all_cores"] = {"error": str(e)}
    try:
        import multiprocessing as mp

        import pandas as pd

        n = min(rows, 1_000_000)
        nb, dtp = _pandas_chunk((0, n))
        res["pandas"] = {"value": round(nb / dtp / 1e9, 5), "unit": "GB/s", "cores": 1, "rows": n, "seconds": round(dtp, 1),
                         "host_cpus": os.cpu_count(), "version": pd.__version__,
                         "extrapolated_s_for_100M_rows": round(dtp * 1e8 / n, 0)}
        chunk = max(50_000, n // 4)
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_pandas_chunk, [(i * 1000, 1000) for i in range(cores)])  # start the workers, import pandas
            t0 = time.perf_counter()
            parts = pool.map(_pandas_chunk, [(i * chunk, chunk) for i in range(cores)])
            dtq = time.perf_counter() - t0
        res["pandas_pool"] = {"value": round(sum(p[0] for p in parts) / dtq / 1e9, 5), "unit": "GB/s", "cores": cores,
                              "rows": chunk * cores, "seconds": round(dtq, 1)}
    except Exception as e:  # pandas is a courtesy figure, never fatal
        res.setdefault("pandas", {"error": str(e)})
        res["pandas_pool"] = {"error": str(e)}
    return res


if __name__ == "__main__":
    main()
