#!/usr/bin/env python3
"""Benchmark of the CSparse.py hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json metric): cs_gaxpy achieved HBM GB/s on the 5M x 5M,
64-nnz-per-column matrix ("G-rand"), plus batched cs_cholsol
solves/s on the 5M x 5M 64-nnz/row SPD matrix ("G-spd").  One "step" is one
cs_gaxpy pass y += A x over the whole matrix.

N > 1, one rank per GPU: either the driver starts the ranks (`python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, WORLD_SIZE set) or
`python bench.py --gpus N` starts them itself (N child processes, before the parent
touches a GPU).  A world size that differs from --gpus is an error (exit 2), never a
silent 1-GPU run.  The path shards as independent matrices / right-hand-side blocks
(SURVEY 8e): every rank owns its own matrix and its own RHS block, no collective in
the timed headline region, which is bracketed by barriers and timed by the slowest
rank ("weak").  The exchange steps SURVEY 8e names are measured as their own legs
and reported under "exchange": factor once on rank 0 + RCCL broadcast of L.p / L.i /
L.x against factoring redundantly, the RHS blocks leaving the root, the solution
blocks gathered to the root.

Rank 0 prints one JSON line.  `value` = algorithmic bytes of all ranks' steps /
wall time.  `roofline.achieved` = algorithmic bytes of one step / average step
duration measured with HIP events on the stream the kernels run on.
`cpu_baseline` = the pure-Python port (oracle/csparse_oracle.py: the reference is
pure Python, single-threaded) on a bounded sample of the same generator.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "csparse.py_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s copy-achievable)


def measured_traffic(kernel_substr, **match):
    """HBM-side bytes per launch of a kernel from the newest committed counter summary that was taken at THIS
    size: profiles/*_traffic.json (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes).  None when no summary matches --
    a figure measured on another kernel or size is not reported."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        meta = d.get("_meta", {})
        if any(meta.get(k) != v for k, v in match.items()):
            continue
        for name, rec in d.get("kernels", {}).items():
            if kernel_substr in name:
                best = {"bytes": rec["hbm_bytes_per_launch"], "source": os.path.relpath(path, ROOT)}
    return best


def gaxpy_bytes(m, n, nnz):
    # 12 nnz + 4 (n+1) + 8 n (x) + 16 m (y read + write)
    return 12 * nnz + 4 * (n + 1) + 8 * n + 16 * m


def cholsol_bytes(lnz, n, k):
    # forward + backward factor reads, b -> y -> x each read and written
    return 2 * 12 * lnz + 4 * 8 * n * k


def cpu_baseline(n_cpu, per_col, budget_s, gen="uniform"):
    """Pure-Python port of cs_gaxpy (1 core) on G-rand (same row draw as the headline) at n_cpu; also the plain-C port."""
    import numpy as np
    import csparse_oracle as O
    import c_oracle as CO
    import synth
    Ap, Ai, Ax = (synth.grand_uniform if gen == "uniform" else synth.grand)(n_cpu, per_col, 20240601)
    x = synth.vec(n_cpu, 1, 0.5, 1.5)
    A = O.cs_spalloc(n_cpu, n_cpu, len(Ai), True, False)
    A.p, A.i, A.x = Ap.tolist(), Ai.tolist(), Ax.tolist()
    xl, yl = x.tolist(), [0.0] * n_cpu
    nnz = len(Ai)
    best = None
    t_end = time.perf_counter() + budget_s
    runs = 0
    while runs < 1 or (time.perf_counter() < t_end and runs < 5):
        t0 = time.perf_counter()
        O.cs_gaxpy(A, xl, yl)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        runs += 1
    by = gaxpy_bytes(n_cpu, n_cpu, nnz)
    y0 = np.zeros(n_cpu)
    CO.gaxpy(n_cpu, n_cpu, Ap, Ai, Ax, x, y0)
    tc = None
    for _ in range(3):
        t0 = time.perf_counter()
        CO.gaxpy(n_cpu, n_cpu, Ap, Ai, Ax, x, y0)
        dt = time.perf_counter() - t0
        tc = dt if tc is None else min(tc, dt)
    py = {"value": by / best / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
          "sample": "cs_gaxpy, pure-Python port (list-based, as the reference), G-rand n=%d, %d nnz/col "
                    "(%d nnz), best of %d, %.3f s/pass = %.2f M nnz/s" % (n_cpu, per_col, nnz, runs, best, nnz / best / 1e6)}
    c = {"value": by / tc / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
         "sample": "same sample, plain-C port (oracle/oracle.c, gcc -O2), %.4f s/pass" % tc}
    return py, c


def cpu_baseline_cholsol(nblocks, bs, budget_s, lnz_full):
    """The cholsol leg on one host core: pure-Python port of cs_schol + cs_chol + (cs_lsolve, cs_ltsolve) on
    G-spd with `nblocks` blocks, solves repeated within the time budget.  A solve costs ~2 lnz operations, so
    the rate on the full 5M-row factor is the sample's rate x lnz(sample) / lnz(full) (reported, not measured)."""
    import numpy as np
    import csparse_oracle as O
    import c_oracle as CO
    import synth
    n = nblocks * bs
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 20240601 + 5)
    A = O.cs_spalloc(n, n, len(Ai), True, False)
    A.p, A.i, A.x = Ap.tolist(), Ai.tolist(), Ax.tolist()
    t0 = time.perf_counter()
    S = O.cs_schol(0, A)
    N = O.cs_chol(A, S)
    t_factor = time.perf_counter() - t0
    lnz = N.L.p[n]
    B = synth.rhs(n, 8, 0)
    t_end = time.perf_counter() + budget_s
    solves, t_solve = 0, 0.0
    while solves < 1 or (time.perf_counter() < t_end and solves < 8):
        x = B[:, solves].tolist()
        t0 = time.perf_counter()
        O.cs_lsolve(N.L, x)
        O.cs_ltsolve(N.L, x)
        t_solve += time.perf_counter() - t0
        solves += 1
    rate = solves / t_solve
    # plain-C port, same sample
    parent, cp = CO.schol(n, Ap, Ai)
    t0 = time.perf_counter()
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    tc_factor = time.perf_counter() - t0
    t0 = time.perf_counter()
    for r in range(8):
        CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, B[:, r]))
    tc = (time.perf_counter() - t0) / 8
    return {"value": rate, "unit": "solves/s", "cores": 1, "kind": "port",
            "sample": "cs_cholsol solve phase (cs_lsolve + cs_ltsolve), pure-Python port, G-spd %d blocks of %d (n=%d, "
                      "lnz=%d), %d solves, %.3f s each; cs_schol + cs_chol %.2f s" % (nblocks, bs, n, lnz, solves,
                                                                                    t_solve / solves, t_factor),
            "solves_per_s_scaled_to_full_lnz": rate * lnz / lnz_full,
            "plain_c_port": {"solves_per_s": 1.0 / tc, "solves_per_s_scaled_to_full_lnz": lnz / lnz_full / tc,
                             "chol_s": tc_factor}}


class Deadline(object):
    """A bound on everything that runs after the headline measurement.  The legs reported beside the headline
    (G-spd, cholsol, the exchange legs, the column-sharded SpMV) are collectives at N > 1: a rank that fails inside
    one leaves the others waiting.  The headline must not be lost to that, so after `seconds` rank 0 prints the
    JSON line as it stood when the last leg completed (with "extras_cut_short_after_s") and every rank exits 0."""

    def __init__(self, seconds, rank):
        import threading
        self.line, self.printed, self.rank, self.seconds = None, False, rank, seconds
        self.lock = threading.Lock()
        self.timer = threading.Timer(seconds + (0.0 if rank == 0 else 5.0), self.fire)
        self.timer.daemon = True

    def arm(self, out):
        self.checkpoint(out)
        if self.seconds > 0:
            self.timer.start()

    def checkpoint(self, out):
        with self.lock:
            self.line = json.dumps(dict(out, extras_cut_short_after_s=self.seconds))

    def emit(self, out):
        with self.lock:
            if self.rank == 0 and not self.printed:
                print(json.dumps(out), flush=True)
            self.printed = True

    def fire(self):
        with self.lock:
            if self.rank == 0 and not self.printed and self.line:
                sys.stdout.write(self.line + "\n")
                sys.stdout.flush()
            os._exit(0)


def launch_ranks(n):
    """`python bench.py --gpus N` run directly: start N ranks of this same command line, one per GPU, with
    the torch.distributed.run environment (rendezvous on 127.0.0.1), and return the worst exit code.  The
    parent makes no HIP call.  Rank 0's stdout is the one JSON line; other ranks' stdout is dropped."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def rehearse(args):
    """--rehearse: the N-rank control flow and every exchange leg on small CPU tensors over gloo.  Proves the
    launcher starts N ranks, that a size mismatch fails, and that the collectives pair up; measures nothing."""
    import shard
    os.environ.setdefault("CSX_COMM_BACKEND", "gloo")
    comm = shard.Comm(backend="gloo")
    import torch
    rank, world = comm.rank, comm.world
    n, k = 1000, 3
    legs = {}
    Lx = torch.arange(n, dtype=torch.float64) if rank == 0 else torch.zeros(n, dtype=torch.float64)
    comm.barrier()
    comm.broadcast_tensor(Lx)
    legs["broadcast_ok"] = bool(Lx[-1].item() == n - 1)
    mine = torch.empty(n, k, dtype=torch.float64)
    blocks = [torch.full((n, k), float(r)) .double() for r in range(world)] if rank == 0 else None
    comm.scatter_blocks(mine, blocks)
    legs["scatter_ok"] = bool(comm.sum(float(mine[0, 0].item() == rank)) == world)
    got = comm.gather_to_root(mine + 1.0)
    if rank == 0:
        legs["gather_ok"] = [float(g[0, 0].item()) for g in got] == [r + 1.0 for r in range(world)]
    tmax = comm.max(float(rank))
    if rank == 0:
        print(json.dumps({"metric": "cs_gaxpy achieved HBM GB/s (algorithmic bytes / time), 5M x 5M CSC, 64 nnz/col",
                          "value": None, "unit": "GB/s", "n_gpus": world, "steps": 0, "warmup": 0, "rehearsal": True,
                          "exchange": legs, "slowest_rank": tmax}))
    comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=5000000)
    ap.add_argument("--per-col", type=int, default=64)
    ap.add_argument("--nrhs", type=int, default=128, help="right-hand sides per GPU for batched cs_cholsol")
    ap.add_argument("--mode", default="auto", choices=["auto", "wave", "tiled", "atomic"])
    ap.add_argument("--cpu-n", type=int, default=100000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-chol-blocks", type=int, default=2000, help="G-spd blocks in the CPU baseline of the cholsol leg")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-cholsol", action="store_true")
    ap.add_argument("--skip-configs", action="store_true", help="skip BASELINE's other configs (bench_configs.py sections)")
    ap.add_argument("--skip-gspd", action="store_true")
    ap.add_argument("--skip-sharded", action="store_true", help="skip the column-sharded single SpMV (N > 1 only)")
    ap.add_argument("--force-sharded", action="store_true", help="run the column-sharded SpMV code path at N = 1 too")
    ap.add_argument("--gen", default="uniform", choices=["uniform", "stratified"],
                    help="G-rand row draw of the headline matrix: SURVEY 8d's uniform distinct rows (default) or one "
                         "row per stratum of n/per_col rows (round 1's generator); the other one is timed beside it")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher and exchange plumbing only, on CPU tensors (gloo): no GPU, no kernels, value = null")
    ap.add_argument("--extras-deadline", type=float, default=420.0,
                    help="seconds the legs after the headline may take before rank 0 prints what it has (0: no bound)")
    ap.add_argument("--dry-exchange", action="store_true",
                    help="only the exchange preflight (every exchange leg at n = 100 000, 8 right-hand sides, pass / fail per leg): "
                         "no headline, value = null.  At N > 1 the preflight always runs before the full-size legs.")
    ap.add_argument("--exchange-nrhs", type=int, default=None,
                    help="right-hand sides per GPU in the scatter / gather legs (default: --nrhs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # parent: starts the ranks, touches no GPU
    import shard
    if shard.env_rank()[1] != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: refusing to report a run of another size\n"
                         % (args.gpus, shard.env_rank()[1]))
        sys.exit(2)
    if args.rehearse:
        return rehearse(args)
    comm = shard.Comm()  # RCCL inside libcsx (csx_comm_*) when there is more than one rank, no-op at N = 1
    rank, world, local = comm.rank, comm.world, comm.local
    import numpy as np
    import _csx
    import csparse as cs
    _csx.init(local)
    lib = _csx.lib()

    def barrier():
        comm.barrier(_csx.sync)

    max_over_ranks = comm.max

    if args.dry_exchange:
        out = {"metric": "cs_gaxpy achieved HBM GB/s (algorithmic bytes / time), 5M x 5M CSC, 64 nnz/col", "value": None,
               "unit": "GB/s", "n_gpus": world, "steps": 0, "warmup": 0, "dry_exchange": True}
        deadline = Deadline(args.extras_deadline, rank)
        deadline.arm(out)
        exchange_preflight(args, lib, cs, comm, barrier, deadline, out)
        deadline.emit(out)
        comm.close()
        deadline.timer.cancel()
        return

    n, per_col = args.n, args.per_col
    nnz = n * per_col
    # ---- headline: cs_gaxpy on G-rand; every rank owns an independent matrix ----
    gens = {"uniform": lib.csx_gen_grand_uniform, "stratified": lib.csx_gen_grand}
    gen_words = {"uniform": "%d distinct rows per column drawn uniformly from [0, n), ascending (SURVEY 8d)" % per_col,
                 "stratified": "%d rows per column, one uniformly random row per stratum of n/%d rows" % (per_col, per_col)}
    hA = _csx.new_handle()
    _csx.check(gens[args.gen](n, per_col, 20240601 + 1 + rank, hA), "gen_grand")
    hx, hy = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_gen_vec(n, 7 + rank, 0.5, 1.5, hx), "gen_vec")
    _csx.check(lib.csx_vec_alloc(n, hy), "vec_alloc")
    modes = {"wave": cs.GAXPY_WAVE, "tiled": cs.GAXPY_TILED, "atomic": cs.GAXPY_ATOMIC}
    cand = ["tiled", "wave"] if args.mode == "auto" else [args.mode]
    trial = {}
    for name in cand:
        t0 = time.perf_counter()
        _csx.check(lib.csx_gaxpy_prepare(hA, modes[name]), "prepare " + name)
        _csx.sync()
        prep = time.perf_counter() - t0
        _csx.check(lib.csx_gaxpy(hA, hx, hy, modes[name]), "gaxpy " + name)
        with _csx.Timer() as tm:
            for _ in range(3):
                _csx.check(lib.csx_gaxpy(hA, hx, hy, modes[name]), "gaxpy " + name)
        trial[name] = {"ms": tm.ms / 3, "prepare_s": prep}
    chosen = min(trial, key=lambda k: trial[k]["ms"])
    chosen = comm.broadcast_object(chosen)  # all ranks run the kernel rank 0 measured fastest
    mode = modes[chosen]

    _csx.check(lib.csx_vec_fill(hy, 0.0), "fill")
    for _ in range(args.warmup):
        _csx.check(lib.csx_gaxpy(hA, hx, hy, mode), "gaxpy")
    barrier()
    t0 = time.perf_counter()
    _csx.check(lib.csx_timer_start(), "timer")
    for _ in range(args.steps):
        _csx.check(lib.csx_gaxpy(hA, hx, hy, mode), "gaxpy")
    ev_ms = _csx.C.c_double(0.0)
    _csx.check(lib.csx_timer_stop(ev_ms), "timer")
    barrier()
    wall = max_over_ranks(time.perf_counter() - t0)
    step_ms_events = max_over_ranks(ev_ms.value / args.steps)
    by = gaxpy_bytes(n, n, nnz)
    value = by * args.steps * world / wall / 1e9
    achieved = by / (step_ms_events * 1e-3) / 1e9
    tr = measured_traffic({"tiled": "k_gaxpy_tiled<0,", "wave": "k_gaxpy_rows", "atomic": "k_gaxpy_atomic"}[chosen],
                          n=n, nnz=nnz, kernel="gaxpy_" + chosen)

    # parity of the timed kernel ON THE TIMED MATRIX, after the timed region: one pass of the chosen kernel from y = 0
    # against one pass of the reference-order kernel (GAXPY_EXACT: thread per row, terms in ascending (column, position)
    # order, multiply and add rounded separately -- bit-identical to csparse.py:1210-1212, tests/test_gpu_parity.py).
    hy1, hy2 = _csx.new_handle(), _csx.new_handle()
    _csx.check(lib.csx_vec_alloc(n, hy1), "vec_alloc")
    _csx.check(lib.csx_vec_alloc(n, hy2), "vec_alloc")
    _csx.check(lib.csx_gaxpy(hA, hx, hy1, mode), "gaxpy check")
    _csx.check(lib.csx_gaxpy(hA, hx, hy2, cs.GAXPY_EXACT), "gaxpy exact")
    ya, ye = np.empty(n), np.empty(n)
    _csx.check(lib.csx_vec_download(hy1, _csx.pd(ya), n), "vec_download")
    _csx.check(lib.csx_vec_download(hy2, _csx.pd(ye), n), "vec_download")
    for h in (hy1, hy2):
        _csx.free(h)
    nzr = ye > 0                                          # all-positive data; a row may be empty under the uniform draw
    parity_err = float(np.max(np.abs(ya[nzr] - ye[nzr]) / ye[nzr])) if nzr.any() else 0.0
    parity_ok = bool(np.all(np.isfinite(ye)) and np.all(ya[~nzr] == 0) and parity_err < 1e-12)
    if not (parity_ok or os.environ.get("CSX_TILED_VARIANT")):
        raise SystemExit("bench.py: the timed kernel disagrees with the reference-order kernel on the timed matrix "
                         "(max relative error %.3e)" % parity_err)
    del ya, ye

    try:
        rccl_ranks = comm.info()
    except Exception as e:                                # reported, never fatal
        rccl_ranks = {"error": "%s: %s" % (type(e).__name__, e)}
    key_bytes = _csx.C.c_int(0)
    _csx.check(lib.csx_gaxpy_plan_info(hA, None, None, key_bytes), "plan_info")
    shape, shape_ms = _csx.C.c_int(-1), (_csx.C.c_double * 4)()
    if chosen == "tiled":
        _csx.check(lib.csx_gaxpy_plan_shape(hA, shape, shape_ms), "plan_shape")
    out = {
        "metric": "cs_gaxpy achieved HBM GB/s (algorithmic bytes / time), 5M x 5M CSC, 64 nnz/col",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        # what the exchange transport itself says (csx_comm_info of the communicator libcsx made, and a sum of ones over it):
        # at N > 1 "uses_rccl" must be true and "ranks_counted" == n_gpus
        "rccl_ranks": rccl_ranks,
        "config": {"workload": "cs_gaxpy y += A x on G-rand: %d x %d CSC, %s, int32 indices, fp64 values; one "
                               "independent matrix per GPU" % (n, n, gen_words[args.gen]), "row_draw": args.gen,
                   "n": n, "nnz": nnz, "kernel": "gaxpy_" + chosen, "plan_key_bytes": key_bytes.value,
                   "algorithmic_bytes_per_step": by,
                   "parallelism": "independent matrices, 1 per GPU, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     # bytes per launch crossing the L2 -> fabric boundary, read from the committed counter summary
                     # of this kernel at this size (measured_traffic); null if there is none
                     "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                     "step_ms_hip_events": round(step_ms_events, 4)},
        "gaxpy_trials_ms": {k: round(v["ms"], 4) for k, v in trial.items()},
        "gaxpy_prepare_s": {k: round(v["prepare_s"], 3) for k, v in trial.items()},
        "parity_on_the_timed_matrix": {"against": "GAXPY_EXACT (reference summation order, csparse.py:1210-1212)",
                                       "max_rel_err": parity_err, "tolerance": 1e-12, "ok": parity_ok},
    }
    if shape.value >= 0:                                  # only when the plan timed its launch shapes (gaxpy.tune_shape)
        out["config"]["plan_launch_shape"] = {"picked": ["4x5", "2x10", "8x4", "2x8"][shape.value],
                                              "ms_when_the_plan_was_built": {k: round(v, 4) for k, v in
                                                                             zip(("4x5", "2x10", "8x4", "2x8"), shape_ms)}}
    # the spread of the headline with the device's state: three more trials of the same K steps, each behind a different
    # kernel (a transposing sort, the exact kernel, nothing), HIP-event timed.  `value` stays the contract's timed region.
    spread = []
    for k in range(3):
        if k == 0:
            _csx.check(lib.csx_gaxpy(hA, hx, hy, cs.GAXPY_EXACT), "gaxpy exact")
        elif k == 1:
            hT = _csx.new_handle()
            _csx.check(lib.csx_transpose(hA, 1, hT), "transpose")
            _csx.free(hT)
        with _csx.Timer() as tm:
            for _ in range(args.steps):
                _csx.check(lib.csx_gaxpy(hA, hx, hy, mode), "gaxpy")
        spread.append(max_over_ranks(tm.ms / args.steps))
    allt = sorted(spread + [step_ms_events])
    med = allt[len(allt) // 2] if len(allt) % 2 else 0.5 * (allt[len(allt) // 2 - 1] + allt[len(allt) // 2])
    out["headline_trials"] = {"ms_per_step": [round(step_ms_events, 4)] + [round(t, 4) for t in spread],
                              "median_ms": round(med, 4), "median_frac_of_peak": round(by / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "first = the timed region of `value`; then behind the exact kernel, behind a transpose, back to back"}
    _csx.free(hA)
    deadline = Deadline(args.extras_deadline, rank)
    deadline.arm(out)
    preflight = None
    if world > 1 or args.force_sharded:
        # every exchange leg at a small size first: a hang at N > 1 is then attributed to a leg (see exchange_preflight)
        preflight = exchange_preflight(args, lib, cs, comm, barrier, deadline, out)
    # the same kernel on the other row draw (not part of `value`)
    other = "stratified" if args.gen == "uniform" else "uniform"
    hA2 = _csx.new_handle()
    _csx.check(gens[other](n, per_col, 20240601 + 1 + rank, hA2), "gen_grand")
    _csx.check(lib.csx_gaxpy_prepare(hA2, mode), "prepare")
    for _ in range(max(1, args.warmup)):
        _csx.check(lib.csx_gaxpy(hA2, hx, hy, mode), "gaxpy")
    barrier()
    with _csx.Timer() as tm:
        for _ in range(args.steps):
            _csx.check(lib.csx_gaxpy(hA2, hx, hy, mode), "gaxpy")
    ms2 = max_over_ranks(tm.ms / args.steps)
    out["gaxpy_grand_other_row_draw"] = {"row_draw": other, "workload": gen_words[other], "ms_per_step": round(ms2, 4),
                                         "achieved_GBps_per_gpu": round(by / (ms2 * 1e-3) / 1e9, 2),
                                         "frac_of_peak": round(by / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    _csx.free(hA2)
    _csx.free(hx)
    _csx.free(hy)
    deadline.checkpoint(out)

    # ---- cs_gaxpy on G-spd (block-diagonal, best-case locality), same size ----
    if not args.skip_gspd:
        bs = 64
        nb = n // bs
        hB = _csx.new_handle()
        _csx.check(lib.csx_gen_gspd(nb, bs, 20240601 + 5, hB), "gen_gspd")  # the same matrix on every rank: its right-hand sides are what is sharded
        hx, hy = _csx.new_handle(), _csx.new_handle()
        _csx.check(lib.csx_gen_vec(nb * bs, 9, 0.5, 1.5, hx), "gen_vec")
        _csx.check(lib.csx_vec_alloc(nb * bs, hy), "vec_alloc")
        _csx.check(lib.csx_gaxpy_prepare(hB, cs.GAXPY_WAVE), "prepare")
        for _ in range(max(1, args.warmup)):
            _csx.check(lib.csx_gaxpy(hB, hx, hy, cs.GAXPY_WAVE), "gaxpy")
        barrier()
        with _csx.Timer() as tm:
            for _ in range(args.steps):
                _csx.check(lib.csx_gaxpy(hB, hx, hy, cs.GAXPY_WAVE), "gaxpy")
        ms = max_over_ranks(tm.ms / args.steps)
        byb = gaxpy_bytes(nb * bs, nb * bs, nb * bs * bs)
        gbs = byb / (ms * 1e-3) / 1e9
        out["gaxpy_gspd"] = {"workload": "cs_gaxpy on G-spd (the 5M x 5M 64-nnz/row SPD matrix of the cholsol leg): "
                                         "%d dense %dx%d SPD blocks (n=%d)" % (nb, bs, bs, nb * bs),
                             "ms_per_step": round(ms, 4), "achieved_GBps_per_gpu": round(gbs, 2),
                             "whole_job_GBps": round(gbs * world, 2), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4),
                             "kernel": "gaxpy_wave (k_gaxpy_rows4: 16-byte loads on the row-major copy)",
                             "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(gbs / HBM_PEAK_GBS, 4),
                                          "traffic": (measured_traffic("k_gaxpy_rows4", n=n, nnz=nnz) or {}).get("bytes")}}
        if not args.skip_cholsol:
            extra = cholsol_section(args, lib, cs, comm, hB, nb, bs, barrier, max_over_ranks, preflight)
            if extra:
                out["cholsol"] = extra
        _csx.free(hB)
        _csx.free(hx)
        _csx.free(hy)
        deadline.checkpoint(out)

    if (world > 1 or args.force_sharded) and not args.skip_sharded:
        legs = (preflight or {}).get("legs", {})
        if not legs.get("sharded_gaxpy_reduce_scatter", {}).get("ok", True):
            out["gaxpy_one_matrix_column_sharded"] = {"skipped": "the reduce-scatter form failed the exchange preflight"}
        else:
            p2p_ok = legs.get("sharded_gaxpy_row_pieces_p2p", {}).get("ok", True)
            out["gaxpy_one_matrix_column_sharded"] = sharded_spmv_section(args, lib, cs, comm, barrier, max_over_ranks,
                                                                          forms=(0, 1) if p2p_ok else (0,))
            if not p2p_ok:
                out["gaxpy_one_matrix_column_sharded"]["fallback"] = ("row_pieces_overlapped_p2p failed its row-for-row check "
                                                                      "in the preflight: only spmv_then_reduce_scatter was run")

    if rank == 0 and world == 1 and not args.skip_configs:
        # BASELINE's other configs (2: bcsstk16 SpMV, 3: cs_lusol on W, 4: A*A' on S) and cs_transpose at the headline
        # size, each checking itself; never part of `value`.  bench_configs.py is the stand-alone form.
        import bench_configs as bc
        bc.SKIP_CPU = args.skip_cpu
        other = {}
        for name, fn in (("config2_gaxpy_bcsstk16", bc.config2), ("config3_lusol_W", bc.config3),
                         ("transpose_grand_5M", bc.transpose_grand), ("config4_multiply_S", bc.config4),
                         ("cholsol_connected", bc.cholsol_connected), ("lu_connected", bc.lu_connected)):
            try:
                other[name] = fn()
            except Exception as e:                        # never take the headline down with it
                other[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["other_configs"] = other
        deadline.checkpoint(out)
    if rank == 0 and world == 1 and not args.skip_cpu:
        py, c = cpu_baseline(args.cpu_n, per_col, args.cpu_seconds, args.gen)
        out["cpu_baseline"] = py
        out["cpu_baseline_c"] = c
    deadline.emit(out)
    comm.close()
    deadline.timer.cancel()


def exchange_preflight(args, lib, cs, comm, barrier, deadline, out):
    """Every exchange leg of the N > 1 run at a SMALL size (n = 100 032, 8 right-hand sides) before the full-size ones,
    pass / fail per leg.  The first RCCL run on more than one GPU is the driver's scaling bench: if a leg hangs there --
    a rank failing inside a collective leaves the others waiting until --extras-deadline -- the line rank 0 then prints
    names the leg that was running ("running"), instead of the deadline eating the attribution; a leg that completes but
    fails its check is reported and its full-size form is skipped or replaced (the p2p sharded SpMV falls back to the
    reduce-scatter form).  Legs: factor broadcast (csx_comm_bcast_csc) with the receiver's solve reproduced bit for bit;
    right-hand-side scatter; solution gather; the column-sharded SpMV in both forms, row for row against the unsharded
    product; cholsol_factor(A).solve(B, comm=...) against the unsharded solve."""
    import numpy as np
    import _csx
    import shard
    rank, world = comm.rank, comm.world
    res = {"world": world, "backend": comm.backend, "n": None, "legs": {}, "running": None}
    out["exchange_preflight"] = res

    def leg(name, fn):
        res["running"] = name
        deadline.checkpoint(out)
        t0 = time.perf_counter()
        err = None
        try:
            ok = fn()
        except Exception as e:                                    # reported, never fatal
            ok, err = False, "%s: %s" % (type(e).__name__, e)
        # EVERY rank takes part in the vote, also the one whose fn() raised: a rank that skipped it would enter the next
        # leg's collectives one step ahead of the others
        ok = bool(comm.sum(1.0 if ok else 0.0) == world)
        res["legs"][name] = {"ok": ok, "s": round(time.perf_counter() - t0, 4)}
        if err:
            res["legs"][name]["error"] = err
        res["running"] = None
        deadline.checkpoint(out)

    nb, bs, k = 1563, 64, 8
    n = nb * bs
    res["n"], res["nrhs_per_gpu"] = n, k
    hB = _csx.new_handle()
    _csx.check(lib.csx_gen_gspd(nb, bs, 20240601 + 5, hB), "gen_gspd")
    A = cs._from_device(hB, lambda z: max(z, 1))
    A._pinned = True
    F = cs.cholsol_factor(A, exact=True)

    def own_solution(col0):
        h = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, k, col0, h), "gen_rhs")
        _csx.check(lib.csx_cholsol_solve(F.plan_handle, h, k), "cholsol_solve")
        a = np.empty(n * k)
        _csx.check(lib.csx_vec_download(h, _csx.pd(a), n * k), "vec_download")
        _csx.free(h)
        return a

    X_own = own_solution(rank * k)

    def bcast_factor():
        hL2 = comm.bcast_csc(F.L._dev.handle if rank == 0 else None, 0)
        plan2, h = _csx.new_handle(), _csx.new_handle()
        _csx.check(lib.csx_cholsol_plan(hL2, None, plan2), "cholsol_plan")
        _csx.check(lib.csx_gen_rhs(n, k, rank * k, h), "gen_rhs")
        _csx.check(lib.csx_cholsol_solve(plan2, h, k), "cholsol_solve")
        a = np.empty(n * k)
        _csx.check(lib.csx_vec_download(h, _csx.pd(a), n * k), "vec_download")
        for q in (plan2, h) + ((hL2,) if rank != 0 else ()):       # the root's handle is its own factor
            _csx.free(q)
        return a.tobytes() == X_own.tobytes()

    state = {}

    def scatter():
        mine = cs.dvec(n * k)
        src = None
        if rank == 0:
            src = cs.dvec(n * k * world)
            for r in range(world):
                h, slot = _csx.new_handle(), _csx.new_handle()
                _csx.check(lib.csx_gen_rhs(n, k, r * k, h), "gen_rhs")
                _csx.check(lib.csx_vec_wrap(_csx.C.c_void_p(src.device_ptr() + 8 * n * k * r), n * k, slot), "vec_wrap")
                _csx.check(lib.csx_vec_copy(h, slot), "vec_copy")
                _csx.free(slot)
                _csx.free(h)
        comm.scatter_vec_blocks(src.handle if rank == 0 else None, mine.handle, n * k, 0)
        h = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, k, rank * k, h), "gen_rhs")
        want = np.empty(n * k)
        _csx.check(lib.csx_vec_download(h, _csx.pd(want), n * k), "vec_download")
        _csx.free(h)
        state["src"], state["mine"] = src, mine
        return mine.numpy().tobytes() == want.tobytes()

    def gather():
        import hashlib
        mine, src = state["mine"], state.get("src")
        _csx.check(lib.csx_cholsol_solve(F.plan_handle, mine.handle, k), "cholsol_solve")
        digests = comm.all_gather_object(hashlib.sha256(mine.numpy().tobytes()).hexdigest())
        comm.gather_vec_blocks(mine.handle, src.handle if rank == 0 else None, n * k, 0)
        ok = True
        if rank == 0:
            allb = src.numpy().reshape(world, n * k)
            ok = all(hashlib.sha256(allb[r].tobytes()).hexdigest() == digests[r] for r in range(world))
        return ok and mine.numpy().tobytes() == X_own.tobytes()

    def sharded_solve():
        K = k * world - (1 if world > 1 else 0)                   # an uneven last block at N > 1
        hfull = _csx.new_handle()
        _csx.check(lib.csx_gen_rhs(n, K, 0, hfull), "gen_rhs")
        B = cs.dvec(n, K, _handle=hfull)
        ref = None
        if rank == 0:
            R = B.copy()
            assert F.solve(R)
            ref = R.numpy().tobytes()
        assert F.solve(B if rank == 0 else None, comm=comm, nrhs=K)
        return True if rank != 0 else B.numpy().tobytes() == ref

    def sharded_gaxpy(how):
        def run():
            m = 100003                                            # uneven column blocks and row chunks
            first, count = shard.strong_block(rank, world, m)
            hFull, hA, hxF, hyR = (_csx.new_handle() for _ in range(4))
            _csx.check(lib.csx_gen_grand(m, 16, 20240601 + 78, hFull), "gen_grand")
            _csx.check(lib.csx_csc_col_block(hFull, first, count, hA), "col_block")
            _csx.check(lib.csx_gen_vec(m, 7, 0.5, 1.5, hxF), "gen_vec")
            _csx.check(lib.csx_vec_alloc(m, hyR), "vec_alloc")
            _csx.check(lib.csx_gaxpy(hFull, hxF, hyR, cs.GAXPY_EXACT), "gaxpy exact")
            yref = np.empty(m)
            _csx.check(lib.csx_vec_download(hyR, _csx.pd(yref), m), "vec_download")
            xs = np.empty(m)
            _csx.check(lib.csx_vec_download(hxF, _csx.pd(xs), m), "vec_download")
            sg = shard.ShardedGaxpy(comm, hA, m)
            f, c = sg.rows()
            y, dx = cs.dvec(sg.chunk), cs.dvec(np.ascontiguousarray(xs[first:first + count]))
            sg.run(dx.handle, y.handle, how)
            got, want = y.numpy()[:c], yref[f:f + c]
            nz = want > 0
            ok = bool(np.all(got[~nz] == 0) and (not nz.any() or np.max(np.abs(got[nz] - want[nz]) / want[nz]) < 1e-12))
            sg.free()
            for h in (hFull, hA, hxF, hyR):
                _csx.free(h)
            return ok
        return run

    leg("factor_broadcast", bcast_factor)
    leg("rhs_scatter", scatter)
    leg("solution_gather", gather)
    leg("sharded_solve_api", sharded_solve)
    leg("sharded_gaxpy_reduce_scatter", sharded_gaxpy(0))
    leg("sharded_gaxpy_row_pieces_p2p", sharded_gaxpy(1))
    res["all_ok"] = all(v.get("ok") for v in res["legs"].values())
    del res["running"]
    state.clear()
    del F, A
    return res


def sharded_spmv_section(args, lib, cs, comm, barrier, max_over_ranks, forms=(0, 1)):
    """ONE n x n G-rand matrix sharded by columns over the ranks (second bullet): rank r owns
    columns [r n/W, (r+1) n/W) and the matching slice of x; y = sum of the ranks' partial products, rank r keeping
    rows [r ceil(n/W), ...).  Both exchange forms of csx_gaxpy_sharded are timed: one SpMV + one RCCL reduce-scatter,
    and row pieces leaving over direct links while the next piece is computed.  The result is checked ROW FOR ROW
    against the unsharded cs_gaxpy of the whole matrix (exact-order kernel).  Strong scaling; reported next to
    (never instead of) the headline.  Every stage is guarded: a failure is reported as {"error": ...}."""
    import numpy as np
    import _csx
    import shard
    C = _csx.C
    rank, world = comm.rank, comm.world
    n, per_col = args.n, args.per_col
    try:
        first, count = shard.strong_block(rank, world, n)
        hFull = _csx.new_handle()
        _csx.check(lib.csx_gen_grand(n, per_col, 20240601 + 77, hFull), "gen_grand")     # same matrix on every rank
        hA = _csx.new_handle()
        _csx.check(lib.csx_csc_col_block(hFull, first, count, hA), "col_block")
        hxFull = _csx.new_handle()
        _csx.check(lib.csx_gen_vec(n, 7, 0.5, 1.5, hxFull), "gen_vec")
        # the unsharded answer, reference summation order, this rank's rows of it
        r0, rc = shard.row_chunk(rank, world, n)
        chunk = (n + world - 1) // world
        hyRef = _csx.new_handle()
        _csx.check(lib.csx_vec_alloc(n, hyRef), "vec_alloc")
        _csx.check(lib.csx_gaxpy(hFull, hxFull, hyRef, cs.GAXPY_EXACT), "gaxpy exact")
        yref = np.empty(n)
        _csx.check(lib.csx_vec_download(hyRef, _csx.pd(yref), n), "vec_download")
        yref = yref[r0:r0 + rc]
        _csx.free(hyRef)
        _csx.free(hFull)
        ptr, ln = C.c_void_p(), C.c_int64()
        _csx.check(lib.csx_vec_ptr(hxFull, ptr, ln), "vec_ptr")
        hx = _csx.new_handle()
        _csx.check(lib.csx_vec_wrap(C.c_void_p(ptr.value + 8 * first), count, hx), "vec_wrap")
        hy = _csx.new_handle()
        _csx.check(lib.csx_vec_alloc(chunk, hy), "vec_alloc")
        sg = shard.ShardedGaxpy(comm, hA, n)
        assert sg.rows() == (r0, rc)
        by = gaxpy_bytes(n, n, n * per_col)
        res = {"workload": "one %d x %d G-rand matrix, columns sharded over %d GPU(s), partial y summed over %s "
                           "(%d MB of partial sums per rank)" % (n, n, world, comm.backend, chunk * world * 8 // 1000000),
               "scaling": "strong", "rows_owned_per_rank": chunk, "forms": {}}
        steps = min(args.steps, 20)
        for how, name in ((0, "spmv_then_reduce_scatter"), (1, "row_pieces_overlapped_p2p")):
            if how not in forms:
                continue
            _csx.check(lib.csx_vec_fill(hy, 0.0), "fill")
            sg.run(hx, hy, how)
            got = np.empty(chunk)
            _csx.check(lib.csx_vec_download(hy, _csx.pd(got), chunk), "vec_download")
            got = got[:rc]
            nz = yref > 0
            err = float(np.max(np.abs(got[nz] - yref[nz]) / yref[nz])) if nz.any() else 0.0
            ok = bool(err < 1e-12 and np.all(got[~nz] == 0))
            ok = bool(comm.sum(1.0 if ok else 0.0) == world)
            err = comm.max(err)
            for _ in range(max(1, args.warmup)):
                sg.run(hx, hy, how)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                sg.run(hx, hy, how)
            barrier()
            wall = max_over_ranks(time.perf_counter() - t0) / steps
            res["forms"][name] = {"ms_per_spmv": round(wall * 1e3, 4),
                                  "whole_job_algorithmic_GBps": round(by / wall / 1e9, 2),
                                  "rows_equal_unsharded_exact_order": ok, "max_rel_err": err}
        with _csx.Timer() as tm:                          # the block's kernel alone, no exchange
            for _ in range(steps):
                _csx.check(lib.csx_gaxpy(hA, hx, hy_scratch(lib, _csx, n), cs.GAXPY_AUTO), "gaxpy")
        res["ms_kernel_of_the_block_alone"] = round(max_over_ranks(tm.ms / steps), 4)
        res["rows_equal_unsharded"] = all(f["rows_equal_unsharded_exact_order"] for f in res["forms"].values())
        sg.free()
        for h in (hA, hx, hy, hxFull) + tuple(_SCRATCH.values()):
            _csx.free(h)
        _SCRATCH.clear()
        return res
    except Exception as e:                                # never take the headline down with it
        return {"error": "%s: %s" % (type(e).__name__, e)}


_SCRATCH = {}


def hy_scratch(lib, _csx, n):
    """One full-length scratch y (kept for the section)."""
    if n not in _SCRATCH:
        h = _csx.new_handle()
        _csx.check(lib.csx_vec_alloc(n, h), "vec_alloc")
        _SCRATCH[n] = h
    return _SCRATCH[n]


def cholsol_section(args, lib, cs, comm, hB, nb, bs, barrier, max_over_ranks, preflight=None):
    """Batched cs_cholsol on G-spd: factor once per rank, solve nrhs right-hand sides per GPU."""
    import numpy as np
    import _csx
    C = _csx.C
    rank, world = comm.rank, comm.world
    n = nb * bs
    nnz = n * bs
    k = args.nrhs
    # The product's flow for a batch (round 5): ONE library call -- cs_schol + cs_chol + the solve plan, S never leaving the device
    # (csx_cholsol_factor; csparse.cholsol_factor(A) / cs_cholsol(0, A, b) go through it).  Called three times (factor and plan
    # freed in between, as a refactorisation loop would): every call is reported, schol_chol_plan_total is their median.
    fused_ms, fused_info = [], None
    hL = plan = None
    for rep in range(3):
        if hL is not None:
            _csx.free(plan)
            _csx.free(hL)
        hL, plan = _csx.new_handle(), _csx.new_handle()
        # (a successful call leaves its analysis -- tree, counts, block list -- on the matrix for the next factorisation of it;
        # dropped here, so that each of the three calls is the whole pipeline: analysis included)
        _csx.check(lib.csx_csc_invalidate(hB), "csc_invalidate")
        _csx.sync()
        t0 = time.perf_counter()
        _csx.check(lib.csx_cholsol_factor(hB, 0, hL, plan), "cholsol_factor")
        _csx.sync()
        fused_ms.append((time.perf_counter() - t0) * 1e3)
        fpath, fa, fn, fc = C.c_int32(-1), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        _csx.check(lib.csx_cholsol_factor_info(fpath, fa, fn, fc), "cholsol_factor_info")
        fused_info = {"path": fpath.value, "analysis_ms": round(fa.value, 3), "numeric_kernel_ms": round(fn.value, 4),
                      "call_ms_library_clock": round(fc.value, 3)}
    t_fused = sorted(fused_ms)[1] * 1e-3
    # a REfactorisation (same pattern: the analysis found on the matrix): what a loop over new values pays per factor
    refactor_ms = []
    for rep in range(3):
        hLr, planr = _csx.new_handle(), _csx.new_handle()
        _csx.sync()
        t0 = time.perf_counter()
        _csx.check(lib.csx_cholsol_factor(hB, 0, hLr, planr), "cholsol_factor")
        _csx.sync()
        refactor_ms.append((time.perf_counter() - t0) * 1e3)
        _csx.free(planr)
        _csx.free(hLr)
    chol_path, chol_kernel_ms = C.c_int32(-1), C.c_double(0.0)
    _csx.check(lib.csx_chol_info(chol_path, chol_kernel_ms), "chol_info")      # (of the default kernel: before the opt-in one runs)
    # the same call with "chol.exact" = 0 (opt-in): fused multiply-adds and refined reciprocal square roots in the block kernel,
    # L.x equal to the default's to rounding (checked on the spot: a digest would differ, so a sample of L.x is compared)
    relaxed_chol = {}
    try:
        def lx_sample(h):
            px = _csx.C.c_void_p()
            _csx.check(lib.csx_csc_ptrs(h, None, None, _csx.C.byref(px)), "csc_ptrs")
            hv = _csx.new_handle()
            _csx.check(lib.csx_vec_wrap(px, 1 << 20, hv), "vec_wrap")
            a = np.empty(1 << 20)
            _csx.check(lib.csx_vec_download(hv, _csx.pd(a), 1 << 20), "vec_download")
            _csx.free(hv)
            return a
        lx_exact = lx_sample(hL)
        with _csx.option("chol.exact", 0):
            ms_r = []
            for rep in range(3):
                hL3, plan3 = _csx.new_handle(), _csx.new_handle()
                _csx.sync()
                t0 = time.perf_counter()
                _csx.check(lib.csx_cholsol_factor(hB, 0, hL3, plan3), "cholsol_factor")
                _csx.sync()
                ms_r.append((time.perf_counter() - t0) * 1e3)
                fpath, fa, fn, fc = C.c_int32(-1), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
                _csx.check(lib.csx_cholsol_factor_info(fpath, fa, fn, fc), "cholsol_factor_info")
                if rep == 2:
                    lx_r = lx_sample(hL3)
                _csx.free(plan3)
                _csx.free(hL3)
        relaxed_chol = {"option": "chol.exact = 0 (opt-in)", "numeric_kernel_ms": round(fn.value, 4), "fused_calls_ms": [round(v, 3) for v in ms_r],
                        "Lx_max_rel_diff_vs_default_first_1M_entries": float(np.max(np.abs(lx_r - lx_exact) / np.abs(lx_exact)))}
    except Exception as e:                                # a comparison figure: never take the section down
        relaxed_chol = {"error": "%s: %s" % (type(e).__name__, e)}
    nL_m, nL_n, nL_nnz, nL_hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
    _csx.check(lib.csx_csc_info(hL, nL_m, nL_n, nL_nnz, nL_hv), "csc_info")
    lnz = nL_nnz.value
    # The same through round 4's three calls (csx_schol hands parent / cp to the host, csx_chol takes them back and compares,
    # csx_cholsol_plan re-reads L): kept as the C ABI's separate steps and timed beside the fused call.
    sep = {}
    try:
        t0 = time.perf_counter()
        parent = np.empty(n, dtype=np.int32)
        cp = np.empty(n + 1, dtype=np.int32)
        _csx.check(lib.csx_schol(hB, _csx.pi(parent), _csx.pi(cp)), "schol")
        sep["symbolic_cs_schol"] = round(time.perf_counter() - t0, 4)
        t0 = time.perf_counter()
        hL2 = _csx.new_handle()
        _csx.check(lib.csx_chol(hB, _csx.pi(parent), _csx.pi(cp), None, hL2), "chol")
        _csx.sync()
        sep["numeric_cs_chol"] = round(time.perf_counter() - t0, 4)
        t0 = time.perf_counter()
        plan2 = _csx.new_handle()
        _csx.check(lib.csx_cholsol_plan(hL2, None, plan2), "cholsol_plan")
        _csx.sync()
        sep["solve_plan"] = round(time.perf_counter() - t0, 4)
        t0 = time.perf_counter()
        _csx.check(lib.csx_cholsol_set_order(plan2, 0), "cholsol_set_order")
        _csx.sync()
        sep["matrix_core_fragments"] = round(time.perf_counter() - t0, 4)
        sep["total"] = round(sum(sep.values()), 4)
        assert int(cp[n]) == lnz
        _csx.free(plan2)
        _csx.free(hL2)
        del parent, cp
    except Exception as e:                                # a comparison figure: never take the section down
        sep["error"] = "%s: %s" % (type(e).__name__, e)
    # the reference's own driver is exact throughout -- cs_cholsol(order, A, b): the same one call with exact = 1 (no W tiles; the
    # plan is the block list, the exact kernel reads L.x itself), whole pipeline each time (analysis dropped), then ONE batch
    exact_flow = {}
    try:
        ef_ms, es_ms = [], []
        for rep in range(3):
            hLe, plane = _csx.new_handle(), _csx.new_handle()
            _csx.check(lib.csx_csc_invalidate(hB), "csc_invalidate")
            _csx.sync()
            t0 = time.perf_counter()
            _csx.check(lib.csx_cholsol_factor(hB, 1, hLe, plane), "cholsol_factor")
            _csx.sync()
            ef_ms.append((time.perf_counter() - t0) * 1e3)
            hRe = _csx.new_handle()
            _csx.check(lib.csx_gen_rhs(n, k, rank * k, hRe), "gen_rhs")
            _csx.sync()
            t0 = time.perf_counter()
            _csx.check(lib.csx_cholsol_solve(plane, hRe, k), "cholsol_solve")
            _csx.sync()
            es_ms.append((time.perf_counter() - t0) * 1e3)
            _csx.free(hRe)
            _csx.free(plane)
            _csx.free(hLe)
        tf_e, ts_e = sorted(ef_ms)[1], sorted(es_ms)[1]
        exact_flow = {"factor_calls_ms": [round(v, 3) for v in ef_ms], "first_batch_ms": [round(v, 3) for v in es_ms],
                      "end_to_end_solves_per_s_per_gpu": round(k / ((tf_e + ts_e) * 1e-3), 1),
                      "note": "csx_cholsol_factor(exact = 1) + ONE batch of %d, medians of three; every bit the reference's" % k}
    except Exception as e:                                # a comparison figure: never take the section down
        exact_flow = {"error": "%s: %s" % (type(e).__name__, e)}
    # exact (default) order first: bit-identical to cs_lsolve + cs_ltsolve, substitution kernels
    hR0 = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, rank * k, hR0), "gen_rhs")
    t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_set_order(plan, 1), "cholsol_set_order")
    _csx.check(lib.csx_cholsol_solve(plan, hR0, k), "cholsol_solve")    # (until late in round 5 the substitution programs were cut out of L.x here; the kernel now reads L.x)
    _csx.sync()
    t_first_exact = time.perf_counter() - t0
    with _csx.Timer() as tm0:
        for _ in range(5):
            _csx.check(lib.csx_cholsol_solve(plan, hR0, k), "cholsol_solve")
    ms_exact = max_over_ranks(tm0.ms / 5)
    _csx.free(hR0)
    # the timed leg: rounding-equal order (1e-10 budget of BASELINE.json), dense blocks on the matrix cores
    t0 = time.perf_counter()
    _csx.check(lib.csx_cholsol_set_order(plan, 0), "cholsol_set_order")
    _csx.sync()
    t_plan_mfma = time.perf_counter() - t0
    fused, trees, mx = C.c_int32(), C.c_int32(), C.c_int32()
    _csx.check(lib.csx_cholsol_info(plan, fused, trees, mx), "cholsol_info")
    # this rank's block of right-hand sides: columns [rank*k, (rank+1)*k) of the global B
    hR = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, rank * k, hR), "gen_rhs")
    for _ in range(max(1, args.warmup)):
        _csx.check(lib.csx_cholsol_solve(plan, hR, k), "cholsol_solve")
    steps = min(args.steps, 40)
    barrier()
    t0 = time.perf_counter()
    _csx.check(lib.csx_timer_start(), "timer")
    for _ in range(steps):
        _csx.check(lib.csx_cholsol_solve(plan, hR, k), "cholsol_solve")
    ev = C.c_double(0.0)
    _csx.check(lib.csx_timer_stop(ev), "timer")
    barrier()
    wall = max_over_ranks(time.perf_counter() - t0)
    ms = max_over_ranks(ev.value / steps)
    fused_bytes = 12 * lnz + 4 * (n + 1) + 16 * n * k  # L read once, B read once, X written once
    # cs_chol: read the upper triangle of A (index + value), write L (index + value)
    nnz_triu = n * (bs + 1) // 2
    chol_bytes = 12 * nnz_triu + 12 * lnz
    t_factor = t_fused
    out = {"workload": "batched cs_cholsol solve phase on G-spd (n=%d, lnz=%d): %d right-hand sides per GPU, "
                       "factor once per GPU, row-major n x k block" % (n, lnz, k),
           "solves_per_s": round(k * world * steps / wall, 1), "nrhs_per_gpu": k, "ms_per_batch": round(ms, 4),
           "path": {0: "level-scheduled", 1: "fused per-tree (X in LDS)", 2: "dense-block substitution (X in registers)",
                    3: "dense-block blocked TRSM, fp64 MFMA (X in registers)", 4: "supernodal schedule",
                    5: "small trees made dense by size class, fp64 MFMA"}.get(fused.value, str(fused.value)),
           "trees": trees.value, "max_tree": mx.value,
           "algorithmic_bytes_fused": fused_bytes,
           "achieved_GBps_per_gpu": round(fused_bytes / (ms * 1e-3) / 1e9, 2),
           "frac_of_peak": round(fused_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "roofline": {"bound": "hbm", "achieved": round(fused_bytes / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(fused_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": (measured_traffic("k_cholsol_", n=n, nrhs_per_gpu=k) or {}).get("bytes")},
           "survey_bytes_unfused": cholsol_bytes(lnz, n, k),
           "factor_s": {"schol_chol_plan_total": round(t_factor, 5),
                        "note": "csx_cholsol_factor: cs_schol + cs_chol + the solve plan (matrix-core operands included) in one "
                                "call, median of the three calls listed; the end-to-end figure below uses it",
                        "fused_calls_ms": [round(v, 3) for v in fused_ms], "fused_info_last_call": fused_info,
                        "refactor_calls_ms_analysis_kept_on_the_matrix": [round(v, 3) for v in refactor_ms],
                        "first_exact_solve_s": round(t_first_exact, 5),
                        "set_order_rounding_equal_s": round(t_plan_mfma, 5),
                        "separate_calls_round4_flow": sep},
           "chol_roofline": {"bound": "hbm", "kernel": {1: "k_chol_clique (forest of cliques: a block in the registers of a wave)",
                                                        2: "k_chol_clique (forest of small sparse trees)",
                                                        0: "general path (pattern of L + column kernels)"}.get(chol_path.value),
                             "algorithmic_bytes": chol_bytes, "numeric_kernel_ms": round(chol_kernel_ms.value, 4),
                             "achieved": round(chol_bytes / (chol_kernel_ms.value * 1e-3) / 1e9, 2) if chol_kernel_ms.value > 0 else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(chol_bytes / (chol_kernel_ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if chol_kernel_ms.value > 0 else None,
                             "note": "inside csx_cholsol_factor the kernel writes the inverses of the diagonal tiles (0.64 GB: all the matrix-core "
                                     "solve needs beyond L.x) and leaves L.i to be made on demand; algorithmic bytes stay SURVEY 8d's 12 nnz(triu A) + 12 lnz",
                             "factor_call_ms": round(t_factor * 1e3, 3),
                             "frac_whole_call": round(chol_bytes / t_factor / 1e9 / HBM_PEAK_GBS, 4),
                             "traffic": (measured_traffic("k_chol_clique", n=n) or {}).get("bytes"),
                             "rounding_equal_kernel": dict(relaxed_chol, **({"achieved": round(chol_bytes / (relaxed_chol["numeric_kernel_ms"] * 1e-3) / 1e9, 2),
                                                                             "frac": round(chol_bytes / (relaxed_chol["numeric_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                                                            if relaxed_chol.get("numeric_kernel_ms") else {}))},
           "end_to_end_solves_per_s_per_gpu": round(k / (t_factor + ms * 1e-3), 1),
           # BASELINE config 5 as stated: 1 024 right-hand sides on 8 GPUs = 128 per GPU, ONE batch; every rank factors for itself
           # (csx_cholsol_factor, no exchange: DESIGN 5), its block generated in place; whole-job figures with the slowest rank's times
           "config5_as_stated": (lambda tf: {"total_rhs": k * world, "rhs_per_gpu": k, "gpus": world,
                                             "factor_s_slowest_rank": round(tf, 5), "batch_ms_slowest_rank": round(ms, 4),
                                             "end_to_end_solves_per_s_whole_job": round(k * world / (tf + ms * 1e-3), 1),
                                             "solve_phase_solves_per_s_whole_job": round(k * world / (ms * 1e-3), 1)})(max_over_ranks(t_factor)),
           "end_to_end_note": "cs_schol + cs_chol + plan + ONE batch of %d right-hand sides (the solve phase alone: solves_per_s)" % k,
           "exact_order": {"ms_per_batch": round(ms_exact, 4), "solves_per_s_per_gpu": round(k / (ms_exact * 1e-3), 1),
                           "note": "default order of every plan: bit-identical to cs_lsolve + cs_ltsolve",
                           "one_batch_end_to_end": exact_flow}}
    if rank == 0 and world == 1 and not args.skip_cpu:
        out["cpu_baseline"] = cpu_baseline_cholsol(args.cpu_chol_blocks, bs, args.cpu_seconds, lnz)
    bad = [name for name in ("factor_broadcast", "rhs_scatter", "solution_gather")
           if not ((preflight or {}).get("legs", {}).get(name, {}).get("ok", True))]
    if bad:
        out["exchange"] = {"skipped": "failed the exchange preflight at small size: %s" % ", ".join(bad)}
    elif world > 1 or args.force_sharded:
        try:
            out["exchange"] = exchange_section(args, lib, comm, hL, plan, n, lnz, t_factor,
                                               barrier, max_over_ranks)
        except Exception as e:                            # never take the headline down with it
            out["exchange"] = {"error": "%s: %s" % (type(e).__name__, e)}
    _csx.free(plan)
    _csx.free(hR)
    _csx.free(hL)
    return out


def exchange_section(args, lib, comm, hL, plan, n, lnz, t_factor_redundant, barrier, max_over_ranks):
    """The exchange steps of a batched cs_cholsol sharded by right-hand-side block, each timed on
    its own, none of them inside `value`, all through libcsx's own RCCL calls (csx_comm_*):
      1. factor once on rank 0 and ship L (csx_comm_bcast_csc: sizes, then p / i / x), against every rank
         factoring redundantly (what the timed solve leg does);
      2. the right-hand-side blocks leaving the root (rank r gets columns [r k, (r+1) k));
      3. the solution blocks gathered to the root.
    Every leg is checked: a rank rebuilds its solve plan from the RECEIVED factor and must reproduce its own
    solution bit for bit; received RHS blocks must equal the ones the rank would generate itself; the root
    checks the gathered blocks against each rank's own bytes (a digest)."""
    import hashlib
    import numpy as np
    import _csx
    C = _csx.C
    rank, world = comm.rank, comm.world
    k = args.exchange_nrhs or args.nrhs
    res = {"world": world, "backend": comm.backend}

    def block_np(h):
        a = np.empty(n * k)
        _csx.check(lib.csx_vec_download(h, _csx.pd(a), n * k), "vec_download")
        return a

    # ---- 1. factor once + broadcast ----
    hw = comm.bcast_csc(warm_csc(lib, _csx) if rank == 0 else None, 0)     # first collective sets up the rings: not timed
    _csx.free(hw)
    barrier()
    t0 = time.perf_counter()
    hL2 = comm.bcast_csc(hL if rank == 0 else None, 0)
    barrier()
    t_bcast = max_over_ranks(time.perf_counter() - t0)
    bytes_bcast = 4 * (n + 1) + 12 * lnz
    hR = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, rank * k, hR), "gen_rhs")
    _csx.check(lib.csx_cholsol_solve(plan, hR, k), "cholsol_solve")       # this rank's own factor
    X_own = block_np(hR)
    same, t_replan = 1.0, 0.0
    if rank != 0 or world == 1:
        if world == 1:                                    # a world of one: "receive" a copy so the re-plan is exercised
            hL2 = _csx.new_handle()
            m_, n_, z_, hv_ = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
            _csx.check(lib.csx_csc_info(hL, m_, n_, z_, hv_), "csc_info")
            _csx.check(lib.csx_csc_col_block(hL, 0, n_.value, hL2), "col_block")
        plan2, hR2 = _csx.new_handle(), _csx.new_handle()
        t1 = time.perf_counter()
        _csx.check(lib.csx_cholsol_plan(hL2, None, plan2), "cholsol_plan")
        _csx.check(lib.csx_cholsol_set_order(plan2, 0), "cholsol_set_order")   # same order as the rank's own plan
        _csx.sync()
        t_replan = time.perf_counter() - t1
        _csx.check(lib.csx_gen_rhs(n, k, rank * k, hR2), "gen_rhs")
        _csx.check(lib.csx_cholsol_solve(plan2, hR2, k), "cholsol_solve")  # the factor that came over the wire
        same = 1.0 if block_np(hR2).tobytes() == X_own.tobytes() else 0.0
        for h in (plan2, hR2, hL2):
            _csx.free(h)
    res["factor_once_broadcast"] = {
        "bytes": bytes_bcast, "s_broadcast": round(t_bcast, 5),
        "GBps_per_receiver": round(bytes_bcast / t_bcast / 1e9, 2) if t_bcast > 0 else None,
        "s_solve_plan_on_receiver": round(max_over_ranks(t_replan), 4),
        "s_factor_redundant_per_rank": round(max_over_ranks(t_factor_redundant), 4),
        "receivers_reproduce_own_solution_bit_for_bit": bool(comm.sum(same) == world)}

    # ---- 2. right-hand-side blocks leave the root ----
    mine = _csx.new_handle()
    _csx.check(lib.csx_vec_alloc(n * k, mine), "vec_alloc")
    src = None
    if rank == 0:
        src = _csx.new_handle()
        _csx.check(lib.csx_vec_alloc(n * k * world, src), "vec_alloc")
        sp, sl = C.c_void_p(), C.c_int64()
        _csx.check(lib.csx_vec_ptr(src, sp, sl), "vec_ptr")
        for r in range(world):
            h = _csx.new_handle()
            _csx.check(lib.csx_gen_rhs(n, k, r * k, h), "gen_rhs")
            slot = _csx.new_handle()
            _csx.check(lib.csx_vec_wrap(C.c_void_p(sp.value + 8 * n * k * r), n * k, slot), "vec_wrap")
            _csx.check(lib.csx_vec_copy(h, slot), "vec_copy")
            _csx.free(slot)
            _csx.free(h)
    barrier()
    t0 = time.perf_counter()
    comm.scatter_vec_blocks(src, mine, n * k, 0)
    barrier()
    t_scatter = max_over_ranks(time.perf_counter() - t0)
    hB2 = _csx.new_handle()
    _csx.check(lib.csx_gen_rhs(n, k, rank * k, hB2), "gen_rhs")
    ok = 1.0 if block_np(hB2).tobytes() == block_np(mine).tobytes() else 0.0
    _csx.free(hB2)
    _csx.free(mine)
    res["rhs_scatter_from_root"] = {"bytes_out_of_root": 8 * n * k * (world - 1), "s": round(t_scatter, 5),
                                    "GBps_out_of_root": round(8 * n * k * (world - 1) / t_scatter / 1e9, 2)
                                    if world > 1 else None,
                                    "blocks_equal_locally_generated": bool(comm.sum(ok) == world)}

    # ---- 3. solutions gathered to the root (into the buffer the blocks left from) ----
    digests = comm.all_gather_object(hashlib.sha256(X_own.tobytes()).hexdigest())
    barrier()
    t0 = time.perf_counter()
    comm.gather_vec_blocks(hR, src, n * k, 0)
    barrier()
    t_gather = max_over_ranks(time.perf_counter() - t0)
    okg = True
    if rank == 0:
        for r in range(world):
            slot = _csx.new_handle()
            _csx.check(lib.csx_vec_wrap(C.c_void_p(sp.value + 8 * n * k * r), n * k, slot), "vec_wrap")
            okg = okg and hashlib.sha256(block_np(slot).tobytes()).hexdigest() == digests[r]
            _csx.free(slot)
        _csx.free(src)
    res["solutions_gather_to_root"] = {"bytes_into_root": 8 * n * k * (world - 1), "s": round(t_gather, 5),
                                       "GBps_into_root": round(8 * n * k * (world - 1) / t_gather / 1e9, 2)
                                       if world > 1 else None,
                                       "nrhs_per_gpu": k, "checksums_match": bool(comm.broadcast_object(okg))}
    _csx.free(hR)
    return res


def warm_csc(lib, _csx):
    """A 1 x 1 matrix: the first broadcast of a communicator pays its set-up."""
    import numpy as np
    h = _csx.new_handle()
    p, i, x = np.asarray([0, 1], np.int32), np.asarray([0], np.int32), np.asarray([1.0])
    _csx.check(lib.csx_csc_upload(1, 1, _csx.pi(p), _csx.pi(i), _csx.pd(x), h), "csc_upload")
    return h


if __name__ == "__main__":
    main()
