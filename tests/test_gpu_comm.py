"""The multi-GPU entry points of libcsx (include/csx.h, csx_comm_*) on the one-GPU box.

RCCL wants one GPU per rank, so two things are run: (1) the whole csx_comm_* API through a REAL RCCL communicator of
one rank (csx_comm_unique_id -> csx_comm_init: every collective goes through librccl); (2) the sharded operations at
world size 2 -- cholsol_factor(A).solve(B, comm=...) and the column-sharded cs_gaxpy in both exchange forms -- with two
processes on the one device and the host stand-in (gloo) as the wire.  Results are compared with the plain-C oracle:
solutions bit for bit (a sharded solve is the unsharded one column by column), the SpMV row for row."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

HEAD = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path[:0] = [os.path.join(r"{root}", "csparse.py_amd"), os.path.join(r"{root}", "oracle"),
                    os.path.join(r"{root}", "tests")]
    import shard, synth, _csx
    import csparse as cs
    import c_oracle as CO
""")

RCCL_ONE = HEAD + textwrap.dedent("""
    comm = shard.Comm()                      # CSX_FORCE_DIST=1: a real RCCL communicator of one rank inside libcsx
    lib = _csx.lib()
    C = _csx.C
    r, w, u = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    _csx.check(lib.csx_comm_info(r, w, u))
    out = dict(info=[r.value, w.value, u.value], backend=comm.backend)
    # control plane
    out["max"] = comm.max(3.5); out["sum"] = comm.sum(2.0); out["obj"] = comm.broadcast_object({{"k": [1, 2]}})
    out["gather"] = comm.all_gather_object("x")
    comm.barrier(_csx.sync)
    # factor shipped: the root keeps its handle
    Ap, Ai, Ax = synth.grand(3001, 7, 11)
    hA = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(3001, 3001, _csx.pi(Ap), _csx.pi(Ai), _csx.pd(Ax), hA))
    h2 = comm.bcast_csc(hA, 0)
    out["bcast_same_handle"] = h2.value == hA.value
    # blocks out and back
    n, k = 1000, 3
    src = cs.dvec(np.arange(n * k, dtype=np.float64)); dst = cs.dvec(n * k); back = cs.dvec(n * k)
    comm.scatter_vec_blocks(src.handle, dst.handle, n * k, 0)
    comm.gather_vec_blocks(dst.handle, back.handle, n * k, 0)
    out["blocks_roundtrip"] = bool((back.numpy() == np.arange(n * k)).all())
    full = cs.dvec(np.arange(64, dtype=np.float64)); piece = cs.dvec(64)
    comm.reduce_scatter_vec(full.handle, piece.handle, 64)
    out["reduce_scatter"] = bool((piece.numpy() == np.arange(64)).all())
    v = cs.dvec(np.ones(10)); _csx.check(lib.csx_comm_allreduce_vec(v.handle)); _csx.check(lib.csx_comm_bcast_vec(v.handle, 0))
    out["allreduce_vec"] = bool((v.numpy() == 1).all())
    # column-sharded cs_gaxpy, both exchange forms, against the oracle
    x = synth.vec(3001, 3, 0.5, 1.5)
    yref = CO.gaxpy(3001, 3001, Ap, Ai, Ax, x, np.zeros(3001))
    sg = shard.ShardedGaxpy(comm, hA, 3001)
    errs = []
    for how in (0, 1):
        y = cs.dvec(sg.chunk); dx = cs.dvec(x)
        sg.run(dx.handle, y.handle, how)
        sg.run(dx.handle, y.handle, how)                  # y += : twice the product
        f, c = sg.rows()
        errs.append(float(np.max(np.abs(y.numpy()[:c] - 2 * yref[f:f + c]) / np.abs(yref[f:f + c]))))
    out["gaxpy_err"] = errs
    sg.free()
    print("RESULT " + json.dumps(out))
    comm.close()
""")

TWO_RANKS = HEAD + textwrap.dedent("""
    comm = shard.Comm(backend="gloo")        # two ranks, one device: the host stand-in carries the exchange
    lib = _csx.init(0)
    rank, world = comm.rank, comm.world
    out = dict(rank=rank)
    # ---- cholsol_factor(A).solve(B, comm=...): K = 5 right-hand sides over 2 ranks (3 + 2, the last block padded) ----
    nb, bs, K = 6, 8, 5
    n = nb * bs
    Ap, Ai, Ax = synth.gspd(nb, bs, 5)
    A = cs.cs_spalloc(n, n, len(Ai), True, False)
    A.p, A.i, A.x = Ap.tolist(), Ai.tolist(), Ax.tolist()
    F = cs.cholsol_factor(cs.cs_pin(A), exact=True)                  # every rank factors the same matrix
    B = synth.rhs(n, K, 0)
    dB = cs.dvec(B) if rank == 0 else None
    assert F.solve(dB, comm=comm, nrhs=K)
    if rank == 0:
        parent, cp = CO.schol(n, Ap, Ai)
        Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
        ref = np.stack([CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, B[:, r])) for r in range(K)], axis=1)
        out["solve_bit_identical"] = dB.numpy().tobytes() == ref.tobytes()
    # ---- lusol_factor / qrsol_factor: the same sharding for cs_lusol (csparse.py:1474-1477) and cs_qrsol (:1893-1897) ----
    from conftest import golden, unpack
    import csparse_oracle as O
    gw = golden("west0067")
    W = cs.cs_pin(unpack(cs, gw, "C"))
    nw = W.n
    FL = cs.lusol_factor(W, 0, 1.0)
    BW = synth.rhs(nw, K, 3)
    dW = cs.dvec(BW) if rank == 0 else None
    assert FL.solve(dW, comm=comm, nrhs=K)
    if rank == 0:
        ok = True
        for r in range(K):
            col = BW[:, r].tolist()
            assert O.cs_lusol(0, unpack(O, gw, "C"), col, 1.0)
            ok = ok and np.asarray(col).tobytes() == np.ascontiguousarray(dW.numpy().reshape(nw, K)[:, r]).tobytes()
        one = BW[:, 1].tolist()                          # ... and the unsharded solver on a list, and cs_lusol on a block
        assert FL.solve(one)
        ok = ok and np.asarray(one).tobytes() == np.ascontiguousarray(dW.numpy().reshape(nw, K)[:, 1]).tobytes()
        dW2 = cs.dvec(BW)
        assert cs.cs_lusol(0, W, dW2, 1.0)
        out["lusol_bit_identical"] = ok and dW2.numpy().tobytes() == dW.numpy().tobytes()
    ga = golden("ash219")
    Q = cs.cs_pin(unpack(cs, ga, "C"))
    mq, nq = Q.m, Q.n
    FQ = cs.qrsol_factor(Q, 0)
    BQ = synth.rhs(mq, K, 1)
    XQ = FQ.solve(cs.dvec(BQ) if rank == 0 else None, comm=comm, nrhs=K)
    if rank == 0:
        ok = True
        for r in range(K):
            col = BQ[:, r].tolist()
            assert cs.cs_qrsol(0, Q, col)
            ok = ok and np.asarray(col[:nq]).tobytes() == np.ascontiguousarray(XQ.numpy().reshape(nq, K)[:, r]).tobytes()
        out["qrsol_bit_identical"] = ok
    else:
        assert XQ is None
    # ---- the factor shipped from rank 0 instead of factored twice ----
    hL = comm.bcast_csc(F.L._dev.handle if rank == 0 else None, 0)
    p, i, x = np.empty(n + 1, np.int32), np.empty(F.L.p[n], np.int32), np.empty(F.L.p[n])
    _csx.check(lib.csx_csc_download(hL, _csx.pi(p), _csx.pi(i), _csx.pd(x)))
    out["factor_arrived"] = p.tolist() == F.L.p and i.tolist() == F.L.i[:F.L.p[n]] and x.tolist() == F.L.x[:F.L.p[n]]
    # ---- ONE cs_gaxpy sharded by columns: n = 3001 (uneven column blocks and row chunks), both forms ----
    m = 3001
    Gp, Gi, Gx = synth.grand(m, 7, 11)
    hG = _csx.new_handle()
    _csx.check(lib.csx_csc_upload(m, m, _csx.pi(Gp), _csx.pi(Gi), _csx.pd(Gx), hG))
    first, count = shard.strong_block(rank, world, m)
    hBk = _csx.new_handle()
    _csx.check(lib.csx_csc_col_block(hG, first, count, hBk))
    xfull = synth.vec(m, 3, 0.5, 1.5)
    yref = CO.gaxpy(m, m, Gp, Gi, Gx, xfull, np.zeros(m))
    sg = shard.ShardedGaxpy(comm, hBk, m)
    f, c = sg.rows()
    errs = []
    for how in (0, 1):
        y = cs.dvec(sg.chunk); dx = cs.dvec(xfull[first:first + count])
        sg.run(dx.handle, y.handle, how)
        errs.append(float(np.max(np.abs(y.numpy()[:c] - yref[f:f + c]) / np.abs(yref[f:f + c]))))
    out["gaxpy_rows"] = [f, c]
    out["gaxpy_err"] = errs
    sg.free()
    print("RESULT " + json.dumps(out))
    comm.close()
""")


def _env(**kw):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CSX_FORCE_DIST", "CSX_COMM_BACKEND"):
        e.pop(k, None)
    e.update(kw)
    return e


def _result(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("RESULT ")][0][7:])


def test_comm_api_through_a_real_rccl_communicator_of_one(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(RCCL_ONE.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], env=_env(CSX_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                                                               MASTER_ADDR="127.0.0.1", MASTER_PORT="29711"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _result(r.stdout)
    assert d["info"] == [0, 1, 1] and d["backend"] == "rccl (libcsx)"          # uses_rccl = 1: librccl is really underneath
    assert d["max"] == 3.5 and d["sum"] == 2.0 and d["obj"] == {"k": [1, 2]} and d["gather"] == ["x"]
    assert d["bcast_same_handle"] and d["blocks_roundtrip"] and d["reduce_scatter"] and d["allreduce_vec"]
    assert max(d["gaxpy_err"]) < 1e-12


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_solve_and_sharded_gaxpy_at_world_size_two_and_three(tmp_path, world):
    """world 3: 5 right-hand sides as 2 + 2 + 1, column blocks of 1001 / 1000 / 1000, row chunks of 1001 / 1001 / 999, and
    two exchange steps per SpMV in rotated order (world 2 has one)."""
    sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
    import shard
    script = tmp_path / "w.py"
    script.write_text(TWO_RANKS.format(root=ROOT))
    procs = []
    for rk in range(world):
        procs.append(subprocess.Popen([sys.executable, str(script)],
                                      env=_env(RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                                               MASTER_PORT=str(29713 + world), CSX_SINGLE_DEVICE="1",
                                               CSX_COMM_BACKEND="gloo"),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = {d["rank"]: d for d in (_result(so) for so, _ in outs)}
    assert res[0]["solve_bit_identical"] is True
    assert res[0]["lusol_bit_identical"] is True and res[0]["qrsol_bit_identical"] is True
    assert all(res[r]["factor_arrived"] for r in range(world))
    assert [res[r]["gaxpy_rows"] for r in range(world)] == [list(shard.row_chunk(r, world, 3001)) for r in range(world)]
    assert max(e for r in range(world) for e in res[r]["gaxpy_err"]) < 1e-12
