"""The N > 1 path on CPU: two gloo ranks shard a batched cs_cholsol the way bench.py shards
it across GPUs (one block of right-hand sides per rank, no data-path collective), with the
oracle standing in for the device solver.  Checks the partition, the control collectives and
that gathering the blocks reproduces the serial answer."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path[:0] = [os.path.join(r"{root}", "csparse.py_amd"), os.path.join(r"{root}", "oracle"),
                    os.path.join(r"{root}", "tests")]
    import shard, synth
    import c_oracle as CO
    comm = shard.Comm(backend="gloo")
    assert comm.world == 2
    nblocks, bs, per_rank = 6, 8, 3
    n = nblocks * bs
    Ap, Ai, Ax = synth.gspd(nblocks, bs, 5)              # every rank factors the same matrix
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    col0, k = shard.weak_block(comm.rank, per_rank)
    B = synth.rhs(n, k, col0)                             # this rank's block of the global RHS
    X = np.stack([CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, B[:, r])) for r in range(k)], axis=1)
    comm.barrier()
    t = comm.max(1.0 + comm.rank)                         # slowest rank's time
    total = comm.sum(k)
    choice = comm.broadcast_object("tiled" if comm.rank == 0 else "wave")
    full = comm.gather_blocks(X)
    # the exchange step of a column-sharded SpMV: partial y's summed, each rank keeps its rows
    import torch
    part = torch.arange(8, dtype=torch.float64) * (comm.rank + 1)     # rank 0: 0..7, rank 1: 0,2,..14
    mine = comm.reduce_scatter_sum(part)
    out = dict(rank=comm.rank, tmax=t, total=total, choice=choice, col0=col0, rs=mine.tolist(),
               strong=[shard.strong_block(r, 3, 10) for r in range(3)])
    if comm.rank == 0:
        out["full"] = np.asarray(full).tolist()
    print("RESULT " + json.dumps(out))
    comm.close()
""")


def test_two_rank_sharded_cholsol(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    import json
    res = {}
    for so, _ in outs:
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][0]
        d = json.loads(line[7:])
        res[d["rank"]] = d
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0
    assert res[0]["total"] == res[1]["total"] == 6.0
    assert res[0]["choice"] == res[1]["choice"] == "tiled"
    assert (res[0]["col0"], res[1]["col0"]) == (0, 3)
    assert res[0]["strong"] == [[0, 4], [4, 3], [7, 3]]
    assert res[0]["rs"] == [0.0, 3.0, 6.0, 9.0] and res[1]["rs"] == [12.0, 15.0, 18.0, 21.0]
    # the gathered block equals the serial solve of all six right-hand sides
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle as CO
    import synth
    n = 48
    Ap, Ai, Ax = synth.gspd(6, 8, 5)
    parent, cp = CO.schol(n, Ap, Ai)
    Lp, Li, Lx = CO.chol(n, Ap, Ai, Ax, parent, cp)
    B = synth.rhs(n, 6, 0)
    ref = np.stack([CO.ltsolve(n, Lp, Li, Lx, CO.lsolve(n, Lp, Li, Lx, B[:, r])) for r in range(6)], axis=1)
    assert np.asarray(res[0]["full"]).tobytes() == ref.tobytes()


def test_single_rank_comm_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
    import shard
    old = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        c = shard.Comm()
        assert (c.rank, c.world) == (0, 1) and c.dist is None
        assert c.max(3.5) == 3.5 and c.sum(2) == 2.0 and c.broadcast_object("x") == "x"
        c.barrier()
        blk = np.ones((2, 2))
        assert c.gather_blocks(blk) is blk
        c.close()
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v
    assert shard.weak_block(3, 128) == (384, 128)
    assert [shard.strong_block(r, 8, 1024) for r in (0, 7)] == [(0, 128), (896, 128)]


def test_rendezvous_carries_the_unique_id_and_skips_strangers():
    """shard.Rendezvous: the 128-byte RCCL id from rank 0 to the other ranks over TCP.  A stranger already listening
    on the first candidate port (another job, an old launch) is skipped: its answer lacks this launch's token."""
    import socket
    import threading
    sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
    import shard
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        base = probe.getsockname()[1]
    stranger = socket.socket()
    stranger.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        stranger.bind(("127.0.0.1", base + 1))           # the first candidate port is taken by someone else
    except OSError:
        pytest.skip("port in use")
    stranger.listen(8)
    stop = threading.Event()

    def strange():
        stranger.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = stranger.accept()
                c.sendall(b"\x05\x00\x00\x00HELLO")
                c.close()
            except OSError:
                pass

    th = threading.Thread(target=strange, daemon=True)
    th.start()
    blob = bytes(range(128))
    got = {}

    def run(rank):
        rv = shard.Rendezvous(rank, 3, addr="127.0.0.1", port=base, token="launch-42", timeout=30)
        got[rank] = rv.share(blob if rank == 0 else None)

    ts = [threading.Thread(target=run, args=(r,)) for r in (1, 2, 0)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(40)
    stop.set()
    stranger.close()
    assert got == {0: blob, 1: blob, 2: blob}
    # a client of ANOTHER launch (different token) is refused by rank 0 and does not consume a slot
    assert shard.Rendezvous(0, 1).share(b"x") == b"x"                     # world of one: nothing to do
    assert shard.row_chunk(1, 2, 3001) == (1501, 1500) and shard.row_chunk(7, 8, 10) == (10, 0)


def test_rendezvous_token_comes_from_the_launcher_and_a_mismatch_names_both_tokens(monkeypatch):
    """Ranks started by DIFFERENT parents (torchrun across nodes, per-rank wrapper shells) share the launcher's run id or
    MASTER_ADDR:MASTER_PORT, not a parent pid; a client that only finds a rank 0 of another launch says whose port it met."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "csparse.py_amd"))
    import shard
    for k in ("CSX_RDV_TOKEN", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert shard.Rendezvous(0, 2).token == b"ppid:%d/2" % os.getppid()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29411")
    assert shard.Rendezvous(1, 4).token == b"master:127.0.0.1:29411/4"
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job-7")
    assert shard.Rendezvous(1, 4).token == b"run:job-7@127.0.0.1:29411/4"
    monkeypatch.setenv("CSX_RDV_TOKEN", "mine")
    assert shard.Rendezvous(1, 4).token == b"mine/4"
    import socket
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        base = probe.getsockname()[1]
    res = {}

    def server():
        try:
            shard.Rendezvous(0, 2, addr="127.0.0.1", port=base, token="launch-A", timeout=6).share(b"id")
        except Exception as e:               # noqa: BLE001 -- nobody of its own launch comes: accept() times out
            res["server"] = type(e).__name__

    def client():
        try:
            shard.Rendezvous(1, 2, addr="127.0.0.1", port=base, token="launch-B", timeout=3).share(None)
        except Exception as e:               # noqa: BLE001
            res["client"] = str(e)

    ts = [threading.Thread(target=server), threading.Thread(target=client)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert "launch-B/2" in res["client"] and "launch-A/2" in res["client"] and "CSX_RDV_TOKEN" in res["client"]
    assert res["server"] in ("timeout", "TimeoutError")
    # the server socket was closed on the way out: the port can be bound again at once
    with socket.socket() as again:
        again.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        again.bind(("127.0.0.1", base + 1))


def _run_bench(extra, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_gpus_n_really_starts_n_ranks(world):
    """`python bench.py --gpus N` must start N ranks itself (no outer launcher) and say n_gpus = N;
    --rehearse keeps it off the GPU: launcher, world check and every exchange leg on CPU tensors over gloo.  N = 8 is the
    node the driver's scaling run uses: uneven blocks of right-hand sides and columns over eight ranks."""
    import json
    r = _run_bench(["--gpus", str(world), "--rehearse"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["rehearsal"] is True and d["value"] is None
    assert d["exchange"] == {"broadcast_ok": True, "scatter_ok": True, "gather_ok": True}
    assert d["slowest_rank"] == float(world - 1)


def test_bench_refuses_a_world_of_another_size():
    """A launcher that set up 1 rank for `--gpus 2` (or 3 for 2) is an error, not a silent smaller run."""
    r = _run_bench(["--gpus", "2", "--rehearse"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing" in r.stderr
    r = _run_bench(["--gpus", "1", "--rehearse"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing" in r.stderr


def test_bench_deadline_prints_the_headline_when_a_later_leg_hangs():
    """bench.py's legs after the headline are collectives at N > 1: if one hangs, rank 0 must still print the JSON line
    as it stood after the last finished leg, marked, and exit 0 (bench.Deadline)."""
    import json
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import bench\n"
        "out = {'metric': 'm', 'value': 1.0}\n"
        "d = bench.Deadline(0.3, 0); d.arm(out)\n"
        "out['leg1'] = 'done'; d.checkpoint(out)\n"
        "time.sleep(30)\n"          # a leg that never returns
        "print('not reached')\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=20)
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "not reached" not in r.stdout
    got = json.loads(lines[0])
    assert got["value"] == 1.0 and got["leg1"] == "done" and got["extras_cut_short_after_s"] == 0.3
    # a rank other than 0 exits 0 without printing
    code2 = code.replace("bench.Deadline(0.3, 0)", "bench.Deadline(0.3, 1)")
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=30)
    assert r2.returncode == 0 and r2.stdout.strip() == ""
