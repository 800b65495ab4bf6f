"""bench.py --gpus 2 on the one-GPU box: two real ranks (started by bench.py itself), both on device 0
(CSX_SINGLE_DEVICE), exchanging over gloo (the host stand-in of shard.Comm) with device buffers staged through the host.  Everything but
the transport is the N > 1 code path of the 8-GPU run: the world check, the rank-sharded right-hand
sides, the three exchange legs of the batched cs_cholsol and the column-sharded SpMV with its
reduce-scatter.  RCCL itself needs one GPU per rank; its call sequence is covered at world size 1 by
test_exchange_legs_on_a_real_rccl_group_of_one."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _clean_env(**kw):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(kw)
    return e


def test_bench_two_ranks_on_one_device():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "200000", "--steps", "3",
                        "--warmup", "1", "--nrhs", "64"],
                       env=_clean_env(CSX_SINGLE_DEVICE="1", CSX_COMM_BACKEND="gloo"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    pre = d["exchange_preflight"]                      # every exchange leg at a small size first, pass / fail per leg
    assert pre["world"] == 2 and pre["all_ok"] is True and "running" not in pre
    assert set(pre["legs"]) == {"factor_broadcast", "rhs_scatter", "solution_gather", "sharded_solve_api",
                                "sharded_gaxpy_reduce_scatter", "sharded_gaxpy_row_pieces_p2p"}
    ex = d["cholsol"]["exchange"]
    assert "error" not in ex, ex
    assert ex["world"] == 2
    assert ex["factor_once_broadcast"]["receivers_reproduce_own_solution_bit_for_bit"] is True
    assert ex["rhs_scatter_from_root"]["blocks_equal_locally_generated"] is True
    assert ex["solutions_gather_to_root"]["checksums_match"] is True
    sh = d["gaxpy_one_matrix_column_sharded"]
    assert "error" not in sh and sh["rows_equal_unsharded"] is True       # row for row against the unsharded cs_gaxpy
    assert set(sh["forms"]) == {"spmv_then_reduce_scatter", "row_pieces_overlapped_p2p"}


def test_exchange_legs_on_a_real_rccl_group_of_one():
    """CSX_FORCE_DIST: a real torch.distributed process group (backend nccl = RCCL) with one member, so the
    exchange legs run through RCCL's API on device tensors that are views of libcsx buffers."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--n", "200000", "--steps", "2",
                        "--warmup", "1", "--nrhs", "64", "--force-sharded", "--skip-cpu", "--skip-configs"],
                       env=_clean_env(CSX_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                                      MASTER_ADDR="127.0.0.1", MASTER_PORT="29671"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["exchange_preflight"]["all_ok"] is True and d["exchange_preflight"]["backend"] == "rccl (libcsx)"
    ex = d["cholsol"]["exchange"]
    assert "error" not in ex, ex
    assert ex["backend"] == "rccl (libcsx)"                  # csx_comm_*: RCCL bound inside the library, no torch
    assert ex["factor_once_broadcast"]["receivers_reproduce_own_solution_bit_for_bit"] is True
    assert ex["rhs_scatter_from_root"]["blocks_equal_locally_generated"] is True
    assert ex["solutions_gather_to_root"]["checksums_match"] is True
    sh = d["gaxpy_one_matrix_column_sharded"]
    assert "error" not in sh and sh["rows_equal_unsharded"] is True


def test_dry_exchange_alone_on_three_ranks():
    """bench.py --gpus 3 --dry-exchange: only the preflight (no headline, value null), three ranks on the one device."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-exchange"],
                       env=_clean_env(CSX_SINGLE_DEVICE="1", CSX_COMM_BACKEND="gloo"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] is None and d["dry_exchange"] is True and d["n_gpus"] == 3
    assert d["exchange_preflight"]["all_ok"] is True and len(d["exchange_preflight"]["legs"]) == 6
