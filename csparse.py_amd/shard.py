"""One-process-per-GPU sharding of the hot path.

The path shards as independent units -- right-hand-side blocks of a batched solve (csparse.py:640-643), whole
independent matrices -- so there is NO collective on that data path.  Exchange steps exist around it (factor once and
ship the factor, right-hand-side blocks leaving the root, solutions returning to it) and inside ONE operation: a single
cs_gaxpy (csparse.py:1210-1212) sharded by columns, whose partial y vectors must be summed.

Transport.  On the GPU node every exchange is RCCL over xGMI INSIDE libcsx (csx_comm_*, csparse.py_amd/csrc/csx_comm.hip):
device buffers behind csx handles, enqueued on the library's own stream in order with its kernels.  No torch, no second
HIP runtime, no stream hand-over.  The only thing this module carries itself is the communicator's 128-byte unique id
from rank 0 to the other ranks: a TCP hand-shake on MASTER_ADDR (Rendezvous), ports MASTER_PORT + 1 ... + 16 (MASTER_PORT
itself belongs to whoever launched the ranks).  Control data after that (barriers, the max of a timing, a broadcast
choice) also travels through libcsx (csx_comm_allreduce_host / csx_comm_bcast_host).

backend "gloo" is the HOST STAND-IN: torch.distributed over gloo on CPU tensors, device buffers staged through the host.
It carries the CPU tests of the N > 1 control flow (tests/test_shard_gloo.py, bench.py --rehearse) and the two-ranks-on-
one-GPU rehearsal (RCCL wants one GPU per rank); same call sequence, same results, different wire.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np


def env_rank():
    """(rank, world, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def weak_block(rank, per_rank):
    """Weak scaling: rank r owns global columns [r*per_rank, (r+1)*per_rank)."""
    return rank * per_rank, per_rank


def strong_block(rank, world, total):
    """Strong scaling: `total` columns split as evenly as possible, contiguous, in rank order."""
    base, rem = divmod(total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def row_chunk(rank, world, m):
    """Rows of y a rank owns after a column-sharded SpMV: [r*chunk, min((r+1)*chunk, m)), chunk = ceil(m / world)
    (csx_gaxpy_sharded_rows: equal pieces, so that one reduce-scatter serves)."""
    chunk = (m + world - 1) // world
    first = min(rank * chunk, m)
    return first, min(chunk, m - first)


# ------------------------------------------------------------------ rendezvous ----

_MAGIC = b"CSX1"


class Rendezvous(object):
    """Carries one small blob (the RCCL unique id) from rank 0 to the other ranks of THIS launch.

    Rank 0 listens on the first free port of MASTER_PORT + 1 ... + 16 and serves world - 1 clients; a client tries
    the candidates in turn until a server answers the hand-shake with this launch's token, so a stranger on one of the
    ports is skipped, not believed.  The token is something every rank of ONE launch shares, whatever started them:
    CSX_RDV_TOKEN, else the launcher's run id (TORCHELASTIC_RUN_ID; torchrun, also across nodes), else MASTER_ADDR :
    MASTER_PORT when the launcher set them (srun / mpirun wrappers), else -- ranks forked by one parent on one host, as
    bench.py and the tests do -- the parent's pid; always with the world size.  A server that sees ANOTHER token says
    whose port it is, and a client that finds only such servers names both tokens in its error."""

    def __init__(self, rank, world, addr=None, port=None, token=None, timeout=300.0):
        self.rank, self.world = rank, world
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        tok = token if token is not None else os.environ.get("CSX_RDV_TOKEN")
        if tok is None and os.environ.get("TORCHELASTIC_RUN_ID"):
            # (torchrun without --rdzv-id gives every job the id "none": the master address and port tell two such jobs apart)
            tok = "run:%s@%s:%s" % (os.environ["TORCHELASTIC_RUN_ID"], os.environ.get("MASTER_ADDR", ""), os.environ.get("MASTER_PORT", ""))
        if tok is None and "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ and addr is None and port is None:
            tok = "master:%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])
        if tok is None:
            tok = "ppid:%d" % os.getppid()
        self.token = ("%s/%d" % (tok, world)).encode()
        self.timeout = timeout

    def _candidates(self):
        return [self.port + 1 + k for k in range(16)]

    @staticmethod
    def _send(sk, blob):
        sk.sendall(struct.pack("<I", len(blob)) + blob)

    @staticmethod
    def _recv(sk):
        def take(n):
            buf = b""
            while len(buf) < n:
                part = sk.recv(n - len(buf))
                if not part:
                    raise ConnectionError("peer closed")
                buf += part
            return buf
        (n,) = struct.unpack("<I", take(4))
        return take(n)

    def share(self, blob):
        """Rank 0: blob -> everyone; returns the blob on every rank."""
        if self.world == 1:
            return blob
        if self.rank == 0:
            srv = None
            for port in self._candidates():
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((self.addr, port))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("shard.Rendezvous: no free port in %s" % self._candidates())
            try:
                srv.listen(self.world)
                srv.settimeout(self.timeout)
                served = 0
                while served < self.world - 1:
                    conn, _ = srv.accept()             # socket.timeout after self.timeout: srv is closed below
                    try:
                        conn.settimeout(10.0)
                        hello = self._recv(conn)
                        if hello == _MAGIC + self.token:
                            self._send(conn, _MAGIC + blob)
                            served += 1
                        else:                          # a client of another launch (or a stranger): tell it whose port this is
                            self._send(conn, b"NOPE" + self.token)
                    except (OSError, ConnectionError, struct.error):
                        pass
                    finally:
                        conn.close()
            finally:
                srv.close()
            return blob
        deadline = time.time() + self.timeout
        refused = None
        while time.time() < deadline:
            for port in self._candidates():
                try:
                    left = deadline - time.time()
                    if left <= 0:
                        break
                    # (a stranger that accepts and says nothing must not hold this rank past its deadline: the waits are cut to what is left)
                    with socket.create_connection((self.addr, port), timeout=min(2.0, max(0.2, left))) as sk:
                        sk.settimeout(min(10.0, max(0.2, left)))
                        self._send(sk, _MAGIC + self.token)
                        ans = self._recv(sk)
                        if ans[:4] == _MAGIC:
                            return ans[4:]
                        if ans[:4] == b"NOPE":
                            refused = (port, ans[4:])
                except (OSError, ConnectionError, struct.error):
                    continue
            time.sleep(0.05)
        if refused is not None:
            # the only csx servers found belong to another launch: the ranks of this job do not share a token (started by
            # different parents without a common run id?)
            raise TimeoutError("shard.Rendezvous: rank %d's token %r is not the token %r of the rank 0 listening on %s:%d, and no "
                               "other rank 0 answered; set CSX_RDV_TOKEN to one value for all ranks of the job"
                               % (self.rank, self.token, refused[1], self.addr, refused[0]))
        raise TimeoutError("shard.Rendezvous: rank 0 did not answer on %s:%s" % (self.addr, self._candidates()))


# ------------------------------------------------------------------------ comm ----

class Comm(object):
    """The ranks' exchange.  world == 1 (and CSX_FORCE_DIST unset): everything is a no-op / local copy and nothing is
    imported.  backend "csx" (default at world > 1): RCCL inside libcsx.  backend "gloo": the host stand-in."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local = env_rank()
        if os.environ.get("CSX_SINGLE_DEVICE"):   # rehearsal of the N > 1 control flow on a one-GPU box
            self.local = 0
        self.dist = None          # torch.distributed when the transport is gloo
        self.csx = None           # libcsx when the transport is RCCL
        self.backend = "none"
        self._local_ready = False
        forced = bool(os.environ.get("CSX_FORCE_DIST"))
        if self.world > 1 or forced:
            backend = backend or os.environ.get("CSX_COMM_BACKEND") or "csx"
            if backend == "nccl":                 # the old name of the RCCL transport
                backend = "csx"
            if backend == "gloo":
                import torch
                import torch.distributed as dist
                dist.init_process_group("gloo")
                self.dist, self.torch = dist, torch
            elif backend == "csx":
                import _csx
                lib = _csx.init(self.local)
                idbuf = (_csx.C.c_uint8 * 128)()
                if self.rank == 0:
                    _csx.check(lib.csx_comm_unique_id(idbuf), "csx_comm_unique_id")
                blob = Rendezvous(self.rank, self.world).share(bytes(idbuf))
                idbuf = (_csx.C.c_uint8 * 128).from_buffer_copy(blob)
                _csx.check(lib.csx_comm_init(self.rank, self.world, idbuf), "csx_comm_init")
                self.csx, self._csx = lib, _csx
                self._local_ready = True
            else:
                raise ValueError("shard.Comm: unknown backend %r" % backend)
            self.backend = "rccl (libcsx)" if backend == "csx" else backend

    def info(self):
        """What the transport itself reports: {"backend", "world", "rank", "uses_rccl", "ranks_counted"} -- world / rank / uses_rccl
        from csx_comm_info (the communicator libcsx made), ranks_counted = a sum of ones over that transport."""
        d = {"backend": self.backend, "world": self.world, "rank": self.rank, "uses_rccl": False}
        if self.csx is not None:
            C = self._csx.C
            r, w, u = C.c_int(-1), C.c_int(-1), C.c_int(0)
            self._csx.check(self.csx.csx_comm_info(r, w, u), "csx_comm_info")
            d.update(world=w.value, rank=r.value, uses_rccl=bool(u.value))
        d["ranks_counted"] = int(round(self.sum(1.0)))
        return d

    # ---- control plane -------------------------------------------------------------------------------------------

    def barrier(self, sync=None):
        """Device sync (callable) then a rank barrier."""
        if sync is not None:
            sync()
        if self.csx is not None:
            self._csx.check(self.csx.csx_comm_barrier(), "csx_comm_barrier")
        elif self.dist is not None:
            self.dist.barrier()

    def _reduce(self, value, op):
        if self.csx is not None:
            v = self._csx.C.c_double(float(value))
            self._csx.check(self.csx.csx_comm_allreduce_host(self._csx.C.byref(v), 1, op), "csx_comm_allreduce_host")
            return float(v.value)
        if self.dist is not None:
            t = self.torch.tensor([float(value)], dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op else self.dist.ReduceOp.SUM)
            return float(t.item())
        return float(value)

    def max(self, value):
        return self._reduce(value, 1)

    def sum(self, value):
        return self._reduce(value, 0)

    def broadcast_object(self, obj, src=0):
        if self.csx is not None:
            C = self._csx.C
            blob = pickle.dumps(obj) if self.rank == src else b""
            n = C.c_int64(len(blob))
            self._csx.check(self.csx.csx_comm_bcast_host(C.byref(n), 8, src), "csx_comm_bcast_host")
            buf = C.create_string_buffer(blob, n.value) if self.rank == src else C.create_string_buffer(n.value)
            self._csx.check(self.csx.csx_comm_bcast_host(buf, n.value, src), "csx_comm_bcast_host")
            return obj if self.rank == src else pickle.loads(buf.raw)
        if self.dist is not None:
            box = [obj]
            self.dist.broadcast_object_list(box, src=src)
            return box[0]
        return obj

    def all_gather_object(self, obj):
        if self.csx is not None:
            return [self.broadcast_object(obj, src=r) for r in range(self.world)]
        if self.dist is not None:
            out = [None] * self.world
            self.dist.all_gather_object(out, obj)
            return out
        return [obj]

    # ---- data path on libcsx handles (device buffers) ------------------------------------------------------------------

    def _lib(self):
        """libcsx with a communicator: the RCCL one, or (world of one / gloo stand-in) the local one."""
        import _csx
        lib = _csx.init(self.local)
        if not self._local_ready:
            _csx.check(lib.csx_comm_init(0, 1, None), "csx_comm_init")
            self._local_ready = True
        return _csx, lib

    def _vec_np(self, h, count):
        _csx, lib = self._lib()
        out = np.empty(count)
        _csx.check(lib.csx_vec_download(h, _csx.pd(out), count), "csx_vec_download")
        return out

    def bcast_csc(self, h, root=0):
        """The root's CSC matrix on every rank: returns the handle to use (the root's own, a new one elsewhere)."""
        _csx, lib = self._lib()
        if self.dist is None:
            hh = _csx.H(h.value if self.rank == root and h is not None else 0)
            _csx.check(lib.csx_comm_bcast_csc(_csx.C.byref(hh), root), "csx_comm_bcast_csc")
            return hh
        C = _csx.C
        meta = None
        if self.rank == root:
            m, n, nnz, hv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int()
            _csx.check(lib.csx_csc_info(h, m, n, nnz, hv), "csx_csc_info")
            meta = (m.value, n.value, nnz.value, hv.value)
        m, n, nnz, hv = self.broadcast_object(meta, root)
        p, i = np.empty(n + 1, np.int32), np.empty(max(nnz, 1), np.int32)
        x = np.empty(max(nnz, 1)) if hv else None
        if self.rank == root:
            _csx.check(lib.csx_csc_download(h, _csx.pi(p), _csx.pi(i), _csx.pd(x)), "csx_csc_download")
        for a in (p, i) + ((x,) if hv else ()):
            self.dist.broadcast(self.torch.from_numpy(a), src=root)
        if self.rank == root:
            return h
        out = _csx.new_handle()
        _csx.check(lib.csx_csc_upload(m, n, _csx.pi(p), _csx.pi(i), _csx.pd(x), out), "csx_csc_upload")
        return out

    def scatter_vec_blocks(self, src, dst, length, root=0):
        """Rank r's `dst` <- block r (length doubles) of the root's `src` (world blocks, rank order)."""
        _csx, lib = self._lib()
        if self.dist is None:
            _csx.check(lib.csx_comm_scatter_blocks(src if self.rank == root else _csx.H(0), dst, length, root),
                       "csx_comm_scatter_blocks")
            return
        mine = self.torch.empty(length, dtype=self.torch.float64)
        blocks = None
        if self.rank == root:
            full = self._vec_np(src, length * self.world)
            blocks = [self.torch.from_numpy(full[r * length:(r + 1) * length].copy()) for r in range(self.world)]
        self.scatter_blocks(mine, blocks, src=root)
        a = mine.numpy()
        _csx.check(lib.csx_vec_write(dst, _csx.pd(a), length), "csx_vec_write")

    def gather_vec_blocks(self, block, out, length, root=0):
        """The root's `out` (world blocks, rank order) <- every rank's `block` (length doubles)."""
        _csx, lib = self._lib()
        if self.dist is None:
            _csx.check(lib.csx_comm_gather_blocks(block, out if self.rank == root else _csx.H(0), length, root),
                       "csx_comm_gather_blocks")
            return
        got = self.gather_to_root(self.torch.from_numpy(self._vec_np(block, length)), dst=root)
        if self.rank == root:
            full = np.concatenate([g.numpy() for g in got])
            _csx.check(lib.csx_vec_write(out, _csx.pd(full), length * self.world), "csx_vec_write")

    def reduce_scatter_vec(self, full, out, length):
        """out (length doubles) <- this rank's piece of the sum of the ranks' `full` vectors (world * length)."""
        _csx, lib = self._lib()
        if self.dist is None:
            _csx.check(lib.csx_comm_reduce_scatter_vec(full, out), "csx_comm_reduce_scatter_vec")
            return
        mine = self.reduce_scatter_sum(self.torch.from_numpy(self._vec_np(full, length * self.world)))
        a = np.ascontiguousarray(mine.numpy())
        _csx.check(lib.csx_vec_write(out, _csx.pd(a), length), "csx_vec_write")

    # ---- the host stand-in's tensor calls (gloo; CPU tests and --rehearse) -------------------------------------------

    def gather_blocks(self, block, dst=0):
        """Gather equally shaped 2-D float64 blocks (n x k_r, same k_r on every rank) to `dst`,
        concatenated along columns in rank order.  Returns the full block on dst, None elsewhere."""
        if self.dist is None:
            return block
        t = self.torch.as_tensor(block, dtype=self.torch.float64).contiguous()
        out = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(t, out, dst=dst)
        if self.rank != dst:
            return None
        return self.torch.cat(out, dim=1)

    def reduce_scatter_sum(self, full):
        """Sum the ranks' full-length 1-D float64 CPU tensors and leave rank r with rows
        [r*len/world, (r+1)*len/world).  gloo has no reduce-scatter: an all-reduce followed by a slice."""
        if self.dist is None:
            return full
        n = full.numel()
        if n % self.world:
            raise ValueError("reduce_scatter_sum: length %d is not a multiple of world %d" % (n, self.world))
        chunk = n // self.world
        tmp = full.detach().clone()
        self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM)
        return tmp[self.rank * chunk:(self.rank + 1) * chunk].clone()

    def broadcast_tensor(self, t, src=0):
        if self.dist is not None:
            self.dist.broadcast(t, src=src)
        return t

    def scatter_blocks(self, out, blocks, src=0):
        """Rank r receives blocks[r] (held by `src`, same shape as `out` everywhere) into `out`: W - 1 sends."""
        if self.dist is None:
            out.copy_(blocks[0])
            return out
        if self.rank == src:
            reqs = []
            for r in range(self.world):
                if r == src:
                    out.copy_(blocks[r])
                else:
                    reqs.append(self.dist.isend(blocks[r], dst=r))
            for q in reqs:
                q.wait()
        else:
            self.dist.recv(out, src=src)
        return out

    def gather_to_root(self, block, dst=0):
        """Every rank's equally shaped CPU block to `dst`: list of W tensors there (rank order), None elsewhere."""
        if self.dist is None:
            return [block]
        if self.rank == dst:
            outs, reqs = [], []
            for r in range(self.world):
                if r == dst:
                    outs.append(block)
                else:
                    o = self.torch.empty(block.shape, dtype=block.dtype)
                    outs.append(o)
                    reqs.append(self.dist.irecv(o, src=r))
            for q in reqs:
                q.wait()
            return outs
        self.dist.send(block, dst=dst)
        return None

    def close(self):
        if self.csx is not None:
            self.barrier()
            self.csx.csx_comm_finalize()
            self.csx = None
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


class ShardedGaxpy(object):
    """ONE cs_gaxpy (csparse.py:1199-1213) over the ranks of `comm`: this rank holds the column block `block` (a libcsx
    CSC handle, m x count: csx_csc_col_block of the whole matrix) and the matching slice of x.
    run(hx, hy_mine, how): hy_mine += this rank's rows (rows()) of sum_r A_r x_r.  how 0: one SpMV + one reduce-scatter;
    how 1: row pieces in rotated order, each leaving for its owner while the next is computed (csx_gaxpy_sharded)."""

    def __init__(self, comm, block, m):
        self.comm, self.block, self.m = comm, block, m
        self._csx, self.lib = comm._lib()
        self.plan = None
        self.first, self.count = row_chunk(comm.rank, comm.world, m)
        self.chunk = (m + comm.world - 1) // comm.world
        if comm.dist is None:
            self.plan = self._csx.new_handle()
            self._csx.check(self.lib.csx_gaxpy_sharded_plan(block, self.plan), "csx_gaxpy_sharded_plan")
            f, c = self._csx.C.c_int32(), self._csx.C.c_int32()
            self._csx.check(self.lib.csx_gaxpy_sharded_rows(self.plan, f, c), "csx_gaxpy_sharded_rows")
            assert (f.value, c.value) == (self.first, self.count)
        else:
            # host stand-in: the library's own plan cut for this world (the same row pieces, work and arrival buffers and
            # rank-ordered sum as under RCCL); only the wire is gloo, with the pieces staged through the host
            C = self._csx.C
            self.standin = self._csx.new_handle()
            self._csx.check(self.lib.csx_gaxpy_sharded_plan_for(block, comm.world, self.standin), "csx_gaxpy_sharded_plan_for")
            w, r, ch = C.c_void_p(), C.c_void_p(), C.c_int64()
            self._csx.check(self.lib.csx_gaxpy_sharded_buffers(self.standin, w, r, ch), "csx_gaxpy_sharded_buffers")
            assert ch.value == self.chunk
            self._work_ptr, self._recv_ptr = w.value, r.value

    def rows(self):
        return self.first, self.count

    def _slot(self, base, k):
        h = self._csx.new_handle()
        self._csx.check(self.lib.csx_vec_wrap(self._csx.C.c_void_p(base + 8 * self.chunk * k), self.chunk, h), "csx_vec_wrap")
        return h

    def run(self, hx, hy_mine, how=0):
        _csx, lib, comm = self._csx, self.lib, self.comm
        if self.plan is not None:
            _csx.check(lib.csx_gaxpy_sharded(self.plan, hx, hy_mine, how), "csx_gaxpy_sharded")
            return
        W, rank, torch = comm.world, comm.rank, comm.torch
        if how == 0:
            # one SpMV of the whole block into the work buffer, then the sum (gloo: all-reduce + slice), added into y
            full = self._csx.new_handle()
            _csx.check(lib.csx_vec_wrap(_csx.C.c_void_p(self._work_ptr), self.chunk * W, full), "csx_vec_wrap")
            _csx.check(lib.csx_vec_fill(full, 0.0), "csx_vec_fill")
            _csx.check(lib.csx_gaxpy(self.block, hx, full, _csx.GAXPY_AUTO), "csx_gaxpy")
            mine = comm.reduce_scatter_sum(torch.from_numpy(comm._vec_np(full, self.chunk * W))).numpy()
            _csx.free(full)
            y = comm._vec_np(hy_mine, self.chunk)
            y[:self.count] += mine[:self.count]
            _csx.check(lib.csx_vec_write(hy_mine, _csx.pd(y), self.chunk), "csx_vec_write")
            return
        # how == 1, step for step as csx_gaxpy_sharded does it: rotated order, own piece last
        for step in range(1, W + 1):
            q = (rank + step) % W
            _csx.check(lib.csx_gaxpy_sharded_piece(self.standin, q, hx), "csx_gaxpy_sharded_piece")
            if step == W:
                break
            src = (rank - step + W) % W
            hs = self._slot(self._work_ptr, q)
            out = torch.from_numpy(comm._vec_np(hs, self.chunk))
            _csx.free(hs)
            got = torch.empty(self.chunk, dtype=torch.float64)
            rq = comm.dist.isend(out, dst=q)
            comm.dist.recv(got, src=src)
            rq.wait()
            hr = self._slot(self._recv_ptr, src if src < rank else src - 1)
            a = got.numpy()
            _csx.check(lib.csx_vec_write(hr, _csx.pd(a), self.chunk), "csx_vec_write")
            _csx.free(hr)
        _csx.check(lib.csx_gaxpy_sharded_sum(self.standin, rank, W, hy_mine), "csx_gaxpy_sharded_sum")

    def free(self):
        if self.plan is not None:
            self._csx.free(self.plan)
            self.plan = None
        elif getattr(self, "standin", None) is not None:
            self._csx.free(self.standin)
            self.standin = None
