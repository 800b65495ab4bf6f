"""One-process-per-GPU sharding of the hot path.

The path shards as independent units -- right-hand-side blocks of a batched
solve, or whole independent matrices -- so there is NO collective on that data
path.  The one case with a real exchange step is a single SpMV sharded by
columns: every rank produces a full-length partial y and a reduce-scatter sums
them (reduce_scatter_sum).  What ranks exchange is control only: a barrier around timed regions, the
max of a timing over ranks, a small object broadcast (e.g. the kernel choice made
by rank 0), and optionally a gather of per-block results to rank 0.  On the GPU
node that traffic goes over RCCL (torch.distributed backend "nccl"); the same
code runs on "gloo" for the CPU tests.

Nothing here touches libcsx: it is pure plumbing and is unit-tested with gloo,
world_size 2 (tests/test_shard_gloo.py).
"""
import os


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


class _DevArray(object):
    """A device buffer owned by libcsx, described to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_from_ptr(ptr, count, dtype, device):
    """torch tensor VIEW of `count` elements at device pointer `ptr` (libcsx keeps ownership): how a factor
    that lives behind csx handles is handed to RCCL.  dtype: 'f64' or 'i32'."""
    import torch
    t = torch.as_tensor(_DevArray(ptr, count, {"f64": "<f8", "i32": "<i4"}[dtype]), device=device)
    assert t.data_ptr() == int(ptr), "torch copied instead of viewing the libcsx buffer"
    return t


class Comm(object):
    """Thin wrapper over torch.distributed; a no-op when world == 1 (no torch import)."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local = env_rank()
        if os.environ.get("CSX_SINGLE_DEVICE"):   # rehearsal of the N > 1 control flow on a one-GPU box
            self.local = 0
        self.dist = None
        self.device = device
        if self.world > 1 or os.environ.get("CSX_FORCE_DIST"):   # CSX_FORCE_DIST: a real process group of one
            import torch
            import torch.distributed as dist
            if backend is None:
                backend = os.environ.get("CSX_COMM_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.local)
                self.device = torch.device("cuda", self.local)
                dist.init_process_group(backend, device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend)
            self.dist = dist
            self.torch = torch

    def barrier(self, sync=None):
        """Device sync (callable) then a rank barrier."""
        if sync is not None:
            sync()
        if self.dist is not None:
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def broadcast_object(self, obj, src=0):
        if self.dist is None:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def all_gather_object(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def gather_blocks(self, block, dst=0):
        """Gather equally shaped 2-D float64 blocks (n x k_r, same k_r on every rank) to `dst`,
        concatenated along columns in rank order.  Returns the full block on dst, None elsewhere.
        This is the optional 'solutions back to the root' step; it is never inside a timed region."""
        if self.dist is None:
            return block
        t = self.torch.as_tensor(block, dtype=self.torch.float64, device=self.device).contiguous()
        out = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(t, out, dst=dst)
        if self.rank != dst:
            return None
        return self.torch.cat(out, dim=1)

    def reduce_scatter_sum(self, full):
        """Sum the ranks' full-length 1-D float64 tensors and leave rank r with rows
        [r*len/world, (r+1)*len/world) -- the exchange step of a column-sharded SpMV.
        len(full) must be a multiple of world.  RCCL reduce-scatter on the GPU node; gloo (the CPU tests)
        has no reduce-scatter, so there it is an all-reduce followed by a slice.  world == 1: the input."""
        if self.dist is None:
            return full
        n = full.numel()
        if n % self.world:
            raise ValueError("reduce_scatter_sum: length %d is not a multiple of world %d" % (n, self.world))
        chunk = n // self.world
        if self.dist.get_backend() == "nccl":
            out = self.torch.empty(chunk, dtype=full.dtype, device=full.device)
            self.dist.reduce_scatter_tensor(out, full, op=self.dist.ReduceOp.SUM)
            return out
        tmp = full.detach().to("cpu", copy=True)       # gloo: staged through the host
        self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM)
        return tmp[self.rank * chunk:(self.rank + 1) * chunk].to(full.device, copy=True)

    # ---- data-path exchanges of the batched cs_cholsol (first bullet) -----------------
    # On the GPU node these are RCCL collectives on device tensors over xGMI.  gloo (CPU tests, and the
    # one-GPU rehearsal with CSX_COMM_BACKEND=gloo) carries host tensors only, so device tensors are
    # staged through the host there: same call sequence, same results.

    def _stage(self, t):
        return self.dist.get_backend() != "nccl" and t.device.type != "cpu"

    def broadcast_tensor(self, t, src=0):
        """In-place broadcast of one contiguous tensor from `src` (factor once, ship L.p / L.i / L.x)."""
        if self.dist is None:
            return t
        if self._stage(t):
            h = t.detach().to("cpu", copy=True)
            self.dist.broadcast(h, src=src)
            if self.rank != src:
                t.copy_(h)
            return t
        self.dist.broadcast(t, src=src)
        return t

    def scatter_blocks(self, out, blocks, src=0):
        """Rank r receives blocks[r] (held by `src`, same shape as `out` everywhere) into `out`: the
        right-hand-side blocks of a batched solve leaving the root.  RCCL has no native scatter in every
        torch build, so it is written as the point-to-point pattern scatter is: W - 1 sends from the root."""
        if self.dist is None:
            out.copy_(blocks[0])
            return out
        stage = self._stage(out)
        if self.rank == src:
            reqs = []
            for r in range(self.world):
                if r == src:
                    out.copy_(blocks[r])
                else:
                    b = blocks[r].detach().to("cpu", copy=True) if stage else blocks[r]
                    reqs.append(self.dist.isend(b, dst=r))
            for q in reqs:
                q.wait()
        else:
            if stage:
                h = self.torch.empty(out.shape, dtype=out.dtype, device="cpu")
                self.dist.recv(h, src=src)
                out.copy_(h)
            else:
                self.dist.recv(out, src=src)
        return out

    def gather_to_root(self, block, dst=0):
        """Every rank's equally shaped block to `dst`: list of W tensors there (rank order), None elsewhere.
        Point-to-point like scatter_blocks (W - 1 receives at the root; the root's inbound links are the
        bound, SURVEY 8e: ~36 GB at 8 x 5.12 GB)."""
        if self.dist is None:
            return [block]
        stage = self._stage(block)
        if self.rank == dst:
            outs, reqs = [], []
            for r in range(self.world):
                if r == dst:
                    outs.append(block)
                else:
                    o = self.torch.empty(block.shape, dtype=block.dtype, device="cpu" if stage else block.device)
                    outs.append(o)
                    reqs.append(self.dist.irecv(o, src=r))
            for q in reqs:
                q.wait()
            return [o.to(block.device) if stage and o is not block else o for o in outs]
        b = block.detach().to("cpu", copy=True) if stage else block
        self.dist.send(b, dst=dst)
        return None

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
