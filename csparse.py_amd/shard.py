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


class Comm(object):
    """Thin wrapper over torch.distributed; a no-op when world == 1 (no torch import)."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local = env_rank()
        if os.environ.get("CSX_SINGLE_DEVICE"):   # rehearsal of the N > 1 control flow on a one-GPU box
            self.local = 0
        self.dist = None
        self.device = device
        if self.world > 1:
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

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
