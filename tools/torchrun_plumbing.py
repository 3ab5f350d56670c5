"""Launcher plumbing for `python -m torch.distributed.run ... bench.py --gpus N` (one process per GPU).

NOT part of the product: pygsp_amd imports no torch.  The Chebyshev recurrence needs no exchange between graphs
(or between signal columns of one graph), so the units of a batch are sharded across ranks with NO data-path
collective (pygsp_amd.dist.shard_units); the only collective is the final gather of the outputs to the root, and
that is RCCL inside libgspx (`make_comm` -> engine.Comm -> gspx_comm_gather: grouped ncclSend / ncclRecv over
xGMI).  torch.distributed is what the driver's launch form brings along, used here for rendezvous, barrier,
scalar reductions of the timings and for carrying the 128-byte RCCL id from rank 0 to the other ranks.
`gather_to_root` is the same exchange through torch.distributed - bench.py's fallback when the in-library
communicator cannot be built (two ranks sharing one GPU in the tests, gloo on CPU).  The single-process
multi-GPU path (`bench.py --gpus N` without a launcher, engine.gather, pygsp_amd.multi) needs none of this.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pygsp_amd.dist import env_world, shard_units  # noqa: E402,F401  (re-exported for the launcher's callers)

def init_process_group(backend=None):
    """Initialise torch.distributed from the environment.  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over all ranks (identity for a single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_to_root(tensor, dst=0):
    """The path's one collective: every rank's output block to `dst`.  Returns the list of blocks
    on the root (rank order), None elsewhere.  All blocks must have the same shape/dtype."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [tensor]
    world = dist.get_world_size()
    rank = dist.get_rank()
    if dist.get_backend() == "nccl":
        # RCCL gather as grouped send/recv: each peer's block lands on its own xGMI link
        if rank == dst:
            out = [torch.empty_like(tensor) for _ in range(world)]
            out[dst].copy_(tensor)
            ops = [dist.P2POp(dist.irecv, out[r], r) for r in range(world) if r != dst]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return out
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, tensor, dst)]):
            req.wait()
        return None
    out = [torch.empty_like(tensor) for _ in range(world)] if rank == dst else None
    dist.gather(tensor, out, dst=dst)
    return out


def exchange_comm_id():
    """The 128-byte RCCL id of a new communicator: made on rank 0 (gspx_comm_unique_id) and handed to the other
    ranks through the launcher's process group - the only thing torch.distributed carries for the gather.
    None on every rank when rank 0 cannot make one (RCCL not loadable)."""
    from pygsp_amd import engine
    rank, world, _ = env_world()
    if world == 1:
        return engine.comm_unique_id()
    import torch.distributed as dist
    box = [None]
    if rank == 0:
        try:
            box[0] = engine.comm_unique_id()
        except Exception:  # RCCL not loadable: every rank learns it (None) instead of waiting for rank 0
            box[0] = None
    dist.broadcast_object_list(box, src=0)
    return box[0]


def make_comm(ctx, unique_id=None):
    """The in-library RCCL communicator of this rank (engine.Comm over gspx_comm_*); the gather itself is RCCL
    inside libgspx.  Single process: a one-rank communicator (its gather is a self send / recv)."""
    from pygsp_amd import engine
    rank, world, _ = env_world()
    return engine.Comm(ctx, world, rank, unique_id if unique_id is not None else exchange_comm_id())
