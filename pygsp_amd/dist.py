"""Sharding of a batch over GPUs / ranks.  The Chebyshev recurrence needs no exchange between graphs (or between
signal columns of one graph): the units of a batch are split with NO data-path collective, and the one collective
of the path - the final gather of the outputs - is RCCL inside libgspx (engine.Comm, engine.gather).  Nothing here
(or anywhere in pygsp_amd) imports torch; the torch.distributed plumbing the driver's `torch.distributed.run`
launch form needs lives beside bench.py in tools/torchrun_plumbing.py.
"""
import os


def env_world():
    """(rank, world_size, local_rank) from the launcher's environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_units(n_units, rank, world):
    """Contiguous slice of `n_units` independent units (graphs or signal columns) for `rank`:
    sizes differ by at most one, every unit is owned by exactly one rank."""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))
