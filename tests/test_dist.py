"""world_size-2 gloo test of the multi-GPU batch driver (sharding + final gather) on CPU.
The per-rank outputs are produced by the oracle here (no GPU in this container); on the GPU box
bench.py feeds the same helpers with libgspx outputs over RCCL."""
import os
import socket

import numpy as np
import pytest

from pygsp_amd import dist as gdist


def test_shard_units_partition():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(gdist.shard_units(n, r, world))
            assert seen == list(range(n))
            sizes = [len(gdist.shard_units(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from scipy import sparse
    from oracle import cheby_oracle as orc
    from tools import torchrun_plumbing as gd
    r, w, _ = gd.init_process_group("gloo")
    assert (r, w) == (rank, world)
    # a batch of 3 independent graphs, sharded 2 + 1
    units = list(gd.shard_units(3, r, w))
    outs = []
    for u in units:
        rng = np.random.default_rng(u)
        A = sparse.random(40, 40, 0.2, random_state=u, format="csr")
        W = A + A.T
        L = orc.laplacian(W)
        lmax = 2 * float(np.ravel(W.sum(0)).max()) + 1e-9
        c = orc.compute_cheby_coeff(orc.heat_kernel(5, lmax), lmax, 8)
        outs.append(orc.cheby_op(L, lmax, c, rng.standard_normal((40, 3))))
    pad = np.zeros((2, 40, 3))
    for i, o in enumerate(outs):
        pad[i] = o
    gd.barrier()
    t = gd.max_over_ranks(float(rank + 1))
    assert t == float(world)
    assert gd.sum_over_ranks(len(units)) == 3.0
    blocks = gd.gather_to_root(torch.from_numpy(pad), dst=0)
    if rank == 0:
        assert len(blocks) == world
        np.save(os.path.join(tmp, "gathered.npy"), torch.stack(blocks).numpy())
    else:
        assert blocks is None
    torch.distributed.destroy_process_group()


def test_gloo_world2_shard_and_gather(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")  # (world, 2, 40, 3)
    from scipy import sparse
    from oracle import cheby_oracle as orc
    flat = [got[0, 0], got[0, 1], got[1, 0]]
    for u in range(3):
        rng = np.random.default_rng(u)
        A = sparse.random(40, 40, 0.2, random_state=u, format="csr")
        W = A + A.T
        L = orc.laplacian(W)
        lmax = 2 * float(np.ravel(W.sum(0)).max()) + 1e-9
        c = orc.compute_cheby_coeff(orc.heat_kernel(5, lmax), lmax, 8)
        ref = orc.cheby_op(L, lmax, c, rng.standard_normal((40, 3)))
        np.testing.assert_allclose(flat[u], ref, rtol=0, atol=0)
    assert np.all(got[1, 1] == 0)
