"""the path's one collective.  gspx_gather across contexts in one process (peer copies and
the RCCL form), the one-process-per-GPU communicator (gspx_comm_*: RCCL inside the library), and two
ranks under torch.distributed.run sharing this box's one GPU (bench.py's N > 1 code path with libgspx
outputs; RCCL refuses two ranks on one device, so that launch gathers over gloo).  `-m gpu`."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from scipy import sparse

from conftest import csr_from, rel_err
from gpu_helpers import BAR, TOL, ctx, random_graph, upper_lmax  # noqa: F401 (ctx is a fixture)
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, engine, filters, graphs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _flush_c_stdio():
    """RCCL prints a version banner through printf when a communicator is created; flush it inside the test
    that caused it (pytest captures it there) instead of at process exit, after pytest's summary line."""
    yield
    import ctypes
    ctypes.CDLL(None).fflush(None)


def test_gather_and_batch_across_contexts(ctx):
    """gspx_gather: buffers of several contexts concatenated on the root (single-process form of the
    final gather; two contexts on this box's one GPU), and engine.filter_batch on top of it."""
    ctx2 = engine.Context(ctx.device)
    try:
        rng = np.random.default_rng(31)
        a = rng.standard_normal(1000)
        b = rng.standard_normal(333).astype(np.float32)
        pa, pb, pc = ctx.upload(a), ctx2.upload(b), ctx2.alloc(0)
        root = ctx.alloc(a.nbytes + b.nbytes + 16)
        engine.gather([pa, pc, pb], root)
        flat = root.download((a.nbytes + b.nbytes + 16,), np.uint8)
        assert np.array_equal(flat[:a.nbytes].view(np.float64), a)
        assert np.array_equal(flat[a.nbytes:a.nbytes + b.nbytes].view(np.float32), b)
        small = ctx.alloc(8)
        with pytest.raises(ValueError):
            engine.gather([pa, pb], small)
        with pytest.raises(ValueError):
            engine.gather([root], root)
        jobs, refs = [], []
        for i, c_ in enumerate((ctx, ctx2, ctx)):
            W, coords = graphs.sensor_weights(3000 + 500 * i, k=6, seed=40 + i)
            lmax = upper_lmax(W)
            dev = engine.DeviceGraph.from_w(W, dtype=np.float64, perm=engine.locality_order(W, coords), ctx=c_)
            c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, 12)
            x = rng.standard_normal((W.shape[0], 5 + i))
            jobs.append((dev, c, x, lmax))
            refs.append(orc.cheby_op(orc.laplacian(W), lmax, c, x))
        outs = engine.filter_batch(jobs, root_ctx=ctx)
        for o, r in zip(outs, refs):
            assert o.shape == (1,) + r.shape and rel_err(o[0], r) < 1e-12
        for d, _, _, _ in jobs:
            d.destroy()
        for buf in (pa, pb, pc, root, small):  # a context must outlive its buffers
            buf.free()
    finally:
        ctx2.close()


def test_gather_through_rccl_in_the_library(ctx):
    """Option gather_rccl = 2 routes every block of gspx_gather through RCCL (ncclCommInitAll over the
    parts' devices, grouped ncclSend / ncclRecv) - on this one-GPU box as self send / recv, which still
    runs RCCL's communicator set-up, group launch and copy kernels on hardware."""
    if not engine.comm_available():
        pytest.skip("RCCL cannot be loaded on this box")
    ctx2 = engine.Context(ctx.device)
    try:
        rng = np.random.default_rng(5)
        a, b = rng.standard_normal(70000), rng.standard_normal(12345).astype(np.float32)
        pa, pb = ctx.upload(a), ctx2.upload(b)
        root = ctx.alloc(a.nbytes + b.nbytes)
        ctx.set_option("gather_rccl", 2)
        try:
            engine.gather([pa, pb], root)
        finally:
            ctx.set_option("gather_rccl", 1)
        flat = root.download((a.nbytes + b.nbytes,), np.uint8)
        assert np.array_equal(flat[:a.nbytes].view(np.float64), a)
        assert np.array_equal(flat[a.nbytes:].view(np.float32), b)
        for buf in (pa, pb, root):
            buf.free()
    finally:
        ctx2.close()


def test_comm_gather_one_rank(ctx):
    """gspx_comm_*: unique id -> ncclCommInitRank -> grouped send / recv.  One rank here (one GPU): the
    block of a filter output travels to the root buffer through RCCL and arrives bit for bit."""
    if not engine.comm_available():
        pytest.skip("RCCL cannot be loaded on this box")
    W, coords = graphs.sensor_weights(40000, k=6, seed=9)
    lmax = upper_lmax(W)
    dev = engine.DeviceGraph.from_w(W, dtype=np.float64, perm=engine.locality_order(W, coords), ctx=ctx)
    c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, 20)
    x = np.random.default_rng(3).standard_normal((W.shape[0], 16))
    bx, by, broot = ctx.upload(x), ctx.alloc(x.nbytes), ctx.alloc(x.nbytes)
    comm = engine.Comm(ctx, 1, 0, engine.comm_unique_id())
    try:
        dev.cheby_filter_dev(c, bx.ptr, by.ptr, 16, lmax)
        ms = comm.gather(by.ptr, [x.nbytes], 0, broot.ptr)
        assert ms >= 0
        y = broot.download(x.shape, np.float64)
        assert np.array_equal(y, by.download(x.shape, np.float64))
        assert rel_err(y, orc.cheby_op(orc.laplacian(W), lmax, c, x)) < 1e-12
        with pytest.raises(ValueError):
            comm.gather(by.ptr, [x.nbytes, 0], 0, broot.ptr)
    finally:
        comm.close()
        dev.destroy()
        for b in (bx, by, broot):
            b.free()


def test_two_ranks_on_this_gpu_through_bench():
    """bench.py --gpus 2 under torch.distributed.run, both ranks on device 0 (gloo process group): the
    N > 1 path - independent graph per rank, barrier-bracketed timing, MAX over ranks, final gather - runs
    end to end with libgspx outputs; rank 0 prints one JSON line with n_gpus = 2 and a gather time."""
    pytest.importorskip("torch")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # GSPX_BENCH_LIB_GATHER=force: the in-library RCCL gather is attempted although RCCL must refuse two ranks
    # on one device - the attempt (id exchange, communicator creation under the watchdog), the agreement of the
    # ranks on its failure and the fall-back to the launcher's gather all run
    env = dict(os.environ, GSPX_ALL_RANKS_DEVICE0="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GSPX_BENCH_LIB_GATHER="force")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--vertices", "100000", "--nsig", "16",
           "--backend", "gloo", "--no-newton"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["gather_ms"] is not None and out["gather_ms"] > 0 and "torch.distributed (gloo)" in out["gather_impl"]
    assert out["config"]["N"] == 100000 and out["roofline"]["frac"] > 0
    assert out["parity_vs_oracle"]["ranks"] == 2 and out["parity_vs_oracle"]["max_rel_err"] < 1e-11
    b5 = out["batch_config4"]  # BASELINE configs[4]: 8 graphs sharded 4 + 4
    assert b5["n_gpus"] == 2 and b5["n_graphs"] == 8 and b5["value"] > 0 and "4/4" in b5["workload"]
    assert b5["parity_vs_oracle"]["max_rel_err"] < 1e-11
