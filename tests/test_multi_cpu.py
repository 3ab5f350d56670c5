"""pygsp_amd.multi.filter_columns without a GPU: the column sharding, the per-context driver threads and the
reassembly of the gathered blocks into the caller's column order, with oracle-backed stand-ins for the contexts,
buffers and device graphs (test infrastructure: the product path has no such fallback).  The same function runs on
hardware in tests/test_gpu_6_multi.py."""
import numpy as np
import pytest

from conftest import csr_from, rel_err
from oracle import cheby_oracle as orc
from pygsp_amd import _capi, multi


class _Buf:
    _next = [1]
    live = {}

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.data = np.zeros(self.nbytes, dtype=np.uint8)
        self.ptr = _Buf._next[0] << 40
        _Buf._next[0] += 1
        _Buf.live[self.ptr] = self

    def free(self):
        _Buf.live.pop(self.ptr, None)

    def download(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.data[:n].view(dtype).reshape(shape).copy()


class _Ctx:
    def __init__(self, device):
        self.device, self._h = device, object()

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        b = _Buf(self, arr.nbytes)
        b.data[:] = arr.view(np.uint8).reshape(-1)
        return b

    def alloc(self, nbytes):
        return _Buf(self, nbytes)

    def sync(self):
        pass

    def get_option(self, key):
        return 1


class _Dev:
    """engine.DeviceGraph's call contract, arithmetic by the oracle."""

    def __init__(self, L, dtype, ctx):
        self.L, self.dtype, self.ctx, self._h = L, np.dtype(dtype), ctx, object()

    def cheby_filter_dev(self, c, x_ptr, y_ptr, nsig, lmax, mode=_capi.ANALYSIS):
        N, nf = self.L.shape[0], c.shape[0]
        xb, yb = _Buf.live[x_ptr], _Buf.live[y_ptr]
        if mode == _capi.ANALYSIS:
            x = xb.data[:N * nsig * self.dtype.itemsize].view(self.dtype).reshape(N, nsig)
            y = orc.cheby_op(self.L, lmax, c, x.astype(np.float64)).reshape(nf, N, nsig)
        else:
            x = xb.data[:nf * N * nsig * self.dtype.itemsize].view(self.dtype).reshape(nf, N, nsig)
            y = sum(orc.cheby_op(self.L, lmax, c[f], x[f].astype(np.float64)) for f in range(nf)).reshape(N, nsig)
        y = np.ascontiguousarray(y, dtype=self.dtype)
        yb.data[:y.nbytes] = y.view(np.uint8).reshape(-1)
        return 1.0


class _Group(multi.DeviceGroup):
    def __init__(self, devices):  # no libgspx, no device count
        self.devices = list(devices)
        self.ctxs = [_Ctx(d) for d in devices]
        self.n_distinct = len(set(devices))

    def gather(self, parts, root=0):
        parts = [p for p in parts if p is not None]
        out = self.ctxs[root].alloc(max(sum(p.nbytes for p in parts), 16))
        off = 0
        for p in parts:
            out.data[off:off + p.nbytes] = p.data
            off += p.nbytes
        return out, 1e-3, "stand-in"


class _Graph:
    def __init__(self, L, lmax):
        self.L, self.N, self.lmax = L, L.shape[0], lmax


@pytest.mark.parametrize("n_dev,nsig", [(2, 7), (3, 2), (4, 16), (8, 64), (1, 5)])
def test_columns_are_split_and_come_back_in_order(monkeypatch, golden_sensor123, n_dev, nsig):
    g = golden_sensor123
    L, lmax = csr_from(g, "Lcomb"), float(g["lmax"])
    G = _Graph(L, lmax)
    group = _Group(list(range(n_dev)))
    monkeypatch.setattr(multi, "_replicas", lambda G_, grp: [_Dev(L, np.float64, c) for c in grp.ctxs])
    rng = np.random.default_rng(n_dev * 100 + nsig)
    c = np.array([orc.compute_cheby_coeff(k, lmax, 12) for k in orc.mexican_hat_kernels(lmax, 3)])
    x = rng.standard_normal((G.N, nsig))
    tm = {}
    y, _ = multi.filter_columns(G, c, x, group, timings=tm)
    assert sum(tm["columns"]) == nsig and max(tm["columns"]) - min(tm["columns"]) <= 1
    ref = orc.cheby_op(L, lmax, c, x).reshape(3, G.N, nsig)
    assert y.shape == ref.shape and rel_err(y, ref) < 1e-14
    # synthesis: (Nf, N, Nsig) -> (N, Nsig)
    s = rng.standard_normal((3, G.N, nsig))
    ys, _ = multi.filter_columns(G, c, s, group, mode=_capi.SYNTHESIS)
    refs = sum(orc.cheby_op(L, lmax, c[f], s[f]) for f in range(3)).reshape(G.N, nsig)
    assert ys.shape == refs.shape and rel_err(ys, refs) < 1e-13
    assert not _Buf.live  # every buffer of the call was released
    with pytest.raises(ValueError):
        multi.filter_columns(G, c, x[:-1], group)
    with pytest.raises(ValueError):
        multi.filter_columns(G, c, x, group, collect="elsewhere")


def test_a_failing_device_surfaces_its_error(monkeypatch, golden_sensor123):
    g = golden_sensor123
    L, lmax = csr_from(g, "Lcomb"), float(g["lmax"])
    G = _Graph(L, lmax)
    group = _Group([0, 1, 2])

    class Broken(_Dev):
        def cheby_filter_dev(self, *a, **k):
            raise RuntimeError("device 1 fell over")

    monkeypatch.setattr(multi, "_replicas",
                        lambda G_, grp: [(Broken if i == 1 else _Dev)(L, np.float64, c) for i, c in enumerate(grp.ctxs)])
    c = np.atleast_2d(orc.compute_cheby_coeff(orc.heat_kernel(5, lmax), lmax, 8))
    with pytest.raises(RuntimeError, match="device 1"):
        multi.filter_columns(G, c, np.ones((G.N, 6)), group)
    assert not _Buf.live  # the other devices' buffers do not leak


def test_numa_node_of_a_gpu_from_sysfs(tmp_path):
    """multi.numa_cpus_of / parse_cpulist: PCI address -> numa_node -> cpulist, on a fabricated sysfs tree (the
    address itself comes from gspx_device_pci_bus_id on a GPU box); unknown nodes and missing files give None."""
    from pygsp_amd import multi
    assert multi.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert multi.parse_cpulist("") == set()
    dev = tmp_path / "bus/pci/devices/0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices/system/node/node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127,192-255\n")
    cpus = multi.numa_cpus_of(0, sysfs=str(tmp_path), address="0000:C1:00.0")
    assert cpus == set(range(64, 128)) | set(range(192, 256))
    (dev / "numa_node").write_text("-1\n")
    assert multi.numa_cpus_of(0, sysfs=str(tmp_path), address="0000:c1:00.0") is None
    assert multi.numa_cpus_of(0, sysfs=str(tmp_path), address="0000:ff:00.0") is None
    assert multi.numa_cpus_of(0) is None or isinstance(multi.numa_cpus_of(0), set)  # no GPU here: None, never raises
