"""One process driving several GPUs: SURVEY.md 8(e) without a launcher.

The Chebyshev recurrence needs no exchange between independent graphs nor between the signal columns of one
graph, so both partitionings of SURVEY 8(e) run with NO data-path collective:

* graph-parallel  - a batch of independent graphs, one (or several) per GPU  (``engine.filter_batch``,
  ``bench.py --gpus N``);
* signal-parallel - ONE graph, its device CSR replicated on every GPU (<= 0.5 GB), the Nsig columns split
  N ways (``filter_columns``; ``Filter.filter(..., devices=[...])``; ``plugin.install(devices=[...])``).

A ``DeviceGroup`` holds one libgspx context (= one device + one stream) per entry of `devices` and drives
each from its own thread (ctypes releases the GIL inside libgspx).  The only collective is the final gather
of the outputs onto the root context: ``gspx_gather`` - RCCL (``ncclCommInitAll``, grouped ncclSend / ncclRecv,
one xGMI link per source) inside the library.  No torch anywhere on this path.

An entry of `devices` may repeat a device id (``[0, 0]``): every entry still gets its own context and stream.
That is how the one-GPU test box exercises the threaded path (tests/test_gpu_6_multi.py); it is of no use in
production.
"""
import threading
import time

import numpy as np

from . import _capi, engine
from .dist import shard_units

_extra_ctx = {}
_extra_lock = threading.Lock()


def context_for(device, occurrence=0):
    """The context of the `occurrence`-th entry naming `device`: the process-wide default context of that
    device for the first one, a further long-lived context for every repeat."""
    if occurrence == 0:
        return engine.default_context(device)
    with _extra_lock:
        key = (int(device), int(occurrence))
        ctx = _extra_ctx.get(key)
        if ctx is None or not getattr(ctx, "_h", None):
            ctx = engine.Context(device)
            _extra_ctx[key] = ctx
        return ctx


def numa_cpus_of(device, sysfs="/sys", address=None):
    """The host cores next to HIP device `device`: the cpulist of the NUMA node its PCI function hangs off
    (gspx_device_pci_bus_id -> <sysfs>/bus/pci/devices/<address>/numa_node -> <sysfs>/devices/system/node/node<k>/
    cpulist).  None when the platform does not say (numa_node -1, no sysfs, one node)."""
    import ctypes
    import os
    buf = ctypes.create_string_buffer(64)
    try:
        if address is None:
            _capi.check(_capi.load().gspx_device_pci_bus_id(int(device), buf, 64))
            address = buf.value.decode()
        address = address.strip().lower()
        with open(os.path.join(sysfs, "bus/pci/devices", address, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node/node{}".format(node), "cpulist")) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError, _capi.GspxError):
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11} (the kernel's cpulist format)."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_thread_near(device):
    """Restrict the CALLING thread (and the threads libgspx starts from it: the packers of the host pipeline) to
    the cores of the GPU's NUMA node; pageable-memory staging then reads the local memory controller.  A no-op
    when the node is unknown, when it would leave no allowed core, or with GSPX_NUMA_PIN=0.  Returns the cores
    or None."""
    import os
    if os.environ.get("GSPX_NUMA_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = numa_cpus_of(device)
    if not cpus:
        return None
    allowed = cpus & os.sched_getaffinity(0)
    if not allowed:
        return None
    try:
        os.sched_setaffinity(0, allowed)  # pid 0: the calling thread
    except OSError:
        return None
    return allowed


class DeviceGroup:
    """Contexts of a device list, one driver thread per context."""

    def __init__(self, devices):
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        visible = _capi.device_count()
        bad = [d for d in devices if d < 0 or d >= visible]
        if bad:
            raise ValueError("device {} requested but {} HIP device(s) visible".format(bad[0], visible))
        self.devices = devices
        seen = {}
        self.ctxs = []
        for d in devices:
            self.ctxs.append(context_for(d, seen.get(d, 0)))
            seen[d] = seen.get(d, 0) + 1
        self.n_distinct = len(seen)
        self.pinned = [None] * len(devices)  # cores each driver thread was restricted to (None: not pinned)

    def __len__(self):
        return len(self.ctxs)

    def run(self, fn):
        """fn(i, ctx) on one thread per context; returns the list of results.  The first exception of any
        thread is re-raised here (after every thread has finished)."""
        n = len(self.ctxs)
        out, errs = [None] * n, []

        def work(i):
            try:
                if n > 1 and self.n_distinct > 1:  # a driver thread per GPU: next to its GPU's memory controller
                    self.__dict__.setdefault("pinned", [None] * n)[i] = pin_thread_near(self.devices[i])
                out[i] = fn(i, self.ctxs[i])
            except BaseException as e:  # surfaced on the caller's thread
                errs.append((i, e))

        if n == 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(i,), name="gspx-dev{}".format(i)) for i in range(n)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errs:
            errs.sort(key=lambda p: p[0])
            raise errs[0][1]
        return out

    def sync(self):
        for c in self.ctxs:
            c.sync()

    def gather(self, parts, root=0):
        """parts[i] (a DeviceBuffer of context i, or None) concatenated, in order, on a new buffer of the
        root context: (buffer, seconds, implementation note)."""
        parts = [p for p in parts if p is not None]
        total = sum(p.nbytes for p in parts)
        out = self.ctxs[root].alloc(max(total, 16))
        self.sync()
        t0 = time.perf_counter()
        engine.gather(parts, out)
        dt = time.perf_counter() - t0
        devs = {p.ctx.device for p in parts} | {self.ctxs[root].device}
        if len(devs) > 1 and engine.comm_available() and self.ctxs[root].get_option("gather_rccl") > 0:
            impl = ("libgspx gspx_gather: RCCL (ncclCommInitAll over {} devices), grouped ncclSend/ncclRecv, one xGMI "
                    "link per source".format(len(devs)))
        elif len(devs) > 1:
            impl = "libgspx gspx_gather: peer copies (hipMemcpyPeerAsync on the source streams)"
        else:
            impl = "libgspx gspx_gather: device copies (every part lives on the root's device)"
        return out, dt, impl


# ---- signal-parallel filtering of ONE graph ---------------------------------------------------------------
def _replicas(G, group):
    """One DeviceGraph of `G` per context of the group, cached on the graph object (the CSR is uploaded and the
    Laplacian / tiles are built once per context)."""
    cache = G.__dict__.setdefault("_gspx_replicas", {})
    mirror = hasattr(G, "device_graph")  # pygsp_amd.graphs.Graph; otherwise a reference graph (plugin mode)
    stamp = (id(G._adjacency), G.lap_type, np.dtype(G.compute_dtype).str) if mirror else None
    reps = []
    for ctx in group.ctxs:
        hit = cache.get(id(ctx))
        if mirror:
            # (the entry remembers the context object itself: an id() can be recycled by a later context)
            if (hit is not None and hit[0] == stamp and hit[2] is ctx and getattr(ctx, "_h", None)
                    and getattr(hit[1], "_h", None)):
                reps.append(hit[1])
                continue
            own = G.device_graph()
            if own.ctx is ctx:
                dev = own
            else:
                dev = engine.DeviceGraph.from_w(G._symmetric_w(), G.lap_type, dtype=G.compute_dtype,
                                                perm=G._internal_order(), ctx=ctx)
                if G.tiles == "auto":
                    dev.auto_gather_tiles()
                elif G.tiles:
                    dev.enable_gather_tiles()
            cache[id(ctx)] = (stamp, dev, ctx)
        else:
            from . import plugin
            dev = plugin.device_graph_for(G, ctx=ctx)
        reps.append(dev)
    return reps


def filter_columns(G, coeffs, x, devices, mode=_capi.ANALYSIS, root=0, timings=None, collect="device"):
    """cheby_op of ONE graph with the signal columns split over `devices`.

    coeffs (Nf, M) float64; analysis: x (N, Nsig) -> (Nf, N, Nsig); synthesis: x (Nf, N, Nsig) -> (N, Nsig).
    Columns come back in the caller's order.  `timings` (a dict, optional) receives per-device kernel
    milliseconds, the gather time and its implementation.  `collect`: "device" (default; what the north star
    describes) - every GPU's block travels to the root GPU over xGMI (gspx_gather: RCCL) and the root ships the
    panel to the host; "host" - every GPU ships its own columns to the host itself through the pipelined
    host-pointer path (N PCIe links in parallel, no collective: the better choice when the result is wanted on
    the host anyway)."""
    group = devices if isinstance(devices, DeviceGroup) else DeviceGroup(devices)
    c = np.ascontiguousarray(np.atleast_2d(np.asarray(coeffs, dtype=np.float64)))
    Nf = c.shape[0]
    reps = _replicas(G, group)
    dtype = reps[0].dtype
    x = np.asarray(x)
    analysis = mode == _capi.ANALYSIS
    if analysis:
        if x.ndim != 2 or x.shape[0] != G.N:
            raise ValueError("analysis input must be (N, Nsig), got {}".format(x.shape))
    elif x.ndim != 3 or x.shape[0] != Nf or x.shape[1] != G.N:
        raise ValueError("synthesis input must be (Nf, N, Nsig), got {}".format(x.shape))
    nsig = x.shape[-1]
    n = len(group)
    cols = [shard_units(nsig, r, n) for r in range(n)]
    lmax = float(G.lmax)
    elt = np.dtype(dtype).itemsize
    kernel_ms = [0.0] * n
    if collect not in ("device", "host"):
        raise ValueError("collect must be 'device' or 'host'")
    if collect == "host":
        def direct(i, ctx):
            cr = cols[i]
            if len(cr) == 0:
                return None
            y, kernel_ms[i] = reps[i].cheby_filter(c, x[..., cr.start:cr.stop], lmax, mode)
            return y

        blocks = group.run(direct)
        out = np.empty((Nf, G.N, nsig) if analysis else (G.N, nsig), dtype=dtype)
        for cr, y in zip(cols, blocks):
            if y is not None:
                out[..., cr.start:cr.stop] = y
        if timings is not None:
            timings.update(kernel_ms=list(kernel_ms), gather_ms=0.0, columns=[len(cr) for cr in cols],
                           gather_impl="none: every GPU ships its columns to the host itself",
                           devices=list(group.devices))
        return out, max(kernel_ms)

    parts = [None] * n  # filled by the driver threads; released below also when one of them failed

    def work(i, ctx):
        cr = cols[i]
        w = len(cr)
        if w == 0:
            return
        xs = np.ascontiguousarray(x[..., cr.start:cr.stop], dtype=dtype)
        bx = ctx.upload(xs)
        try:
            parts[i] = ctx.alloc((Nf if analysis else 1) * G.N * w * elt)
            kernel_ms[i] = reps[i].cheby_filter_dev(c, bx.ptr, parts[i].ptr, w, lmax, mode)
        finally:
            bx.free()

    root_buf = None
    try:
        group.run(work)
        root_buf, t_gather, impl = group.gather(parts, root)
        total = sum(p.nbytes for p in parts if p is not None)
        flat = root_buf.download((max(total, 16),), np.uint8)[:total]
    finally:
        for p in parts:
            if p is not None:
                p.free()
        if root_buf is not None:
            root_buf.free()
    out = np.empty((Nf, G.N, nsig) if analysis else (G.N, nsig), dtype=dtype)
    off = 0
    for i, cr in enumerate(cols):
        w = len(cr)
        if w == 0:
            continue
        nb = (Nf if analysis else 1) * G.N * w * elt
        block = flat[off:off + nb].view(dtype)
        out[..., cr.start:cr.stop] = block.reshape((Nf, G.N, w) if analysis else (G.N, w))
        off += nb
    if timings is not None:
        timings.update(kernel_ms=list(kernel_ms), gather_ms=t_gather * 1e3, gather_impl=impl,
                       columns=[len(cr) for cr in cols], devices=list(group.devices))
    return out, max(kernel_ms)
