"""Thin object layer over the libgspx C-ABI: Context, DeviceBuffer, DeviceGraph.

Nothing here computes: every numeric result comes out of the HIP kernels in
pygsp_amd/csrc.  Vertex reordering (a graph-setup step, like building the Laplacian) is the
only host-side preparation, see ``locality_order``.
"""
import ctypes
import sys
import threading

import numpy as np
from scipy import sparse

from . import _capi


class Context:
    """One device + one HIP stream (gspx_ctx).  Calls on one Context must be serialised."""

    def __init__(self, device=0):
        lib = _capi.load()
        h = ctypes.c_void_p()
        _capi.check(lib.gspx_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self._init_pool()

    def _init_pool(self):
        """The recycled-buffer pool and its lock (also called by host-only tests that build a Context without a
        device)."""
        self._pool, self._pooled, self._pool_lock = {}, 0, threading.RLock()

    def close(self):
        """Explicit teardown.  (No __del__: graphs and buffers hold a pointer to their context, so
        a context must outlive them; default contexts simply live until the process exits.)"""
        if getattr(self, "_h", None):
            self.clear_pool()
            _capi.load().gspx_ctx_destroy(self._h)
            self._h = None

    def set_option(self, key, value):
        _capi.check(_capi.load().gspx_ctx_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_int64(0)
        _capi.check(_capi.load().gspx_ctx_get_option(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    def sync(self):
        _capi.check(_capi.load().gspx_ctx_sync(self._h))

    def last_timing(self):
        """dict of HIP-event timings (ms) of the last filter call on this context."""
        out = (ctypes.c_double * 5)()
        _capi.check(_capi.load().gspx_last_timing(self._h, out))
        return {"total_ms": out[0], "steps_ms": out[1], "step_launches": int(out[2]),
                "permute_ms": out[3], "combine_ms": out[4]}

    def last_host_timing(self):
        """Stage times (ms) of the last host-array filter call on this context (gspx_last_host_timing); None
        when that call was not pipelined."""
        out = (ctypes.c_double * 9)()
        _capi.check(_capi.load().gspx_last_host_timing(self._h, out))
        if out[6] == 0:
            return None
        return {"wall_ms": out[0], "pack_ms": out[1], "h2d_ms": out[2], "kernel_ms": out[3], "d2h_ms": out[4],
                "unpack_ms": out[5], "batches": int(out[6]), "signals_per_batch": int(out[7]),
                "host_threads_per_direction": int(out[8])}

    def last_host_timeline(self):
        """(batches, 6) array: host clock (ms since the call began) at which every batch of the last pipelined
        host-array call was packed, had its H2D issued, began / finished its kernels, finished its D2H, was unpacked."""
        n = ctypes.c_int(0)
        _capi.check(_capi.load().gspx_last_host_timeline(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros((max(n.value, 0), 6))
        if n.value > 0:
            _capi.check(_capi.load().gspx_last_host_timeline(self._h, _capi.ptr(out), out.size, ctypes.byref(n)))
        return out

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # Buffers of device-resident arrays (DeviceArray) are recycled: a chain of filters allocates and drops a few
    # panels of the same sizes per call, and hipMalloc / hipFree of gigabyte buffers cost milliseconds each (and a
    # device synchronisation).  Exact-size matches only, a bounded amount kept (POOL_BYTES; the pool never holds more
    # than that, and less as soon as device memory is short: every library call that may allocate goes through
    # call(), which empties the pool and retries once when the allocation fails); clear_pool() frees it.  The pool is
    # touched from DeviceArray.__del__ (any thread the collector runs on) and from user threads: one lock.
    POOL_BYTES = 8 << 30

    def _pool_state(self):
        return self._pool_lock

    def pooled_bytes(self):
        with self._pool_state():
            return self._pooled

    def take(self, nbytes):
        nbytes = max(int(nbytes), 16)
        with self._pool_state():
            stack = self._pool.get(nbytes)
            if stack:
                self._pooled -= nbytes
                return stack.pop()
        return DeviceBuffer(self, nbytes)  # (through call(): trims the pool and retries if the device is full)

    def give(self, buf):
        with self._pool_state():
            if not getattr(buf, "_h", None) or buf.ctx is not self or self._pooled + buf.nbytes > self.POOL_BYTES:
                buf.free()
                return
            self._pool.setdefault(buf.nbytes, []).append(buf)
            self._pooled += buf.nbytes

    def clear_pool(self):
        with self._pool_state():
            stacks, self._pool, self._pooled = list(self._pool.values()), {}, 0
        for stack in stacks:
            for buf in stack:
                buf.free()

    def call(self, fn, *args):
        """rc = fn(*args) for a libgspx entry point that may allocate device memory (buffers, workspaces, graph
        builds), mapped to the reference's exception types.  An ALLOCATION failure (GSPX_ERR_OOM: nothing has been
        launched or written by then - every entry point allocates its workspaces before its first kernel) while the
        pool holds recycled buffers empties the pool and runs the call once more: memory kept for reuse must never be
        the reason a call fails that would have succeeded without the pool.  Any other failure surfaces at
        once: a call that failed midway is never repeated on inputs it may have overwritten (workspaces
        are grow-only and sized by a call's first batch, so an allocation can only fail before the first kernel)."""
        rc = fn(*args)
        if rc == _capi.ERR_OOM and self.pooled_bytes() > 0:
            self.clear_pool()
            rc = fn(*args)
        _capi.check(rc)

    def bench_copy(self, nbytes=1 << 30, iters=10):
        """Measured read+write GB/s of the engine's streaming copy kernel (HBM ceiling)."""
        v = ctypes.c_double(0)
        _capi.check(_capi.load().gspx_bench_copy(self._h, int(nbytes), int(iters), ctypes.byref(v)))
        return v.value

    def bench_read(self, nbytes, passes=50):
        """Read-only GB/s of an nbytes buffer streamed `passes` times in one launch."""
        v = ctypes.c_double(0)
        _capi.check(_capi.load().gspx_bench_read(self._h, int(nbytes), int(passes), ctypes.byref(v)))
        return v.value

    def bench_streams(self, bytes_per_stream, n_read, n_write, nt=0, workgroups_per_cu=8, iters=5):
        """Total GB/s of n_read read streams + n_write write streams walked together (gspx_bench_streams)."""
        v = ctypes.c_double(0)
        _capi.check(_capi.load().gspx_bench_streams(self._h, int(bytes_per_stream), int(n_read), int(n_write), int(nt),
                                                    int(workgroups_per_cu), int(iters), ctypes.byref(v)))
        return v.value

    def bench_gather(self, panel_rows, row_bytes, n_gathers, in_flight=8, blocks=1, p_intra=0.0, workgroups_per_cu=8,
                     iters=5):
        """Rate of random row gathers on this device (gspx_bench_gather): (ms per launch, GB/s of row bytes)."""
        ms, gb = ctypes.c_double(0), ctypes.c_double(0)
        _capi.check(_capi.load().gspx_bench_gather(self._h, int(panel_rows), int(row_bytes), int(n_gathers),
                                                   int(in_flight), int(blocks), float(p_intra), int(workgroups_per_cu),
                                                   int(iters), ctypes.byref(ms), ctypes.byref(gb)))
        return ms.value, gb.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes)
        buf.upload(arr)
        return buf

    def identity_panel(self, buf, N, j0, w, dtype):
        """Write columns [j0, j0 + w) of the N x N identity into `buf` (row-major N x w, on the device)."""
        _capi.check(_capi.load().gspx_identity_panel_dev(self._h, _capi.dtype_code(dtype), int(N), int(j0), int(w),
                                                        ctypes.c_void_p(buf.ptr)))


_default_ctx = {}
_default_lock = threading.Lock()


def default_context(device=0):
    with _default_lock:
        ctx = _default_ctx.get(device)
        if ctx is None:
            ctx = Context(device)
            _default_ctx[device] = ctx
        return ctx


class DeviceBuffer:
    """Device memory owned by libgspx (gspx_buf)."""

    def __init__(self, ctx, nbytes):
        h = ctypes.c_void_p()
        ctx.call(_capi.load().gspx_buf_alloc, ctx._h, int(nbytes), ctypes.byref(h))
        self._h = h
        self.ctx = ctx
        self.nbytes = int(nbytes)

    def free(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):  # a closed context took the device state with it:
                _capi.load().gspx_buf_free(self._h)  # never hand libgspx a dangling context
            self._h = None

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may already be gone; the OS reclaims the memory
            return
        try:
            self.free()
        except Exception:
            pass

    @property
    def ptr(self):
        p = ctypes.c_void_p()
        _capi.check(_capi.load().gspx_buf_ptr(self._h, ctypes.byref(p)))
        return p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        _capi.check(_capi.load().gspx_buf_upload(self._h, _capi.ptr(arr), arr.nbytes))

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        _capi.check(_capi.load().gspx_buf_download(self._h, _capi.ptr(out), out.nbytes))
        return out


class DeviceArray:
    """A signal tensor that lives on the device between calls: what ``Filter.filter`` / ``analyze`` /
    ``synthesize`` return when they are given one, so that a chain of filters (the doctest of filter.py:232-256:
    heat -> analysis -> synthesis) costs one upload and one download instead of a PCIe round trip per call.

    `shape` is the shape the reference's call would have returned (the squeezed (vertices, signals, features)
    cube of filter.py:328); the memory is the engine's layout - `n_features` planes [feature][vertex][signal] in
    the compute dtype, exactly what gspx_cheby_filter_dev reads and writes, so chaining moves nothing.
    ``np.asarray(a)`` / ``a.numpy()`` download it as the float64 array the reference would have returned."""

    __array_priority__ = 100

    def __init__(self, buf, cube, dtype):
        self._buf, self.ctx = buf, buf.ctx
        self.cube = tuple(int(d) for d in cube)  # (vertices, signals, features)
        self.dtype = np.dtype(dtype)
        self.shape = tuple(d for d in self.cube if d != 1)

    @classmethod
    def from_host(cls, ctx, array, dtype=np.float64):
        """Upload a host signal: (N,), (N, Nsig) or (N, Nsig, Nfeat) of any real dtype / memory order."""
        a = np.asanyarray(array)
        if np.iscomplexobj(a):
            raise TypeError("complex signals are not supported by the Chebyshev path")
        if a.ndim < 1 or a.ndim > 3:
            raise ValueError("At most 3 dimensions: #nodes x #signals x #features.")
        cube = a.reshape(a.shape + (1,) * (3 - a.ndim))
        planes = np.ascontiguousarray(np.moveaxis(cube, 2, 0), dtype=dtype)
        buf = ctx.take(max(planes.nbytes, 16))
        if planes.nbytes:
            buf.upload(planes)
        out = cls(buf, cube.shape, dtype)
        out.shape = tuple(a.shape)  # as given (not squeezed): the caller's own shape rules apply to it
        return out

    @classmethod
    def empty(cls, ctx, cube, dtype):
        n = int(np.prod(cube)) * np.dtype(dtype).itemsize
        return cls(ctx.take(max(n, 16)), cube, dtype)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.cube))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ptr(self):
        if self._buf is None:
            raise ValueError("the device array has been freed")
        return self._buf.ptr

    def planes(self, signals, features):
        """Device pointer (+ the object keeping it alive) of this tensor as `features` planes of (N, signals):
        itself when that is how it is stored, else a repacked copy (gspx_planes_pack_dev) - the per-vertex
        elements are re-read in row-major order, as ``reshape`` of the host array would."""
        N, S0, F0 = self.cube
        if S0 * F0 != signals * features:
            raise ValueError("cannot view {} signals x {} features as {} x {}".format(S0, F0, signals, features))
        if (S0, F0) == (signals, features) or self.size == 0:
            return self.ptr, self
        lib, code, cur = _capi.load(), _capi.dtype_code(self.dtype), self
        if F0 != 1:  # planes -> the row-major cube (N, S0 * F0)
            flat = DeviceArray.empty(self.ctx, (N, S0 * F0, 1), self.dtype)
            _capi.check(lib.gspx_planes_pack_dev(self.ctx._h, code, N, S0, F0, ctypes.c_void_p(self.ptr),
                                                 ctypes.c_void_p(flat.ptr), 0))
            cur = flat
        if features != 1:  # the cube (N, signals, features) -> planes
            out = DeviceArray.empty(self.ctx, (N, signals, features), self.dtype)
            _capi.check(lib.gspx_planes_pack_dev(self.ctx._h, code, N, signals, features, ctypes.c_void_p(cur.ptr),
                                                 ctypes.c_void_p(out.ptr), 1))
            cur = out
        return cur.ptr, cur

    def numpy(self):
        """The float64 host array the reference would have returned (shape `self.shape`)."""
        N, S, F = self.cube
        if self.size == 0:
            return np.zeros(self.shape)
        planes = self._buf_download((F, N, S))
        return np.asarray(np.moveaxis(planes, 0, 2), dtype=np.float64).reshape(self.shape)

    def _buf_download(self, shape):
        if self._buf is None:
            raise ValueError("the device array has been freed")
        return self._buf.download(shape, self.dtype)

    def __array__(self, dtype=None, copy=None):
        out = self.numpy()
        return out if dtype is None else out.astype(dtype, copy=False)

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __repr__(self):
        return "DeviceArray(shape={}, dtype={}, device={})".format(self.shape, self.dtype, self.ctx.device)

    def free(self):
        """Give the memory back (to the context's pool of recycled buffers: Context.give)."""
        if self._buf is not None:
            buf, self._buf = self._buf, None
            if getattr(self.ctx, "_h", None):
                self.ctx.sync()  # (nothing queued on the stream may still read or write it when it is handed out again)
                self.ctx.give(buf)
            else:
                buf.free()

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.free()
        except Exception:
            pass


def gather(parts, root_out):
    """Concatenate device buffers living on (possibly) different contexts / GPUs into `root_out`:
    the single-process form of the path's one collective (gspx_gather; peer copies over xGMI)."""
    n = len(parts)
    arr = (ctypes.c_void_p * max(n, 1))(*[b._h for b in parts])
    _capi.check(_capi.load().gspx_gather(None, n, arr, root_out._h))
    return root_out


COMM_ID_BYTES = 128


def comm_available():
    """True when RCCL could be loaded by libgspx (gspx_comm_available)."""
    return bool(_capi.load().gspx_comm_available())


def comm_unique_id():
    """A fresh RCCL unique id (bytes): made on rank 0, handed to the other ranks by the launcher."""
    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    _capi.check(_capi.load().gspx_comm_unique_id(buf))
    return bytes(buf)


def comm_info(comm=None):
    """{"rccl_version", "nranks", "rank"} as RCCL reports them (gspx_comm_info): of `comm`, or - without one - of the
    device set this process's last RCCL gspx_gather ran on (nranks 0: none has run).  Version 0: RCCL is unavailable."""
    out = np.zeros(3, dtype=np.int64)
    _capi.check(_capi.load().gspx_comm_info(comm._h if comm is not None else None, _capi.ptr(out)))
    return {"rccl_version": int(out[0]), "nranks": int(out[1]), "rank": int(out[2])}


class Comm:
    """RCCL communicator of one rank (gspx_comm): the one-process-per-GPU form of the path's only
    collective, the gather of the ranks' output blocks to a root over xGMI."""

    def __init__(self, ctx, nranks, rank, unique_id):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be {} bytes".format(COMM_ID_BYTES))
        buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = ctypes.c_void_p()
        _capi.check(_capi.load().gspx_comm_create(ctx._h, int(nranks), int(rank), buf, ctypes.byref(h)))
        self._h, self.ctx, self.nranks, self.rank = h, ctx, int(nranks), int(rank)

    def gather(self, part_ptr, nbytes_per_rank, root=0, root_out_ptr=None):
        """Collective: this rank's block (device pointer, nbytes_per_rank[rank] bytes) lands in the root's
        buffer at the offset of its rank.  Returns the device milliseconds of the exchange on this rank."""
        table = np.ascontiguousarray(nbytes_per_rank, dtype=np.int64)
        if table.shape != (self.nranks,):
            raise ValueError("nbytes_per_rank must have one entry per rank")
        ms = ctypes.c_double(0)
        _capi.check(_capi.load().gspx_comm_gather(
            self._h, ctypes.c_void_p(part_ptr), _capi.ptr(table), int(root),
            ctypes.c_void_p(root_out_ptr) if root_out_ptr else None, ctypes.byref(ms)))
        return ms.value

    def info(self):
        """What RCCL itself reports about this communicator (gspx_comm_info): version code, ranks, own rank."""
        return comm_info(self)

    def close(self):
        # (after its context is gone the handle is an empty shell inside libgspx - gspx_ctx_destroy took the RCCL
        # communicator down with the stream - and gspx_comm_destroy only frees it)
        if getattr(self, "_h", None):
            _capi.load().gspx_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def filter_batch(jobs, root_ctx=None):
    """Independent (graph, coefficients, signals, lmax) jobs, one driver thread per context: job i
    runs on the context its DeviceGraph lives on (ctypes releases the GIL inside libgspx), the
    outputs are gathered onto `root_ctx` and returned as one list of host arrays.

    jobs: list of (DeviceGraph, c (Nf, K+1) or (K+1,), x (N, Nsig) host array, lmax).
    This is the single-process counterpart of `bench.py --gpus N` (one process per GPU there)."""
    import threading as _th
    root_ctx = root_ctx or jobs[0][0].ctx
    outs = [None] * len(jobs)
    errs = []

    def work(i):
        bx = by = None
        try:
            dev, c, x, lmax = jobs[i]
            c2 = np.atleast_2d(np.asarray(c, dtype=np.float64))
            x = np.ascontiguousarray(x, dtype=dev.dtype)
            bx = dev.ctx.upload(x)
            by = dev.ctx.alloc(x.nbytes * c2.shape[0])
            dev.cheby_filter_dev(c2, bx.ptr, by.ptr, x.shape[1], lmax)
            outs[i] = (by, (c2.shape[0],) + x.shape, dev.dtype)
            by = None  # owned by outs from here on
        except Exception as e:  # surfaced on the caller's thread
            errs.append(e)
        finally:
            for b in (bx, by):
                if b is not None:
                    b.free()

    by_ctx = {}
    for i, j in enumerate(jobs):
        by_ctx.setdefault(id(j[0].ctx), []).append(i)
    threads = [_th.Thread(target=lambda idx=idx: [work(i) for i in idx]) for idx in by_ctx.values()]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    root = None
    try:
        if errs:
            raise errs[0]
        total = sum(b.nbytes for b, _, _ in outs)
        root = root_ctx.alloc(total)
        gather([b for b, _, _ in outs], root)
        flat = root.download((total,), np.uint8)
        res, off = [], 0
        for b, shape, dt in outs:
            res.append(flat[off:off + b.nbytes].view(dt).reshape(shape))
            off += b.nbytes
        return res
    finally:  # every device buffer of the batch is released, also when a job failed
        for o in outs:
            if o is not None:
                o[0].free()
        if root is not None:
            root.free()


def _canonical_csr(M):
    M = sparse.csr_matrix(M)
    if not M.has_canonical_format:
        M = M.copy()
        M.sum_duplicates()
    if M.indices.dtype != np.int32 or M.indptr.dtype != np.int32:
        if M.shape[0] >= 2 ** 31 - 1 or M.nnz >= 2 ** 31 - 1:
            raise ValueError("graph too large for int32 CSR indices")
        M = sparse.csr_matrix((M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32)),
                              shape=M.shape)
    return M


def hilbert_order(coords, bits=16):
    """Vertex order along a 2-D Hilbert curve (no long jumps, unlike the Z-curve)."""
    c = np.asarray(coords, dtype=np.float64)[:, :2]
    lo = c.min(axis=0)
    span = c.max(axis=0) - lo
    span[span == 0] = 1.0
    q = np.minimum(((c - lo) / span * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    x, y = q[:, 0].copy(), q[:, 1].copy()
    d = np.zeros(len(x), dtype=np.int64)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x)
        y = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return np.argsort(d, kind="stable").astype(np.int32)


def locality_order(W, coords=None, curve="auto", device=0, ctx=None):
    """Vertex order used INSIDE the engine (perm[new] = old) so that neighbour gathers hit cache.

    * with coordinates (NN graphs such as Sensor): a space-filling curve through the first two /
      three coordinates - median |i-j| over stored entries drops from ~N/4 to a handful.
      curve="auto": Hilbert in 2-D (no long jumps: a 64-row block of the k=8 sensor graph gathers
      117 distinct rows instead of 123 under the Z-curve, and the LDS-staged step kernel runs 5-8 %
      faster), Morton (Z-order) in 3-D; "morton" / "hilbert" force one;
    * otherwise reverse Cuthill-McKee on the pattern of W.
    The order is invisible to callers: inputs/outputs stay in the graph's own vertex order.
    """
    N = W.shape[0]
    if N < 2:
        return None
    has_coords = coords is not None and np.ndim(coords) == 2 and coords.shape[0] == N and coords.shape[1] >= 2
    if has_coords and N >= 4096 and _capi.device_count() > 0:
        # curve keys and their stable argsort (a radix sort) on the device: gspx_curve_order
        hil = curve == "hilbert" or (curve == "auto" and coords.shape[1] == 2)
        c = np.ascontiguousarray(coords, dtype=np.float64)
        perm = np.empty(N, dtype=np.int32)
        _capi.check(_capi.load().gspx_curve_order((ctx or default_context(device))._h, N, c.shape[1], _capi.ptr(c),
                                                  1 if hil else 0, _capi.ptr(perm)))
        return perm
    if has_coords and (curve == "hilbert" or (curve == "auto" and coords.shape[1] == 2)):
        return hilbert_order(coords)
    if has_coords:
        c = np.asarray(coords, dtype=np.float64)[:, :3]
        lo = c.min(axis=0)
        span = c.max(axis=0) - lo
        span[span == 0] = 1.0
        d = c.shape[1]
        bits = 21 if d == 3 else 31
        q = np.minimum(((c - lo) / span * (2 ** bits - 1)).astype(np.uint64), 2 ** bits - 1)
        code = np.zeros(N, dtype=np.uint64)
        for b in range(bits):
            for k in range(d):
                code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * d + k)
        return np.argsort(code, kind="stable").astype(np.int32)
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    pattern = sparse.csr_matrix((np.ones(W.nnz, dtype=np.int8), W.indices, W.indptr), shape=W.shape)
    return np.asarray(reverse_cuthill_mckee(pattern, symmetric_mode=True), dtype=np.int32)


def locality_score(W, perm=None, reach=None, sample=None):
    """Fraction of stored entries whose two vertices are at most `reach` positions apart in the
    given order (perm[new] = old; None = the graph's own order): a proxy for how many neighbour
    gathers find their row in the L2 of the XCD that sweeps that index range."""
    W = sparse.csr_matrix(W)
    if W.nnz == 0:
        return 1.0
    if reach is None:
        reach = min(8192, max(64, W.shape[0] // 64))  # ~ rows of a 64-signal panel one L2 holds
    if sample is not None and W.nnz > sample:  # every stride-th stored entry
        pos = np.arange(0, W.nnz, W.nnz // sample, dtype=np.int64)
        rows = np.searchsorted(W.indptr, pos, side="right") - 1
        cols = W.indices[pos]
    else:
        coo = W.tocoo()
        rows, cols = coo.row, coo.col
    if perm is None:
        r, c = rows, cols
    else:
        inv = np.empty(W.shape[0], dtype=np.int64)
        inv[np.asarray(perm, dtype=np.int64)] = np.arange(W.shape[0])
        r, c = inv[rows], inv[cols]
    return float(np.mean(np.abs(r.astype(np.int64) - c.astype(np.int64)) <= reach))


def expander_like(W, seeds=12):
    """True when balls around sample vertices grow like in a random graph: the number of NEW vertices at distance 3
    is about (mean degree) x the number at distance 2 - Erdos-Renyi, block models, small-world graphs - instead of
    the r^(d-1) of a graph embedded in a few dimensions (k-NN graphs, meshes: a ratio of 1.5-2.5).  No vertex order
    gives such a graph locality, so reverse Cuthill-McKee (a host algorithm, ~1 s at 10M entries) is not worth
    running on it.  Looks at a few hundred rows of the CSR arrays only; `W` need not be validated."""
    W = W if sparse.isspmatrix_csr(W) else sparse.csr_matrix(W)
    N = W.shape[0]
    if N < 64 or W.nnz == 0:
        return False
    indptr, indices = W.indptr, W.indices
    mean_deg = W.nnz / N
    ratios = []
    for s in np.linspace(0, N - 1, seeds).astype(np.int64):
        ball = frontier = np.array([s], dtype=np.int64)
        sizes = []
        for _ in range(3):
            if frontier.size == 0 or frontier.size > 20000:
                break
            nbrs = np.unique(np.concatenate([indices[indptr[v]:indptr[v + 1]] for v in frontier]).astype(np.int64))
            nbrs = nbrs[(nbrs >= 0) & (nbrs < N)]
            new = np.setdiff1d(nbrs, ball, assume_unique=True)
            ball = np.union1d(ball, new)
            frontier = new
            sizes.append(new.size)
        if len(sizes) == 3 and sizes[1] > 0:
            ratios.append(sizes[2] / sizes[1])
    if len(ratios) < max(3, seeds // 3):
        return False
    return mean_deg >= 4 and float(np.median(ratios)) > 0.5 * mean_deg


def auto_order(W, coords=None, device=0, ctx=None):
    """The internal order `reorder='auto'` picks: Morton order when coordinates exist, otherwise
    reverse Cuthill-McKee - but only if it beats the graph's own order on `locality_score`
    (block-structured graphs such as a sorted SBM are already local; RCM would scramble them)."""
    has_coords = coords is not None and np.ndim(coords) == 2 and np.shape(coords)[0] == W.shape[0] and np.shape(coords)[1] >= 2
    if not has_coords and expander_like(W):
        return None  # a random-like graph: no order helps, skip the reverse Cuthill-McKee pass
    perm = locality_order(W, coords, device=device, ctx=ctx)
    if perm is None:
        return None
    # (scored on a sample of the entries: coordinates may be a plotting layout unrelated to the edges)
    if locality_score(W, perm, sample=200000) < locality_score(W, None, sample=200000) + 0.05:
        return None
    return perm


class DeviceGraph:
    """Device-resident Laplacian (gspx_graph): built on the GPU from W, or uploaded as L."""

    def __init__(self, handle, ctx, N, dtype):
        self._h = handle
        self.ctx = ctx
        self.N = int(N)
        self.dtype = np.dtype(dtype)

    @classmethod
    def from_w(cls, W, lap_type="combinatorial", dtype=np.float64, perm=None, ctx=None):
        """graph.py:510-630 on device.  W must be symmetric (undirected)."""
        ctx = ctx or default_context()
        if lap_type not in ("combinatorial", "normalized"):
            raise ValueError("Unknown Laplacian type {}".format(lap_type))
        W = _canonical_csr(W)
        data = W.data
        if data.dtype not in (np.float32, np.float64):
            data = data.astype(np.float64)  # int64 adjacency (ER/SBM) -> float64, as scipy does
        data = np.ascontiguousarray(data)
        lap = _capi.LAP_COMBINATORIAL if lap_type == "combinatorial" else _capi.LAP_NORMALIZED
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        h = ctypes.c_void_p()
        ctx.call(_capi.load().gspx_graph_create_from_w,
            ctx._h, W.shape[0], W.nnz, _capi.ptr(W.indptr), _capi.ptr(W.indices), _capi.ptr(data),
            _capi.dtype_code(data.dtype), lap, _capi.dtype_code(dtype), _capi.ptr(p),
            ctypes.byref(h))
        return cls(h, ctx, W.shape[0], dtype)

    ORDER_MODES = {"none": 0, None: 0, False: 0, "auto": 1, "morton": 2, "hilbert": 3}

    @classmethod
    def setup(cls, W, lap_type="combinatorial", dtype=np.float64, coords=None, order="auto", ctx=None):
        """Graph set-up in one device call (gspx_graph_setup): W - a scipy CSR matrix, NOT checked on the host - is
        uploaded once, validated, inspected (NaN / inf / negative / zero / diagonal entries, symmetry), given its
        internal vertex order (curve order of `coords`: 'auto' / 'morton' / 'hilbert' / 'none', or a permutation
        array) and turned into the device Laplacian.  Returns (DeviceGraph or None, report): None when the graph
        is directed or stores explicit zeros - the caller prepares W on the host (graph.py:613-616) and uses
        from_w.  Raises ValueError for NaN / inf entries (the reference's messages) and for a non-canonical CSR."""
        ctx = ctx or default_context()
        if lap_type not in ("combinatorial", "normalized"):
            raise ValueError("Unknown Laplacian type {}".format(lap_type))
        if not sparse.isspmatrix_csr(W):
            W = sparse.csr_matrix(W)
        N = W.shape[0]
        indptr, indices, data = W.indptr, W.indices, W.data
        if indptr.dtype != np.int32 or indices.dtype != np.int32:
            if N >= 2 ** 31 - 1 or W.nnz >= 2 ** 31 - 1:
                raise ValueError("graph too large for int32 CSR indices")
            indptr, indices = indptr.astype(np.int32), indices.astype(np.int32)
        if data.dtype == np.float32:
            code = _capi.F32
        elif data.dtype == np.float64:
            code = _capi.F64
        elif data.dtype == np.int64:
            code = 2  # int64 adjacency (ER / SBM): converted on the device
        else:
            data, code = data.astype(np.float64), _capi.F64
        c = np.ascontiguousarray
        mode, xy, d, perm_in = cls._order_args(N, coords, order)
        report = np.zeros(12, dtype=np.int64)
        h = ctypes.c_void_p()
        ctx.call(_capi.load().gspx_graph_setup,
            ctx._h, N, W.nnz, _capi.ptr(c(indptr)), _capi.ptr(c(indices)), _capi.ptr(c(data)), code,
            _capi.LAP_COMBINATORIAL if lap_type == "combinatorial" else _capi.LAP_NORMALIZED, _capi.dtype_code(dtype),
            _capi.ptr(xy), d, mode, _capi.ptr(perm_in), _capi.ptr(report), ctypes.byref(h))
        return (cls(h, ctx, N, dtype) if h.value else None), cls._setup_report(report)

    @classmethod
    def _order_args(cls, N, coords, order):
        """(order_mode, coordinates or None, their dimension, permutation or None) of the set-up calls."""
        perm_in, xy, d = None, None, 0
        if isinstance(order, np.ndarray) or isinstance(order, (list, tuple)):
            perm_in, mode = np.ascontiguousarray(order, dtype=np.int32), 4
        else:
            mode = cls.ORDER_MODES[order]
        if mode in (1, 2, 3):
            ok = coords is not None and np.ndim(coords) == 2 and np.shape(coords)[0] == N and np.shape(coords)[1] >= 2
            if ok and N >= 4096:
                xy = np.ascontiguousarray(coords, dtype=np.float64)
                d = xy.shape[1]
            else:
                mode = 0
        return mode, xy, d, perm_in

    @staticmethod
    def _setup_report(report):
        return {"nan": int(report[0]), "inf": int(report[1]), "negative": int(report[2]), "zeros": int(report[3]),
                "self_loops": int(report[4]), "asymmetric": int(report[5]), "reordered": bool(report[7]),
                "locality_own": report[8] / 1e9, "locality_curve": report[9] / 1e9, "built": report[10] == 0,
                "setup_ms": report[11] / 1e3}

    @classmethod
    def setup_from(cls, adjacency, lap_type="combinatorial", dtype=np.float64, coords=None, order="auto"):
        """The same set-up for a DeviceAdjacency - the W a device builder (knn_graph / radius_graph / sbm_graph with
        keep_on_device=True) left on the device: nothing is downloaded or uploaded (gspx_graph_setup_from_knn)."""
        if lap_type not in ("combinatorial", "normalized"):
            raise ValueError("Unknown Laplacian type {}".format(lap_type))
        N = adjacency.shape[0]
        mode, xy, d, perm_in = cls._order_args(N, coords, order)
        report = np.zeros(12, dtype=np.int64)
        h = ctypes.c_void_p()
        adjacency.ctx.call(_capi.load().gspx_graph_setup_from_knn,
            adjacency._h, _capi.LAP_COMBINATORIAL if lap_type == "combinatorial" else _capi.LAP_NORMALIZED,
            _capi.dtype_code(dtype), _capi.ptr(xy), d, mode, _capi.ptr(perm_in), _capi.ptr(report), ctypes.byref(h))
        return (cls(h, adjacency.ctx, N, dtype) if h.value else None), cls._setup_report(report)

    def lmax_bounds(self):
        """(max W, max dw, max (dw_i + dw_j) over entries, max (dw_i + (W dw)_i / dw_i)) taken on the device while W
        was there (gspx_graph_lmax_bounds), or None when the graph carries none."""
        out = np.zeros(4)
        try:
            _capi.check(_capi.load().gspx_graph_lmax_bounds(self._h, _capi.ptr(out)))
        except ValueError:
            return None
        return out

    def download_perm(self):
        """The internal vertex order (perm[new] = old), or None when the graph keeps its own order."""
        perm = np.empty(self.N, dtype=np.int32)
        try:
            _capi.check(_capi.load().gspx_graph_download_perm(self._h, _capi.ptr(perm)))
        except ValueError:
            return None
        return perm

    @classmethod
    def from_l(cls, L, dtype=np.float64, perm=None, ctx=None):
        ctx = ctx or default_context()
        L = _canonical_csr(L)
        data = L.data
        if data.dtype not in (np.float32, np.float64):
            data = data.astype(np.float64)
        data = np.ascontiguousarray(data)
        p = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
        h = ctypes.c_void_p()
        ctx.call(_capi.load().gspx_graph_create_from_l,
            ctx._h, L.shape[0], L.nnz, _capi.ptr(L.indptr), _capi.ptr(L.indices), _capi.ptr(data),
            _capi.dtype_code(data.dtype), _capi.dtype_code(dtype), _capi.ptr(p), ctypes.byref(h))
        return cls(h, ctx, L.shape[0], dtype)

    def destroy(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):
                _capi.load().gspx_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.destroy()
        except Exception:
            pass

    def _i64(self, fn):
        v = ctypes.c_int64(0)
        _capi.check(fn(self._h, ctypes.byref(v)))
        return v.value

    @property
    def nnz_l(self):
        return self._i64(_capi.load().gspx_graph_nnz_l)

    @property
    def nnz_internal(self):
        return self._i64(_capi.load().gspx_graph_nnz_internal)

    @property
    def build_ms(self):
        v = ctypes.c_double(0)
        _capi.check(_capi.load().gspx_graph_build_ms(self._h, ctypes.byref(v)))
        return v.value

    def download_l(self):
        """Canonical Laplacian as scipy CSR in the graph's own vertex order."""
        nnz = self.nnz_l
        indptr = np.empty(self.N + 1, dtype=np.int32)
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz, dtype=self.dtype)
        _capi.check(_capi.load().gspx_graph_download_l(self._h, _capi.ptr(indptr),
                                                       _capi.ptr(indices), _capi.ptr(data)))
        return sparse.csr_matrix((data, indices, indptr), shape=(self.N, self.N))

    def download_dw(self):
        dw = np.empty(self.N, dtype=self.dtype)
        _capi.check(_capi.load().gspx_graph_download_dw(self._h, _capi.ptr(dw)))
        return dw

    def lanczos_lmax(self, max_iter=60, tol=5e-4):
        """Largest Ritz value of L after a device Lanczos run: (value, iterations).  Raises ValueError
        when the step budget runs out before the residual criterion is met, as Graph.estimate_lmax
        does on ArpackNoConvergence (graph.py:918-919): a value from below lambda_max would put part of
        the spectrum outside [-1, 1], where the Chebyshev recurrence diverges."""
        v = ctypes.c_double(0)
        it = ctypes.c_int(0)
        ok = ctypes.c_int(0)
        self.ctx.call(_capi.load().gspx_lanczos_lmax, self._h, int(max_iter), float(tol), ctypes.byref(v),
                                                   ctypes.byref(it), ctypes.byref(ok))
        if not ok.value:
            raise ValueError("The Lanczos method did not converge. Try to use bounds.")
        return v.value, it.value

    # ---- the hot path -------------------------------------------------------------------------
    def cheby_filter(self, coeffs, x, lmax, mode=_capi.ANALYSIS):
        """Host arrays in / out.  coeffs: (Nf, M) float64.
        analysis:  x (N, Nsig)      -> (Nf, N, Nsig)
        synthesis: x (Nf, N, Nsig)  -> (N, Nsig)
        Returns (y, kernel_ms)."""
        c = np.ascontiguousarray(np.atleast_2d(np.asarray(coeffs, dtype=np.float64)))
        Nf, M = c.shape
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if mode == _capi.ANALYSIS:
            if x.ndim != 2 or x.shape[0] != self.N:
                raise ValueError("analysis input must be (N, Nsig), got {}".format(x.shape))
            nsig = x.shape[1]
            y = np.empty((Nf, self.N, nsig), dtype=self.dtype)
        else:
            if x.ndim != 3 or x.shape[0] != Nf or x.shape[1] != self.N:
                raise ValueError("synthesis input must be (Nf, N, Nsig), got {}".format(x.shape))
            nsig = x.shape[2]
            y = np.empty((self.N, nsig), dtype=self.dtype)
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_cheby_filter,
            self._h, float(lmax), Nf, M, _capi.ptr(c), nsig, _capi.ptr(x), _capi.ptr(y), mode,
            ctypes.byref(ms))
        return y, ms.value

    def cheby_filter_dev(self, coeffs, x_ptr, y_ptr, nsig, lmax, mode=_capi.ANALYSIS):
        """Device pointers in / out (ints).  Returns device milliseconds of the whole call."""
        c = np.ascontiguousarray(np.atleast_2d(np.asarray(coeffs, dtype=np.float64)))
        Nf, M = c.shape
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_cheby_filter_dev,
            self._h, float(lmax), Nf, M, _capi.ptr(c), int(nsig), ctypes.c_void_p(x_ptr),
            ctypes.c_void_p(y_ptr), mode, ctypes.byref(ms))
        return ms.value


    def tune_placement(self, coeffs, x_ptr, y_ptr, nsig, lmax, candidates=6, stride_mb=0):
        """Draw `candidates` physical backings for the context's streamed workspaces and keep the one on which THIS call
        (the arguments of cheby_filter_dev, one filter) runs fastest (gspx_ctx_tune_placement; on MI355X the same call
        runs 0.54-0.60 of 8 TB/s depending on which pages back the work panels).  Returns {"launch_ms": [per
        candidate], "kept": index}; y holds the call's result.  Results are bit-identical whichever backing is kept.
        stride_mb > 0: a pad of that size is held before every further draw, so the candidates sample the card's
        memory at that stride (fast and slow pages come in zones of tens of GB); candidates that no longer fit read 0."""
        c = np.ascontiguousarray(np.asarray(coeffs, dtype=np.float64).ravel())
        out = np.zeros(int(candidates) + 1)
        self.ctx.call(_capi.load().gspx_ctx_tune_placement, self._h, float(lmax), int(c.size), _capi.ptr(c), int(nsig),
                      ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), int(candidates), int(stride_mb), _capi.ptr(out))
        return {"launch_ms": [float(v) for v in out[:-1]], "kept": int(out[-1])}

    def bench_step_mix(self, coeffs, x_ptr, y_ptr, nsig, lmax, mode=1):
        """CALIBRATION, not a filter: the launches cheby_filter_dev(coeffs, ...) would make with the row products
        removed from every wide k_step_tile launch (gspx_bench_step_mix; mode 2: the barriers of a pass too).  y
        receives numbers without meaning.  Returns ctx.last_timing() of that call."""
        c = np.ascontiguousarray(np.asarray(coeffs, dtype=np.float64).ravel())
        self.ctx.call(_capi.load().gspx_bench_step_mix, self._h, float(lmax), int(c.size), _capi.ptr(c), int(nsig),
                      ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), int(mode))
        return self.ctx.last_timing()


def _newton_methods():
    def newton_filter(self, nodes, dcoef, x, lmax):
        """Newton-form evaluation (single filter): host arrays in/out, x (N, Nsig) -> (N, Nsig)."""
        nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        dcoef = np.ascontiguousarray(dcoef, dtype=np.float64)
        if dcoef.size != nodes.size + 1:
            raise ValueError("need K nodes and K+1 coefficients")
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if x.ndim != 2 or x.shape[0] != self.N:
            raise ValueError("input must be (N, Nsig), got {}".format(x.shape))
        y = np.empty_like(x)
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_newton_filter,
            self._h, float(lmax), int(nodes.size), _capi.ptr(nodes), _capi.ptr(dcoef), x.shape[1],
            _capi.ptr(x), _capi.ptr(y), ctypes.byref(ms))
        return y, ms.value

    def program_filter(self, program, x, lmax, old_is_x=False):
        """A polynomial program (gspx_poly_program; filters.cheb_to_product builds the product form's): host arrays in /
        out, x (N, Nsig) -> (N, Nsig).  program: (S, 3) rows of (scale, beta, gamma)."""
        prog = np.ascontiguousarray(program, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=self.dtype)
        if x.ndim != 2 or x.shape[0] != self.N:
            raise ValueError("input must be (N, Nsig), got {}".format(x.shape))
        y = np.empty_like(x)
        ms = ctypes.c_double(0)
        cols = [np.ascontiguousarray(prog[:, k]) for k in range(3)]
        self.ctx.call(_capi.load().gspx_poly_program, self._h, float(lmax), int(prog.shape[0]), _capi.ptr(cols[0]),
                      _capi.ptr(cols[1]), _capi.ptr(cols[2]), int(bool(old_is_x)), x.shape[1], _capi.ptr(x), _capi.ptr(y),
                      ctypes.byref(ms))
        return y, ms.value

    def program_filter_dev(self, program, x_ptr, y_ptr, nsig, lmax, old_is_x=False):
        prog = np.ascontiguousarray(program, dtype=np.float64).reshape(-1, 3)
        cols = [np.ascontiguousarray(prog[:, k]) for k in range(3)]
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_poly_program_dev, self._h, float(lmax), int(prog.shape[0]), _capi.ptr(cols[0]),
                      _capi.ptr(cols[1]), _capi.ptr(cols[2]), int(bool(old_is_x)), int(nsig), ctypes.c_void_p(x_ptr),
                      ctypes.c_void_p(y_ptr), ctypes.byref(ms))
        return ms.value

    def newton_filter_dev(self, nodes, dcoef, x_ptr, y_ptr, nsig, lmax):
        nodes = np.ascontiguousarray(nodes, dtype=np.float64)
        dcoef = np.ascontiguousarray(dcoef, dtype=np.float64)
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_newton_filter_dev,
            self._h, float(lmax), int(nodes.size), _capi.ptr(nodes), _capi.ptr(dcoef), int(nsig),
            ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), ctypes.byref(ms))
        return ms.value

    def download_internal(self):
        """(rowptr, col) of the engine's internal padded CSR (internal vertex order)."""
        rp = np.empty(self.N + 1, dtype=np.int32)
        col = np.empty(self.nnz_internal, dtype=np.int32)
        _capi.check(_capi.load().gspx_graph_download_internal(self._h, _capi.ptr(rp), _capi.ptr(col)))
        return rp, col

    def enable_gather_tiles(self):
        """Build (numpy) and upload the one-level row tiles of the LDS-staged recurrence step
        (k_step_tile): per 64-row block the distinct rows it gathers, per entry a 16-bit position in
        that list.  With them (and option "tile_gather" = 1, the default) single-filter Chebyshev
        filtering stages the gathered panel in LDS.  Returns tile statistics."""
        from . import tiling
        rp, col = self.download_internal()
        t = tiling.build_tiles(rp, col, self.N, 64)
        lidx = t["lidx1"].copy()
        lidx[lidx == tiling.PAD] = 0  # pads carry the value 0: any valid position will do
        c = np.ascontiguousarray
        stats = np.zeros(3, dtype=np.int64)
        _capi.check(_capi.load().gspx_graph_set_gather_tiles(
            self._h, 64, t["nb"], _capi.ptr(c(t["s1ptr"])), _capi.ptr(c(t["s1rows"])), _capi.ptr(c(lidx)),
            _capi.ptr(stats)))
        return {"nb": t["nb"], "max_n1": t["max_n1"], "mean_n1": t["mean_n1"],
                "slow_blocks": int(stats[1]), "lds_bytes": int(stats[2])}

    def build_gather_tiles(self):
        """The same tiles as enable_gather_tiles(), computed on the device (gspx_graph_build_gather_tiles):
        milliseconds instead of half a second of numpy at N = 1M."""
        stats = np.zeros(4, dtype=np.int64)
        self.ctx.call(_capi.load().gspx_graph_build_gather_tiles, self._h, _capi.ptr(stats))
        nb = int(stats[0])
        return {"nb": nb, "mean_n1": float(stats[3]) / max(nb, 1), "slow_blocks": int(stats[1]),
                "lds_bytes": int(stats[2])}

    def auto_gather_tiles(self, min_vertices=32768):
        """enable_gather_tiles() when it pays: a large graph whose internal order is local (nearly
        every 64-row block's gather set fits its LDS tile).  Returns the statistics (with
        "enabled") or None when the graph is too small to bother."""
        if self.N < min_vertices:
            return None
        st = build_gather_tiles(self)  # on the device: cheap enough to just try
        st["enabled"] = st["slow_blocks"] * 50 <= st["nb"]
        if not st["enabled"]:
            disable_gather_tiles(self)
        return st

    def disable_gather_tiles(self):
        _capi.check(_capi.load().gspx_graph_set_gather_tiles(self._h, 0, 0, None, None, None, None))

    DeviceGraph.newton_filter = newton_filter
    DeviceGraph.newton_filter_dev = newton_filter_dev
    DeviceGraph.program_filter = program_filter
    DeviceGraph.program_filter_dev = program_filter_dev
    DeviceGraph.download_internal = download_internal
    DeviceGraph.enable_gather_tiles = enable_gather_tiles
    DeviceGraph.disable_gather_tiles = disable_gather_tiles
    DeviceGraph.auto_gather_tiles = auto_gather_tiles
    DeviceGraph.build_gather_tiles = build_gather_tiles


_newton_methods()


def _ops_methods():
    """Operators that reuse the device CSR next to the Chebyshev path (include/gspx_ext.h, SURVEY 8(f)
    row 3).  Host arrays in / out; the *_dev variants take device pointers."""

    def _panel(self, x, rows, what):
        x = np.asarray(x)
        one_d = x.ndim == 1
        x2 = np.ascontiguousarray(x.reshape(x.shape[0], -1), dtype=self.dtype)
        if x2.shape[0] != rows:
            raise ValueError("{}: first dimension must be {}, got {}".format(what, rows, x.shape))
        return x2, one_d

    def laplacian_apply_dev(self, x_ptr, y_ptr, nsig):
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_laplacian_apply_dev,
            self._h, int(nsig), ctypes.c_void_p(x_ptr), ctypes.c_void_p(y_ptr), ctypes.byref(ms))
        return ms.value

    def laplacian_apply(self, x):
        """L x for x of shape (N,) or (N, Nsig)."""
        x2, one_d = _panel(self, x, self.N, "laplacian_apply")
        bx, by = self.ctx.upload(x2), self.ctx.alloc(max(x2.nbytes, 16))
        try:
            laplacian_apply_dev(self, bx.ptr, by.ptr, x2.shape[1])
            y = by.download(x2.shape, self.dtype)
        finally:
            bx.free()
            by.free()
        return y[:, 0] if one_d else y

    def dirichlet_energy_dev(self, x_ptr, nsig):
        gram = np.zeros((int(nsig), int(nsig)), dtype=np.float64)
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_dirichlet_energy_dev,
            self._h, int(nsig), ctypes.c_void_p(x_ptr), _capi.ptr(gram), ctypes.byref(ms))
        return gram, ms.value

    def dirichlet_energy(self, x):
        """x^T L x: a float for one signal, the (Nsig, Nsig) matrix x.T @ (L @ x) for a panel
        (what Graph.dirichlet_energy returns for 2-D input, graph.py:701-702)."""
        x2, one_d = _panel(self, x, self.N, "dirichlet_energy")
        bx = self.ctx.upload(x2)
        try:
            gram, _ = dirichlet_energy_dev(self, bx.ptr, x2.shape[1])
        finally:
            bx.free()
        return float(gram[0, 0]) if one_d else gram

    def tikhonov_cg_dev(self, tau, mask_ptr, y_ptr, x_ptr, nsig, rtol=1e-5, atol=0.0, maxiter=None):
        maxiter = 10 * self.N if maxiter is None else int(maxiter)
        iters = np.zeros(max(int(nsig), 1), dtype=np.int32)
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_tikhonov_cg_dev,
            self._h, float(tau), ctypes.c_void_p(mask_ptr), int(nsig), ctypes.c_void_p(y_ptr),
            ctypes.c_void_p(x_ptr), float(rtol), float(atol), maxiter, _capi.ptr(iters), ctypes.byref(ms))
        return iters[:int(nsig)], ms.value

    def tikhonov_cg(self, tau, mask, y, rtol=1e-5, atol=0.0, maxiter=None):
        """Solve (diag(mask) + tau L) x = mask * y per column by conjugate gradients (scipy's cg
        recurrence and stopping rule).  Returns (x, iterations per column, device ms)."""
        y2, one_d = _panel(self, y, self.N, "tikhonov_cg")
        m = np.ascontiguousarray(np.asarray(mask).reshape(-1) != 0, dtype=self.dtype)
        if m.size != self.N:
            raise ValueError("M should be of size [G.n_vertices,]")
        bm, by = self.ctx.upload(m), self.ctx.upload(y2)
        bx = self.ctx.alloc(max(y2.nbytes, 16))
        try:
            iters, ms = tikhonov_cg_dev(self, tau, bm.ptr, by.ptr, bx.ptr, y2.shape[1], rtol, atol, maxiter)
            x = bx.download(y2.shape, self.dtype)
        finally:
            bm.free()
            by.free()
            bx.free()
        return (x[:, 0] if one_d else x), iters, ms

    def n_edges(self):
        v = ctypes.c_int64(0)
        _capi.check(_capi.load().gspx_graph_n_edges(self._h, ctypes.byref(v)))
        return v.value

    def edge_list(self, with_d=False):
        """(sources, targets, weights) in Graph.get_edge_list order; with_d adds D's two values per edge."""
        E = n_edges(self)
        src, dst = np.empty(E, dtype=np.int32), np.empty(E, dtype=np.int32)
        w, ds, dt = (np.empty(E, dtype=self.dtype) for _ in range(3))
        _capi.check(_capi.load().gspx_graph_download_edges(
            self._h, _capi.ptr(src), _capi.ptr(dst), _capi.ptr(w), _capi.ptr(ds), _capi.ptr(dt)))
        return (src, dst, w, ds, dt) if with_d else (src, dst, w)

    def differential_operator(self):
        """D as scipy csc (N x n_edges), assembled on the host from the device edge arrays."""
        src, dst, _, ds, dt = edge_list(self, with_d=True)
        E = src.size
        rows = np.concatenate([src, dst])
        cols = np.concatenate([np.arange(E), np.arange(E)])
        D = sparse.csc_matrix((np.concatenate([ds, dt]), (rows, cols)), shape=(self.N, E))
        D.eliminate_zeros()  # the two values of a self-loop cancel (difference.py:166)
        return D

    def set_edge_list(self, sources, targets, weights, directed):
        """Replace the edge list of grad / div / D by the caller's (gspx_graph_set_edge_list): the edges of a
        directed graph or of a graph with self-loops, in Graph.get_edge_list order (sources non-decreasing)."""
        src = np.ascontiguousarray(sources, dtype=np.int32)
        dst = np.ascontiguousarray(targets, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if not (src.shape == dst.shape == w.shape and src.ndim == 1):
            raise ValueError("sources, targets and weights must be vectors of one length")
        _capi.check(_capi.load().gspx_graph_set_edge_list(self._h, src.size, _capi.ptr(src), _capi.ptr(dst),
                                                          _capi.ptr(w), 1 if directed else 0))

    def grad_dev(self, x_ptr, y_ptr, nsig):
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_grad_dev, self._h, int(nsig), ctypes.c_void_p(x_ptr),
                                               ctypes.c_void_p(y_ptr), ctypes.byref(ms))
        return ms.value

    def div_dev(self, y_ptr, z_ptr, nsig):
        ms = ctypes.c_double(0)
        self.ctx.call(_capi.load().gspx_div_dev, self._h, int(nsig), ctypes.c_void_p(y_ptr),
                                              ctypes.c_void_p(z_ptr), ctypes.byref(ms))
        return ms.value

    def _edge_op(self, x, rows_in, rows_out, fn, what):
        x2, one_d = _panel(self, x, rows_in, what)
        out_bytes = rows_out * x2.shape[1] * np.dtype(self.dtype).itemsize
        bx, by = self.ctx.upload(x2), self.ctx.alloc(max(out_bytes, 16))
        try:
            fn(self, bx.ptr, by.ptr, x2.shape[1])
            y = by.download((rows_out, x2.shape[1]), self.dtype)
        finally:
            bx.free()
            by.free()
        return y[:, 0] if one_d else y

    def grad(self, x):
        """D^T x: (N,) or (N, Nsig) -> (n_edges,) or (n_edges, Nsig)."""
        return _edge_op(self, x, self.N, n_edges(self), grad_dev, "grad")

    def div(self, y):
        """D y: (n_edges,) or (n_edges, Nsig) -> (N,) or (N, Nsig)."""
        return _edge_op(self, y, n_edges(self), self.N, div_dev, "div")

    for f in (laplacian_apply_dev, laplacian_apply, dirichlet_energy_dev, dirichlet_energy,
              tikhonov_cg_dev, tikhonov_cg, n_edges, edge_list, differential_operator, set_edge_list, grad_dev,
              div_dev, grad, div):
        setattr(DeviceGraph, f.__name__, f)


_ops_methods()


METRICS = {"euclidean": 0, "manhattan": 1, "max_dist": 2}
SYMMETRIZE = {"average": 0, "maximum": 1, "fill": 1, "tril": 2, "triu": 3}


class DeviceAdjacency:
    """The weight matrix a device builder produced, still on the device (a gspx_knn handle).  Graph() takes it in
    place of a scipy matrix (DeviceGraph.setup_from): the host copy is made by download(), once, when somebody
    reads G.W.  `weights`: dtype of the host copy (the block-model samplers hand out int64 ones like the
    reference)."""

    def __init__(self, handle, ctx, N, nnz, weights=np.float64):
        self._h, self.ctx, self.shape, self.nnz, self.weights = handle, ctx, (int(N), int(N)), int(nnz), np.dtype(weights)
        self._host = None

    def download(self):
        if self._host is None:
            if not self._h:
                raise RuntimeError("the device adjacency has been released")
            N = self.shape[0]
            indptr = np.empty(N + 1, dtype=np.int32)
            indices = np.empty(self.nnz, dtype=np.int32)
            data = np.empty(self.nnz, dtype=np.float64)
            _capi.check(_capi.load().gspx_knn_download_w(self._h, _capi.ptr(indptr), _capi.ptr(indices), _capi.ptr(data)))
            if self.weights != np.float64:
                data = np.ones(self.nnz, dtype=self.weights) if self.weights.kind in "iu" else data.astype(self.weights)
            self._host = sparse.csr_matrix((data, indices, indptr), shape=self.shape)
            self.close()
        return self._host

    def close(self):
        h, self._h = self._h, None
        if h:
            _capi.load().gspx_knn_destroy(h)

    def __del__(self):
        if sys.is_finalizing():  # the HIP runtime may already be gone; the OS reclaims the memory
            return
        try:
            self.close()
        except Exception:
            pass


def knn_graph(coords, k, sigma=None, ctx=None, neighbors=False, metric="euclidean", symmetrize="average",
              keep_on_device=False):
    """k-nearest-neighbour weights on the device (gspx_knn_build): the KD-tree query, Gaussian weights
    and symmetrisation of NNGraph (nngraph.py:213-226, 289-297) in 1 to 64 dimensions (a uniform grid in 1-3-D;
    beyond that a tiled brute force whose pair distances run on the matrix cores).  coords: (N, d), already centred / rescaled.  Returns (W csr float64, sigma, info)
    where info = {"build_ms": ...} plus "NN", "D" (N x k, nearest first) when neighbors=True.  keep_on_device: W is
    returned as a DeviceAdjacency (not downloaded)."""
    ctx = ctx or default_context()
    X = np.ascontiguousarray(coords, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("coords must be (N, d)")
    N, d = X.shape
    lib = _capi.load()
    h = ctypes.c_void_p()
    _capi.check(lib.gspx_knn_build(ctx._h, N, d, _capi.ptr(X), int(k), float(sigma or 0.0), METRICS[metric],
                                   SYMMETRIZE[symmetrize], ctypes.byref(h)))
    try:
        nnz, sg, ms = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        _capi.check(lib.gspx_knn_info(h, ctypes.byref(nnz), ctypes.byref(sg), ctypes.byref(ms)))
        info = {"build_ms": ms.value}
        if neighbors:
            NN = np.empty((N, int(k)), dtype=np.int32)
            D = np.empty((N, int(k)), dtype=np.float64)
            _capi.check(lib.gspx_knn_download_neighbors(h, _capi.ptr(NN), _capi.ptr(D)))
            info["NN"], info["D"] = NN, D
        W = DeviceAdjacency(h, ctx, N, nnz.value)
        h = None
    finally:
        if h:
            lib.gspx_knn_destroy(h)
    return (W if keep_on_device else W.download()), sg.value, info


def radius_graph(coords, epsilon, sigma=None, ctx=None, metric="euclidean", keep_on_device=False):
    """Radius-graph weights on the device (gspx_radius_build): NNtype='radius' of NNGraph
    (nngraph.py:228-287) in 1 to 64 dimensions (a grid of epsilon-sized cells up to 3-D; beyond that the candidates of
    an MFMA distance sweep, tested in the KD-tree's arithmetic).  Returns (W csr float64, sigma, info)."""
    ctx = ctx or default_context()
    X = np.ascontiguousarray(coords, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("coords must be (N, d)")
    N, d = X.shape
    lib = _capi.load()
    h = ctypes.c_void_p()
    _capi.check(lib.gspx_radius_build(ctx._h, N, d, _capi.ptr(X), float(epsilon), float(sigma or 0.0),
                                      METRICS[metric], ctypes.byref(h)))
    try:
        nnz, sg, ms = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        _capi.check(lib.gspx_knn_info(h, ctypes.byref(nnz), ctypes.byref(sg), ctypes.byref(ms)))
        W = DeviceAdjacency(h, ctx, N, nnz.value)
        h = None
    finally:
        if h:
            lib.gspx_knn_destroy(h)
    return (W if keep_on_device else W.download()), sg.value, {"build_ms": ms.value}


def sbm_graph(z, M, seed=None, ctx=None, keep_on_device=False, directed=False, self_loops=False):
    """Stochastic-block-model adjacency sampled on the device (gspx_sbm_build_ex): every unordered pair
    of distinct vertices (r, c) is an edge with probability M[z[r], z[c]], unit weights
    (stochasticblockmodel.py:125-144; one block = Erdos-Renyi).  directed: every ORDERED pair is an entry
    W[r, c] on its own (M need not be symmetric); self_loops: the pairs r == c take part.
    z: (N,) block labels in 0..k-1, M: (k, k).  Returns (W csr float64, build_ms)."""
    ctx = ctx or default_context()
    z = np.asarray(z)
    M = np.ascontiguousarray(M, dtype=np.float64)
    k = M.shape[0]
    if M.shape != (k, k) or z.ndim != 1 or (z.size and (z.min() < 0 or z.max() >= k)):
        raise ValueError("z must hold labels 0..k-1 and M must be k x k")
    N = z.size
    order = np.ascontiguousarray(np.argsort(z, kind="stable"), dtype=np.int32)
    bounds = np.ascontiguousarray(np.searchsorted(z[order], np.arange(k + 1)), dtype=np.int64)
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
    lib = _capi.load()
    h = ctypes.c_void_p()
    flags = (1 if directed else 0) | (2 if self_loops else 0)
    _capi.check(lib.gspx_sbm_build_ex(ctx._h, N, k, _capi.ptr(order), _capi.ptr(bounds), _capi.ptr(M),
                                      ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), flags, ctypes.byref(h)))
    try:
        nnz, sg, ms = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        _capi.check(lib.gspx_knn_info(h, ctypes.byref(nnz), ctypes.byref(sg), ctypes.byref(ms)))
        # (keep_on_device: unit int64 weights in the host copy, like the reference's W)
        W = DeviceAdjacency(h, ctx, N, nnz.value, weights=np.int64 if keep_on_device else np.float64)
        h = None
    finally:
        if h:
            lib.gspx_knn_destroy(h)
    return (W if keep_on_device else W.download()), ms.value


def plan_describe(coeffs, ctx=None):
    """The engine's step schedule for these coefficients (host-only; for schedule tests)."""
    c = np.ascontiguousarray(np.atleast_2d(np.asarray(coeffs, dtype=np.float64)))
    Nf, M = c.shape
    if M < 2:
        _capi.check(_capi.load().gspx_plan_describe(None, Nf, M, _capi.ptr(c), None))
    plan = np.zeros((M - 1, 4 + 3 * Nf), dtype=np.float64)
    _capi.check(_capi.load().gspx_plan_describe(ctx._h if ctx else None, Nf, M, _capi.ptr(c),
                                                _capi.ptr(plan)))
    return plan
