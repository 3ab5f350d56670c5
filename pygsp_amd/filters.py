"""pygsp.filters surface for the Chebyshev path, backed by libgspx.

Mirrors (names, argument meaning, shapes, exceptions):

* ``compute_cheby_coeff``   pygsp/filters/approximations.py:9-55   (host quadrature, K+1 points)
* ``cheby_op``              pygsp/filters/approximations.py:58-114 -> gspx_cheby_filter (device)
* ``Filter.filter`` & co    pygsp/filters/filter.py:146-391, 506-600
* ``Heat``                  pygsp/filters/heat.py:102-119
* ``MexicanHat``            pygsp/filters/mexicanhat.py:55-84 (+ utils.compute_log_scales)

Only ``method='chebyshev'`` is implemented: it is the accelerated path.  The exact (Fourier)
method is a different algorithm (dense eigendecomposition) and stays with the reference.
"""
import functools

import numpy as np

from . import _capi


def filterbank_handler(func):
    """utils.py:37-53: call `func` once per filter of the bank unless an index `i` is given."""
    @functools.wraps(func)
    def inner(f, *args, **kwargs):
        if "i" in kwargs or f.Nf <= 1:
            return func(f, *args, **kwargs)
        return [func(f, *args, i=i, **kwargs) for i in range(f.Nf)]
    return inner


@filterbank_handler
def compute_cheby_coeff(f, m=30, N=None, *args, **kwargs):
    """Chebyshev coefficients of kernel `i` of filterbank `f` on [0, lmax] (host, float64).

    c[o] = 2/N * sum_j g(a1 cos(pi (j+1/2)/N) + a2) cos(pi o (j+1/2)/N),  a1 = a2 = lmax/2.
    """
    G = f.G
    i = kwargs.pop("i", 0)
    if not N:
        N = m + 1
    a1 = a2 = G.lmax / 2
    j = np.arange(N)
    samples = f._kernels[i](a1 * np.cos(np.pi * (j + 0.5) / N) + a2)
    c = np.empty(m + 1)
    for o in range(m + 1):
        # same operation order as the reference, so the coefficients are bit-identical
        c[o] = 2.0 / N * np.dot(samples, np.cos(np.pi * o * (j + 0.5) / N))
    return c


# How a SINGLE filter's polynomial is evaluated on the device (filterbanks and synthesis always
# use the three-term recurrence):
#   'recurrence' - the reference's three-term Chebyshev recurrence (approximations.py:99-112);
#   'newton'     - the identical polynomial in Newton form on Leja-ordered Chebyshev nodes, by
#                  Horner: a two-term recurrence, 3 instead of 3 2/3 panel passes per order.
EVALUATION = "recurrence"


def set_evaluation(mode):
    global EVALUATION
    if mode not in ("recurrence", "newton"):
        raise ValueError("evaluation must be 'recurrence' or 'newton'")
    EVALUATION = mode


def _leja_order(points):
    """Greedy Leja ordering: each next point maximises the product of distances to the chosen ones
    (keeps the partial products of the Newton basis bounded: stable Horner evaluation)."""
    pts = np.asarray(points, dtype=np.float64)
    n = pts.size
    chosen = np.empty(n)
    left = np.ones(n, dtype=bool)
    first = int(np.argmax(np.abs(pts)))
    chosen[0] = pts[first]
    left[first] = False
    logp = np.full(n, -np.inf)
    with np.errstate(divide="ignore"):
        logp[left] = np.log(np.abs(pts[left] - chosen[0]))
    for k in range(1, n):
        cand = np.where(left, logp, -np.inf)
        i = int(np.argmax(cand))
        chosen[k] = pts[i]
        left[i] = False
        with np.errstate(divide="ignore"):
            logp[left] += np.log(np.abs(pts[left] - chosen[k]))
    return chosen


_newton_cache = {}


def cheb_to_newton(c):
    """Chebyshev coefficients c_0..c_K of p(t) = c_0/2 + sum_k c_k T_k(t) (as compute_cheby_coeff
    returns them) -> (nodes r_0..r_{K-1}, divided differences d_0..d_K) of the SAME polynomial,
    p(t) = sum_j d_j prod_{i<j} (t - r_i).  Computed in 80+4K-digit decimal arithmetic from the
    float64 inputs, so the only rounding is the final conversion of d_j to float64."""
    import decimal
    c = np.asarray(c, dtype=np.float64).ravel()
    key = c.tobytes()
    hit = _newton_cache.get(key)
    if hit is not None:
        return hit
    K = c.size - 1
    if K < 1:
        raise TypeError("The coefficients have an invalid shape")
    D = decimal.Decimal
    with decimal.localcontext() as ctx:
        ctx.prec = 80 + 4 * K
        nodes = _leja_order(np.cos(np.pi * (np.arange(K) + 0.5) / K))
        extra = 0.0 if np.all(np.abs(nodes) > 1e-3) else 0.987654321
        r = [D(float(v)) for v in nodes] + [D(extra)]
        cd = [D(float(v)) for v in c]

        def p(t):
            t0, t1 = D(1), t
            acc = cd[0] / 2 + cd[1] * t1
            for k in range(2, K + 1):
                t0, t1 = t1, 2 * t * t1 - t0
                acc += cd[k] * t1
            return acc

        d = [p(t) for t in r]
        for j in range(1, K + 1):
            for i in range(K, j - 1, -1):
                d[i] = (d[i] - d[i - 1]) / (r[i] - r[i - j])
        out = (np.ascontiguousarray(nodes), np.array([float(v) for v in d]))
    if len(_newton_cache) > 64:
        _newton_cache.clear()
    _newton_cache[key] = out
    return out


def _as_coeff_matrix(c):
    if not isinstance(c, np.ndarray):
        c = np.array(c)
    c = np.atleast_2d(c)
    if c.shape[1] < 2:
        raise TypeError("The coefficients have an invalid shape")
    return np.ascontiguousarray(c, dtype=np.float64)


def cheby_op(G, c, signal, **kwargs):
    """Chebyshev polynomial of the graph Laplacian applied to `signal`, on the GPU.

    Same contract as approximations.py:58-114: `c` is one coefficient vector, a 2-D array or a
    list of vectors; `signal` is (N,) or (N, Nsig) of any real dtype / memory order; returns a
    float64 array of shape (Nf*N,) or (Nf*N, Nsig), block f = rows [f*N, (f+1)*N).
    """
    c = _as_coeff_matrix(c)
    Nf = c.shape[0]
    signal = np.asanyarray(signal)
    if np.iscomplexobj(signal):
        raise TypeError("complex signals are not supported by the Chebyshev path")
    if signal.ndim not in (1, 2) or signal.shape[0] != G.N:
        raise ValueError("First dimension must be the number of vertices "
                         "G.N = {}, got {}.".format(G.N, signal.shape))
    one_d = signal.ndim == 1
    x = signal.reshape(G.N, 1) if one_d else signal
    dev = _device_graph_of(G)
    evaluation = kwargs.get("evaluation") or EVALUATION
    if evaluation not in ("recurrence", "newton"):
        raise ValueError("evaluation must be 'recurrence' or 'newton'")
    if evaluation == "newton" and Nf == 1:
        nodes, dcoef = cheb_to_newton(c[0])
        y, ms = dev.newton_filter(nodes, dcoef, x, G.lmax)
        y = y[None]
    else:
        y, ms = dev.cheby_filter(c, x, G.lmax, _capi.ANALYSIS)
    _record_timing(G, ms)
    r = np.asarray(y, dtype=np.float64).reshape(Nf * G.N, x.shape[1])
    return r[:, 0] if one_d else r


def cheby_rect(G, bounds, signal, **kwargs):
    """Ideal band-pass [bounds[0], bounds[1]] by its Chebyshev expansion (approximations.py:117-163):
    the same recurrence as cheby_op with closed-form coefficients
    c_0 = 2 (b1 - b2)/pi,  c_k = 2/(k pi) (sin k b1 - sin k b2),  b = arccos(2 bounds/lmax - 1)."""
    if not (isinstance(bounds, (list, np.ndarray)) and len(bounds) == 2):
        raise ValueError("Bounds of wrong shape.")
    bounds = np.array(bounds, dtype=np.float64)
    order = int(kwargs.pop("order", 30))
    b1, b2 = np.arccos(2.0 * bounds / G.lmax - 1.0)
    k = np.arange(1, order + 1)
    c = np.empty(order + 1)
    c[0] = 2.0 * (b1 - b2) / np.pi  # cheby_op halves c_0
    c[1:] = 2.0 / (k * np.pi) * (np.sin(k * b1) - np.sin(k * b2))
    return cheby_op(G, c, signal, **kwargs)


def compute_jackson_cheby_coeff(filter_bounds, delta_lambda, m):
    """Chebyshev and Jackson-damped coefficients of the ideal band-pass [a, b] on
    [lambda_min, lambda_max] (approximations.py:166-225).  Unlike the reference this does not
    rescale the caller's `filter_bounds` list in place."""
    if delta_lambda[0] > filter_bounds[0] or delta_lambda[1] < filter_bounds[1]:
        raise ValueError("Bounds of the filter are out of the lambda values")
    if delta_lambda[0] > delta_lambda[1]:
        raise ValueError("lambda_min is greater than lambda_max")
    a1 = (delta_lambda[1] - delta_lambda[0]) / 2
    a2 = (delta_lambda[1] + delta_lambda[0]) / 2
    lo = (filter_bounds[0] - a2) / a1
    hi = (filter_bounds[1] - a2) / a1
    ch = np.empty(m + 1, dtype=float)
    ch[0] = (2 / np.pi) * (np.arccos(lo) - np.arccos(hi))
    i = np.arange(1, m + 1)
    ch[1:] = (2 / (np.pi * i)) * (np.sin(i * np.arccos(lo)) - np.sin(i * np.arccos(hi)))
    alpha = np.pi / (m + 2)
    i = np.arange(m + 1)
    jch = (1 / np.sin(alpha)) * ((1 - i / (m + 2)) * np.sin(alpha) * np.cos(i * alpha)
                                + (1 / (m + 2)) * np.cos(alpha) * np.sin(i * alpha))
    return ch, ch * jch


def _device_graph_of(G):
    """The libgspx graph of `G`: our own Graph builds it on device from W; a reference
    pygsp.graphs.Graph (plugin mode) gets one attached lazily from its host-built G.L."""
    if hasattr(G, "device_graph"):
        return G.device_graph()
    from . import plugin
    return plugin.device_graph_for(G)


def _record_timing(G, ms):
    try:
        G._gspx_last_kernel_ms = ms
    except Exception:
        pass


class Filter:
    """Filterbank of kernels g_i(lambda) on a graph (filter.py:16-110)."""

    def __init__(self, G, kernels):
        self.G = G
        try:
            iter(kernels)
        except TypeError:
            kernels = [kernels]
        self._kernels = kernels
        self.n_features_in, self.n_features_out = (1, len(kernels))
        self.shape = (self.n_features_out, self.n_features_in)
        self.n_filters = self.n_features_in * self.n_features_out
        self.Nf = self.n_filters

    def __len__(self):
        return self.n_filters

    def __getitem__(self, key):
        return Filter(self.G, self._kernels[key])

    def __add__(self, other):
        if not isinstance(other, Filter):
            return NotImplemented
        return Filter(self.G, self._kernels + other._kernels)

    def __matmul__(self, other):
        return self.filter(other)

    def __repr__(self):
        return "{}(in={}, out={})".format(self.__class__.__name__, self.n_features_in,
                                          self.n_features_out)

    def evaluate(self, x):
        """Frequency response, shape (Nf, len(x)) (filter.py:112-144)."""
        x = np.asanyarray(x)
        y = np.empty([self.Nf] + list(x.shape))
        for i, kernel in enumerate(self._kernels):
            y[i] = kernel(x)
        return y

    def filter(self, s, method="chebyshev", order=30):
        """Filter signals (analysis or synthesis), filter.py:146-328.

        Shapes follow the reference exactly: `s` is (N,), (N, Nsig) or (N, Nsig, Nfeat) with
        Nfeat in {1, Nf}; a trailing dimension equal to Nf means synthesis.  The result is
        squeezed.
        """
        s = self.G._check_signal(s)
        if s.ndim == 1 or s.shape[-1] not in [1, self.Nf]:
            if s.ndim == 3:
                raise ValueError("Third dimension (#features) should be either 1 or the number "
                                 "of filters Nf = {}, got {}.".format(self.Nf, s.shape))
            s = np.expand_dims(s, -1)
        n_features_in = s.shape[-1]
        if s.ndim < 3:
            s = np.expand_dims(s, 1)
        n_signals = s.shape[1]
        if s.ndim > 3:
            raise ValueError("At most 3 dimensions: #nodes x #signals x #features.")
        assert s.ndim == 3
        n_features_out = self.Nf if n_features_in == 1 else 1

        if method == "chebyshev":
            c = compute_cheby_coeff(self, m=order)
            if n_features_in == 1:  # analysis
                s = s.squeeze(axis=2)
                s = cheby_op(self.G, c, s)
                # [filter][vertex][signal] buffer -> zero-copy (N, Nsig, Nf) view
                s = s.reshape((self.G.N, n_features_out, n_signals), order="F")
                s = s.swapaxes(1, 2)
            else:  # synthesis: out = sum_f p_f(L) s[:, :, f]  (filter.py:313-322)
                cm = _as_coeff_matrix(c)
                if np.iscomplexobj(s):
                    raise TypeError("complex signals are not supported by the Chebyshev path")
                x = np.ascontiguousarray(np.moveaxis(s, 2, 0))  # (Nf, N, Nsig)
                dev = _device_graph_of(self.G)
                y, ms = dev.cheby_filter(cm, x, self.G.lmax, _capi.SYNTHESIS)
                _record_timing(self.G, ms)
                s = np.expand_dims(np.asarray(y, dtype=np.float64), 2)
        elif method == "exact":
            raise NotImplementedError(
                "method='exact' (dense Fourier filtering, filter.py:292-301) is outside the "
                "accelerated path; use the reference implementation for it.")
        else:
            raise ValueError("Unknown method {}.".format(method))
        return s.squeeze()

    def analyze(self, s, method="chebyshev", order=30):
        if s.ndim == 3 and s.shape[-1] != 1:
            raise ValueError("Last dimension (#features) should be 1, got {}.".format(s.shape))
        return self.filter(s, method, order)

    def synthesize(self, s, method="chebyshev", order=30):
        if s.shape[-1] != self.Nf:
            raise ValueError("Last dimension (#features) should be the number of filters "
                             "Nf = {}, got {}.".format(self.Nf, s.shape))
        return self.filter(s, method, order)

    def localize(self, i, **kwargs):
        """sqrt(N) * filter(delta_i)  (filter.py:350-391)."""
        s = np.zeros(self.G.N)
        s[i] = 1
        return np.sqrt(self.G.N) * self.filter(s, **kwargs)

    def compute_frame(self, **kwargs):
        """Filter the identity: (Nf*N, N) frame matrix (filter.py:506-600)."""
        s = np.identity(self.G.N)
        return self.filter(s, **kwargs).T.reshape(-1, self.G.N)


class Heat(Filter):
    """g(x) = min(exp(-scale * x / lmax), 1) [/ norm]  (heat.py:102-119).  lmax is read when the
    kernel is EVALUATED, like the reference's closure."""

    def __init__(self, G, scale=10, normalize=False):
        try:
            iter(scale)
        except TypeError:
            scale = [scale]
        self.scale = scale
        self.normalize = normalize

        def kernel(x, scale):
            return np.minimum(np.exp(-scale * x / G.lmax), 1)

        kernels = []
        for s in scale:
            norm = np.linalg.norm(kernel(G.e, s)) if normalize else 1
            kernels.append(lambda x, s=s, norm=norm: kernel(x, s) / norm)
        super().__init__(G, kernels)


def compute_log_scales(lmin, lmax, Nscales, t1=1, t2=2):
    """utils.py:312-339: log-spaced wavelet scales from t2/lmin down to t1/lmax."""
    return np.exp(np.linspace(np.log(t2 / lmin), np.log(t1 / lmax), Nscales))


class MexicanHat(Filter):
    """Low-pass 1.2 e^-1 exp(-(x / (0.4 lmin))^4) plus Nf-1 band-passes s x exp(-s x)
    (mexicanhat.py:55-84).  lmin = lmax / lpfactor is captured at construction."""

    def __init__(self, G, Nf=6, lpfactor=20, scales=None, normalize=False):
        self.lpfactor = lpfactor
        self.normalize = normalize
        lmin = G.lmax / lpfactor
        if scales is None:
            scales = compute_log_scales(lmin, G.lmax, Nf - 1)
        self.scales = scales
        if len(scales) != Nf - 1:
            raise ValueError("len(scales) should be Nf-1.")

        kernels = [lambda x: 1.2 * np.exp(-1) * np.exp(-((x / 0.4 / lmin) ** 4))]
        for i in range(Nf - 1):
            def kernel(x, i=i):
                norm = np.sqrt(scales[i]) if normalize else 1
                return norm * (scales[i] * x) * np.exp(-(scales[i] * x))
            kernels.append(kernel)
        super().__init__(G, kernels)
