"""Mirror of pygsp.learning for the solver loop that runs on the device.

regression_tikhonov / classification_tikhonov with tau > 0 solve (diag(M) + tau L) x = M y by
conjugate gradients (pygsp/learning.py:324-337, one scipy.sparse.linalg.cg call per column); here
all columns advance together on the GPU with the same recurrence and stopping rule
(gspx_tikhonov_cg_dev).  The tau = 0 branch of the reference is a direct sparse solve
(spsolve, learning.py:342-367), not a sparse-product loop: it is not part of this engine and raises.
"""
import numpy as np


def _to_logits(x):
    """learning.py:36-39."""
    logits = np.zeros([len(x), np.max(x) + 1])
    logits[range(len(x)), x] = 1
    return logits


def regression_tikhonov(G, y, M, tau=0, rtol=1e-5, atol=0.0, maxiter=None):
    """argmin_x ||M x - y||^2 + tau x^T L x for tau > 0 (learning.py:254-337).

    y: (N,) or (N, Nsig) measurements, M: boolean mask of the measured vertices.  rtol / atol /
    maxiter are scipy.sparse.linalg.cg's (the reference uses its defaults).
    """
    y = np.asarray(y)
    M = np.asarray(M)
    if np.prod(M.shape) != G.n_vertices:
        raise ValueError("M should be of size [G.n_vertices,]")
    if not tau > 0:
        raise NotImplementedError("tau = 0 is a direct sparse solve in the reference "
                                  "(learning.py:342-367), not a device path; use tau > 0")
    y = y.copy()
    y[M == False] = 0  # noqa: E712  (learning.py:325-326)
    x, _, _ = G.device_graph().tikhonov_cg(tau, M, y, rtol=rtol, atol=atol, maxiter=maxiter)
    return x


def classification_tikhonov(G, y, M, tau=0, **kwargs):
    """Tikhonov regression of the one-hot encoded labels (learning.py:170-251)."""
    y = np.asarray(y).copy()
    y[np.asarray(M) == False] = 0  # noqa: E712
    Y = _to_logits(y.astype(int))
    return regression_tikhonov(G, Y, M, tau, **kwargs)
