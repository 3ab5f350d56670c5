"""Mirror of pygsp.learning for the solver loop that runs on the device.

regression_tikhonov / classification_tikhonov with tau > 0 solve (diag(M) + tau L) x = M y by
conjugate gradients (pygsp/learning.py:324-337, one scipy.sparse.linalg.cg call per column); here
all columns advance together on the GPU with the same recurrence and stopping rule
(gspx_tikhonov_cg_dev).  The tau = 0 branch of the reference is a direct sparse solve
(spsolve, learning.py:342-367), not a sparse-product loop: it is not part of this engine and raises.
"""
import numpy as np


def _one_hot(labels):
    """Integer class labels -> (N, classes) indicator matrix (what learning.py:36-39 calls logits)."""
    labels = np.asarray(labels, dtype=np.int64)
    return (labels[:, np.newaxis] == np.arange(labels.max() + 1)[np.newaxis, :]).astype(np.float64)


def _measured_only(y, mask):
    """A float copy of y with every unmeasured vertex set to zero (learning.py:325-326)."""
    keep = np.asarray(mask).reshape(-1).astype(bool)
    out = np.array(y, copy=True)
    out[~keep] = 0
    return out, keep


def regression_tikhonov(G, y, M, tau=0, rtol=1e-5, atol=0.0, maxiter=None):
    """argmin_x ||M x - y||^2 + tau x^T L x for tau > 0 (learning.py:254-337).

    y: (N,) or (N, Nsig) measurements, M: boolean mask of the measured vertices.  rtol / atol /
    maxiter are scipy.sparse.linalg.cg's (the reference uses its defaults).
    """
    if np.size(M) != G.n_vertices:
        raise ValueError("M should be of size [G.n_vertices,]")
    if not tau > 0:
        raise NotImplementedError("tau = 0 is a direct sparse solve in the reference "
                                  "(learning.py:342-367), not a device path; use tau > 0")
    rhs, keep = _measured_only(y, M)
    solution, _, _ = G.device_graph().tikhonov_cg(tau, keep, rhs, rtol=rtol, atol=atol, maxiter=maxiter)
    return solution


def classification_tikhonov(G, y, M, tau=0, **kwargs):
    """Tikhonov regression of the one-hot encoded labels (learning.py:170-251)."""
    labels, _ = _measured_only(y, M)
    return regression_tikhonov(G, _one_hot(labels), M, tau, **kwargs)
