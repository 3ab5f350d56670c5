"""CPU oracle for nearest-neighbour graph construction.
TEST INFRASTRUCTURE ONLY - see oracle/cheby_oracle.py for who may import this.

Restatement of epfl-lts2/pygsp v0.6.1, pygsp/graphs/nngraphs/nngraph.py:113-297 (NNtype='knn',
use_flann=False) and nngraphs/sensor.py:50-75.  The neighbour search is scipy.spatial.KDTree itself
(the reference's own call, nngraph.py:213-216).

Parity status: PINNED.  tests/test_oracle.py::test_knn_* check it against tests/golden/knn.npz
(NNGraph on a 3-D and a 1-D point cloud with centring and rescaling, Sensor(123, seed=42)), generated
by tests/golden/gen_golden.py from the real reference.
"""
import numpy as np
from scipy import sparse, spatial


def preprocess(Xin, center=True, rescale=True):
    """nngraph.py:120-137."""
    Xin = np.asanyarray(Xin)
    N, d = np.shape(Xin)
    Xout = Xin
    if center:
        Xout = Xin - np.kron(np.ones((N, 1)), np.mean(Xin, axis=0))
    if rescale:
        bounding_radius = 0.5 * np.linalg.norm(np.amax(Xout, axis=0) - np.amin(Xout, axis=0), 2)
        scale = np.power(N, 1.0 / float(min(d, 3))) / 10.0
        Xout = Xout * (scale / bounding_radius)
    return Xout


P_OF = {"euclidean": 2, "manhattan": 1, "max_dist": np.inf}  # dist_translation, nngraph.py:139-145


def knn_query(Xout, k, dist_type="euclidean"):
    """nngraph.py:213-216: D, NN of shape (N, k + 1), self first."""
    kdt = spatial.KDTree(Xout)
    return kdt.query(Xout, k=(k + 1), p=P_OF[dist_type])


def symmetrize(W, method="average"):
    """utils.symmetrize for sparse input, utils.py:247-275."""
    if method == "average":
        return (W + W.T) / 2
    if method == "maximum":
        bigger = W.T > W
        return W - W.multiply(bigger) + W.T.multiply(bigger)
    if method == "fill":
        A = W > 0
        mask = (A + A.T) - A
        W = W + mask.multiply(W.T)
        return symmetrize(W, "average")
    if method in ("tril", "triu"):
        return symmetrize(getattr(sparse, method)(W), "maximum")
    raise ValueError("Unknown symmetrization method {}.".format(method))


def knn_weights(Xout, k, sigma=None, dist_type="euclidean", symmetrize_type="average"):
    """nngraph.py:139-226, 289-297.  Returns (W csr, sigma, NN[:, 1:], D[:, 1:])."""
    N = Xout.shape[0]
    if k >= N:
        raise ValueError("The number of neighbors (k={}) must be smaller "
                         "than the number of nodes ({}).".format(k, N))
    D, NN = knn_query(Xout, k, dist_type)
    if sigma is None:
        sigma = np.mean(D[:, 1:])
    spi = np.repeat(np.arange(N), k)
    spj = NN[:, 1:].ravel()
    spv = np.exp(-np.power(D[:, 1:].ravel(), 2) / float(sigma))
    W = sparse.csc_matrix((spv, (spi, spj)), shape=(N, N))
    W = symmetrize(W, symmetrize_type)  # utils.py:247-275
    W = sparse.csr_matrix(W)
    W.eliminate_zeros()
    return W, float(sigma), NN[:, 1:], D[:, 1:]


def sensor_coords(N, seed=None, distributed=False):
    """nngraphs/sensor.py:56-70."""
    rng = np.random.default_rng(seed)
    if distributed:
        m = np.sqrt(N)
        coords = np.mgrid[0:1:1 / m, 0:1:1 / m].reshape(2, -1).T
        coords += rng.uniform(0, 1 / m, (N, 2))
        return coords
    return rng.uniform(0, 1, (N, 2))


def radius_weights(Xout, epsilon, sigma=None, dist_type="euclidean"):
    """nngraph.py:228-297 (NNtype='radius', euclidean): ball query, distances by
    scipy.spatial.distance.minkowski, sigma = mean neighbour distance, Gaussian weights, 'average'
    symmetrisation.  Vectorised over the neighbour lists (the reference loops in Python; same values).
    Returns (W csr, sigma)."""
    from scipy.spatial import distance
    N = Xout.shape[0]
    kdt = spatial.KDTree(Xout)
    pn = P_OF[dist_type]
    NN = [kdt.query_ball_point(p, r=epsilon, p=pn) for p in Xout]
    rows, cols, dists = [], [], []
    for i, nb in enumerate(NN):
        for j in nb:
            if j != i:
                rows.append(i)
                cols.append(j)
                dists.append(distance.minkowski(Xout[i], Xout[j], p=pn))
    if not dists and sigma is None:
        raise ValueError("No neighbors found")
    dists = np.asarray(dists, dtype=np.float64)
    if sigma is None:
        sigma = np.mean(dists)
    spv = np.exp(-np.power(dists, 2) / float(sigma))
    W = sparse.csc_matrix((spv, (np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64))), shape=(N, N))
    W = (W + W.T) / 2
    return sparse.csr_matrix(W), float(sigma)
