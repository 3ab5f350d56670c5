#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (pygsp v0.6.1 from /root/reference).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/gen_golden.py
The fixtures are committed; tests read them, never /root/reference.
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "agg")
import pygsp  # noqa: E402
from pygsp import filters, graphs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def csr_parts(M, prefix):
    M = M.tocsr()
    M.sort_indices()
    return {prefix + "_indptr": M.indptr.astype(np.int32), prefix + "_indices": M.indices.astype(np.int32),
            prefix + "_data": M.data, prefix + "_shape": np.array(M.shape)}


def logo():
    """BASELINE.json configs[0]: graphs.Logo() + Heat(scale=50), 3 deltas, order 30."""
    G = graphs.Logo()
    out = csr_parts(G.W, "W")
    out.update(csr_parts(G.L, "L"))
    s = np.zeros(G.N)
    s[[20, 30, 1090]] = 1
    out["signal"] = s
    out["dw"] = G.dw
    for tag, setter in (("bounds", lambda: G.estimate_lmax("bounds")),
                        ("fourier", lambda: G.compute_fourier_basis())):
        setter()
        g = filters.Heat(G, scale=50)
        out["lmax_" + tag] = np.float64(G.lmax)
        out["coeff_" + tag] = filters.compute_cheby_coeff(g, m=30)
        out["y_" + tag] = g.filter(s, method="chebyshev", order=30)
    np.savez_compressed(os.path.join(OUT, "logo_heat50.npz"), **out)


def sensor123():
    """The fixture of pygsp/tests/test_filters.py:12-29: Sensor(123, seed=42), exact lmax,
    uniform signal from default_rng(42)."""
    G = graphs.Sensor(123, seed=42)
    G.compute_fourier_basis()
    rng = np.random.default_rng(42)
    sig = rng.uniform(size=G.N)
    out = csr_parts(G.W, "W")
    out.update(csr_parts(G.L, "Lcomb"))
    out["coords"] = G.coords
    out["lmax"] = np.float64(G.lmax)
    out["dw"] = G.dw
    out["signal"] = sig
    sigs = rng.standard_normal((G.N, 5))
    out["signals5"] = sigs
    # single filter
    g = filters.Heat(G, scale=10)
    out["heat10_c"] = filters.compute_cheby_coeff(g, m=30)
    out["heat10_y"] = g.filter(sig, method="chebyshev", order=30)
    out["heat10_y5"] = g.filter(sigs, method="chebyshev", order=30)
    out["heat10_exact"] = g.filter(sig, method="exact")
    # two scales (test_frame, test_filters.py:157-168)
    g2 = filters.Heat(G, scale=[8, 9])
    out["heat89_frame"] = g2.compute_frame(method="chebyshev", order=30)
    # filterbank: analysis + synthesis
    mh = filters.MexicanHat(G, Nf=6)
    out["mh6_c"] = np.array(filters.compute_cheby_coeff(mh, m=40))
    a = mh.filter(sigs, method="chebyshev", order=40)          # (N, 5, 6)
    out["mh6_analysis"] = a
    out["mh6_synthesis"] = mh.filter(a, method="chebyshev", order=40)  # (N, 5)
    out["mh6_analysis1"] = mh.filter(sig, method="chebyshev", order=40)  # (N, 6)
    out["mh6_synthesis1"] = mh.filter(out["mh6_analysis1"], method="chebyshev", order=40)
    # low orders (order=1, 2 work; order=0 raises TypeError)
    out["heat10_order1"] = g.filter(sig, method="chebyshev", order=1)
    out["heat10_order2"] = g.filter(sig, method="chebyshev", order=2)
    # same-recurrence siblings (approximations.py:117-225)
    from pygsp.filters import approximations as apx
    b = [0.2 * G.lmax, 0.6 * G.lmax]
    out["rect_bounds"] = np.array(b)
    out["rect_y"] = apx.cheby_rect(G, list(b), sig, order=30)
    out["rect_y5"] = apx.cheby_rect(G, list(b), sigs, order=25)
    ch, jch = apx.compute_jackson_cheby_coeff(list(b), [0, G.lmax], 30)
    out["jackson_ch"], out["jackson_jch"] = ch, jch
    # normalized Laplacian
    G.compute_laplacian("normalized")
    out.update(csr_parts(G.L, "Lnorm"))
    G.estimate_lmax("bounds")
    out["lmax_norm"] = np.float64(G.lmax)
    gn = filters.Heat(G, scale=10)
    out["heat10_norm_y"] = gn.filter(sig, method="chebyshev", order=30)
    np.savez_compressed(os.path.join(OUT, "sensor123.npz"), **out)


def doctest_sensor30():
    """filter.py:232-256: Sensor(30, seed=42), MexicanHat Nf=4, ||s1 - s2|| = 0.27649."""
    G = graphs.Sensor(30, seed=42)
    G.compute_fourier_basis()
    out = csr_parts(G.W, "W")
    out["coords"] = G.coords
    out["lmax"] = np.float64(G.lmax)
    s1 = np.zeros(G.N)
    s1[13] = 1
    s1 = filters.Heat(G, 3).filter(s1)
    g = filters.MexicanHat(G, Nf=4)
    s2 = g.analyze(s1)
    s3 = g.synthesize(s2)
    out["s1"], out["s2"], out["s3"] = s1, s2, s3
    out["norm"] = np.float64(np.linalg.norm(s1 - s3))
    np.savez_compressed(os.path.join(OUT, "doctest_sensor30.npz"), **out)


def laplacians4():
    """pygsp/tests/test_graphs.py:195-254: hand-written 4x4 adjacency, directed and undirected,
    combinatorial and normalized; plus a graph with an isolated vertex and one with self-loops."""
    out = {}
    W_und = np.array([[0, 2, 0, 0], [2, 0, 4, 0], [0, 4, 0, 5], [0, 0, 5, 0]], dtype=float)
    W_dir = np.array([[0, 2, 0, 0], [0, 0, 4, 0], [0, 4, 0, 5], [0, 0, 0, 0]], dtype=float)
    W_iso = np.array([[0, 1, 0, 0], [1, 0, 3, 0], [0, 3, 0, 0], [0, 0, 0, 0]], dtype=float)
    W_loop = np.array([[1, 2, 0, 0], [2, 0, 4, 0], [0, 4, 3, 5], [0, 0, 5, 0]], dtype=float)
    for name, W in (("und", W_und), ("dir", W_dir), ("iso", W_iso), ("loop", W_loop)):
        out["W_" + name] = W
        for lt in ("combinatorial", "normalized"):
            G = graphs.Graph(W, lap_type=lt)
            out["L_{}_{}".format(name, lt)] = G.L.toarray()
            out["dw_" + name] = G.dw
    np.savez_compressed(os.path.join(OUT, "laplacians4.npz"), **out)


def ops_sensor123():
    """SURVEY 8(f) row 3 on the same fixture graph: Dirichlet energy (graph.py:642-702), differential
    operator / grad / div (difference.py), Tikhonov regression and classification (learning.py)."""
    from pygsp import learning
    out = {}
    rng = np.random.default_rng(7)
    G = graphs.Sensor(123, seed=42)
    out.update(csr_parts(G.W, "W"))
    x = rng.standard_normal(G.N)
    X5 = rng.standard_normal((G.N, 5))
    out["x"], out["X5"] = x, X5
    for lt in ("combinatorial", "normalized"):
        G.compute_laplacian(lt)
        G.compute_differential_operator()
        src, dst, w = G.get_edge_list()
        out["edges_src"], out["edges_dst"], out["edges_w"] = src, dst, w
        out["D_" + lt] = G.D.toarray()
        out["energy_" + lt] = np.float64(G.dirichlet_energy(x))
        out["energy5_" + lt] = G.dirichlet_energy(X5)
        out["grad_" + lt] = G.grad(x)
        out["grad5_" + lt] = G.grad(X5)
        out["div_" + lt] = G.div(G.grad(x))
        out["div5_" + lt] = G.div(G.grad(X5))
        out["Lx_" + lt] = G.L.dot(X5)
    # Tikhonov (the reference reads G.L: use the combinatorial Laplacian, as its doctests do)
    G.compute_laplacian("combinatorial")
    mask = rng.uniform(0, 1, G.N) > 0.5
    out["mask"] = mask
    meas = x.copy()
    meas[~mask] = np.nan
    out["measures"] = meas
    m0 = np.nan_to_num(meas)
    for tau in (0.5, 5.0):
        out["reg_tau%g" % tau] = learning.regression_tikhonov(G, m0.copy(), mask, tau=tau)
    M3 = np.nan_to_num(np.where(mask[:, None], X5[:, :3], np.nan))
    out["reg3_in"] = M3
    out["reg3_tau0.5"] = learning.regression_tikhonov(G, M3.copy(), mask, tau=0.5)
    out["reg_tau0"] = learning.regression_tikhonov(G, m0.copy(), mask, tau=0)
    labels = (G.coords[:, 0] > 0.5).astype(int) + (G.coords[:, 1] > 0.5).astype(int)
    out["labels"] = labels
    lab_meas = labels.astype(float)
    lab_meas[~mask] = np.nan
    out["class_tau0.1"] = learning.classification_tikhonov(G, lab_meas.copy(), mask, tau=0.1)
    out["class_tau0"] = learning.classification_tikhonov(G, lab_meas.copy(), mask, tau=0)
    np.savez_compressed(os.path.join(OUT, "ops_sensor123.npz"), **out)


def ops_directed():
    """SURVEY 8(f) row 3, the branches of difference.py:160-166: the differential operator, grad and div of a
    DIRECTED weighted graph (all stored entries are edges, values / sqrt(2)) and of an undirected graph with
    self-loops (diagonal entries are edges whose stored zeros are eliminated); Laplacian and degrees beside them."""
    from scipy import sparse
    out = {}
    rng = np.random.default_rng(11)
    n = 60
    A = sparse.random(n, n, 0.08, random_state=5, format="lil")
    A.setdiag(0)
    A[3, 3], A[17, 17] = 0.8, 1.7  # two self-loops in the directed graph too
    Wd = sparse.csr_matrix(A)
    Wd.eliminate_zeros()
    B = sparse.random(n, n, 0.05, random_state=6, format="csr")
    Wu = sparse.lil_matrix(B + B.T)
    Wu[5, 5], Wu[20, 20], Wu[59, 59] = 1.5, 0.25, 2.0
    Wu = sparse.csr_matrix(Wu)
    x, X4 = rng.standard_normal(n), rng.standard_normal((n, 4))
    out["x"], out["X4"] = x, X4
    for name, W in (("dir", Wd), ("loops", Wu)):
        out.update(csr_parts(W, "W" + name))
        for lt in ("combinatorial", "normalized"):
            G = graphs.Graph(W, lap_type=lt)
            assert G.is_directed() == (name == "dir")
            G.compute_differential_operator()
            src, dst, w = G.get_edge_list()
            key = "{}_{}".format(name, lt)
            out["src_" + name], out["dst_" + name], out["w_" + name] = src, dst, w
            out["ne_" + name] = np.int64(G.n_edges)
            out["dw_" + name] = G.dw
            out["D_" + key] = G.D.toarray()
            out["L_" + key] = G.L.toarray()
            out["grad_" + key], out["grad4_" + key] = G.grad(x), G.grad(X4)
            out["div_" + key], out["div4_" + key] = G.div(G.grad(x)), G.div(G.grad(X4))
    np.savez_compressed(os.path.join(OUT, "ops_directed.npz"), **out)


def lmax_directed():
    """Graph._get_upper_bound (graph.py:933-960) on DIRECTED graphs: the 60-vertex directed fixture of ops_directed()
    and a small dense directed graph on which the FIRST candidate, N * max(W) of the unsymmetrised W (graph.py:941), is
    the minimum of the four - the case where symmetrising first would return a smaller bound than the reference."""
    from scipy import sparse
    out = {}
    A = sparse.random(60, 60, 0.08, random_state=5, format="lil")
    A.setdiag(0)
    A[3, 3], A[17, 17] = 0.8, 1.7
    Wd = sparse.csr_matrix(A)
    Wd.eliminate_zeros()
    rng = np.random.default_rng(3)
    D = rng.uniform(0.9, 1.0, (6, 6))
    np.fill_diagonal(D, 0.0)
    D[0, 1], D[1, 0] = 1.0, 0.2  # the largest entry has a small mirror: max(W) > max((W + W.T) / 2)
    Wdense = sparse.csr_matrix(D)
    for name, W in (("sparse60", Wd), ("dense6", Wdense)):
        G = graphs.Graph(W)
        assert G.is_directed()
        out.update(csr_parts(W, "W_" + name))
        out["bound_" + name] = np.float64(G._get_upper_bound())
        G.estimate_lmax("bounds")
        out["lmax_bounds_" + name] = np.float64(G.lmax)
    G = graphs.Graph(Wdense)
    assert out["bound_dense6"] == G.n_vertices * np.max(G.W)  # the first candidate is the minimum here
    np.savez_compressed(os.path.join(OUT, "lmax_directed.npz"), **out)


def knn():
    """NNGraph (nngraph.py:113-297) on small point clouds, Sensor variants."""
    out = {}
    rng = np.random.default_rng(3)
    X3 = rng.standard_normal((200, 3)) * np.array([1.0, 2.0, 0.5]) + 4.0
    G = graphs.NNGraph(X3, k=5)  # center=True, rescale=True
    out["X3"] = X3
    out.update(csr_parts(G.W, "W3"))
    out["X3_coords"] = G.coords
    out["sigma3"] = np.float64(G.sigma)
    X1 = rng.uniform(0, 10, (64, 1))
    G = graphs.NNGraph(X1, k=3, center=False, rescale=False, sigma=0.7)
    out["X1"] = X1
    out.update(csr_parts(G.W, "W1"))
    G = graphs.Sensor(123, seed=42)
    out.update(csr_parts(G.W, "Wsensor"))
    out["sensor_coords"] = G.coords
    out["sensor_sigma"] = np.float64(G.sigma)
    G = graphs.Sensor(144, k=4, distributed=True, seed=7)
    out.update(csr_parts(G.W, "Wdist"))
    out["dist_coords"] = G.coords
    # radius graphs (nngraph.py:228-287)
    Xr = rng.uniform(0, 1, (300, 3))
    G = graphs.NNGraph(Xr, NNtype="radius", epsilon=0.35)  # centred and rescaled: extents ~ 0.7 here
    out["Xr"] = Xr
    out.update(csr_parts(G.W, "Wr"))
    out["sigma_r"] = np.float64(G.sigma)
    X2 = rng.uniform(0, 1, (400, 2))
    G = graphs.NNGraph(X2, NNtype="radius", epsilon=0.08, center=False, rescale=False, sigma=0.01)
    out["X2r"] = X2
    out.update(csr_parts(G.W, "W2r"))
    # other metrics (dist_type, nngraph.py:139-145)
    G = graphs.NNGraph(X3, k=6, dist_type="manhattan")
    out.update(csr_parts(G.W, "W3_manhattan"))
    G = graphs.NNGraph(X3, k=4, dist_type="max_dist", center=False, rescale=False)
    out.update(csr_parts(G.W, "W3_maxdist"))
    G = graphs.NNGraph(X2, NNtype="radius", epsilon=0.07, dist_type="manhattan", center=False, rescale=False)
    out.update(csr_parts(G.W, "W2r_manhattan"))
    for st in ("maximum", "fill", "tril", "triu"):  # utils.symmetrize, utils.py:247-275
        G = graphs.NNGraph(X3, k=5, symmetrize_type=st)
        out.update(csr_parts(G.W, "W3_" + st))
    np.savez_compressed(os.path.join(OUT, "knn.npz"), **out)


def knn_highdim():
    """NNGraph beyond three dimensions (nngraph.py:113-297 takes any dimension; nngraphs/imgpatches.py builds such
    clouds from image patches): 3 x 3 and 5 x 5 patches of a smooth random image (9 and 25 dimensions; the
    reference's own patch extraction needs scikit-image, which this container lacks, so the patches are cut with
    numpy - NNGraph only sees the feature matrix), a 6-D Gaussian cloud under the manhattan metric, a 40-D one."""
    out = {}
    rng = np.random.default_rng(11)
    img = rng.standard_normal((26, 26))
    for _ in range(3):  # smooth it a little: neighbouring patches resemble each other, like in a real image
        img = (img + np.roll(img, 1, 0) + np.roll(img, -1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 1)) / 5

    def patches(width):
        h = width // 2
        pad = np.pad(img, h, mode="symmetric")
        rows = [pad[i:i + width, j:j + width].ravel() for i in range(img.shape[0]) for j in range(img.shape[1])]
        return np.array(rows)

    for tag, width, k in (("p9", 3, 8), ("p25", 5, 10)):
        X = patches(width)
        G = graphs.NNGraph(X, k=k)
        out["X_" + tag] = X
        out.update(csr_parts(G.W, "W_" + tag))
        out["sigma_" + tag] = np.float64(G.sigma)
        out["coords_" + tag] = G.coords
    X6 = rng.standard_normal((500, 6)) * np.array([1, 2, 0.5, 1, 3, 1.0])
    G = graphs.NNGraph(X6, k=7, dist_type="manhattan")
    out["X6"] = X6
    out.update(csr_parts(G.W, "W6_manhattan"))
    G = graphs.NNGraph(X6, k=5, symmetrize_type="maximum", center=False, rescale=False)
    out.update(csr_parts(G.W, "W6_maximum"))
    X40 = rng.standard_normal((300, 40))
    G = graphs.NNGraph(X40, k=12, sigma=2.5)
    out["X40"] = X40
    out.update(csr_parts(G.W, "W40"))
    # radius graphs beyond three dimensions (nngraph.py:228-287): about eight neighbours per point
    G = graphs.NNGraph(X6, NNtype="radius", epsilon=0.125)
    out.update(csr_parts(G.W, "W6_radius"))
    out["sigma6_radius"] = np.float64(G.sigma)
    G = graphs.NNGraph(out["X_p9"], NNtype="radius", epsilon=0.3, dist_type="manhattan")
    out.update(csr_parts(G.W, "Wp9_radius_manhattan"))
    np.savez_compressed(os.path.join(OUT, "knn_highdim.npz"), **out)


if __name__ == "__main__":
    print("pygsp", pygsp.__version__)
    if len(sys.argv) > 1:  # e.g. `gen_golden.py ops_sensor123`: regenerate only the named fixtures
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    logo()
    sensor123()
    doctest_sensor30()
    laplacians4()
    ops_sensor123()
    ops_directed()
    lmax_directed()
    knn()
    knn_highdim()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
