"""pytest plugin (test infrastructure, loaded with `-p seam_plugin`) that runs the REAL pygsp's own
test files through the product's seam on a box without a GPU:

  * `pygsp_amd.plugin.install(pygsp)` patches `pygsp.filters.approximations.cheby_op`
    (looked up at call time from pygsp/filters/filter.py:309, 319) and the alias
    `pygsp.filters.cheby_op` (pygsp/filters/__init__.py:115) exactly as on a GPU box;
  * the only thing replaced is the device object the product's `cheby_op` hands its panel to:
    `plugin.device_graph_for` returns an oracle-backed stand-in (oracle/cheby_oracle.py), so what is
    exercised is the product's host side of the seam - the `c` contract (1-D, 2-D, list of arrays,
    approximations.py:77-84), dtype / order normalisation, the [filter][vertex][signal] output
    buffer that filter.py:310-311 reshapes, error types.

The number of calls that went through the seam is written to $GSPX_SEAM_REPORT at session end.
"""
import json
import os

import numpy as np

CALLS = {"cheby_op": 0, "graphs": 0, "synthesis_device_calls": 0, "synthesis_filters": 0, "frames": 0}


class OracleDevice:
    """Stand-in for engine.DeviceGraph: same call contract, arithmetic by the oracle."""

    def __init__(self, G):
        from scipy import sparse
        self.L = sparse.csr_matrix(G.L)
        self.N = G.N
        self.dtype = np.dtype(np.float64)
        CALLS["graphs"] += 1

    def cheby_filter(self, coeffs, x, lmax, mode=0):
        from oracle import cheby_oracle as orc
        c = np.atleast_2d(np.asarray(coeffs, dtype=np.float64))
        x = np.ascontiguousarray(x, dtype=np.float64)
        CALLS["cheby_op"] += 1
        CALLS["synthesis_device_calls"] += int(mode == 1)
        if mode == 0:
            return orc.cheby_op(self.L, lmax, c, x).reshape(c.shape[0], self.N, -1), 0.0
        return sum(orc.cheby_op(self.L, lmax, c[f], x[f]) for f in range(c.shape[0])), 0.0


def pytest_configure(config):
    import pygsp
    from pygsp_amd import filters as product_filters, plugin

    def device_graph_for(G, ctx=None, dtype=None):
        cached = getattr(G, "_gspx_dev", None)
        if cached is not None and cached[0] is G.L:
            return cached[1]
        dev = OracleDevice(G)
        G._gspx_dev = (G.L, dev)
        return dev

    plugin.device_graph_for = device_graph_for
    wrap = os.environ.get("GSPX_SEAM_WRAP_FILTER", "1") != "0"
    plugin.install(pygsp, wrap_filter=wrap)
    assert pygsp.filters.approximations.cheby_op is pygsp.filters.cheby_op
    assert pygsp.filters.cheby_op.__module__ == "pygsp_amd.filters"
    assert (pygsp.filters.Filter.filter.__module__ == "pygsp_amd.plugin") == wrap
    assert (pygsp.filters.Filter.compute_frame.__module__ == "pygsp_amd.plugin") == wrap

    # count the Chebyshev synthesis calls and frames that go through Filter.filter / compute_frame
    patched_filter, patched_frame = pygsp.filters.Filter.filter, pygsp.filters.Filter.compute_frame

    def counting_filter(self, s, method="chebyshev", order=30):
        if method == "chebyshev":
            try:
                CALLS["synthesis_filters"] += int(product_filters._cube_shape(self.G, self.Nf, np.shape(s))[2] != 1)
            except ValueError:
                pass
        return patched_filter(self, s, method=method, order=order)

    def counting_frame(self, **kwargs):
        CALLS["frames"] += int(kwargs.get("method", "chebyshev") == "chebyshev")
        return patched_frame(self, **kwargs)

    pygsp.filters.Filter.filter, pygsp.filters.Filter.compute_frame = counting_filter, counting_frame


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("GSPX_SEAM_REPORT")
    if path:
        with open(path, "w") as f:
            json.dump(CALLS, f)
