"""pytest plugin (test infrastructure, loaded with `-p seam_device_plugin`): the REAL pygsp's own test files on the
REAL device - `pygsp_amd.plugin.install(pygsp)` with nothing replaced.  Used by tests/test_gpu_b_real_pygsp.py,
which arms itself only on a box that has both a GPU and an importable pygsp (the reference checkout or an
installed package); tests/seam_plugin.py is the CPU form of the same seam with an oracle-backed device object.

The number of calls that went through the seam is written to $GSPX_SEAM_REPORT at session end."""
import json
import os

CALLS = {"cheby_op": 0, "filter": 0, "frames": 0, "device_graphs": 0}


def pytest_configure(config):
    import pygsp
    from pygsp_amd import filters as product_filters, plugin

    plugin.install(pygsp, wrap_filter=os.environ.get("GSPX_SEAM_WRAP_FILTER", "1") != "0",
                   lmax=os.environ.get("GSPX_SEAM_LMAX", "reference"))
    assert pygsp.filters.approximations.cheby_op is product_filters.cheby_op

    real_cheby_op, real_for = product_filters.cheby_op, plugin.device_graph_for

    def counting_cheby_op(*args, **kwargs):
        CALLS["cheby_op"] += 1
        return real_cheby_op(*args, **kwargs)

    def counting_for(G, *args, **kwargs):
        had = bool(G.__dict__.get("_gspx_dev"))
        dev = real_for(G, *args, **kwargs)
        CALLS["device_graphs"] += int(not had)
        return dev

    # (the names the reference resolves at call time, and the one the product's own code goes through)
    pygsp.filters.approximations.cheby_op = pygsp.filters.cheby_op = counting_cheby_op
    plugin.device_graph_for = counting_for
    patched_filter, patched_frame = pygsp.filters.Filter.filter, pygsp.filters.Filter.compute_frame

    def counting_filter(self, s, method="chebyshev", order=30):
        CALLS["filter"] += int(method == "chebyshev")
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
