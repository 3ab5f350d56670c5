"""The drop-in seam against the REAL reference: the reference's own
`pygsp/tests/test_filters.py` is run, unmodified, in a subprocess with `pygsp_amd.plugin.install()`
applied to the real `pygsp` package; every `Filter.filter(method='chebyshev')` in it then goes through
`pygsp_amd.filters.cheby_op` (filter.py:305-322 -> approximations.cheby_op).  There is no GPU in the
build container, so the device object is replaced by an oracle-backed stand-in (tests/seam_plugin.py,
test infrastructure only); on the GPU box `/root/reference` does not exist and the test is skipped -
there the same seam is exercised on hardware by tests/test_gpu_2_kernels.py::test_plugin_patches_a_pygsp_like_module."""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pygsp", "tests")),
                                reason="needs the reference checkout at /root/reference")


def _run_reference_tests(tmp_path, test_file, wrap_filter=True):
    report = tmp_path / "seam.json"
    env = dict(os.environ)
    env["GSPX_SEAM_WRAP_FILTER"] = "1" if wrap_filter else "0"
    env["PYTHONPATH"] = os.pathsep.join([REF, ROOT, os.path.join(ROOT, "tests")])
    env["PYTHONDONTWRITEBYTECODE"] = "1"  # nothing is written under /root/reference
    env["GSPX_SEAM_REPORT"] = str(report)
    env["MPLBACKEND"] = "Agg"
    cmd = [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-p", "seam_plugin", "-q", "--no-header",
           "-o", "addopts=", "--rootdir", str(tmp_path), os.path.join(REF, "pygsp", "tests", test_file)]
    res = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1500)
    calls = json.loads(report.read_text()) if report.exists() else {}
    return res, calls


def test_reference_test_filters_passes_through_the_patched_seam(tmp_path):
    res, calls = _run_reference_tests(tmp_path, "test_filters.py")
    tail = res.stdout[-3000:] + res.stderr[-2000:]
    assert res.returncode == 0, tail
    assert " passed" in res.stdout and "failed" not in res.stdout, tail
    # the reference's file holds 25 tests; all of them must have run
    n_passed = int(res.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n_passed >= 25, tail
    # ... and the Chebyshev calls really went through the product's cheby_op
    assert calls.get("cheby_op", 0) >= 50 and calls.get("graphs", 0) >= 1, calls
    # the secondary seam (filter.py:313-322): every Chebyshev synthesis was ONE device call, not Nf cheby_op calls
    assert calls["synthesis_filters"] >= 5 and calls["synthesis_device_calls"] == calls["synthesis_filters"], calls
    assert calls["frames"] >= 1, calls


def test_reference_test_filters_passes_with_the_primary_seam_alone(tmp_path):
    """install(wrap_filter=False): cheby_op is the only patched name; the reference's own synthesis loop
    (filter.py:318-321) then calls it once per filter and no fused synthesis call is made."""
    res, calls = _run_reference_tests(tmp_path, "test_filters.py", wrap_filter=False)
    tail = res.stdout[-3000:] + res.stderr[-2000:]
    assert res.returncode == 0 and " passed" in res.stdout and "failed" not in res.stdout, tail
    assert calls.get("cheby_op", 0) >= 50 and calls["synthesis_device_calls"] == 0 and calls["synthesis_filters"] >= 5, calls


def test_reference_doctest_value_through_the_seam(tmp_path):
    """filter.py:255-256 (the doctest of Filter.filter): Sensor(30, seed=42), MexicanHat x 3 signals,
    analysis then synthesis, 0.27649 - through the patched real pygsp."""
    script = tmp_path / "doctest_case.py"
    script.write_text(
        "import numpy as np\n"
        "from pygsp import graphs, filters\n"
        "def test_case():\n"
        "    G = graphs.Sensor(30, seed=42)\n"
        "    G.compute_fourier_basis()  # reproducible lmax, as the doctest does\n"
        "    s1 = np.zeros(G.N); s1[13] = 1\n"
        "    s1 = filters.Heat(G, 3).filter(s1)\n"
        "    g = filters.MexicanHat(G, Nf=4)\n"
        "    s2 = g.analyze(s1)\n"
        "    assert s2.shape == (30, 4)\n"
        "    s3 = g.synthesize(s2)\n"
        "    assert s3.shape == (30,)\n"
        "    assert '{:.5f}'.format(np.linalg.norm(s1 - s3)) == '0.27649'\n")
    res, calls = _run_reference_tests(tmp_path, str(script))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert calls.get("cheby_op", 0) >= 3, calls


def test_estimate_lmax_seam_on_the_real_graph_class():
    """plugin.install(pygsp, lmax='device') replaces the REAL ``pygsp.graphs.Graph.estimate_lmax`` (graph.py:858) and
    uninstall() / a default install() put the reference's method back; with the seam in place the 'bounds' method is
    still the reference's own code (no device here, so the 'lanczos' branch itself runs in `-m gpu`,
    test_plugin_estimate_lmax_seam)."""
    code = (
        "import pygsp\n"
        "from pygsp_amd import plugin\n"
        "orig = pygsp.graphs.Graph.estimate_lmax\n"
        "plugin.install(pygsp, lmax='device')\n"
        "assert pygsp.graphs.Graph.estimate_lmax is not orig\n"
        "G = pygsp.graphs.Logo()\n"
        "G.estimate_lmax('bounds')\n"
        "assert abs(G.lmax - 18.583333333333332) < 1e-12 and G._lmax_method == 'bounds'\n"
        "plugin.install(pygsp)\n"
        "assert pygsp.graphs.Graph.estimate_lmax is orig\n"
        "plugin.install(pygsp, lmax='device'); plugin.uninstall(pygsp)\n"
        "assert pygsp.graphs.Graph.estimate_lmax is orig\n"
        "print('seam ok')\n")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([REF, ROOT])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    env["MPLBACKEND"] = "Agg"
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "seam ok" in res.stdout, res.stdout[-1500:] + res.stderr[-1500:]


def test_reference_doctests_give_the_same_results_through_the_seam(tmp_path):
    """The reference's own doctests of the modules that reach the Chebyshev path (pygsp/filters/*.py, reduction.py,
    features.py: ~350 examples, collected the way pygsp/tests/test_docstrings.py does) printed by the real package with
    and without `plugin.install()`: the same examples pass and the same few fail either way (those need packages this
    container lacks), and with the seam > 100 of the calls went through the product's cheby_op / Filter.filter."""
    def run(seam):
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([REF, ROOT, os.path.join(ROOT, "tests")])
        env["PYTHONDONTWRITEBYTECODE"] = "1"
        env["MPLBACKEND"] = "Agg"
        env["GSPX_SEAM"] = "1" if seam else "0"
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "seam_doctests.py")], cwd=str(tmp_path), env=env,
                             capture_output=True, text=True, timeout=1500)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
        return json.loads(res.stdout.strip().splitlines()[-1])

    plain, seamed = run(False), run(True)
    attempted = sum(v["attempted"] for v in plain["files"].values())
    assert attempted >= 300 and plain["files"]["filters/filter.py"]["attempted"] >= 100
    for rel, v in plain["files"].items():
        w = seamed["files"][rel]
        assert (w["attempted"], w["failed"], w["failing"]) == (v["attempted"], v["failed"], v["failing"]), rel
    assert sum(v["failed"] for v in plain["files"].values()) <= attempted // 20  # (missing optional packages only)
    calls = seamed["calls"]
    assert calls["cheby_op"] >= 100 and calls["graphs"] >= 10 and calls["frames"] >= 1, calls
    assert calls["synthesis_device_calls"] == calls["synthesis_filters"] >= 1, calls
