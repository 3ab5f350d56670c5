"""Run the REAL pygsp's doctests (the reference's own examples: pygsp/tests/test_docstrings.py walks the package with
doctest.testfile and this namespace) for the modules that reach the Chebyshev path, with or without the product's seam
installed, and print one JSON line: per file (failed, attempted) plus the sources of the failing examples.
Test infrastructure (used by tests/test_seam_real_pygsp.py); with GSPX_SEAM=1 the device object is the oracle-backed
stand-in of tests/seam_plugin.py - there is no GPU in the build container."""
import doctest
import json
import os

import numpy

import pygsp

FILES = ["filters/filter.py", "filters/approximations.py", "filters/heat.py", "filters/mexicanhat.py", "filters/abspline.py",
         "filters/meyer.py", "filters/itersine.py", "filters/expwin.py", "filters/rectangular.py", "filters/halfcosine.py",
         "filters/simoncelli.py", "filters/papadakis.py", "filters/regular.py", "filters/held.py", "filters/simpletight.py",
         "filters/gabor.py", "filters/modulation.py", "filters/wave.py", "filters/chebyshev.py", "features.py", "reduction.py"]


def main():
    calls = None
    if os.environ.get("GSPX_SEAM") == "1":
        import seam_plugin
        seam_plugin.pytest_configure(None)
        calls = seam_plugin.CALLS
    root = os.path.dirname(pygsp.__file__)
    globs = {"graphs": pygsp.graphs, "filters": pygsp.filters, "utils": pygsp.utils, "np": numpy}
    flags = doctest.ELLIPSIS | doctest.NORMALIZE_WHITESPACE | doctest.IGNORE_EXCEPTION_DETAIL
    out = {}
    for rel in FILES:
        path = os.path.join(root, rel)
        if not os.path.exists(path):
            continue
        failing = []

        class Runner(doctest.DocTestRunner):
            def report_failure(self, o, test, example, got):
                failing.append(example.source.strip())

            def report_unexpected_exception(self, o, test, example, exc_info):
                failing.append(example.source.strip())

        parser = doctest.DocTestParser()
        text = open(path).read()
        test = parser.get_doctest(text, dict(globs), rel, path, 0)
        runner = Runner(verbose=False, optionflags=flags)
        with open(os.devnull, "w") as null:
            res = runner.run(test, out=null.write)
        out[rel] = {"failed": res.failed, "attempted": res.attempted, "failing": failing}
        pygsp.plotting.close_all()
    print(json.dumps({"files": out, "calls": calls}))


if __name__ == "__main__":
    main()
