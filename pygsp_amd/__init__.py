"""pygsp_amd - MI355X-native Chebyshev graph filtering behind the PyGSP API.

Scope: ``pygsp.filters.Filter.filter(method='chebyshev')`` and the Laplacian it consumes.
``pygsp_amd.graphs`` / ``pygsp_amd.filters`` mirror the reference's classes for that path;
``pygsp_amd.plugin.install()`` patches a real pygsp installation in place.
"""
from . import _capi, engine, filters, graphs, plugin  # noqa: F401

__version__ = "0.1.0"
