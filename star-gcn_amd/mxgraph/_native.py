"""Single import point of the native-backed modules for the `mxgraph` mirror, so the mirror works both as
`star_gcn_amd.mxgraph` and as a top-level `mxgraph` package (PYTHONPATH=star-gcn_amd, drop-in for the
reference's `import mxgraph.layers`)."""
try:
    from .. import _lib, contrib, dist, functional, ops, plan  # noqa: F401
except (ImportError, ValueError):  # imported as top-level `mxgraph`
    import os
    import sys
    import types

    _pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "star_gcn_amd" not in sys.modules:
        _pkg = types.ModuleType("star_gcn_amd")
        _pkg.__path__ = [_pkg_dir]
        sys.modules["star_gcn_amd"] = _pkg
    from star_gcn_amd import _lib, contrib, dist, functional, ops, plan  # noqa: F401
