"""Import shim: makes the package directory `star-gcn_amd/` (a name Python cannot import directly because
of the hyphen) available as `import star_gcn_amd`.  `star-gcn_amd/` itself can also be put on PYTHONPATH, in
which case `import mxgraph.layers` resolves to the drop-in operator API."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "star-gcn_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
del _f
