"""How close are the small-graph whole-network tests (tests/test_gpu_network.py) to their tolerances?  Runs them through pytest with
rel_close replaced by a recording version (same scale rule: max(|ref|max, 1e-3)) and prints the worst ratio per kind."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
import tests.test_gpu_network as T

rec = []


def rel_close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = max(float(ref.abs().max()), 1e-3)
    rec.append((float((got - ref).abs().max()) / scale, tol, what, scale))


T.rel_close = rel_close
rc = pytest.main([os.path.join(os.path.dirname(T.__file__), "test_gpu_network.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider"])
g = sorted([r for r in rec if r[2].startswith("grad")], reverse=True)[:6]
o = sorted([r for r in rec if not r[2].startswith("grad")], reverse=True)[:6]
print("rc", rc, "records", len(rec))
print("worst gradients (err / scale, tolerance, tensor, scale):")
for r in g:
    print("   %.2e  %.0e  %-50s %.2e" % r)
print("worst outputs:")
for r in o:
    print("   %.2e  %.0e  %-50s %.2e" % r)
