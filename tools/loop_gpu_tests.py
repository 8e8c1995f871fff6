"""The whole `-m gpu` suite N times in ONE process (round 4: after the freeze fix -- DESIGN section 5): pass times must not grow and
the process must keep its CPU mask.   python tools/loop_gpu_tests.py"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import pytest
for k in range(3):
    t = time.time()
    rc = pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", "tests", "-x"])
    print("PASS %d rc=%s %.1fs affinity=%d" % (k, rc, time.time() - t, len(os.sched_getaffinity(0))), flush=True)
    if rc != 0: break
