"""50 x pmv_pnp_ransac on one synthetic problem (300 points, 20 % outliers): a small target for rocprofv3 --pmc on k_pnp_hyp."""
import sys, os, importlib, time, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import scenes
pmv = importlib.import_module("practical-multi-view_amd")
ctx = pmv.Context(64, 64, n_slots=1)
P = scenes.pnp_problem(1, m=300, outlier_frac=0.2)
N = int(os.environ.get("N", "50"))
t0 = time.perf_counter()
for _ in range(N):
    ctx.pnp_ransac(P["obj"], P["img"], scenes.K, np.array([0.3, -0.2, 0.1]), np.array([1.0, 2.0, -30.0]))
print("%.1f us per call" % ((time.perf_counter() - t0) / N * 1e6))
