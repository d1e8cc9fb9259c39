"""Device cost of one five-point RANSAC round (k_fivepoint_hyp + k_fivepoint_score) against the host solver — the measurement behind
DESIGN.md §8 "five-point on the device". usage: python scripts/ubench/fivepoint_device.py"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import orc_binding
pmv = importlib.import_module("practical-multi-view_amd")
orc = orc_binding.load()
rng = np.random.default_rng(1)
n = 400
X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 2, n), rng.uniform(5, 40, n)], 1)
t = np.array([0.08, -0.03, -1.0]); t /= np.linalg.norm(t)
q1 = X[:, :2] / X[:, 2:3] + rng.normal(0, 4e-4, (n, 2))
Xc = X + t
q2 = Xc[:, :2] / Xc[:, 2:3] + rng.normal(0, 4e-4, (n, 2))
ctx = pmv.Context(640, 480, n_slots=1, max_tracks=1024)
thr = np.float32((1 / 718.856) ** 2)
for nh in (1, 8, 17, 32, 64):
    s = np.zeros((nh, 5), np.int32)
    orc.lib.orc_host_five_point_samples(n, nh, s.ctypes.data_as(C.POINTER(C.c_int)))
    ctx.fivepoint_hypotheses(q1, q2, s, thr)
    t0 = time.perf_counter()
    for _ in range(20):
        m, nm, c = ctx.fivepoint_hypotheses(q1, q2, s, thr)
    dev = (time.perf_counter() - t0) / 20
    f64p = C.POINTER(C.c_double)
    Es = np.zeros(90)
    t0 = time.perf_counter()
    for _ in range(20):
        for h in range(nh):
            s1 = np.ascontiguousarray(q1[s[h]]); s2 = np.ascontiguousarray(q2[s[h]])
            orc.lib.orc_host_five_point(s1.ctypes.data_as(f64p), s2.ctypes.data_as(f64p), Es.ctypes.data_as(f64p))
    host = (time.perf_counter() - t0) / 20
    print(f"{nh:3d} hypotheses: device round {dev * 1e6:8.1f} us ({dev / nh * 1e6:7.1f} us per hypothesis), host solver only (one core, incl. ctypes) {host * 1e6:8.1f} us "
          f"({host / nh * 1e6:6.1f} per hypothesis); models per hypothesis {nm.mean():.1f}")
ctx.close()
