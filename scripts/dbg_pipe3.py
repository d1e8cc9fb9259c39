"""Per-frame pose/landmark differences HIP pipeline vs oracle on the config3-like case (800 tracks, bundle 10)."""
import sys, importlib, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import orc_binding as ob
pmv = importlib.import_module("practical-multi-view_amd")
cfg = dict(w=1241, h=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)
n = 36
kw = dict(min_tracked=800, tol=300, bundle_size=10)
frames, poses = pmv.synth_sequence(1000, 0, n, cfg["w"], cfg["h"], cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"])
K = np.array([cfg["fx"], 0, cfg["cx"], 0, cfg["fy"], cfg["cy"], 0, 0, 1.0])
ctx = pmv.Context(cfg["w"], cfg["h"], n_slots=n, max_tracks=4096, max_ba_cams=32, max_ba_points=8192, max_ba_obs=65536)
ctx.frames_stage(0, frames)
g = ctx.pipeline_run(n, cfg["w"], cfg["h"], K, poses, **kw)
o = ob.run_pipeline(frames, K, poses, n_threads=8, **kw)
for k in range(len(g.features)):
    a, b = g.features[k], o.features[k]
    pd = np.abs(g.poses[k] - o.poses[k]).max() if k < len(g.poses) and k < len(o.poses) else -1
    print(k, "xy", np.array_equal(a[:, :2], b[:, :2]), "lm", np.array_equal(a[:, 2], b[:, 2]), "n3d", (a[:, 2] >= 0).sum(), (b[:, 2] >= 0).sum(), "posediff %.3e" % pd)
