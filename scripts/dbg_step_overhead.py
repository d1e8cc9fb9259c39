"""Where does a bench step spend time outside OdometryPipeline::run_threaded? (setup, result extraction, teardown)"""
import sys, time, importlib, ctypes as C, numpy as np
sys.path.insert(0, ".")
pmv = importlib.import_module("practical-multi-view_amd")
w, h, n = 1241, 376, 1101
fx = 718.856; cx, cy = 607.1928, 185.2157
frames, gt = pmv.synth_sequence(1007, 0, n, w, h, fx, fx, cx, cy, nthreads=16)
K = np.array([fx, 0, cx, 0, fx, cy, 0, 0, 1.0])
ctx = pmv.Context(w, h, n_slots=n, max_tracks=4096, max_ba_cams=32, max_ba_points=8192, max_ba_obs=65536)
ctx.frames_stage(0, frames)
for rep in range(3):
    P = pmv.PipelineParams(n, w, h, 400, 150, 5, 5, 5, 0, 1, 8, 1)
    out = C.c_void_p()
    Kd = np.ascontiguousarray(K); g = np.ascontiguousarray(gt, np.float64).reshape(n, 12)
    f64p = C.POINTER(C.c_double)
    t0 = time.perf_counter()
    rc = ctx.lib.pmv_pipeline_run(ctx.h, C.byref(P), Kd.ctypes.data_as(f64p), g.ctypes.data_as(f64p), C.byref(out))
    t1 = time.perf_counter()
    res = pmv.PipelineResult(ctx.lib, out, False)
    t2 = time.perf_counter()
    ctx.lib.pmv_pipeline_free(out)
    t3 = time.perf_counter()
    print("rc", rc, "run call %.4f  (pipeline seconds %.4f)  extract %.4f  free %.4f" % (t1 - t0, res.stats["seconds"], t2 - t1, t3 - t2))
