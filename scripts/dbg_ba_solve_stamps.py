"""Phase timers (shader clock) of k_bam_solve: prologue / assembly / Cholesky / back-substitution / candidate rotations."""
import sys, os, importlib, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["PMV_BA_STAMPS"] = "1"
import scenes
pmv = importlib.import_module("practical-multi-view_amd")
ctx = pmv.Context(64, 64, n_slots=1)
names = ["prologue", "assemble", "cholesky", "backsub", "candrot"]
for nc, npts in ((5, 400), (10, 1000), (20, 1500)):
    P = scenes.ba_problem(4, nc=nc, npts=npts)
    st0 = np.zeros(32, np.uint64)
    ctx.lib.pmv_debug_ba_stamps(ctx.h, st0.ctypes.data_as(C.POINTER(C.c_uint64)))
    _, _, s = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, 5)
    st = np.zeros(32, np.uint64)
    ctx.lib.pmv_debug_ba_stamps(ctx.h, st.ctypes.data_as(C.POINTER(C.c_uint64)))
    d = [(int(st[16 + i]) - int(st0[16 + i])) / max(1, s.iterations) for i in range(5)]
    db = [(int(st[10 + i]) - int(st0[10 + i])) / max(1, s.iterations) for i in range(4)]
    print("   backsub kernel (block 0): phaseA=%d phaseB=%d phaseC=%d reduce=%d" % tuple(db))
    dp = [(int(st[4 + i]) - int(st0[4 + i])) / max(1, s.iterations) for i in range(4)]
    print("   point role (block 0): phase1=%d wait=%d rescale(first)=%d phase2=%d" % tuple(dp))
    print("nc", nc, "iterations", s.iterations, " ".join("%s=%d" % (n, v) for n, v in zip(names, d)), "cycles/iteration (100 MHz counter?)")
