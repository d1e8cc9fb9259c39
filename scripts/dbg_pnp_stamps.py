"""Phase timers (shader clock, hypothesis 0) of k_pnp_hyp."""
import sys, os, importlib, ctypes as C, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["PMV_BA_STAMPS"] = "1"
import scenes
pmv = importlib.import_module("practical-multi-view_amd")
ctx = pmv.Context(64, 64, n_slots=1)
names = ["setup(lane0)", "MtM+jacobi", "sort+L+rho", "pinv-betas", "gauss-newton", "R_and_t", "final"]
P = scenes.pnp_problem(1, m=300, outlier_frac=0.2)
st0 = np.zeros(32, np.uint64)
ctx.lib.pmv_debug_ba_stamps(ctx.h, st0.ctypes.data_as(C.POINTER(C.c_uint64)))
N = 5
for _ in range(N):
    ctx.pnp_ransac(P["obj"], P["img"], scenes.K, np.array([0.3, -0.2, 0.1]), np.array([1.0, 2.0, -30.0]))
st = np.zeros(32, np.uint64)
ctx.lib.pmv_debug_ba_stamps(ctx.h, st.ctypes.data_as(C.POINTER(C.c_uint64)))
d = [(int(st[22 + i]) - int(st0[22 + i])) / N for i in range(7)]
print(" ".join("%s=%d" % (n, v) for n, v in zip(names, d)), "total", sum(d))
print("refit: pre-LM cycles %d, LM cycles %d, LM passes %.1f per call" % ((int(st[30]) - int(st0[30])) / N, (int(st[31]) - int(st0[31])) / N, (int(st[21]) - int(st0[21])) / N))
print("refit per pass: points %d, block_sum %d, decide+solve %d cycles" % tuple((int(st[k]) - int(st0[k])) / max(1, (int(st[21]) - int(st0[21]))) for k in (10, 11, 12)))
