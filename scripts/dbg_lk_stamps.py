"""Phase timers (shader clock) of k_lk for track 0 over a short sequence (PMV_LK_STAMPS=1)."""
import sys, os, importlib, ctypes as C, numpy as np
sys.path.insert(0, ".")
os.environ["PMV_LK_STAMPS"] = "1"
pmv = importlib.import_module("practical-multi-view_amd")
w, h, n = 1241, 376, 12
fx = 718.856; cx, cy = 607.1928, 185.2157
frames, gt = pmv.synth_sequence(1007, 0, n, w, h, fx, fx, cx, cy, nthreads=8)
ctx = pmv.Context(w, h, n_slots=n)
ctx.frames_stage(0, frames); ctx.frames_build(0, n)
cells = pmv.grid_cells(w, h)
pts = np.concatenate([c for c in ctx.detect_gftt(0, cells, 40) if len(c)]).astype(np.float32)
calls = 0
for k in range(n - 1):
    nxt, st, err = ctx.lk_track(k, k + 1, pts)
    calls += 1
    pts = np.floor(nxt[st > 0]).astype(np.float32)
out = np.zeros(16, np.uint64)
ctx.lib.pmv_debug_lk_stamps(ctx.h, out.ctypes.data_as(C.POINTER(C.c_uint64)))
names = ["level-entry", "I-tile", "scharr", "samples+A", "iterations(+J tiles)", "err-pass"]
print("tracks", len(pts), "calls", calls, " ".join("%s=%d" % (nm, int(out[i]) / calls) for i, nm in enumerate(names)), "iterations/call %.1f" % (int(out[8]) / calls), "cycles per call")
it = max(1, int(out[8]))
print("per iteration: top(tile check)=%d sample+diff=%d block-sum=%d update=%d cycles" % tuple(int(out[k]) / it for k in (9, 10, 11, 12)))
