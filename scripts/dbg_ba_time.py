import sys, os, importlib, time, ctypes as C, numpy as np
sys.path.insert(0,"."); sys.path.insert(0,"tests")
os.environ["PMV_BA_STAMPS"]="1"
import scenes
pmv = importlib.import_module("practical-multi-view_amd")
ctx = pmv.Context(64,64,n_slots=1)
P = scenes.ba_problem(4, nc=5, npts=400)
print("nobs", len(P["obs"]), "np", len(P["pts"]))
for rep in range(3):
    t=time.perf_counter()
    for _ in range(20): ctx.ba_solve(P["cams"],P["pts"],P["obs"],P["cam_idx"],P["pt_idx"],scenes.K)
    print("host ms per call", (time.perf_counter()-t)/20*1e3)
st=np.zeros(32,np.uint64)
ctx.lib.pmv_debug_ba_stamps(ctx.h, st.ctypes.data_as(C.POINTER(C.c_uint64)))
calls=60
names=["eval","camblocks","diag","pointblocks","gmax","gemm","chol","pt-backsub","modelchange","cand"]
tot=sum(int(st[i]) for i in range(10))
for i,nm in enumerate(names): print("%-12s %10.0f cycles/call  %5.1f%%"%(nm, int(st[i])/calls, 100.0*int(st[i])/tot))
print("last call: wall ticks(100MHz)", int(st[28]), "shader cycles", int(st[29]), "=> clock MHz", int(st[29])/ (int(st[28])/100.0))
