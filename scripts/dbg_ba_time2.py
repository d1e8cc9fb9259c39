import sys, os, importlib, time, ctypes as C, numpy as np
sys.path.insert(0,"."); sys.path.insert(0,"tests")
os.environ["PMV_BA_STAMPS"]="1"
import scenes
pmv = importlib.import_module("practical-multi-view_amd")
ctx = pmv.Context(64,64,n_slots=1)
P = scenes.ba_problem(4, nc=5, npts=400)
names=["eval","camblocks","diag","pointblocks","gmax","gemm","chol","pt-backsub","modelchange","cand"]
prev=np.zeros(32,np.uint64)
for iters in (1,2,5,10,20):
    _,_,s=ctx.ba_solve(P["cams"],P["pts"],P["obs"],P["cam_idx"],P["pt_idx"],scenes.K,1.0,iters)
    st=np.zeros(32,np.uint64)
    ctx.lib.pmv_debug_ba_stamps(ctx.h, st.ctypes.data_as(C.POINTER(C.c_uint64)))
    d=[int(st[i])-int(prev[i]) for i in range(10)]
    prev=st.copy()
    print("max_iter",iters,"ran",s.iterations,"succ",s.successful_steps,"total cycles",int(st[29]), " ".join("%s=%d"%(n,v) for n,v in zip(names,d)))
