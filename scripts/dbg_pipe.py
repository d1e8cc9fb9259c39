import sys, importlib, numpy as np
sys.path.insert(0,"."); sys.path.insert(0,"tests")
import orc_binding as ob
pmv = importlib.import_module("practical-multi-view_amd")
cfg=dict(w=1226,h=370,fx=707.0912,fy=707.0912,cx=601.8873,cy=183.1104)
n=int(sys.argv[1]) if len(sys.argv)>1 else 30
kw=dict(min_tracked=200,tol=75,bundle_size=3)
frames,poses=pmv.synth_sequence(1007,0,n,cfg["w"],cfg["h"],cfg["fx"],cfg["fy"],cfg["cx"],cfg["cy"])
K=np.array([cfg["fx"],0,cfg["cx"],0,cfg["fy"],cfg["cy"],0,0,1.0])
ctx=pmv.Context(cfg["w"],cfg["h"],n_slots=n)
ctx.frames_stage(0,frames)
g=ctx.pipeline_run(n,cfg["w"],cfg["h"],K,poses,**kw)
o=ob.run_pipeline(frames,K,poses,n_threads=8,**kw)
print(g.stats); print(o.stats)
for k in range(len(g.features)):
    a,b=g.features[k],o.features[k]
    same_xy=np.array_equal(a[:,:2],b[:,:2]); same_l=np.array_equal(a[:,2],b[:,2])
    pd = np.abs(g.poses[k]-o.poses[k]).max() if k<len(g.poses) and k<len(o.poses) else -1
    print(k, "xy",same_xy,"lm",same_l, "n3d",(a[:,2]>=0).sum(),(b[:,2]>=0).sum(), "posediff %.3e"%pd)
