import sys, os, importlib, struct, numpy as np, ctypes as C
sys.path.insert(0,"."); sys.path.insert(0,"tests")
import orc_binding as ob
pmv = importlib.import_module("practical-multi-view_amd")
cfg=dict(w=1226,h=370,fx=707.0912,fy=707.0912,cx=601.8873,cy=183.1104)
n=30
frames,poses=pmv.synth_sequence(1007,0,n,cfg["w"],cfg["h"],cfg["fx"],cfg["fy"],cfg["cx"],cfg["cy"])
K=np.array([cfg["fx"],0,cfg["cx"],0,cfg["fy"],cfg["cy"],0,0,1.0])
dump="/tmp/pnp_dump.bin"
if os.path.exists(dump): os.remove(dump)
os.environ["PMV_DUMP_PNP"]=dump
ctx=pmv.Context(cfg["w"],cfg["h"],n_slots=n)
ctx.frames_stage(0,frames)
g=ctx.pipeline_run(n,cfg["w"],cfg["h"],K,poses,min_tracked=200,tol=75,bundle_size=3)
del os.environ["PMV_DUMP_PNP"]
data=open(dump,"rb").read(); off=0; calls=[]
while off<len(data):
    m=struct.unpack_from("i",data,off)[0]; off+=4
    obj=np.frombuffer(data,np.float32,3*m,off).reshape(m,3).copy(); off+=12*m
    img=np.frombuffer(data,np.float32,2*m,off).reshape(m,2).copy(); off+=8*m
    Kd=np.frombuffer(data,np.float64,9,off).copy(); off+=72
    rv=np.frombuffer(data,np.float64,3,off).copy(); off+=24
    tv=np.frombuffer(data,np.float64,3,off).copy(); off+=24
    calls.append((obj,img,Kd,rv,tv))
print("pnp calls",len(calls))
lib=ob.load().lib
fp=C.POINTER(C.c_float); dp=C.POINTER(C.c_double); ip=C.POINTER(C.c_int)
for ci,(obj,img,Kd,rv,tv) in enumerate(calls):
    m=len(obj)
    grv,gtv,ginl=ctx.pnp_ransac(obj,img,Kd,rv,tv)
    gm,gc=ctx.pnp_hypotheses(100)
    orv,otv,oinl,hyp=ob.pnp_ransac(obj,img,Kd,rv,tv)
    om=np.zeros((100,6)); oc=np.zeros(100,np.int32)
    lib.orc_pnp_hypotheses(obj.ctypes.data_as(fp),img.ctypes.data_as(fp),m,Kd.ctypes.data_as(dp),100,C.c_float(8.0),om.ctypes.data_as(dp),oc.ctypes.data_as(ip))
    dm=np.abs(gm-om).max(1)
    bad=np.where((dm>1e-7)|(gc!=oc))[0]
    same=np.array_equal(ginl,oinl)
    print(ci,"m",m,"inl g/o",len(ginl),len(oinl),"same",same,"hyp used",hyp,"bad hyps",bad[:10], "maxdiff",dm.max(), "pose diff", np.abs(grv-orv).max(), np.abs(gtv-otv).max())
    if len(bad):
        for b in bad[:3]: print("   hyp",b,"g",gm[b],gc[b],"o",om[b],oc[b])
