import importlib, sys, numpy as np
sys.path[:0]=["/root/repo","/root/repo/tests"]
pkg = importlib.import_module("bwd-nlkalman_amd"); synth = importlib.import_module("bwd-nlkalman_amd.synth")
w,h,ch,sigma=1920,1080,3,20.0
n0,n1,_=synth.noisy_pair(w,h,ch,sigma,1)
o0,o1=pkg.rgb2opp(n0),pkg.rgb2opp(n1)
p=pkg.default_params(sigma,pkg.FLT1); p2=pkg.default_params(sigma,pkg.FLT2); ps=pkg.default_params(sigma,pkg.SMO1)
def frame(c,smo,cur,prev,basic,pp):
    d=[c.upload(a) if a is not None else None for a in (cur,prev,basic)]
    o=c.alloc(cur.nbytes); (c.smooth_frame if smo else c.filter_frame)(o,d[0],d[1],d[2],w,h,ch,sigma,pp)
    out=c.download(o,cur.shape)
    for x in d+[o]:
        if x: c.free(x)
    return out
for det in (False, True):
    c=pkg.Context(0); c.set_deterministic(det)
    print("deterministic",det,flush=True)
    f0=frame(c,False,o0,None,None,p); print(" spatial ok",flush=True)
    f1=frame(c,False,o1,f0,None,p); print(" temporal ok",flush=True)
    hole=f0.copy(); hole[500:540,900:1000]=np.nan; hole[:1]=np.nan; hole[-2:]=np.nan; hole[:,:1]=np.nan; hole[:,-2:]=np.nan
    f1h=frame(c,False,o1,hole,None,p); print(" temporal+holes ok",flush=True)
    f2=frame(c,False,o1,hole,f1h,p2); print(" flt2 ok",flush=True)
    s0=frame(c,True,f0,f2,None,ps); print(" smo ok",flush=True)
    c.close()
