"""Where a frame of the bench spends its wall time on the host: thread hand-over to the two ORB extractor threads and
the matching thread, enqueue of the network, arrival of the class map (tools/; GPU box only).  DROP=<signature names>
removes entries from the ctypes table, FREEZE=1 freezes the garbage collector first."""
import os, sys, time, threading, numpy as np, torch
sys.path.insert(0, os.getcwd())
from bench import make_inputs
from sivo_amd import _lib as _L
for k in os.environ.get("DROP", "").split(","):
    if k: _L.SIGNATURES.pop(k)
from sivo_amd import netspec, orb, weights as wts
from sivo_amd.segnet import BayesianSegNet
T,H,W=12,352,1024
text=netspec.standard_prototxt(T,H,W); layers=netspec.parse_layers(text); w=wts.synth_weights(layers,42)
sn=BayesianSegNet(prototxt=text, weights=wts.pack(layers,w), T=T)
bgr,left,right=make_inputs(H,W)
d_bgr=torch.from_numpy(bgr).cuda(); d_left=torch.from_numpy(left).cuda(); d_right=torch.from_numpy(right).cuda()
maps=(torch.empty((H,W),dtype=torch.uint8,device="cuda"),torch.empty((H,W),dtype=torch.float64,device="cuda"),torch.empty((H,W),dtype=torch.float64,device="cuda"))
ex_l,ex_r=orb.ORBextractor(),orb.ORBextractor()
T0={}
def frame(seed, rec):
    res={}; t0=time.perf_counter()
    def ext(k,e,im):
        res[k]=e(im); rec[k].append(time.perf_counter()-t0)
    th=[threading.Thread(target=ext,args=(k,e,im)) for k,e,im in (("l",ex_l,d_left),("r",ex_r,d_right))]
    [t.start() for t in th]
    def match():
        [t.join() for t in th]
        (kl,dl),(kr,dr)=res["l"],res["r"]
        res["m"]=orb.stereo_match_begin(ex_l,ex_r,kl,dl,kr,dr,386.1448,386.1448/718.856); rec["m"].append(time.perf_counter()-t0)
    tm=threading.Thread(target=match); tm.start()
    rec["started"].append(time.perf_counter()-t0)
    sn.segment_into(d_bgr, seed, maps); rec["enq"].append(time.perf_counter()-t0)
    cls=maps[0].cpu().numpy(); rec["net"].append(time.perf_counter()-t0)
    tm.join(); rec["join"].append(time.perf_counter()-t0)
import gc
if os.environ.get("FREEZE"): gc.collect(); gc.freeze()
for i in range(3): frame(i, {k:[] for k in ("l","r","m","started","enq","net","join")})
rec={k:[] for k in ("l","r","m","started","enq","net","join")}
for i in range(20): frame(100+i, rec)
import gc; print(os.environ.get("DROP"), threading.active_count(), gc.get_count(), {k: round(1e3*float(np.mean(v)),2) for k,v in rec.items()})
