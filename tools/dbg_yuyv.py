import sys, numpy as np, torch
sys.path.insert(0, '.')
import backscrub_amd as bs
from tests.conftest import model_path
sys.path.insert(0,'tests')
from test_gpu_parity import _camera_yuyv, _dev
W,H,n=640,480,3
rng=np.random.default_rng(41)
a=bs.MaskGen(model_path("lite"),W,H,n_streams=n); b=bs.MaskGen(model_path("lite"),W,H,n_streams=n)
raw=_dev(_camera_yuyv(a,W,H,n,0,rng)); bgr=a.yuyv_to_bgr(raw)
a.run_stage(0,bgr); b.run_stage(4,raw)
ia,ib=a.input_tensor(),b.input_tensor()
print("prep input equal:", torch.equal(ia,ib), int((ia!=ib).sum()), float((ia-ib).abs().max()))
if not torch.equal(ia,ib):
    d=(ia!=ib).any(-1).nonzero()[:10]; print(d)
bg=_dev(rng.integers(0,256,size=(n,H,W,3),dtype=np.uint8))
oa=torch.empty((n,H,W,3),dtype=torch.uint8,device='cuda'); ob=torch.empty_like(oa)
a.step(bgr,bg,oa); b.step_ex(raw,bg,ob,yuyv_in=True)
print("masks equal", torch.equal(a.masks(),b.masks()), "ofinal", torch.equal(a.ofinal(),b.ofinal()), "out", torch.equal(oa,ob), int((oa!=ob).sum()))
d=(oa!=ob).any(-1)
print("pixels differing", int(d.sum()), "per stream", d.flatten(1).sum(1).tolist())
idx=d.nonzero()[:8]
for i in idx.tolist():
    s_,y,x=i
    print(i, oa[s_,y,x].tolist(), ob[s_,y,x].tolist(), "mask", int(a.masks()[s_,y,x]), "bgr", bgr[s_,y,x].tolist(), "raw", raw[s_,y,(x&~1):(x&~1)+2].tolist(), "bg", bg[s_,y,x].tolist())
# columns histogram
xs=d.nonzero()[:,2]; print("x mod 4 histogram", torch.bincount(xs%4).tolist()); ys=d.nonzero()[:,1]; print("y mod 8", torch.bincount(ys%8).tolist())
