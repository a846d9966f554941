import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np, torch
from oracle import nerf_oracle as O
from nerf import _hip
from nerf.models import NeRF
DEV="cuda"
def rel(a,b):
    a=a.detach().cpu().double().reshape(-1); b=b.detach().cpu().double().reshape(-1)
    return float((a-b).norm()/(b.norm()+1e-30))
M=1000
params = O.closed_form_params(O.nerf_param_shapes())
model = NeRF(); model.load_state_dict(params); model=model.to(DEV)
g = torch.Generator().manual_seed(M)
pts = torch.rand(M, 3, generator=g) * 2 - 1
dirs = torch.randn(M, 3, generator=g); dirs = dirs / dirs.norm(dim=1, keepdim=True)
gc, gd = torch.randn(M, 3, generator=g), torch.randn(M, generator=g) * 0.1
for p in params.values(): p.requires_grad_(True)
c_ref, d_ref, acts = O.nerf_forward(params, pts, dirs, return_acts=True)
((c_ref * gc).sum() + (d_ref * gd).sum()).backward()
# raw path
packed = model.packed_weights()
colors, dens, ws = _hip.mlp_fwd(0, packed, M, points=pts.to(DEV), dirs=dirs.to(DEV), train=True)
flat = _hip.mlp_bwd(0, packed, M, gc.to(DEV), gd.to(DEV), colors, ws)
off=0
print("raw path")
for n,p in params.items():
    k=p.numel(); r=rel(flat[off:off+k], p.grad); off+=k
    if r>1e-5: print("  ", n, r)
# per-sample check: which samples' density mask differ
sr = acts["f"][:, -1]
Mp=1024
sraw = ws[(96+10*256+128)*Mp:(96+10*256+128)*Mp+Mp][:M].cpu()
print("sraw rel", rel(sraw, sr), "sign mismatches", int(((sraw>0)!=(sr>0)).sum()), "min |sraw|", float(sr.abs().min()))
for l in range(9):
    h = ws[(96+256*l)*Mp:(96+256*l)*Mp+Mp*256].view(Mp,256)[:M].cpu()
    hr = acts[f"h{l}"]
    mism = ((h>0)!=(hr>0))
    print(f"h{l} mask mismatches", int(mism.sum()), "max |ref| at mismatches", float(hr[mism].abs().max()) if mism.any() else 0.0)
gm = ws[(96+10*256)*Mp:(96+10*256)*Mp+Mp*128].view(Mp,128)[:M].cpu()
print("g mask mismatches", int(((gm>0)!=(acts['g']>0)).sum()))
# autograd path
colors2, dens2 = model(pts.to(DEV), dirs.to(DEV))
((colors2 * gc.to(DEV)).sum() + (dens2 * gd.to(DEV)).sum()).backward()
print("autograd path")
for (n,p),pr in zip(model.named_parameters(), params.values()):
    r=rel(p.grad, pr.grad)
    if r>1e-5: print("  ", n, r)
