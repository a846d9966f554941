"""Layer-wise path (csrc/linear.hip) against the oracle in float64 on the device, per gradient tensor, at growing
sample counts: where does its error come from?  python tests/diag/layerwise_accuracy.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
from oracle import nerf_oracle as O  # noqa: E402
from nerf.models import NeRF  # noqa: E402

DEV = "cuda"
params = O.closed_form_params(O.nerf_param_shapes())
model = NeRF()
model.load_state_dict(params)
model = model.to(DEV)
p64 = {k: v.to(DEV).double().requires_grad_(True) for k, v in params.items()}


def r(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


for M in (4096, 65536, 262144, 1048576):
    g = torch.Generator().manual_seed(777)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.randn(M, 3, generator=g)
    dirs = (dirs / dirs.norm(dim=1, keepdim=True)).to(DEV)
    gc, gd = torch.randn(M, 3, generator=g).to(DEV), (torch.randn(M, generator=g) * 0.1).to(DEV)
    with torch.no_grad():
        _, _, acts = O.nerf_forward(p64, pts.double(), dirs.double(), return_acts=True)
        keep = O.relu_kink_distance(acts) > 3e-6
        del acts
    pts, dirs, gc, gd = (x[keep].contiguous() for x in (pts, dirs, gc, gd))
    for p in p64.values():
        p.grad = None
    c64, d64 = O.nerf_forward(p64, pts.double(), dirs.double())
    ((c64 * gc.double()).sum() + (d64 * gd.double()).sum()).backward()
    ref = {k: p.grad.clone() for k, p in p64.items()}
    out = {}
    for name, call in (("fused", model), ("layer", model._layerwise)):
        model.zero_grad(set_to_none=True)
        c, d = call(pts, dirs)
        ((c * gc).sum() + (d * gd).sum()).backward()
        out[name] = ({n: r(p.grad, ref[n]) for n, p in model.named_parameters()}, r(c, c64), r(d, d64))
    print(f"M {M} kept {int(keep.sum())}: forward colors/density fused {out['fused'][1]:.1e}/{out['fused'][2]:.1e} "
          f"layer {out['layer'][1]:.1e}/{out['layer'][2]:.1e}")
    for n in out["fused"][0]:
        print(f"   {n:22s} fused {out['fused'][0][n]:.2e}  layer {out['layer'][0][n]:.2e}")
    torch.cuda.empty_cache()
