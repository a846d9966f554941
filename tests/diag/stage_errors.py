"""Stage-by-stage GPU vs oracle error report (not a pytest; run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np, torch
from oracle import nerf_oracle as O
from nerf import _hip
from nerf.models import NeRF, Siren

dev = "cuda"
torch.manual_seed(0)

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30)), float((a - b).abs().max())

def run(tag, M=int(os.environ.get("DBG_M", "200"))):
    is_nerf = tag == "nerf"
    shapes = O.nerf_param_shapes() if is_nerf else O.siren_param_shapes()
    params = O.closed_form_params(shapes, gains=None if is_nerf else O.SIREN_GAINS)
    model = (NeRF() if is_nerf else Siren())
    model.load_state_dict(params); model = model.to(dev)
    g = torch.Generator().manual_seed(1)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1)
    dirs = torch.randn(M, 3, generator=g); dirs = dirs / dirs.norm(dim=1, keepdim=True)
    gc = torch.randn(M, 3, generator=g); gd = torch.randn(M, generator=g) * 0.1
    for p in params.values(): p.requires_grad_(True)
    fwd = O.nerf_forward if is_nerf else O.siren_forward
    c_ref, d_ref, acts = fwd(params, pts, dirs, return_acts=True)
    ((c_ref * gc).sum() + (d_ref * gd).sum()).backward()
    mid = 0 if is_nerf else 1
    plist = model.ordered_parameters()
    packed = model.packed_weights()
    colors, dens, ws = _hip.mlp_fwd(mid, packed, M, points=pts.to(dev), dirs=dirs.to(dev), train=True)
    torch.cuda.synchronize()
    print(f"[{tag}] colors", rel(colors, c_ref), "density", rel(dens, d_ref))
    Mp = (M + 63) // 64 * 64
    def arr(off, width): return ws[off * Mp: off * Mp + Mp * width].view(Mp, width)[:M]
    if is_nerf:
        print("  pe", rel(arr(0, 64)[:, :63], acts["pe"]), "de", rel(arr(64, 32)[:, :27], acts["de"]))
        for l in range(9): print(f"  h{l}", rel(arr(96 + 256 * l, 256), acts[f"h{l}"]))
        print("  f", rel(arr(96 + 9 * 256, 256), acts["f"][:, :256]), "g", rel(arr(96 + 10 * 256, 128), acts["g"]))
    else:
        for l in range(8): print(f"  h{l}", rel(arr(36 + 256 * l, 256), acts[f"h{l}"]))
        print("  f", rel(arr(36 + 16 * 256, 256), acts["f"]), "g", rel(arr(36 + 17 * 256, 128), acts["g"]))
    flat = _hip.mlp_bwd(mid, packed, M, gc.to(dev), gd.to(dev), colors, ws)
    torch.cuda.synchronize()
    off = 0
    for (name, p) in params.items():
        n = p.numel(); gg = flat[off:off + n].view(p.shape); off += n
        r = rel(gg, p.grad)
        print(f"  grad {name:28s} rel {r[0]:.2e} maxabs {r[1]:.2e}")
    # inference mode equals training mode
    c2, d2, _ = _hip.mlp_fwd(mid, packed, M, points=pts.to(dev), dirs=dirs.to(dev), train=False)
    print("  infer==train", bool(torch.equal(c2, colors) and torch.equal(d2, dens)))

for tag in sys.argv[1:] or ["nerf", "siren"]:
    run(tag)
