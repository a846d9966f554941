"""Which samples does the layer-wise path get wrong at 1,042,555 samples?  Intermediate gradients (retain_grad) of
model._layerwise against the float64 oracle, row by row.  python tests/diag/layerwise_rows.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
from oracle import nerf_oracle as O  # noqa: E402
from nerf import _hip  # noqa: E402
from nerf.models import NeRF, _dense  # noqa: E402
from nerf.encoding import positional_encoding  # noqa: E402

DEV = "cuda"
params = O.closed_form_params(O.nerf_param_shapes())
model = NeRF()
model.load_state_dict(params)
model = model.to(DEV)
p64 = {k: v.to(DEV).double() for k, v in params.items()}

for M in (786432, 1042555):
    g = torch.Generator().manual_seed(777)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.randn(M, 3, generator=g)
    dirs = (dirs / dirs.norm(dim=1, keepdim=True)).to(DEV)
    gc, gd = torch.randn(M, 3, generator=g).to(DEV), (torch.randn(M, generator=g) * 0.1).to(DEV)
    # float64: gradient wrt the rgb head's hidden activations g and the trunk features
    c64, d64, acts = O.nerf_forward(p64, pts.double(), dirs.double(), return_acts=True)
    # layer-wise, by hand, with retained intermediate gradients
    pe = positional_encoding(pts, 10)
    de = positional_encoding(dirs, 4)
    h = pe
    for lin in model.block1[0::2]:
        h = _dense(h, lin, _hip.ACT_RELU)
    h = torch.cat([h, pe], dim=1)
    trunk = list(model.block2[0::2])
    for lin in trunk[:-1]:
        h = _dense(h, lin, _hip.ACT_RELU)
    h.retain_grad()
    feat = _dense(h, trunk[-1], _hip.ACT_NONE, rows=slice(0, 256))
    feat.retain_grad()
    density = _dense(h, trunk[-1], _hip.ACT_RELU, rows=slice(256, 257)).squeeze(-1)
    gg = _dense(torch.cat([feat, de], dim=1), model.rgb_head[0], _hip.ACT_RELU)
    gg.retain_grad()
    colors = _dense(gg, model.rgb_head[2], _hip.ACT_SIGMOID)
    ((colors * gc).sum() + (density * gd).sum()).backward()
    # float64 reference of the same intermediate gradients
    c = c64
    dzo = (gc.double() * c * (1 - c))
    dg_ref = dzo @ p64["rgb_head.2.weight"]
    dzg = dg_ref * (acts["z_g"] > 0)
    dfeat_ref = (dzg @ p64["rgb_head.0.weight"])[:, :256]
    for name, got, ref in (("d g", gg.grad, dg_ref), ("d feat", feat.grad, dfeat_ref)):
        err = (got.double() - ref).norm(dim=1)
        scale = ref.norm(dim=1).mean()
        bad = (err > 1e-4 * scale).nonzero().flatten()
        print(f"M {M} {name}: rel {float(err.norm() / ref.norm()):.2e}; rows off by > 1e-4 of the mean row norm: {bad.numel()}"
              f" first {bad[:8].tolist()} last {bad[-8:].tolist()}", flush=True)
    print(f"   colors rel {float((colors.double() - c64).norm() / c64.norm()):.2e}")
    model.zero_grad(set_to_none=True)
    del acts, c64, d64, pe, de, h, feat, gg, colors, density
    torch.cuda.empty_cache()
