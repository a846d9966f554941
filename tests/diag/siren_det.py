import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/nerf-pytorch_amd")
import torch
from nerf import _hip
from nerf.models import Siren
from oracle import nerf_oracle as O
m = Siren(); m.load_state_dict(O.closed_form_params(O.siren_param_shapes(), gains=O.SIREN_GAINS)); m = m.cuda()
packed = m.packed_weights(True)
for M in (64, 128, 4096, 65536, 1048576):
    g = torch.Generator().manual_seed(M)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).cuda(); dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).cuda()
    dc = torch.randn(M, 3, generator=g).cuda(); dd = torch.randn(M, generator=g).cuda()
    outs = []
    for rep in range(3):
        colors, dens, ws = _hip.mlp_fwd(1, packed, M, points=pts, dirs=dirs, train=True)
        ws_copy = ws.clone()
        grads = _hip.mlp_bwd(1, packed, M, dc, dd, colors, ws)
        torch.cuda.synchronize()
        outs.append((ws_copy, grads.clone()))
    same_ws = torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32))
    same_g = torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][1], outs[2][1])
    print(M, "ws identical:", same_ws, "grads identical:", same_g, "max grad diff", float((outs[0][1] - outs[1][1]).abs().max()))
M = 65536
Mp = M
g = torch.Generator().manual_seed(M)
pts = (torch.rand(M, 3, generator=g) * 2 - 1).cuda(); dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).cuda()
a = _hip.mlp_fwd(1, packed, M, points=pts, dirs=dirs, train=True)[2].clone()
b = _hip.mlp_fwd(1, packed, M, points=pts, dirs=dirs, train=True)[2].clone()
H = 256
regions = [("xn", 0, 4), ("de", 4, 32)] + [(f"h{l}", 36 + l * H, H) for l in range(8)] + [(f"c{l}", 36 + 8 * H + l * H, H) for l in range(8)] + \
          [("f", 36 + 16 * H, H), ("g", 36 + 17 * H, 128), ("cg", 36 + 17 * H + 128, 128), ("sraw", 36 + 17 * H + 256, 1)]
for name, off, w in regions:
    ra, rb = a[off * Mp:(off + w) * Mp].view(torch.int32), b[off * Mp:(off + w) * Mp].view(torch.int32)
    nd = int((ra != rb).sum())
    if nd:
        idx = (ra != rb).nonzero().flatten()
        print(name, "differs in", nd, "of", ra.numel(), "first float index", int(idx[0]), "tile", int(idx[0]) // (64 * w), "last tile", int(idx[-1]) // (64 * w))
off = 36 + 8 * H + 1 * H
ra, rb = a[off * Mp:(off + H) * Mp].view(torch.int32), b[off * Mp:(off + H) * Mp].view(torch.int32)
idx = (ra != rb).nonzero().flatten().cpu()
tile = idx // 16384
t0 = int(tile[0])
sel = idx[tile == t0] % 16384
print("tile", t0, "differing floats", sel.numel(), "piece index q:", sorted(set((sel // 4096).tolist())), "threads:", sorted(set(((sel % 4096) // 4).tolist()))[:20], "...")
import collections
print("per-tile counts (first 10):", collections.Counter(tile.tolist()).most_common(5))
va, vb = a[off * Mp:(off + H) * Mp][idx[:8].cuda()], b[off * Mp:(off + H) * Mp][idx[:8].cuda()]
print(va.tolist(), vb.tolist())
