import sys, torch
sys.path.insert(0, "nerf-pytorch_amd")
from nerf.models import NeRF
from nerf.rendering import render_nerf_hierarchical, render_nerf
dev = "cuda"
for seed in (11, 12, 13):
    torch.manual_seed(seed)
    ms = [NeRF().to(dev) for _ in range(2)]
    g = torch.Generator().manual_seed(7)
    o = torch.randn(512, 3, generator=g); o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.2 * torch.randn(512, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    o, d = o.to(dev), d.to(dev)
    rc, rf = render_nerf_hierarchical(ms[0], ms[1], o, d, 2.0, 6.0, 32, 64, device=dev)
    (rc.sum() + rf.sum()).backward()
    print(seed, "rc mean", float(rc.mean()), "rf mean", float(rf.mean()),
          "grad norms", [float(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).norm()) for m in ms])
    for i, m in enumerate(ms):
        with torch.no_grad():
            x = torch.rand(4096, 3, device=dev) * 2 - 1
            v = torch.nn.functional.normalize(torch.randn(4096, 3, device=dev), dim=1)
            c, s = m(x, v)
        print("   model", i, "frac density>0", float((s > 0).float().mean()), "max", float(s.max()))
