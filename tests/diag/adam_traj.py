"""Deviation of the five-step Adam loss trajectory from the golden one, per step (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_parity import make_model, T
from nerf import parallel
from nerf.rendering import render_nerf
for tag in ("nerf", "siren"):
    for optim in ("torch", "fused"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"g7_{tag}.npz"))
        R, S, steps = (int(v) for v in z["cfg"])
        model, _ = make_model(tag)
        o, d, tgt = T(z["o"]).cuda(), T(z["d"]).cuda(), T(z["target"]).cuda()
        lr = 5e-4
        if optim == "torch":
            opt = torch.optim.Adam(model.parameters(), lr=lr)
        else:
            opt = parallel.FusedAdam(parallel.FlatParameters(model), lr=lr)
        devs = []
        for it in range(steps):
            pred = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device="cuda", chunk_size=8192, u=[T(z["us"][it]).cuda()])
            opt.zero_grad()
            loss = torch.nn.functional.mse_loss(pred, tgt)
            loss.backward()
            opt.step()
            devs.append(abs(float(loss.detach()) - z["losses"][it]) / z["losses"][it])
        pn = max(abs(float(p.detach().norm()) - float(z["pnorm." + n])) / (float(z["pnorm." + n]) + 1e-12) for n, p in model.named_parameters())
        print(tag, optim, ["%.1e" % v for v in devs], "max param-norm deviation %.1e" % pn)
