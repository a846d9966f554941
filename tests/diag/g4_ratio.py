"""How far the G4 gradient subsamples are from the golden values, in units of the test tolerance (GPU box)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_parity import make_model, T
for tag in ("nerf", "siren"):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"g4_{tag}.npz"))
    model, _ = make_model(tag)
    colors, dens = model(T(z["pts"]).cuda(), T(z["dirs"]).cuda())
    ((colors * T(z["gc"]).cuda()).sum() + (dens * T(z["gd"]).cuda()).sum()).backward()
    worst = 0; worstn = 0
    for n, p in model.named_parameters():
        gn = float(z["gnorm." + n]); f = p.grad.reshape(-1); step = max(1, f.numel() // 97)
        sub = f[::step][:97].cpu().numpy()
        r = np.abs(sub - z["gsub." + n]).max() / (1e-5 * gn / math.sqrt(f.numel()))
        rn = abs(float(p.grad.norm()) - gn) / (1e-5 * gn)
        worst = max(worst, r); worstn = max(worstn, rn)
    print(tag, "max subsample ratio (units of RTOL*rms):", round(worst, 2), " max norm ratio (units of RTOL):", round(worstn, 3))
