"""How reproducible is a 120-step training curve at all?  The oracle against itself with 1e-7 relative noise on the initial
weights (CPU, about a minute): PSNR gaps per 20-step window, for the tolerance of test_training_psnr_curve_matches_the_oracle."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nerf_oracle as O
torch.set_num_threads(8)
def rays(R, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(R, 3, generator=g); o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.15 * torch.randn(R, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    return o.contiguous(), d.contiguous()
def run(tag, eps):
    is_nerf = tag == "nerf"
    shapes = O.nerf_param_shapes() if is_nerf else O.siren_param_shapes()
    params = O.closed_form_params(shapes, gains=None if is_nerf else O.SIREN_GAINS)
    fwd = O.nerf_forward if is_nerf else O.siren_forward
    R, S, steps, lr = 64, 32, 120, 5e-4
    o, d = rays(R, 900)
    tgt = 0.5 + 0.5 * torch.sin(3.0 * d + torch.tensor([0.0, 1.0, 2.0]))
    g = torch.Generator().manual_seed(901)
    us = [torch.rand(S, generator=g) for _ in range(steps)]
    gp = torch.Generator().manual_seed(5)
    plist = [(p * (1 + eps * torch.randn(p.shape, generator=gp))).clone().requires_grad_(True) for p in params.values()]
    pdict = dict(zip(params.keys(), plist))
    m = [torch.zeros_like(p) for p in plist]; v = [torch.zeros_like(p) for p in plist]
    curve = []
    for it in range(steps):
        rgb = O.render(lambda x, dd: fwd(pdict, x, dd), o, d, 2.0, 6.0, S, [us[it]])
        loss = torch.nn.functional.mse_loss(rgb, tgt)
        grads = torch.autograd.grad(loss, plist)
        with torch.no_grad():
            for p, gr, mm, vv in zip(plist, grads, m, v):
                O.adam_step(p, gr, mm, vv, it + 1, lr * O.lr_lambda(it, lr, 300.0, 0.1, 1e-4))
        curve.append(float(loss.detach()))
    return curve
psnr = lambda x: 20 * math.log10(1 / math.sqrt(x))
for tag in ("siren", "nerf"):
    a, b = run(tag, 0.0), run(tag, 1e-7)
    win = lambda c, k: psnr(sum(c[k:k + 20]) / 20)
    print(tag, "oracle vs oracle with 1e-7 relative weight noise: window gaps", ["%.3f" % abs(win(a, k) - win(b, k)) for k in range(0, 120, 20)])
