"""Generate the golden vectors in this directory by importing the REFERENCE.

Run once in the build container (the reference is mounted at /root/reference
there and nowhere else):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The outputs (*.npz) are data only: seeded inputs and the reference's outputs.
Weights are not stored; they come from oracle.nerf_oracle.closed_form_params
(a pure index hash), which this script and the tests share.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# NERF_GOLDEN_OUT redirects the output (tests/test_oracle_golden.py's "live" source regenerates the vectors on the
# host it runs on, into a temporary directory, and holds the oracle to them bit for bit).
OUT = os.environ.get("NERF_GOLDEN_OUT", HERE)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from nerf.rendering import (  # noqa: E402  (reference)
    stratified_sampling, generate_sample_positions, normalize_positions,
    compute_accumulated_transmittance, composite_volume, render_nerf)
from nerf.encoding import positional_encoding  # noqa: E402  (reference)
from nerf.models import NeRF, Siren  # noqa: E402  (reference)
from oracle import nerf_oracle as O  # noqa: E402  (weight/input generators only)

torch.set_num_threads(8)


def u01(n, stream):
    return torch.from_numpy(((O.hash_uniform(n, stream) + 1.0) * 0.5).astype(np.float32))


def sym(n, stream, scale=1.0):
    return torch.from_numpy((O.hash_uniform(n, stream) * scale).astype(np.float32))


def save(name, **kw):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in kw.items()}
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


def rays(R, stream):
    """Origins on the radius-4 sphere, unit directions toward the origin + jitter."""
    o = sym(R * 3, stream).reshape(R, 3)
    o = 4.0 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4.0 + 0.15 * sym(R * 3, stream + 1).reshape(R, 3)
    d = d / d.norm(dim=1, keepdim=True)
    return o.contiguous(), d.contiguous()


def subsample(t, n=97):
    f = t.reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].clone()


# ---- G1: stratified_sampling ------------------------------------------------
g1 = {}
for ci, (near, far, S, seed) in enumerate([(2.0, 6.0, 64, 1), (2.0, 6.0, 128, 2), (2.0, 6.0, 192, 3),
                                           (2.0, 6.0, 256, 4), (0.5, 7.3, 193, 5), (0.1, 1000.0, 256, 6),
                                           (1.3, 2.9, 7, 7)]):
    torch.manual_seed(seed)
    u = torch.rand(S)
    torch.manual_seed(seed)
    t = stratified_sampling(near, far, S)
    g1[f"c{ci}_cfg"] = np.array([near, far, S, seed], dtype=np.float64)
    g1[f"c{ci}_u"] = u
    g1[f"c{ci}_t"] = t
    g1[f"c{ci}_edges"] = torch.linspace(near, far, S + 1)
save("g1_stratified.npz", **g1)

# ---- G2: generate_sample_positions -----------------------------------------
o, d = rays(32, 10)
g2 = {"o": o, "d": d}
for S, seed in [(64, 11), (256, 12)]:
    torch.manual_seed(seed)
    u = torch.rand(S)
    torch.manual_seed(seed)
    pos, deltas = generate_sample_positions(o, d, 2.0, 6.0, S)
    g2[f"S{S}_u"] = u
    g2[f"S{S}_pos"] = pos
    g2[f"S{S}_deltas"] = deltas
    g2[f"S{S}_posn"] = normalize_positions(pos, 2.0, 6.0)
save("g2_positions.npz", **g2)

# ---- G3: positional_encoding ------------------------------------------------
x = sym(61 * 3, 20, 1.2).reshape(61, 3)
x = torch.cat([x, torch.tensor([[2.0, -2.0, 0.0], [1.0, -1.0, 0.5], [1e-3, -1e-6, 1.999]])], 0)
save("g3_posenc.npz", x=x, L4=positional_encoding(x, 4), L10=positional_encoding(x, 10))


# ---- G4: model forward + parameter gradients --------------------------------
def model_golden(tag, model, params, M, stream):
    model.load_state_dict(params)
    pts = sym(M * 3, stream, 1.0).reshape(M, 3)
    dirs = sym(M * 3, stream + 1).reshape(M, 3)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    gc = sym(M * 3, stream + 2).reshape(M, 3)
    gd = sym(M, stream + 3, 0.1)
    colors, dens = model(pts, dirs)
    (colors * gc).sum().add((dens * gd).sum()).backward()
    out = {"pts": pts, "dirs": dirs, "gc": gc, "gd": gd, "colors": colors, "density": dens}
    for n, p in model.named_parameters():
        out["gnorm." + n] = p.grad.norm().double()
        out["gsub." + n] = subsample(p.grad)
    save(f"g4_{tag}.npz", **out)


model_golden("nerf", NeRF(), O.closed_form_params(O.nerf_param_shapes()), 96, 30)
model_golden("nerf_h64", NeRF(hidden_dim=64), O.closed_form_params(O.nerf_param_shapes(hidden=64)), 80, 40)
model_golden("siren", Siren(), O.closed_form_params(O.siren_param_shapes(), gains=O.SIREN_GAINS), 96, 50)
model_golden("siren_h64", Siren(hidden_dim=64),
             O.closed_form_params(O.siren_param_shapes(hidden=64), gains=O.SIREN_GAINS), 80, 60)

# ---- G5: composite_volume forward + grads, transmittance --------------------
R, S = 24, 64
torch.manual_seed(70)
dens = torch.relu(torch.randn(R, S)) * 10
dens[0] = 0.0                       # empty ray
dens[1, 10] = 5000.0                # beta == 0 in mid-ray
dens[2] = 1e-3                      # nearly transparent
dens[3, :] = 50.0                   # saturates at first sample
cols = torch.rand(R, S, 3)
t = stratified_sampling(2.0, 6.0, S)
deltas = torch.cat([t[1:] - t[:-1], torch.tensor([1e10])])
grgb = sym(R * 3, 71).reshape(R, 3)
g5 = {"colors": cols, "densities": dens, "deltas": deltas, "grad_rgb": grgb, "t": t}
for wb in (True, False):
    c = cols.clone().requires_grad_(True)
    s = dens.clone().requires_grad_(True)
    rgb = composite_volume(c, s, deltas, wb)
    (rgb * grgb).sum().backward()
    alpha = 1 - torch.exp(-dens * deltas.unsqueeze(0))
    w = compute_accumulated_transmittance(1 - alpha) * alpha
    g5[f"wb{int(wb)}_rgb"] = rgb
    g5[f"wb{int(wb)}_dcolors"] = c.grad
    g5[f"wb{int(wb)}_ddens"] = s.grad
    g5["weights"] = w
    g5["trans"] = compute_accumulated_transmittance(1 - alpha)
save("g5_composite.npz", **g5)


# ---- G6: render_nerf end to end ---------------------------------------------
def render_golden(tag, model, params, R, S, chunk, seed, wb=True):
    model.load_state_dict(params)
    model.zero_grad()
    o, d = rays(R, 80)
    target = u01(R * 3, 82).reshape(R, 3)
    nchunks = (R + chunk - 1) // chunk
    torch.manual_seed(seed)
    us = torch.stack([torch.rand(S) for _ in range(nchunks)])
    torch.manual_seed(seed)
    rgb = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device="cpu", white_background=wb, chunk_size=chunk)
    loss = torch.nn.functional.mse_loss(rgb, target)
    loss.backward()
    out = {"o": o, "d": d, "target": target, "us": us, "rgb": rgb, "loss": loss.double(),
           "cfg": np.array([R, S, chunk, seed, int(wb)])}
    for n, p in model.named_parameters():
        out["gnorm." + n] = p.grad.norm().double()
        out["gsub." + n] = subsample(p.grad)
    save(f"g6_{tag}.npz", **out)


# density-friendly weights: scale the density row so that rays are neither empty nor opaque
pn = O.closed_form_params(O.nerf_param_shapes())
ps = O.closed_form_params(O.siren_param_shapes(), gains=O.SIREN_GAINS)
render_golden("nerf_S64_onechunk", NeRF(), pn, 64, 64, 8192, 90)
render_golden("nerf_S256_chunk24", NeRF(), pn, 64, 256, 24, 91)
render_golden("nerf_S64_black", NeRF(), pn, 40, 64, 16, 92, wb=False)
render_golden("siren_S64_onechunk", Siren(), ps, 64, 64, 8192, 93)
render_golden("siren_S256_chunk24", Siren(), ps, 64, 256, 24, 94)


# ---- G7: five Adam steps (train.py:123-131,166-183) --------------------------
def train_golden(tag, model, params, R, S, steps=5, lr=5e-4, lr_decay=300.0, lr_decay_factor=0.1, lr_min=1e-4):
    model.load_state_dict(params)
    o, d = rays(R, 100)
    target = u01(R * 3, 102).reshape(R, 3)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    gamma = lr_decay_factor ** (1 / (lr_decay * 1000))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: max(gamma ** s, lr_min / lr))
    mse = torch.nn.MSELoss()
    torch.manual_seed(123)
    us = torch.stack([torch.rand(S) for _ in range(steps)])
    torch.manual_seed(123)
    losses, lrs = [], []
    for _ in range(steps):
        pred = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device="cpu", white_background=True, chunk_size=8192)
        opt.zero_grad()
        loss = mse(pred, target)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
        lrs.append(sched.get_last_lr()[0])
    out = {"o": o, "d": d, "target": target, "us": us, "losses": np.array(losses, dtype=np.float64),
           "lrs": np.array(lrs, dtype=np.float64), "cfg": np.array([R, S, steps])}
    for n, p in model.named_parameters():
        out["pnorm." + n] = p.detach().norm().double()
        out["psub." + n] = subsample(p.detach())
    save(f"g7_{tag}.npz", **out)


train_golden("nerf", NeRF(), pn, 48, 64)
train_golden("siren", Siren(), ps, 48, 64)
print("done")
