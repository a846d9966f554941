"""CPU oracle for the NeRF ray-march -> encode -> MLP -> composite path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
and the cpu_baseline leg of bench.py may import it; the product path
(nerf-pytorch_amd/) never does and fails loudly without its HIP library.

It is a functional, fp32, CPU-only restatement of the algorithm of the
reference repository (josedelrey/nerf-pytorch).  Every function names the
reference lines it follows.  The arithmetic primitives (sin, cos, exp, matmul,
linspace) are ATen's CPU kernels, as in the reference itself, which has no
arithmetic of its own.

Pinning: tests/test_oracle_golden.py checks every function here against the
golden vectors in tests/golden/*.npz, which were produced by importing the
reference itself (tests/golden/make_golden.py, run once in the build
container).  Functions that have no counterpart in the reference
(sample_pdf, depth output, coarse+fine rendering) are marked
"parity unpinned" in their docstrings.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# ----------------------------------------------------------------------------
# deterministic closed-form weights (shared by golden generation and GPU tests)
# ----------------------------------------------------------------------------

NERF_SHAPES: List[Tuple[str, Tuple[int, ...]]] = []
SIREN_SHAPES: List[Tuple[str, Tuple[int, ...]]] = []


def nerf_param_shapes(hidden: int = 256, pos_L: int = 10, dir_L: int = 4):
    """Parameter names/shapes of the reference NeRF module (models.py:25-57)."""
    pe, de = 3 + 6 * pos_L, 3 + 6 * dir_L
    shapes = []
    ins = [pe, hidden, hidden, hidden, hidden]
    for i, k in enumerate(ins):
        shapes += [(f"block1.{2*i}.weight", (hidden, k)), (f"block1.{2*i}.bias", (hidden,))]
    ins = [hidden + pe, hidden, hidden, hidden]
    for i, k in enumerate(ins):
        shapes += [(f"block2.{2*i}.weight", (hidden, k)), (f"block2.{2*i}.bias", (hidden,))]
    shapes += [("block2.8.weight", (hidden + 1, hidden)), ("block2.8.bias", (hidden + 1,))]
    shapes += [("rgb_head.0.weight", (hidden // 2, hidden + de)), ("rgb_head.0.bias", (hidden // 2,))]
    shapes += [("rgb_head.2.weight", (3, hidden // 2)), ("rgb_head.2.bias", (3,))]
    return shapes


def siren_param_shapes(hidden: int = 256, num_layers: int = 8, dir_L: int = 4):
    """Parameter names/shapes of the reference Siren module (models.py:163-183)."""
    de = 3 + 6 * dir_L
    shapes = [("block1.0.layer.weight", (hidden, 3)), ("block1.0.layer.bias", (hidden,))]
    for i in range(1, num_layers):
        shapes += [(f"block1.{i}.layer.weight", (hidden, hidden)), (f"block1.{i}.layer.bias", (hidden,))]
    shapes += [("density_branch.0.weight", (1, hidden)), ("density_branch.0.bias", (1,))]
    shapes += [("feature_remap.0.weight", (hidden, hidden)), ("feature_remap.0.bias", (hidden,))]
    shapes += [("rgb_head.0.layer.weight", (hidden // 2, hidden + de)), ("rgb_head.0.layer.bias", (hidden // 2,))]
    shapes += [("rgb_head.1.weight", (3, hidden // 2)), ("rgb_head.1.bias", (3,))]
    return shapes


def hash_uniform(n: int, stream: int) -> np.ndarray:
    """n deterministic uniforms in [-1,1) from a 64-bit integer mix of
    (index, stream): no RNG state, no files, identical on every machine."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) * np.uint64(6364136223846793005)
        x = x + np.uint64(1442695040888963407) * np.uint64(2 * stream + 1)
        x ^= x >> np.uint64(33)
        x = x * np.uint64(0xFF51AFD7ED558CCD)
        x ^= x >> np.uint64(33)
        x = x * np.uint64(0xC4CEB9FE1A85EC53)
        x ^= x >> np.uint64(33)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24)) * 2.0 - 1.0


def closed_form_params(shapes, gain: float = 1.0, seed: int = 0,
                       gains: Optional[Dict[str, float]] = None) -> Params:
    """Deterministic full-rank pseudo-random weights without any file or RNG
    state: weight ~ U(+-gain*sqrt(6/fan_in)), bias ~ U(+-0.05).  The exact law
    does not matter, only that golden generation and the tests agree on it."""
    out: Params = {}
    for ti, (name, shape) in enumerate(shapes):
        n = int(np.prod(shape))
        g = gain if not gains or name not in gains else gains[name]
        bound = g * math.sqrt(6.0 / shape[-1]) if len(shape) == 2 else 0.05
        raw = hash_uniform(n, 1000 * seed + ti) * bound
        out[name] = torch.from_numpy(raw.astype(np.float32)).reshape(shape).contiguous()
    return out


SIREN_GAINS = {"block1.0.layer.weight": 1.0 / (3.0 * math.sqrt(2.0))}


# ----------------------------------------------------------------------------
# sampling (nerf/rendering.py:6-64)
# ----------------------------------------------------------------------------

def _round_to_f32(x: Fraction) -> np.float32:
    """Round an exact rational to the nearest fp32 (ties to even)."""
    if x == 0:
        return np.float32(0.0)
    sign = -1 if x < 0 else 1
    x = abs(x)
    e = math.floor(math.log2(float(x)))
    # fix possible off-by-one from the float log
    while Fraction(2) ** e > x:
        e -= 1
    while Fraction(2) ** (e + 1) <= x:
        e += 1
    e = max(e, -126)
    scale = Fraction(2) ** (e - 23)
    q = x / scale
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(sign * float(Fraction(n) * scale))


def linspace_fma(near: float, far: float, steps: int) -> np.ndarray:
    """Explicit restatement of ATen's CPU fp32 linspace as used at rendering.py:23.

    step = (far-near)/(steps-1) rounded to fp32; element i is
    fma(step, i, near) for i < steps//2 and fma(-step, steps-1-i, far) otherwise,
    each with ONE rounding.  (Checked equal to torch.linspace in
    tests/test_oracle_golden.py; the HIP sampling kernel implements this law.)
    """
    n32, f32 = np.float32(near), np.float32(far)
    step = np.float32((f32 - n32) / np.float32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = _round_to_f32(Fraction(float(step)) * i + Fraction(float(n32)))
        else:
            out[i] = _round_to_f32(Fraction(float(f32)) - Fraction(float(step)) * (steps - 1 - i))
    return out


def stratified_t(near: float, far: float, num_bins: int, u: Tensor) -> Tensor:
    """t = lower + (upper-lower)*u over S equal bins (rendering.py:23-27).

    `u` is the caller-supplied uniform draw of shape [S] (the reference draws it
    with torch.rand at rendering.py:26).  One t-vector is shared by all rays.
    """
    edges = torch.linspace(near, far, num_bins + 1)
    lo, hi = edges[:-1], edges[1:]
    return lo + (hi - lo) * u.to(torch.float32)


def sample_deltas(t: Tensor) -> Tensor:
    """Successive differences with a trailing 1e10 (rendering.py:54-57)."""
    return torch.cat([t[1:] - t[:-1], torch.full((1,), 1e10, dtype=t.dtype)])


def sample_positions(rays_o: Tensor, rays_d: Tensor, t: Tensor) -> Tensor:
    """pos[r,s,:] = o[r] + t[s]*d[r] (rendering.py:59-62). -> [R,S,3]"""
    return rays_o[:, None, :] + t[None, :, None] * rays_d[:, None, :]


def normalize_positions(p: Tensor, near: float, far: float) -> Tensor:
    """2*(p-near)/(far-near)-1, in that operation order (rendering.py:82)."""
    return 2 * (p - near) / (far - near) - 1


# ----------------------------------------------------------------------------
# encoding (nerf/encoding.py:4-20)
# ----------------------------------------------------------------------------

def positional_encoding(x: Tensor, L: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] on dim 1."""
    cols = [x]
    for j in range(L):
        f = float(2 ** j)
        cols += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(cols, dim=1)


# ----------------------------------------------------------------------------
# models (nerf/models.py)
# ----------------------------------------------------------------------------

def _lin(x: Tensor, p: Params, key: str) -> Tensor:
    return torch.addmm(p[key + ".bias"], x, p[key + ".weight"].t())


def nerf_forward(p: Params, points: Tensor, dirs: Tensor, pos_L: int = 10, dir_L: int = 4,
                 return_acts: bool = False):
    """NeRF MLP (models.py:62-75): 5-layer block, skip-concat [features, pe],
    5-layer block whose LAST output column is the raw density, rgb head on
    [features, de] with sigmoid."""
    pe = positional_encoding(points, pos_L)
    de = positional_encoding(dirs, dir_L)
    acts = {"pe": pe, "de": de}
    h = pe
    for i in range(5):
        z = _lin(h, p, f"block1.{2*i}")
        h = torch.relu(z)
        acts[f"z{i}"], acts[f"h{i}"] = z, h
    h = torch.cat([h, pe], dim=1)
    for i in range(4):
        z = _lin(h, p, f"block2.{2*i}")
        h = torch.relu(z)
        acts[f"z{5+i}"], acts[f"h{5+i}"] = z, h
    f = _lin(h, p, "block2.8")
    acts["f"] = f
    acts["z_sigma"] = f[:, -1:]
    density = torch.relu(f[:, -1])
    zg = _lin(torch.cat([f[:, :-1], de], dim=1), p, "rgb_head.0")
    g = torch.relu(zg)
    acts["z_g"], acts["g"] = zg, g
    colors = torch.sigmoid(_lin(g, p, "rgb_head.2"))
    if return_acts:
        return colors, density, acts
    return colors, density


def bf16_round(x: Tensor) -> Tensor:
    """fp32 -> bf16 (round to nearest even) -> fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


def nerf_forward_bf16(p: Params, points: Tensor, dirs: Tensor, terms: int = 1):
    """The opt-in bf16 inference mode (no counterpart in the reference, which is fp32 throughout):
    nerf_forward with every weight and every layer input (encodings, activations) rounded to bf16,
    products accumulated in higher precision (float64 here), biases / sigmoid / inputs in fp32.
    terms = 3: the three-term mode instead (linear_bf16x3), whose result is fp32-accurate."""
    def lin(x, name):
        if terms == 3:
            return linear_bf16x3(x, p[name + ".weight"], p[name + ".bias"])
        w = bf16_round(p[name + ".weight"]).double()
        return (bf16_round(x).double() @ w.t() + p[name + ".bias"].double()).float()

    pe = positional_encoding(points, 10)
    de = positional_encoding(dirs, 4)
    h = pe
    for i in range(5):
        h = torch.relu(lin(h, f"block1.{2*i}"))
    h = torch.cat([h, pe], dim=1)
    for i in range(4):
        h = torch.relu(lin(h, f"block2.{2*i}"))
    f = lin(h, "block2.8")
    density = torch.relu(f[:, -1])
    g = torch.relu(lin(torch.cat([f[:, :-1], de], dim=1), "rgb_head.0"))
    colors = torch.sigmoid(lin(g, "rgb_head.2"))
    return colors, density


def bf16_terms(x: Tensor, terms: int = 3):
    """The three-term mode's expansion of fp32 values (no counterpart in the reference): x0 = bf16(x),
    x1 = bf16(x - x0), x2 = bf16(x - x0 - x1), every subtraction exact in fp32.  Returns the terms as fp32 tensors."""
    out, r = [], x.to(torch.float32)
    for _ in range(terms):
        t = bf16_round(r)
        out.append(t)
        r = r - t
    return out


def linear_bf16x3(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """x @ w.T + b as the three-term kernels form it: the six term products w0x0, w0x1, w0x2, w1x0, w1x1, w2x0 (each
    exact in fp32), accumulated here in float64 -- what remains relative to the exact product is the three dropped
    term products, below 2^-24 of |w||x|."""
    xs, ws = bf16_terms(x), bf16_terms(w)
    acc = torch.zeros(x.shape[0], w.shape[0], dtype=torch.float64)
    for i, j in ((0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (2, 0)):
        acc += xs[j].double() @ ws[i].double().t()
    return (acc + b.double()).float()


def siren_forward(p: Params, points: Tensor, dirs: Tensor, num_layers: int = 8, dir_L: int = 4,
                  w0: float = 30.0, hidden_w0: float = 1.0, sigma_mul: float = 10.0,
                  rgb_mul: float = 1.0, return_acts: bool = False):
    """Siren MLP (models.py:185-203): sine layers sin(w0*(Wx+b)) on raw xyz,
    density = relu(linear)*sigma_mul, rgb = sigmoid(linear(sin-layer([remap, de]))*rgb_mul)."""
    acts = {}
    h = points
    for i in range(num_layers):
        w = w0 if i == 0 else hidden_w0
        h = torch.sin(w * _lin(h, p, f"block1.{i}.layer"))
        acts[f"h{i}"] = h
    sigma = _lin(h, p, "density_branch.0")
    acts["z_sigma"] = sigma
    density = (torch.relu(sigma) * sigma_mul).squeeze(-1)
    feat = _lin(h, p, "feature_remap.0")
    acts["f"] = feat
    de = positional_encoding(dirs, dir_L)
    acts["de"] = de
    g = torch.sin(hidden_w0 * _lin(torch.cat([feat, de], dim=-1), p, "rgb_head.0.layer"))
    acts["g"] = g
    colors = torch.sigmoid(_lin(g, p, "rgb_head.1") * rgb_mul)
    if return_acts:
        return colors, density, acts
    return colors, density


def relu_kink_distance(acts) -> Tensor:
    """Per sample, the smallest |pre-activation| over every ReLU input ("z*" entries of the
    activations dict).  A sample whose value is within fp32 rounding of zero has an
    implementation-dependent ReLU decision, hence an implementation-dependent gradient: tests
    that compare gradients element for element drop such samples."""
    zs = [v.detach().abs().min(dim=1).values for k, v in acts.items() if k.startswith("z")]
    return torch.stack(zs, dim=0).min(dim=0).values


# ----------------------------------------------------------------------------
# compositing (nerf/rendering.py:110-153)
# ----------------------------------------------------------------------------

def exclusive_transmittance(betas: Tensor) -> Tensor:
    """Inclusive cumprod along dim 1 shifted right by one with leading ones
    (rendering.py:120-122); no epsilon is added."""
    inc = torch.cumprod(betas, dim=1)
    return torch.cat([torch.ones_like(inc[:, :1]), inc[:, :-1]], dim=1)


def composite_weights(densities: Tensor, deltas: Tensor) -> Tuple[Tensor, Tensor]:
    """alpha = 1-exp(-sigma*delta); w = T_excl(1-alpha)*alpha (rendering.py:143-146)."""
    alpha = 1 - torch.exp(-densities * deltas[None, :])
    w = exclusive_transmittance(1 - alpha) * alpha
    return w, alpha


def composite(colors: Tensor, densities: Tensor, deltas: Tensor, white_background: bool) -> Tensor:
    """rgb = sum_s w*c (+ 1 - sum_s w on white) (rendering.py:148-153)."""
    w, _ = composite_weights(densities, deltas)
    rgb = (w[..., None] * colors).sum(dim=1)
    if white_background:
        rgb = rgb + (1 - w.sum(dim=1, keepdim=True))
    return rgb


def composite_depth(weights: Tensor, t: Tensor) -> Tensor:
    """Expected depth sum_s w*t.  PARITY UNPINNED: the reference never computes depth."""
    return (weights * t[None, :]).sum(dim=1)


def composite_backward_manual(colors: Tensor, densities: Tensor, deltas: Tensor,
                              white_background: bool, grad_rgb: Tensor) -> Tuple[Tensor, Tensor]:
    """Closed-form gradient of `composite` wrt (colors, densities), written as
    the zero-safe reverse recurrence the HIP kernel uses:
        R[S-1] = 0;  R[j] = dT[j+1] + beta[j+1]*R[j+1];  dbeta[j] = T[j]*R[j].
    Used only to cross-check that recurrence against autograd of `composite`.
    """
    c64, s64, d64, g64 = (x.double() for x in (colors, densities, deltas, grad_rgb))
    e = torch.exp(-s64 * d64[None, :])
    alpha = 1 - e
    beta = 1 - alpha
    T = exclusive_transmittance(beta)
    w = T * alpha
    G = (g64[:, None, :] * c64).sum(-1)
    if white_background:
        G = G - g64.sum(-1, keepdim=True)
    dcol = g64[:, None, :] * w[..., None]
    dT = G * alpha
    R = torch.zeros_like(T)
    S = T.shape[1]
    for j in range(S - 2, -1, -1):
        R[:, j] = dT[:, j + 1] + beta[:, j + 1] * R[:, j + 1]
    dalpha = G * T - T * R
    dsig = dalpha * d64[None, :] * e
    return dcol.float(), dsig.float()


# ----------------------------------------------------------------------------
# renderer (nerf/rendering.py:156-226)
# ----------------------------------------------------------------------------

def render(model_fn, rays_o: Tensor, rays_d: Tensor, near: float, far: float, num_samples: int,
           us: Sequence[Tensor], white_background: bool = True, chunk_size: int = 8192,
           return_extras: bool = False):
    """Chunked render loop of render_nerf.  `us[i]` is the uniform draw the
    reference would take for chunk i (one torch.rand(S) per chunk, rendering.py:26
    reached through :195).  model_fn(points_normalised[M,3], dirs[M,3]) -> (colors, density)."""
    outs, extras = [], []
    for ci, i in enumerate(range(0, rays_o.shape[0], chunk_size)):
        o, d = rays_o[i:i + chunk_size], rays_d[i:i + chunk_size]
        t = stratified_t(near, far, num_samples, us[ci])
        deltas = sample_deltas(t)
        pos = sample_positions(o, d, t).reshape(-1, 3)
        dirs = d[:, None, :].expand(-1, num_samples, -1).reshape(-1, 3)
        colors, dens = model_fn(normalize_positions(pos, near, far), dirs)
        colors = colors.reshape(o.shape[0], num_samples, 3)
        dens = dens.reshape(o.shape[0], num_samples)
        outs.append(composite(colors, dens, deltas, white_background))
        if return_extras:
            w, _ = composite_weights(dens, deltas)
            extras.append((w, composite_depth(w, t)))
    rgb = torch.cat(outs, dim=0)
    if return_extras:
        return rgb, torch.cat([e[0] for e in extras]), torch.cat([e[1] for e in extras])
    return rgb


# ----------------------------------------------------------------------------
# hierarchical sampling -- PARITY UNPINNED (absent from the reference)
# ----------------------------------------------------------------------------

def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tuple[Tensor, Tensor]:
    """Inverse-CDF sampling as published with the original NeRF (Mildenhall et al.
    2020, run_nerf_helpers.sample_pdf): pdf = (w+1e-5)/sum, cdf = [0, cumsum(pdf)],
    idx = searchsorted(cdf, u, right=True), linear interpolation inside the bin.

    bins [R,B+1], weights [R,B], u [R,N] -> (samples [R,N], idx [R,N] int64 = the
    `right=True` insertion index, which the HIP kernel must reproduce bit-exactly).
    The reference repository has no such function; nothing pins this but itself.
    """
    w = weights + 1e-5
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    idx = torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True)
    below = (idx - 1).clamp(min=0)
    above = idx.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0), idx


def render_hierarchical(coarse_fn, fine_fn, rays_o: Tensor, rays_d: Tensor, near: float, far: float,
                        num_coarse: int, num_fine: int, u_coarse: Tensor, u_fine: Tensor,
                        white_background: bool = True):
    """Coarse + fine render of the published NeRF (run_nerf.render_rays), restated on top of
    this file's single-pass pieces.  PARITY UNPINNED: the reference has no such path.
    u_coarse [Nc], u_fine [R,Nf].  Returns rgb_coarse, rgb_fine, weights_fine, t_all."""
    R = rays_o.shape[0]
    t_c = stratified_t(near, far, num_coarse, u_coarse)

    def run(fn, t_rays, deltas_rays):
        S = t_rays.shape[1]
        pos = rays_o[:, None, :] + t_rays[:, :, None] * rays_d[:, None, :]
        dirs = rays_d[:, None, :].expand(-1, S, -1).reshape(-1, 3)
        col, den = fn(normalize_positions(pos.reshape(-1, 3), near, far), dirs)
        col, den = col.reshape(R, S, 3), den.reshape(R, S)
        alpha = 1 - torch.exp(-den * deltas_rays)
        w = exclusive_transmittance(1 - alpha) * alpha
        rgb = (w[..., None] * col).sum(1)
        if white_background:
            rgb = rgb + (1 - w.sum(1, keepdim=True))
        return rgb, w

    d_c = sample_deltas(t_c)
    rgb_c, w_c = run(coarse_fn, t_c[None].expand(R, -1), d_c[None].expand(R, -1))
    t_mid = 0.5 * (t_c[1:] + t_c[:-1])
    t_f, _ = sample_pdf(t_mid[None].expand(R, -1).contiguous(), w_c[:, 1:-1].detach(), u_fine)
    t_all, _ = torch.sort(torch.cat([t_c[None].expand(R, -1), t_f.detach()], dim=1), dim=1)
    d_all = torch.cat([t_all[:, 1:] - t_all[:, :-1], torch.full((R, 1), 1e10)], dim=1)
    rgb_f, w_f = run(fine_fn, t_all, d_all)
    return rgb_c, rgb_f, w_f, t_all


# ----------------------------------------------------------------------------
# train step (train.py:123-131, 166-183)
# ----------------------------------------------------------------------------

def lr_lambda(step: int, lr: float, lr_decay: float, lr_decay_factor: float, lr_min: float) -> float:
    """LambdaLR factor of train.py:127-131."""
    gamma = lr_decay_factor ** (1 / (lr_decay * 1000))
    return max(gamma ** step, lr_min / lr)


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam single-tensor update (train.py:123,182), in place;
    `step` is 1-based."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def compute_rays_np(H: int, W: int, c2w: np.ndarray, focal: float):
    """Pinhole rays of nerf/data.py:87-112 for one pose: x=u-W/2, y=-(v-H/2), z=-f,
    rotate by c2w[:3,:3], L2-normalise; origin = c2w[:3,3].  (data.py cannot be
    imported here -- imageio is absent -- so this row is pinned by reading it only.)"""
    u = np.arange(W, dtype=np.float32)
    v = np.arange(H, dtype=np.float32)
    ug, vg = np.meshgrid(u, v, indexing="xy")
    dirs = np.stack([ug - 0.5 * W, -(vg - 0.5 * H), -np.full_like(ug, focal)], axis=-1)
    rd = np.einsum("ij,hwj->hwi", c2w[:3, :3], dirs)
    rd = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    ro = np.tile(c2w[:3, 3][None, None, :], (H, W, 1))
    return ro.reshape(-1, 3).astype(np.float32), rd.reshape(-1, 3).astype(np.float32)


def pose_spherical(theta_deg: float, phi_deg: float, radius: float) -> np.ndarray:
    """Camera-to-world of eval.py:14-41 (translate z, rotate x by phi, rotate y by
    theta, axis swap)."""
    t = np.eye(4, dtype=np.float32); t[2, 3] = radius
    ph = phi_deg / 180.0 * np.pi
    rx = np.eye(4, dtype=np.float32)
    rx[1, 1] = rx[2, 2] = np.cos(ph); rx[1, 2] = -np.sin(ph); rx[2, 1] = np.sin(ph)
    th = theta_deg / 180.0 * np.pi
    ry = np.eye(4, dtype=np.float32)
    ry[0, 0] = ry[2, 2] = np.cos(th); ry[0, 2] = -np.sin(th); ry[2, 0] = np.sin(th)
    swap = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    return swap @ (ry @ (rx @ t))
