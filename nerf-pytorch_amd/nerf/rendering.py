"""Volume renderer on HIP kernels (drop-in for the reference's nerf/rendering.py).

Every public function of the reference file exists here with the same name, argument
order and return value; extras are keyword-only with reference-preserving defaults.
render_nerf draws exactly one torch.rand(num_samples) per ray chunk, as the reference
does (rendering.py:26 reached through :195), so RNG consumption is unchanged.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _hip
from .ops import CompositeFunction, RenderFunction, _f32c, wants_grad


def stratified_sampling(near: float, far: float, num_bins: int, device: str = 'cpu', *,
                        u: Optional[Tensor] = None) -> Tensor:
    """t[s] = lo[s] + (hi[s]-lo[s])*u[s] over num_bins equal bins of [near, far]
    (reference rendering.py:6-27).  `u` overrides the torch.rand draw (parity tests)."""
    if u is None:
        u = torch.rand(num_bins, device=device)
    t, _ = _hip.stratified_t(near, far, num_bins, _f32c(u), want_deltas=False)
    return t


def _t_and_deltas(near, far, num_samples, device, u):
    if u is None:
        u = torch.rand(num_samples, device=device)
    return _hip.stratified_t(near, far, num_samples, _f32c(u), want_deltas=True)


def generate_sample_positions(rays_o_batch: Tensor, rays_d_batch: Tensor, near: float, far: float,
                              num_samples: int, device: str = 'cpu', *,
                              u: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(positions [R,S,3], deltas [S]) with one t-vector shared by all rays and a trailing
    delta of 1e10 (reference rendering.py:30-64)."""
    t, deltas = _t_and_deltas(near, far, num_samples, device, u)
    pos = _hip.sample_positions(_f32c(rays_o_batch), _f32c(rays_d_batch), t, False, near, far)
    return pos, deltas


def normalize_positions(positions: Tensor, near: float, far: float) -> Tensor:
    """2*(p-near)/(far-near)-1 (reference rendering.py:67-82).  Elementwise glue on
    whatever device `positions` lives on; the fused path normalises inside the MLP kernel."""
    return 2 * (positions - near) / (far - near) - 1


def query_model(model: torch.nn.Module, sample_positions_flat: Tensor, directions_flat: Tensor, near: float,
                far: float) -> Tuple[Tensor, Tensor]:
    """Normalise positions, then model.forward(positions, directions) (reference rendering.py:85-107)."""
    return model.forward(normalize_positions(sample_positions_flat, near, far), directions_flat)


def compute_accumulated_transmittance(betas: Tensor) -> Tensor:
    """Exclusive cumulative product along dim 1: T[:,0] = 1, T[:,s] = prod_{j<s} betas[:,j]
    (reference rendering.py:110-122); one ray per wavefront, forward only (inside the renderer
    the transmittance and its gradient live in the compositing kernels)."""
    if betas.requires_grad:
        raise NotImplementedError("use composite_volume for a differentiable transmittance")
    return _hip.exclusive_cumprod(_f32c(betas))


def composite_volume(colors: Tensor, densities: Tensor, deltas: Tensor, white_background: bool) -> Tensor:
    """Alpha compositing (reference rendering.py:125-153): colors [R,S,3], densities [R,S],
    deltas [S] -> rgb [R,3]; differentiable wrt colors and densities."""
    return CompositeFunction.apply(colors, densities, deltas, bool(white_background))


def composite_outputs(colors: Tensor, densities: Tensor, deltas: Tensor, t: Tensor, white_background: bool):
    """rgb, weights [R,S], depth [R] (= sum w*t), acc [R] (= sum w) in one kernel; forward only.
    weights match compute_accumulated_transmittance(1-alpha)*alpha of the reference
    (rendering.py:146); depth and acc have no counterpart there."""
    return _hip.composite_fwd(_f32c(colors), _f32c(densities), _f32c(deltas), bool(white_background), t=_f32c(t),
                              want_weights=True, want_depth=True, want_acc=True)


def _fused_model(model) -> bool:
    return hasattr(model, "model_id") and hasattr(model, "packed_weights")


def render_nerf(model: torch.nn.Module, rays_o: Tensor, rays_d: Tensor, near: float, far: float,
                num_samples: int = 256, device: str = 'cpu', white_background: bool = True,
                chunk_size: int = 8192, *, u: Optional[Sequence[Tensor]] = None,
                return_extras: bool = False):
    """Render rays to RGB [N,3] (reference rendering.py:156-226).

    Per chunk of `chunk_size` rays: one uniform draw -> t, deltas; then for this package's NeRF /
    Siren modules a single fused path (positions generated inside the encode+MLP kernel,
    compositing kernel on its outputs), or for any other nn.Module the reference's sequence
    sample -> flatten -> model.forward -> composite with HIP sampling/compositing kernels.

    Keyword-only extras: `u` = list of per-chunk uniform draws (replaces torch.rand, for parity
    tests and for data-parallel ranks that must share the draw); `return_extras` = also return
    (weights [N,S], depth [N], acc [N]).
    """
    rays_o = _f32c(rays_o.to(device))
    rays_d = _f32c(rays_d.to(device))
    if not rays_o.is_cuda:
        raise RuntimeError("render_nerf: this build has no CPU path; pass device='cuda'")
    fused = _fused_model(model)
    if fused:
        model._check_supported()
        params = model.ordered_parameters()
        packed = model.packed_weights()
        train = wants_grad(params)
    outs, extras = [], []
    for ci, i in enumerate(range(0, rays_o.shape[0], chunk_size)):
        o_chunk = rays_o[i:i + chunk_size]
        d_chunk = rays_d[i:i + chunk_size]
        t, deltas = _t_and_deltas(near, far, num_samples, rays_o.device, None if u is None else u[ci].to(rays_o.device))
        if fused:
            res = RenderFunction.apply(model.model_id, packed, o_chunk, d_chunk, t, deltas, float(near), float(far),
                                       bool(white_background), bool(return_extras), train, *params)
            if return_extras:
                outs.append(res[0])
                extras.append(res[1:])
            else:
                outs.append(res)
        else:
            pos = _hip.sample_positions(o_chunk, d_chunk, t, False, near, far).reshape(-1, 3)
            dirs = d_chunk.unsqueeze(1).expand(-1, num_samples, -1).reshape(-1, 3)
            colors, dens = query_model(model, pos, dirs, near, far)
            colors = colors.reshape(o_chunk.shape[0], num_samples, 3)
            dens = dens.reshape(o_chunk.shape[0], num_samples)
            outs.append(composite_volume(colors, dens, deltas, white_background))
            if return_extras:
                extras.append(composite_outputs(colors.detach(), dens.detach(), deltas, t, white_background)[1:])
    rgb = torch.cat(outs, dim=0)
    if return_extras:
        return (rgb,) + tuple(torch.cat([e[k] for e in extras], dim=0) for k in range(3))
    return rgb
