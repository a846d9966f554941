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

from . import _hip, ops
from .ops import CompositeFunction, ExclusiveCumprodFunction, RenderFunction, _f32c, grad_sink, wants_grad


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
    (reference rendering.py:110-122); one ray per wavefront.  Differentiable like the reference's
    cumprod + shift: the backward is a zero-safe suffix scan (rows of up to 1024 samples).  Inside the
    renderer the transmittance and its gradient live in the compositing kernels."""
    if betas.dim() != 2:
        raise ValueError("compute_accumulated_transmittance expects betas [R,S]")
    if torch.is_grad_enabled() and betas.requires_grad:
        return ExclusiveCumprodFunction.apply(betas)
    return _hip.exclusive_cumprod(_f32c(betas.detach()))


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
    """This package's NeRF/Siren with the architecture the fused kernels are built for; anything else
    (other constructor arguments, foreign modules) goes through model.forward + the compositing kernel."""
    return hasattr(model, "model_id") and hasattr(model, "packed_weights") and model.fused_supported()


def render_nerf(model: torch.nn.Module, rays_o: Tensor, rays_d: Tensor, near: float, far: float,
                num_samples: int = 256, device: str = 'cpu', white_background: bool = True,
                chunk_size: int = 8192, *, u: Optional[Sequence[Tensor]] = None,
                return_extras: bool = False):
    """Render rays to RGB [N,3] (reference rendering.py:156-226).

    Per chunk of `chunk_size` rays: one uniform draw -> t, deltas; then for this package's NeRF /
    Siren modules a single fused path (positions generated inside the encode+MLP kernel,
    compositing kernel on its outputs), or for any other nn.Module the reference's sequence
    sample -> flatten -> model.forward -> composite with HIP sampling/compositing kernels.

    Rays that require grad take the second route as well, with positions built from differentiable tensor
    operations, so that d rgb / d rays_o and d rays_d exist as they do in the reference.

    Keyword-only extras: `u` = list of per-chunk uniform draws (replaces torch.rand, for parity
    tests and for data-parallel ranks that must share the draw); `return_extras` = also return
    (weights [N,S], depth [N], acc [N]).
    """
    rays_o = _f32c(rays_o.to(device))
    rays_d = _f32c(rays_d.to(device))
    if not rays_o.is_cuda:
        raise RuntimeError("render_nerf: this build has no CPU path; pass device='cuda'")
    if rays_o.shape[0] == 0:   # nothing to render (the reference would fail in torch.cat of no chunks)
        empty = rays_o.new_zeros((0, 3))
        if return_extras:
            return empty, rays_o.new_zeros((0, num_samples)), rays_o.new_zeros((0,)), rays_o.new_zeros((0,))
        return empty
    # Gradients with respect to the rays (the reference's render_nerf is transparent to autograd in rays_o / rays_d,
    # rendering.py:59-62,223; none of its callers asks for them): positions from differentiable tensor operations, the
    # model through its layer-wise kernels (which have input gradients), the compositing kernel on top.
    rays_need_grad = torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)
    fused = _fused_model(model) and not rays_need_grad
    if fused:
        model._check_supported()
        params = model.ordered_parameters()
        train = wants_grad(params)
        packed = model.packed_weights(train)
        sink = grad_sink(model, params) if train else None
    outs, extras = [], []
    for ci, i in enumerate(range(0, rays_o.shape[0], chunk_size)):
        o_chunk = rays_o[i:i + chunk_size]
        d_chunk = rays_d[i:i + chunk_size]
        t, deltas = _t_and_deltas(near, far, num_samples, rays_o.device, None if u is None else u[ci].to(rays_o.device))
        if fused:
            res = RenderFunction.apply(model.model_id, packed, o_chunk, d_chunk, t, deltas, float(near), float(far),
                                       bool(white_background), bool(return_extras), train, sink, *params)
            if return_extras:
                outs.append(res[0])
                extras.append(res[1:])
            else:
                outs.append(res)
        else:
            if rays_need_grad:   # o + t*d as the reference forms it: one multiplication, one addition per element
                pos = (o_chunk.unsqueeze(1) + t.view(1, -1, 1) * d_chunk.unsqueeze(1)).reshape(-1, 3)
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


def _disjoint(a: Optional[Tensor], b: Optional[Tensor]) -> bool:
    """Two gradient sinks (flat fp32 buffers, or None) occupy different memory."""
    if a is None or b is None:
        return True
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a0 + a.numel() * a.element_size() <= b0 or b0 + b.numel() * b.element_size() <= a0


def render_nerf_hierarchical(coarse: torch.nn.Module, fine: torch.nn.Module, rays_o: Tensor, rays_d: Tensor,
                             near: float, far: float, num_coarse: int = 64, num_fine: int = 128,
                             device: str = 'cuda', white_background: bool = True, chunk_size: int = 8192, *,
                             u: Optional[Sequence[Tensor]] = None, u_fine: Optional[Sequence[Tensor]] = None,
                             return_extras: bool = False, coarse_rgb: bool = True):
    """Coarse + fine rendering with inverse-CDF re-sampling.  NOT in the reference (its renderer is
    single pass, rendering.py:156-226); this follows the published NeRF procedure: a stratified
    coarse pass, a per-ray PDF from the interior coarse weights over the mid-points of the coarse
    depths, `num_fine` inverse-CDF samples per ray (no gradient through the sampling), and a fine pass
    over the sorted union (num_coarse + num_fine evaluations per ray).

    Returns (rgb_coarse [N,3], rgb_fine [N,3]) -- train on the sum of both MSEs -- plus, with
    return_extras, the fine pass's (weights [N,S], depth [N], acc [N]).  `u[ci]` [num_coarse] and
    `u_fine[ci]` [rays_in_chunk, num_fine] override the random draws of chunk ci.

    coarse_rgb=False (renders without gradients): the coarse image is not needed, so the coarse pass stops at the
    densities (fp32 kernels: 16 % of that pass's matrix work is skipped) and rgb_coarse comes back as None; the
    re-sampling weights, hence the fine image, are bitwise the same.
    """
    rays_o = _f32c(rays_o.to(device))
    rays_d = _f32c(rays_d.to(device))
    if not rays_o.is_cuda:
        raise RuntimeError("render_nerf_hierarchical: this build has no CPU path; pass device='cuda'")
    if num_coarse < 3:
        raise ValueError("num_coarse must be at least 3 (the PDF is built from interior coarse samples)")
    for m in (coarse, fine):
        if not _fused_model(m):
            raise TypeError("render_nerf_hierarchical needs this package's NeRF/Siren modules with default arguments")
        m._check_supported()
    pc, pf = coarse.ordered_parameters(), fine.ordered_parameters()
    tc_train, tf_train = wants_grad(pc), wants_grad(pf)
    wc, wf = coarse.packed_weights(tc_train), fine.packed_weights(tf_train)
    sink_c = grad_sink(coarse, pc) if tc_train else None
    sink_f = grad_sink(fine, pf) if tf_train else None
    density_only = not coarse_rgb and not tc_train and not tf_train
    if rays_o.shape[0] == 0:   # nothing to render (a rank whose slice of the rays is empty)
        empty = rays_o.new_zeros((0, 3))
        res = (None if density_only else empty, rays_o.new_zeros((0, 3)))
        if return_extras:
            res += (rays_o.new_zeros((0, num_coarse + num_fine)), rays_o.new_zeros((0,)), rays_o.new_zeros((0,)))
        return res
    # Two backward chains may only run side by side when they add into different memory: one module used as both
    # networks (or two modules sharing a bucket slice) would have two streams read-modify-write the same gradients.
    overlap = ops.OVERLAP_BACKWARD and tc_train and tf_train and coarse is not fine and _disjoint(sink_c, sink_f)
    out_c, out_f, extras = [], [], []
    for ci, i in enumerate(range(0, rays_o.shape[0], chunk_size)):
        o, d = rays_o[i:i + chunk_size], rays_d[i:i + chunk_size]
        R = o.shape[0]
        t_c, deltas_c = _t_and_deltas(near, far, num_coarse, rays_o.device, None if u is None else u[ci].to(rays_o.device))
        if overlap:
            # the coarse pass goes to a side stream so that autograd runs its backward there, concurrently with
            # the fine network's backward; in the forward direction the fine pass waits for it as before
            main, side = torch.cuda.current_stream(rays_o.device), ops.side_stream(rays_o.device)
            side.wait_stream(main)
            for x in (o, d, t_c, deltas_c, wc):     # main-stream memory that side-stream kernels read, now and in backward
                x.record_stream(side)
            with torch.cuda.stream(side):
                rgb_c, w_c, _, _ = RenderFunction.apply(coarse.model_id, wc, o, d, t_c, deltas_c, float(near), float(far),
                                                        bool(white_background), True, tc_train, sink_c, *pc)
            main.wait_stream(side)
            rgb_c.record_stream(main)
            w_c.record_stream(main)
        elif density_only:
            # no gradients, no coarse image: densities from the shortened pass, compositing weights only
            fwd = _hip.mlp_fwd_bf16 if wc.dtype == torch.uint8 else _hip.mlp_fwd     # (a bf16 / bf16x3 image is uint8)
            _, dens_c, _ = fwd(coarse.model_id, wc, R * num_coarse, rays_o=o, rays_d=d, t=t_c, S=num_coarse,
                               near=near, far=far, want_colors=False)
            rgb_c = None
            w_c = _hip.composite_fwd(None, dens_c.view(R, num_coarse), deltas_c, bool(white_background))[1]
        else:
            rgb_c, w_c, _, _ = RenderFunction.apply(coarse.model_id, wc, o, d, t_c, deltas_c, float(near), float(far),
                                                    bool(white_background), True, tc_train, sink_c, *pc)
        with torch.no_grad():
            # PDF of the interior coarse weights over the mid-points of the coarse depths, num_fine inverse-CDF
            # draws per ray, sorted union with the coarse depths: one launch (nerf_resample = nerf_weights_to_cdf
            # -> nerf_sample_pdf -> nerf_merge_samples, bitwise)
            uf = torch.rand(R, num_fine, device=rays_o.device) if u_fine is None else _f32c(u_fine[ci].to(rays_o.device))
            t_all, deltas_all = _hip.resample(w_c.detach(), t_c, uf)
        res = RenderFunction.apply(fine.model_id, wf, o, d, t_all, deltas_all, float(near), float(far),
                                   bool(white_background), bool(return_extras), tf_train, sink_f, *pf)
        out_c.append(rgb_c)
        if return_extras:
            out_f.append(res[0])
            extras.append(res[1:])
        else:
            out_f.append(res)
    rgb_c = None if density_only else torch.cat(out_c, dim=0)
    rgb_f = torch.cat(out_f, dim=0)
    if return_extras:
        return (rgb_c, rgb_f) + tuple(torch.cat([e[k] for e in extras], dim=0) for k in range(3))
    return rgb_c, rgb_f
