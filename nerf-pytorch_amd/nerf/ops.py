"""Autograd bridges between PyTorch and the HIP kernels (no arithmetic here)."""
from __future__ import annotations

from typing import Sequence

import torch

from . import _hip


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 + contiguous view of a tensor the kernels can take (expand() views and
    slices with a stride are copied; chunk slices along dim 0 are already contiguous)."""
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


class PackedWeights:
    """Caches the fragment-ordered weight image of one module and re-packs it whenever
    a parameter was replaced or modified in place (optimizer step, load_state_dict)."""

    def __init__(self, model_id: int, pack=_hip.mlp_pack):
        self.model_id = model_id
        self._pack = pack      # _hip.mlp_pack (fp32 image) or _hip.mlp_pack_bf16
        self._key = None
        self._img = None

    def invalidate(self) -> None:
        self._key = None

    def get(self, params: Sequence[torch.Tensor]) -> torch.Tensor:
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key or self._img is None or self._img.device != params[0].device:
            datas = [_f32c(p.detach()) for p in params]
            reuse = self._img is not None and self._img.device == params[0].device
            self._img = self._pack(self.model_id, datas, self._img if reuse else None)
            self._key = key
        return self._img


def wants_grad(params: Sequence[torch.Tensor]) -> bool:
    """Training mode = autograd is recording and some parameter asks for a gradient.  Decided by
    the caller: inside autograd.Function.forward grad mode is always off and needs_input_grad
    ignores torch.no_grad()."""
    return torch.is_grad_enabled() and any(p.requires_grad for p in params)


def _mlp_fwd(model_id, packed, M, train, **inputs):
    """The fp32 kernel, or the opt-in bf16 inference kernel when handed a bf16 image (uint8)."""
    if packed.dtype == torch.uint8:
        if train:
            raise RuntimeError("the bf16 mode is inference-only; training runs the fp32 kernels")
        return _hip.mlp_fwd_bf16(model_id, packed, M, **inputs)
    return _hip.mlp_fwd(model_id, packed, M, train=train, **inputs)


def grad_sink(module, params: Sequence[torch.Tensor]):
    """The module's flat gradient bucket (set by parallel.FlatParameters) if the parameters' .grad are
    still views of it; the backward kernels then add into the bucket directly and autograd is handed
    no per-tensor gradients (no 24 accumulation kernels, no temporary)."""
    sink = getattr(module, "_grad_sink", None)
    if sink is None or not params:
        return None
    g = params[0].grad
    if g is None or g.data_ptr() != sink.data_ptr() or sink.device != params[0].device:
        return None
    return sink


def _check_sink(sink, model_id) -> None:
    if sink is not None and (not isinstance(sink, torch.Tensor) or sink.dim() != 1
                             or sink.numel() != _hip.param_floats(model_id) or sink.dtype != torch.float32):
        raise ValueError("gradient sink must be None or the flat fp32 gradient buffer of the whole model")


def _param_grads(ctx, flat):
    if ctx.sink_used:
        return (None,) * ctx.nparams
    grads, off = [], 0
    for shp in ctx.shapes:
        n = 1
        for s in shp:
            n *= s
        grads.append(flat[off:off + n].view(shp))
        off += n
    return tuple(grads)


class MLPFunction(torch.autograd.Function):
    """(colors, density) = model(points, dirs) through the fused encode+MLP kernel;
    backward yields the 24 parameter gradients (flat buffer, state_dict order)."""

    @staticmethod
    def forward(ctx, model_id, packed, points, dirs, train, sink, *params):
        M = points.shape[0]
        colors, density, ws = _mlp_fwd(model_id, packed, M, train, points=points, dirs=dirs)
        ctx.model_id, ctx.M, ctx.train = model_id, M, train
        ctx.nparams = len(params)
        _check_sink(sink, model_id)
        ctx.sink, ctx.sink_used = sink, sink is not None
        if train:
            ctx.save_for_backward(packed, colors, ws)
            ctx.shapes = [p.shape for p in params]
        return colors, density

    @staticmethod
    def backward(ctx, dcolors, ddensity):
        if not ctx.train:
            raise RuntimeError("backward through an inference-mode forward")
        packed, colors, ws = ctx.saved_tensors
        if dcolors is None:
            dcolors = torch.zeros_like(colors)
        if ddensity is None:
            ddensity = torch.zeros((ctx.M,), dtype=torch.float32, device=colors.device)
        flat = _hip.mlp_bwd(ctx.model_id, packed, ctx.M, _f32c(dcolors), _f32c(ddensity), colors, ws, ctx.sink,
                            ctx.sink_used)
        return (None,) * 6 + _param_grads(ctx, flat)


class LinearActFunction(torch.autograd.Function):
    """y = post * f(pre * (x W^T + b)): one nn.Linear + activation of an architecture that has no fused
    kernel, forward and backward on the layer-wise HIP kernels."""

    @staticmethod
    def forward(ctx, x, W, b, act, pre, post):
        x, Wc = _f32c(x), _f32c(W)
        bc = _f32c(b) if b is not None else None
        if x.shape[0] == 0:
            ctx.empty = True
            return x.new_zeros((0, W.shape[0]))
        y, aux = _hip.linear_fwd(x, Wc, bc, act, pre, post)
        ctx.empty = False
        ctx.cfg = (act, float(pre), float(post), b is not None)
        ctx.save_for_backward(x, Wc, y, aux if aux is not None else y.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.empty:
            raise RuntimeError("backward through an empty batch")
        x, W, y, aux = ctx.saved_tensors
        act, pre, post, has_b = ctx.cfg
        need_dx = ctx.needs_input_grad[0]
        dx, dW, db = _hip.linear_bwd(_f32c(dy), y, aux if aux.numel() else None, act, pre, post, x, W, need_dx,
                                     has_b and ctx.needs_input_grad[2])
        return dx, dW, db, None, None, None


class RenderFunction(torch.autograd.Function):
    """rgb = composite(model(sample(rays, t))) for one ray chunk: positions are generated
    inside the MLP kernel, colors/densities go straight to the compositing kernel."""

    @staticmethod
    def forward(ctx, model_id, packed, rays_o, rays_d, t, deltas, near, far, white, want_extras, train, sink, *params):
        R, S = rays_o.shape[0], t.shape[-1]   # t is [S] (shared) or [R,S] (per ray)
        M = R * S
        colors, density, ws = _mlp_fwd(model_id, packed, M, train, rays_o=rays_o, rays_d=rays_d, t=t, S=S, near=near,
                                       far=far)
        colors = colors.view(R, S, 3)
        density = density.view(R, S)
        rgb, weights, depth, acc = _hip.composite_fwd(colors, density, deltas, white, t=t, want_weights=want_extras,
                                                      want_depth=want_extras, want_acc=want_extras)
        ctx.model_id, ctx.M, ctx.train, ctx.white = model_id, M, train, bool(white)
        ctx.nparams = len(params)
        _check_sink(sink, model_id)
        ctx.sink, ctx.sink_used = sink, sink is not None
        if train:
            ctx.save_for_backward(packed, colors, density, deltas, ws)
            ctx.shapes = [p.shape for p in params]
        if want_extras:
            ctx.mark_non_differentiable(weights, depth, acc)
            return rgb, weights, depth, acc
        return rgb

    @staticmethod
    def backward(ctx, grad_rgb, *unused):
        if not ctx.train:
            raise RuntimeError("backward through an inference-mode forward")
        packed, colors, density, deltas, ws = ctx.saved_tensors
        dcolors, ddens = _hip.composite_bwd(colors, density, deltas, ctx.white, _f32c(grad_rgb))
        flat = _hip.mlp_bwd(ctx.model_id, packed, ctx.M, dcolors.view(-1, 3), ddens.view(-1), colors.view(-1, 3), ws,
                            ctx.sink, ctx.sink_used)
        return (None,) * 12 + _param_grads(ctx, flat)


class CompositeFunction(torch.autograd.Function):
    """composite_volume for arbitrary upstream models (colors/densities carry grad)."""

    @staticmethod
    def forward(ctx, colors, densities, deltas, white):
        colors, densities, deltas = _f32c(colors), _f32c(densities), _f32c(deltas)
        rgb, _, _, _ = _hip.composite_fwd(colors, densities, deltas, white)
        ctx.save_for_backward(colors, densities, deltas)
        ctx.white = bool(white)
        return rgb

    @staticmethod
    def backward(ctx, grad_rgb):
        colors, densities, deltas = ctx.saved_tensors
        dcolors, ddens = _hip.composite_bwd(colors, densities, deltas, ctx.white, _f32c(grad_rgb))
        return dcolors, ddens, None, None


class ExclusiveCumprodFunction(torch.autograd.Function):
    """T[:, s] = prod_{j<s} betas[:, j] with the gradient autograd derives for the reference's
    cumprod + shift (rendering.py:120-122), exact for betas that contain zeros."""

    @staticmethod
    def forward(ctx, betas):
        betas = _f32c(betas)
        ctx.save_for_backward(betas)
        return _hip.exclusive_cumprod(betas)

    @staticmethod
    def backward(ctx, grad_out):
        (betas,) = ctx.saved_tensors
        return _hip.exclusive_cumprod_bwd(betas, _f32c(grad_out))
