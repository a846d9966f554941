"""The HIP kernels as PyTorch custom operators (torch.library, namespace `nerf`) with their autograd
formulas and shape-only fake implementations, so that the dispatcher, torch.compile and opcheck see
them; no arithmetic happens here.  Operators:

    nerf::posenc(x, L)                                   nerf/encoding.py:4-20 (+ nerf::posenc_backward)
    nerf::composite(colors, densities, deltas, white)    nerf/rendering.py:125-153 (+ nerf::composite_backward)
    nerf::mlp(model, packed, points, dirs, params, ...)  nerf/models.py:62-75 / :185-203 (+ nerf::mlp_backward)
    nerf::render(model, packed, rays_o, rays_d, t, ...)  one ray chunk of nerf/rendering.py:191-223: sample ->
                                                         encode -> MLP -> composite (+ nerf::render_backward)

`MLPFunction`, `RenderFunction` and `CompositeFunction` keep the call forms the rest of the package uses.
"""
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

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
        self._pack = pack      # _hip.mlp_pack (fp32 image), _hip.mlp_pack_bf16 or _hip.mlp_pack_bf16x3
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
    """The fp32 kernel, or an opt-in bf16 / bf16x3 inference kernel when handed such an image (uint8)."""
    if packed.dtype == torch.uint8:
        if train:
            raise RuntimeError("the bf16 modes are inference-only; training runs the fp32 kernels")
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


# A coarse + fine pair has two independent backward chains (each network's data- and weight-gradient
# kernels).  Run on two streams they overlap on the device: the tail of one launch, where compute units run
# out of tiles, is filled with the other chain's workgroups.  The coarse pass of render_nerf_hierarchical is
# therefore issued on a side stream; autograd runs its backward there too.
_side_streams = {}
OVERLAP_BACKWARD = os.environ.get("NERF_OVERLAP", "1") != "0"


def side_stream(device) -> "torch.cuda.Stream":
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _side_streams:
        _side_streams[idx] = torch.cuda.Stream(device=idx)
    return _side_streams[idx]


def _join_after_backward(device) -> None:
    """Called inside a backward that wrote into a gradient sink (which autograd does not track).  If it ran on the
    side stream, make the stream that called backward() wait for it once the whole backward pass is queued: callbacks
    queued on the autograd engine run at the end of the pass with the caller's streams current."""
    idx = torch.device(device).index
    side = _side_streams.get(idx if idx is not None else torch.cuda.current_device())
    if side is not None and torch.cuda.current_stream(device) == side:
        done = torch.cuda.Event()
        done.record(side)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream(device).wait_event(done))


def _split(flat: Tensor, shapes) -> List[Tensor]:
    """Per-tensor views of a flat gradient buffer (state_dict order)."""
    grads, off = [], 0
    for shp in shapes:
        n = 1
        for s in shp:
            n *= int(s)
        grads.append(flat[off:off + n].view(shp))
        off += n
    return grads


def _none_tensor(like: Tensor) -> Tensor:
    """Stand-in for an output the call does not produce (operators cannot return None)."""
    return like.new_empty((0,), dtype=torch.float32)


def _opaque(ws, like: Tensor) -> Tensor:
    """The forward workspace as an operator output: raw 32-bit words (it mixes activations with ReLU mask
    bits, which are not numbers), empty for an inference call."""
    return ws.view(torch.int32) if ws is not None else like.new_empty((0,), dtype=torch.int32)


# ------------------------------------------------------------------------------------------- posenc
@custom_op("nerf::posenc", mutates_args=())
def posenc_op(x: Tensor, L: int) -> Tensor:
    return _hip.posenc(_f32c(x), int(L))


@posenc_op.register_fake
def _(x, L):
    return x.new_empty((x.shape[0], x.shape[1] * (1 + 2 * L)))


@custom_op("nerf::posenc_backward", mutates_args=())
def posenc_backward_op(x: Tensor, grad_out: Tensor, L: int) -> Tensor:
    return _hip.posenc_bwd(_f32c(x), _f32c(grad_out), int(L))


@posenc_backward_op.register_fake
def _(x, grad_out, L):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


def _posenc_setup(ctx, inputs, output):
    x, L = inputs
    ctx.save_for_backward(x)
    ctx.L = int(L)


def _posenc_backward(ctx, grad_out):
    (x,) = ctx.saved_tensors
    return posenc_backward_op(x, grad_out, ctx.L), None


posenc_op.register_autograd(_posenc_backward, setup_context=_posenc_setup)


# ---------------------------------------------------------------------------------------- composite
@custom_op("nerf::composite", mutates_args=())
def composite_op(colors: Tensor, densities: Tensor, deltas: Tensor, white: bool) -> Tensor:
    rgb, _, _, _ = _hip.composite_fwd(_f32c(colors), _f32c(densities), _f32c(deltas), white)
    return rgb


@composite_op.register_fake
def _(colors, densities, deltas, white):
    return densities.new_empty((densities.shape[0], 3))


@custom_op("nerf::composite_backward", mutates_args=())
def composite_backward_op(colors: Tensor, densities: Tensor, deltas: Tensor, white: bool,
                          grad_rgb: Tensor) -> Tuple[Tensor, Tensor]:
    return _hip.composite_bwd(_f32c(colors), _f32c(densities), _f32c(deltas), white, _f32c(grad_rgb))


@composite_backward_op.register_fake
def _(colors, densities, deltas, white, grad_rgb):
    return torch.empty_like(colors, memory_format=torch.contiguous_format), \
        torch.empty_like(densities, memory_format=torch.contiguous_format)


def _composite_setup(ctx, inputs, output):
    colors, densities, deltas, white = inputs
    ctx.save_for_backward(colors, densities, deltas)
    ctx.white = bool(white)


def _composite_backward(ctx, grad_rgb):
    colors, densities, deltas = ctx.saved_tensors
    dcolors, ddens = composite_backward_op(colors, densities, deltas, ctx.white, grad_rgb)
    return dcolors, ddens, None, None


composite_op.register_autograd(_composite_backward, setup_context=_composite_setup)


# ---------------------------------------------------------------------------------------------- MLP
def _ws_floats(model: int, M: int, train: bool) -> int:
    return int(_hip.lib.nerf_mlp_fwd_workspace_floats(model, M)) if train else 0


@custom_op("nerf::mlp", mutates_args=())
def mlp_op(model: int, packed: Tensor, points: Tensor, dirs: Tensor, params: Sequence[Tensor], train: bool,
           sink: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """(colors [M,3], density [M], workspace) = model(points, dirs) through the fused encode + MLP kernel.
    `packed` is the weight image of `params` (which are inputs only so that autograd routes the 24 parameter
    gradients); `sink`, when given, is the flat gradient bucket the backward adds into instead."""
    _check_sink(sink, model)
    colors, density, ws = _mlp_fwd(model, packed, points.shape[0], train, points=_f32c(points), dirs=_f32c(dirs))
    return colors, density, _opaque(ws, colors)


@mlp_op.register_fake
def _(model, packed, points, dirs, params, train, sink):
    M = points.shape[0]
    return (points.new_empty((M, 3)), points.new_empty((M,)),
            points.new_empty((_ws_floats(model, M, train),), dtype=torch.int32))


@custom_op("nerf::mlp_backward", mutates_args=("sink",))
def mlp_backward_op(model: int, packed: Tensor, dcolors: Tensor, ddensity: Tensor, colors: Tensor, workspace: Tensor,
                    sink: Optional[Tensor]) -> Tensor:
    """Flat parameter gradient (state_dict order) of nerf::mlp; with `sink` it is ADDED to that buffer and an
    empty tensor is returned."""
    flat = _hip.mlp_bwd(model, packed, colors.shape[0], _f32c(dcolors), _f32c(ddensity), colors,
                        workspace.view(torch.float32), sink, sink is not None)
    return _none_tensor(colors) if sink is not None else flat


@mlp_backward_op.register_fake
def _(model, packed, dcolors, ddensity, colors, workspace, sink):
    return colors.new_empty((0 if sink is not None else _hip.param_floats(model),))


def _mlp_setup(ctx, inputs, output):
    model, packed, points, dirs, params, train, sink = inputs
    colors, density, ws = output
    ctx.model, ctx.train, ctx.sink = model, bool(train), sink
    ctx.shapes = [p.shape for p in params]
    ctx.set_materialize_grads(False)   # no zero tensors for the outputs nobody differentiates (the workspace!)
    ctx.save_for_backward(packed, colors, ws)


def _mlp_backward(ctx, dcolors, ddensity, dws):
    if not ctx.train:
        raise RuntimeError("backward through an inference-mode forward")
    packed, colors, ws = ctx.saved_tensors
    if dcolors is None and ddensity is None:
        return None, None, None, None, [None] * len(ctx.shapes), None, None
    if dcolors is None:
        dcolors = torch.zeros_like(colors)
    if ddensity is None:
        ddensity = colors.new_zeros((colors.shape[0],))
    flat = mlp_backward_op(ctx.model, packed, dcolors, ddensity, colors, ws, ctx.sink)
    grads = [None] * len(ctx.shapes) if ctx.sink is not None else _split(flat, ctx.shapes)
    return None, None, None, None, grads, None, None


mlp_op.register_autograd(_mlp_backward, setup_context=_mlp_setup)


class MLPFunction:
    """(colors, density) = model(points, dirs); call form kept from the autograd.Function this replaced."""

    @staticmethod
    def apply(model_id, packed, points, dirs, train, sink, *params):
        colors, density, _ = mlp_op(model_id, packed, points, dirs, list(params), bool(train), sink)
        return colors, density


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


# ------------------------------------------------------------------------------------------- render
@custom_op("nerf::render", mutates_args=())
def render_op(model: int, packed: Tensor, rays_o: Tensor, rays_d: Tensor, t: Tensor, deltas: Tensor, near: float,
              far: float, white: bool, extras: bool, params: Sequence[Tensor], train: bool,
              sink: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """One ray chunk: rgb = composite(model(sample(rays, t))).  Positions are generated inside the MLP kernel,
    colors / densities go straight to the compositing kernel.  t, deltas: [S] (shared) or [R,S] (per ray).
    Returns rgb [R,3], weights [R,S], depth [R], acc [R] (empty unless `extras`), and what the backward needs:
    colors [R,S,3], density [R,S], workspace (empty unless `train`)."""
    _check_sink(sink, model)
    R, S = rays_o.shape[0], t.shape[-1]
    t, deltas = _f32c(t), _f32c(deltas)
    colors, density, ws = _mlp_fwd(model, packed, R * S, train, rays_o=_f32c(rays_o), rays_d=_f32c(rays_d), t=t, S=S,
                                   near=near, far=far)
    colors, density = colors.view(R, S, 3), density.view(R, S)
    rgb, weights, depth, acc = _hip.composite_fwd(colors, density, deltas, white, t=t, want_weights=extras,
                                                  want_depth=extras, want_acc=extras)
    none = _none_tensor(rgb)
    return (rgb, weights if extras else none, depth if extras else _none_tensor(rgb), acc if extras else _none_tensor(rgb),
            colors, density, _opaque(ws, rgb))


@render_op.register_fake
def _(model, packed, rays_o, rays_d, t, deltas, near, far, white, extras, params, train, sink):
    R, S = rays_o.shape[0], t.shape[-1]
    e = rays_o.new_empty
    return (e((R, 3)), e((R, S) if extras else (0,)), e((R,) if extras else (0,)), e((R,) if extras else (0,)),
            e((R, S, 3)), e((R, S)), e((_ws_floats(model, R * S, train),), dtype=torch.int32))


@custom_op("nerf::render_backward", mutates_args=("sink",))
def render_backward_op(model: int, packed: Tensor, colors: Tensor, density: Tensor, deltas: Tensor, white: bool,
                       grad_rgb: Tensor, workspace: Tensor, sink: Optional[Tensor]) -> Tensor:
    """Flat parameter gradient of nerf::render: compositing backward, then the MLP backward; with `sink` the
    result is ADDED to that buffer and an empty tensor is returned."""
    dcolors, ddens = _hip.composite_bwd(colors, density, deltas, white, _f32c(grad_rgb))
    flat = _hip.mlp_bwd(model, packed, density.numel(), dcolors.view(-1, 3), ddens.view(-1), colors.view(-1, 3),
                        workspace.view(torch.float32), sink, sink is not None)
    return _none_tensor(colors) if sink is not None else flat


@render_backward_op.register_fake
def _(model, packed, colors, density, deltas, white, grad_rgb, workspace, sink):
    return colors.new_empty((0 if sink is not None else _hip.param_floats(model),))


def _render_setup(ctx, inputs, output):
    model, packed, rays_o, rays_d, t, deltas, near, far, white, extras, params, train, sink = inputs
    rgb, weights, depth, acc, colors, density, ws = output
    ctx.model, ctx.train, ctx.sink, ctx.white = model, bool(train), sink, bool(white)
    ctx.shapes = [p.shape for p in params]
    ctx.set_materialize_grads(False)   # no zero tensors for the outputs nobody differentiates (the workspace!)
    ctx.save_for_backward(packed, colors, density, _f32c(deltas), ws)


def _render_backward(ctx, grad_rgb, *unused):
    if not ctx.train:
        raise RuntimeError("backward through an inference-mode forward")
    packed, colors, density, deltas, ws = ctx.saved_tensors
    if grad_rgb is None:   # only the extras were used downstream: they carry no gradient
        return (None,) * 10 + ([None] * len(ctx.shapes), None, None)
    flat = render_backward_op(ctx.model, packed, colors, density, deltas, ctx.white, grad_rgb, ws, ctx.sink)
    if ctx.sink is not None:
        _join_after_backward(colors.device)
    grads = [None] * len(ctx.shapes) if ctx.sink is not None else _split(flat, ctx.shapes)
    return (None,) * 10 + (grads, None, None)


render_op.register_autograd(_render_backward, setup_context=_render_setup)


class RenderFunction:
    """rgb (or rgb, weights, depth, acc with want_extras) of one ray chunk; call form kept from the
    autograd.Function this replaced.  Only rgb carries a gradient."""

    @staticmethod
    def apply(model_id, packed, rays_o, rays_d, t, deltas, near, far, white, want_extras, train, sink, *params):
        out = render_op(model_id, packed, rays_o, rays_d, t, deltas, float(near), float(far), bool(white),
                        bool(want_extras), list(params), bool(train), sink)
        if want_extras:
            return out[0], out[1].detach(), out[2].detach(), out[3].detach()
        return out[0]


class CompositeFunction:
    """composite_volume for arbitrary upstream models (colors / densities carry grad)."""

    @staticmethod
    def apply(colors, densities, deltas, white):
        return composite_op(colors, densities, deltas, bool(white))


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
