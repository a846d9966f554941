"""ctypes binding of libnerf_hip.so (C ABI declared in include/nerf_hip.h).

There is no CPU fallback: importing this module without the built library, or calling
an op with a tensor that is not a contiguous fp32 tensor on a HIP device, raises.
PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NERF_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libnerf_hip.so")

MODEL_NERF = 0
MODEL_SIREN = 1
NUM_PARAM_TENSORS = 24
ABI_VERSION = 10

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f"{LIB_PATH} is missing: build it with `make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` "
        "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")

lib = ctypes.CDLL(LIB_PATH)

_P = c_void_p
_SIGS = {
    "nerf_abi_version": (c_int, []),
    "nerf_last_error": (c_char_p, []),
    "nerf_stratified_t": (c_int, [c_double, c_double, c_int, _P, _P, _P, _P]),
    "nerf_sample_positions": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_double, c_double, _P, _P]),
    "nerf_compute_rays": (c_int, [_P, c_int64, c_int, c_int, c_double, _P, _P, _P]),
    "nerf_exclusive_cumprod": (c_int, [_P, c_int64, c_int, _P, _P]),
    "nerf_exclusive_cumprod_bwd": (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    "nerf_posenc_fwd": (c_int, [_P, c_int64, c_int, c_int, _P, _P]),
    "nerf_posenc_bwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "nerf_composite_fwd": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "nerf_composite_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P]),
    "nerf_mlp_param_floats": (c_size_t, [c_int]),
    "nerf_mlp_packed_floats": (c_size_t, [c_int]),
    "nerf_mlp_fwd_workspace_floats": (c_size_t, [c_int, c_int64]),
    "nerf_mlp_bwd_workspace_floats": (c_size_t, [c_int, c_int64]),
    "nerf_mlp_pack": (c_int, [c_int, ctypes.POINTER(c_void_p), _P, _P]),
    "nerf_mlp_fwd": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, _P, _P, _P, _P]),
    "nerf_mlp_packed_bf16_bytes": (c_size_t, [c_int]),
    "nerf_mlp_pack_bf16": (c_int, [c_int, ctypes.POINTER(c_void_p), _P, _P]),
    "nerf_mlp_fwd_bf16": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, _P, _P, _P]),
    "nerf_mlp_packed_bf16x3_bytes": (c_size_t, [c_int]),
    "nerf_mlp_pack_bf16x3": (c_int, [c_int, ctypes.POINTER(c_void_p), _P, _P]),
    "nerf_mlp_fwd_bf16x3": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, _P, _P, _P]),
    "nerf_mlp_packed_x3_floats": (c_size_t, [c_int]),
    "nerf_mlp_pack_x3": (c_int, [c_int, ctypes.POINTER(c_void_p), _P, _P]),
    "nerf_mlp_fwd_x3": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, c_int, c_int, c_double, c_double, _P, _P, _P, _P]),
    "nerf_mlp_bwd_data_x3": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "nerf_mlp_bwd_weights_x3": (c_int, [c_int, c_int64, _P, _P, _P, c_int, _P]),
    "nerf_linear_bwd_workspace_floats": (c_size_t, [c_int64, c_int, c_int]),
    "nerf_linear_fwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_int, c_int, c_double, c_double, _P, c_int64, _P, _P]),
    "nerf_linear_bwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_double, c_double, _P, c_int64, _P, c_int64, c_int,
                                c_int, _P, _P, c_int64, _P, _P, _P]),
    "nerf_linear_bwd_noalloc": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, c_double, c_double, _P, c_int64, _P, c_int64,
                                        c_int, c_int, _P, c_int64, _P, _P, _P]),
    "nerf_merge_samples": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "nerf_mlp_bwd": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "nerf_mlp_bwd_data": (c_int, [c_int, _P, c_int64, _P, _P, _P, _P, _P, _P]),
    "nerf_mlp_bwd_weights": (c_int, [c_int, c_int64, _P, _P, _P, c_int, _P]),
    "nerf_adam_step": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_double, c_double, c_double, c_double, c_double, _P]),
    "nerf_weights_to_cdf": (c_int, [_P, c_int64, c_int, _P, _P]),
    "nerf_sample_pdf": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "nerf_resample": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P]),
}
EXPORTS = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

if lib.nerf_abi_version() != ABI_VERSION:
    raise RuntimeError(f"libnerf_hip.so ABI {lib.nerf_abi_version()} != expected {ABI_VERSION}")


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib.nerf_last_error().decode()}")


def _dev(t: torch.Tensor, name: str, dtype=torch.float32) -> int:
    """Device pointer of a tensor after validating what the kernels assume."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; this build runs on a HIP device only (no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str) -> Optional[int]:
    return None if t is None else _dev(t, name)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimers:
    """Optional HIP-event brackets around named launches, recorded on the stream the kernels
    run on (the current torch stream).  Used by bench.py for per-kernel roofline numbers;
    disabled (zero cost) by default."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    def start(self, name, work=0):
        if not self.enabled:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if torch.cuda.current_stream() != torch.cuda.default_stream():
            name += "@side"   # launched on a side stream: runs concurrently with the main stream's kernels
        self.events.setdefault(name, []).append((a, b, work))
        return b

    @staticmethod
    def stop(ev):
        if ev is not None:
            ev.record()

    def summary(self):
        """name -> (launches, total ms, total work units); call after a device synchronize."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v), sum(w for _, _, w in v))
                for k, v in self.events.items() if v}

    def reset(self):
        self.events = {}


timers = KernelTimers()


# --------------------------------------------------------------------------- sampling
def stratified_t(near: float, far: float, S: int, u: torch.Tensor, want_deltas: bool = True):
    if u.shape != (S,):
        raise ValueError(f"u must have shape ({S},), got {tuple(u.shape)}")
    t = torch.empty_like(u)
    deltas = torch.empty_like(u) if want_deltas else None
    _check(lib.nerf_stratified_t(float(near), float(far), S, _dev(u, "u"), _dev(t, "t"), _opt(deltas, "deltas"),
                                 _stream()), "nerf_stratified_t")
    return t, deltas


def sample_positions(rays_o: torch.Tensor, rays_d: torch.Tensor, t: torch.Tensor, normalize: bool, near: float,
                     far: float) -> torch.Tensor:
    R, S = rays_o.shape[0], t.shape[0]
    if rays_o.shape != (R, 3) or rays_d.shape != (R, 3):
        raise ValueError("rays_o and rays_d must be [R,3]")
    pos = torch.empty((R, S, 3), dtype=torch.float32, device=rays_o.device)
    _check(lib.nerf_sample_positions(_dev(rays_o, "rays_o"), _dev(rays_d, "rays_d"), _dev(t, "t"), R, S,
                                     int(normalize), float(near), float(far), _dev(pos, "pos"), _stream()),
           "nerf_sample_positions")
    return pos


def compute_rays(c2w: torch.Tensor, H: int, W: int, focal: float):
    """c2w [N,4,4] -> (rays_o, rays_d) [N, H*W, 3] on the device of c2w."""
    if c2w.dim() != 3 or c2w.shape[1:] != (4, 4):
        raise ValueError("c2w must be [N,4,4]")
    N = c2w.shape[0]
    ro = torch.empty((N, H * W, 3), dtype=torch.float32, device=c2w.device)
    rd = torch.empty_like(ro)
    _check(lib.nerf_compute_rays(_dev(c2w, "c2w"), N, H, W, float(focal), _dev(ro, "rays_o"), _dev(rd, "rays_d"),
                                 _stream()), "nerf_compute_rays")
    return ro, rd


def exclusive_cumprod(betas: torch.Tensor) -> torch.Tensor:
    R, S = betas.shape
    out = torch.empty_like(betas)
    _check(lib.nerf_exclusive_cumprod(_dev(betas, "betas"), R, S, _dev(out, "out"), _stream()), "nerf_exclusive_cumprod")
    return out


def exclusive_cumprod_bwd(betas: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    R, S = betas.shape
    if grad_out.shape != betas.shape:
        raise ValueError("exclusive_cumprod_bwd: grad_out must have the shape of betas")
    dbetas = torch.empty_like(betas)
    _check(lib.nerf_exclusive_cumprod_bwd(_dev(betas, "betas"), _dev(grad_out, "grad_out"), R, S, _dev(dbetas, "dbetas"),
                                          _stream()), "nerf_exclusive_cumprod_bwd")
    return dbetas


def posenc(x: torch.Tensor, L: int) -> torch.Tensor:
    if x.dim() != 2:
        raise ValueError("positional_encoding expects [M,D]")
    M, D = x.shape
    out = torch.empty((M, D * (1 + 2 * L)), dtype=torch.float32, device=x.device)
    _check(lib.nerf_posenc_fwd(_dev(x, "x"), M, D, L, _dev(out, "out"), _stream()), "nerf_posenc_fwd")
    return out


def posenc_bwd(x: torch.Tensor, grad_out: torch.Tensor, L: int) -> torch.Tensor:
    M, D = x.shape
    if grad_out.shape != (M, D * (1 + 2 * L)):
        raise ValueError("posenc_bwd: grad_out must be [M, D*(1+2L)]")
    dx = torch.empty_like(x)
    _check(lib.nerf_posenc_bwd(_dev(x, "x"), _dev(grad_out, "grad_out"), M, D, L, _dev(dx, "dx"), _stream()),
           "nerf_posenc_bwd")
    return dx


# --------------------------------------------------------------------------- compositing
def composite_fwd(colors, densities, deltas, white: bool, t=None, want_weights=False, want_depth=False,
                  want_acc=False):
    """colors=None: weights only (rgb comes back as None; want_weights is implied)."""
    R, S = densities.shape
    per_ray = deltas.dim() == 2
    if (colors is not None and colors.shape != (R, S, 3)) or deltas.shape != ((R, S) if per_ray else (S,)) or \
            (t is not None and t.shape != deltas.shape):
        raise ValueError(f"composite: colors {None if colors is None else tuple(colors.shape)} densities "
                         f"{tuple(densities.shape)} deltas {tuple(deltas.shape)} do not agree")
    dev = densities.device
    want_weights = want_weights or colors is None
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev) if colors is not None else None
    weights = torch.empty((R, S), dtype=torch.float32, device=dev) if want_weights else None
    depth = torch.empty((R,), dtype=torch.float32, device=dev) if want_depth else None
    acc = torch.empty((R,), dtype=torch.float32, device=dev) if want_acc else None
    if want_depth and t is None:
        raise ValueError("depth output needs t")
    _check(lib.nerf_composite_fwd(_opt(colors, "colors"), _dev(densities, "densities"), _dev(deltas, "deltas"),
                                  _opt(t, "t"), R, S, int(per_ray), int(bool(white)), _opt(rgb, "rgb"), _opt(weights, "weights"),
                                  _opt(depth, "depth"), _opt(acc, "acc"), _stream()), "nerf_composite_fwd")
    return rgb, weights, depth, acc


def composite_bwd(colors, densities, deltas, white: bool, grad_rgb):
    R, S = densities.shape
    dcolors = torch.empty_like(colors)
    ddens = torch.empty_like(densities)
    _check(lib.nerf_composite_bwd(_dev(colors, "colors"), _dev(densities, "densities"), _dev(deltas, "deltas"), R, S,
                                  int(deltas.dim() == 2), int(bool(white)), _dev(grad_rgb, "grad_rgb"), _dev(dcolors, "dcolors"),
                                  _dev(ddens, "ddensities"), _stream()), "nerf_composite_bwd")
    return dcolors, ddens


# --------------------------------------------------------------------------- MLP
def param_floats(model: int) -> int:
    return lib.nerf_mlp_param_floats(model)


def packed_floats(model: int) -> int:
    return lib.nerf_mlp_packed_floats(model)


def mlp_pack(model: int, params: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if len(params) != NUM_PARAM_TENSORS:
        raise ValueError(f"expected {NUM_PARAM_TENSORS} parameter tensors, got {len(params)}")
    dev = params[0].device
    if out is None:
        out = torch.empty(packed_floats(model), dtype=torch.float32, device=dev)
    ptrs = (c_void_p * NUM_PARAM_TENSORS)(*[_dev(p, f"params[{i}]") for i, p in enumerate(params)])
    _check(lib.nerf_mlp_pack(model, ptrs, _dev(out, "packed"), _stream()), "nerf_mlp_pack")
    return out


def mlp_pack_x3(model: int, params: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight image of the opt-in bf16x3 TRAINING mode: the fp32 image of mlp_pack followed by the three-term bf16
    images of every dense layer that the mode's forward and data-gradient kernels stream.  mlp_fwd / mlp_bwd recognise it by its size; a pass without a
    workspace (inference) reads the fp32 part alone."""
    if len(params) != NUM_PARAM_TENSORS:
        raise ValueError(f"expected {NUM_PARAM_TENSORS} parameter tensors, got {len(params)}")
    n = lib.nerf_mlp_packed_x3_floats(model)
    if n == 0:
        raise NotImplementedError("this model has no bf16x3 training kernels")
    if out is None or out.numel() != n:
        out = torch.empty(n, dtype=torch.float32, device=params[0].device)
    ptrs = (c_void_p * NUM_PARAM_TENSORS)(*[_dev(p, f"params[{i}]") for i, p in enumerate(params)])
    _check(lib.nerf_mlp_pack_x3(model, ptrs, _dev(out, "packed"), _stream()), "nerf_mlp_pack_x3")
    return out


def is_x3_image(model: int, packed: torch.Tensor) -> bool:
    n = lib.nerf_mlp_packed_x3_floats(model)
    return n != 0 and packed.dtype == torch.float32 and packed.numel() == n


def mlp_fwd(model: int, packed: torch.Tensor, M: int, *, points=None, dirs=None, rays_o=None, rays_d=None, t=None,
            S: int = 0, near: float = 0.0, far: float = 1.0, train: bool = False, want_colors: bool = True):
    """want_colors=False (inference only): the pass stops after the density row / branch; colors come back as None."""
    dev = packed.device
    if not want_colors and train:
        raise ValueError("the density-only pass is an inference variant")
    colors = torch.empty((M, 3), dtype=torch.float32, device=dev) if want_colors else None
    density = torch.empty((M,), dtype=torch.float32, device=dev)
    ws = None
    if train:
        ws = torch.empty(lib.nerf_mlp_fwd_workspace_floats(model, M), dtype=torch.float32, device=dev)
    x3 = train and is_x3_image(model, packed)     # the opt-in bf16x3 training kernels; inference reads the fp32 part
    ev = timers.start(("mlp_fwd_train_x3" if x3 else "mlp_fwd_train") if train else ("mlp_fwd_infer" if want_colors else "mlp_fwd_infer_density"), M)
    _check((lib.nerf_mlp_fwd_x3 if x3 else lib.nerf_mlp_fwd)(
        model, _dev(packed, "packed"), M, _opt(points, "points"), _opt(dirs, "dirs"),
        _opt(rays_o, "rays_o"), _opt(rays_d, "rays_d"), _opt(t, "t"), S,
        int(t is not None and t.dim() == 2), float(near), float(far),
        _opt(colors, "colors"), _dev(density, "density"), _opt(ws, "workspace"), _stream()),
        "nerf_mlp_fwd_x3" if x3 else "nerf_mlp_fwd")
    timers.stop(ev)
    return colors, density, ws


def _pack_terms(terms: int, model: int, params: Sequence[torch.Tensor], out: Optional[torch.Tensor]) -> torch.Tensor:
    name = "bf16" if terms == 1 else "bf16x3"
    if len(params) != NUM_PARAM_TENSORS:
        raise ValueError(f"expected {NUM_PARAM_TENSORS} parameter tensors, got {len(params)}")
    nbytes = getattr(lib, f"nerf_mlp_packed_{name}_bytes")(model)
    if nbytes == 0:
        raise NotImplementedError(f"the {name} inference mode covers the NeRF model only")
    if out is None or out.numel() != nbytes:
        out = torch.empty(nbytes, dtype=torch.uint8, device=params[0].device)
    ptrs = (c_void_p * NUM_PARAM_TENSORS)(*[_dev(p, f"params[{i}]") for i, p in enumerate(params)])
    _check(getattr(lib, f"nerf_mlp_pack_{name}")(model, ptrs, _dev(out, "packed", torch.uint8), _stream()), f"nerf_mlp_pack_{name}")
    return out


def mlp_pack_bf16(model: int, params: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 weight image (a uint8 tensor) for mlp_fwd_bf16."""
    return _pack_terms(1, model, params, out)


def mlp_pack_bf16x3(model: int, params: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Three-term bf16 weight image (a uint8 tensor, three times the bf16 one) for mlp_fwd_bf16."""
    return _pack_terms(3, model, params, out)


def mlp_fwd_bf16(model: int, packed: torch.Tensor, M: int, *, points=None, dirs=None, rays_o=None, rays_d=None, t=None,
                 S: int = 0, near: float = 0.0, far: float = 1.0, want_colors: bool = True):
    """Inference forward on the bf16 matrix pipe; the image's size says which of the two modes it was packed for.
    want_colors=False: the pass stops after the layer that holds the density row; colors come back as None."""
    dev = packed.device
    if packed.numel() == lib.nerf_mlp_packed_bf16_bytes(model):
        name = "bf16"
    elif packed.numel() == lib.nerf_mlp_packed_bf16x3_bytes(model):
        name = "bf16x3"
    else:
        raise ValueError(f"packed image of {packed.numel()} bytes is neither a bf16 nor a bf16x3 image of model {model}")
    colors = torch.empty((M, 3), dtype=torch.float32, device=dev) if want_colors else None
    density = torch.empty((M,), dtype=torch.float32, device=dev)
    ev = timers.start(f"mlp_fwd_infer_{name}" + ("" if want_colors else "_density"), M)
    _check(getattr(lib, f"nerf_mlp_fwd_{name}")(model, _dev(packed, "packed", torch.uint8), M, _opt(points, "points"),
                                               _opt(dirs, "dirs"), _opt(rays_o, "rays_o"), _opt(rays_d, "rays_d"), _opt(t, "t"), S,
                                               int(t is not None and t.dim() == 2), float(near), float(far),
                                               _opt(colors, "colors"), _dev(density, "density"), _stream()), f"nerf_mlp_fwd_{name}")
    timers.stop(ev)
    return colors, density, None


def mlp_bwd(model: int, packed: torch.Tensor, M: int, dcolors, ddensity, colors, ws, grads: Optional[torch.Tensor] = None,
            accumulate: bool = False):
    """Flat parameter gradient (state_dict order).  With `grads` given and accumulate=True the result is
    added to that buffer (the optimizer's bucket) instead of a fresh one."""
    dev = packed.device
    if grads is None:
        grads = torch.empty(param_floats(model), dtype=torch.float32, device=dev)
    elif grads.dim() != 1 or grads.numel() != param_floats(model):
        raise ValueError(f"gradient buffer must be a flat tensor of {param_floats(model)} floats, got {tuple(grads.shape)}")
    bw = torch.empty(lib.nerf_mlp_bwd_workspace_floats(model, M), dtype=torch.float32, device=dev)
    x3 = is_x3_image(model, packed)   # the workspace then comes from the bf16x3 forward (its mask words are laid out differently)
    ev = timers.start("mlp_bwd_data_x3" if x3 else "mlp_bwd_data", M)
    _check((lib.nerf_mlp_bwd_data_x3 if x3 else lib.nerf_mlp_bwd_data)(
        model, _dev(packed, "packed"), M, _dev(dcolors, "dcolors"), _dev(ddensity, "ddensity"),
        _dev(colors, "colors"), _dev(ws, "fwd_workspace"), _dev(bw, "bwd_workspace"),
        _stream()), "nerf_mlp_bwd_data_x3" if x3 else "nerf_mlp_bwd_data")
    timers.stop(ev)
    ev = timers.start("mlp_bwd_weights_x3" if x3 else "mlp_bwd_weights", M)
    _check((lib.nerf_mlp_bwd_weights_x3 if x3 else lib.nerf_mlp_bwd_weights)(
        model, M, _dev(ws, "fwd_workspace"), _dev(bw, "bwd_workspace"),
        _dev(grads, "grads"), int(bool(accumulate)), _stream()), "nerf_mlp_bwd_weights_x3" if x3 else "nerf_mlp_bwd_weights")
    timers.stop(ev)
    return grads


ACT_NONE, ACT_RELU, ACT_SIN, ACT_SIGMOID = 0, 1, 2, 3


def linear_fwd(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], act: int, pre: float, post: float):
    """y = post*f(pre*(x W^T + b)) for contiguous x [M,K], W [N,K]; returns (y, aux)."""
    M, K = x.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise ValueError(f"linear: x is [*,{K}] but W is {tuple(W.shape)}")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    aux = torch.empty((M, N), dtype=torch.float32, device=x.device) if act == ACT_SIN else None
    ev = timers.start("linear_fwd", M)
    _check(lib.nerf_linear_fwd(_dev(x, "x"), K, _dev(W, "W"), _opt(b, "b"), M, K, N, act, float(pre), float(post),
                               _dev(y, "y"), N, _opt(aux, "aux"), _stream()), "nerf_linear_fwd")
    timers.stop(ev)
    return y, aux


def linear_bwd(dy, y, aux, act: int, pre: float, post: float, x, W, need_dx: bool, need_db: bool):
    M, K = x.shape
    N = W.shape[0]
    dev = x.device
    ws = torch.empty(lib.nerf_linear_bwd_workspace_floats(M, K, N), dtype=torch.float32, device=dev)
    dx = torch.empty((M, K), dtype=torch.float32, device=dev) if need_dx else None
    dW = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev) if need_db else None
    ev = timers.start("linear_bwd", M)
    _check(lib.nerf_linear_bwd(_dev(dy, "dy"), N, _opt(y, "y"), _opt(aux, "aux"), N, act, float(pre), float(post),
                               _dev(x, "x"), K, _dev(W, "W"), M, K, N, _dev(ws, "workspace"), _opt(dx, "dx"), K,
                               _dev(dW, "dW"), _opt(db, "db"), _stream()), "nerf_linear_bwd")
    timers.stop(ev)
    return dx, dW, db


NOALLOC_SCRATCH_FLOATS = 8 << 20


def linear_bwd_noalloc(dy, y, aux, act: int, pre: float, post: float, x, W, dx, dW, db) -> None:
    """nerf_linear_bwd_noalloc: the caller brings every output (dx / db may be None); the library's built-in
    scratch replaces the workspace, so nothing is allocated between entry and return."""
    M, K = x.shape
    N = W.shape[0]
    _check(lib.nerf_linear_bwd_noalloc(_dev(dy, "dy"), N, _opt(y, "y"), _opt(aux, "aux"), N, act, float(pre), float(post),
                                       _dev(x, "x"), K, _dev(W, "W"), M, K, N, _opt(dx, "dx"), K, _dev(dW, "dW"),
                                       _opt(db, "db"), _stream()), "nerf_linear_bwd_noalloc")


def adam_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0) -> None:
    n = p.numel()
    if not (g.numel() == m.numel() == v.numel() == n):
        raise ValueError("adam_step: p, g, m, v must have the same number of elements")
    _check(lib.nerf_adam_step(_dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"), n, step, lr, beta1, beta2, eps,
                              grad_scale, _stream()), "nerf_adam_step")


# --------------------------------------------------------------------------- hierarchical sampling
def weights_to_cdf(weights: torch.Tensor) -> torch.Tensor:
    R, B = weights.shape
    cdf = torch.empty((R, B + 1), dtype=torch.float32, device=weights.device)
    _check(lib.nerf_weights_to_cdf(_dev(weights, "weights"), R, B, _dev(cdf, "cdf"), _stream()), "nerf_weights_to_cdf")
    return cdf


def merge_samples(t_coarse: torch.Tensor, t_fine: torch.Tensor):
    """Sorted union of shared coarse depths [Sc] and per-ray fine depths [R,Sf] -> (t_all, deltas) [R,Sc+Sf]."""
    R, Sf = t_fine.shape
    Sc = t_coarse.shape[0]
    t_all = torch.empty((R, Sc + Sf), dtype=torch.float32, device=t_fine.device)
    deltas = torch.empty_like(t_all)
    _check(lib.nerf_merge_samples(_dev(t_coarse, "t_coarse"), _dev(t_fine, "t_fine"), R, Sc, Sf, _dev(t_all, "t_all"),
                                  _dev(deltas, "deltas"), _stream()), "nerf_merge_samples")
    return t_all, deltas


def sample_pdf(bins: torch.Tensor, cdf: torch.Tensor, u: torch.Tensor):
    R, B1 = bins.shape
    if cdf.shape != (R, B1) or u.shape[0] != R:
        raise ValueError("sample_pdf: bins/cdf must be [R,B+1] and u [R,N]")
    N = u.shape[1]
    samples = torch.empty((R, N), dtype=torch.float32, device=bins.device)
    idx = torch.empty((R, N), dtype=torch.int32, device=bins.device)
    _check(lib.nerf_sample_pdf(_dev(bins, "bins"), _dev(cdf, "cdf"), _dev(u, "u"), R, B1 - 1, N, _dev(samples, "samples"),
                               _dev(idx, "idx", torch.int32), _stream()), "nerf_sample_pdf")
    return samples, idx


def resample(weights: torch.Tensor, t_coarse: torch.Tensor, u: torch.Tensor, want_fine: bool = False):
    """Coarse weights [R,Sc], shared coarse depths [Sc], draws [R,Sf] -> (t_all, deltas) [R,Sc+Sf] (+ t_fine [R,Sf],
    idx [R,Sf] with want_fine): CDF, inverse-CDF draws and sorted union in one launch."""
    R, Sc = weights.shape
    if t_coarse.shape != (Sc,) or u.dim() != 2 or u.shape[0] != R:
        raise ValueError("resample: weights [R,Sc], t_coarse [Sc], u [R,Sf]")
    Sf = u.shape[1]
    t_all = torch.empty((R, Sc + Sf), dtype=torch.float32, device=weights.device)
    deltas = torch.empty_like(t_all)
    t_fine = torch.empty((R, Sf), dtype=torch.float32, device=weights.device) if want_fine else None
    idx = torch.empty((R, Sf), dtype=torch.int32, device=weights.device) if want_fine else None
    _check(lib.nerf_resample(_dev(weights, "weights"), _dev(t_coarse, "t_coarse"), _dev(u, "u"), R, Sc, Sf, _dev(t_all, "t_all"),
                             _dev(deltas, "deltas"), _opt(t_fine, "t_fine"),
                             None if idx is None else _dev(idx, "idx", torch.int32), _stream()), "nerf_resample")
    return (t_all, deltas, t_fine, idx) if want_fine else (t_all, deltas)
