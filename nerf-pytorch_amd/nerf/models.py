"""NeRF and Siren modules whose forward/backward run as fused HIP kernels.

Drop-in for the reference's nerf/models.py: same class names, constructor arguments,
sub-module names (hence identical state_dict keys and shapes, so the reference's
checkpoints load and torch.optim.Adam sees the same 24 tensors), same
forward(points, rays_d) -> (colors[M,3], density[M]).  The architecture the reference's
callers build (train.py:101-103, all-default arguments) runs on the fused kernels; any other
constructor arguments run layer by layer on the generic dense HIP kernels (slower, same results).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn as nn

from . import _hip
from .ops import LinearActFunction, MLPFunction, PackedWeights, _f32c, grad_sink, wants_grad


def _dense(x: torch.Tensor, lin: nn.Linear, act: int = _hip.ACT_NONE, pre: float = 1.0, post: float = 1.0,
           rows: slice = slice(None)) -> torch.Tensor:
    """post*f(pre*(x W^T + b)) on the layer-wise kernels; `rows` selects output rows of the layer."""
    return LinearActFunction.apply(x, lin.weight[rows], lin.bias[rows], act, pre, post)


def _relu_stack(dims: List[int], last_act: bool) -> nn.Sequential:
    mods: List[nn.Module] = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or last_act:
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class _FusedMLP(nn.Module):
    """Shared plumbing: ordered parameter list, packed-weight cache, kernel dispatch."""

    model_id: int = -1

    def _init_fused(self) -> None:
        self._packed = PackedWeights(self.model_id)
        self._packed_bf16 = None
        self.inference_precision = "fp32"
        self.training_precision = "fp32"

    def set_training_precision(self, precision: str) -> "_FusedMLP":
        """'fp32' (default: the fp32 MFMA kernels, the path every headline figure is measured on) or the opt-in 'bf16x3':
        forward and data-gradient chain on the bf16 matrix pipe with every fp32 operand split into three bf16 terms and
        every product formed from six term products accumulated in fp32 -- fp32 accuracy (the mode holds the fp32
        path's 1e-5 bar, gradients included) at a fraction of the matrix-pipe time; so do the wide weight-gradient
        products (the narrow ones stay on fp32 MFMA).  Both models, default architecture.  Passes without gradients are
        not affected (see set_inference_precision)."""
        if precision not in ("fp32", "bf16x3"):
            raise ValueError("training precision must be 'fp32' or 'bf16x3'")
        if precision == "bf16x3":
            self._check_supported()
            if _hip.lib.nerf_mlp_packed_x3_floats(self.model_id) == 0:
                raise NotImplementedError("this model has no bf16x3 training kernels")
        if precision != self.training_precision:
            self._packed = PackedWeights(self.model_id, _hip.mlp_pack_x3 if precision == "bf16x3" else _hip.mlp_pack)
        self.training_precision = precision
        return self

    def set_inference_precision(self, precision: str) -> "_FusedMLP":
        """'fp32' (default, the parity path) or one of two opt-in modes for renders without gradients
        (torch.no_grad() / parameters that do not require grad) on the bf16 matrix pipe.  'bf16': weights,
        encodings and activations are rounded to bf16 and accumulated in fp32 (tolerance of its own).
        'bf16x3': every fp32 operand is split into three bf16 terms and a product is formed from six term
        products accumulated in fp32 -- fp32 accuracy (it holds the fp32 path's parity bar) at about 1.7 times
        the fp32 kernel's rate; both models.  'bf16' covers NeRF only.  Training always runs fp32."""
        if precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("precision must be 'fp32', 'bf16' or 'bf16x3'")
        if precision != "fp32":
            self._check_supported()
            if getattr(_hip.lib, f"nerf_mlp_packed_{precision}_bytes")(self.model_id) == 0:
                raise NotImplementedError(f"the {precision} inference mode covers the NeRF model only")
            if self._packed_bf16 is None or precision != self.inference_precision:
                self._packed_bf16 = PackedWeights(self.model_id, _hip.mlp_pack_bf16 if precision == "bf16" else _hip.mlp_pack_bf16x3)
        self.inference_precision = precision
        return self

    def ordered_parameters(self) -> List[torch.Tensor]:
        """The 24 tensors in state_dict order (the order the C ABI expects)."""
        ps = list(self.parameters())
        if len(ps) != _hip.NUM_PARAM_TENSORS:
            raise RuntimeError(f"expected {_hip.NUM_PARAM_TENSORS} parameter tensors, found {len(ps)}")
        return ps

    def packed_weights(self, train: bool = True) -> torch.Tensor:
        """Weight image for the kernels: the fp32 image, or the bf16 / bf16x3 one for a gradient-free pass of
        a module switched to that inference mode."""
        if not train and self.inference_precision != "fp32":
            return self._packed_bf16.get(self.ordered_parameters())
        return self._packed.get(self.ordered_parameters())

    def invalidate_packed(self) -> None:
        """Call after modifying parameter memory behind autograd's back (fused optimizer)."""
        self._packed.invalidate()
        if self._packed_bf16 is not None:
            self._packed_bf16.invalidate()

    def fused_supported(self) -> bool:
        """True when the constructor arguments are the ones the fused kernels are built for."""
        raise NotImplementedError

    def _check_supported(self) -> None:
        if not self.fused_supported():
            raise NotImplementedError("the fused kernels cover the reference's default architecture only")

    def _layerwise(self, points: torch.Tensor, rays_d: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def forward(self, points: torch.Tensor, rays_d: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if not points.is_cuda:
            raise RuntimeError("this build has no CPU path: move the model and its inputs to the HIP device "
                               "(model.to('cuda'))")
        if points.dim() != 2 or points.shape[1] != 3 or rays_d.shape != points.shape:
            raise ValueError("points and rays_d must both be [M,3]")
        inputs_need_grad = torch.is_grad_enabled() and (points.requires_grad or rays_d.requires_grad)
        if inputs_need_grad or not self.fused_supported():
            # The fused kernels produce parameter gradients only (the reference's callers never ask for more).
            # Gradients with respect to points / directions, which the reference's forward also supports, come
            # from the layer-wise kernels: every dense layer and the encoding have an input gradient there.
            return self._layerwise(_f32c(points), _f32c(rays_d))
        params = self.ordered_parameters()
        train = wants_grad(params)
        return MLPFunction.apply(self.model_id, self.packed_weights(train), _f32c(points), _f32c(rays_d), train,
                                 grad_sink(self, params) if train else None, *params)


class NeRF(_FusedMLP):
    """ReLU MLP with positional encoding (reference nerf/models.py:9-75):
    block1 63->256 x5, block2 [256+63]->256 x4 ->257 (last column = raw density),
    rgb_head [256+27]->128->3 with sigmoid."""

    model_id = _hip.MODEL_NERF

    def __init__(self, pos_encoding_dim: int = 10, dir_encoding_dim: int = 4, hidden_dim: int = 256) -> None:
        super().__init__()
        pe, de, h = 3 + 6 * pos_encoding_dim, 3 + 6 * dir_encoding_dim, hidden_dim
        self.block1 = _relu_stack([pe, h, h, h, h, h], last_act=True)
        self.block2 = _relu_stack([h + pe, h, h, h, h, h + 1], last_act=False)
        self.rgb_head = nn.Sequential(nn.Linear(h + de, h // 2), nn.ReLU(), nn.Linear(h // 2, 3), nn.Sigmoid())
        self.pos_encoding_dim, self.dir_encoding_dim, self.hidden_dim = pos_encoding_dim, dir_encoding_dim, hidden_dim
        self._init_fused()

    def fused_supported(self) -> bool:
        return (self.pos_encoding_dim, self.dir_encoding_dim, self.hidden_dim) == (10, 4, 256)

    def _layerwise(self, points, rays_d):
        """nerf/models.py:62-75 layer by layer; the 257-row last trunk layer is evaluated as its 256
        feature rows (no activation) and its density row (ReLU) so that no arithmetic is left to ATen."""
        from .encoding import positional_encoding
        pe = positional_encoding(points, self.pos_encoding_dim)
        de = positional_encoding(rays_d, self.dir_encoding_dim)
        h = pe
        for lin in self.block1[0::2]:
            h = _dense(h, lin, _hip.ACT_RELU)
        h = torch.cat([h, pe], dim=1)
        trunk = list(self.block2[0::2])
        for lin in trunk[:-1]:
            h = _dense(h, lin, _hip.ACT_RELU)
        feat = _dense(h, trunk[-1], _hip.ACT_NONE, rows=slice(0, self.hidden_dim))
        density = _dense(h, trunk[-1], _hip.ACT_RELU, rows=slice(self.hidden_dim, self.hidden_dim + 1)).squeeze(-1)
        g = _dense(torch.cat([feat, de], dim=1), self.rgb_head[0], _hip.ACT_RELU)
        return _dense(g, self.rgb_head[2], _hip.ACT_SIGMOID), density


class Sine(nn.Module):
    """sin(w0 x) (reference nerf/models.py:78-90); kept for state_dict/module-tree parity."""

    def __init__(self, w0: float = 30.0) -> None:
        super().__init__()
        self.w0 = w0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sin(self.w0 * x)


class SirenLayer(nn.Module):
    """Linear + sine with the SIREN initialisation of the reference (nerf/models.py:117-122):
    weight AND bias ~ U(+-1/in) for the first layer, U(+-sqrt(6/in)/w0) otherwise."""

    def __init__(self, input_dim: int, hidden_dim: int, w0: float = 1.0, is_first: bool = False) -> None:
        super().__init__()
        self.layer = nn.Linear(input_dim, hidden_dim, bias=True)
        self.activation = Sine(w0)
        self.is_first, self.input_dim, self.w0, self.c = is_first, input_dim, w0, 6
        self.reset_parameters()

    def reset_parameters(self) -> None:
        bound = 1.0 / self.input_dim if self.is_first else math.sqrt(self.c / self.input_dim) / self.w0
        with torch.no_grad():
            self.layer.weight.uniform_(-bound, bound)
            self.layer.bias.uniform_(-bound, bound)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.activation(self.layer(x))


class Siren(_FusedMLP):
    """Sine MLP on raw xyz (reference nerf/models.py:130-203): 8 sine layers (w0 30 then 1),
    density = relu(linear)*sigma_mul, rgb = sigmoid(linear(sine-layer([remap(base), enc(dir)]))*rgb_mul)."""

    model_id = _hip.MODEL_SIREN

    def __init__(self, num_layers: int = 8, hidden_dim: int = 256, dir_encoding_dim: int = 4, sigma_mul: float = 10.,
                 rgb_mul: float = 1., w0: float = 30., hidden_w0: float = 1.) -> None:
        super().__init__()
        self.dir_encoding_dim, self.sigma_mul, self.rgb_mul = dir_encoding_dim, sigma_mul, rgb_mul
        self._cfg = (num_layers, hidden_dim, dir_encoding_dim, float(sigma_mul), float(rgb_mul), float(w0),
                     float(hidden_w0))
        layers = [SirenLayer(3, hidden_dim, w0=w0, is_first=True)]
        layers += [SirenLayer(hidden_dim, hidden_dim, w0=hidden_w0) for _ in range(num_layers - 1)]
        self.block1 = nn.Sequential(*layers)
        self.density_branch = nn.Sequential(nn.Linear(hidden_dim, 1))
        self.feature_remap = nn.Sequential(nn.Linear(hidden_dim, hidden_dim))
        self.rgb_head = nn.Sequential(
            SirenLayer(hidden_dim + 6 * dir_encoding_dim + 3, hidden_dim // 2, w0=hidden_w0),
            nn.Linear(hidden_dim // 2, 3))
        self._init_fused()

    def fused_supported(self) -> bool:
        return self._cfg == (8, 256, 4, 10.0, 1.0, 30.0, 1.0)

    def _layerwise(self, points, rays_d):
        """nerf/models.py:185-203 layer by layer (sin(w0 z), relu*sigma_mul and sigmoid(rgb_mul z) are
        epilogues of the dense kernel)."""
        from .encoding import positional_encoding
        h = points
        for layer in self.block1:
            h = _dense(h, layer.layer, _hip.ACT_SIN, pre=layer.w0)
        density = _dense(h, self.density_branch[0], _hip.ACT_RELU, post=self.sigma_mul).squeeze(-1)
        feat = _dense(h, self.feature_remap[0])
        de = positional_encoding(rays_d, self.dir_encoding_dim)
        head = self.rgb_head[0]
        g = _dense(torch.cat([feat, de], dim=1), head.layer, _hip.ACT_SIN, pre=head.w0)
        return _dense(g, self.rgb_head[1], _hip.ACT_SIGMOID, pre=self.rgb_mul), density
