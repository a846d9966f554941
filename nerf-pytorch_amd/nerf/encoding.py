"""Positional encoding as a HIP kernel (drop-in for the reference's nerf/encoding.py)."""
import torch

from .ops import posenc_op


def positional_encoding(x: torch.Tensor, L: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] along dim 1
    (reference nerf/encoding.py:4-20); x is [M,D] with D <= 4.  Differentiable in x like the
    reference's torch ops (the fused MLP kernels compute their encodings internally and never need
    that gradient: the reference's callers do not ask for gradients with respect to positions)."""
    return posenc_op(x, int(L))
