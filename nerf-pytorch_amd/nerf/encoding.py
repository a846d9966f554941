"""Positional encoding as a HIP kernel (drop-in for the reference's nerf/encoding.py)."""
import torch

from .ops import posenc_op


def positional_encoding(x: torch.Tensor, L: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] along dim 1
    (reference nerf/encoding.py:4-20); x is [M,D] with D <= 4.  Forward only: the fused
    MLP kernels never route gradients through the encoding, as the reference's callers
    never need gradients with respect to positions."""
    if x.requires_grad:
        raise NotImplementedError("positional_encoding: gradient with respect to x is not provided")
    return posenc_op(x, int(L))
