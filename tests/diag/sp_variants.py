"""Timing of nerf_sample_positions with and without the normalisation (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nerf-pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nerf import _hip
from bench_kernels import timeit
dev = torch.device("cuda", 0)
R, S = 262144, 256
g = torch.Generator(device=dev).manual_seed(0)
o = torch.randn(R, 3, device=dev, generator=g)
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=1)
t, _ = _hip.stratified_t(2.0, 6.0, S, torch.rand(S, device=dev, generator=g))
for norm in (True, False):
    sec = timeit(lambda: _hip.sample_positions(o, d, t, norm, 2.0, 6.0), 20)
    print("normalize", norm, "us", sec * 1e6, "GB/s", R * (24 + 12 * S) / sec / 1e9)
out = torch.empty(R, S, 3, device=dev)
sec = timeit(lambda: out.fill_(1.0), 20)
print("fill of the same size: us", sec * 1e6, "GB/s", out.numel() * 4 / sec / 1e9)
src = torch.rand(R, S, 3, device=dev)
sec = timeit(lambda: out.copy_(src), 20)
print("copy of the same size: us", sec * 1e6, "GB/s (read+write)", 2 * out.numel() * 4 / sec / 1e9)
