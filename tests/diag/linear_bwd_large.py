"""nerf_linear_bwd (csrc/linear.hip) against float64 torch products at large sample counts, output by output.
python tests/diag/linear_bwd_large.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
from nerf import _hip  # noqa: E402

DEV = "cuda"


def r(a, b):
    return float((a.double() - b).norm() / (b.norm() + 1e-300))


for M in (262144, 524288, 786432, 1042555, 1048576):
    for K, N in ((128, 3), (283, 128), (256, 256)):
        g = torch.Generator().manual_seed(M + K)
        x = torch.randn(M, K, generator=g).to(DEV)
        W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
        dy = torch.randn(M, N, generator=g).to(DEV)
        y = torch.relu(torch.randn(M, N, generator=g)).to(DEV)
        dx, dW, db = _hip.linear_bwd(dy, y, None, _hip.ACT_RELU, 1.0, 1.0, x, W, True, True)
        dz = torch.where(y > 0, dy, torch.zeros_like(dy)).double()
        print(f"M {M} K {K} N {N}: dx {r(dx, dz @ W.double()):.2e} dW {r(dW, dz.t() @ x.double()):.2e} db {r(db, dz.sum(0)):.2e}",
              flush=True)
        del x, W, dy, y, dx, dW, db, dz
        torch.cuda.empty_cache()
