"""What a workgroup pass of the three-term kernels costs once: forward time at 1, 2, 4, 32 passes per CU (128 samples per
workgroup, 256 CUs), from HIP events around back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF

dev = "cuda"
torch.manual_seed(0)
m = NeRF().to(dev)
params = [p.detach() for p in m.parameters()]
img3 = _hip.mlp_pack_bf16x3(_hip.MODEL_NERF, params)
img1 = _hip.mlp_pack_bf16(_hip.MODEL_NERF, params)
for name, img, per in (("bf16x3", img3, 128), ("bf16", img1, 256)):
    for passes in (1, 2, 4, 8, 32):
        M = 256 * per * passes
        x = torch.rand(M, 3, device=dev) * 2 - 1
        d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=1)
        f = lambda: _hip.mlp_fwd_bf16(_hip.MODEL_NERF, img, M, points=x, dirs=d)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        n = max(4, 64 // passes)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{name:7s} {passes:3d} passes per CU: {us:9.1f} us per launch, {us / passes:8.1f} us per pass", flush=True)
