"""Forward-kernel rates of the three inference modes on the same samples: fp32 MFMA, bf16, bf16x3 (three-term split)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF, Siren

dev = "cuda"
torch.manual_seed(0)
SIREN = len(sys.argv) > 1 and sys.argv[1] == "siren"
m = (Siren() if SIREN else NeRF()).to(dev)
MODEL = _hip.MODEL_SIREN if SIREN else _hip.MODEL_NERF
M = 1 << 20
x = torch.rand(M, 3, device=dev) * 2 - 1
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=1)
params = [p.detach() for p in m.parameters()]
img32 = _hip.mlp_pack(MODEL, params)
img16 = None if SIREN else _hip.mlp_pack_bf16(MODEL, params)
img3 = _hip.mlp_pack_bf16x3(MODEL, params)
FLOP = 2 * (561920 if SIREN else 658944)


def run(name, f):
    out = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name:8s} {ms:8.3f} ms  {M / ms / 1e3:8.1f} M samples/s  {M * FLOP / ms / 1e9:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
    return out


c32, s32, _ = run("fp32", lambda: _hip.mlp_fwd(MODEL, img32, M, points=x, dirs=d, train=False))
c16, s16 = c32, s32
if not SIREN:
    c16, s16, _ = run("bf16", lambda: _hip.mlp_fwd_bf16(MODEL, img16, M, points=x, dirs=d))
c3, s3, _ = run("bf16x3", lambda: _hip.mlp_fwd_bf16(MODEL, img3, M, points=x, dirs=d))
for name, c, s in (("bf16", c16, s16), ("bf16x3", c3, s3)):
    print(f"{name:8s} colors rel {float((c - c32).norm() / c32.norm()):.3e} max abs {float((c - c32).abs().max()):.3e}; "
          f"density rel {float((s - s32).norm() / (s32.norm() + 1e-30)):.3e} max abs {float((s - s32).abs().max()):.3e}")
