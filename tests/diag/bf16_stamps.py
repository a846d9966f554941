"""Per-phase shader-clock stamps of the bf16 kernel (needs lib/diag.so built with -DNERF_DIAG)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NERF_HIP_LIB"] = os.path.join(ROOT, "nerf-pytorch_amd", "lib", "diag.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF
from nerf.rendering import render_nerf
PREC = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = NeRF().cuda().set_inference_precision(PREC)
R, S = 4096, 256
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=1)
WG = 256 if PREC == "bf16" else 128     # samples per workgroup
nwg = R * S // WG
dbg = torch.zeros(nwg * 32, dtype=torch.int64, device="cuda")
_hip.lib.nerf_diag_bf16_stamps.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3): render_nerf(m, o, d, 2.0, 6.0, num_samples=S, device="cuda")
    _hip.lib.nerf_diag_bf16_stamps(dbg.data_ptr())
    render_nerf(m, o, d, 2.0, 6.0, num_samples=S, device="cuda")
    torch.cuda.synchronize()
g = dbg.view(nwg, 32).cpu().double()
def span(a, b): return float((g[:, b] - g[:, a]).mean())
print("total", span(0, 5), " prologue", span(0, 1), " L0", span(1, 2), " L1-4", span(2, 3), " L5..L11", span(3, 4), " out", span(4, 5))
for l in range(4):
    print(f"  L{l+1}: bias {span(10+3*l-1 if l else 2, 10+3*l):.0f}  mma {span(10+3*l, 11+3*l):.0f}  frags {span(11+3*l, 12+3*l):.0f}")
first = g[:, 0]; last = g[:, 5]
print("kernel span cycles", float(last.max() - first.min()), " tiles/CU", nwg / 256)
