"""Isolated timing of the fused forward kernel in inference and training mode (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODEL = 1 if (len(sys.argv) > 1 and sys.argv[1] == "siren") else 0
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF, Siren
m = (Siren() if MODEL else NeRF()).cuda()
R, S = 4096, 256
M = R * S
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=1)
t = torch.linspace(2, 6, S, device="cuda")
packed = m.packed_weights()
colors = torch.empty(M, 3, device="cuda"); dens = torch.empty(M, device="cuda")
ws = torch.empty(_hip.lib.nerf_mlp_fwd_workspace_floats(MODEL, M), device="cuda")
def run(train):
    _hip.lib.nerf_mlp_fwd(MODEL, packed.data_ptr(), M, None, None, o.data_ptr(), d.data_ptr(), t.data_ptr(), S, 0, 2.0, 6.0,
                          colors.data_ptr(), dens.data_ptr(), ws.data_ptr() if train else None, None)
for rep in range(2):
    for train in (False, True):
        for _ in range(3): run(train)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(train)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("train" if train else "infer", f"{ms:.3f} ms  {(1123840 if MODEL else 1317888) * M / ms / 1e9:.1f} TFLOP/s")
