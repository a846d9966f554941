"""Training-forward kernel time in isolation vs between backward passes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF
m = NeRF().cuda()
R, S = 4096, 256
M = R * S
if len(sys.argv) > 1 and sys.argv[1] == "scene":   # the benchmark's rays and a stratified draw instead of random ones
    import bench
    ro, rd, _ = bench.make_rays(2, 800, 800, torch.device("cuda"), seed=42)
    idx = torch.randint(0, ro.shape[0], (R,), device="cuda")
    o, d = ro[idx].contiguous(), rd[idx].contiguous()
    t, _ = _hip.stratified_t(2.0, 6.0, S, torch.rand(S, device="cuda"))
else:
    o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=1)
    t = torch.linspace(2, 6, S, device="cuda")
packed = m.packed_weights()
colors = torch.empty(M, 3, device="cuda"); dens = torch.empty(M, device="cuda")
ws = torch.empty(_hip.lib.nerf_mlp_fwd_workspace_floats(0, M), device="cuda")
bw = torch.empty(_hip.lib.nerf_mlp_bwd_workspace_floats(0, M), device="cuda")
grads = torch.empty(_hip.param_floats(0), device="cuda")
dc = torch.randn(M, 3, device="cuda") * 1e-3; dd = torch.randn(M, device="cuda") * 1e-3
L = _hip.lib
REALLOC = "realloc" in sys.argv   # a fresh workspace per call, as the autograd path allocates it
def fwd():
    global ws, bw
    if REALLOC:
        ws = None; bw = None
        ws = torch.empty(_hip.lib.nerf_mlp_fwd_workspace_floats(0, M), device="cuda")
        bw = torch.empty(_hip.lib.nerf_mlp_bwd_workspace_floats(0, M), device="cuda")
    L.nerf_mlp_fwd(0, packed.data_ptr(), M, None, None, o.data_ptr(), d.data_ptr(), t.data_ptr(), S, 0, 2.0, 6.0,
                   colors.data_ptr(), dens.data_ptr(), ws.data_ptr(), None)
def dgrad():
    L.nerf_mlp_bwd_data(0, packed.data_ptr(), M, dc.data_ptr(), dd.data_ptr(), colors.data_ptr(), ws.data_ptr(), bw.data_ptr(), None)
def wgrad():
    L.nerf_mlp_bwd_weights(0, M, ws.data_ptr(), bw.data_ptr(), grads.data_ptr(), 0, None)
def timed(seq, names, iters=8):
    tot = {n: 0.0 for n in names}
    for it in range(iters + 2):
        evs = []
        for fn, n in zip(seq, names):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); evs.append((n, e0, e1))
        torch.cuda.synchronize()
        if it >= 2:
            for n, e0, e1 in evs: tot[n] += e0.elapsed_time(e1) / iters
    return tot
print("fwd only      ", timed([fwd], ["fwd"]))
print("fwd,dgrad     ", timed([fwd, dgrad], ["fwd", "dgrad"]))
print("fwd,dgrad,wgrad", timed([fwd, dgrad, wgrad], ["fwd", "dgrad", "wgrad"]))
print("fwd,wgrad     ", timed([fwd, wgrad], ["fwd", "wgrad"]))
print("dgrad only    ", timed([dgrad], ["dgrad"]))
print("wgrad only    ", timed([wgrad], ["wgrad"]))
