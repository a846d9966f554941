"""Why is the training forward slower inside a step than in a tight loop?  Runs training steps through the normal
Python path (timers on), then the same forward launch in a tight loop in the same process, first with the
trained weights and the step's rays, then again after a pause.  (GPU box.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
import bench
from nerf import _hip, parallel
from nerf.models import NeRF
from nerf.rendering import render_nerf

dev = torch.device("cuda")
torch.manual_seed(0)
model = NeRF().to(dev)
flat = parallel.FlatParameters(model)
opt = parallel.FusedAdam(flat, lr=5e-4)
ro, rd, tg = bench.make_rays(12, 800, 800, dev, seed=42)
R, S = 4096, 256
M = R * S
gen = torch.Generator(device=dev).manual_seed(1)


def step():
    idx = torch.randint(0, ro.shape[0], (R,), device=dev, generator=gen)
    o, d, t = ro[idx], rd[idx], tg[idx]
    u = torch.rand(S, device=dev, generator=gen)
    rgb = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=dev, chunk_size=8192, u=[u])
    opt.zero_grad()
    torch.nn.functional.mse_loss(rgb, t).backward()
    opt.step()
    return o, d, u


for _ in range(3):
    o, d, u = step()
torch.cuda.synchronize()
_hip.timers.enabled = True
_hip.timers.reset()
for _ in range(8):
    o, d, u = step()
torch.cuda.synchronize()
_hip.timers.enabled = False
print({k: round(v[1] / v[0], 3) for k, v in _hip.timers.summary().items()})

packed = model.packed_weights(True)
t, _ = _hip.stratified_t(2.0, 6.0, S, u)
colors = torch.empty(M, 3, device=dev); dens = torch.empty(M, device=dev)
ws = torch.empty(_hip.lib.nerf_mlp_fwd_workspace_floats(0, M), device=dev)


def loop(tag, n=8):
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.lib.nerf_mlp_fwd(0, packed.data_ptr(), M, None, None, o.data_ptr(), d.data_ptr(), t.data_ptr(), S, 0, 2.0, 6.0,
                              colors.data_ptr(), dens.data_ptr(), ws.data_ptr(), None)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    print(tag, [round(a.elapsed_time(b), 2) for a, b in evs])


loop("tight loop right after training")
time.sleep(1.0)
loop("tight loop after 1 s idle")
fresh = NeRF().to(dev)
packed = fresh.packed_weights(True)
loop("tight loop, freshly initialised weights")
