"""Core clock over time while whole training steps run (GPU box): the one-wave probe of clock_probe.hip samples
the shader clock against the 100 MHz counter every 10 us on a second stream during three default bench steps
(4096 rays, 64 coarse + 192 fine evaluations, two networks).  Needs tests/diag/libclockprobe.so."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np
import torch
from nerf import parallel
from nerf.models import NeRF
from nerf.rendering import render_nerf_hierarchical

probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libclockprobe.so"))
probe.clock_probe_launch.restype = ctypes.c_int
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
models = [NeRF().to(dev) for _ in range(2)]
opts = [parallel.FusedAdam(parallel.FlatParameters(torch.nn.ModuleList(models)), lr=5e-4)]
B = 4096
o = torch.randn(B, 3, device=dev); o = 4 * o / o.norm(dim=1, keepdim=True)
d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(B, 3, device=dev), dim=1)
tgt = torch.rand(B, 3, device=dev)


def step():
    rc, rf = render_nerf_hierarchical(models[0], models[1], o, d, 2.0, 6.0, 64, 128, device=dev, chunk_size=8192)
    for op in opts: op.zero_grad()
    loss = torch.nn.functional.mse_loss(rc, tgt) + torch.nn.functional.mse_loss(rf, tgt)
    loss.backward()
    for op in opts: op.step()


for _ in range(6): step()
torch.cuda.synchronize()
side = torch.cuda.Stream()
n, spin = 10000, 1000                      # 100 ms window
out = torch.zeros(2 * n + 2, dtype=torch.int64, device=dev)
for _ in range(3): step()                  # the device is in its loaded state when the window opens
ev = torch.cuda.Event(); ev.record()
side.wait_event(ev)
assert probe.clock_probe_launch(out.data_ptr(), 1, n, spin, side.cuda_stream) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): step()
e1.record()
torch.cuda.synchronize()
print(f"step {e0.elapsed_time(e1) / 3:.2f} ms")
v = out.cpu().numpy()[2:].reshape(n, 2).astype(np.float64)
t = (v[:, 1] - v[0, 1]) * 1e-5             # ms
ghz = np.diff(v[:, 0]) / np.diff(v[:, 1]) * 0.1
busy = t[1:] < e0.elapsed_time(e1)
print(f"clock while the three steps run: mean {ghz[busy].mean():.3f} GHz, p5 {np.percentile(ghz[busy], 5):.3f}, "
      f"p50 {np.median(ghz[busy]):.3f}, p95 {np.percentile(ghz[busy], 95):.3f}")
print("timeline, mean GHz per 1 ms:")
edges = np.arange(0, min(t[-1], 100.0), 1.0)
line = []
for a in edges:
    m = (t[1:] >= a) & (t[1:] < a + 1.0)
    line.append(f"{ghz[m].mean():.2f}" if m.any() else " -- ")
for i in range(0, len(line), 20):
    print(f"  {int(edges[i]):3d} ms: " + " ".join(line[i:i + 20]))
