"""Core clock sustained under each fused kernel (GPU box): a one-wave probe samples the shader clock
counter against the 100 MHz counter on a second stream while the kernel under test runs back to back.
Build tests/diag/libclockprobe.so first (command in clock_probe.hip)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODEL = 1 if (len(sys.argv) > 1 and sys.argv[1] == "siren") else 0
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np
import torch
from nerf import _hip
from nerf.models import NeRF, Siren

probe = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libclockprobe.so"))
probe.clock_probe_launch.restype = ctypes.c_int
probe.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

m = (Siren() if MODEL else NeRF()).cuda()
R, S = 4096, 256
M = R * S
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=1)
t = torch.linspace(2, 6, S, device="cuda")
packed = m.packed_weights()
colors = torch.empty(M, 3, device="cuda"); dens = torch.empty(M, device="cuda")
dcol = torch.randn(M, 3, device="cuda") * 1e-3; dden = torch.randn(M, device="cuda") * 1e-3
ws = torch.empty(_hip.lib.nerf_mlp_fwd_workspace_floats(MODEL, M), device="cuda")
bws = torch.empty(_hip.lib.nerf_mlp_bwd_workspace_floats(MODEL, M), device="cuda")
grads = torch.zeros(_hip.lib.nerf_mlp_param_floats(MODEL), device="cuda")
main, side = torch.cuda.Stream(), torch.cuda.Stream()
L = _hip.lib


def fwd(train):
    rc = L.nerf_mlp_fwd(MODEL, packed.data_ptr(), M, None, None, o.data_ptr(), d.data_ptr(), t.data_ptr(), S, 0, 2.0, 6.0,
                        colors.data_ptr(), dens.data_ptr(), ws.data_ptr() if train else None, main.cuda_stream)
    assert rc == 0, L.nerf_last_error()


def dgrad():
    rc = L.nerf_mlp_bwd_data(MODEL, packed.data_ptr(), M, dcol.data_ptr(), dden.data_ptr(), colors.data_ptr(), ws.data_ptr(),
                             bws.data_ptr(), main.cuda_stream)
    assert rc == 0, L.nerf_last_error()


def wgrad():
    rc = L.nerf_mlp_bwd_weights(MODEL, M, ws.data_ptr(), bws.data_ptr(), grads.data_ptr(), 0, main.cuda_stream)
    assert rc == 0, L.nerf_last_error()


PROBES = 16


def measure(name, fn, reps, flop):
    n, spin = 1500, 1000                       # 1500 samples 10 us apart = 15 ms window
    out = torch.zeros(PROBES * (2 * n + 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(reps): fn()                 # bring the device to its steady state under this load
    e0.record(main)
    side.wait_event(e0)                        # the probe window opens when the timed launches start
    rc = probe.clock_probe_launch(out.data_ptr(), PROBES, n, spin, side.cuda_stream)
    assert rc == 0
    for _ in range(reps): fn()
    e1.record(main)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    raw = out.cpu().numpy().reshape(PROBES, 2 * n + 2)
    per_xcc = {}
    for p in range(PROBES):
        v = raw[p, 2:].reshape(n, 2).astype(np.float64)
        ghz = np.diff(v[:, 0]) / np.diff(v[:, 1]) * 0.1     # cycles per 10 ns tick -> GHz
        per_xcc.setdefault(int(raw[p, 0]), []).append(float(np.median(ghz)))
    line = f"{name:14s} {ms:7.3f} ms/launch"
    if flop:
        line += f"  {flop * M / ms / 1e9:6.1f} TFLOP/s"
    line += "  median clock by XCC: " + " ".join(f"{x}:{np.mean(c):.3f}" for x, c in sorted(per_xcc.items()))
    print(line, flush=True)


FWD = 1123840 if MODEL else 1317888
DGRAD = (3363072 - 2 * 1123840) if MODEL else 2 * 623232
measure("idle", lambda: None, 1, 0)
measure("fwd infer", lambda: fwd(False), 4, FWD)
measure("fwd train", lambda: fwd(True), 4, FWD)
measure("dgrad", dgrad, 4, DGRAD)
measure("wgrad group", wgrad, 4, FWD)
measure("fwd infer", lambda: fwd(False), 4, FWD)

# sustained load: two seconds of training-forward launches, probed every 40 launches
print("sustained fwd train:", flush=True)
for block in range(5):
    measure(f"  after {40 * block:3d}", lambda: fwd(True), 20, FWD)
