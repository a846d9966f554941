import os, sys, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["NERF_HIP_LIB"] = os.path.join(ROOT, "nerf-pytorch_amd", "lib", "exp_v8.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF
from nerf.rendering import render_nerf
m = NeRF().cuda()
R, S = 4096, 256
o = torch.randn(R, 3, device="cuda"); d = torch.randn(R, 3, device="cuda")
for _ in range(3):
    with torch.no_grad():
        rgb = render_nerf(m, o, d, 2.0, 6.0, num_samples=S, device="cuda")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 32))()
_hip.lib.nerf_debug_stamps.argtypes = [ctypes.c_void_p]
print("rc", _hip.lib.nerf_debug_stamps(buf))
rows = [[buf[b * 32 + i] for i in range(32)] for b in range(1024)]
t0 = min(r[0] for r in rows)
groups = collections.defaultdict(list)
for b, r in enumerate(rows):
    hw = r[30]; xcc = r[31]
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; slot = hw & 0xF; simd = (hw >> 4) & 3
    groups[(xcc, se, sh, cu)].append((b, slot, simd, r))
print("distinct CUs seen", len(groups), "blocks per CU", collections.Counter(len(v) for v in groups.values()))
names = ["start", "inputs", "encode", "L0", "L1", "L2", "L3", "L4", "L5", "L6-8", "L9", "x", "L10", "head"]
shown = 0
for key, v in sorted(groups.items()):
    if shown >= 3: break
    shown += 1
    print("CU", key)
    for b, slot, simd, r in sorted(v, key=lambda x: x[3][0]):
        ts = [r[i] - t0 for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13)]
        print(f"   block {b:4d} slot {slot} simd {simd}: " + " ".join(f"{x/100:.1f}" for x in ts), "(us)")
        idx = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13)
        ghz = []
        for a_, b_ in zip(idx[:-1], idx[1:]):
            dt = r[b_] - r[a_]; dc = r[14 + b_] - r[14 + a_]
            ghz.append(dc / max(dt, 1) / 10.0)
        print("        clock GHz per phase: " + " ".join(f"{g:.2f}" for g in ghz))
