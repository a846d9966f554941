"""Phase timeline of the fp32 NeRF forward kernel from in-kernel 100 MHz stamps (GPU box; needs
nerf-pytorch_amd/lib/diag.so: `make diag` in nerf-pytorch_amd/csrc).  Usage: fwd_stamps.py [train]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NERF_HIP_LIB"] = os.path.join(ROOT, "nerf-pytorch_amd", "lib", "diag.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np
import torch
from nerf import _hip
from nerf.models import NeRF

TRAIN = len(sys.argv) > 1 and sys.argv[1] == "train"
m = NeRF().cuda()
R, S = 4096, 256
M = R * S
o = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=1)
t = torch.linspace(2, 6, S, device="cuda")
packed = m.packed_weights()
colors = torch.empty(M, 3, device="cuda"); dens = torch.empty(M, device="cuda")
ws = torch.empty(_hip.lib.nerf_mlp_fwd_workspace_floats(0, M), device="cuda") if TRAIN else None
ntile = M // 64
dbg = torch.zeros(ntile * 32, dtype=torch.int64, device="cuda")
_hip.lib.nerf_diag_fwd_stamps.argtypes = [ctypes.c_void_p]


def run():
    rc = _hip.lib.nerf_mlp_fwd(0, packed.data_ptr(), M, None, None, o.data_ptr(), d.data_ptr(), t.data_ptr(), S, 0, 2.0, 6.0,
                               colors.data_ptr(), dens.data_ptr(), ws.data_ptr() if TRAIN else None, None)
    assert rc == 0


for _ in range(4): run()
_hip.lib.nerf_diag_fwd_stamps(dbg.data_ptr())
run()
torch.cuda.synchronize()
_hip.lib.nerf_diag_fwd_stamps(None)
g = dbg.view(ntile, 32).cpu().numpy().astype(np.int64)
T = g[:, :19].astype(np.float64) * 0.01          # microseconds
names = ["inputs", "encode pe", "L0", "L1", "L2", "L3", "L4", "L5a loop", "encode pe 2", "L5b+epilogue", "L6", "L7", "L8", "L9",
         "L10a loop", "encode de", "L10b+epilogue", "head"]
print(("train" if TRAIN else "infer"), "kernel span %.1f us" % (T[:, 18].max() - T[:, 0].min()), " tiles", ntile)
print("tile duration mean %.1f us  (p10 %.1f, p90 %.1f)" % ((T[:, 18] - T[:, 0]).mean(), np.percentile(T[:, 18] - T[:, 0], 10),
                                                           np.percentile(T[:, 18] - T[:, 0], 90)))
for i, n in enumerate(names):
    dur = T[:, i + 1] - T[:, i]
    print(f"  {n:14s} mean {dur.mean():7.2f} us   p10 {np.percentile(dur, 10):7.2f}  p90 {np.percentile(dur, 90):7.2f}")

L = g[:, 20:25].astype(np.float64) * 0.01
print("inside layer 2 (wave 0): bias + MFMA loop %.2f us, wait at barrier %.2f, epilogue %.2f, wait at barrier %.2f" % tuple(
    (L[:, i + 1] - L[:, i]).mean() for i in range(4)))
print("   p10/p90: loop %.2f/%.2f  epilogue %.2f/%.2f" % (np.percentile(L[:, 1] - L[:, 0], 10), np.percentile(L[:, 1] - L[:, 0], 90),
                                                          np.percentile(L[:, 3] - L[:, 2], 10), np.percentile(L[:, 3] - L[:, 2], 90)))

# per compute unit: how many resident tiles, and how many of them inside an MFMA phase, over time
MFMA = [(2, 8), (9, 15), (16, 17)]               # stamp index ranges that contain MFMA loops (with their epilogues)
cu_key = (g[:, 30] << 16) | (g[:, 31] & 0xff00)
t0 = T[:, 0].min()
res_hist = np.zeros(4); mf_hist = np.zeros(4); gaps = []
for key in np.unique(cu_key):
    idx = np.where(cu_key == key)[0]
    ev = []
    for i in idx:
        ev.append((T[i, 0], 0, +1)); ev.append((T[i, 18], 0, -1))
        for a, b in MFMA:
            ev.append((T[i, a], 1, +1)); ev.append((T[i, b], 1, -1))
    ev.sort()
    res = mf = 0
    prev = ev[0][0]
    for tm, kind, dlt in ev:
        res_hist[min(res, 3)] += tm - prev
        mf_hist[min(mf, 3)] += tm - prev
        prev = tm
        if kind == 0: res += dlt
        else: mf += dlt
    # turnover: time from a tile's end to the next tile start on this CU
    starts = np.sort(T[idx, 0]); ends = np.sort(T[idx, 18])
    for e in ends[:-2]:
        nxt = starts[starts >= e - 1e-9]
        if len(nxt): gaps.append(nxt[0] - e)
print("compute units seen:", len(np.unique(cu_key)))
print("share of CU time with n resident tiles   0/1/2/3+:", np.round(res_hist / res_hist.sum(), 3))
print("share of CU time with n tiles in MFMA phases 0/1/2/3+:", np.round(mf_hist / mf_hist.sum(), 3))
gaps = np.array(gaps)
print("turnover (tile end -> next tile start on the CU): mean %.2f us  p50 %.2f  p90 %.2f" % (gaps.mean(), np.median(gaps),
                                                                                              np.percentile(gaps, 90)))
