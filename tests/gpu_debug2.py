import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import numpy as np, torch
from oracle import nerf_oracle as O
from nerf import _hip
z = np.load(os.path.join(ROOT, "tests/golden/g1_stratified.npz"))
for ci in (4, 5, 6):
    near, far, S, _ = z[f"c{ci}_cfg"]; S = int(S)
    u = torch.from_numpy(z[f"c{ci}_u"])
    t, _ = _hip.stratified_t(float(near), float(far), S, u.cuda())
    t = t.cpu().numpy(); ref = z[f"c{ci}_t"]
    bad = np.nonzero(t != ref)[0]
    print(ci, "t mismatches", len(bad), bad[:10], (t - ref)[bad[:5]])
    t0, _ = _hip.stratified_t(float(near), float(far), S, torch.zeros(S).cuda())
    e = z[f"c{ci}_edges"]
    bad0 = np.nonzero(t0.cpu().numpy() != e[:-1])[0]
    print("   edge mismatches", len(bad0), bad0[:10], (t0.cpu().numpy() - e[:-1])[bad0[:5]])
    step = np.float32((np.float32(far) - np.float32(near)) / np.float32(S))
    print("   step", repr(step), "edges[1]-edges[0]", repr(e[1] - e[0]))
