"""Hierarchical renders of the fp32 kernels and of the three-term mode on the same rays and draws: how far do the fine
images differ when only the coarse pass / both passes change precision?  (The re-sampling amplifies rounding-level
differences of the coarse weights on a few rays.)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for sub in ("tests", "nerf-pytorch_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
import test_gpu_parity as T
from nerf.rendering import render_nerf_hierarchical
DEV = "cuda"
for tag in ("nerf", "siren"):
    model, _ = T.make_model(tag); fine, _ = T.make_model(tag)
    o, d = T.rays(2048, 71); o, d = o.to(DEV), d.to(DEV)
    with torch.no_grad():
        torch.manual_seed(5); ref = render_nerf_hierarchical(model, fine, o, d, 2.0, 6.0, 64, 128, device=DEV, return_extras=True)
        torch.manual_seed(5); ref2 = render_nerf_hierarchical(model, fine, o, d, 2.0, 6.0, 64, 128, device=DEV, return_extras=True)
        model.set_inference_precision("bf16x3")
        torch.manual_seed(5); mid = render_nerf_hierarchical(model, fine, o, d, 2.0, 6.0, 64, 128, device=DEV, return_extras=True)
        fine.set_inference_precision("bf16x3")
        torch.manual_seed(5); out = render_nerf_hierarchical(model, fine, o, d, 2.0, 6.0, 64, 128, device=DEV, return_extras=True)
    for name, a in (("rerun", ref2), ("coarse x3", mid), ("both x3", out)):
        e0 = (a[0] - ref[0]).abs(); e1 = (a[1] - ref[1]).abs()
        print(tag, name, "coarse img max %.3e" % float(e0.max()), "fine img max %.3e" % float(e1.max()), "rays > 1e-5: %d" % int((e1.max(dim=1)[0] > 1e-5).sum()),
              "fine rel %.3e" % float((a[1] - ref[1]).norm() / ref[1].norm()))
