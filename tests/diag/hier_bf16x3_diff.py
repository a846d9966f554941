import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/nerf-pytorch_amd"); sys.path.insert(0, "/root/repo")
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
