"""bf16x3 training mode (opt-in, no counterpart in the reference): forward with workspace and the data-gradient chain
on the bf16 matrix pipe must hold the fp32 path's 1e-5 bar against the oracle, every gradient tensor included."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_parity import DEV, RTOL, T, make_model, rays, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M", [1, 63, 65, 200, 1000, 4096])
def test_bf16x3_training_holds_fp32_bar(M):
    model, params = make_model("nerf")
    model.set_training_precision("bf16x3")
    ref_model, _ = make_model("nerf")
    g = torch.Generator().manual_seed(M)
    pts = torch.rand(M, 3, generator=g) * 2 - 1
    dirs = torch.randn(M, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    gc, gd = torch.randn(M, 3, generator=g), torch.randn(M, generator=g) * 0.1
    with torch.no_grad():
        _, _, acts = O.nerf_forward(params, pts, dirs, return_acts=True)
    keep = O.relu_kink_distance(acts) > 3e-6
    pts, dirs, gc, gd = pts[keep].contiguous(), dirs[keep].contiguous(), gc[keep].contiguous(), gd[keep].contiguous()
    for p in params.values():
        p.requires_grad_(True)
    c_ref, d_ref = O.nerf_forward(params, pts, dirs)
    ((c_ref * gc).sum() + (d_ref * gd).sum()).backward()
    colors, dens = model(pts.to(DEV), dirs.to(DEV))
    ((colors * gc.to(DEV)).sum() + (dens * gd.to(DEV)).sum()).backward()
    c32, d32 = ref_model(pts.to(DEV), dirs.to(DEV))
    ((c32 * gc.to(DEV)).sum() + (d32 * gd.to(DEV)).sum()).backward()
    print("fwd vs oracle", rel(colors, c_ref), rel(dens, d_ref), "vs fp32 kernel", rel(colors, c32), rel(dens, d32))
    assert rel(colors, c_ref) < RTOL and rel(dens, d_ref) < RTOL
    worst = 0.0
    for (n, p), pr, (_, p32) in zip(model.named_parameters(), params.values(), ref_model.named_parameters()):
        e, e32 = rel(p.grad, pr.grad), rel(p.grad, p32.grad)
        print(f"  {n:20s} vs oracle {e:.2e}  vs fp32 kernel {e32:.2e}")
        worst = max(worst, e)
    assert worst < RTOL


def test_bf16x3_render_step_matches_oracle():
    """One training step's gradients through render_nerf (ray mode: positions generated in the kernel, compositing and
    its backward in between), rays chosen clear of every ReLU kink under the draw; both modes against the oracle."""
    from nerf.rendering import render_nerf
    m3, params = make_model("nerf")
    m3.set_training_precision("bf16x3")
    m32, _ = make_model("nerf")
    S = 64
    o, d = rays(320, 5)
    torch.manual_seed(11)
    u = torch.rand(S, device=DEV).cpu()
    t = O.stratified_t(2.0, 6.0, S, u)
    with torch.no_grad():
        pos = O.normalize_positions(O.sample_positions(o, d, t), 2.0, 6.0).reshape(-1, 3)
        _, _, acts = O.nerf_forward(params, pos, d[:, None, :].expand(-1, S, -1).reshape(-1, 3), return_acts=True)
    keep = (O.relu_kink_distance(acts).reshape(-1, S) > 3e-6).all(dim=1)
    o, d = o[keep][:256].contiguous(), d[keep][:256].contiguous()
    R = o.shape[0]
    assert R >= 128
    tgt = torch.rand(R, 3, generator=torch.Generator().manual_seed(6))
    for p in params.values():
        p.requires_grad_(True)
    rgb_ref = O.render(lambda x, v: O.nerf_forward(params, x, v), o, d, 2.0, 6.0, S, [u])
    torch.nn.functional.mse_loss(rgb_ref, tgt).backward()
    for name, m in (("bf16x3", m3), ("fp32", m32)):
        torch.manual_seed(11)
        rgb = render_nerf(m, o.to(DEV), d.to(DEV), 2.0, 6.0, num_samples=S, device=DEV, chunk_size=8192)
        torch.nn.functional.mse_loss(rgb, tgt.to(DEV)).backward()
        errs = {n: rel(p.grad, pr.grad) for (n, p), pr in zip(m.named_parameters(), params.values())}
        print(name, "rgb", rel(rgb, rgb_ref), "worst gradient", max(errs.values()), max(errs, key=errs.get))
        assert rel(rgb, rgb_ref) < RTOL
        assert max(errs.values()) < RTOL, (name, errs)
    # inference through a module in this mode reads the fp32 part of the image: bitwise the fp32 module's render
    with torch.no_grad():
        torch.manual_seed(11)
        a = render_nerf(m3, o.to(DEV), d.to(DEV), 2.0, 6.0, num_samples=S, device=DEV)
        torch.manual_seed(11)
        b = render_nerf(m32, o.to(DEV), d.to(DEV), 2.0, 6.0, num_samples=S, device=DEV)
    assert torch.equal(a, b)
