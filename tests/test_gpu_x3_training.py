"""bf16x3 training mode (opt-in, no counterpart in the reference): forward with workspace and the data-gradient chain
on the bf16 matrix pipe must hold the fp32 path's 1e-5 bar against the oracle, every gradient tensor included."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from test_gpu_parity import DEV, RTOL, T, make_model, rays, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["nerf", "siren"])
@pytest.mark.parametrize("M", [1, 63, 65, 200, 1000, 4096])
def test_bf16x3_training_holds_fp32_bar(tag, M):
    from test_gpu_parity import oracle_fwd
    model, params = make_model(tag)
    model.set_training_precision("bf16x3")
    ref_model, _ = make_model(tag)
    g = torch.Generator().manual_seed(M)
    pts = torch.rand(M, 3, generator=g) * 2 - 1
    dirs = torch.randn(M, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    gc, gd = torch.randn(M, 3, generator=g), torch.randn(M, generator=g) * 0.1
    with torch.no_grad():
        _, _, acts = oracle_fwd(tag)(params, pts, dirs, return_acts=True)
    keep = O.relu_kink_distance(acts) > 3e-6
    pts, dirs, gc, gd = pts[keep].contiguous(), dirs[keep].contiguous(), gc[keep].contiguous(), gd[keep].contiguous()
    for p in params.values():
        p.requires_grad_(True)
    c_ref, d_ref = oracle_fwd(tag)(params, pts, dirs)
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


@pytest.mark.parametrize("tag", ["nerf", "siren"])
def test_bf16x3_render_step_matches_oracle(tag):
    """One training step's gradients through render_nerf (ray mode: positions generated in the kernel, compositing and
    its backward in between), rays chosen clear of every ReLU kink under the draw; both modes against the oracle."""
    from nerf.rendering import render_nerf
    from test_gpu_parity import oracle_fwd
    fwd = oracle_fwd(tag)
    m3, params = make_model(tag)
    m3.set_training_precision("bf16x3")
    m32, _ = make_model(tag)
    S = 64
    o, d = rays(320, 5)
    torch.manual_seed(11)
    u = torch.rand(S, device=DEV).cpu()
    t = O.stratified_t(2.0, 6.0, S, u)
    with torch.no_grad():
        pos = O.normalize_positions(O.sample_positions(o, d, t), 2.0, 6.0).reshape(-1, 3)
        _, _, acts = fwd(params, pos, d[:, None, :].expand(-1, S, -1).reshape(-1, 3), return_acts=True)
    keep = (O.relu_kink_distance(acts).reshape(-1, S) > 3e-6).all(dim=1)
    o, d = o[keep][:256].contiguous(), d[keep][:256].contiguous()
    R = o.shape[0]
    assert R >= 128
    tgt = torch.rand(R, 3, generator=torch.Generator().manual_seed(6))
    for p in params.values():
        p.requires_grad_(True)
    rgb_ref = O.render(lambda x, v: fwd(params, x, v), o, d, 2.0, 6.0, S, [u])
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


def test_bf16x3_training_is_reproducible_and_tracks_fp32_over_adam_steps():
    """Coarse + fine pair, one bucket, fused Adam: two runs of the bf16x3 mode are bit-identical (the weight-gradient
    partial slabs are summed in a fixed order, no atomics), and after 5 steps its parameters sit where the fp32
    kernels put them, to rounding amplified by Adam (the bound the two-rank tests use: a fraction of the distance
    travelled)."""
    from nerf import parallel
    from nerf.models import NeRF
    from nerf.rendering import render_nerf_hierarchical
    o, d = rays(512, 21)
    o, d = o.to(DEV), d.to(DEV)
    tgt = torch.rand(512, 3, generator=torch.Generator().manual_seed(22)).to(DEV)

    def run(precision):
        torch.manual_seed(3)
        nets = torch.nn.ModuleDict({"coarse": NeRF(), "fine": NeRF()}).to(DEV)
        for n in nets.values():
            n.set_training_precision(precision)
        flat = parallel.FlatParameters(nets)
        start = flat.flat.clone()
        opt = parallel.FusedAdam(flat, lr=5e-4)
        gen = torch.Generator(device=DEV).manual_seed(5)
        losses = []
        for _ in range(5):
            u = torch.rand(32, device=DEV, generator=gen)
            uf = torch.rand(512, 64, device=DEV, generator=gen)
            opt.zero_grad()
            rc, rf = render_nerf_hierarchical(nets["coarse"], nets["fine"], o, d, 2.0, 6.0, 32, 64, device=DEV, u=[u], u_fine=[uf])
            loss = torch.nn.functional.mse_loss(rc, tgt) + torch.nn.functional.mse_loss(rf, tgt)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return flat.flat.clone(), start, losses

    p3a, start, l3 = run("bf16x3")
    p3b, _, _ = run("bf16x3")
    p32, _, l32 = run("fp32")
    assert torch.equal(p3a, p3b)
    moved = float((p32 - start).norm())
    print("losses", l3, l32, "distance between modes / distance travelled", float((p3a - p32).norm()) / moved)
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(l3[:2], l32[:2]))
    assert float((p3a - p32).norm()) <= 2e-2 * moved


def test_bf16x3_mode_is_switchable():
    from nerf.models import NeRF
    with pytest.raises(NotImplementedError):
        NeRF(hidden_dim=64).to(DEV).set_training_precision("bf16x3")     # the fused kernels' architecture only
    with pytest.raises(ValueError):
        NeRF().to(DEV).set_training_precision("fp16")
    m, _ = make_model("nerf")
    x, v = torch.rand(100, 3, device=DEV) * 2 - 1, torch.nn.functional.normalize(torch.randn(100, 3, device=DEV), dim=1)
    outs = []
    for prec in ("fp32", "bf16x3", "fp32"):
        m.set_training_precision(prec)
        m.zero_grad()
        c, s = m(x, v)
        (c.sum() + s.sum()).backward()
        outs.append((c.detach().clone(), m.block1[0].weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])   # back on the fp32 kernels, bit for bit
    assert rel(outs[1][0], outs[0][0]) < RTOL


@pytest.mark.parametrize("M", [1, 100, 129, 1000, 4133])
def test_bf16x3_forward_fills_the_workspace_of_the_fp32_forward(M):
    """NeRF model: the bf16x3 training forward (16-sample tiles, 128 samples per workgroup) writes every workspace field the
    gradient kernels read where the fp32 forward writes it (pe, de, h0..h8, f, g, density pre-activation; rows up to the
    padded sample count, which need not be a multiple of 128), to 1e-5; its ReLU decisions, in its own layout (32 bytes
    per sample: word 2q + (ot >> 3), bit 31 - (4 (ot & 7) + g) = feature 16 ot + 4 q + g), are exactly h > 0."""
    from nerf import _hip
    model, _ = make_model("nerf")
    g = torch.Generator().manual_seed(M)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).to(DEV)
    c32, d32, ws32 = _hip.mlp_fwd(model.model_id, model.packed_weights(True), M, points=pts, dirs=dirs, train=True)
    model.set_training_precision("bf16x3")
    c3, d3, ws3 = _hip.mlp_fwd(model.model_id, model.packed_weights(True), M, points=pts, dirs=dirs, train=True)
    assert _hip.is_x3_image(model.model_id, model.packed_weights(True))
    assert rel(c3, c32) < RTOL and rel(d3, d32) < RTOL
    Mp = -(-M // 64) * 64
    assert ws3.numel() == ws32.numel() == Mp * (64 + 32 + 9 * 256 + 256 + 128 + 1 + 72)
    fields = [("pe", 0, 64), ("de", 64, 32)] + [(f"h{l}", 96 + 256 * l, 256) for l in range(9)]
    fields += [("f", 96 + 9 * 256, 256), ("g", 96 + 10 * 256, 128), ("sraw", 96 + 10 * 256 + 128, 1)]
    for name, off, width in fields:
        a = ws3[off * Mp:(off + width) * Mp].view(Mp, width)
        b = ws32[off * Mp:(off + width) * Mp].view(Mp, width)
        assert rel(a[:M], b[:M]) < RTOL, name
        scale = float(b[:M].abs().max())
        assert float((a[:M] - b[:M]).abs().max()) <= 2e-5 * max(scale, 1.0), name
        assert torch.isfinite(a).all(), name                      # rows M..Mp are written too (zero inputs)
    assert float(ws3[:64 * Mp].view(Mp, 64)[:, 63].abs().max()) == 0.0                               # padding columns
    de = ws3[64 * Mp:96 * Mp].view(Mp, 32)
    assert float(de[:, 27:].abs().max()) == 0.0
    mask0 = 96 + 10 * 256 + 128 + 1
    feat = torch.arange(256, device=DEV)
    ot, q, gg = feat // 16, (feat % 16) // 4, feat % 4
    word, bit = 2 * q + ot // 8, 31 - (4 * (ot % 8) + gg)
    for l in range(9):
        words = ws3[(mask0 + 8 * l) * Mp:(mask0 + 8 * l + 8) * Mp].view(torch.int32).view(Mp, 8)
        decided = (words[:, word] >> bit) & 1                                      # [Mp, 256]
        h = ws3[(96 + 256 * l) * Mp:(96 + 256 * (l + 1)) * Mp].view(Mp, 256)
        assert torch.equal(decided.bool(), h > 0), f"mask of h{l}"


@pytest.mark.parametrize("M", [1, 129, 1000])
def test_bf16x3_siren_forward_workspace(M):
    """Siren: x, de, h0..h7, f, g and the density pre-activation where the fp32 forward writes them, to 1e-5; the
    derivative fields (row-major like the activations in this mode; the fp32 kernels keep theirs in accumulator order)
    against the activations they belong to: (w0 cos)^2 / w0^2 + sin^2 = 1."""
    from nerf import _hip
    model, _ = make_model("siren")
    g = torch.Generator().manual_seed(M)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).to(DEV)
    c32, d32, ws32 = _hip.mlp_fwd(model.model_id, model.packed_weights(True), M, points=pts, dirs=dirs, train=True)
    model.set_training_precision("bf16x3")
    c3, d3, ws3 = _hip.mlp_fwd(model.model_id, model.packed_weights(True), M, points=pts, dirs=dirs, train=True)
    assert rel(c3, c32) < RTOL and rel(d3, d32) < RTOL
    Mp = -(-M // 64) * 64
    H, HH = 256, 128
    WS_H, WS_C = 36, 36 + 8 * H
    WS_F = WS_C + 8 * H
    WS_G, WS_CG, WS_SRAW = WS_F + H, WS_F + H + HH, WS_F + H + 2 * HH
    assert ws3.numel() == ws32.numel() == Mp * (WS_SRAW + 1)

    def field(ws, off, width):
        return ws[off * Mp:(off + width) * Mp].view(Mp, width)

    same = [("x", 0, 4), ("de", 4, 32)] + [(f"h{l}", WS_H + H * l, H) for l in range(8)] + [("f", WS_F, H), ("g", WS_G, HH), ("sraw", WS_SRAW, 1)]
    for name, off, width in same:
        a, b = field(ws3, off, width), field(ws32, off, width)
        assert rel(a[:M], b[:M]) < RTOL, name
        assert float((a[:M] - b[:M]).abs().max()) <= 2e-5 * max(float(b[:M].abs().max()), 1.0), name
        assert torch.isfinite(a).all(), name
    for l in range(8):
        w0 = 30.0 if l == 0 else 1.0
        h, c = field(ws3, WS_H + H * l, H).double(), field(ws3, WS_C + H * l, H).double()
        assert float(((c / w0) ** 2 + h ** 2 - 1).abs().max()) < 1e-6, f"derivative of h{l}"
    gg, cg = field(ws3, WS_G, HH).double(), field(ws3, WS_CG, HH).double()
    assert float((cg ** 2 + gg ** 2 - 1).abs().max()) < 1e-6


def test_bf16x3_training_psnr_curve_matches_the_fp32_kernels():
    """120 optimizer steps (train.py:166-183: render, MSE, backward, Adam + LambdaLR) on a small analytic scene, NeRF
    model: the bf16x3 mode next to the fp32 kernels from the same weights, rays and stratified draws.  The first ten
    steps step for step (PSNR within 0.01 dB), afterwards the 20-step windows within 0.15 dB -- the bounds the fp32
    kernels are held to against the CPU oracle (test_training_psnr_curve_matches_the_oracle)."""
    from nerf import parallel
    from nerf.loss import mse_to_psnr
    from nerf.rendering import render_nerf
    R, S, steps, lr = 64, 32, 120, 5e-4
    o, d = rays(R, 900)
    tgt = 0.5 + 0.5 * torch.sin(3.0 * d + torch.tensor([0.0, 1.0, 2.0]))
    g = torch.Generator().manual_seed(901)
    us = [torch.rand(S, generator=g).to(DEV) for _ in range(steps)]
    od, dd, td = o.to(DEV), d.to(DEV), tgt.to(DEV)

    def run(precision):
        model, _ = make_model("nerf")
        model.set_training_precision(precision)
        flat = parallel.FlatParameters(model)
        opt = parallel.FusedAdam(flat, lr=lr)
        curve = []
        for it in range(steps):
            rgb = render_nerf(model, od, dd, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=8192, u=[us[it]])
            opt.zero_grad()
            loss = torch.nn.functional.mse_loss(rgb, td)
            loss.backward()
            opt.lr = lr * parallel.lr_lambda(it, lr, 300.0, 0.1, 1e-4)
            opt.step()
            curve.append(float(loss.detach()))
        return curve

    c3, c32 = run("bf16x3"), run("fp32")
    win = lambda c, k: mse_to_psnr(sum(c[k:k + 20]) / 20.0)
    gaps = [abs(win(c3, k) - win(c32, k)) for k in range(0, steps, 20)]
    early = max(abs(mse_to_psnr(c3[i]) - mse_to_psnr(c32[i])) for i in range(10))
    print(f"PSNR {mse_to_psnr(c32[0]):.2f} -> {win(c32, steps - 20):.2f} dB (fp32), {mse_to_psnr(c3[0]):.2f} -> "
          f"{win(c3, steps - 20):.2f} dB (bf16x3); gaps " + " ".join(f"{x:.3f}" for x in gaps) + f" dB; first 10 steps {early:.4f} dB")
    assert win(c3, steps - 20) > mse_to_psnr(c3[0]) + 6.0
    assert early < 0.01 and max(gaps) < 0.15, (early, gaps)


@pytest.mark.parametrize("tag", ["nerf", "siren"])
@pytest.mark.parametrize("per_ray", [False, True])
def test_bf16x3_training_forward_in_ray_mode(tag, per_ray):
    """Positions generated in the kernel (rays_o + t * rays_d, normalised), shared or per-ray depths, a sample count that is
    no multiple of the 16-sample tiles or the 128-sample workgroups: outputs and the first / last hidden activations
    against the fp32 forward on the same rays."""
    from nerf import _hip
    model, _ = make_model(tag)
    R, S = 7, 50
    o, d = rays(R, 77)
    g = torch.Generator().manual_seed(78)
    t = torch.sort(2.0 + 4.0 * torch.rand((R, S) if per_ray else (S,), generator=g), dim=-1)[0]
    o, d, t = o.to(DEV), d.to(DEV), t.to(DEV).contiguous()
    M = R * S
    kw = dict(rays_o=o, rays_d=d, t=t, S=S, near=2.0, far=6.0, train=True)
    c32, d32, ws32 = _hip.mlp_fwd(model.model_id, model.packed_weights(True), M, **kw)
    model.set_training_precision("bf16x3")
    c3, d3, ws3 = _hip.mlp_fwd(model.model_id, model.packed_weights(True), M, **kw)
    assert rel(c3, c32) < RTOL and rel(d3, d32) < RTOL
    Mp = -(-M // 64) * 64
    h0 = 96 if tag == "nerf" else 36
    last = 8 if tag == "nerf" else 7
    for l in (0, last):
        a = ws3[(h0 + 256 * l) * Mp:(h0 + 256 * (l + 1)) * Mp].view(Mp, 256)[:M]
        b = ws32[(h0 + 256 * l) * Mp:(h0 + 256 * (l + 1)) * Mp].view(Mp, 256)[:M]
        assert rel(a, b) < RTOL, f"h{l}"
