"""The benchmark configurations at their full sizes, where the CPU oracle would take minutes: checked through
size-independent properties plus an oracle comparison on a slice.
  * training batch in its headline shape: 4096 rays, 64 coarse + 128 fine samples, two networks;
  * one full 800x800 pose (640,000 rays) through the eval driver's render path;
  * the large-batch stress shape 262,144 rays x 256 samples in the opt-in bf16 mode against fp32."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = a.detach().cpu().double().reshape(-1)
    b = b.detach().cpu().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def rays(R, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(R, 3, generator=g)
    o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.15 * torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    return o.contiguous(), d.contiguous()


def nerf_pair(scale_fine=0.97):
    from nerf.models import NeRF
    pc = O.closed_form_params(O.nerf_param_shapes())
    pf = {k: v * scale_fine for k, v in pc.items()}
    coarse, fine = NeRF(), NeRF()
    coarse.load_state_dict(pc)
    fine.load_state_dict(pf)
    return coarse.to(DEV), fine.to(DEV), pc, pf


def test_training_batch_hierarchical_full_size():
    """4096 rays x (64 coarse + 128 fine), coarse + fine NeRF, both losses -- the shape the headline rate is
    quoted on.  Properties:
      * two runs are bit-identical (images and both networks' gradients: no atomics anywhere);
      * gradients are linear in the loss: grads(2 * loss) == 2 * grads(loss) bit for bit;
      * the inverse-CDF indices equal torch.searchsorted on the GPU's own CDF copied to the CPU (pure
        comparisons: bit-exact), and the fine depths are sorted with the coarse depths among them;
      * a 64-ray slice agrees with the oracle's coarse + fine render (coarse 1e-5; fine 2e-4 because its
        depths went through the re-sampling, see test_hierarchical_render_vs_oracle)."""
    from nerf import _hip
    from nerf.rendering import render_nerf_hierarchical
    coarse, fine, pc, pf = nerf_pair()
    R, Nc, Nf = 4096, 64, 128
    o, d = rays(R, 131)
    g = torch.Generator().manual_seed(132)
    u_c, u_f = torch.rand(Nc, generator=g), torch.rand(R, Nf, generator=g)
    tgt = torch.rand(R, 3, generator=g).to(DEV)
    od, dd, ucd, ufd = o.to(DEV), d.to(DEV), u_c.to(DEV), u_f.to(DEV)

    def run(scale):
        coarse.zero_grad(set_to_none=True)
        fine.zero_grad(set_to_none=True)
        rc, rf = render_nerf_hierarchical(coarse, fine, od, dd, 2.0, 6.0, Nc, Nf, device=DEV, chunk_size=8192,
                                          u=[ucd], u_fine=[ufd])
        ((torch.nn.functional.mse_loss(rc, tgt) + torch.nn.functional.mse_loss(rf, tgt)) * scale).backward()
        grads = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.parameters()])
        return rc.detach(), rf.detach(), grads

    rc1, rf1, g1 = run(1.0)
    rc2, rf2, g2 = run(2.0)
    rc3, rf3, g3 = run(1.0)
    assert torch.equal(rc1, rc3) and torch.equal(rf1, rf3) and torch.equal(g1, g3)
    assert torch.equal(g2, 2 * g1)
    n_c = sum(p.numel() for p in coarse.parameters())
    assert torch.isfinite(g1).all() and float(g1[:n_c].abs().max()) > 0 and float(g1[n_c:].abs().max()) > 0
    for img in (rc1, rf1):
        assert float(img.min()) >= -1e-5 and float(img.max()) <= 1 + 1e-5

    # the re-sampling step on its own, as render_nerf_hierarchical chains it
    with torch.no_grad():
        t_c, deltas_c = _hip.stratified_t(2.0, 6.0, Nc, ucd)
        colors, dens, _ = _hip.mlp_fwd(coarse.model_id, coarse.packed_weights(False), R * Nc, rays_o=od, rays_d=dd, t=t_c,
                                       S=Nc, near=2.0, far=6.0)
        _, w_c, _, _ = _hip.composite_fwd(colors.view(R, Nc, 3), dens.view(R, Nc), deltas_c, True, want_weights=True)
        cdf = _hip.weights_to_cdf(w_c[:, 1:-1].contiguous())
        t_mid = (0.5 * (t_c[1:] + t_c[:-1])).expand(R, -1).contiguous()
        t_f, idx = _hip.sample_pdf(t_mid, cdf, ufd)
        t_all, deltas_all = _hip.merge_samples(t_c, t_f)
    idx_ref = torch.searchsorted(cdf.cpu().contiguous(), u_f.contiguous(), right=True)
    assert torch.equal(idx.cpu().long(), idx_ref)
    assert bool((t_all[:, 1:] >= t_all[:, :-1]).all()) and t_all.shape == (R, Nc + Nf)
    assert bool((cdf[:, 1:] >= cdf[:, :-1]).all()) and float((cdf[:, -1] - 1).abs().max()) < 1e-5
    assert float(t_f.min()) >= float(t_mid.min()) - 1e-6 and float(t_f.max()) <= float(t_mid.max()) + 1e-6
    assert bool((deltas_all[:, -1] == 1e10).all())

    fwd = O.nerf_forward
    with torch.no_grad():
        rc_ref, rf_ref, _, _ = O.render_hierarchical(lambda x, v: fwd(pc, x, v), lambda x, v: fwd(pf, x, v), o[:64], d[:64],
                                                     2.0, 6.0, Nc, Nf, u_c, u_f[:64])
    assert rel(rc1[:64], rc_ref) < RTOL
    assert rel(rf1[:64], rf_ref) < 2e-4


def test_full_image_render_800x800():
    """One 800x800 pose = 640,000 rays x 256 samples through the render path of the eval driver (device ray
    generation, chunked no-grad render, uint8 quantisation).  Properties:
      * chunking is invisible: with the same draw per chunk, chunk sizes 8192 and 65536 give the same
        image bit for bit, and any scan line rendered alone equals its rows of the full image;
      * the quantised frame is the truncation (clip(rgb,0,1)*255).astype(uint8) of eval.py:149-150;
      * the rays equal the oracle's compute_rays (origins bit for bit, directions to 5e-7), and 48 pixels spread
        over the image agree with the oracle's render."""
    sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
    import eval as eval_driver
    from nerf.data import compute_rays_device
    from nerf.models import NeRF
    from nerf.rendering import render_nerf
    params = O.closed_form_params(O.nerf_param_shapes())
    model = NeRF()
    model.load_state_dict(params)
    model = model.to(DEV).eval()
    H = W = 800
    S = 256
    focal = 0.5 * W / math.tan(0.5 * 0.6911112)
    c2w = eval_driver.pose_spherical(35.0, -30.0, 4.0)
    ro, rd = compute_rays_device(H, W, torch.from_numpy(c2w)[None].to(DEV), focal)
    ro, rd = ro[0], rd[0]
    assert ro.shape == (H * W, 3)
    u = torch.rand(S, generator=torch.Generator().manual_seed(77)).to(DEV)
    with torch.no_grad():
        img_a = render_nerf(model, ro, rd, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=8192, u=[u] * 79)
        img_b = render_nerf(model, ro, rd, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=65536, u=[u] * 10)
        assert img_a.shape == (H * W, 3) and torch.equal(img_a, img_b)
        row = 417
        line = render_nerf(model, ro[row * W:(row + 1) * W], rd[row * W:(row + 1) * W], 2.0, 6.0, num_samples=S,
                           device=DEV, chunk_size=8192, u=[u])
        assert torch.equal(line, img_a[row * W:(row + 1) * W])
    assert torch.isfinite(img_a).all() and float(img_a.min()) >= -1e-5 and float(img_a.max()) <= 1 + 1e-5
    frame = eval_driver.quantise(img_a.reshape(H, W, 3)).cpu().numpy()
    want = (np.clip(img_a.reshape(H, W, 3).cpu().numpy(), 0.0, 1.0) * 255).astype(np.uint8)
    assert frame.dtype == np.uint8 and frame.shape == (H, W, 3) and np.array_equal(frame, want)
    # rays against the oracle's restatement of compute_rays (its normalisation differs in the last bit, which the
    # positional encoding multiplies by up to 2^9: the render slice below is therefore fed the device's own rays)
    ro_ref, rd_ref = O.compute_rays_np(H, W, c2w, focal)
    assert np.array_equal(ro.cpu().numpy(), ro_ref) and float(np.abs(rd.cpu().numpy() - rd_ref).max()) <= 5e-7
    pick = (torch.arange(0, H * W, H * W // 48)[:48] + 123).to(DEV)
    with torch.no_grad():
        ref = O.render(lambda x, v: O.nerf_forward(params, x, v), ro[pick].cpu(), rd[pick].cpu(), 2.0, 6.0, S, [u.cpu()])
    assert rel(img_a[pick], ref) < RTOL


def _lively(model, gain, density_bias=0.5):
    """Default-initialised networks render almost empty space; scale the weights and lift the density bias so
    that the two precisions have structure to disagree on."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("weight"):
                p.mul_(gain)
        model.block2[-1].bias[-1] += density_bias
    return model


@pytest.mark.parametrize("seed,gain,min_psnr,q9999,max_bad_rays", [(0, 1.5, 70.0, 2e-3, 0), (3, 3.0, 55.0, 2e-2, 32)])
def test_bf16_large_batch_tolerance(seed, gain, min_psnr, q9999, max_bad_rays):
    """262,144 rays x 256 samples, the opt-in bf16 inference mode against the fp32 kernels on the same rays and
    draws (no counterpart in the reference, which is fp32 throughout: parity unpinned).  Stated tolerance of the
    mode at this shape (measured values in profiles/r1_bf16_tolerance.json: 80.8 dB / 3.4e-4 / 0 rays at gain
    1.5; 65.1 dB / 9e-3 / 4 rays at gain 3): PSNR between the renders above 70 dB / 55 dB, 99.99 % of the channel
    values within 2e-3 / 2e-2, at most 0 / 32 of the 262,144 rays with a channel off by more than 3e-2, mean
    absolute deviation below 1e-3."""
    from nerf.models import NeRF
    from nerf.rendering import render_nerf
    R, S, chunk = 262144, 256, 32768
    o, d = rays(R, 7)
    o, d = o.to(DEV), d.to(DEV)
    torch.manual_seed(seed)
    model = _lively(NeRF().to(DEV), gain)
    u = [torch.rand(S, generator=torch.Generator().manual_seed(100 + seed)).to(DEV)] * (R // chunk)
    with torch.no_grad():
        ref = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=chunk, u=u)
        model.set_inference_precision("bf16")
        out = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=chunk, u=u)
        model.set_inference_precision("fp32")
    assert out.shape == (R, 3) and torch.isfinite(out).all()
    err = (out - ref).abs().float()
    psnr = -10.0 * math.log10(max(float(((out - ref) ** 2).mean()), 1e-20))
    flat = err.reshape(-1)
    q = float(flat.kthvalue(int(0.9999 * flat.numel()))[0])
    bad = int((err.max(dim=1)[0] > 3e-2).sum())
    print(f"bf16 vs fp32 at {R} x {S}: PSNR {psnr:.1f} dB, mean abs {float(err.mean()):.2e}, q99.99 {q:.2e}, "
          f"rays off by > 3e-2: {bad}, image std {float(ref.std()):.3f}")
    assert float(ref.std()) > 0.01                       # the render has structure
    assert psnr > min_psnr and float(err.mean()) < 1e-3 and q < q9999 and bad <= max_bad_rays


@pytest.mark.parametrize("tag", ["nerf", "siren"])
def test_training_psnr_curve_matches_the_oracle(tag):
    """PSNR parity over a short training run: 120 optimizer steps (train.py:166-183: render, MSE, backward, Adam +
    LambdaLR) on a small analytic scene, the HIP path next to the CPU oracle with the same initial weights, rays and
    stratified draws.  The trajectories are chaotic in the last bits (Adam divides by sqrt(v)): the first ten steps are
    compared step for step (PSNR, nerf/loss.py mse_to_psnr, within 0.01 dB), after that as curves -- PSNR of the mean loss
    of every 20-step window, within 0.15 dB for NeRF and within the run's own sensitivity for Siren (see below) -- and both
    must have gained more than 6 dB."""
    from nerf import parallel
    from nerf.loss import mse_to_psnr
    from nerf.models import NeRF, Siren
    from nerf.rendering import render_nerf
    is_nerf = tag == "nerf"
    shapes = O.nerf_param_shapes() if is_nerf else O.siren_param_shapes()
    params = O.closed_form_params(shapes, gains=None if is_nerf else O.SIREN_GAINS)
    model = (NeRF if is_nerf else Siren)()
    model.load_state_dict(params)
    model = model.to(DEV)
    fwd = O.nerf_forward if is_nerf else O.siren_forward
    R, S, steps, lr = 64, 32, 120, 5e-4
    o, d = rays(R, 900)
    tgt = 0.5 + 0.5 * torch.sin(3.0 * d + torch.tensor([0.0, 1.0, 2.0]))      # a smooth colour per direction
    g = torch.Generator().manual_seed(901)
    us = [torch.rand(S, generator=g) for _ in range(steps)]

    plist = [p.clone().requires_grad_(True) for p in params.values()]
    pdict = dict(zip(params.keys(), plist))
    m = [torch.zeros_like(p) for p in plist]
    v = [torch.zeros_like(p) for p in plist]
    ref_curve = []
    for it in range(steps):
        rgb = O.render(lambda x, dd: fwd(pdict, x, dd), o, d, 2.0, 6.0, S, [us[it]])
        loss = torch.nn.functional.mse_loss(rgb, tgt)
        grads = torch.autograd.grad(loss, plist)
        step_lr = lr * O.lr_lambda(it, lr, 300.0, 0.1, 1e-4)
        with torch.no_grad():
            for p, gr, mm, vv in zip(plist, grads, m, v):
                O.adam_step(p, gr, mm, vv, it + 1, step_lr)
        ref_curve.append(float(loss.detach()))

    flat = parallel.FlatParameters(model)
    opt = parallel.FusedAdam(flat, lr=lr)
    od, dd, td = o.to(DEV), d.to(DEV), tgt.to(DEV)
    curve = []
    for it in range(steps):
        rgb = render_nerf(model, od, dd, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=8192, u=[us[it].to(DEV)])
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(rgb, td)
        loss.backward()
        opt.lr = lr * parallel.lr_lambda(it, lr, 300.0, 0.1, 1e-4)
        opt.step()
        curve.append(float(loss.detach()))
    # windows of 20 steps: the per-step loss also moves with each step's stratified draw
    win = lambda c, k: mse_to_psnr(sum(c[k:k + 20]) / 20.0)
    gaps = [abs(win(curve, k) - win(ref_curve, k)) for k in range(0, steps, 20)]
    early = max(abs(mse_to_psnr(curve[i]) - mse_to_psnr(ref_curve[i])) for i in range(10))
    print(f"{tag}: PSNR {mse_to_psnr(ref_curve[0]):.2f} -> {win(ref_curve, steps - 20):.2f} dB (oracle), "
          f"{mse_to_psnr(curve[0]):.2f} -> {win(curve, steps - 20):.2f} dB (HIP); gaps per 20-step window "
          + " ".join(f"{x:.3f}" for x in gaps) + f" dB; first 10 steps {early:.4f} dB")
    assert win(curve, steps - 20) > mse_to_psnr(curve[0]) + 6.0 and win(ref_curve, steps - 20) > mse_to_psnr(ref_curve[0]) + 6.0
    assert early < 0.01                       # before the trajectories decorrelate: step for step
    # afterwards: the same curve, as far as the run itself is reproducible.  tests/diag/psnr_sensitivity.py perturbs the
    # ORACLE's initial weights by 1e-7 relative: NeRF's window PSNRs then move by <= 0.04 dB, Siren's (which reaches
    # 37 dB, MSE 2e-4, within 100 steps) by 0.01 / 0.4 / 1.8 / 3.6 dB from the second window on.
    if is_nerf:
        assert max(gaps) < 0.15, gaps        # measured 0.007 .. 0.034
    else:
        # measured 0.000 0.041 0.153 0.463 1.562 0.898 (polynomial sine epilogues), 0.000 0.174 0.164 0.023 0.833 0.880
        # (table form): the second window already moves by 0.4 dB under a 1e-7 perturbation
        assert gaps[0] < 0.15 and gaps[1] < 0.5 and max(gaps) < 5.0, gaps


@pytest.mark.parametrize("seed,gain", [(0, 1.5), (3, 3.0)])
def test_bf16x3_large_batch_matches_fp32(seed, gain):
    """The same 262,144 x 256 stress shape through the three-term mode: its render and the fp32 kernels' differ by fp32
    accumulation rounding only -- relative deviation of the image below the fp32 parity bar, no channel off by 1e-4
    (lively networks: a rounding-level density difference is amplified by the compositing of grazing rays)."""
    from nerf.models import NeRF
    from nerf.rendering import render_nerf
    R, S, chunk = 262144, 256, 32768
    o, d = rays(R, 7)
    o, d = o.to(DEV), d.to(DEV)
    torch.manual_seed(seed)
    model = _lively(NeRF().to(DEV), gain)
    u = [torch.rand(S, generator=torch.Generator().manual_seed(100 + seed)).to(DEV)] * (R // chunk)
    with torch.no_grad():
        ref = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=chunk, u=u)
        model.set_inference_precision("bf16x3")
        out = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=chunk, u=u)
        model.set_inference_precision("fp32")
    err = (out - ref).abs().float()
    relerr = float((out - ref).norm() / ref.norm())
    print(f"bf16x3 vs fp32 at {R} x {S}: relative {relerr:.2e}, max abs {float(err.max()):.2e}, image std {float(ref.std()):.3f}")
    assert float(ref.std()) > 0.01
    assert relerr < RTOL and float(err.max()) < 1e-4


@pytest.mark.parametrize("tag", ["nerf", "siren"])
def test_full_size_gradients_three_ways(tag):
    """1,048,576 MLP evaluations (4096 rays x 256): every parameter-gradient tensor of the fused kernels against
    two INDEPENDENT computations of the same gradients on the same inputs, 1e-5 per tensor --
      * the layer-wise kernels of csrc/linear.hip (`model._layerwise`: other tiling, other summation order, no
        shared code with the fused forward / data-gradient / weight-gradient kernels);
      * the oracle run in float64 on the device (pure torch; nerf/models.py:62-75 / :185-203 restated).
    Run-to-run equality (test_full_size_properties) would not notice a corruption that repeats; these do.
    Samples whose float64 pre-activation lies within 1e-5 of a ReLU kink are dropped first (about 12 in 1000 for
    NeRF): the decision there differs between fp32 implementations and moves a whole gradient path -- ONE flipped
    decision among 1,048,576 samples shows as 1.3e-4 in every tensor upstream of it (tests/diag/layerwise_rows.py), and
    at the 3e-6 margin of the small-size tests the layer-wise kernels still flip one here."""
    from nerf.models import NeRF, Siren
    if tag == "nerf":
        model, params, fwd = NeRF(), O.closed_form_params(O.nerf_param_shapes()), O.nerf_forward
    else:
        model, params, fwd = Siren(), O.closed_form_params(O.siren_param_shapes(), gains=O.SIREN_GAINS), O.siren_forward
    model.load_state_dict(params)
    model = model.to(DEV)
    M = 4096 * 256
    g = torch.Generator().manual_seed(777)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.randn(M, 3, generator=g)
    dirs = (dirs / dirs.norm(dim=1, keepdim=True)).to(DEV)
    gc, gd = torch.randn(M, 3, generator=g).to(DEV), (torch.randn(M, generator=g) * 0.1).to(DEV)
    p64 = {k: v.to(DEV).double() for k, v in params.items()}
    with torch.no_grad():
        _, _, acts = fwd(p64, pts.double(), dirs.double(), return_acts=True)
        keep = O.relu_kink_distance(acts) > 1e-5
        del acts
    assert int(keep.sum()) >= int(0.95 * M)
    pts, dirs, gc, gd = (x[keep].contiguous() for x in (pts, dirs, gc, gd))

    for p in p64.values():
        p.requires_grad_(True)
    c64, d64 = fwd(p64, pts.double(), dirs.double())
    ((c64 * gc.double()).sum() + (d64 * gd.double()).sum()).backward()
    ref = {k: p.grad.clone() for k, p in p64.items()}
    c64, d64 = c64.detach(), d64.detach()
    for p in p64.values():
        p.grad = None
    torch.cuda.empty_cache()

    def grads(call):
        model.zero_grad(set_to_none=True)
        colors, dens = call(pts, dirs)
        ((colors * gc).sum() + (dens * gd).sum()).backward()
        return colors.detach(), dens.detach(), {n: p.grad.clone() for n, p in model.named_parameters()}

    cf, df, gf = grads(model)                 # fused kernels
    cl, dl, gl = grads(model._layerwise)      # one dense kernel per layer
    torch.cuda.empty_cache()

    def r(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))

    assert r(cf, c64) < RTOL and r(df, d64) < RTOL and r(cl, c64) < RTOL and r(dl, d64) < RTOL
    worst = {}
    for n in gf:
        e = (r(gf[n], ref[n]), r(gl[n], ref[n]), r(gf[n], gl[n]))
        worst[n] = e
    bad = {n: e for n, e in worst.items() if max(e) >= RTOL}
    assert not bad, bad
    print(f"[three ways {tag}] worst fused-vs-fp64 {max(e[0] for e in worst.values()):.2e}, layer-wise-vs-fp64 "
          f"{max(e[1] for e in worst.values()):.2e}, fused-vs-layer-wise {max(e[2] for e in worst.values()):.2e}")
