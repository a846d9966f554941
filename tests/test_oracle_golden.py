"""Pins oracle/nerf_oracle.py to the golden vectors the reference produced
(tests/golden/make_golden.py).  CPU only.

Two sources of vectors, every test runs against both:
* "live": when the reference is mounted (the build container only), make_golden.py is run once per session into a
  temporary directory, i.e. the reference and the oracle run on the SAME host -- the oracle is then held bit for bit
  where it restates elementwise arithmetic (sampling, positions, encoding, compositing and its gradients) and to 1e-6
  where matrix products are involved.  This is the pin proper.  Skipped where /root/reference does not exist.
* "committed": the files under tests/golden/ (what the GPU tests compare the kernels with).  They were written on
  another host; ATen's vectorised exp / sin and MKL's GEMM blocking differ by an ulp or two between CPU models
  (seen: the same AVX-512 build of torch on two EPYC generations disagrees in 3 of 72 composited channels by one
  ulp), so against these files the transcendental paths are held to 4 ulp / 5e-6 relative instead of bit for bit --
  still inside the 1e-5 bar of the GPU parity tests.  Pure IEEE add / mul / fma paths stay bit-exact for both sources.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

T = torch.from_numpy
REFERENCE = "/root/reference"


@pytest.fixture(scope="session")
def live_dir(tmp_path_factory):
    if not os.path.isdir(os.path.join(REFERENCE, "nerf")):
        pytest.skip("reference not mounted here: the committed vectors are the pin")
    out = tmp_path_factory.mktemp("golden_live")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, NERF_GOLDEN_OUT=str(out), PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(here, "golden", "make_golden.py")], check=True, env=env,
                   stdout=subprocess.DEVNULL, timeout=600)
    return str(out)


@pytest.fixture(params=["committed", "live"])
def golden(request):
    """(loader, strict): strict = the vectors come from this very host."""
    if request.param == "live":
        base = request.getfixturevalue("live_dir")
    else:
        base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    load = lambda name: np.load(os.path.join(base, name))    # noqa: E731
    load.strict = request.param == "live"
    return load


def _same(a, b, strict, ulps=4):
    """Bit for bit on the host that produced `b`; on any other host within `ulps` units in the last place of the value
    or of 1.0, whichever is larger (values far below 1 here are differences of nearly equal numbers, `1 - (1 - e)`
    with e close to 1, whose relative error is unbounded while their absolute error stays at the ulp of the operands)."""
    a, b = np.asarray(a), np.asarray(b)
    if strict:
        return np.array_equal(a, b)
    tol = ulps * 2.0 ** -23 * np.maximum(np.maximum(np.abs(a), np.abs(b)), 1.0)
    return bool((np.abs(a.astype(np.float64) - b) <= tol).all())


def test_g1_stratified_bit_exact(golden):
    z = golden("g1_stratified.npz")
    n = len([k for k in z.files if k.endswith("_cfg")])
    assert n == 7
    for ci in range(n):
        near, far, S, _ = z[f"c{ci}_cfg"]
        t = O.stratified_t(float(near), float(far), int(S), T(z[f"c{ci}_u"]))
        assert np.array_equal(t.numpy(), z[f"c{ci}_t"])
        # explicit restatement of the linspace law the HIP kernel implements
        assert np.array_equal(O.linspace_fma(float(near), float(far), int(S) + 1), z[f"c{ci}_edges"])


def test_g2_positions_bit_exact(golden):
    z = golden("g2_positions.npz")
    o, d = T(z["o"]), T(z["d"])
    for S in (64, 256):
        t = O.stratified_t(2.0, 6.0, S, T(z[f"S{S}_u"]))
        assert np.array_equal(O.sample_deltas(t).numpy(), z[f"S{S}_deltas"])
        pos = O.sample_positions(o, d, t)
        assert np.array_equal(pos.numpy(), z[f"S{S}_pos"])
        assert np.array_equal(O.normalize_positions(pos, 2.0, 6.0).numpy(), z[f"S{S}_posn"])


def test_g3_posenc_bit_exact(golden):
    z = golden("g3_posenc.npz")
    x = T(z["x"])
    assert np.array_equal(O.positional_encoding(x, 4).numpy(), z["L4"])
    assert np.array_equal(O.positional_encoding(x, 10).numpy(), z["L10"])


def _model_case(tag):
    if tag.startswith("nerf"):
        hidden = 64 if tag.endswith("h64") else 256
        return O.closed_form_params(O.nerf_param_shapes(hidden=hidden)), O.nerf_forward
    hidden = 64 if tag.endswith("h64") else 256
    return O.closed_form_params(O.siren_param_shapes(hidden=hidden), gains=O.SIREN_GAINS), O.siren_forward


def _check_grads(z, params, rtol=2e-5):
    for n, p in params.items():
        ref_norm = float(z["gnorm." + n])
        assert abs(float(p.grad.norm()) - ref_norm) <= rtol * ref_norm + 1e-12, n
        f = p.grad.reshape(-1)
        step = max(1, f.numel() // 97)
        sub = f[::step][:97].numpy()
        assert np.abs(sub - z["gsub." + n]).max() <= rtol * max(ref_norm / np.sqrt(f.numel()), 1e-12) * 50, n


@pytest.mark.parametrize("tag", ["nerf", "nerf_h64", "siren", "siren_h64"])
def test_g4_model_forward_and_grads(golden, tag):
    z = golden(f"g4_{tag}.npz")
    params, fwd = _model_case(tag)
    for p in params.values():
        p.requires_grad_(True)
    colors, dens = fwd(params, T(z["pts"]), T(z["dirs"]))
    rtol = 1e-6 if golden.strict else 5e-6
    np.testing.assert_allclose(colors.detach().numpy(), z["colors"], rtol=rtol, atol=1e-7)
    # densities are differences of large terms cut at zero: the absolute tolerance follows their scale
    np.testing.assert_allclose(dens.detach().numpy(), z["density"], rtol=rtol,
                               atol=1e-7 if golden.strict else 2e-6 * float(np.abs(z["density"]).max()))
    ((colors * T(z["gc"])).sum() + (dens * T(z["gd"])).sum()).backward()
    _check_grads(z, params)


@pytest.mark.parametrize("wb", [True, False])
def test_g5_composite(golden, wb):
    z = golden("g5_composite.npz")
    c = T(z["colors"]).requires_grad_(True)
    s = T(z["densities"]).requires_grad_(True)
    deltas = T(z["deltas"])
    rgb = O.composite(c, s, deltas, wb)
    assert _same(rgb.detach().numpy(), z[f"wb{int(wb)}_rgb"], golden.strict)
    (rgb * T(z["grad_rgb"])).sum().backward()
    assert _same(c.grad.numpy(), z[f"wb{int(wb)}_dcolors"], golden.strict)
    assert _same(s.grad.numpy(), z[f"wb{int(wb)}_ddens"], golden.strict)
    w, _ = O.composite_weights(T(z["densities"]), deltas)
    assert _same(w.numpy(), z["weights"], golden.strict)
    # the zero-safe reverse recurrence (what the HIP backward implements) == autograd
    dcol, dsig = O.composite_backward_manual(T(z["colors"]), T(z["densities"]), deltas, wb, T(z["grad_rgb"]))
    np.testing.assert_allclose(dcol.numpy(), z[f"wb{int(wb)}_dcolors"], rtol=1e-5, atol=1e-7)
    ref = z[f"wb{int(wb)}_ddens"]
    assert np.abs(dsig.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("tag", ["nerf_S64_onechunk", "nerf_S256_chunk24", "nerf_S64_black",
                                 "siren_S64_onechunk", "siren_S256_chunk24"])
@pytest.mark.parametrize("family", ["g6", "g6k"])   # g6k: rays chosen clear of every ReLU kink (make_golden_kinkfree.py)
def test_g6_render_end_to_end(golden, tag, family):
    if family == "g6k" and golden.strict:
        pytest.skip("make_golden_kinkfree.py picks its rays by this host's kink distances; only the committed set exists")
    z = golden(f"{family}_{tag}.npz")
    R, S, chunk, _, wb = (int(v) for v in z["cfg"])
    params, fwd = _model_case(tag.split("_")[0])
    for p in params.values():
        p.requires_grad_(True)
    rgb = O.render(lambda x, d: fwd(params, x, d), T(z["o"]), T(z["d"]), 2.0, 6.0, S,
                   [T(u) for u in z["us"]], white_background=bool(wb), chunk_size=chunk)
    rtol = 1e-6 if golden.strict else 5e-6
    np.testing.assert_allclose(rgb.detach().numpy(), z["rgb"], rtol=rtol, atol=1e-7)
    loss = torch.nn.functional.mse_loss(rgb, T(z["target"]))
    assert abs(float(loss.detach()) - float(z["loss"])) <= rtol * float(z["loss"])
    loss.backward()
    _check_grads(z, params)


@pytest.mark.parametrize("tag", ["nerf", "siren"])
def test_g7_adam_trajectory(golden, tag):
    z = golden(f"g7_{tag}.npz")
    R, S, steps = (int(v) for v in z["cfg"])
    params, fwd = _model_case(tag)
    for p in params.values():
        p.requires_grad_(True)
    m = {n: torch.zeros_like(p) for n, p in params.items()}
    v = {n: torch.zeros_like(p) for n, p in params.items()}
    lr0 = 5e-4
    for it in range(steps):
        lr = lr0 * O.lr_lambda(it, lr0, 300.0, 0.1, 1e-4)
        rgb = O.render(lambda x, d: fwd(params, x, d), T(z["o"]), T(z["d"]), 2.0, 6.0, S, [T(z["us"][it])])
        loss = torch.nn.functional.mse_loss(rgb, T(z["target"]))
        # Adam amplifies rounding-level gradient differences step by step (every parameter moves by ~lr whatever its
        # gradient's size): 2e-5 on the producing host, 1e-4 by the fifth step on another one
        assert abs(float(loss) - z["losses"][it]) <= (2e-5 if golden.strict else 1e-4) * z["losses"][it], it
        grads = torch.autograd.grad(loss, list(params.values()))
        with torch.no_grad():
            for (n, p), g in zip(params.items(), grads):
                O.adam_step(p, g, m[n], v[n], it + 1, lr)
        assert abs(lr0 * O.lr_lambda(it + 1, lr0, 300.0, 0.1, 1e-4) - z["lrs"][it]) < 1e-12
    for n, p in params.items():
        # bias vectors are small and every entry moves by ~lr per step: a rounding-level gradient difference that flips
        # one early update's direction shows at 2e-5 of a bias norm on a host other than the producing one
        assert abs(float(p.detach().norm()) - float(z["pnorm." + n])) <= (1e-5 if golden.strict else 5e-5) * float(z["pnorm." + n]), n


def test_sample_pdf_properties():
    """PARITY UNPINNED (absent from the reference): self-consistency only."""
    torch.manual_seed(0)
    R, B, N = 16, 63, 128
    bins = torch.linspace(2.0, 6.0, B + 1).expand(R, B + 1).contiguous()
    w = torch.rand(R, B)
    u = torch.rand(R, N)
    s, idx = O.sample_pdf(bins, w, u)
    assert s.min() >= 2.0 and s.max() <= 6.0
    assert idx.min() >= 1 and idx.max() <= B + 1
    us, _ = torch.sort(u, dim=1)
    ss, _ = O.sample_pdf(bins, w, us)
    assert (ss[:, 1:] >= ss[:, :-1] - 1e-6).all()


def test_compute_rays_and_pose():
    c2w = O.pose_spherical(30.0, -30.0, 4.0).astype(np.float32)
    ro, rd = O.compute_rays_np(8, 8, c2w, 0.5 * 8 / np.tan(0.5 * 0.6911112))
    assert ro.shape == (64, 3) and rd.shape == (64, 3)
    np.testing.assert_allclose(np.linalg.norm(rd, axis=1), 1.0, rtol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(ro, axis=1), 4.0, rtol=1e-6)


def test_configuration_1_cpu_plumbing(tmp_path):
    """The CPU-runnable configuration: a config_lego-style file at half resolution (400x400), 1024 rays per batch,
    64 coarse + 128 fine samples -- config parsing, pinhole rays of one pose, a shuffled 1024-ray batch, one coarse + fine
    render with both losses, gradients and an Adam step with the LambdaLR factor, all on the CPU restatement (the HIP path
    needs a GPU; this is the plumbing around it)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "nerf-pytorch_amd"))
    from nerf.data import BatchCursor
    from nerf.loss import mse_to_psnr
    from nerf.utils import parse_config
    cfg_path = tmp_path / "config_half_res.txt"
    cfg_path.write_text("dataset_path = ./datasets/lego   # not read here\nnum_random_rays = 1024\nchunk_size = 8192\n"
                        "num_samples = 64\nnum_fine_samples = 128\nlearning_rate = 5e-4\nnear = 2.0\nfar = 6.0\n"
                        "lr_decay = 150\nlr_decay_factor = 0.1\nlr_min = 1e-5\nmodel_type = nerf\n")
    cfg = parse_config(str(cfg_path))
    B, Nc, Nf = int(cfg["num_random_rays"]), int(cfg["num_samples"]), int(cfg["num_fine_samples"])
    near, far, lr = float(cfg["near"]), float(cfg["far"]), float(cfg["learning_rate"])
    assert (B, Nc, Nf, cfg["model_type"]) == (1024, 64, 128, "nerf")
    H = W = 400
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = O.compute_rays_np(H, W, O.pose_spherical(30.0, -30.0, 4.0).astype(np.float32), focal)
    assert ro.shape == (H * W, 3) and np.allclose(np.linalg.norm(rd, axis=1), 1.0, atol=1e-6)
    idx = BatchCursor(H * W, "cpu", torch.Generator().manual_seed(42)).next(B)
    o, d = torch.from_numpy(ro)[idx], torch.from_numpy(rd)[idx]
    g = torch.Generator().manual_seed(1)
    tgt = torch.rand(B, 3, generator=g)
    pc = O.closed_form_params(O.nerf_param_shapes())
    pf = {k: v.clone() for k, v in pc.items()}
    for p in list(pc.values()) + list(pf.values()):
        p.requires_grad_(True)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    rc, rf, w_f, t_all = O.render_hierarchical(lambda x, v: O.nerf_forward(pc, x, v), lambda x, v: O.nerf_forward(pf, x, v),
                                               o, d, near, far, Nc, Nf, torch.rand(Nc, generator=g),
                                               torch.rand(B, Nf, generator=g))
    assert rc.shape == (B, 3) and rf.shape == (B, 3) and t_all.shape == (B, Nc + Nf) and w_f.shape == (B, Nc + Nf)
    assert bool((t_all[:, 1:] >= t_all[:, :-1]).all()) and float(t_all.min()) >= near and float(t_all.max()) <= far
    loss = torch.nn.functional.mse_loss(rc, tgt) + torch.nn.functional.mse_loss(rf, tgt)
    allp = list(pc.values()) + list(pf.values())
    grads = torch.autograd.grad(loss, allp)
    assert all(torch.isfinite(gr).all() for gr in grads) and sum(float(gr.abs().sum()) for gr in grads) > 0
    factor = O.lr_lambda(0, lr, float(cfg["lr_decay"]), float(cfg["lr_decay_factor"]), float(cfg["lr_min"]))
    assert factor == 1.0
    before = float(allp[0].detach().norm())
    with torch.no_grad():
        for p, gr in zip(allp, grads):
            O.adam_step(p, gr, torch.zeros_like(p), torch.zeros_like(p), 1, lr * factor)
    assert float(allp[0].detach().norm()) != before
    assert 0.0 < mse_to_psnr(float(torch.nn.functional.mse_loss(rf, tgt).detach())) < 60.0


def test_three_term_expansion_properties():
    """The arithmetic of the opt-in bf16x3 kernels, restated (no counterpart in the reference): the three bf16 terms of
    an fp32 number reconstruct it exactly, every term product fits fp32, and a chain of layers formed from the six
    products per product stays as close to a float64 evaluation as plain fp32 arithmetic does."""
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-6, 1e-2, 1.0, 37.0, 1e5)])
    t0, t1, t2 = O.bf16_terms(x)
    assert torch.equal((t0.double() + t1.double() + t2.double()).float(), x)          # exact, not merely close
    assert torch.equal(t0 + t1 + t2, x)                                                # even summed in fp32
    for t in (t0, t1, t2):
        assert torch.equal(O.bf16_round(t), t)                                         # every term IS a bf16 number
    assert float((t1.abs() / x.abs()).max()) <= 2.0 ** -8 and float((t2.abs() / x.abs()).max()) <= 2.0 ** -16
    w = torch.randn(4096, generator=g)
    for a in O.bf16_terms(w):
        for b in (t0, t1, t2):
            pr = a[:4096].double() * b[:4096].double()
            assert torch.equal(pr.float().double(), pr)                                # 8 x 8 significant bits: exact in fp32
    # an 8-layer 256-wide ReLU chain against float64
    Ws = [torch.randn(256, 256, generator=g) * (2.0 / 256) ** 0.5 for _ in range(8)]
    bs = [torch.randn(256, generator=g) * 0.1 for _ in range(8)]
    h0 = torch.randn(512, 256, generator=g)
    ref, f32, x3 = h0.double(), h0, h0
    for W, b in zip(Ws, bs):
        ref = torch.relu(ref @ W.double().t() + b.double())
        f32 = torch.relu(f32 @ W.t() + b)
        x3 = torch.relu(O.linear_bf16x3(x3, W, b))
    e32 = float((f32.double() - ref).norm() / ref.norm())
    e3 = float((x3.double() - ref).norm() / ref.norm())
    assert e3 < 1e-6 and e3 <= 1.5 * e32, (e3, e32)
