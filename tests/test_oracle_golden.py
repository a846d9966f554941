"""Pins oracle/nerf_oracle.py to the golden vectors the reference produced
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

T = torch.from_numpy


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
    np.testing.assert_allclose(colors.detach().numpy(), z["colors"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dens.detach().numpy(), z["density"], rtol=1e-6, atol=1e-7)
    ((colors * T(z["gc"])).sum() + (dens * T(z["gd"])).sum()).backward()
    _check_grads(z, params)


@pytest.mark.parametrize("wb", [True, False])
def test_g5_composite(golden, wb):
    z = golden("g5_composite.npz")
    c = T(z["colors"]).requires_grad_(True)
    s = T(z["densities"]).requires_grad_(True)
    deltas = T(z["deltas"])
    rgb = O.composite(c, s, deltas, wb)
    assert np.array_equal(rgb.detach().numpy(), z[f"wb{int(wb)}_rgb"])
    (rgb * T(z["grad_rgb"])).sum().backward()
    assert np.array_equal(c.grad.numpy(), z[f"wb{int(wb)}_dcolors"])
    assert np.array_equal(s.grad.numpy(), z[f"wb{int(wb)}_ddens"])
    w, _ = O.composite_weights(T(z["densities"]), deltas)
    assert np.array_equal(w.numpy(), z["weights"])
    # the zero-safe reverse recurrence (what the HIP backward implements) == autograd
    dcol, dsig = O.composite_backward_manual(T(z["colors"]), T(z["densities"]), deltas, wb, T(z["grad_rgb"]))
    np.testing.assert_allclose(dcol.numpy(), z[f"wb{int(wb)}_dcolors"], rtol=1e-5, atol=1e-7)
    ref = z[f"wb{int(wb)}_ddens"]
    assert np.abs(dsig.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("tag", ["nerf_S64_onechunk", "nerf_S256_chunk24", "nerf_S64_black",
                                 "siren_S64_onechunk", "siren_S256_chunk24"])
def test_g6_render_end_to_end(golden, tag):
    z = golden(f"g6_{tag}.npz")
    R, S, chunk, _, wb = (int(v) for v in z["cfg"])
    params, fwd = _model_case(tag.split("_")[0])
    for p in params.values():
        p.requires_grad_(True)
    rgb = O.render(lambda x, d: fwd(params, x, d), T(z["o"]), T(z["d"]), 2.0, 6.0, S,
                   [T(u) for u in z["us"]], white_background=bool(wb), chunk_size=chunk)
    np.testing.assert_allclose(rgb.detach().numpy(), z["rgb"], rtol=1e-6, atol=1e-7)
    loss = torch.nn.functional.mse_loss(rgb, T(z["target"]))
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-6 * float(z["loss"])
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
        assert abs(float(loss) - z["losses"][it]) <= 2e-5 * z["losses"][it], it
        grads = torch.autograd.grad(loss, list(params.values()))
        with torch.no_grad():
            for (n, p), g in zip(params.items(), grads):
                O.adam_step(p, g, m[n], v[n], it + 1, lr)
        assert abs(lr0 * O.lr_lambda(it + 1, lr0, 300.0, 0.1, 1e-4) - z["lrs"][it]) < 1e-12
    for n, p in params.items():
        assert abs(float(p.detach().norm()) - float(z["pnorm." + n])) <= 1e-5 * float(z["pnorm." + n]), n


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
