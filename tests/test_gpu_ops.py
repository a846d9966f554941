"""The HIP kernels as registered PyTorch operators (torch.library, namespace `nerf`): schema, fake (shape-only)
implementation, autograd registration and eager-versus-traced agreement, checked with torch.library.opcheck for
the operators behind the reference's call sites (train.py:166-176 render_nerf, models.py:62-75 forward)."""
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(tag="nerf"):
    from nerf.models import NeRF, Siren
    if tag == "nerf":
        m, params = NeRF(), O.closed_form_params(O.nerf_param_shapes())
    else:
        m, params = Siren(), O.closed_form_params(O.siren_param_shapes(), gains=O.SIREN_GAINS)
    m.load_state_dict(params)
    return m.to(DEV)


def _rays(R, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(R, 3, generator=g)
    o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.15 * torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    return o.to(DEV), d.to(DEV)


def test_ops_are_registered():
    from nerf import ops  # noqa: F401  (registers the operators)
    for name in ("posenc", "composite", "composite_backward", "mlp", "mlp_backward", "render", "render_backward"):
        assert hasattr(torch.ops.nerf, name), name
    assert "Tensor[] params" in str(torch.ops.nerf.render.default._schema)


def test_opcheck_posenc():
    from nerf import ops  # noqa: F401
    x = torch.rand(100, 3, device=DEV) * 2 - 1
    torch.library.opcheck(torch.ops.nerf.posenc.default, (x, 10))
    torch.library.opcheck(torch.ops.nerf.posenc.default, (x[:, :2].contiguous(), 4))
    torch.library.opcheck(torch.ops.nerf.posenc.default, (x.clone().requires_grad_(True), 4))


def test_opcheck_composite():
    from nerf import ops  # noqa: F401
    g = torch.Generator().manual_seed(1)
    R, S = 70, 48
    colors = torch.rand(R, S, 3, generator=g).to(DEV).requires_grad_(True)
    dens = (torch.relu(torch.randn(R, S, generator=g)) * 5).to(DEV).requires_grad_(True)
    t = O.stratified_t(2.0, 6.0, S, torch.rand(S, generator=g))
    deltas = O.sample_deltas(t).to(DEV)
    torch.library.opcheck(torch.ops.nerf.composite.default, (colors, dens, deltas, True))
    torch.library.opcheck(torch.ops.nerf.composite.default, (colors.detach(), dens.detach(), deltas, False))
    torch.library.opcheck(torch.ops.nerf.composite_backward.default,
                          (colors.detach(), dens.detach(), deltas, True, torch.randn(R, 3, device=DEV)))


@pytest.mark.parametrize("tag", ["nerf", "siren"])
def test_opcheck_mlp(tag):
    from nerf import ops  # noqa: F401
    model = _model(tag)
    M = 128
    g = torch.Generator().manual_seed(2)
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).to(DEV)
    params = model.ordered_parameters()
    packed = model.packed_weights(True)
    torch.library.opcheck(torch.ops.nerf.mlp.default, (model.model_id, packed, pts, dirs, params, True, None))
    with torch.no_grad():
        torch.library.opcheck(torch.ops.nerf.mlp.default,
                              (model.model_id, packed, pts, dirs, [p.detach() for p in params], False, None))


def test_opcheck_render_and_gradients_match_the_module_path():
    from nerf import ops  # noqa: F401
    from nerf.rendering import render_nerf
    model = _model("nerf")
    R, S = 64, 32
    o, d = _rays(R, 3)
    u = torch.rand(S, generator=torch.Generator().manual_seed(4)).to(DEV)
    from nerf import _hip
    t, deltas = _hip.stratified_t(2.0, 6.0, S, u)
    params = model.ordered_parameters()
    packed = model.packed_weights(True)
    args = (model.model_id, packed, o, d, t, deltas, 2.0, 6.0, True, True, params, True, None)
    torch.library.opcheck(torch.ops.nerf.render.default, args)
    # the operator called directly gives what render_nerf gives, forward and parameter gradients
    out = torch.ops.nerf.render(*args)
    rgb = render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=DEV, u=[u])
    assert torch.equal(out[0], rgb)
    g1 = torch.autograd.grad(out[0].square().sum(), params)
    g2 = torch.autograd.grad(rgb.square().sum(), params)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert out[1].shape == (R, S) and out[2].shape == (R,) and out[3].shape == (R,)
