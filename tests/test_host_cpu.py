"""CPU-only checks of the host side: the C-ABI library loads and exports every declared
symbol, argument validation fails loudly, nothing silently falls back to a CPU path."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nerf import _hip
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(_hip.lib, s), f"libnerf_hip.so does not export {s}"
        assert s in _hip.EXPORTS, f"{s} declared in nerf_hip.h but not bound in nerf/_hip.py"
    assert _hip.lib.nerf_abi_version() == 10


def test_sizes_match_reference_parameter_counts():
    from nerf import _hip
    from nerf.models import NeRF, Siren
    assert _hip.param_floats(_hip.MODEL_NERF) == 661636 == sum(p.numel() for p in NeRF().parameters())
    assert _hip.param_floats(_hip.MODEL_SIREN) == 564356 == sum(p.numel() for p in Siren().parameters())
    assert _hip.lib.nerf_mlp_fwd_workspace_floats(0, 100) == 128 * (64 + 32 + 9 * 256 + 256 + 128 + 1 + 72)
    assert _hip.lib.nerf_mlp_fwd_workspace_floats(0, -1) == 0
    assert _hip.lib.nerf_mlp_packed_floats(7) == 0


def test_state_dict_keys_match_reference_layout():
    from nerf.models import NeRF, Siren
    from oracle import nerf_oracle as O
    assert [(k, tuple(v.shape)) for k, v in NeRF().state_dict().items()] == O.nerf_param_shapes()
    assert [(k, tuple(v.shape)) for k, v in Siren().state_dict().items()] == O.siren_param_shapes()


def test_argument_errors_are_reported_without_a_gpu():
    from nerf import _hip
    lib = _hip.lib
    assert lib.nerf_stratified_t(2.0, 6.0, 0, None, None, None, None) == -1
    assert b"S=0" in lib.nerf_last_error()
    assert lib.nerf_stratified_t(2.0, 6.0, 8, None, None, None, None) == -2
    assert lib.nerf_composite_fwd(None, None, None, None, 4, 2000, 0, 1, None, None, None, None, None) == -1
    assert lib.nerf_mlp_fwd(5, None, 10, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None, None) == -1
    assert lib.nerf_posenc_fwd(None, 10, 9, 4, None, None) == -1
    assert lib.nerf_mlp_pack(0, None, None, None) == -2
    # zero-sized work is a no-op that needs no pointers
    assert lib.nerf_composite_fwd(None, None, None, None, 0, 64, 0, 1, None, None, None, None, None) == 0
    assert lib.nerf_sample_positions(None, None, None, 0, 64, 0, 2.0, 6.0, None, None) == 0


def test_argument_errors_of_the_optional_paths():
    """bf16 mode and layer-wise path: sizes and argument checks that need no GPU."""
    from nerf import _hip
    lib = _hip.lib
    assert lib.nerf_mlp_packed_bf16_bytes(_hip.MODEL_NERF) == 43 * 32768 + 86 * 32 * 4
    assert lib.nerf_mlp_packed_bf16_bytes(_hip.MODEL_SIREN) == 0
    assert lib.nerf_mlp_fwd_bf16(_hip.MODEL_SIREN, None, 8, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None) == -1
    assert b"NeRF model only" in lib.nerf_last_error()
    assert lib.nerf_mlp_fwd_bf16(_hip.MODEL_NERF, None, 8, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None) == -2
    # three-term mode: three 1 KiB units per weight fragment (16 rows x 32 k), chunks of 24 units, 16 bias floats per output tile
    assert lib.nerf_mlp_packed_bf16x3_bytes(_hip.MODEL_NERF) == 163 * 24576 + 170 * 16 * 4
    assert lib.nerf_mlp_packed_bf16x3_bytes(_hip.MODEL_SIREN) == 141 * 24576 + 154 * 16 * 4      # both models
    assert lib.nerf_mlp_packed_bf16x3_bytes(7) == 0
    assert lib.nerf_mlp_fwd_bf16x3(7, None, 8, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None) == -1
    assert lib.nerf_mlp_fwd_bf16x3(_hip.MODEL_SIREN, None, 8, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None) == -2
    assert lib.nerf_mlp_fwd_bf16x3(_hip.MODEL_NERF, None, 8, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None) == -2
    assert lib.nerf_mlp_fwd_bf16x3(_hip.MODEL_NERF, None, 0, None, None, None, None, None, 0, 0, 0.0, 1.0, None, None, None) == 0
    assert lib.nerf_mlp_pack_bf16x3(7, None, None, None) == -1
    assert lib.nerf_mlp_pack_bf16x3(_hip.MODEL_SIREN, None, None, None) == -2
    assert lib.nerf_linear_fwd(None, 4, None, None, 10, 8, 3, 0, 1.0, 1.0, None, 3, None, None) == -1   # ldx < K
    assert lib.nerf_linear_fwd(None, 8, None, None, 10, 8, 3, 9, 1.0, 1.0, None, 3, None, None) == -1   # activation
    assert lib.nerf_linear_fwd(None, 8, None, None, 0, 8, 3, 0, 1.0, 1.0, None, 3, None, None) == 0     # empty batch
    assert lib.nerf_linear_bwd_workspace_floats(100, 8, 3) == 100 * 3 + 1 * (3 * 8 + 3)
    assert lib.nerf_linear_bwd_workspace_floats(-1, 8, 3) == 0


def test_no_cpu_fallback():
    from nerf.models import NeRF
    from nerf.rendering import render_nerf, composite_volume
    from nerf.encoding import positional_encoding
    m = NeRF()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(4, 3), torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        render_nerf(m, torch.zeros(4, 3), torch.zeros(4, 3), 2.0, 6.0, num_samples=8, device="cpu")
    with pytest.raises(RuntimeError, match="HIP device only"):
        positional_encoding(torch.zeros(4, 3), 4)
    with pytest.raises(RuntimeError, match="HIP device only"):
        composite_volume(torch.zeros(2, 4, 3), torch.zeros(2, 4), torch.zeros(4), True)


def test_unsupported_architectures_raise():
    from nerf.models import NeRF, Siren
    with pytest.raises(NotImplementedError):
        NeRF(hidden_dim=64)._check_supported()
    with pytest.raises(NotImplementedError):
        Siren(num_layers=4)._check_supported()
    NeRF()._check_supported()
    Siren()._check_supported()


def test_siren_init_law():
    """SirenLayer draws weight and bias from U(+-1/in) (first) or U(+-sqrt(6/in)/w0)."""
    import math
    from nerf.models import Siren
    torch.manual_seed(0)
    s = Siren()
    assert s.block1[0].layer.weight.abs().max() <= 1 / 3 + 1e-7
    assert s.block1[0].layer.bias.abs().max() <= 1 / 3 + 1e-7
    b = math.sqrt(6 / 256)
    assert s.block1[3].layer.weight.abs().max() <= b + 1e-7
    assert 0.9 * b < s.block1[3].layer.weight.abs().max()


def test_sincos_host_sweep(tmp_path):
    """The kernels' sine/cosine (csrc/sincos.h) swept on the host against double precision."""
    import subprocess
    exe = tmp_path / "sweep"
    subprocess.run(["g++", "-O2", "-mfma", "-o", str(exe), os.path.join(ROOT, "tests", "host", "sincos_sweep.cpp")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    m = re.search(r"^max_abs ([0-9.e+-]+)", out.stdout, re.M)
    assert float(m.group(1)) < 2.5e-7
    m = re.search(r"^table_max_abs ([0-9.e+-]+)", out.stdout, re.M)     # the table form of the sine epilogues
    assert float(m.group(1)) < 1.0e-7


def test_lr_schedule_and_shard_helpers():
    from nerf import parallel
    from oracle import nerf_oracle as O
    for s in (0, 10, 299999):
        assert parallel.lr_lambda(s, 5e-4, 300.0, 0.1, 1e-4) == O.lr_lambda(s, 5e-4, 300.0, 0.1, 1e-4)
    assert parallel.shard_slice(32768, 3, 8) == slice(12288, 16384)
    with pytest.raises(ValueError):
        parallel.shard_slice(10, 0, 3)


def test_flat_bucket_slices_for_several_networks():
    """parallel.FlatParameters over a container: parameters become views of one buffer in state_dict order and
    every fused network gets the slice of the gradient bucket that holds its own parameters (host logic only)."""
    import torch
    from nerf import parallel
    from nerf.models import NeRF, Siren
    pair = torch.nn.ModuleDict({"coarse": NeRF(), "fine": Siren()})
    before = {k: v.clone() for k, v in pair.state_dict().items()}
    flat = parallel.FlatParameters(pair)
    n_c = sum(p.numel() for p in pair["coarse"].parameters())
    n_f = sum(p.numel() for p in pair["fine"].parameters())
    assert (n_c, n_f) == (661636, 564356) and flat.flat.numel() == n_c + n_f
    assert pair["coarse"]._grad_sink.data_ptr() == flat.grad.data_ptr() and pair["coarse"]._grad_sink.numel() == n_c
    assert pair["fine"]._grad_sink.data_ptr() == flat.grad.data_ptr() + 4 * n_c and pair["fine"]._grad_sink.numel() == n_f
    for k, v in pair.state_dict().items():               # values unchanged, storage shared with the bucket
        assert torch.equal(v, before[k])
    off = 0
    for p in pair.parameters():
        assert p.data_ptr() == flat.flat.data_ptr() + 4 * off and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * off
        off += p.numel()
    flat.grad.fill_(1.0)
    flat.zero_grad()
    assert float(flat.grad.abs().max()) == 0.0 and len(flat.fused) == 2


def test_eval_pose_sharding_covers_every_pose_once():
    """Inference over N GPUs shards the camera path by pose with no collective: 200 poses over 8 ranks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("eval_driver", os.path.join(ROOT, "nerf-pytorch_amd", "eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    for n, world in ((200, 8), (40, 3), (5, 8), (0, 2)):
        seen = sorted(i for r in range(world) for i in ev.poses_for_rank(n, r, world))
        assert seen == list(range(n))
        sizes = [len(ev.poses_for_rank(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        ev.poses_for_rank(10, 3, 3)
