"""nerf_linear_bwd_noalloc (the entry point that allocates nothing) against nerf_linear_bwd, through the
C ABI: same kernels in the same order, the workspace replaced by the library's built-in scratch, so the
results must be bit-identical.  Two calls in a row, the second with a smaller input (the scratch keeps the
first call's contents), then the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(hip, M, K, N, act, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    dy = torch.randn(M, N, generator=g).to(DEV)
    pre, post = (3.0, 1.0) if act == hip.ACT_SIN else (1.5, 2.0) if act in (hip.ACT_RELU, hip.ACT_SIGMOID) else (1.0, 1.0)
    y, aux = hip.linear_fwd(x, W, b, act, pre, post)
    ref = hip.linear_bwd(dy, y, aux, act, pre, post, x, W, True, True)
    dx, dW, db = torch.full_like(x, float("nan")), torch.full_like(W, float("nan")), torch.full_like(b, float("nan"))
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    hip.linear_bwd_noalloc(dy, y, aux, act, pre, post, x, W, dx, dW, db)
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before   # torch's allocator was not touched either
    torch.cuda.synchronize()
    for got, want, name in ((dx, ref[0], "dx"), (dW, ref[1], "dW"), (db, ref[2], "db")):
        assert torch.equal(got, want), (name, M, K, N, act)


def test_noalloc_matches_workspace_entry_point_twice():
    from nerf import _hip as hip
    _case(hip, 20000, 67, 129, hip.ACT_SIN, 1)     # several partial slabs (Z = 9), odd sizes
    _case(hip, 300, 64, 64, hip.ACT_RELU, 2)       # smaller second call on the scratch the first one left
    _case(hip, 70000, 63, 64, hip.ACT_RELU, 3)     # hidden_dim = 64 first layer at a training-batch size
    _case(hip, 1, 3, 5, hip.ACT_NONE, 4)
    # optional outputs skipped
    g = torch.Generator().manual_seed(5)
    x, W, dy = (torch.randn(s, generator=g).to(DEV) for s in ((500, 32), (16, 32), (500, 16)))
    y, aux = hip.linear_fwd(x, W, None, hip.ACT_SIGMOID, 1.0, 1.0)
    ref = hip.linear_bwd(dy, y, aux, hip.ACT_SIGMOID, 1.0, 1.0, x, W, False, False)
    dW = torch.empty_like(W)
    hip.linear_bwd_noalloc(dy, y, aux, hip.ACT_SIGMOID, 1.0, 1.0, x, W, None, dW, None)
    assert torch.equal(dW, ref[1])


def test_noalloc_rejects_sizes_beyond_the_scratch():
    from nerf import _hip as hip
    M, K, N = 70000, 256, 256     # M*N alone is 17.9 M floats > 8 Mi floats
    assert hip.lib.nerf_linear_bwd_workspace_floats(M, K, N) > hip.NOALLOC_SCRATCH_FLOATS
    x = torch.zeros(M, K, device=DEV)
    W = torch.zeros(N, K, device=DEV)
    dy = torch.zeros(M, N, device=DEV)
    dW = torch.empty_like(W)
    with pytest.raises(RuntimeError, match="scratch"):
        hip.linear_bwd_noalloc(dy, None, None, hip.ACT_NONE, 1.0, 1.0, x, W, None, dW, None)
