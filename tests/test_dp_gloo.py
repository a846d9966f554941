"""world_size-2 data-parallel path on CPU (gloo): flat gradient bucket, shared uniform draw,
ray sharding.  The per-rank gradient producer is the oracle model (the HIP kernels need a GPU;
tests/test_a_dp_gpu.py runs the same check with the kernels); what is under test is the product's host
logic in nerf/parallel.py: after all_reduce_mean the bucket equals the single-process gradient of
the full batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf import parallel
    from oracle import nerf_oracle as O

    class Tiny(torch.nn.Module):  # same parameter names/shapes as the product's NeRF
        def __init__(self):
            super().__init__()
            for i, (n, shp) in enumerate(O.nerf_param_shapes()):
                self.register_parameter(n.replace(".", "_"), torch.nn.Parameter(torch.zeros(shp)))

    torch.manual_seed(100 + rank)   # deliberately different initial weights per rank
    net = Tiny()
    for p in net.parameters():
        p.data.normal_(0, 0.05)
    flat = parallel.FlatParameters(net)
    flat.broadcast_parameters(0)
    names = [n for n, _ in O.nerf_param_shapes()]
    params = {n: p for n, p in zip(names, net.parameters())}

    R, S = 16, 8
    g = torch.Generator().manual_seed(5)
    o = torch.randn(R, 3, generator=g); o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.1 * torch.randn(R, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    tgt = torch.rand(R, 3, generator=g)
    torch.manual_seed(1000 + rank)  # different global RNG state per rank: the draw must still agree
    parallel.check_shared_draws("cpu")
    u = parallel.shared_uniform(S, "cpu")
    bad = torch.Generator().manual_seed(7 + rank)   # a generator that is NOT seeded alike must be caught
    try:
        parallel.check_shared_draws("cpu", bad)
        caught = rank == 0                           # rank 0 compares with itself
    except RuntimeError:
        caught = True
    sl = parallel.shard_slice(R, rank, world)

    def loss_of(oo, dd, tt):
        rgb = O.render(lambda x, v: O.nerf_forward(params, x, v), oo, dd, 2.0, 6.0, S, [u])
        return torch.nn.functional.mse_loss(rgb, tt)

    flat.zero_grad()
    loss_of(o[sl], d[sl], tgt[sl]).backward()
    assert all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in net.parameters())
    local_grad = flat.grad.clone()
    flat.all_reduce_mean()
    dp_grad = flat.grad.clone()
    # the 1/world factor left to the optimizer: the bucket holds the sum, pending_scale the factor
    flat.zero_grad()
    flat.grad.copy_(local_grad)
    flat.all_reduce_mean(scale_in_optimizer=True)
    deferred_ok = flat.pending_scale == 0.5 and torch.allclose(flat.grad * flat.pending_scale, dp_grad, rtol=1e-6, atol=0)
    flat.zero_grad()
    deferred_ok = deferred_ok and flat.pending_scale == 1.0
    # the torch.optim path: per-tensor .grad detached from the bucket, one flattened all-reduce
    for p in net.parameters():
        p.grad = None
    loss_of(o[sl], d[sl], tgt[sl]).backward()
    parallel.all_reduce_module_grads(net)
    per_tensor = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    flat.zero_grad()
    loss_of(o, d, tgt).backward()
    full_grad = flat.grad.clone()
    ret[rank] = (float((dp_grad - full_grad).norm() / full_grad.norm()), u.numpy().copy(), flat.flat.clone().numpy(),
                 caught, deferred_ok, float((per_tensor - full_grad).norm() / full_grad.norm()))
    dist.destroy_process_group()


def test_dp2_gradient_equals_single_process():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2
    for r in range(world):
        assert ret[r][0] < 1e-5, ret[r][0]
    assert np.array_equal(ret[0][1], ret[1][1])          # shared uniform draw
    assert np.array_equal(ret[0][2], ret[1][2])          # broadcast parameters
    for r in range(world):
        assert ret[r][3], "a differently seeded generator was not detected"
        assert ret[r][4], "deferred 1/world scale"
        assert ret[r][5] < 1e-5, ret[r][5]


def test_batch_cursor_and_shard_slice_non_divisible_pool():
    """A pool that the batch does not divide: with drop_last every batch is full and shards evenly (what the
    data-parallel driver uses); without it the epoch ends on a short batch as DataLoader(shuffle=True) does."""
    sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
    from nerf import parallel
    from nerf.data import BatchCursor
    n, batch, world = 103, 16, 4
    cur = BatchCursor(n, "cpu", torch.Generator().manual_seed(1), drop_last=True)
    seen = []
    for _ in range(20):
        idx = cur.next(batch)
        assert idx.numel() == batch and idx.unique().numel() == batch
        parts = [idx[parallel.shard_slice(batch, r, world)] for r in range(world)]
        assert all(p.numel() == batch // world for p in parts) and torch.equal(torch.cat(parts), idx)
        seen.append(idx)
    first_epoch = torch.cat(seen[:n // batch])
    assert first_epoch.unique().numel() == (n // batch) * batch       # no repeats inside an epoch
    cur = BatchCursor(n, "cpu", torch.Generator().manual_seed(1), drop_last=False)
    sizes = [cur.next(batch).numel() for _ in range(8)]
    assert sizes == [16, 16, 16, 16, 16, 16, 7, 16]
    with pytest.raises(ValueError):
        parallel.shard_slice(7, 0, 2)
    with pytest.raises(ValueError):
        BatchCursor(5, "cpu", drop_last=True).next(6)
