"""world_size-2 data-parallel path on CPU (gloo): flat gradient bucket, shared uniform draw,
ray sharding.  The per-rank gradient producer is the oracle model (the HIP kernels need a GPU);
what is under test is the product's host logic in nerf/parallel.py: after all_reduce_mean the
bucket equals the single-process gradient of the full batch."""
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
    torch.manual_seed(1000 + rank)  # different RNG state per rank: the draw must still agree
    u = parallel.shared_uniform(S, "cpu")
    sl = parallel.shard_slice(R, rank, world)

    def loss_of(oo, dd, tt):
        rgb = O.render(lambda x, v: O.nerf_forward(params, x, v), oo, dd, 2.0, 6.0, S, [u])
        return torch.nn.functional.mse_loss(rgb, tt)

    flat.zero_grad()
    loss_of(o[sl], d[sl], tgt[sl]).backward()
    assert all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in net.parameters())
    flat.all_reduce_mean()
    dp_grad = flat.grad.clone()
    flat.zero_grad()
    loss_of(o, d, tgt).backward()
    full_grad = flat.grad.clone()
    ret[rank] = (float((dp_grad - full_grad).norm() / full_grad.norm()), u.numpy().copy(), flat.flat.clone().numpy())
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
