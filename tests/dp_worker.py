"""One rank of the 2-rank data-parallel check (started by tests/test_a_dp_gpu.py through torch.distributed.run).

Both ranks drive the product path -- fused HIP kernels, FlatParameters bucket, FusedAdam -- on cuda:0 and
exchange through gloo (a one-GPU box has no second device for RCCL; the collective calls in nerf/parallel.py
are the same for both backends).  Each rank writes its findings to <out>/rank<r>.pt; the parent asserts.
"""
import datetime
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    from nerf import parallel
    from nerf.models import NeRF
    from nerf.rendering import render_nerf
    from oracle import nerf_oracle as O

    def quiesce():
        # rehearsal only: gloo stages device tensors through the host; with both ranks on ONE device the
        # collective is started on an idle device (an earlier 2-rank rehearsal stalled without this)
        torch.cuda.synchronize()

    params = O.closed_form_params(O.nerf_param_shapes())
    model = NeRF()
    model.load_state_dict(params)
    if rank != 0:                       # deliberately different weights: the broadcast must repair them
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.0 + 0.01 * rank)
    model = model.to(dev)
    flat = parallel.FlatParameters(model)
    quiesce()
    flat.broadcast_parameters(0)
    start = flat.flat.clone()

    R, S, lr = 64, 32, 5e-4
    g = torch.Generator().manual_seed(5)
    o = torch.randn(R, 3, generator=g); o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.1 * torch.randn(R, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    tgt = torch.rand(R, 3, generator=g)
    o, d, tgt = o.to(dev), d.to(dev), tgt.to(dev)
    torch.manual_seed(1000 + rank)      # global RNG differs per rank; the shared draw must not depend on it
    ugen = torch.Generator(device=dev).manual_seed(77)
    quiesce()
    parallel.check_shared_draws(dev, ugen)
    sl = parallel.shard_slice(R, rank, world)

    def loss_of(oo, dd, tt, u):
        rgb = render_nerf(model, oo, dd, 2.0, 6.0, num_samples=S, device=dev, chunk_size=8192, u=[u])
        return torch.nn.functional.mse_loss(rgb, tt)

    # (1) bucket after all_reduce_mean == single-process gradient of the concatenated batch
    u = parallel.shared_uniform(S, dev, ugen)
    flat.zero_grad()
    loss_of(o[sl], d[sl], tgt[sl], u).backward()
    quiesce()
    flat.all_reduce_mean()
    dp_grad = flat.grad.clone()
    flat.zero_grad()
    loss_of(o, d, tgt, u).backward()
    full_grad = flat.grad.clone()
    grad_err = float((dp_grad - full_grad).double().norm() / full_grad.double().norm())

    # (2) three data-parallel steps (1/world folded into the Adam kernel) against three single-process steps
    opt = parallel.FusedAdam(flat, lr=lr)
    draws = []
    for step in range(3):
        u = parallel.shared_uniform(S, dev, ugen)
        draws.append(u.clone())
        opt.zero_grad()
        loss_of(o[sl], d[sl], tgt[sl], u).backward()
        quiesce()
        flat.all_reduce_mean(scale_in_optimizer=True)
        opt.step()
    dp_params = flat.flat.clone()

    single = NeRF().to(dev)
    flat1 = parallel.FlatParameters(single)
    flat1.flat.copy_(start)
    flat1._touch()
    opt1 = parallel.FusedAdam(flat1, lr=lr)
    model_backup, model = model, single
    for step in range(3):
        opt1.zero_grad()
        loss_of(o, d, tgt, draws[step]).backward()
        opt1.step()
    model = model_backup
    single_params = flat1.flat.clone()
    torch.cuda.synchronize()
    torch.save({"grad_err": grad_err, "dp_params": dp_params.cpu(), "single_params": single_params.cpu(),
                "start": start.cpu(), "draw": draws[0].cpu(), "grad_norm": float(full_grad.norm())},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
