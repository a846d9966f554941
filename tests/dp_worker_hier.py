"""One rank of the 2-rank data-parallel check of the HEADLINE shape (started by tests/test_a_dp_gpu.py through
torch.distributed.run): a coarse + fine NeRF pair in one FlatParameters bucket over a ModuleDict,
render_nerf_hierarchical with the two-stream backward overlap on, one all-reduce and one fused Adam launch per step
-- the step bench.py times (counterpart of train.py:179-183 of the reference for one process).

Both ranks run on cuda:0 and exchange through gloo (a one-GPU box has no second device for RCCL).  Each rank writes
its findings to <out>/rank<r>.pt; the parent asserts."""
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
    from nerf import ops, parallel
    from nerf.models import NeRF
    from nerf.rendering import render_nerf_hierarchical
    from oracle import nerf_oracle as O

    assert ops.OVERLAP_BACKWARD, "the two-stream backward is part of what is under test"

    def quiesce():
        # rehearsal only: gloo stages device tensors through the host; with both ranks on ONE device the collective
        # is started on an idle device
        torch.cuda.synchronize()

    def make_pair(perturb):
        pc = O.closed_form_params(O.nerf_param_shapes())
        coarse, fine = NeRF(), NeRF()
        coarse.load_state_dict(pc)
        fine.load_state_dict({k: v * 0.97 for k, v in pc.items()})
        pair = torch.nn.ModuleDict({"coarse": coarse, "fine": fine})
        if perturb:
            with torch.no_grad():
                for p in pair.parameters():
                    p.mul_(1.0 + perturb)
        return pair.to(dev)

    pair = make_pair(0.01 * rank)       # rank 1 starts from different weights: the broadcast must repair them
    flat = parallel.FlatParameters(pair)
    quiesce()
    flat.broadcast_parameters(0)
    start = flat.flat.clone()

    R, Nc, Nf, lr = 128, 16, 24, 5e-4
    g = torch.Generator().manual_seed(5)
    o = torch.randn(R, 3, generator=g); o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.1 * torch.randn(R, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    tgt = torch.rand(R, 3, generator=g)
    ufs = [torch.rand(R, Nf, generator=g) for _ in range(4)]          # per-ray fine draws: a rank uses its rows
    o, d, tgt = o.to(dev), d.to(dev), tgt.to(dev)
    ufs = [u.to(dev) for u in ufs]
    ugen = torch.Generator(device=dev).manual_seed(77)
    quiesce()
    parallel.check_shared_draws(dev, ugen)
    sl = parallel.shard_slice(R, rank, world)

    def loss_of(nets, rows, u, uf):
        rc, rf = render_nerf_hierarchical(nets["coarse"], nets["fine"], o[rows], d[rows], 2.0, 6.0, Nc, Nf, device=dev,
                                          white_background=True, chunk_size=8192, u=[u], u_fine=[uf[rows]])
        return torch.nn.functional.mse_loss(rc, tgt[rows]) + torch.nn.functional.mse_loss(rf, tgt[rows])

    everything = slice(0, R)
    # (1) the shared bucket after all_reduce_mean == single-process gradient of the concatenated batch
    u = parallel.shared_uniform(Nc, dev, ugen)
    flat.zero_grad()
    loss_of(pair, sl, u, ufs[0]).backward()
    quiesce()
    flat.all_reduce_mean()
    dp_grad = flat.grad.clone()
    flat.zero_grad()
    loss_of(pair, everything, u, ufs[0]).backward()
    torch.cuda.synchronize()
    full_grad = flat.grad.clone()
    n = sum(p.numel() for p in pair["coarse"].parameters())
    err = lambda a, b: float((a - b).double().norm() / b.double().norm())
    grad_err = (err(dp_grad[:n], full_grad[:n]), err(dp_grad[n:], full_grad[n:]))

    # (2) three data-parallel steps with the 1/world factor folded into the Adam kernel
    opt = parallel.FusedAdam(flat, lr=lr)
    draws, exp_avg_err = [], None
    for step in range(3):
        u = parallel.shared_uniform(Nc, dev, ugen)
        draws.append(u.clone())
        opt.zero_grad()
        loss_of(pair, sl, u, ufs[1 + step]).backward()
        quiesce()
        flat.all_reduce_mean(scale_in_optimizer=True)
        summed = flat.grad.clone()                      # still the SUM over ranks: the scale is pending
        opt.step()
        if step == 0:   # after the first step exp_avg = (1 - beta1) * mean gradient: a dropped or wrong 1/world shows here
            want = 0.1 * summed / world
            exp_avg_err = err(opt.exp_avg, want)
    dp_params = flat.flat.clone()

    # the same three steps in one process on the whole batch
    single = make_pair(0.0)
    flat1 = parallel.FlatParameters(single)
    flat1.flat.copy_(start)
    flat1._touch()
    opt1 = parallel.FusedAdam(flat1, lr=lr)
    for step in range(3):
        opt1.zero_grad()
        loss_of(single, everything, draws[step], ufs[1 + step]).backward()
        opt1.step()
    single_params = flat1.flat.clone()
    torch.cuda.synchronize()
    torch.save({"grad_err": grad_err, "exp_avg_err": exp_avg_err, "dp_params": dp_params.cpu(),
                "single_params": single_params.cpu(), "start": start.cpu(), "draw": draws[0].cpu(),
                "grad_norm": (float(full_grad[:n].norm()), float(full_grad[n:].norm()))},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
