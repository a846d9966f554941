"""Single-rank RCCL check (started by tests/test_y_rccl_single_rank.py in a fresh process): the collective calls of
nerf/parallel.py on backend "nccl" (= RCCL on ROCm) with a world of one.  No second GPU is needed; what it shows
is that the library initialises on this box and that the flat bucket goes through an RCCL all-reduce / broadcast
and comes back unchanged."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=90))
    from nerf import parallel
    from nerf.models import NeRF
    model = NeRF().to(dev)
    flat = parallel.FlatParameters(model)
    flat.grad.copy_(torch.arange(flat.grad.numel(), device=dev, dtype=torch.float32) % 1000)
    before_g, before_p = flat.grad.clone(), flat.flat.clone()
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)          # what all_reduce_mean issues for world > 1
    dist.broadcast(flat.flat, src=0)                           # what broadcast_parameters issues
    parallel.check_shared_draws(dev, torch.Generator(device=dev).manual_seed(3))
    dist.barrier()
    torch.cuda.synchronize()
    assert torch.equal(flat.grad, before_g) and torch.equal(flat.flat, before_p)
    dist.destroy_process_group()
    print("rccl single rank ok")


if __name__ == "__main__":
    main()
