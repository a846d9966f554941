"""Ray-sharded data parallelism: one process per GPU, weights replicated, one flat fp32
gradient bucket all-reduced per step (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" on CPU for tests).

The reference has no distributed code.  Equivalence to its single-process step
(train.py:166-183): MSELoss is a mean over B*3, so equal contiguous ray slices + gradient
averaging reproduce the single-process gradient up to summation order -- provided every
rank uses the same uniform draw u[S], because the reference shares one t-vector across the
whole batch (rendering.py:26,59-62).  `shared_uniform` provides that draw.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_slice(n: int, rank: int, world_size: int) -> slice:
    """Contiguous equal slice of n rays for `rank`; n must divide evenly so that the mean of
    per-rank means equals the global mean."""
    if n % world_size != 0:
        raise ValueError(f"batch of {n} rays does not divide over {world_size} ranks")
    per = n // world_size
    return slice(rank * per, (rank + 1) * per)


def shared_uniform(num_samples: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The per-chunk uniform draw, identical on every rank: rank 0 draws, everyone receives."""
    u = torch.rand(num_samples, device=device, generator=generator)
    if world()[1] > 1:
        dist.broadcast(u, src=0)
    return u


class FlatParameters:
    """Re-homes a module's parameters (and their .grad) as views of two flat fp32 buffers, in
    state_dict order.  One all-reduce and one optimizer launch then cover the whole model (or several
    networks at once: pass a ModuleDict / ModuleList of them), while state_dict()/load_state_dict()/
    torch.optim see the usual 24 tensors per network.
    For this package's fused modules the backward kernels add their result straight into the gradient
    buffer (no per-tensor accumulation by autograd): drive training with loss.backward(), and zero
    the gradients with zero_grad() of this class or of FusedAdam (never set_to_none)."""

    def __init__(self, module: torch.nn.Module):
        params = list(module.parameters())
        if not params:
            raise ValueError("module has no parameters")
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.module = module
        self.params = params
        self.flat = torch.empty(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        off, start = 0, {}
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            start[id(p)] = off
            off += n
        # every fused network inside `module` (the module itself, or e.g. the coarse and fine members of a
        # ModuleDict) gets the slice of the bucket that holds its parameters as its gradient sink
        self.fused = [m for m in module.modules() if hasattr(m, "model_id")]
        for m in self.fused:
            own = list(m.parameters())
            first, count = start[id(own[0])], sum(p.numel() for p in own)
            at = first
            for p in own:
                if start[id(p)] != at:
                    raise ValueError("parameters of a fused network are not contiguous in the bucket")
                at += p.numel()
            m._grad_sink = self.grad[first:first + count]

    def zero_grad(self) -> None:
        """One memset; keeps .grad views attached (do not use set_to_none)."""
        self.grad.zero_()
        off = 0
        for p in self.params:  # re-attach in case something replaced .grad
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * self.grad.element_size():
                p.grad = self.grad[off:off + n].view(p.shape)
            off += n

    def all_reduce_mean(self) -> None:
        """Average the gradient bucket over ranks (no-op for a single process)."""
        _, ws = world()
        if ws > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
            self.grad.mul_(1.0 / ws)

    def broadcast_parameters(self, src: int = 0) -> None:
        if world()[1] > 1:
            dist.broadcast(self.flat, src=src)
            self._touch()

    def _touch(self) -> None:
        for m in self.fused:
            m.invalidate_packed()


class FusedAdam:
    """torch.optim.Adam semantics (train.py:123: betas (0.9, 0.999), eps 1e-8, no weight decay)
    as ONE HIP launch over FlatParameters; `lr` may be changed between steps (LambdaLR-style
    schedules multiply base_lr by lr_lambda(step))."""

    def __init__(self, flat: FlatParameters, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        from . import _hip
        self._hip = _hip
        self.flat = flat
        self.base_lr = float(lr)
        self.lr = float(lr)
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    def step(self) -> None:
        self.step_count += 1
        self._hip.adam_step(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                            self.betas[0], self.betas[1], self.eps)
        self.flat._touch()

    def state_dict(self) -> dict:
        return {"step": self.step_count, "lr": self.lr, "base_lr": self.base_lr, "exp_avg": self.exp_avg,
                "exp_avg_sq": self.exp_avg_sq}

    def load_state_dict(self, sd: dict) -> None:
        self.step_count, self.lr, self.base_lr = int(sd["step"]), float(sd["lr"]), float(sd["base_lr"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def lr_lambda(step: int, lr: float, lr_decay: float, lr_decay_factor: float, lr_min: float) -> float:
    """LambdaLR factor of the reference schedule (train.py:127-131)."""
    gamma = lr_decay_factor ** (1 / (lr_decay * 1000))
    return max(gamma ** step, lr_min / lr)
