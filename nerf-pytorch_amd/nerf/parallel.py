"""Ray-sharded data parallelism: one process per GPU, weights replicated, one flat fp32
gradient bucket all-reduced per step (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" on CPU for tests).

The reference has no distributed code.  Equivalence to its single-process step
(train.py:166-183): MSELoss is a mean over B*3, so equal contiguous ray slices + gradient
averaging reproduce the single-process gradient up to summation order -- provided every
rank uses the same uniform draw u[S], because the reference shares one t-vector across the
whole batch (rendering.py:26,59-62).  `shared_uniform` provides that draw from identically seeded
per-rank generators: no collective per draw, one broadcast at start-up to verify the agreement.
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


_SHARED_SEED = 0x5EED0A11
_shared_generators = {}


def shared_generator(device, seed: int = _SHARED_SEED) -> torch.Generator:
    """The generator on `device` that shared_uniform draws from when none is passed: seeded with the same
    constant on every rank, so the ranks' draw sequences are identical without any communication."""
    key = (str(torch.device(device)), seed)
    if key not in _shared_generators:
        _shared_generators[key] = torch.Generator(device=device).manual_seed(seed)
    return _shared_generators[key]


def shared_uniform(num_samples: int, device, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The per-chunk uniform draw, identical on every rank.  Every rank draws from its own generator;
    the generators carry the same seed and are advanced by the same calls, so no collective is needed
    (a broadcast per chunk would be one latency-bound collective per chunk and step).  Pass the same
    seeded generator on all ranks, or none to use shared_generator(device); check_shared_draws()
    verifies the agreement once."""
    if generator is None:
        generator = shared_generator(device)
    return torch.rand(num_samples, device=device, generator=generator)


def check_shared_draws(device, generator: Optional[torch.Generator] = None, n: int = 64) -> None:
    """One broadcast at start-up: raises on every rank's own comparison if its generator is not in
    step with rank 0's.  Consumes n draws on every rank alike."""
    mine = shared_uniform(n, device, generator)
    if world()[1] > 1:
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(ref, mine):
            raise RuntimeError("data-parallel ranks do not share the stratified-sampling draw: seed the "
                               "generator passed to shared_uniform identically on every rank")


def broadcast_module_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Rank src's parameters to every rank (for modules that are not re-homed by FlatParameters)."""
    if world()[1] > 1:
        with torch.no_grad():
            for p in module.parameters():
                dist.broadcast(p.data, src=src)
        for m in module.modules():
            if hasattr(m, "invalidate_packed"):
                m.invalidate_packed()


def all_reduce_module_grads(module: torch.nn.Module) -> None:
    """Average the per-tensor .grad of a module over ranks through ONE flattened all-reduce (the path of
    torch.optim optimizers, whose zero_grad detaches .grad from any flat bucket)."""
    ws = world()[1]
    if ws <= 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if len(grads) != sum(1 for _ in module.parameters()):
        raise RuntimeError("all_reduce_module_grads: a parameter has no gradient on this rank")
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / ws)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


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
        self.pending_scale = 1.0   # factor the optimizer still has to apply to the bucket (see all_reduce_mean)
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
        self.pending_scale = 1.0
        off = 0
        for p in self.params:  # re-attach in case something replaced .grad
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * self.grad.element_size():
                p.grad = self.grad[off:off + n].view(p.shape)
            off += n

    def all_reduce_mean(self, scale_in_optimizer: bool = False) -> None:
        """Average the gradient bucket over ranks (no-op for a single process): one SUM all-reduce, then
        the 1/world factor -- applied here by one more elementwise launch, or, with scale_in_optimizer,
        left in `pending_scale` for FusedAdam.step(), which folds it into the Adam kernel's gradient
        scale (the bucket then holds the SUM until zero_grad)."""
        _, ws = world()
        if ws > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
            if scale_in_optimizer:
                self.pending_scale *= 1.0 / ws
            else:
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
                            self.betas[0], self.betas[1], self.eps, grad_scale=self.flat.pending_scale)
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
