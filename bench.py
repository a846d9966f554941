#!/usr/bin/env python3
"""Benchmark of the NeRF hot path on MI355X: train rays/s (fwd + bwd + optimizer).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic rays already resident in HBM.
Default workload (the single-GPU training configuration the headline rate is quoted on): lego-shaped
synthetic 800x800 scene, 4096 rays per GPU per step, 64 coarse + 128 fine samples per ray through a
coarse and a fine NeRF MLP, fp32 -- draw u, stratified t, coarse pass (fused sample+encode+MLP forward,
compositing with weights), inverse-CDF re-sampling, fine pass over the merged 192 depths, both MSE
losses, compositing backward, fused MLP data-gradient chains, per-layer weight gradients of both
networks, (N>1: one RCCL all-reduce of the shared flat gradient bucket), one fused Adam launch, weight
re-pack: 256 MLP evaluations per ray.  `--sampling single` is the reference's own single pass of
num_samples = 256 (config_lego.txt:15; same FLOPs), `--model siren` its other network, `--mode infer`
the gradient-free render (`--precision bf16`: the opt-in mode).  Weak scaling: every rank renders its
own 4096-ray slice of a 4096*N batch with the same stratified draw.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))

import numpy as np
import torch
import torch.distributed as dist

# algorithmic FLOPs per MLP evaluation (2 * multiply-accumulates)
FLOPS = {
    "nerf": {"fwd": 2 * 658944, "dgrad": 2 * 623232, "wgrad": 2 * 658944, "density": 2 * 556800},
    "siren": {"fwd": 2 * 561920, "dgrad": 2 * (561920 - 3 * 256 - 27 * 128), "wgrad": 2 * 561920, "density": 2 * 459776},
}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak (without sparsity)
PEAK_HBM_GBS = 8000.0
PROFILES = os.path.join(ROOT, "profiles")
# (function name, leading template arguments) of the timed launches as profiles/kernel_names.py keys them; trailing
# template parameters a kernel may gain later do not break the match
PMC_KERNEL = {("nerf", "mlp_fwd_train"): ("nerf_fwd_kernel", ("true",)), ("nerf", "mlp_bwd_data"): ("nerf_dgrad_kernel", ()),
              ("nerf", "mlp_fwd_infer"): ("nerf_fwd_kernel", ("false", "false")),
              ("siren", "mlp_fwd_train"): ("siren_fwd_kernel", ("true",)), ("siren", "mlp_bwd_data"): ("siren_dgrad_kernel", ()),
              ("siren", "mlp_fwd_infer"): ("siren_fwd_kernel", ("false", "false")),
              # (template arguments: model 0 = NeRF, all outputs, with the workspace)
              ("nerf", "mlp_fwd_train_x3"): ("fwd_x3i_kernel", ("0", "false", "true")), ("nerf", "mlp_bwd_data_x3"): ("dgrad_x3i_kernel", ("0",))}
# HBM bytes per MLP evaluation the kernels move BY DESIGN (csrc/mlp_layout.h): the forward writes its workspace
# (2857 / 4517 floats per sample) and reads 24 B of ray data per 256 samples; the data gradient writes every layer's
# dY (2693 / 2437 floats) and reads the head's inputs, the ReLU mask words or the stored cosines.  The packed
# weights (5 MB) stay in the caches.
DESIGN_BYTES = {("nerf", "mlp_fwd_train"): 4 * 2857, ("nerf", "mlp_bwd_data"): 4 * (2693 + 128 + 72 + 8),
                ("nerf", "mlp_fwd_train_x3"): 4 * 2857, ("nerf", "mlp_bwd_data_x3"): 4 * (2693 + 128 + 72 + 8),
                ("siren", "mlp_fwd_train"): 4 * 4517, ("siren", "mlp_bwd_data"): 4 * (2437 + 8 * 256 + 256 + 8)}


def pmc_traffic_per_sample(model, kernel, profiles=None):
    """HBM bytes per MLP evaluation of one kernel from the newest committed `rocprofv3 --pmc` passes
    (profiles/r<N>_pmc[_siren]_fetch.csv, ..._write.csv: profiles/summarize_pmc.py output, KiB per dispatch;
    r<N>_pmc_meta.json names the samples per dispatch of that profiled command).  FETCH_SIZE is doubled:
    on gfx950 it reports half the bytes of wide coalesced reads.  Counters cannot be read from inside
    this process; the figure is per sample of the profiled build.  (None, None) when no pass covers the kernel --
    with a line on stderr when counter files exist but hold no row for it, so that a renamed kernel cannot
    silently empty roofline.traffic."""
    import csv
    import glob
    import re
    profiles = profiles or PROFILES
    if profiles not in sys.path:
        sys.path.insert(0, profiles)
    from kernel_names import matches
    want = PMC_KERNEL.get((model, kernel))
    if want is None:
        return None, None
    # (the opt-in bf16x3 training kernels are profiled on the NeRF network only: files r<N>_pmc_x3_*.csv)
    infix = "_x3" if kernel.endswith("_x3") else ("" if model == "nerf" else f"_{model}")
    metas = sorted(glob.glob(os.path.join(profiles, "r*_pmc_meta.json")),
                   key=lambda f: int(re.search(r"r(\d+)_pmc_meta", f).group(1)), reverse=True)
    for meta_path in metas:       # newest round that profiled this model
        tag = os.path.basename(meta_path).split("_pmc_meta")[0]
        paths = [os.path.join(profiles, f"{tag}_pmc{infix}_{which}.csv") for which in ("fetch", "write")]
        if not all(os.path.exists(p) for p in paths):
            continue
        with open(meta_path) as f:
            meta = json.load(f)
        kib = {}
        for path, counter in zip(paths, ("FETCH_SIZE", "WRITE_SIZE")):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if row["counter"] == counter and matches(row.get("kernel_key") or row["kernel"], *want):
                        kib[counter] = float(row["mean_per_dispatch"])
        if len(kib) != 2:
            print(f"bench.py: {[os.path.basename(p) for p in paths]} hold no FETCH_SIZE/WRITE_SIZE row for "
                  f"{want[0]}<{','.join(want[1])}...>: roofline.traffic left empty", file=sys.stderr)
            return None, None
        per_sample = (2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024.0 / float(meta["samples_per_dispatch"])
        root = os.path.dirname(os.path.abspath(profiles))
        return per_sample, [os.path.relpath(p, root) for p in paths + [meta_path]]
    return None, None


def make_rays(n_views, H, W, device, seed=42):
    """Pinhole rays as nerf/data.py:87-112 would compute them for spherical poses
    (eval.py:36-41), generated on the device with torch ops; targets uniform in [0,1)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    fov = 0.6911112
    focal = 0.5 * W / np.tan(0.5 * fov)
    thetas = np.linspace(-180.0, 180.0, n_views + 1)[:-1]
    u = torch.arange(W, dtype=torch.float32, device=device)
    v = torch.arange(H, dtype=torch.float32, device=device)
    vg, ug = torch.meshgrid(v, u, indexing="ij")
    dirs_cam = torch.stack([ug - 0.5 * W, -(vg - 0.5 * H), -torch.full_like(ug, float(focal))], dim=-1).reshape(-1, 3)
    ro, rd = [], []
    for th in thetas:
        c2w = torch.from_numpy(_pose_spherical(float(th), -30.0, 4.0)).to(device)
        d = dirs_cam @ c2w[:3, :3].T
        d = d / d.norm(dim=-1, keepdim=True)
        rd.append(d)
        ro.append(c2w[:3, 3].expand_as(d))
    rays_o = torch.cat(ro).contiguous()
    rays_d = torch.cat(rd).contiguous()
    target = torch.rand(rays_o.shape[0], 3, generator=g).to(device)
    return rays_o, rays_d, target


def _pose_spherical(theta, phi, radius):
    t = np.eye(4, dtype=np.float32); t[2, 3] = radius
    ph = phi / 180.0 * np.pi
    rx = np.eye(4, dtype=np.float32); rx[1, 1] = rx[2, 2] = np.cos(ph); rx[1, 2] = -np.sin(ph); rx[2, 1] = np.sin(ph)
    th = theta / 180.0 * np.pi
    ry = np.eye(4, dtype=np.float32); ry[0, 0] = ry[2, 2] = np.cos(th); ry[0, 2] = -np.sin(th); ry[2, 0] = np.sin(th)
    swap = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return (swap @ (ry @ (rx @ t))).astype(np.float32)


def cpu_baseline(model_name, rays, samples, hier=False, n_coarse=64, n_fine=128, steps=5, warmup=2):
    """The CPU restatement (oracle) timed on the host cores: same step (render, MSE, backward,
    Adam) on a bounded sample of the same workload; median of `steps` steps after `warmup` untimed ones."""
    from oracle import nerf_oracle as O
    torch.manual_seed(0)
    is_nerf = model_name == "nerf"
    shapes = O.nerf_param_shapes() if is_nerf else O.siren_param_shapes()
    params = O.closed_form_params(shapes, gains=None if is_nerf else O.SIREN_GAINS)
    params2 = {k: v.clone() for k, v in params.items()}
    for p in list(params.values()) + list(params2.values()):
        p.requires_grad_(True)
    fwd = O.nerf_forward if is_nerf else O.siren_forward
    o = torch.randn(rays, 3); o = 4 * o / o.norm(dim=1, keepdim=True)
    d = -o / 4 + 0.1 * torch.randn(rays, 3); d = d / d.norm(dim=1, keepdim=True)
    target = torch.rand(rays, 3)
    allp = list(params.values()) + (list(params2.values()) if hier else [])
    m = [torch.zeros_like(p) for p in allp]
    v = [torch.zeros_like(p) for p in allp]
    times = []
    for it in range(steps + warmup):
        t0 = time.perf_counter()
        if hier:
            rc, rf, _, _ = O.render_hierarchical(lambda x, dd: fwd(params, x, dd), lambda x, dd: fwd(params2, x, dd), o, d,
                                                 2.0, 6.0, n_coarse, n_fine, torch.rand(n_coarse), torch.rand(rays, n_fine))
            loss = torch.nn.functional.mse_loss(rc, target) + torch.nn.functional.mse_loss(rf, target)
        else:
            rgb = O.render(lambda x, dd: fwd(params, x, dd), o, d, 2.0, 6.0, samples, [torch.rand(samples)])
            loss = torch.nn.functional.mse_loss(rgb, target)
        grads = torch.autograd.grad(loss, allp)
        with torch.no_grad():
            for i, (p, g) in enumerate(zip(allp, grads)):
                O.adam_step(p, g, m[i], v[i], it + 1, 5e-4)
        times.append(time.perf_counter() - t0)
    best = float(np.median(times[warmup:]))
    return {"value": rays / best, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{rays} rays x {samples} MLP evaluations/ray ({'coarse+fine' if hier else 'single pass'}), {model_name}, "
                      f"fwd+bwd+Adam, median of {steps} steps after {warmup} warm-ups (oracle/nerf_oracle.py, torch CPU fp32, "
                      f"{torch.get_num_threads()} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="nerf", choices=["nerf", "siren"])
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step")
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--sampling", default="hier", choices=["hier", "single"],
                    help="hier: 64 coarse + (64+128) fine evaluations per ray through two networks (the metric's "
                         "'64+128 samples/ray'); single: the reference's own single pass of --samples (config_lego.txt:15)")
    ap.add_argument("--coarse", type=int, default=64)
    ap.add_argument("--fine", type=int, default=128)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"],
                    help="bf16 / bf16x3: the opt-in inference modes on the bf16 matrix pipe (--mode infer, nerf only; bf16x3 = "
                         "three-term split at fp32 accuracy); never the default metric")
    ap.add_argument("--train-precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="opt-in (never the default line): a training step's forward, data gradient and wide weight-gradient "
                         "products on the bf16 matrix pipe at fp32 accuracy (three bf16 terms per operand); both networks")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device rehearses the multi-rank code path on a box with a single GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-steps", type=int, default=5)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from nerf import _hip, parallel
    from nerf.models import NeRF, Siren
    from nerf.rendering import render_nerf, render_nerf_hierarchical

    torch.manual_seed(42)
    np.random.seed(42)
    verbose = bool(os.environ.get("BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    note("process group up")
    hier = args.sampling == "hier"
    Model = NeRF if args.model == "nerf" else Siren
    models = [Model().to(dev) for _ in range(2 if hier else 1)]   # [coarse, fine] or [single]
    model = models[0]
    if args.precision != "fp32":
        if args.mode != "infer":
            raise SystemExit(f"--precision {args.precision} is an inference-only mode: add --mode infer")
        for m in models:
            m.set_inference_precision(args.precision)
    if args.train_precision != "fp32":
        if args.mode != "train":
            raise SystemExit("--train-precision applies to --mode train")
        for m in models:
            m.set_training_precision(args.train_precision)
    flat = parallel.FlatParameters(torch.nn.ModuleList(models))   # one bucket, one all-reduce, one Adam launch
    flat.broadcast_parameters(0)
    opt = parallel.FusedAdam(flat, lr=5e-4)
    near, far, B = 2.0, 6.0, args.rays
    S = args.coarse + (args.coarse + args.fine) if hier else args.samples     # MLP evaluations per ray

    # ray pool resident in HBM; ranks own disjoint view ranges of the same scene
    views_per_rank = max(1, min(args.views // max(world, 1), 12))
    rays_o, rays_d, target = make_rays(views_per_rank, args.res, args.res, dev, seed=42 + rank)
    n_pool = rays_o.shape[0]
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ugen = torch.Generator(device=dev).manual_seed(999)  # same seed on all ranks: identical draws, no collective
    parallel.check_shared_draws(dev, ugen)

    def host_sync_for_gloo():
        # rehearsal backend only: gloo stages device tensors through the host and, with several ranks
        # on ONE device, was seen to stall unless the device is idle when the collective starts
        if world > 1 and args.dist_backend == "gloo":
            torch.cuda.synchronize()

    def render(o, d):
        host_sync_for_gloo()
        if hier:
            u = parallel.shared_uniform(args.coarse, dev, ugen)
            # inference mode renders the fine image only (what eval.py does): the coarse pass stops at the densities
            rc, rf = render_nerf_hierarchical(models[0], models[1], o, d, near, far, args.coarse, args.fine, device=dev,
                                              white_background=True, chunk_size=max(B, 8192), u=[u],
                                              coarse_rgb=args.mode == "train")
            return rc, rf
        u = parallel.shared_uniform(S, dev, ugen)
        return (render_nerf(model, o, d, near, far, num_samples=S, device=dev, white_background=True,
                            chunk_size=max(B, 8192), u=[u]),)

    def train_step(step):
        idx = torch.randint(0, n_pool, (B,), device=dev, generator=gen)
        o, d, tgt = rays_o[idx], rays_d[idx], target[idx]
        preds = render(o, d)
        opt.zero_grad()
        loss = sum(torch.nn.functional.mse_loss(p, tgt) for p in preds)
        loss.backward()
        host_sync_for_gloo()
        flat.all_reduce_mean(scale_in_optimizer=True)   # 1/world rides in the Adam kernel's gradient scale
        opt.lr = opt.base_lr * parallel.lr_lambda(step, opt.base_lr, 300.0, 0.1, 1e-4)
        opt.step()
        return loss

    def infer_step(step):
        idx = torch.randint(0, n_pool, (B,), device=dev, generator=gen)
        with torch.no_grad():
            return render(rays_o[idx], rays_d[idx])[-1]

    step_fn = train_step if args.mode == "train" else infer_step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note("rays ready")
    for i in range(args.warmup):
        step_fn(i)
        note(f"warmup step {i} done")
    barrier()
    note("timed region starts")
    _hip.timers.enabled = True
    _hip.timers.reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step_fn(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    _hip.timers.enabled = False
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final = float(out.float().mean().item())

    # Per-kernel rates.  With a coarse + fine pair the two networks' backward chains run on two streams and
    # share the device, so an event bracket in the timed region also covers the other stream's kernels: those
    # entries are flagged, and the same launches are timed once more, serialised (overlap off), in two extra
    # steps after the timed region.
    def kernel_table(tm, steps):
        fl = FLOPS[args.model]
        if args.mode == "train":
            specs = [("mlp_fwd_train", fl["fwd"], "fused sample+encode+MLP forward (training, stores activations)"),
                     ("mlp_bwd_data", fl["dgrad"], "fused data-gradient chain"),
                     ("mlp_fwd_train_x3", 6 * fl["fwd"], "bf16x3 opt-in: the training forward with every fp32 operand as three bf16 "
                                                         "terms; matrix-pipe flops = 6 term products per product of the fp32 algorithm"),
                     ("mlp_bwd_data_x3", 6 * fl["dgrad"], "bf16x3 opt-in: the data-gradient chain, same arithmetic"),
                     ("mlp_bwd_weights_x3", 6 * fl["wgrad"], "bf16x3 opt-in: weight-gradient launch group, wide products with three-term "
                                                             "operands (flops counted as 6 term products throughout; the narrow products run on fp32 MFMA)"),
                     ("mlp_bwd_weights", fl["wgrad"], "per-layer weight-gradient GEMMs + one reduction (launch group)")]
        else:
            specs = [("mlp_fwd_infer", fl["fwd"], "fused sample+encode+MLP forward (inference)"),
                     ("mlp_fwd_infer_density", fl["density"], "the same pass stopped after the density row (coarse pass of a "
                                                           "hierarchical render; no feature layer, rgb layers or head)"),
                     ("mlp_fwd_infer_bf16", fl["fwd"], "bf16 opt-in: register-chain encode+MLP forward (inference)"),
                     ("mlp_fwd_infer_bf16_density", fl["density"], "bf16 opt-in, stopped after the density row"),
                     ("mlp_fwd_infer_bf16x3", 6 * fl["fwd"], "bf16x3 opt-in: the same chain with every fp32 operand as three bf16 terms; "
                                                             "matrix-pipe flops = 6 term products per product of the fp32 algorithm"),
                     ("mlp_fwd_infer_bf16x3_density", 6 * fl["density"], "bf16x3 opt-in, stopped after the density row")]
        table = {}
        for base, flop_per_sample, desc in specs:
            for name in (base, base + "@side"):
                if name in tm:
                    n, ms_total, samples = tm[name]
                    flop_total = flop_per_sample * samples
                    table[name] = {"launches_per_step": n / steps, "ms_per_step": ms_total / steps,
                                   "ms_per_launch": ms_total / n, "flop_per_launch": flop_total / n,
                                   "flop_per_sample": flop_per_sample, "tflops": flop_total / ms_total / 1e9, "what": desc}
        return table

    from nerf import ops as nerf_ops
    overlapped = hier and args.mode == "train" and nerf_ops.OVERLAP_BACKWARD
    tm_timed = _hip.timers.summary()
    tm_serial = None
    if overlapped:
        nerf_ops.OVERLAP_BACKWARD = False
        step_fn(args.warmup + args.steps)           # untimed, settles the allocator pools of the serial order
        torch.cuda.synchronize()
        _hip.timers.reset()
        _hip.timers.enabled = True
        for i in range(2):
            step_fn(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        _hip.timers.enabled = False
        tm_serial = _hip.timers.summary()
        nerf_ops.OVERLAP_BACKWARD = True

    if rank == 0:
        kernels = kernel_table(tm_timed, args.steps)
        if overlapped:
            for name, k in kernels.items():
                if name.split("@")[0] not in ("mlp_fwd_train", "mlp_fwd_train_x3"):
                    k["concurrent"] = True          # bracket includes the other stream's kernels
        # The roofline object describes ONE kernel over all its launches of the timed region (what a rocprofv3
        # --kernel-trace --stats average of the same command shows): the fused forward, which runs alone on the
        # device (its coarse launch on the side stream, its fine launch on the main one) and is the slowest
        # single kernel of the step; the weight-gradient entry is a launch group.
        alone = {}
        for name, k in kernels.items():
            if k.get("concurrent") or name.startswith("mlp_bwd_weights"):   # (also mlp_bwd_weights_x3)
                continue
            m = alone.setdefault(name.split("@")[0], {"n": 0.0, "ms": 0.0, "flop": 0.0, "per_sample": k["flop_per_sample"]})
            m["n"] += k["launches_per_step"]
            m["ms"] += k["ms_per_step"]
            m["flop"] += k["flop_per_launch"] * k["launches_per_step"]
        dom = max(alone, key=lambda b: alone[b]["ms"]) if alone else None
        roof = None
        if dom:
            m = alone[dom]
            ach = m["flop"] / m["ms"] / 1e9
            samples_per_launch = m["flop"] / m["n"] / m["per_sample"]
            tb, tb_files = pmc_traffic_per_sample(args.model, dom)
            peak = PEAK_BF16_MFMA_TFLOPS if ("bf16" in dom or dom.endswith("_x3")) else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                    "frac": ach / peak, "traffic": tb * samples_per_launch if tb else None,
                    "traffic_unit": "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) per sample of the rocprofv3 --pmc "
                                    "passes in traffic_source x mean samples per launch",
                    "traffic_source": tb_files, "launches_per_step": m["n"],
                    "ms_per_launch": m["ms"] / m["n"], "flop_per_launch": m["flop"] / m["n"],
                    "flop_per_sample": m["per_sample"], "samples_per_launch": samples_per_launch}
        total_flop = sum(k["flop_per_launch"] * k["launches_per_step"] for k in kernels.values())
        ms_step = elapsed / args.steps * 1e3
        res = {
            "metric": (f"train rays/sec (fwd+bwd+Adam, {args.coarse}+{args.fine} samples/ray)" if hier else
                       f"train rays/sec (fwd+bwd+Adam, {S} samples/ray single pass)") if args.mode == "train"
                      else (f"inference rays/sec ({S} MLP evaluations/ray"
                            + (f", the {args.coarse} coarse ones density only)" if hier else ")")),
            "value": world * B * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3" if args.train_precision == "bf16x3"
            else {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3"}[args.precision], "data": "synthetic",
            "config": {"workload": f"lego-shaped synthetic {args.res}x{args.res} scene, {B} rays/GPU/step, "
                                   + (f"{args.coarse} coarse + ({args.coarse}+{args.fine}) fine = {S} MLP evaluations/ray, "
                                      f"coarse+fine {args.model} MLPs" if hier else
                                      f"{S} samples/ray single pass, one {args.model} MLP")
                                   + {"fp32": " (fp32 MFMA)", "bf16": " (bf16 MFMA, fp32 accumulate)",
                                      "bf16x3": " (fp32 operands as three bf16 terms, six bf16 MFMAs per product, fp32 accumulate)"}[args.precision]
                                   + (" -- opt-in training mode: every fp32 operand as three bf16 terms on the bf16 matrix pipe (narrow weight-gradient products on fp32 MFMA)"
                                      if args.train_precision == "bf16x3" else "") + f", {args.mode}", "network": args.model, "sampling": args.sampling,
                       "rays_per_gpu": B, "rays_per_step": B * world, "mlp_evals_per_ray": S,
                       "parallelism": f"dp{world} ray-sharded, flat-bucket RCCL all-reduce" if world > 1 else "single GPU"},
            "roofline": roof,
            "kernels": kernels,
            "kernels_serialized": kernel_table(tm_serial, 2) if tm_serial else None,
            "step_mfma_frac": (total_flop / (ms_step * 1e-3) / 1e12)
                              / (PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" and args.train_precision == "fp32" else PEAK_BF16_MFMA_TFLOPS),
            "final_value": final,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "train":
            res["cpu_baseline"] = cpu_baseline(args.model, args.cpu_rays, S, hier, args.coarse, args.fine, steps=args.cpu_steps)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
