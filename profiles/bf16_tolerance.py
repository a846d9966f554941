"""Tolerance-versus-fp32 report of the opt-in inference modes (bf16: one term per operand; bf16x3: three terms,
fp32 accuracy) at the large-batch stress shape: 262,144 rays x 256 MLP evaluations per ray, same rays and same
stratified draws in every precision.
Run on the GPU box:  python profiles/bf16_tolerance.py > profiles/r2_bf16_tolerance.json"""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
from nerf.models import NeRF
from nerf.rendering import render_nerf, render_nerf_hierarchical

DEV = torch.device("cuda", 0)
R, S, CHUNK = 262144, 256, 32768
HIER_GAIN, DENSITY_BIAS = 3.0, 0.5     # weight scale of the hierarchical case; the single-pass cases sweep it


def rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g)
    o = 4.0 * o / o.norm(dim=1, keepdim=True)                     # camera centres on the radius-4 sphere
    d = -o / 4.0 + 0.15 * torch.randn(n, 3, generator=g)          # towards the origin, jittered
    return o.to(DEV), (d / d.norm(dim=1, keepdim=True)).to(DEV)


def lively(model, gain, density_bias):
    """Default-initialised networks render almost empty space (density is a ReLU of a near-zero value);
    scale the weights and lift the density bias so that the renders have structure to disagree on."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("weight"):
                p.mul_(gain)
        model.block2[-1].bias[-1] += density_bias      # the density column is the last output of block2
    return model


def stats(out, ref):
    err = (out - ref).abs().float()
    mse = float(((out - ref) ** 2).mean())
    flat = err.reshape(-1)
    k = lambda q: float(flat.kthvalue(max(1, int(q * flat.numel())))[0])
    return {"fp32_image_mean": float(ref.mean()), "fp32_image_std": float(ref.std()), "psnr_db": -10.0 * math.log10(max(mse, 1e-20)), "mean_abs": float(err.mean()), "q99": k(0.99), "q99.9": k(0.999),
            "q99.99": k(0.9999), "max_abs": float(err.max()), "rays_with_any_channel_above_3e-2": int((err.max(dim=1)[0] > 3e-2).sum())}


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    return out, time.perf_counter() - t0


report = {"shape": {"rays": R, "mlp_evaluations_per_ray": S, "chunk_size": CHUNK}, "cases": []}
o, d = rays(R, 7)
nchunks = (R + CHUNK - 1) // CHUNK
with torch.no_grad():
    for seed, gain in ((0, 1.5), (1, 2.0), (2, 2.5), (3, 3.0)):
        torch.manual_seed(seed)
        model = lively(NeRF().to(DEV), gain, DENSITY_BIAS)
        u = [torch.rand(S, generator=torch.Generator().manual_seed(100 + seed)).to(DEV)] * nchunks
        single = lambda: render_nerf(model, o, d, 2.0, 6.0, num_samples=S, device=DEV, chunk_size=CHUNK, u=u)
        model.set_inference_precision("fp32"); ref, t32 = timed(single)
        model.set_inference_precision("bf16"); out, t16 = timed(single)
        model.set_inference_precision("bf16x3"); out3, t3 = timed(single)
        case = {"weights": f"default init x {gain}, density bias + {DENSITY_BIAS}, seed {seed}", "sampling": "single pass, 256 stratified samples",
                "fp32_rays_per_s": R / t32, "bf16_rays_per_s": R / t16, "bf16x3_rays_per_s": R / t3, **stats(out, ref),
                "bf16x3": stats(out3, ref)}
        report["cases"].append(case)
        print(json.dumps(case), file=sys.stderr, flush=True)
    torch.manual_seed(10)
    coarse, fine = lively(NeRF().to(DEV), HIER_GAIN, DENSITY_BIAS), lively(NeRF().to(DEV), HIER_GAIN, DENSITY_BIAS)
    uc = [torch.rand(64, generator=torch.Generator().manual_seed(200)).to(DEV)] * nchunks
    uf = [torch.rand(CHUNK, 192, generator=torch.Generator().manual_seed(201)).to(DEV)] * nchunks
    hier = lambda: render_nerf_hierarchical(coarse, fine, o, d, 2.0, 6.0, 64, 192, device=DEV, chunk_size=CHUNK, u=uc, u_fine=uf)[1]
    for m in (coarse, fine): m.set_inference_precision("fp32")
    ref, t32 = timed(hier)
    for m in (coarse, fine): m.set_inference_precision("bf16")
    out, t16 = timed(hier)
    for m in (coarse, fine): m.set_inference_precision("bf16x3")
    out3, t3 = timed(hier)
    case = {"weights": f"default init x {HIER_GAIN}, density bias + {DENSITY_BIAS}, seed 10 (coarse + fine)", "sampling": "64 coarse + 192 importance samples, fine pass over the merged 256",
            "fp32_rays_per_s": R / t32, "bf16_rays_per_s": R / t16, "bf16x3_rays_per_s": R / t3, **stats(out, ref), "bf16x3": stats(out3, ref),
            "note": "the fine sample positions follow the coarse weights, which differ between the precisions"}
    report["cases"].append(case)
print(json.dumps(report, indent=1))
