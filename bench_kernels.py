#!/usr/bin/env python3
"""Stand-alone timings of the HBM-bound kernels (sampling, encoding, compositing) at sizes where
they fill the device, as achieved GB/s of ALGORITHMIC bytes against the 8 TB/s HBM3E peak.
Not the headline benchmark (that is bench.py); one JSON line per kernel.

    python bench_kernels.py [--rays 262144] [--samples 256] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
HBM_PEAK_GBS = 8000.0


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=262144)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from nerf import _hip
    dev = torch.device("cuda", 0)
    R, S = args.rays, args.samples
    g = torch.Generator(device=dev).manual_seed(0)
    colors = torch.rand(R, S, 3, device=dev, generator=g)
    dens = torch.relu(torch.randn(R, S, device=dev, generator=g)) * 10
    u = torch.rand(S, device=dev, generator=g)
    t, deltas = _hip.stratified_t(2.0, 6.0, S, u)
    grad = torch.rand(R, 3, device=dev, generator=g)
    o = torch.randn(R, 3, device=dev, generator=g)
    d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=1)
    x = torch.rand(R * 16, 3, device=dev, generator=g)
    w = torch.rand(R, 64, device=dev, generator=g)
    uu = torch.rand(R, 128, device=dev, generator=g)
    bins = torch.linspace(2, 6, 65, device=dev).expand(R, 65).contiguous()
    cdf = _hip.weights_to_cdf(w)
    tf = torch.sort(torch.rand(R, 128, device=dev, generator=g) * 4 + 2, dim=1).values.contiguous()
    tc = t[:64].contiguous()
    cases = [
        ("composite_fwd", lambda: _hip.composite_fwd(colors, dens, deltas, True), R * (16 * S + 12)),
        ("composite_fwd + weights/depth/acc", lambda: _hip.composite_fwd(colors, dens, deltas, True, t=t, want_weights=True,
                                                                         want_depth=True, want_acc=True),
         R * (16 * S + 12 + 4 * S + 8)),
        ("composite_bwd", lambda: _hip.composite_bwd(colors, dens, deltas, True, grad), R * (16 * S + 12 + 16 * S)),
        ("sample_positions", lambda: _hip.sample_positions(o, d, t, True, 2.0, 6.0), R * (24 + 12 * S)),
        ("posenc L=10", lambda: _hip.posenc(x, 10), x.shape[0] * (12 + 63 * 4)),
        ("weights_to_cdf B=64", lambda: _hip.weights_to_cdf(w), R * (64 * 4 + 65 * 4)),
        ("sample_pdf B=64 N=128", lambda: _hip.sample_pdf(bins, cdf, uu), R * (2 * 65 * 4 + 128 * 4 + 128 * 8)),
        ("merge_samples 64+128", lambda: _hip.merge_samples(tc, tf), R * (128 * 4 + 2 * 192 * 4)),
        # the three steps above in one launch, as the hierarchical renderer chains them: reads the coarse weights and the
        # draws, writes the merged depths and their differences
        ("resample 64+128 (cdf + draws + merge, fused)", lambda: _hip.resample(w, tc, uu), R * (64 * 4 + 128 * 4 + 2 * 192 * 4)),
    ]
    for name, fn, nbytes in cases:
        sec = timeit(fn, args.iters)
        gbs = nbytes / sec / 1e9
        print(json.dumps({"kernel": name, "rays": R, "samples": S, "algorithmic_bytes": nbytes, "us": sec * 1e6,
                          "achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}))


if __name__ == "__main__":
    main()
