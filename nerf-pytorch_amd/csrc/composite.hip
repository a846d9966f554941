// Alpha compositing with exclusive-cumprod transmittance, forward and backward.
// Replaces nerf/rendering.py:110-153 (compute_accumulated_transmittance, composite_volume)
// and what autograd derives from them (train.py:181).
//
// Layout: one ray per wavefront.  Lane l owns EPL consecutive samples
// [l*EPL, (l+1)*EPL); the transmittance is a wave-wide exclusive product scan of the
// per-lane products, the backward is a wave-wide suffix scan of affine maps
// (zero-safe: no division by beta or T).  HBM-bound: 16*S B/ray read forward.
#include "common.h"

namespace {

constexpr int kWavesPerBlock = 4;

// wave-exclusive product over lanes: returns prod of v over lanes < lane
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(inc, off);
        if (lane >= off) inc *= o;
    }
    const float ex = __shfl_up(inc, 1);
    return lane == 0 ? 1.0f : ex;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Loads the EPL samples this lane owns.  VEC: S == 64*EPL and EPL % 4 == 0 -> 16-byte loads.  COLORS = false (the
// weights-only call, colors == NULL) is a compile-time variant: a run-time test in front of every colour load keeps the
// loads from being issued back to back (forward 180 -> 215 us at 262,144 rays x 256).
template <int EPL, bool VEC, bool COLORS = true>
__device__ __forceinline__ void load_ray(const float* __restrict__ colors, const float* __restrict__ dens,
                                         const float* __restrict__ deltas, int64_t r, int S, int lane,
                                         float (&sig)[EPL], float (&dl)[EPL], float (&col)[EPL][3]) {
    // `deltas` already points at this ray's row when depths are per ray
    const int s0 = lane * EPL;
    if constexpr (VEC) {
        const f32x4* dp = reinterpret_cast<const f32x4*>(dens + r * S + s0);
        const f32x4* lp = reinterpret_cast<const f32x4*>(deltas + s0);
        const f32x4* cp = reinterpret_cast<const f32x4*>(colors + (r * S + s0) * 3);
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            const f32x4 dv = dp[q], lv = lp[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) { sig[4 * q + k] = dv[k]; dl[4 * q + k] = lv[k]; }
        }
        float flat[EPL * 3];
#pragma unroll
        for (int q = 0; q < EPL * 3 / 4; ++q) {
            f32x4 cv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (COLORS) cv = cp[q];
#pragma unroll
            for (int k = 0; k < 4; ++k) flat[4 * q + k] = cv[k];
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e)
#pragma unroll
            for (int c = 0; c < 3; ++c) col[e][c] = flat[3 * e + c];
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int s = s0 + e;
            const bool ok = s < S;
            sig[e] = ok ? dens[r * S + s] : 0.0f;   // sigma = 0 -> alpha = 0, beta = 1: identity
            dl[e] = ok ? deltas[s] : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) col[e][c] = (ok && COLORS) ? colors[(r * S + s) * 3 + c] : 0.0f;
        }
    }
}

template <int EPL, bool VEC, bool COLORS>
__global__ __launch_bounds__(64 * kWavesPerBlock) void composite_fwd_kernel(
    const float* __restrict__ colors, const float* __restrict__ dens, const float* __restrict__ deltas,
    const float* __restrict__ tvals, int64_t R, int S, int per_ray, int white, float* __restrict__ rgb,
    float* __restrict__ weights, float* __restrict__ depth, float* __restrict__ accout) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;  // wave-uniform; no block barrier in this kernel
    float sig[EPL], dl[EPL], col[EPL][3];
    if (per_ray) {
        deltas += r * S;
        if (tvals) tvals += r * S;
    }
    load_ray<EPL, VEC, COLORS>(colors, dens, deltas, r, S, lane, sig, dl, col);

    float alpha[EPL], beta[EPL];
    float prod = 1.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float ex = expf(mul_rn(-sig[e], dl[e]));
        alpha[e] = sub_rn(1.0f, ex);
        beta[e] = sub_rn(1.0f, alpha[e]);  // 1-(1-e), as the reference computes it
        prod *= beta[e];
    }
    float T = wave_excl_prod(prod, lane);
    float sr = 0.f, sg = 0.f, sb = 0.f, sw = 0.f, sd = 0.f;
    float w[EPL], tv[EPL];
    if (depth) {
        if constexpr (VEC) {
            const f32x4* tp = reinterpret_cast<const f32x4*>(tvals + lane * EPL);
#pragma unroll
            for (int q = 0; q < EPL / 4; ++q) {
                const f32x4 v = tp[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) tv[4 * q + k] = v[k];
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) tv[e] = (lane * EPL + e < S) ? tvals[lane * EPL + e] : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        w[e] = T * alpha[e];
        T *= beta[e];
        sr += w[e] * col[e][0];
        sg += w[e] * col[e][1];
        sb += w[e] * col[e][2];
        sw += w[e];
        if (depth) sd += w[e] * tv[e];
    }
    if (weights) {
        if constexpr (VEC) {   // S == 64*EPL, EPL % 4 == 0: the lane's weights as 16-byte stores
            f32x4* wp = reinterpret_cast<f32x4*>(weights + r * S + lane * EPL);
#pragma unroll
            for (int q = 0; q < EPL / 4; ++q) wp[q] = f32x4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int s = lane * EPL + e;
                if (s < S) weights[r * S + s] = w[e];
            }
        }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sw = wave_sum(sw);
    if (depth) sd = wave_sum(sd);
    if (lane == 0) {
        const float bg = white ? (1.0f - sw) : 0.0f;
        if constexpr (COLORS) {
            rgb[r * 3 + 0] = sr + bg;
            rgb[r * 3 + 1] = sg + bg;
            rgb[r * 3 + 2] = sb + bg;
        }
        if (depth) depth[r] = sd;
        if (accout) accout[r] = sw;
    }
}

template <int EPL, bool VEC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void composite_bwd_kernel(
    const float* __restrict__ colors, const float* __restrict__ dens, const float* __restrict__ deltas, int64_t R,
    int S, int per_ray, int white, const float* __restrict__ grad_rgb, float* __restrict__ dcolors,
    float* __restrict__ ddens) {
    __shared__ __attribute__((aligned(16))) float stage[VEC ? kWavesPerBlock * 64 * EPL * 3 : 4];   // VEC: one row of color gradients per wave
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    float sig[EPL], dl[EPL], col[EPL][3];
    if (per_ray) deltas += r * S;
    load_ray<EPL, VEC>(colors, dens, deltas, r, S, lane, sig, dl, col);
    const float g0 = grad_rgb[r * 3 + 0], g1 = grad_rgb[r * 3 + 1], g2 = grad_rgb[r * 3 + 2];
    const float gbg = white ? (g0 + g1 + g2) : 0.0f;

    float ex[EPL], alpha[EPL], beta[EPL], Tk[EPL];
    float prod = 1.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        ex[e] = expf(mul_rn(-sig[e], dl[e]));
        alpha[e] = sub_rn(1.0f, ex[e]);
        beta[e] = sub_rn(1.0f, alpha[e]);
        prod *= beta[e];
    }
    float T = wave_excl_prod(prod, lane);
    // G_k = d loss / d w_k ; dT_k = G_k * alpha_k ; F_k(x) = dT_k + beta_k x ; R_{k-1} = F_k(R_k)
    float G[EPL], dT[EPL];
    float ca = 1.0f, cb = 0.0f;  // this lane's composed map C = F_first o ... o F_last : x -> cb + ca x
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        Tk[e] = T;
        T *= beta[e];
        G[e] = g0 * col[e][0] + g1 * col[e][1] + g2 * col[e][2] - gbg;
        dT[e] = G[e] * alpha[e];
    }
#pragma unroll
    for (int e = EPL - 1; e >= 0; --e) {
        // prepend F_e: (F_e o C)(x) = dT_e + beta_e (cb + ca x)
        cb = dT[e] + beta[e] * cb;
        ca = beta[e] * ca;
    }
    // inclusive suffix composition over lanes: S_l = C_l o C_{l+1} o ... o C_63
    float sa = ca, sb = cb;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float oa = __shfl_down(sa, off), ob = __shfl_down(sb, off);
        if (lane + off < 64) {
            sb = sb + sa * ob;
            sa = sa * oa;
        }
    }
    // R at this lane's last sample = S_{l+1}(0)
    float x = __shfl_down(sb, 1);
    if (lane == 63) x = 0.0f;
    float dsig[EPL], dcol[EPL][3];
#pragma unroll
    for (int e = EPL - 1; e >= 0; --e) {
        const float dbeta = Tk[e] * x;
        const float dalpha = G[e] * Tk[e] - dbeta;
        dsig[e] = dalpha * dl[e] * ex[e];
        x = dT[e] + beta[e] * x;
        const float w = Tk[e] * alpha[e];
        dcol[e][0] = g0 * w;
        dcol[e][1] = g1 * w;
        dcol[e][2] = g2 * w;
    }
    const int s0 = lane * EPL;
    if constexpr (VEC) {
        f32x4* dp = reinterpret_cast<f32x4*>(ddens + r * S + s0);
#pragma unroll
        for (int q = 0; q < EPL / 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = dsig[4 * q + k];
            dp[q] = v;
        }
        // the lane's 3*EPL color gradients are contiguous in the ray's row but 12*EPL bytes from the next lane's:
        // they pass through the wave's LDS row so that every store instruction writes 1 KiB contiguous
        // (16-byte pieces 48+ bytes apart reach about half the store rate)
        float* row = stage + (threadIdx.x >> 6) * (64 * EPL * 3);
        f32x4* mine = reinterpret_cast<f32x4*>(row + lane * (EPL * 3));
        float flat[EPL * 3];
#pragma unroll
        for (int e = 0; e < EPL; ++e)
#pragma unroll
            for (int c = 0; c < 3; ++c) flat[3 * e + c] = dcol[e][c];
#pragma unroll
        for (int q = 0; q < EPL * 3 / 4; ++q) mine[q] = f32x4{flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]};
        __builtin_amdgcn_wave_barrier();          // a wave's LDS operations complete in program order
        float* crow = dcolors + r * S * 3;
#pragma unroll
        for (int q = 0; q < EPL * 3 / 4; ++q) {
            const int fl = (q * 64 + lane) * 4;
            *reinterpret_cast<f32x4*>(crow + fl) = *reinterpret_cast<const f32x4*>(row + fl);
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int s = s0 + e;
            if (s < S) {
                ddens[r * S + s] = dsig[e];
#pragma unroll
                for (int c = 0; c < 3; ++c) dcolors[(r * S + s) * 3 + c] = dcol[e][c];
            }
        }
    }
}

// exclusive cumulative product along a ray (compute_accumulated_transmittance on its own)
__global__ __launch_bounds__(64 * kWavesPerBlock) void excl_cumprod_kernel(const float* __restrict__ betas, int64_t R,
                                                                          int S, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const int per = (S + 63) / 64;
    const int s0 = lane * per;
    float prod = 1.0f;
    for (int e = 0; e < per; ++e)
        if (s0 + e < S) prod *= betas[r * S + s0 + e];
    float T = wave_excl_prod(prod, lane);
    for (int e = 0; e < per; ++e)
        if (s0 + e < S) {
            out[r * S + s0 + e] = T;
            T *= betas[r * S + s0 + e];
        }
}

// gradient of the exclusive cumulative product: T_s = prod_{k<s} beta_k, g = dL/dT
//   dL/dbeta_j = T_j * A_j,   A_j = sum_{s>j} g_s prod_{j<k<s} beta_k = g_{j+1} + beta_{j+1} A_{j+1},  A_{S-1} = 0
// the same zero-safe suffix scan of affine maps as the compositing backward (no division by beta or T),
// which is what ATen's cumprod_backward yields, shifted by the exclusive offset.
template <int EPL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void excl_cumprod_bwd_kernel(const float* __restrict__ betas,
                                                                              const float* __restrict__ grad_out,
                                                                              int64_t R, int S,
                                                                              float* __restrict__ dbetas) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= R) return;
    const int s0 = lane * EPL;
    float beta[EPL], g[EPL], Tk[EPL];
    float prod = 1.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const bool ok = s0 + e < S;
        beta[e] = ok ? betas[r * S + s0 + e] : 1.0f;     // beyond the ray: identity map
        g[e] = ok ? grad_out[r * S + s0 + e] : 0.0f;
        prod *= beta[e];
    }
    float T = wave_excl_prod(prod, lane);
    // F_s(x) = g_s + beta_s x maps A_s to A_{s-1}; this lane's composition C = F_first o ... o F_last
    float ca = 1.0f, cb = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        Tk[e] = T;
        T *= beta[e];
    }
#pragma unroll
    for (int e = EPL - 1; e >= 0; --e) {
        cb = g[e] + beta[e] * cb;
        ca = beta[e] * ca;
    }
    float sa = ca, sb = cb;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float oa = __shfl_down(sa, off), ob = __shfl_down(sb, off);
        if (lane + off < 64) {
            sb = sb + sa * ob;
            sa = sa * oa;
        }
    }
    float x = __shfl_down(sb, 1);   // A at this lane's last sample = S_{l+1}(0)
    if (lane == 63) x = 0.0f;
#pragma unroll
    for (int e = EPL - 1; e >= 0; --e) {
        if (s0 + e < S) dbetas[r * S + s0 + e] = Tk[e] * x;
        x = g[e] + beta[e] * x;
    }
}

template <int EPL>
int launch_fwd(const float* colors, const float* dens, const float* deltas, const float* t, int64_t R, int S, int per_ray,
               int white, float* rgb, float* weights, float* depth, float* acc, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div64(R, kWavesPerBlock);
    const bool vec = (EPL % 4 == 0) && (S == 64 * EPL) &&
                     ((((uintptr_t)colors | (uintptr_t)dens | (uintptr_t)deltas | (uintptr_t)weights | (uintptr_t)t) & 15) == 0);
    auto launch = [&](auto kern) { kern<<<grid, 64 * kWavesPerBlock, 0, st>>>(colors, dens, deltas, t, R, S, per_ray, white, rgb, weights, depth, acc); };
    if (vec) {
        if constexpr (EPL % 4 == 0) {
            if (colors) launch(composite_fwd_kernel<EPL, true, true>);
            else launch(composite_fwd_kernel<EPL, true, false>);
        }
    } else {
        if (colors) launch(composite_fwd_kernel<EPL, false, true>);
        else launch(composite_fwd_kernel<EPL, false, false>);
    }
    return nerf_check_launch("nerf_composite_fwd");
}

template <int EPL>
int launch_bwd(const float* colors, const float* dens, const float* deltas, int64_t R, int S, int per_ray, int white,
               const float* grad_rgb, float* dcolors, float* ddens, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div64(R, kWavesPerBlock);
    const bool vec = (EPL % 4 == 0) && (S == 64 * EPL) &&
                     ((((uintptr_t)colors | (uintptr_t)dens | (uintptr_t)deltas | (uintptr_t)dcolors |
                        (uintptr_t)ddens) & 15) == 0);
    if (vec) {
        if constexpr (EPL % 4 == 0)
            composite_bwd_kernel<EPL, true><<<grid, 64 * kWavesPerBlock, 0, st>>>(colors, dens, deltas, R, S, per_ray, white,
                                                                                 grad_rgb, dcolors, ddens);
    } else {
        composite_bwd_kernel<EPL, false><<<grid, 64 * kWavesPerBlock, 0, st>>>(colors, dens, deltas, R, S, per_ray, white,
                                                                              grad_rgb, dcolors, ddens);
    }
    return nerf_check_launch("nerf_composite_bwd");
}

}  // namespace

extern "C" int nerf_composite_fwd(const float* colors, const float* densities, const float* deltas, const float* t,
                                  int64_t R, int S, int per_ray, int white_background, float* rgb, float* weights,
                                  float* depth, float* acc, void* stream) {
    NERF_REQUIRE(R >= 0 && S >= 1 && S <= 1024, "nerf_composite_fwd: bad R=%lld S=%d (1..1024)", (long long)R, S);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(densities);
    NERF_REQUIRE_PTR(deltas);
    NERF_REQUIRE((colors != nullptr && rgb != nullptr) || (colors == nullptr && rgb == nullptr && weights != nullptr),
                 "nerf_composite_fwd: colors and rgb go together; without them (weights only) the weights output is required");
    NERF_REQUIRE(depth == nullptr || t != nullptr, "nerf_composite_fwd: depth output needs t");
    NERF_REQUIRE(ceil_div64(R, kWavesPerBlock) < (1ll << 31), "nerf_composite_fwd: too many rays");
    hipStream_t st = (hipStream_t)stream;
    const int epl = (S + 63) / 64;
    if (epl <= 1) return launch_fwd<1>(colors, densities, deltas, t, R, S, per_ray, white_background, rgb, weights, depth, acc, st);
    if (epl <= 2) return launch_fwd<2>(colors, densities, deltas, t, R, S, per_ray, white_background, rgb, weights, depth, acc, st);
    if (epl <= 4) return launch_fwd<4>(colors, densities, deltas, t, R, S, per_ray, white_background, rgb, weights, depth, acc, st);
    if (epl <= 8) return launch_fwd<8>(colors, densities, deltas, t, R, S, per_ray, white_background, rgb, weights, depth, acc, st);
    return launch_fwd<16>(colors, densities, deltas, t, R, S, per_ray, white_background, rgb, weights, depth, acc, st);
}

extern "C" int nerf_composite_bwd(const float* colors, const float* densities, const float* deltas, int64_t R, int S,
                                  int per_ray, int white_background, const float* grad_rgb, float* dcolors,
                                  float* ddensities, void* stream) {
    NERF_REQUIRE(R >= 0 && S >= 1 && S <= 1024, "nerf_composite_bwd: bad R=%lld S=%d (1..1024)", (long long)R, S);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(colors);
    NERF_REQUIRE_PTR(densities);
    NERF_REQUIRE_PTR(deltas);
    NERF_REQUIRE_PTR(grad_rgb);
    NERF_REQUIRE_PTR(dcolors);
    NERF_REQUIRE_PTR(ddensities);
    NERF_REQUIRE(ceil_div64(R, kWavesPerBlock) < (1ll << 31), "nerf_composite_bwd: too many rays");
    hipStream_t st = (hipStream_t)stream;
    const int epl = (S + 63) / 64;
    if (epl <= 1) return launch_bwd<1>(colors, densities, deltas, R, S, per_ray, white_background, grad_rgb, dcolors, ddensities, st);
    if (epl <= 2) return launch_bwd<2>(colors, densities, deltas, R, S, per_ray, white_background, grad_rgb, dcolors, ddensities, st);
    if (epl <= 4) return launch_bwd<4>(colors, densities, deltas, R, S, per_ray, white_background, grad_rgb, dcolors, ddensities, st);
    if (epl <= 8) return launch_bwd<8>(colors, densities, deltas, R, S, per_ray, white_background, grad_rgb, dcolors, ddensities, st);
    return launch_bwd<16>(colors, densities, deltas, R, S, per_ray, white_background, grad_rgb, dcolors, ddensities, st);
}

extern "C" int nerf_exclusive_cumprod(const float* betas, int64_t R, int S, float* out, void* stream) {
    NERF_REQUIRE(R >= 0 && S >= 1, "nerf_exclusive_cumprod: bad R=%lld S=%d", (long long)R, S);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(betas);
    NERF_REQUIRE_PTR(out);
    NERF_REQUIRE(ceil_div64(R, kWavesPerBlock) < (1ll << 31), "nerf_exclusive_cumprod: too many rays");
    excl_cumprod_kernel<<<(unsigned)ceil_div64(R, kWavesPerBlock), 64 * kWavesPerBlock, 0, (hipStream_t)stream>>>(betas, R, S, out);
    return nerf_check_launch("nerf_exclusive_cumprod");
}

extern "C" int nerf_exclusive_cumprod_bwd(const float* betas, const float* grad_out, int64_t R, int S, float* dbetas,
                                          void* stream) {
    NERF_REQUIRE(R >= 0 && S >= 1 && S <= 1024, "nerf_exclusive_cumprod_bwd: bad R=%lld S=%d (1..1024)", (long long)R, S);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(betas);
    NERF_REQUIRE_PTR(grad_out);
    NERF_REQUIRE_PTR(dbetas);
    NERF_REQUIRE(ceil_div64(R, kWavesPerBlock) < (1ll << 31), "nerf_exclusive_cumprod_bwd: too many rays");
    const unsigned grid = (unsigned)ceil_div64(R, kWavesPerBlock);
    hipStream_t st = (hipStream_t)stream;
    const int epl = (S + 63) / 64;
    if (epl <= 1) excl_cumprod_bwd_kernel<1><<<grid, 64 * kWavesPerBlock, 0, st>>>(betas, grad_out, R, S, dbetas);
    else if (epl <= 2) excl_cumprod_bwd_kernel<2><<<grid, 64 * kWavesPerBlock, 0, st>>>(betas, grad_out, R, S, dbetas);
    else if (epl <= 4) excl_cumprod_bwd_kernel<4><<<grid, 64 * kWavesPerBlock, 0, st>>>(betas, grad_out, R, S, dbetas);
    else if (epl <= 8) excl_cumprod_bwd_kernel<8><<<grid, 64 * kWavesPerBlock, 0, st>>>(betas, grad_out, R, S, dbetas);
    else excl_cumprod_bwd_kernel<16><<<grid, 64 * kWavesPerBlock, 0, st>>>(betas, grad_out, R, S, dbetas);
    return nerf_check_launch("nerf_exclusive_cumprod_bwd");
}
