// Stratified sampling, sample positions and positional encoding as stand-alone,
// HBM-bound kernels (the fused MLP kernel generates the same values in registers).
//
// Replaces nerf/rendering.py:6-64,82 and nerf/encoding.py:4-20 of the reference.
#include "common.h"
#include "sincos.h"

// HIP's __fadd_rn/__fmul_rn/__fsub_rn are plain operators, which hipcc's default
// -ffp-contract=fast would fuse into FMAs; the bit-exact sampling below needs every
// operation rounded on its own, so contraction is off for this whole file.
#pragma clang fp contract(off)

thread_local char g_nerf_err[256] = "";

extern "C" int nerf_abi_version(void) { return NERF_ABI_VERSION; }
extern "C" const char* nerf_last_error(void) { return g_nerf_err; }

// ---------------------------------------------------------------------------------------
// linspace edge i of [near, far] with `steps` points, as ATen's CPU fp32 kernel computes
// it: a fused multiply-add from the nearer end (one rounding).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float linspace_edge(float lo, float hi, float step, int i, int steps) {
    return (i < steps / 2) ? __fmaf_rn(step, (float)i, lo) : __fmaf_rn(-step, (float)(steps - 1 - i), hi);
}

__device__ __forceinline__ float strat_t(float lo_end, float hi_end, float step, int s, int S, float u) {
    const float lo = linspace_edge(lo_end, hi_end, step, s, S + 1);
    const float hi = linspace_edge(lo_end, hi_end, step, s + 1, S + 1);
    // lower + (upper - lower) * u with separately rounded mul and add (rendering.py:27)
    return add_rn(lo, mul_rn(sub_rn(hi, lo), u));
}

__global__ void stratified_t_kernel(float lo_end, float hi_end, float step, int S, const float* __restrict__ u,
                                    float* __restrict__ t, float* __restrict__ deltas) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const float ts = strat_t(lo_end, hi_end, step, s, S, u[s]);
    t[s] = ts;
    if (deltas) {
        if (s + 1 < S) {
            const float tn = strat_t(lo_end, hi_end, step, s + 1, S, u[s + 1]);
            deltas[s] = sub_rn(tn, ts);
        } else {
            deltas[s] = 1e10f;
        }
    }
}

extern "C" int nerf_stratified_t(double near, double far, int S, const float* u, float* t, float* deltas,
                                 void* stream) {
    NERF_REQUIRE(S >= 1 && S <= (1 << 20), "nerf_stratified_t: S=%d out of range", S);
    NERF_REQUIRE_PTR(u);
    NERF_REQUIRE_PTR(t);
    const float lo = (float)near, hi = (float)far;
    const float step = (hi - lo) / (float)S;  // (end-start)/(steps-1) in fp32, steps = S+1
    stratified_t_kernel<<<(S + 255) / 256, 256, 0, (hipStream_t)stream>>>(lo, hi, step, S, u, t, deltas);
    return nerf_check_launch("nerf_stratified_t");
}

// ---------------------------------------------------------------------------------------
// positions: one thread per 4 consecutive output floats (one 16-byte store, a wave writes 1 KiB
// contiguous); output float f belongs to sample f/3, coordinate f%3
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float normalize_coord(float p, float nearf, float spanf) {
    // 2*(p-near)/(far-near)-1: sub, mul by 2, true division, sub 1 (rendering.py:82)
    return sub_rn(div_rn(mul_rn(2.0f, sub_rn(p, nearf)), spanf), 1.0f);
}

__global__ __launch_bounds__(256) void sample_positions_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                              const float* __restrict__ t, int64_t R, int S,
                                                              int normalize, float nearf, float spanf,
                                                              float* __restrict__ pos) {
    // 64-bit index arithmetic only on workgroup-uniform values; per thread 32-bit offsets from them
    const int64_t total = R * S * 3;
    const int64_t fb = (int64_t)blockIdx.x * 1024;        // first output float of this workgroup
    const int64_t mb = fb / 3;                            // its sample ...
    const int cb = (int)(fb - mb * 3);                    // ... and coordinate
    const int64_t rb = mb / S;                            // ray and sample-in-ray of that sample
    const unsigned sb = (unsigned)(mb - rb * S);
    const int64_t f0 = fb + threadIdx.x * 4;
    if (f0 >= total) return;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned fl = threadIdx.x * 4 + i + cb;     // floats past (mb, coordinate 0)
        const unsigned ml = fl / 3, c = fl - ml * 3;
        const unsigned sl = sb + ml, rl = sl / (unsigned)S;
        int64_t r = rb + rl;
        const unsigned sidx = sl - rl * (unsigned)S;
        if (r >= R) r = R - 1;                            // lanes past the end: clamped, not stored
        float p = add_rn(o[r * 3 + c], mul_rn(t[sidx], d[r * 3 + c]));
        if (normalize) p = normalize_coord(p, nearf, spanf);
        v[i] = p;
    }
    if (f0 + 4 <= total) {
        *reinterpret_cast<f32x4*>(pos + f0) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        for (int i = 0; f0 + i < total; ++i) pos[f0 + i] = v[i];
    }
}

// S % 4 == 0: one thread per 4 consecutive samples of a ray -- one 16-byte load of their depths, the ray's origin and
// direction once, one integer division per thread.  The workgroup's 256 quads are contiguous in the output
// (3072 floats); they pass through LDS so that every store instruction of a wave writes 1 KiB contiguous
// (12 floats per lane written directly would put 16-byte pieces 48 bytes apart: about half the store rate).
__global__ __launch_bounds__(256) void sample_positions_quads_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                                    const float* __restrict__ t, int64_t R, int S,
                                                                    int normalize, float nearf, float spanf,
                                                                    float* __restrict__ pos) {
    __shared__ __attribute__((aligned(16))) float stage[256 * 12];
    const unsigned Q = (unsigned)S >> 2;                  // quads per ray
    const int64_t qb = (int64_t)blockIdx.x * 256;         // first quad of this workgroup (uniform)
    const int64_t rb = qb / Q;
    const unsigned x = (unsigned)(qb - rb * Q) + threadIdx.x;
    const unsigned rl = x / Q, sq = x - rl * Q;
    const int64_t r = rb + rl;
    if (r < R) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(t + 4 * sq);
        float oo[3], dd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { oo[c] = o[r * 3 + c]; dd[c] = d[r * 3 + c]; }
        float v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float p = add_rn(oo[c], mul_rn(tv[k], dd[c]));
                if (normalize) p = normalize_coord(p, nearf, spanf);
                v[3 * k + c] = p;
            }
        f32x4* mine = reinterpret_cast<f32x4*>(stage + threadIdx.x * 12);
#pragma unroll
        for (int q = 0; q < 3; ++q) mine[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
    __syncthreads();
    const int64_t f0 = qb * 12, total = R * S * 3;        // both multiples of 4
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int fl = (q * 256 + threadIdx.x) * 4;
        if (f0 + fl < total) *reinterpret_cast<f32x4*>(pos + f0 + fl) = *reinterpret_cast<const f32x4*>(stage + fl);
    }
}

extern "C" int nerf_sample_positions(const float* rays_o, const float* rays_d, const float* t, int64_t R, int S,
                                     int normalize, double near, double far, float* pos, void* stream) {
    NERF_REQUIRE(R >= 0 && S >= 1, "nerf_sample_positions: bad R=%lld S=%d", (long long)R, S);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(rays_o);
    NERF_REQUIRE_PTR(rays_d);
    NERF_REQUIRE_PTR(t);
    NERF_REQUIRE_PTR(pos);
    const int64_t quads = ceil_div64(R * S * 3, 4);
    NERF_REQUIRE(ceil_div64(quads, 256) < (1ll << 31), "nerf_sample_positions: too many samples");
    if (S % 4 == 0 && (((uintptr_t)t | (uintptr_t)pos) & 15) == 0) {
        sample_positions_quads_kernel<<<(unsigned)ceil_div64(R * (S / 4), 256), 256, 0, (hipStream_t)stream>>>(
            rays_o, rays_d, t, R, S, normalize, (float)near, (float)(far - near), pos);
        return nerf_check_launch("nerf_sample_positions");
    }
    sample_positions_kernel<<<(unsigned)ceil_div64(quads, 256), 256, 0, (hipStream_t)stream>>>(
        rays_o, rays_d, t, R, S, normalize, (float)near, (float)(far - near), pos);
    return nerf_check_launch("nerf_sample_positions");
}

// ---------------------------------------------------------------------------------------
// positional encoding: a workgroup encodes 64 points into LDS (one thread per point, frequency
// slot and coordinate; slot 0 copies x, slot j>=1 writes sin and cos of 2^(j-1) x with the
// full-range sincos: arguments reach ~1e3 rad), then writes its 64 rows, which are contiguous in the
// output, with 16-byte stores.
// ---------------------------------------------------------------------------------------
constexpr int PE_POINTS = 64;

template <int D>
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ x, int64_t M, int L, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float rows[];  // [PE_POINTS][width]
    const int per = D * (L + 1), width = D * (1 + 2 * L);
    const int64_t m0 = (int64_t)blockIdx.x * PE_POINTS;
    const int npts = (int)(M - m0 < PE_POINTS ? M - m0 : PE_POINTS);
    // lane = point (LDS row stride `width` is odd: conflict-free), the workgroup's four waves share the point's
    // per = D * (L+1) (slot, coordinate) pairs: wave-uniform slot, no per-thread division
    const int ml = threadIdx.x & 63;
    if (ml < npts) {
        float xv[D];
#pragma unroll
        for (int c = 0; c < D; ++c) xv[c] = x[(m0 + ml) * D + c];
        float* row = rows + ml * width;
        for (int jc = threadIdx.x >> 6; jc < per; jc += 4) {
            const int j = jc / D, c = jc - j * D;
            float v = xv[0];
#pragma unroll
            for (int k = 1; k < D; ++k) v = (c == k) ? xv[k] : v;
            if (j == 0) {
                row[c] = v;
            } else {
                float sn, cs;
                nerf_sincosf(mul_rn((float)(1u << (j - 1)), v), &sn, &cs);
                row[D + 2 * D * (j - 1) + c] = sn;
                row[D + 2 * D * (j - 1) + D + c] = cs;
            }
        }
    }
    __syncthreads();
    const int nfl = npts * width;             // m0 * width is a multiple of 4: the block's base is 16-byte aligned
    float* dst = out + m0 * width;
    for (int i = threadIdx.x * 4; i + 4 <= nfl; i += 1024)
        *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(rows + i);
    for (int i = (nfl & ~3) + threadIdx.x; i < nfl; i += 256) dst[i] = rows[i];
}

extern "C" int nerf_posenc_fwd(const float* x, int64_t M, int D, int L, float* out, void* stream) {
    NERF_REQUIRE(M >= 0 && D >= 1 && D <= 4 && L >= 0 && L <= 16, "nerf_posenc_fwd: bad M=%lld D=%d L=%d",
                 (long long)M, D, L);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE_PTR(x);
    NERF_REQUIRE_PTR(out);
    const int64_t blocks = ceil_div64(M, PE_POINTS);
    NERF_REQUIRE(blocks < (1ll << 31), "nerf_posenc_fwd: too many elements");
    const int lds = PE_POINTS * D * (1 + 2 * L) * 4;   // at most 64 * 132 * 4 = 33,792 bytes
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 1: posenc_kernel<1><<<(unsigned)blocks, 256, lds, st>>>(x, M, L, out); break;
        case 2: posenc_kernel<2><<<(unsigned)blocks, 256, lds, st>>>(x, M, L, out); break;
        case 3: posenc_kernel<3><<<(unsigned)blocks, 256, lds, st>>>(x, M, L, out); break;
        default: posenc_kernel<4><<<(unsigned)blocks, 256, lds, st>>>(x, M, L, out); break;
    }
    return nerf_check_launch("nerf_posenc_fwd");
}

// gradient of the encoding wrt x (what autograd derives for nerf/encoding.py:15-20):
//   dx[m,c] = g[m,c] + sum_j 2^j * (g_sin[j][m,c] * cos(2^j x) - g_cos[j][m,c] * sin(2^j x))
// one thread per input element; sin/cos are recomputed with the forward's routine.
__global__ __launch_bounds__(256) void posenc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        int64_t n, int D, int L, float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t m = i / D;
    const int c = (int)(i - m * D);
    const int width = D * (1 + 2 * L);
    const float* row = g + m * width;
    const float v = x[i];
    float acc = row[c];
    float scale = 1.0f;
#pragma unroll 1
    for (int j = 0; j < L; ++j) {
        float sn, cs;
        nerf_sincosf(mul_rn(scale, v), &sn, &cs);
        acc += scale * (row[D + 2 * D * j + c] * cs - row[D + 2 * D * j + D + c] * sn);
        scale *= 2.0f;
    }
    dx[i] = acc;
}

extern "C" int nerf_posenc_bwd(const float* x, const float* grad_out, int64_t M, int D, int L, float* dx, void* stream) {
    NERF_REQUIRE(M >= 0 && D >= 1 && D <= 4 && L >= 0 && L <= 16, "nerf_posenc_bwd: bad M=%lld D=%d L=%d",
                 (long long)M, D, L);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE_PTR(x);
    NERF_REQUIRE_PTR(grad_out);
    NERF_REQUIRE_PTR(dx);
    const int64_t blocks = ceil_div64(M * D, 256);
    NERF_REQUIRE(blocks < (1ll << 31), "nerf_posenc_bwd: too many elements");
    posenc_bwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, grad_out, M * D, D, L, dx);
    return nerf_check_launch("nerf_posenc_bwd");
}

// ---------------------------------------------------------------------------------------
// inverse-CDF sampling (no reference counterpart; canonical NeRF sample_pdf), in two steps:
//   weights -> CDF   (normalise w+1e-5, running sum; fp32, tolerance-tested)
//   (bins, CDF, u) -> right-sided insertion index + interpolated sample; the index is a pure
//   comparison result, hence bit-exact for an identical CDF and identical draws.
// One ray per wavefront.
// ---------------------------------------------------------------------------------------
// data-parallel-primitive moves (no LDS round trip): lanes the pattern leaves without a source read 0
#define NERF_DPP_ROW_SHR(n) (0x110 + (n))
#define NERF_DPP_ROW_SHL(n) (0x100 + (n))
#define NERF_DPP_ROW_ROR(n) (0x120 + (n))
#define NERF_DPP_ROW_BCAST15 0x142
#define NERF_DPP_ROW_BCAST31 0x143
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_or_zero(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}

// sum over the wave, in every lane
__device__ __forceinline__ float wave_total(float v) {
    v += dpp_or_zero<NERF_DPP_ROW_SHR(1)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_SHR(2)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_SHR(4)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_SHR(8)>(v);           // lane 15 of each row: the row's sum
    v += dpp_or_zero<NERF_DPP_ROW_BCAST15, 0xA>(v);     // rows 1 and 3 add the row before
    v += dpp_or_zero<NERF_DPP_ROW_BCAST31, 0xC>(v);     // rows 2 and 3 add lane 31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// inclusive prefix sum over the lanes of the wave
__device__ __forceinline__ float wave_inclusive_sum(float v) {
    v += dpp_or_zero<NERF_DPP_ROW_SHR(1)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_SHR(2)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_SHR(4)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_SHR(8)>(v);
    v += dpp_or_zero<NERF_DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_or_zero<NERF_DPP_ROW_BCAST31, 0xC>(v);
    return v;
}

// CDF of one ray by one wave: wr [B] weights -> cr [B+1] (global or LDS; each lane re-reads only what it wrote)
__device__ __forceinline__ void cdf_row(const float* __restrict__ wr, int B, float* cr, int lane) {
    float part = 0.f;
    for (int i = lane; i < B; i += 64) part += wr[i] + 1e-5f;
    const float tot = wave_total(part);
    // each lane owns a contiguous run of ceil(B/64) bins: normalised weights parked in cr, wave-exclusive scan of
    // the run sums, then the running sum replaces them
    const int per = (B + 63) / 64;
    const int b0 = lane * per;
    float run = 0.f;
    for (int i = 0; i < per; ++i)
        if (b0 + i < B) {
            const float q = (wr[b0 + i] + 1e-5f) / tot;
            cr[b0 + i + 1] = q;
            run += q;
        }
    float acc = wave_inclusive_sum(run) - run;
    if (lane == 0) cr[0] = 0.f;
    for (int i = 0; i < per; ++i)
        if (b0 + i < B) {
            acc += cr[b0 + i + 1];
            cr[b0 + i + 1] = acc;
        }
}

__global__ __launch_bounds__(256) void weights_to_cdf_kernel(const float* __restrict__ w, int64_t R, int B,
                                                            float* __restrict__ cdf) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= R) return;  // no block-wide barrier below: waves are independent
    cdf_row(w + r * B, B, cdf + r * (B + 1), lane);
}

extern "C" int nerf_weights_to_cdf(const float* weights, int64_t R, int B, float* cdf, void* stream) {
    NERF_REQUIRE(R >= 0 && B >= 1, "nerf_weights_to_cdf: bad R=%lld B=%d", (long long)R, B);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(weights);
    NERF_REQUIRE_PTR(cdf);
    weights_to_cdf_kernel<<<(unsigned)ceil_div64(R, 4), 256, 0, (hipStream_t)stream>>>(weights, R, B, cdf);
    return nerf_check_launch("nerf_weights_to_cdf");
}

// #{a[i] <= v} (LE) or #{a[i] < v} (!LE) of the E values v against the ascending a [n], n >= 1 and uniform over
// the wave: a halving search whose trip count depends on n only, so that the E searches of a lane advance
// together and every lane runs the same instructions.  The count lies in [base, base + len] throughout.
template <int E, bool LE>
__device__ __forceinline__ void count_below(const float* a, int n, const float (&v)[E], int (&cnt)[E]) {
    int base[E];
#pragma unroll
    for (int e = 0; e < E; ++e) base[e] = 0;
    for (int len = n; len > 1;) {
        const int half = len >> 1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float x = a[base[e] + half - 1];
            base[e] += (LE ? (x <= v[e]) : (x < v[e])) ? half : 0;
        }
        len -= half;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float x = a[base[e]];
        cnt[e] = base[e] + ((LE ? (x <= v[e]) : (x < v[e])) ? 1 : 0);
    }
}

// inverse-CDF draws: c, bn [B+1] (global or LDS) -> samples; idx = searchsorted(c, uu, right=True)
template <int E>
__device__ __forceinline__ void pdf_sample(const float* c, const float* bn, int B, const float (&uu)[E], float (&out)[E],
                                           int (&idx)[E]) {
    count_below<E, true>(c, B + 1, uu, idx);     // first index with cdf[idx] > uu
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int below = max(idx[e] - 1, 0), above = min(idx[e], B);
        const float c0 = c[below], c1 = c[above];
        float den = sub_rn(c1, c0);
        if (den < 1e-5f) den = 1.f;
        const float tt = div_rn(sub_rn(uu[e], c0), den);
        out[e] = add_rn(bn[below], mul_rn(tt, sub_rn(bn[above], bn[below])));
    }
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ cdf,
                                                        const float* __restrict__ u, int64_t R, int B, int N,
                                                        float* __restrict__ samples, int32_t* __restrict__ idx_out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * N) return;
    const int64_t r = e / N;
    const float uu[1] = {u[e]};
    float v[1];
    int lo[1];
    pdf_sample<1>(cdf + r * (B + 1), bins + r * (B + 1), B, uu, v, lo);
    samples[e] = v[0];
    idx_out[e] = lo[0];
}

// one ray per wave: the ray's CDF and bin edges staged in LDS (2 * (B+1) floats per wave), the draws two per lane at
// a time so that two searches are in flight
constexpr int PDF_LDS_MAX_EDGES = 2048;
__global__ __launch_bounds__(256) void sample_pdf_rows_kernel(const float* __restrict__ bins, const float* __restrict__ cdf,
                                                             const float* __restrict__ u, int64_t R, int B, int N,
                                                             float* __restrict__ samples, int32_t* __restrict__ idx_out) {
    extern __shared__ float pdf_lds[];     // per wave: cdf [B+1] | bins [B+1]
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= R) return;                    // waves are independent: no block-wide barrier below
    float* c = pdf_lds + (size_t)wv * 2 * (B + 1);
    float* bn = c + (B + 1);
    for (int i = lane; i <= B; i += 64) {
        c[i] = cdf[r * (B + 1) + i];
        bn[i] = bins[r * (B + 1) + i];
    }
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < N; i0 += 128) {
        const int i[2] = {i0 + lane, i0 + 64 + lane};
        const float uu[2] = {i[0] < N ? u[r * N + i[0]] : 0.f, i[1] < N ? u[r * N + i[1]] : 0.f};
        float v[2];
        int lo[2];
        pdf_sample<2>(c, bn, B, uu, v, lo);
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (i[e] < N) {
                samples[r * N + i[e]] = v[e];
                idx_out[r * N + i[e]] = lo[e];
            }
    }
}

extern "C" int nerf_sample_pdf(const float* bins, const float* cdf, const float* u, int64_t R, int B, int N,
                               float* samples, int32_t* idx, void* stream) {
    NERF_REQUIRE(R >= 0 && B >= 1 && N >= 1, "nerf_sample_pdf: bad R=%lld B=%d N=%d", (long long)R, B, N);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(bins);
    NERF_REQUIRE_PTR(cdf);
    NERF_REQUIRE_PTR(u);
    NERF_REQUIRE_PTR(samples);
    NERF_REQUIRE_PTR(idx);
    if (B + 1 <= PDF_LDS_MAX_EDGES) {
        NERF_REQUIRE(ceil_div64(R, 4) < (1ll << 31), "nerf_sample_pdf: too many rays");
        const size_t lds = 4 * 2 * (size_t)(B + 1) * sizeof(float);     // at most 64 KiB
        if (int rc = nerf_set_lds(sample_pdf_rows_kernel, (int)lds)) return rc;
        sample_pdf_rows_kernel<<<(unsigned)ceil_div64(R, 4), 256, lds, (hipStream_t)stream>>>(bins, cdf, u, R, B, N, samples, idx);
        return nerf_check_launch("nerf_sample_pdf");
    }
    NERF_REQUIRE(ceil_div64(R * N, 256) < (1ll << 31), "nerf_sample_pdf: too many samples");
    sample_pdf_kernel<<<(unsigned)ceil_div64(R * N, 256), 256, 0, (hipStream_t)stream>>>(bins, cdf, u, R, B, N, samples,
                                                                                      idx);
    return nerf_check_launch("nerf_sample_pdf");
}

// ---------------------------------------------------------------------------------------
// pinhole rays (nerf/data.py:65-114 compute_rays): per pixel (u, v) of an H x W image,
// camera-space direction (u - W/2, -(v - H/2), -focal), rotated by c2w[:3,:3], L2-normalised;
// origin = c2w[:3,3].  c2w: [N][4][4] row-major.  One thread per ray, 24 B written.
// ---------------------------------------------------------------------------------------
__global__ void compute_rays_kernel(const float* __restrict__ c2w, int64_t N, int H, int W, float focal,
                                    float* __restrict__ ro, float* __restrict__ rd) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)H * W;
    if (idx >= N * per) return;
    const int64_t n = idx / per;
    const int64_t p = idx - n * per;
    const int v = (int)(p / W), u = (int)(p - (int64_t)v * W);
    const float x = (float)u - 0.5f * (float)W;
    const float y = -((float)v - 0.5f * (float)H);
    const float z = -focal;
    const float* m = c2w + n * 16;
    float d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = add_rn(add_rn(mul_rn(m[4 * i], x), mul_rn(m[4 * i + 1], y)), mul_rn(m[4 * i + 2], z));
    const float nrm = sqrtf(add_rn(add_rn(mul_rn(d[0], d[0]), mul_rn(d[1], d[1])), mul_rn(d[2], d[2])));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rd[idx * 3 + i] = div_rn(d[i], nrm);
        ro[idx * 3 + i] = m[4 * i + 3];
    }
}

extern "C" int nerf_compute_rays(const float* c2w, int64_t N, int H, int W, double focal, float* rays_o, float* rays_d,
                                 void* stream) {
    NERF_REQUIRE(N >= 0 && H >= 1 && W >= 1, "nerf_compute_rays: bad N=%lld H=%d W=%d", (long long)N, H, W);
    if (N == 0) return NERF_OK;
    NERF_REQUIRE_PTR(c2w);
    NERF_REQUIRE_PTR(rays_o);
    NERF_REQUIRE_PTR(rays_d);
    const int64_t total = N * H * W;
    NERF_REQUIRE(ceil_div64(total, 256) < (1ll << 31), "nerf_compute_rays: too many rays");
    compute_rays_kernel<<<(unsigned)ceil_div64(total, 256), 256, 0, (hipStream_t)stream>>>(c2w, N, H, W, (float)focal,
                                                                                        rays_o, rays_d);
    return nerf_check_launch("nerf_compute_rays");
}

// ---------------------------------------------------------------------------------------
// sorted union of the shared coarse depths and each ray's fine depths (canonical NeRF:
// sort(cat(z_vals, z_samples))) plus the successive differences the compositor needs.
// One ray per wavefront.  The coarse depths are already ascending, so only the ray's fine depths are
// sorted: a bitonic network over 64*E values held E per lane in registers (padded with +inf), whose
// exchanges are register moves inside a lane and cross-lane shuffles between lanes -- no LDS traffic and
// every lane busy in every stage.  Every element then finds its place in the union by a binary search in
// the OTHER list: fine[i] goes to i + #{coarse <= fine[i]}, coarse[j] to j + #{fine < coarse[j]}.  A wave
// only touches its own LDS rows and LDS operations of one wave are carried out in program order, so the
// stages are separated by wave-level scheduling barriers only.
// ---------------------------------------------------------------------------------------
#define MERGE_MAX 1024

// value of lane ^ M: data-parallel-primitive moves where one exists (quad permutes, rotation by 8 in a row of 16),
// a cross-lane LDS permute otherwise
template <int M>
__device__ __forceinline__ float lane_xor(float v) {
    const int x = __float_as_int(v);
    if constexpr (M == 1) return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {     // banks 0 and 2 of a row (lanes with bit 2 clear) read 4 lanes up, banks 1 and 3 4 lanes down
        const int up = __builtin_amdgcn_update_dpp(x, x, NERF_DPP_ROW_SHL(4), 0xF, 0x5, false);
        return __int_as_float(__builtin_amdgcn_update_dpp(up, x, NERF_DPP_ROW_SHR(4), 0xF, 0xA, false));
    } else if constexpr (M == 8) return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true)); // row_ror:8
    else return __shfl_xor(v, M);
}

// one compare-exchange stage of the bitonic network (K: length of the bitonic runs being merged, J: partner distance)
template <int E, int K, int J>
__device__ __forceinline__ void sort_stage(float (&f)[E], int lane) {
    constexpr int N = 64 * E;
    // median of (a, b, -inf) is min(a, b), of (a, b, +inf) max(a, b): one instruction per value once the lane knows
    // which of the two it keeps in this stage
    if constexpr (J >= E) {             // partner in lane ^ (J/E), same register
        constexpr int M = J / E;
        // the lower index keeps the smaller value in an ascending run ((index & K) == 0), the larger in a descending one
        const bool keep_min = ((lane & M) == 0) == (K >= N || (lane & (K / E)) == 0);
        const float sel = keep_min ? -INFINITY : INFINITY;
#pragma unroll
        for (int e = 0; e < E; ++e) f[e] = __builtin_amdgcn_fmed3f(f[e], lane_xor<M>(f[e]), sel);
    } else if constexpr (K < E) {       // partner in the same lane, direction known at compile time
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((e & J) == 0) {
                const float x = f[e], y = f[e | J];
                const bool up = (e & K) == 0;
                f[e] = up ? fminf(x, y) : fmaxf(x, y);
                f[e | J] = up ? fmaxf(x, y) : fminf(x, y);
            }
    } else {                            // partner in the same lane, direction by lane
        const bool up = K >= N || (lane & (K / E)) == 0;
        const float sel_lo = up ? -INFINITY : INFINITY, sel_hi = up ? INFINITY : -INFINITY;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((e & J) == 0) {
                const float x = f[e], y = f[e | J];
                f[e] = __builtin_amdgcn_fmed3f(x, y, sel_lo);
                f[e | J] = __builtin_amdgcn_fmed3f(x, y, sel_hi);
            }
    }
}

template <int E, int K, int J>
__device__ __forceinline__ void sort_merge(float (&f)[E], int lane) {
    sort_stage<E, K, J>(f, lane);
    if constexpr (J > 1) sort_merge<E, K, J / 2>(f, lane);
}

template <int E, int K>
__device__ __forceinline__ void sort_runs(float (&f)[E], int lane) {
    sort_merge<E, K, K / 2>(f, lane);
    if constexpr (K < 64 * E) sort_runs<E, K * 2>(f, lane);
}

// ascending sort of the wave's 64*E values (no NaNs); value (lane, e) has index lane*E + e before and after
template <int E>
__device__ __forceinline__ void wave_sort(float (&f)[E], int lane) {
    sort_runs<E, 2>(f, lane);
}

// f: the ray's fine depths (Sf of them, the rest +inf), unsorted, E per lane; c [Sc] ascending coarse depths (LDS);
// b [64*E] LDS scratch; writes the sorted union to o [Sc+Sf] (LDS)
template <int E>
__device__ __forceinline__ void merge_row(float (&f)[E], float* b, const float* c, float* o, int Sc, int Sf, int lane) {
    wave_sort<E>(f, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) b[lane * E + e] = f[e];
    int cnt[E];
    count_below<E, true>(c, Sc, f, cnt);        // #{coarse <= fine value}
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        if (i < Sf) o[i + cnt[e]] = f[e];
    }
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < Sc; j += 64) {       // #{fine < coarse value}; b holds 64*E >= 1 sorted values, the padding last
        const float v[1] = {c[j]};
        int lo[1];
        count_below<1, false>(b, Sf > 0 ? Sf : 1, v, lo);
        o[j + (Sf > 0 ? lo[0] : 0)] = v[0];
    }
    __builtin_amdgcn_wave_barrier();
}

// the union and its successive differences, LDS -> global (row r)
__device__ __forceinline__ void store_union(const float* o, int S, int64_t r, float* __restrict__ t_all,
                                            float* __restrict__ deltas, int lane) {
    for (int i = lane; i < S; i += 64) {
        const float v = o[i];
        t_all[r * S + i] = v;
        deltas[r * S + i] = (i + 1 < S) ? sub_rn(o[i + 1], v) : 1e10f;
    }
}

template <int E>
__global__ __launch_bounds__(256) void merge_samples_kernel(const float* __restrict__ tc, const float* __restrict__ tf,
                                                           int64_t R, int Sc, int Sf, float* __restrict__ t_all,
                                                           float* __restrict__ deltas) {
    extern __shared__ float merge_lds[];   // per wave: fine [64*E] | coarse [Sc] | union [Sc+Sf]
    constexpr int N = 64 * E;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar row addresses
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    const int64_t rr = r < R ? r : R - 1;  // out-of-range waves redo the last ray and store nothing
    const int S = Sc + Sf;
    float* b = merge_lds + (size_t)wv * (N + Sc + S);
    float* c = b + N;
    float* o = c + Sc;
    float f[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        f[e] = i < Sf ? tf[rr * Sf + i] : INFINITY;
    }
    for (int i = lane; i < Sc; i += 64) c[i] = tc[i];
    __builtin_amdgcn_wave_barrier();
    merge_row<E>(f, b, c, o, Sc, Sf, lane);
    if (r < R) store_union(o, S, r, t_all, deltas, lane);
}

// Fused re-sampling of one ray per wave: CDF of the interior coarse weights over the mid-points of the coarse
// depths, Nf inverse-CDF draws, sorted union with the coarse depths, successive differences -- the chain
// weights_to_cdf -> sample_pdf -> merge_samples in ONE launch, built from the same device functions (bitwise
// the same results), with CDF, mid-points and fine depths kept in LDS / registers instead of going through HBM.
template <int E>
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ w, const float* __restrict__ tc,
                                                      const float* __restrict__ u, int64_t R, int Sc, int Sf,
                                                      float* __restrict__ t_all, float* __restrict__ deltas,
                                                      float* __restrict__ t_fine, int32_t* __restrict__ idx_out) {
    extern __shared__ float merge_lds[];   // per wave: fine [64*E] | coarse [Sc] | union [Sc+Sf] | cdf [Sc-1] | mid [Sc-1]
    constexpr int N = 64 * E;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: scalar row addresses
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    const int64_t rr = r < R ? r : R - 1;  // out-of-range waves redo the last ray and store nothing
    const int S = Sc + Sf, B = Sc - 2;
    float* b = merge_lds + (size_t)wv * (N + Sc + S + 2 * (Sc - 1));
    float* c = b + N;
    float* o = c + Sc;
    float* cdf = o + S;
    float* mid = cdf + (Sc - 1);
    float uu[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {               // issued first: the draws are the bulk of the ray's input
        const int i = lane * E + e;
        uu[e] = i < Sf ? u[rr * Sf + i] : 0.f;
    }
    for (int i = lane; i < Sc; i += 64) c[i] = tc[i];
    cdf_row(w + rr * Sc + 1, B, cdf, lane);                      // interior weights w[1 .. Sc-2]
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < Sc - 1; i += 64) mid[i] = mul_rn(0.5f, add_rn(c[i + 1], c[i]));
    __builtin_amdgcn_wave_barrier();
    float f[E];
    int lo[E];
    pdf_sample<E>(cdf, mid, B, uu, f, lo);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        if (r < R && i < Sf) {
            if (t_fine) t_fine[r * Sf + i] = f[e];
            if (idx_out) idx_out[r * Sf + i] = lo[e];
        }
        if (i >= Sf) f[e] = INFINITY;
    }
    merge_row<E>(f, b, c, o, Sc, Sf, lane);
    if (r < R) store_union(o, S, r, t_all, deltas, lane);
}

// values per lane of the sorting network for Sf fine depths: the power of two >= Sf / 64
static int sort_width(int Sf) {
    int E = 1;
    while (64 * E < Sf) E <<= 1;
    return E;
}

extern "C" int nerf_merge_samples(const float* t_coarse, const float* t_fine, int64_t R, int Sc, int Sf, float* t_all,
                                  float* deltas, void* stream) {
    NERF_REQUIRE(R >= 0 && Sc >= 1 && Sf >= 0 && Sc + Sf <= MERGE_MAX, "nerf_merge_samples: bad R=%lld Sc=%d Sf=%d",
                 (long long)R, Sc, Sf);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(t_coarse);
    if (Sf > 0) NERF_REQUIRE_PTR(t_fine);
    NERF_REQUIRE_PTR(t_all);
    NERF_REQUIRE_PTR(deltas);
    const int E = sort_width(Sf), N = 64 * E;
    const size_t lds = 4 * (size_t)(N + 2 * Sc + Sf) * sizeof(float);   // at most 4 * 3072 * 4 = 48 KiB
    const unsigned grid = (unsigned)ceil_div64(R, 4);
    hipStream_t st = (hipStream_t)stream;
#define NERF_MERGE_CASE(EE) case EE: merge_samples_kernel<EE><<<grid, 256, lds, st>>>(t_coarse, t_fine, R, Sc, Sf, t_all, deltas); break
    switch (E) { NERF_MERGE_CASE(1); NERF_MERGE_CASE(2); NERF_MERGE_CASE(4); NERF_MERGE_CASE(8); default: NERF_MERGE_CASE(16); }
#undef NERF_MERGE_CASE
    return nerf_check_launch("nerf_merge_samples");
}

extern "C" int nerf_resample(const float* weights, const float* t_coarse, const float* u, int64_t R, int Sc, int Sf,
                             float* t_all, float* deltas, float* t_fine, int32_t* idx, void* stream) {
    NERF_REQUIRE(R >= 0 && Sc >= 3 && Sf >= 1 && Sc + Sf <= MERGE_MAX, "nerf_resample: bad R=%lld Sc=%d Sf=%d (Sc >= 3, Sc+Sf <= %d)",
                 (long long)R, Sc, Sf, MERGE_MAX);
    if (R == 0) return NERF_OK;
    NERF_REQUIRE_PTR(weights);
    NERF_REQUIRE_PTR(t_coarse);
    NERF_REQUIRE_PTR(u);
    NERF_REQUIRE_PTR(t_all);
    NERF_REQUIRE_PTR(deltas);
    const int E = sort_width(Sf), N = 64 * E;
    const size_t lds = 4 * (size_t)(N + 2 * Sc + Sf + 2 * (Sc - 1)) * sizeof(float);   // at most 4 * 5118 * 4 = 80 KiB
    const unsigned grid = (unsigned)ceil_div64(R, 4);
    hipStream_t st = (hipStream_t)stream;
#define NERF_RESAMPLE_CASE(EE)                                                                              \
    case EE:                                                                                                \
        if (int rc = nerf_set_lds(resample_kernel<EE>, (int)lds)) return rc;                                \
        resample_kernel<EE><<<grid, 256, lds, st>>>(weights, t_coarse, u, R, Sc, Sf, t_all, deltas, t_fine, idx); \
        break
    switch (E) { NERF_RESAMPLE_CASE(1); NERF_RESAMPLE_CASE(2); NERF_RESAMPLE_CASE(4); NERF_RESAMPLE_CASE(8); default: NERF_RESAMPLE_CASE(16); }
#undef NERF_RESAMPLE_CASE
    return nerf_check_launch("nerf_resample");
}
