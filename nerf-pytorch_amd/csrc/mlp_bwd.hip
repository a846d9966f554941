// Backward of the fused MLP: a fused data-gradient chain per 64-sample tile (same tile
// geometry as the forward), then one weight-gradient GEMM per layer (dW = dY^T X reduced
// over all samples) with deterministic two-stage reduction, no atomics.
//
// This is what autograd derives for nerf/models.py NeRF.forward / Siren.forward
// (train.py:181 loss.backward()), restricted to the parameter gradients: the reference
// never asks for gradients wrt ray origins or directions.
#include "mlp_device.h"

using namespace dev;

namespace {

struct BwdArgs {
    const float* pk;
    int64_t M;
    size_t Mp;
    const float* dcolors;
    const float* ddensity;
    const float* colors;
    const float* ws;
    float* bw;
};

constexpr int BWD_LDS_BYTES = (ACT_FLOATS + TR) * 4;

// Tile prologue shared by both models: gradient of the 128 -> 3 sigmoid head.
//   dzo[c] = dcolors * c * (1 - c) * out_scale      -> bw_dzo [Mp][4]
//   dsig   = ddensity * dens_scale if sraw > 0      -> bw_dsig [Mp], dsig_s (LDS)
//   dzr[k] = (sum_c dzo[c] Wout[c][k]) * deriv[k]   -> act[:, 0:128], bw_dzr [Mp][128]
// deriv = (g > 0) when SINE == false (g = stored ReLU output), = stored cos otherwise.
template <bool SINE>
__device__ __forceinline__ void head_bwd(const BwdArgs& a, size_t m0, float* act, float* dsig_s,
                                         const float* __restrict__ wout, const float* __restrict__ gsrc,
                                         const float* __restrict__ sraw, float dens_scale, float* __restrict__ bw_dzo,
                                         float* __restrict__ bw_dsig, float* __restrict__ bw_dzr, int tid) {
    const int smp = tid / TPS, q = tid % TPS;
    const size_t m = m0 + smp;
    float dz[3] = {0.f, 0.f, 0.f};
    float ds = 0.f;
    if (m < (size_t)a.M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float col = a.colors[m * 3 + c];
            dz[c] = a.dcolors[m * 3 + c] * col * (1.0f - col);
        }
        ds = sraw[m] > 0.0f ? a.ddensity[m] * dens_scale : 0.0f;
    }
    if (q == 0) {
        f32x4 v = {dz[0], dz[1], dz[2], 0.0f};
        *reinterpret_cast<f32x4*>(bw_dzo + m * 4) = v;
        bw_dsig[m] = ds;
        dsig_s[smp] = ds;
    }
#pragma unroll
    for (int i = 0; i < 32 / TPS; ++i) {
        const int k = 4 * (TPS * i + q);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wout + k);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wout + lay::HH + k);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(wout + 2 * lay::HH + k);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gsrc + m * lay::HH + k);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = dz[0] * w0[e] + dz[1] * w1[e] + dz[2] * w2[e];
            v[e] = SINE ? d * gv[e] : (gv[e] > 0.0f ? d : 0.0f);
        }
        *reinterpret_cast<f32x4*>(act + smp * LDA + k) = v;
        *reinterpret_cast<f32x4*>(bw_dzr + m * lay::HH + k) = v;
    }
}

// one data-gradient step on the tile: dX = dY * W (transposed image), epilogue, barriers.
// COPY_COLS != 0 (= 8*kb_count): the layer's INPUT tile (dY) goes to gin in the backward workspace,
// copied by the MFMA loop that consumes it.
template <int MASK, bool RANK1, int COPY_COLS>
__device__ __forceinline__ void dgrad_layer(float* act, const float* __restrict__ timg, int kb_count, int ncol0,
                                            int tid, const float* hsrc, float* gin, const float* dsig_s,
                                            const float* __restrict__ sigw,
                                            const mask_t* __restrict__ mask_word = nullptr) {
    const int lane = tid & 63;
    f32x16 acc[MT][2];
    mask_t bits = 0;
    if constexpr (MASK == 3) bits = *mask_word;  // one 8-byte load, in flight during the whole K loop
    mma_layer<2, COPY_COLS, ACC_ZERO>(act, timg, lay::H, kb_count, ncol0, lane, acc, gin, tid);
    __syncthreads();
    bwd_epilogue<2, MASK, RANK1>(act, acc, ncol0, lane, hsrc, lay::H, dsig_s, sigw, bits);
    __syncthreads();
}

__global__ __launch_bounds__(256, WG_PER_CU) void nerf_dgrad_kernel(BwdArgs a) {
    using namespace lay;
    using namespace lay::nerf;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    float* dsig_s = smem + ACT_FLOATS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t m0 = (size_t)blockIdx.x * TR;
    const size_t Mp = a.Mp;
    const float* __restrict__ pk = a.pk;
    const float* ws = a.ws;
    float* bw = a.bw;
    const int ncol0 = wave * 64;
    auto dzp = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp + m0 * H; };

    head_bwd<false>(a, m0, act, dsig_s, pk + W11, ws + WS_G * Mp, ws + WS_SRAW * Mp, 1.0f, bw + BW_DZ11 * Mp,
                    bw + BW_DSIG * Mp, bw + BW_DZ10 * Mp, tid);
    __syncthreads();
    // rgb_head.0 (feature part): reduce over its 128 outputs -> gradient of the 256 features (dz9)
    dgrad_layer<0, false, 0>(act, pk + D10, HH / 8, ncol0, tid, nullptr, nullptr, nullptr, nullptr);
    auto mskp = [&](int l) {
        return reinterpret_cast<const mask_t*>(ws + (WS_MASK + 8 * (size_t)l) * Mp + m0 * 8) + tid;
    };
    // block2.8: 256 feature rows on MFMA + the density row as a rank-1 term, ReLU mask of h8 -> dz8
    dgrad_layer<3, true, H>(act, pk + d_img(9), H / 8, ncol0, tid, nullptr, bw + BW_DZ9 * Mp + m0 * H, dsig_s, pk + SIGW,
                            mskp(8));
    // layers 8..1: dz_l -> dz_(l-1) (layer 5 only through its feature columns)
#pragma unroll 1   // one copy of the layer body (instruction cache)
    for (int l = 8; l >= 1; --l)
        dgrad_layer<3, false, H>(act, pk + d_img(l), H / 8, ncol0, tid, nullptr, dzp(l), nullptr, nullptr, mskp(l - 1));
    store_tile(act, H, dzp(0), H, tid);
}

__global__ __launch_bounds__(256, WG_PER_CU) void siren_dgrad_kernel(BwdArgs a) {
    using namespace lay;
    using namespace lay::siren;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    float* dsig_s = smem + ACT_FLOATS;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t m0 = (size_t)blockIdx.x * TR;
    const size_t Mp = a.Mp;
    const float* __restrict__ pk = a.pk;
    const float* ws = a.ws;
    float* bw = a.bw;
    const int ncol0 = wave * 64;
    auto dzp = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp + m0 * H; };

    head_bwd<true>(a, m0, act, dsig_s, pk + WOUT, ws + WS_CG * Mp, ws + WS_SRAW * Mp, 10.0f, bw + BW_DZO * Mp,
                   bw + BW_DSIG * Mp, bw + BW_DZR * Mp, tid);
    __syncthreads();
    // rgb sine layer (feature part) -> gradient of feature_remap's output (dzf)
    dgrad_layer<0, false, 0>(act, pk + D_RGB, HH / 8, ncol0, tid, nullptr, nullptr, nullptr, nullptr);
    // feature_remap + density branch (rank-1) -> dh7, times cos -> dz7
    dgrad_layer<2, true, H>(act, pk + D_FEAT, H / 8, ncol0, tid, ws + (WS_C + 7 * (size_t)H) * Mp + m0 * H,
                            bw + BW_DZF * Mp + m0 * H, dsig_s, pk + SIGW);
#pragma unroll 1   // one copy of the layer body (instruction cache)
    for (int l = 7; l >= 1; --l)
        dgrad_layer<2, false, H>(act, pk + d_img(l), H / 8, ncol0, tid, ws + (WS_C + (size_t)(l - 1) * H) * Mp + m0 * H,
                                 dzp(l), nullptr, nullptr);
    store_tile(act, H, dzp(0), H, tid);
}

// =====================================================================================
// weight gradient: slab[g][n][k] = sum over workgroup g's samples of dY[m][n] * X[m][k]
// =====================================================================================
// 8 waves; wave (wn, wk) owns the (NW/WN) x (KW/WK) output tile.  Samples stream through LDS
// in chunks of 32 rows, double buffered; MFMA contracts 2 samples per instruction with
// A[row n][k = sample] = dY[sample][n], B[k = sample][col] = X[sample][col].
struct WgradArgs {
    const float* dY;   // [Mp][ldy], first NW columns used
    const float* X;    // [Mp][ldx], first KW columns used
    int ldy, ldx;
    int nchunks;       // Mp / 32
    float* slab;       // [G][NW*KW]
    float* bslab;      // [G][NW] column sums of dY, or NULL
    // RANK1 only: one more output row whose dY column q [Mp] lives apart from dY (the density row)
    const float* q;
    float* qslab;      // [G][KW] partials of sum_m q[m] X[m][:]
    float* qbslab;     // [G] partials of sum_m q[m]
};

template <int NW, int KW>
constexpr int wgrad_lds_bytes() { return 2 * (32 * (NW + KW) + 32) * 4; }

template <int NW, int KW, int WN, int WK, bool RANK1>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(WgradArgs a) {
    constexpr int TN = NW / WN / 32, TK = KW / WK / 32;
    constexpr int ROWF = NW + KW;             // floats per staged sample row
    constexpr int BUF = 32 * ROWF + 32;       // floats per buffer: dY, X, then 32 values of q
    static_assert(TN >= 1 && TK >= 1 && WN * WK <= 8 && TN != 3 && TK != 3, "wave layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int c0 = (int)((int64_t)a.nchunks * bid / G), c1 = (int)((int64_t)a.nchunks * (bid + 1) / G);
    const bool active = wave < WN * WK;
    const int wn = wave / WK, wk = wave % WK;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = splat16(0.0f);

    // the next chunk goes global -> LDS directly (no staging registers, no ds_write pass).  A buffer is
    // [32][NW] of dY followed by [32][KW] of X; 16-byte piece v of either part lands at byte 16*v of
    // that part, which is what global_load_lds requires of a wave: a wave-uniform LDS base plus
    // 16 bytes per lane (the per-lane SOURCE address follows the row pitch of the tensor in HBM)
    // Buffer form of the LDS-DMA: the resource covers this workgroup's rows, the chunk advances in the scalar offset and the
    // 32-bit lane offset never changes -- no vector instructions per chunk (they would cost matrix-pipe time).
    const __amdgpu_buffer_rsrc_t y_rsrc = make_rsrc(a.dY + (size_t)c0 * 32 * a.ldy, (unsigned)((size_t)(c1 - c0) * 32 * a.ldy * 4));
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(a.X + (size_t)c0 * 32 * a.ldx, (unsigned)((size_t)(c1 - c0) * 32 * a.ldx * 4));
    // piece i of a part: 512 consecutive 16-byte pieces, one per thread
    auto stage_piece = [&](__amdgpu_buffer_rsrc_t rsrc, int chunk_rel, int ld, auto width, float* dst, int i) {
        constexpr int W4 = decltype(width)::value / 4, TOT = 32 * W4;  // TOT is a multiple of 64
        const int soff = chunk_rel * 32 * ld * 4;
        const int v = tid + i * 512;
        const int voff = ((v / W4) * ld + 4 * (v % W4)) * 4;
        if (v < TOT)  // whole waves are in or out
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + 4 * (64 * wave + i * 512)),
                                                 16, voff, soff, 0, 0);
    };
    constexpr int YPIECES = (32 * NW / 4 + 511) / 512, XPIECES = (32 * KW / 4 + 511) / 512, NPIECES = YPIECES + XPIECES + (RANK1 ? 1 : 0);
    // One of the NPIECES LDS-DMA instructions of a thread for the next chunk.  An LDS-DMA instruction holds the issuing wave
    // for some tens of cycles, and all waves of the workgroup reach this point together: issued in a row at the top of a
    // chunk they leave the matrix pipe idle, one per MFMA block they disappear behind the MFMAs.
    auto stage_one = [&](int chunk, float* buf, int i) {
        if (i < YPIECES) stage_piece(y_rsrc, chunk - c0, a.ldy, std::integral_constant<int, NW>{}, buf, i);
        else if (i < YPIECES + XPIECES) stage_piece(x_rsrc, chunk - c0, a.ldx, std::integral_constant<int, KW>{}, buf + 32 * NW, i - YPIECES);
        else if (RANK1 && tid < 32)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.q + (size_t)chunk * 32 + tid),
                                             (__attribute__((address_space(3))) void*)(buf + 32 * ROWF), 4, 0, 0);
    };
    auto stage = [&](int chunk, float* buf) {
#pragma unroll
        for (int i = 0; i < NPIECES; ++i) stage_one(chunk, buf, i);
    };
    static_assert(NPIECES <= 16, "one piece per MFMA block of a chunk");
    // bias column sums: all 512 threads take part, thread (grp, c4) adds rows grp*BROWS.. of the four
    // columns 4*c4.. (16-byte LDS reads: a quarter of the LDS instructions of a column-per-thread split)
    constexpr int BGRP = 512 / (NW / 4), BROWS = 32 / BGRP;
    const int bc4 = tid % (NW / 4), bgrp = tid / (NW / 4), brow0 = bgrp * BROWS;
    f32x4 bsum4 = {0.f, 0.f, 0.f, 0.f};
    // rank-1 row: thread (grp, c4) adds q[row] * X[row][4*c4..] over rows grp*XROWS..
    constexpr int XGRP = 512 / (KW / 4), XROWS = 32 / XGRP;
    static_assert(BROWS >= 1 && (!RANK1 || XROWS == 4), "row split of the column sums");
    const int xc4 = tid % (KW / 4), xgrp = tid / (KW / 4), xrow0 = xgrp * XROWS;
    f32x4 xsum4 = {0.f, 0.f, 0.f, 0.f};
    float qsum = 0.0f;

    if (c0 < c1) stage(c0, smem);
    __syncthreads();  // also waits for the loads above (an LDS-DMA counts on vmcnt)
    for (int c = c0; c < c1; ++c) {
        const float* buf = smem + ((c - c0) & 1) * BUF;
        float* nbuf = smem + (((c - c0) & 1) ^ 1) * BUF;
        const bool more = c + 1 < c1;        // nbuf was last read before the previous barrier
        if (more && !active) stage(c + 1, nbuf);   // (waves without a tile have no MFMA blocks to hide the pieces behind)
        if (active) {
            // Tile i of this wave covers output rows n0 + TN*rho + i (rho = the MFMA row index) and tile j
            // columns k0 + TK*col + j: a lane's TN A values and TK B values of a sample are then adjacent
            // floats -- one vector read each instead of TN + TK scalar reads per sample pair.
            const float* ap = buf + wn * (NW / WN) + TN * r + h * NW;
            const float* bp = buf + 32 * NW + wk * (KW / WK) + TK * r + h * KW;
            // fragments of sample pair s+1 are read from LDS before the MFMAs of pair s (two named
            // register sets, pinned): otherwise every 4-8 MFMAs wait for an LDS round trip
            float av0[TN], bv0[TK], av1[TN], bv1[TK];
            auto lds_frags = [&](int s, float (&av)[TN], float (&bv)[TK]) {
                typedef float vecn __attribute__((ext_vector_type(TN == 3 ? 4 : TN)));
                typedef float veck __attribute__((ext_vector_type(TK == 3 ? 4 : TK)));
                if constexpr (TN == 1) av[0] = ap[2 * s * NW];
                else {
                    const vecn v = *reinterpret_cast<const vecn*>(ap + 2 * s * NW);
#pragma unroll
                    for (int i = 0; i < TN; ++i) av[i] = v[i];
                }
                if constexpr (TK == 1) bv[0] = bp[2 * s * KW];
                else {
                    const veck v = *reinterpret_cast<const veck*>(bp + 2 * s * KW);
#pragma unroll
                    for (int j = 0; j < TK; ++j) bv[j] = v[j];
                }
            };
            auto mma = [&](float (&av)[TN], float (&bv)[TK]) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            };
            lds_frags(0, av0, bv0);
#pragma unroll
            for (int s = 0; s < 16; s += 2) {
                lds_frags(s + 1, av1, bv1);
                if (s < NPIECES && more) stage_one(c + 1, nbuf, s);
                __builtin_amdgcn_sched_barrier(0);
                mma(av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                lds_frags(s + 2 < 16 ? s + 2 : 15, av0, bv0);
                if (s + 1 < NPIECES && more) stage_one(c + 1, nbuf, s + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (a.bslab) {
#pragma unroll
            for (int s = 0; s < BROWS; ++s) bsum4 += *reinterpret_cast<const f32x4*>(buf + (brow0 + s) * NW + 4 * bc4);
        }
        if constexpr (RANK1) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(buf + 32 * ROWF + xrow0);   // XROWS == 4 consecutive rows
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(buf + 32 * NW + (xrow0 + s) * KW + 4 * xc4);
#pragma unroll
                for (int e = 0; e < 4; ++e) xsum4[e] = fmaf(qv[s], xv[e], xsum4[e]);
                qsum += qv[s];
            }
        }
        __syncthreads();
    }
    if (active) {
        float* out = a.slab + (size_t)bid * NW * KW;
        const int n0 = wn * (NW / WN), k0 = wk * (KW / WK);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int rho = (g & 3) + 8 * (g >> 2) + 4 * h;
                float* dst = out + (size_t)(n0 + TN * rho + i) * KW + k0 + TK * r;   // TK adjacent columns
#pragma unroll
                for (int j = 0; j < TK; ++j) dst[j] = acc[i][j][g];
            }
    }
    // the last barrier of the loop has passed: the staging buffers are free for the cross-group sums
    // (2048 floats of bias partials [BGRP][NW], 2048 of rank-1 partials [XGRP][KW], XGRP of q sums)
    static_assert(2 * BUF >= 4096 + 64, "staging buffers too small for the partial sums");
    if (a.bslab) *reinterpret_cast<f32x4*>(smem + bgrp * NW + 4 * bc4) = bsum4;
    if constexpr (RANK1) {
        *reinterpret_cast<f32x4*>(smem + 2048 + xgrp * KW + 4 * xc4) = xsum4;
        if (xc4 == 0) smem[4096 + xgrp] = qsum;
    }
    __syncthreads();
    if (a.bslab && tid < NW) {
        float t = smem[tid];
#pragma unroll
        for (int g = 1; g < BGRP; ++g) t += smem[tid + g * NW];
        a.bslab[(size_t)bid * NW + tid] = t;
    }
    if constexpr (RANK1) {
        if (tid < KW) {
            float t = smem[2048 + tid];
#pragma unroll
            for (int g = 1; g < XGRP; ++g) t += smem[2048 + tid + g * KW];
            a.qslab[(size_t)bid * KW + tid] = t;
            if (tid == 0) {
                float u = smem[4096];
#pragma unroll
                for (int g = 1; g < XGRP; ++g) u += smem[4096 + g];
                a.qbslab[(size_t)bid * 4] = u;   // row of 4, three of them unused
            }
        }
    }
}


// ---- weight gradient on the bf16 matrix pipe (opt-in bf16x3 training mode) --------------------------------------
// Same product, same partial slabs and reduction as wgrad_kernel, with the arithmetic of mlp_x3.hip: every fp32 value
// of dY and X is split into three bf16 terms and a product is formed from six exact term products accumulated in
// fp32 (v_mfma_f32_32x32x16_bf16 contracts 16 samples per instruction).  The MFMA wants the contraction index
// contiguous per lane -- 8 consecutive SAMPLES of one feature -- while the workspace rows are samples.  The transposition
// rides on the split: a thread loads 8 samples x 4 adjacent features of a 32-sample chunk (eight 16-byte loads, a wave
// covers whole 1 KiB rows), splits its 32 values (each pair of consecutive samples gives one packed bf16 pair per
// term), and writes for each of its 4 features and 3 terms ONE 16-byte piece = 8 consecutive samples into the LDS
// image [term][feature][32 samples] (padded per feature quad, see WX3_QUAD_PITCH).  Every value is split once per workgroup, not once per wave.  The chunk after next is loaded into
// registers while the MFMAs of the current one run.  Bias column sums and the rank-1 row (density) are taken from the
// fp32 registers on the way.  8 waves as 2 x 4: wave (wn, wk) owns rows [wn*NW/2, +NW/2) x columns [wk*64, +64).
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
// A term plane in LDS: feature f's 32 samples (64 bytes) at (f / 4) * 272 + (f % 4) * 64 -- 16 bytes of padding per
// feature QUAD.  The staging writes of 16 consecutive quads (one 16-byte piece each) and the fragment reads of 16
// consecutive features then both fall on 64 distinct banks (measured with an 80-byte pitch per feature: the writes,
// four features = 320 bytes apart, collided four ways and cost a fifth of the kernel's cycles).
constexpr int WX3_QUAD_PITCH = 272;
__device__ __forceinline__ constexpr int wx3_feat_off(int f) { return (f >> 2) * WX3_QUAD_PITCH + (f & 3) * 64; }

template <int NW, int KW>
constexpr int wgrad_x3_lds_bytes() { return 3 * (NW + KW) / 4 * WX3_QUAD_PITCH; }

__device__ __forceinline__ unsigned wx3_pack2(float lo, float hi) {
    typedef float wf32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((wf32x2){lo, hi}, wbf16x2));
}
__device__ __forceinline__ void wx3_split2(float v0, float v1, unsigned (&w)[3]) {
#pragma clang fp contract(off)
    w[0] = wx3_pack2(v0, v1);
#pragma unroll
    for (int p = 1; p < 3; ++p) {
        v0 -= __builtin_bit_cast(float, w[p - 1] << 16);           // exact
        v1 -= __builtin_bit_cast(float, w[p - 1] & 0xffff0000u);
        w[p] = wx3_pack2(v0, v1);
    }
}

template <int NW, int KW, bool RANK1>
__global__ __launch_bounds__(512, 1) void wgrad_x3_kernel(WgradArgs a) {
    static_assert(KW == 256 && (NW == 256 || NW == 128), "shapes of the wide layers");
    constexpr int TN = NW / 2 / 32, TK = KW / 4 / 32;      // 32 x 32 tiles per wave: TN x 2
    constexpr int F = NW + KW, NQ = F / 4;                 // staged features per sample; feature quads
    constexpr int PLANE = F / 4 * WX3_QUAD_PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int c0 = (int)((int64_t)a.nchunks * bid / G), c1 = (int)((int64_t)a.nchunks * (bid + 1) / G);
    const int wn = wave >> 2, wk = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    // staging role: the first NW threads (whole waves) stage dY, the next KW threads X; a thread owns one sample octet
    // `so` of the chunk and one feature quad (`fq` counts quads over dY then X, the order of the LDS image)
    const bool stager = wave < F / 64;
    const bool is_y = wave < NW / 64;                       // wave-uniform: the buffer resources below stay scalar
    const int tt = is_y ? tid : tid - NW;
    const int nq = (is_y ? NW : KW) / 4;
    const int so = tt / nq, fq = (is_y ? 0 : NW / 4) + tt % nq;
    const float* src = is_y ? a.dY : a.X;
    const int ld = is_y ? a.ldy : a.ldx;
    static_assert(NW % 64 == 0 && KW % 64 == 0 && NQ > 0, "whole waves per tensor");

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = splat16(0.0f);
    f32x4 colsum = {0.f, 0.f, 0.f, 0.f};   // dY quads: column sums (bias gradient); X quads: sum_m q[m] X[m][:]
    float qsum = 0.0f;

    // Loads through a buffer resource covering this workgroup's rows: the thread's part of the address (its sample
    // octet and feature quad) is one constant 32-bit offset, the chunk and the sample inside the octet advance in the
    // scalar offset -- no vector address arithmetic per chunk.  One register set: the loads of the next chunk are issued
    // right after the current one has been split out of it, and land during the MFMAs.
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(src + (size_t)c0 * 32 * ld, (unsigned)((size_t)(c1 - c0) * 32 * ld * 4));
    const int voff = stager ? (8 * so * ld + 4 * (tt % nq)) * 4 : 0;
    auto load_chunk = [&](int c, f32x4 (&v)[8]) {
        if (stager) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((c - c0) * 32 + e) * ld * 4, 0));
        }
    };
    // split this thread's 8 samples x 4 features and publish them; bias / rank-1 sums on the way
    auto publish = [&](int c, const f32x4 (&v)[8]) {
        if (stager) {
            if (is_y) {
                if (a.bslab) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) colsum += v[e];
                }
            } else if constexpr (RANK1) {
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(a.q + (size_t)c * 32 + 8 * so);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(a.q + (size_t)c * 32 + 8 * so + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float qe = e < 4 ? q0[e] : q1[e - 4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) colsum[i] = fmaf(qe, v[e][i], colsum[i]);
                    qsum += qe;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wu32x4 piece[3];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    unsigned w[3];
                    wx3_split2(v[2 * e2][i], v[2 * e2 + 1][i], w);
#pragma unroll
                    for (int p = 0; p < 3; ++p) piece[p][e2] = w[p];
                }
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<wu32x4*>(wsm + p * PLANE + fq * WX3_QUAD_PITCH + i * 64 + 16 * so) = piece[p];
            }
        }
    };
    // 2 k-steps of 16 samples: six term products per tile pair
    auto mfma_phase = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wbf16x8 A[TN][3], B[TK][3];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    A[i][p] = *reinterpret_cast<const wbf16x8*>(wsm + p * PLANE + wx3_feat_off(wn * (NW / 2) + 32 * i + r) + 32 * ks + 16 * h);
#pragma unroll
            for (int j = 0; j < TK; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    B[j][p] = *reinterpret_cast<const wbf16x8*>(wsm + p * PLANE + wx3_feat_off(NW + wk * (KW / 4) + 32 * j + r) + 32 * ks + 16 * h);
            constexpr int TA[6] = {0, 0, 0, 1, 1, 2}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][TA[t]], B[j][TB[t]], acc[i][j], 0, 0, 0);
        }
    };
    f32x4 v[8];
    if (c0 < c1) load_chunk(c0, v);
    for (int c = c0; c < c1; ++c) {
        publish(c, v);
        if (c + 1 < c1) load_chunk(c + 1, v);      // in flight during the MFMAs below
        __syncthreads();
        mfma_phase();
        __syncthreads();
    }
    // ---- partial slab of this workgroup
    {
        float* out = a.slab + (size_t)bid * NW * KW;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = wn * (NW / 2) + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * h;
                    out[(size_t)row * KW + wk * (KW / 4) + 32 * j + r] = acc[i][j][g];
                }
    }
    // column sums: the four sample octets of a feature quad through LDS (the planes are free after the last barrier)
    float* red = reinterpret_cast<float*>(wsm);       // [4][F] floats + 4
    if (stager) *reinterpret_cast<f32x4*>(red + so * F + 4 * fq) = colsum;
    if (RANK1 && stager && fq == NW / 4) red[4 * F + so] = qsum;    // (the first X quad of every sample octet)
    __syncthreads();
    if (tid < F) {
        const float t = (red[tid] + red[F + tid]) + (red[2 * F + tid] + red[3 * F + tid]);
        if (tid < NW) {
            if (a.bslab) a.bslab[(size_t)bid * NW + tid] = t;
        } else if constexpr (RANK1) {
            a.qslab[(size_t)bid * KW + tid - NW] = t;
        }
    }
    if constexpr (RANK1) {
        if (tid == 0) a.qbslab[(size_t)bid * 4] = (red[4 * F] + red[4 * F + 1]) + (red[4 * F + 2] + red[4 * F + 3]);
    }
}

// thin products on the VALU: slab[g][b][a] = sum_m P[m][a] * Q[m][b], a < AW (128 or 256), b < NB <= 4,
// Q stored [Mp][4].  Optional column sums of P (wide bias) and of Q (thin bias).
// A thread owns 4 adjacent columns of P (16-byte loads, a whole row per 32 or 64 threads) and every
// RP-th row of the workgroup's row range, 4 rows in flight; the RP row groups are added through LDS.
struct ThinArgs {
    const float* P;  // [Mp][ldp]
    const float* Q;  // [Mp][4]
    int ldp, AW, NB;
    int64_t Mp;
    float* slab;     // [G][NB][AW]
    float* pslab;    // [G][AW] or NULL
    float* qslab;    // [G][4] or NULL
};

__global__ __launch_bounds__(256) void thin_kernel(ThinArgs a) {
    __shared__ __attribute__((aligned(16))) float red[8 * 5 * 128];  // [RP][5][AW]: RP*AW = 1024
    __shared__ float qred[8][4];
    const int tid = threadIdx.x, G = gridDim.x, bid = blockIdx.x;
    const int C4 = a.AW / 4, RP = 256 / C4;
    const int rg = tid / C4, c = tid - rg * C4;
    const int64_t r0 = a.Mp * bid / G, r1 = a.Mp * (bid + 1) / G;
    f32x4 acc[4], ps = {0.f, 0.f, 0.f, 0.f}, qs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    int64_t m = r0 + rg;
    for (; m + (int64_t)(U - 1) * RP < r1; m += (int64_t)U * RP) {
        f32x4 p[U], q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            p[u] = *reinterpret_cast<const f32x4*>(a.P + (m + (int64_t)u * RP) * a.ldp + 4 * c);
            q[u] = *reinterpret_cast<const f32x4*>(a.Q + (m + (int64_t)u * RP) * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ps += p[u];
            qs += q[u];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] += p[u] * q[u][b];
        }
    }
    for (; m < r1; m += RP) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.P + m * a.ldp + 4 * c);
        const f32x4 q = *reinterpret_cast<const f32x4*>(a.Q + m * 4);
        ps += p;
        qs += q;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] += p * q[b];
    }
    // red[rg][b][4c..4c+3], b = 4: column sums of P
#pragma unroll
    for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(red + (rg * 5 + b) * a.AW + 4 * c) = acc[b];
    *reinterpret_cast<f32x4*>(red + (rg * 5 + 4) * a.AW + 4 * c) = ps;
    if (c == 0)
#pragma unroll
        for (int b = 0; b < 4; ++b) qred[rg][b] = qs[b];
    __syncthreads();
    if (tid < a.AW) {
        for (int b = 0; b < 5; ++b) {
            if (b < 4 ? b >= a.NB : a.pslab == nullptr) continue;
            float t = 0.f;
            for (int g = 0; g < RP; ++g) t += red[(g * 5 + b) * a.AW + tid];
            if (b < 4) a.slab[((size_t)bid * a.NB + b) * a.AW + tid] = t;
            else a.pslab[(size_t)bid * a.AW + tid] = t;
        }
    }
    if (a.qslab && tid < 4) {
        float t = 0.f;
        for (int g = 0; g < RP; ++g) t += qred[g][tid];
        a.qslab[(size_t)bid * 4 + tid] = t;
    }
}

// ---- one reduction launch for all partial slabs of a network --------------------------------
// Every weight-gradient launch writes its partials into its own region of the slab area; when all of them
// are queued, ONE launch adds the G partial slabs of every region into the gradient buffer:
//     out[row*s_row + col*s_col] (+)= sum_g slab[g][row][col],  row < rows, col < cols_valid.
// A workgroup owns 64 adjacent float4 of one region; its four 64-lane groups each add every fourth slab
// (8 independent 16-byte loads in flight), and the groups are combined through LDS -- a fixed order, so
// the gradients are bitwise reproducible.  rows*cols is a multiple of 4 for every region.
struct ReduceJob {
    const float* slab;   // [G][rows*cols]
    float* out;
    int G, rows, cols, cols_valid;
    int s_row, s_col;
    int block0;          // first workgroup of this job in the launch
};
constexpr int kMaxReduceJobs = 32;
struct ReduceArgs {
    int njobs, acc;      // acc != 0: add to what `out` holds (gradient accumulation across calls)
    ReduceJob jobs[kMaxReduceJobs];
};

__global__ __launch_bounds__(256) void reduce_all_kernel(ReduceArgs a) {
    __shared__ f32x4 part[4][64];
    int ji = 0;
    for (int k = 1; k < a.njobs; ++k)
        if ((int)blockIdx.x >= a.jobs[k].block0) ji = k;     // block0 ascends with the job index
    const ReduceJob j = a.jobs[ji];
    const int q = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i4 = ((int)blockIdx.x - j.block0) * 64 + q;    // this thread's float4 inside one slab
    const int n4 = j.rows * j.cols / 4;
    f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (i4 < n4) {
        const f32x4* base = reinterpret_cast<const f32x4*>(j.slab) + i4;
        const size_t stride4 = (size_t)n4;
        int g = grp;
        for (; g + 28 < j.G; g += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(g + 4 * u) * stride4];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u & 1] += v[u];
        }
        for (; g < j.G; g += 4) s[0] += base[(size_t)g * stride4];
    }
    part[grp][q] = s[0] + s[1];
    __syncthreads();
    if (grp == 0 && i4 < n4) {
        const f32x4 t = (part[0][q] + part[1][q]) + (part[2][q] + part[3][q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * i4 + e, row = i / j.cols, col = i - row * j.cols;
            if (col < j.cols_valid) {
                float* o = j.out + (size_t)row * j.s_row + (size_t)col * j.s_col;
                *o = (a.acc ? *o : 0.0f) + t[e];
            }
        }
    }
}

struct Ctx {
    hipStream_t st;
    size_t Mp;
    float* slab;      // SLAB_FLOATS: every launch takes the next free region
    size_t used;
    int rc;
    ReduceArgs red;
    int blocks;

    // region of G slabs of n floats each (n a multiple of 4); NULL (and rc set) when the area is exhausted
    float* take(int G, int n) {
        const size_t need = (size_t)G * n;
        if (rc) return nullptr;
        if ((n & 3) || used + need > lay::SLAB_FLOATS) {
            rc = nerf_fail(NERF_E_ARG, "weight gradient: partial-slab area exhausted (%zu + %zu floats)", used, need);
            return nullptr;
        }
        float* p = slab + used;
        used += need;
        return p;
    }
};

void reduce(Ctx& c, const float* slab, int G, int rows, int cols, int cols_valid, float* out, int s_row, int s_col) {
    if (c.rc) return;
    if (c.red.njobs >= kMaxReduceJobs) {
        c.rc = nerf_fail(NERF_E_ARG, "weight gradient: more than %d reduction jobs", kMaxReduceJobs);
        return;
    }
    c.red.jobs[c.red.njobs++] = ReduceJob{slab, out, G, rows, cols, cols_valid, s_row, s_col, c.blocks};
    c.blocks += (rows * cols / 4 + 63) / 64;
}

int reduce_all(Ctx& c) {
    if (c.rc) return c.rc;
    if (c.red.njobs == 0) return NERF_OK;
    reduce_all_kernel<<<c.blocks, 256, 0, c.st>>>(c.red);
    return nerf_check_launch("weight gradient reduction");
}

// dW[:, 0:kvalid] (leading dimension ldw) = dY^T X ; db = column sums of dY (if db).
// RANK1: additionally qrow[0:kvalid] = sum_m q[m] X[m][:] and *qbias = sum_m q[m].
template <int NW, int KW, int WN, int WK, bool RANK1 = false>
void wgrad(Ctx& c, const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int kvalid, float* db,
           const float* q = nullptr, float* qrow = nullptr, float* qbias = nullptr) {
    if (c.rc) return;
    const int nchunks = (int)(c.Mp / 32);
    const int G = nchunks < lay::WG_MAX ? nchunks : lay::WG_MAX;
    float* wslab = c.take(G, NW * KW);
    float* bslab = db ? c.take(G, NW) : nullptr;
    float* qslab = RANK1 ? c.take(G, KW) : nullptr;
    float* qbslab = RANK1 ? c.take(G, 4) : nullptr;   // one value per slab, padded to a 16-byte row
    if (c.rc) return;
    WgradArgs a{dY, X, ldy, ldx, nchunks, wslab, bslab, q, qslab, qbslab};
    auto kern = wgrad_kernel<NW, KW, WN, WK, RANK1>;
    if ((c.rc = nerf_set_lds(kern, wgrad_lds_bytes<NW, KW>()))) return;
    kern<<<G, 512, wgrad_lds_bytes<NW, KW>(), c.st>>>(a);
    if ((c.rc = nerf_check_launch("wgrad"))) return;
    reduce(c, wslab, G, NW, KW, kvalid, dW, ldw, 1);
    if (db) reduce(c, bslab, G, 1, NW, NW, db, 0, 1);
    if (RANK1) {
        reduce(c, qslab, G, 1, KW, kvalid, qrow, 0, 1);
        reduce(c, qbslab, G, 1, 4, 1, qbias, 0, 1);
    }
}

// the same product on the bf16 matrix pipe (wgrad_x3_kernel): same slabs, same reduction jobs
template <int NW, int KW, bool RANK1 = false>
void wgrad_x3(Ctx& c, const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int kvalid, float* db,
              const float* q = nullptr, float* qrow = nullptr, float* qbias = nullptr) {
    if (c.rc) return;
    const int nchunks = (int)(c.Mp / 32);
    const int G = nchunks < lay::WG_MAX ? nchunks : lay::WG_MAX;
    float* wslab = c.take(G, NW * KW);
    float* bslab = db ? c.take(G, NW) : nullptr;
    float* qslab = RANK1 ? c.take(G, KW) : nullptr;
    float* qbslab = RANK1 ? c.take(G, 4) : nullptr;
    if (c.rc) return;
    WgradArgs a{dY, X, ldy, ldx, nchunks, wslab, bslab, q, qslab, qbslab};
    auto kern = wgrad_x3_kernel<NW, KW, RANK1>;
    if ((c.rc = nerf_set_lds(kern, wgrad_x3_lds_bytes<NW, KW>()))) return;
    kern<<<G, 512, wgrad_x3_lds_bytes<NW, KW>(), c.st>>>(a);
    if ((c.rc = nerf_check_launch("wgrad_x3"))) return;
    reduce(c, wslab, G, NW, KW, kvalid, dW, ldw, 1);
    if (db) reduce(c, bslab, G, 1, NW, NW, db, 0, 1);
    if (RANK1) {
        reduce(c, qslab, G, 1, KW, kvalid, qrow, 0, 1);
        reduce(c, qbslab, G, 1, 4, 1, qbias, 0, 1);
    }
}

// out[a*sa + b*sb] = sum_m P[m][a] Q[m][b] (Q stored [Mp][4]); optional db_wide[a] = sum_m P[m][a],
// db_thin[b] = sum_m Q[m][b]
void thin(Ctx& c, const float* P, int ldp, int AW, const float* Q, int NB, float* out, int sa, int sb, float* db_wide,
          float* db_thin) {
    if (c.rc) return;
    if ((AW != 128 && AW != 256) || NB < 1 || NB > 4) {
        c.rc = nerf_fail(NERF_E_ARG, "thin wgrad: unsupported shape %d x %d", AW, NB);
        return;
    }
    int64_t G = (int64_t)c.Mp / 64;
    if (G > lay::THIN_MAX) G = lay::THIN_MAX;
    float* tslab = c.take((int)G, NB * AW);
    float* pslab = db_wide ? c.take((int)G, AW) : nullptr;
    float* qslab = db_thin ? c.take((int)G, 4) : nullptr;
    if (c.rc) return;
    ThinArgs a{P, Q, ldp, AW, NB, (int64_t)c.Mp, tslab, pslab, qslab};
    thin_kernel<<<(unsigned)G, 256, 0, c.st>>>(a);
    if ((c.rc = nerf_check_launch("thin wgrad"))) return;
    reduce(c, tslab, (int)G, NB, AW, AW, out, sb, sa);
    if (db_wide) reduce(c, pslab, (int)G, 1, AW, AW, db_wide, 0, 1);
    if (db_thin) reduce(c, qslab, (int)G, 1, 4, NB, db_thin, 0, 1);
}

}  // namespace

static int bwd_check_args(const char* fn, int model, const float* packed, int64_t M, const float* fwd_workspace,
                          float* bwd_workspace) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "%s: unknown model %d", fn, model);
    NERF_REQUIRE(M >= 1, "%s: M=%lld < 1", fn, (long long)M);
    if (!packed || !fwd_workspace || !bwd_workspace) return nerf_fail(NERF_E_NULL, "%s: NULL pointer argument", fn);
    const size_t Mp = lay::padded_samples(M);
    NERF_REQUIRE(Mp / TR < (1ull << 31), "%s: too many samples", fn);
    return NERF_OK;
}

extern "C" int nerf_mlp_bwd_data(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                                 const float* colors, const float* fwd_workspace, float* bwd_workspace, void* stream) {
    using namespace lay;
    int rc;
    if ((rc = bwd_check_args("nerf_mlp_bwd_data", model, packed, M, fwd_workspace, bwd_workspace))) return rc;
    NERF_REQUIRE_PTR(dcolors);
    NERF_REQUIRE_PTR(ddensity);
    NERF_REQUIRE_PTR(colors);
    const size_t Mp = padded_samples(M);
    const unsigned tiles = (unsigned)(Mp / TR);
    hipStream_t st = (hipStream_t)stream;
    BwdArgs a{packed, M, Mp, dcolors, ddensity, colors, fwd_workspace, bwd_workspace};
    if (model == NERF_MODEL_NERF) {
        if ((rc = nerf_set_lds(nerf_dgrad_kernel, BWD_LDS_BYTES))) return rc;
        nerf_dgrad_kernel<<<tiles, 256, BWD_LDS_BYTES, st>>>(a);
    } else {
        if ((rc = nerf_set_lds(siren_dgrad_kernel, BWD_LDS_BYTES))) return rc;
        siren_dgrad_kernel<<<tiles, 256, BWD_LDS_BYTES, st>>>(a);
    }
    return nerf_check_launch("nerf_mlp_bwd_data");
}

static int bwd_weights_impl(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads, int accumulate,
                            void* stream, bool x3);

extern "C" int nerf_mlp_bwd_weights(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads,
                                    int accumulate, void* stream) {
    return bwd_weights_impl(model, M, fwd_workspace, bwd_workspace, grads, accumulate, stream, false);
}

// the wide products (256 x 256, 128 x 256: 95 % of the flops) on the bf16 matrix pipe with three-term operands; the
// narrow ones and the reduction as in nerf_mlp_bwd_weights
extern "C" int nerf_mlp_bwd_weights_x3(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads,
                                       int accumulate, void* stream) {
    return bwd_weights_impl(model, M, fwd_workspace, bwd_workspace, grads, accumulate, stream, true);
}

static int bwd_weights_impl(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads, int accumulate,
                            void* stream, bool x3) {
    using namespace lay;
    int rc;
    if ((rc = bwd_check_args("nerf_mlp_bwd_weights", model, grads, M, fwd_workspace, bwd_workspace))) return rc;
    const size_t Mp = padded_samples(M);
    hipStream_t st = (hipStream_t)stream;
    const float* ws = fwd_workspace;
    float* bw = bwd_workspace;
    if (model == NERF_MODEL_NERF) {
        using namespace lay::nerf;
        Ctx c{st, Mp, bw + BW_PER_SAMPLE * Mp, 0, 0, {}, 0};
        c.red.njobs = 0;
        c.red.acc = accumulate != 0;
        auto W = [&](int layer) { return grads + param_off(2 * layer); };
        auto B = [&](int layer) { return grads + param_off(2 * layer + 1); };
        auto h = [&](int l) { return ws + (WS_H + (size_t)l * H) * Mp; };
        auto dz = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp; };
        const float* pe = ws + WS_PE * Mp;
        const float* de = ws + WS_DE * Mp;
        // the wide products: fp32 MFMA, or the bf16 matrix pipe in the bf16x3 training mode
        auto wide = [&](const float* dY, const float* X, float* dW, int ldw, float* db) {
            if (x3) wgrad_x3<256, 256>(c, dY, H, X, H, dW, ldw, H, db);
            else wgrad<256, 256, 2, 4>(c, dY, H, X, H, dW, ldw, H, db);
        };
        wgrad<256, 64, 8, 1>(c, dz(0), H, pe, PE_P, W(0), PE, PE, B(0));
        for (int l = 1; l <= 4; ++l) wide(dz(l), h(l - 1), W(l), H, B(l));
        wide(dz(5), h(4), W(5), H + PE, B(5));
        wgrad<256, 64, 8, 1>(c, dz(5), H, pe, PE_P, W(5) + H, H + PE, PE, nullptr);
        for (int l = 6; l <= 8; ++l) wide(dz(l), h(l - 1), W(l), H, B(l));
        // block2.8: rows 0..255 from dz9, row 256 (density) is the rank-1 product of dsig with the same h8
        if (x3)
            wgrad_x3<256, 256, true>(c, bw + BW_DZ9 * Mp, H, h(8), H, W(9), H, H, B(9), bw + BW_DSIG * Mp,
                                     W(9) + (size_t)H * H, B(9) + H);
        else
            wgrad<256, 256, 2, 4, true>(c, bw + BW_DZ9 * Mp, H, h(8), H, W(9), H, H, B(9), bw + BW_DSIG * Mp,
                                        W(9) + (size_t)H * H, B(9) + H);
        if (x3) wgrad_x3<128, 256>(c, bw + BW_DZ10 * Mp, HH, ws + WS_F * Mp, H, W(10), H + DE, H, B(10));
        else wgrad<128, 256, 2, 4>(c, bw + BW_DZ10 * Mp, HH, ws + WS_F * Mp, H, W(10), H + DE, H, B(10));
        wgrad<128, 32, 4, 1>(c, bw + BW_DZ10 * Mp, HH, de, DE_P, W(10) + H, H + DE, DE, nullptr);
        thin(c, ws + WS_G * Mp, HH, HH, bw + BW_DZ11 * Mp, 3, W(11), 1, HH, nullptr, B(11));
        return reduce_all(c);
    } else {
        using namespace lay::siren;
        Ctx c{st, Mp, bw + BW_PER_SAMPLE * Mp, 0, 0, {}, 0};
        c.red.njobs = 0;
        c.red.acc = accumulate != 0;
        auto W = [&](int layer) { return grads + param_off(2 * layer); };
        auto B = [&](int layer) { return grads + param_off(2 * layer + 1); };
        auto h = [&](int l) { return ws + (WS_H + (size_t)l * H) * Mp; };
        auto dz = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp; };
        // layer 0: dW0[n][k] = sum_m dz0[m][n] xn[m][k]
        thin(c, dz(0), H, H, ws + WS_X * Mp, 3, W(0), 3, 1, B(0), nullptr);
        for (int l = 1; l <= 7; ++l) {
            if (x3) wgrad_x3<256, 256>(c, dz(l), H, h(l - 1), H, W(l), H, H, B(l));
            else wgrad<256, 256, 2, 4>(c, dz(l), H, h(l - 1), H, W(l), H, H, B(l));
        }
        // feature_remap and the density branch (rank-1 product of dsig) both contract with h7
        if (x3) wgrad_x3<256, 256, true>(c, bw + BW_DZF * Mp, H, h(7), H, W(9), H, H, B(9), bw + BW_DSIG * Mp, W(8), B(8));
        else wgrad<256, 256, 2, 4, true>(c, bw + BW_DZF * Mp, H, h(7), H, W(9), H, H, B(9), bw + BW_DSIG * Mp, W(8), B(8));
        if (x3) wgrad_x3<128, 256>(c, bw + BW_DZR * Mp, HH, ws + WS_F * Mp, H, W(10), H + DE, H, B(10));
        else wgrad<128, 256, 2, 4>(c, bw + BW_DZR * Mp, HH, ws + WS_F * Mp, H, W(10), H + DE, H, B(10));
        wgrad<128, 32, 4, 1>(c, bw + BW_DZR * Mp, HH, ws + WS_DE * Mp, DE_P, W(10) + H, H + DE, DE, nullptr);
        thin(c, ws + WS_G * Mp, HH, HH, bw + BW_DZO * Mp, 3, W(11), 1, HH, nullptr, B(11));
        return reduce_all(c);
    }
}

extern "C" int nerf_mlp_bwd(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                            const float* colors, const float* fwd_workspace, float* bwd_workspace, float* grads,
                            int accumulate, void* stream) {
    NERF_REQUIRE_PTR(grads);
    int rc = nerf_mlp_bwd_data(model, packed, M, dcolors, ddensity, colors, fwd_workspace, bwd_workspace, stream);
    if (rc) return rc;
    return nerf_mlp_bwd_weights(model, M, fwd_workspace, bwd_workspace, grads, accumulate, stream);
}

// =====================================================================================
// Adam on a flat buffer (torch.optim.Adam defaults as used at train.py:123)
// =====================================================================================
namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_bc1,
                                                   float sqrt_bc2, float b1, float b2, float eps, float gscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    // exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2)
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = p[i] - lr_bc1 * (mi / denom);
}
}  // namespace

extern "C" int nerf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int step, double lr, double beta1,
                              double beta2, double eps, double grad_scale, void* stream) {
    NERF_REQUIRE(n >= 0 && step >= 1, "nerf_adam_step: bad n=%lld step=%d", (long long)n, step);
    if (n == 0) return NERF_OK;
    NERF_REQUIRE_PTR(p);
    NERF_REQUIRE_PTR(g);
    NERF_REQUIRE_PTR(m);
    NERF_REQUIRE_PTR(v);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    adam_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, (hipStream_t)stream>>>(
        p, g, m, v, n, (float)(lr / bc1), (float)sqrt(bc2), (float)beta1, (float)beta2, (float)eps,
        (float)grad_scale);
    return nerf_check_launch("nerf_adam_step");
}
