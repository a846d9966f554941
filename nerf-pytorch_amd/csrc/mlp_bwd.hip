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

constexpr int BWD_LDS_BYTES = (ACT_FLOATS + lay::TILE) * 4;

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
    const int smp = tid >> 2, q = tid & 3;
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
    for (int i = 0; i < 8; ++i) {
        const int k = 4 * (4 * i + q);
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

// one data-gradient step on the tile: dX = dY * W (transposed image), epilogue, barriers
template <int MASK, bool RANK1>
__device__ __forceinline__ void dgrad_layer(float* act, const float* __restrict__ timg, int kb_count, int ncol0,
                                            int lane, const float* hsrc, float* gout, const float* dsig_s,
                                            const float* __restrict__ sigw,
                                            const unsigned long long* __restrict__ mask_word = nullptr) {
    f32x16 acc[2][2];
    init_bias<2>(acc, nullptr, ncol0, lane);
    unsigned long long bits = 0ull;
    if constexpr (MASK == 3) bits = *mask_word;  // one 8-byte load, in flight during the whole K loop
    mma_layer<2>(act, timg, lay::H, kb_count, ncol0, lane, acc);
    __syncthreads();
    bwd_epilogue<2, MASK, RANK1>(act, acc, ncol0, lane, hsrc, gout, lay::H, dsig_s, sigw, bits);
    __syncthreads();
}

__global__ __launch_bounds__(256, 2) void nerf_dgrad_kernel(BwdArgs a) {
    using namespace lay;
    using namespace lay::nerf;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    float* dsig_s = smem + ACT_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t m0 = (size_t)blockIdx.x * TILE;
    stagger_second_resident_group();
    const size_t Mp = a.Mp;
    const float* __restrict__ pk = a.pk;
    const float* ws = a.ws;
    float* bw = a.bw;
    const int ncol0 = wave * 64;

    head_bwd<false>(a, m0, act, dsig_s, pk + W11, ws + WS_G * Mp, ws + WS_SRAW * Mp, 1.0f, bw + BW_DZ11 * Mp,
                    bw + BW_DSIG * Mp, bw + BW_DZ10 * Mp, tid);
    __syncthreads();
    // rgb_head.0 (feature part): reduce over its 128 outputs -> gradient of the 256 features
    dgrad_layer<0, false>(act, pk + D10, HH / 8, ncol0, lane, nullptr, bw + BW_DZ9 * Mp + m0 * H, nullptr, nullptr);
    auto mskp = [&](int l) {
        return reinterpret_cast<const unsigned long long*>(ws + (WS_MASK + 8 * (size_t)l) * Mp + m0 * 8) + tid;
    };
    // block2.8: 256 feature rows on MFMA + the density row as a rank-1 term, ReLU mask of h8
    dgrad_layer<3, true>(act, pk + d_img(9), H / 8, ncol0, lane, nullptr, bw + (BW_DZ + 8 * (size_t)H) * Mp + m0 * H,
                         dsig_s, pk + SIGW, mskp(8));
    // layers 8..1 -> dz7..dz0 (layer 5 only through its feature columns)
    for (int l = 8; l >= 1; --l)
        dgrad_layer<3, false>(act, pk + d_img(l), H / 8, ncol0, lane, nullptr,
                              bw + (BW_DZ + (size_t)(l - 1) * H) * Mp + m0 * H, nullptr, nullptr, mskp(l - 1));
}

__global__ __launch_bounds__(256, 2) void siren_dgrad_kernel(BwdArgs a) {
    using namespace lay;
    using namespace lay::siren;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    float* dsig_s = smem + ACT_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t m0 = (size_t)blockIdx.x * TILE;
    stagger_second_resident_group();
    const size_t Mp = a.Mp;
    const float* __restrict__ pk = a.pk;
    const float* ws = a.ws;
    float* bw = a.bw;
    const int ncol0 = wave * 64;

    head_bwd<true>(a, m0, act, dsig_s, pk + WOUT, ws + WS_CG * Mp, ws + WS_SRAW * Mp, 10.0f, bw + BW_DZO * Mp,
                   bw + BW_DSIG * Mp, bw + BW_DZR * Mp, tid);
    __syncthreads();
    // rgb sine layer (feature part) -> gradient of feature_remap's output
    dgrad_layer<0, false>(act, pk + D_RGB, HH / 8, ncol0, lane, nullptr, bw + BW_DZF * Mp + m0 * H, nullptr, nullptr);
    // feature_remap + density branch (rank-1) -> dh7, times cos -> dz7
    dgrad_layer<2, true>(act, pk + D_FEAT, H / 8, ncol0, lane, ws + (WS_C + 7 * (size_t)H) * Mp + m0 * H,
                         bw + (BW_DZ + 7 * (size_t)H) * Mp + m0 * H, dsig_s, pk + SIGW);
    for (int l = 7; l >= 1; --l)
        dgrad_layer<2, false>(act, pk + d_img(l), H / 8, ncol0, lane, ws + (WS_C + (size_t)(l - 1) * H) * Mp + m0 * H,
                              bw + (BW_DZ + (size_t)(l - 1) * H) * Mp + m0 * H, nullptr, nullptr);
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
};

template <int NW, int KW>
constexpr int wgrad_lds_bytes() { return 2 * 32 * (NW + KW) * 4; }

template <int NW, int KW, int WN, int WK>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(WgradArgs a) {
    constexpr int TN = NW / WN / 32, TK = KW / WK / 32;
    constexpr int ROWF = NW + KW;             // floats per staged sample row
    constexpr int BUF = 32 * ROWF;            // floats per buffer
    constexpr int V4_ROW = ROWF / 4;          // 16-byte pieces per row
    constexpr int V4_TOT = 32 * V4_ROW;
    constexpr int NLD = (V4_TOT + 511) / 512;
    static_assert(TN >= 1 && TK >= 1 && WN * WK <= 8, "wave layout");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int c0 = (int)((int64_t)a.nchunks * bid / G), c1 = (int)((int64_t)a.nchunks * (bid + 1) / G);
    const bool active = wave < WN * WK;
    const bool late = active && wave >= 4;
    const int wn = wave / WK, wk = wave % WK;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = splat16(0.0f);
    float bsum = 0.0f;

    f32x4 stage[NLD];
    auto gload = [&](int chunk) {
        const size_t row0 = (size_t)chunk * 32;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * 512;
            if (v < V4_TOT) {
                const int row = v / V4_ROW, c4 = v - row * V4_ROW;
                const float* src = (c4 < NW / 4) ? a.dY + (row0 + row) * a.ldy + 4 * c4
                                                 : a.X + (row0 + row) * a.ldx + 4 * (c4 - NW / 4);
                stage[i] = *reinterpret_cast<const f32x4*>(src);
            }
        }
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * 512;
            if (v < V4_TOT) *reinterpret_cast<f32x4*>(buf + 4 * v) = stage[i];
        }
    };

    if (c0 < c1) {
        gload(c0);
        lstore(smem);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const float* buf = smem + ((c - c0) & 1) * BUF;
        float* nbuf = smem + (((c - c0) & 1) ^ 1) * BUF;
        const bool more = c + 1 < c1;
        if (more) gload(c + 1);
        if (active) {
            const float* ap = buf + wn * (NW / WN) + r + h * ROWF;
            const float* bp = buf + NW + wk * (KW / WK) + r + h * ROWF;
            // fragments of sample pair s+1 are read from LDS before the MFMAs of pair s (two named
            // register sets, pinned): otherwise every 4-8 MFMAs wait for an LDS round trip
            float av0[TN], bv0[TK], av1[TN], bv1[TK];
            auto lds_frags = [&](int s, float (&av)[TN], float (&bv)[TK]) {
#pragma unroll
                for (int i = 0; i < TN; ++i) av[i] = ap[2 * s * ROWF + 32 * i];
#pragma unroll
                for (int j = 0; j < TK; ++j) bv[j] = bp[2 * s * ROWF + 32 * j];
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
                // waves 4-7 share their SIMDs with waves 0-3: they stage the next chunk in the middle
                // of their MFMA block, waves 0-3 at its end, so the two never leave a pipe idle together
                if (s == 8 && more && late) lstore(nbuf);
                lds_frags(s + 1, av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
                mma(av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                lds_frags(s + 2 < 16 ? s + 2 : 15, av0, bv0);
                __builtin_amdgcn_sched_barrier(0);
                mma(av1, bv1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (a.bslab && tid < NW) {
#pragma unroll 8
            for (int s = 0; s < 32; ++s) bsum += buf[s * ROWF + tid];
        }
        if (more && !late) lstore(nbuf);
        __syncthreads();
    }
    if (active) {
        float* out = a.slab + (size_t)bid * NW * KW;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) {
                const int n0 = wn * (NW / WN) + 32 * i, k0 = wk * (KW / WK) + 32 * j;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
                    out[(size_t)(n0 + row) * KW + k0 + r] = acc[i][j][g];
                }
            }
    }
    if (a.bslab && tid < NW) a.bslab[(size_t)bid * NW + tid] = bsum;
}

// thin products on the VALU: slab[g][b][a] = sum_m P[m][a] * Q[m][b], a < AW (thread), b < NB <= 4.
// Optional column sums of P (wide bias) and of Q (thin bias).
struct ThinArgs {
    const float* P;  // [Mp][ldp]
    const float* Q;  // [Mp][ldq]
    int ldp, ldq, AW, NB;
    int64_t Mp;
    float* slab;     // [G][NB][AW]
    float* pslab;    // [G][AW] or NULL
    float* qslab;    // [G][4] or NULL
};

__global__ __launch_bounds__(256) void thin_kernel(ThinArgs a) {
    const int tid = threadIdx.x, G = gridDim.x, bid = blockIdx.x;
    const int64_t r0 = a.Mp * bid / G, r1 = a.Mp * (bid + 1) / G;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, ps = 0.f, qs[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid < a.AW) {
        constexpr int U = 8;  // rows in flight per thread: the loop is latency-bound otherwise
        int64_t m = r0;
        for (; m + U <= r1; m += U) {
            float p[U], q[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                p[u] = a.P[(m + u) * a.ldp + tid];
#pragma unroll
                for (int b = 0; b < 4; ++b) q[u][b] = b < a.NB ? a.Q[(m + u) * a.ldq + b] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ps += p[u];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[b] = fmaf(p[u], q[u][b], acc[b]);
                    qs[b] += q[u][b];
                }
            }
        }
        for (; m < r1; ++m) {
            const float p = a.P[m * a.ldp + tid];
            ps += p;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < a.NB) {
                    const float qv = a.Q[m * a.ldq + b];
                    acc[b] = fmaf(p, qv, acc[b]);
                    qs[b] += qv;
                }
        }
        for (int b = 0; b < a.NB; ++b) a.slab[((size_t)bid * a.NB + b) * a.AW + tid] = acc[b];
        if (a.pslab) a.pslab[(size_t)bid * a.AW + tid] = ps;
        if (a.qslab && tid == 0)
            for (int b = 0; b < 4; ++b) a.qslab[(size_t)bid * 4 + b] = qs[b];
    }
}

// out[row*s_row + col*s_col] = sum_g slab[g][row][col], row < rows, col < cols_valid
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ slab, int G, int rows, int cols,
                                                     int cols_valid, float* __restrict__ out, int s_row, int s_col) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const int row = i / cols, col = i - row * cols;
    if (col >= cols_valid) return;
    const size_t stride = (size_t)rows * cols;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
        s0 += slab[(size_t)g * stride + i];
        s1 += slab[(size_t)(g + 1) * stride + i];
        s2 += slab[(size_t)(g + 2) * stride + i];
        s3 += slab[(size_t)(g + 3) * stride + i];
    }
    for (; g < G; ++g) s0 += slab[(size_t)g * stride + i];
    out[(size_t)row * s_row + (size_t)col * s_col] = (s0 + s1) + (s2 + s3);
}

template <typename K>
int set_lds(K kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return nerf_fail(NERF_E_LAUNCH, "hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
    return NERF_OK;
}

struct Ctx {
    hipStream_t st;
    size_t Mp;
    float* slab;   // SLAB_FLOATS
    int rc;
};

void reduce(Ctx& c, const float* slab, int G, int rows, int cols, int cols_valid, float* out, int s_row, int s_col) {
    if (c.rc) return;
    reduce_kernel<<<(rows * cols + 255) / 256, 256, 0, c.st>>>(slab, G, rows, cols, cols_valid, out, s_row, s_col);
    c.rc = nerf_check_launch("wgrad reduce");
}

// dW[:, koff : koff+kvalid] (leading dimension ldw) = dY^T X ; db = column sums of dY (if db)
template <int NW, int KW, int WN, int WK>
void wgrad(Ctx& c, const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int kvalid, float* db) {
    if (c.rc) return;
    const int nchunks = (int)(c.Mp / 32);
    const int G = nchunks < lay::WG_MAX ? nchunks : lay::WG_MAX;
    float* bslab = c.slab + (size_t)G * NW * KW;
    WgradArgs a{dY, X, ldy, ldx, nchunks, c.slab, db ? bslab : nullptr};
    auto kern = wgrad_kernel<NW, KW, WN, WK>;
    if ((c.rc = set_lds(kern, wgrad_lds_bytes<NW, KW>()))) return;
    kern<<<G, 512, wgrad_lds_bytes<NW, KW>(), c.st>>>(a);
    if ((c.rc = nerf_check_launch("wgrad"))) return;
    reduce(c, c.slab, G, NW, KW, kvalid, dW, ldw, 1);
    if (db) reduce(c, bslab, G, 1, NW, NW, db, 0, 1);
}

// out[a*sa + b*sb] = sum_m P[m][a] Q[m][b]; optional db_wide[a] = sum_m P[m][a], db_thin[b] = sum_m Q[m][b]
void thin(Ctx& c, const float* P, int ldp, int AW, const float* Q, int ldq, int NB, float* out, int sa, int sb,
          float* db_wide, float* db_thin) {
    if (c.rc) return;
    int64_t G = (int64_t)c.Mp / 64;
    if (G > 512) G = 512;
    float* pslab = c.slab + (size_t)G * NB * AW;
    float* qslab = pslab + (size_t)G * AW;
    ThinArgs a{P, Q, ldp, ldq, AW, NB, (int64_t)c.Mp, c.slab, db_wide ? pslab : nullptr, db_thin ? qslab : nullptr};
    thin_kernel<<<(unsigned)G, 256, 0, c.st>>>(a);
    if ((c.rc = nerf_check_launch("thin wgrad"))) return;
    reduce(c, c.slab, (int)G, NB, AW, AW, out, sb, sa);
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
    NERF_REQUIRE(Mp / lay::TILE < (1ull << 31), "%s: too many samples", fn);
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
    const unsigned tiles = (unsigned)(Mp / TILE);
    hipStream_t st = (hipStream_t)stream;
    BwdArgs a{packed, M, Mp, dcolors, ddensity, colors, fwd_workspace, bwd_workspace};
    if (model == NERF_MODEL_NERF) {
        if ((rc = set_lds(nerf_dgrad_kernel, BWD_LDS_BYTES))) return rc;
        nerf_dgrad_kernel<<<tiles, 256, BWD_LDS_BYTES, st>>>(a);
    } else {
        if ((rc = set_lds(siren_dgrad_kernel, BWD_LDS_BYTES))) return rc;
        siren_dgrad_kernel<<<tiles, 256, BWD_LDS_BYTES, st>>>(a);
    }
    return nerf_check_launch("nerf_mlp_bwd_data");
}

extern "C" int nerf_mlp_bwd_weights(int model, int64_t M, const float* fwd_workspace, float* bwd_workspace, float* grads,
                                    void* stream) {
    using namespace lay;
    int rc;
    if ((rc = bwd_check_args("nerf_mlp_bwd_weights", model, grads, M, fwd_workspace, bwd_workspace))) return rc;
    const size_t Mp = padded_samples(M);
    hipStream_t st = (hipStream_t)stream;
    const float* ws = fwd_workspace;
    float* bw = bwd_workspace;
    if (model == NERF_MODEL_NERF) {
        using namespace lay::nerf;
        Ctx c{st, Mp, bw + BW_PER_SAMPLE * Mp, 0};
        auto W = [&](int layer) { return grads + param_off(2 * layer); };
        auto B = [&](int layer) { return grads + param_off(2 * layer + 1); };
        auto h = [&](int l) { return ws + (WS_H + (size_t)l * H) * Mp; };
        auto dz = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp; };
        const float* pe = ws + WS_PE * Mp;
        const float* de = ws + WS_DE * Mp;
        wgrad<256, 64, 8, 1>(c, dz(0), H, pe, PE_P, W(0), PE, PE, B(0));
        for (int l = 1; l <= 4; ++l) wgrad<256, 256, 2, 4>(c, dz(l), H, h(l - 1), H, W(l), H, H, B(l));
        wgrad<256, 256, 2, 4>(c, dz(5), H, h(4), H, W(5), H + PE, H, B(5));
        wgrad<256, 64, 8, 1>(c, dz(5), H, pe, PE_P, W(5) + H, H + PE, PE, nullptr);
        for (int l = 6; l <= 8; ++l) wgrad<256, 256, 2, 4>(c, dz(l), H, h(l - 1), H, W(l), H, H, B(l));
        wgrad<256, 256, 2, 4>(c, bw + BW_DZ9 * Mp, H, h(8), H, W(9), H, H, B(9));
        thin(c, h(8), H, H, bw + BW_DSIG * Mp, 1, 1, W(9) + (size_t)H * H, 1, 0, nullptr, B(9) + H);
        wgrad<128, 256, 2, 4>(c, bw + BW_DZ10 * Mp, HH, ws + WS_F * Mp, H, W(10), H + DE, H, B(10));
        wgrad<128, 32, 4, 1>(c, bw + BW_DZ10 * Mp, HH, de, DE_P, W(10) + H, H + DE, DE, nullptr);
        thin(c, ws + WS_G * Mp, HH, HH, bw + BW_DZ11 * Mp, 4, 3, W(11), 1, HH, nullptr, B(11));
        return c.rc;
    } else {
        using namespace lay::siren;
        Ctx c{st, Mp, bw + BW_PER_SAMPLE * Mp, 0};
        auto W = [&](int layer) { return grads + param_off(2 * layer); };
        auto B = [&](int layer) { return grads + param_off(2 * layer + 1); };
        auto h = [&](int l) { return ws + (WS_H + (size_t)l * H) * Mp; };
        auto dz = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp; };
        // layer 0: dW0[n][k] = sum_m dz0[m][n] xn[m][k]
        thin(c, dz(0), H, H, ws + WS_X * Mp, 4, 3, W(0), 3, 1, B(0), nullptr);
        for (int l = 1; l <= 7; ++l) wgrad<256, 256, 2, 4>(c, dz(l), H, h(l - 1), H, W(l), H, H, B(l));
        thin(c, h(7), H, H, bw + BW_DSIG * Mp, 1, 1, W(8), 1, 0, nullptr, B(8));
        wgrad<256, 256, 2, 4>(c, bw + BW_DZF * Mp, H, h(7), H, W(9), H, H, B(9));
        wgrad<128, 256, 2, 4>(c, bw + BW_DZR * Mp, HH, ws + WS_F * Mp, H, W(10), H + DE, H, B(10));
        wgrad<128, 32, 4, 1>(c, bw + BW_DZR * Mp, HH, ws + WS_DE * Mp, DE_P, W(10) + H, H + DE, DE, nullptr);
        thin(c, ws + WS_G * Mp, HH, HH, bw + BW_DZO * Mp, 4, 3, W(11), 1, HH, nullptr, B(11));
        return c.rc;
    }
}

extern "C" int nerf_mlp_bwd(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                            const float* colors, const float* fwd_workspace, float* bwd_workspace, float* grads,
                            void* stream) {
    NERF_REQUIRE_PTR(grads);
    int rc = nerf_mlp_bwd_data(model, packed, M, dcolors, ddensity, colors, fwd_workspace, bwd_workspace, stream);
    if (rc) return rc;
    return nerf_mlp_bwd_weights(model, M, fwd_workspace, bwd_workspace, grads, stream);
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
