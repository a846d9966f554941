// Opt-in TRAINING kernels of the NeRF MLP (nerf/models.py:62-75) and, further down, the Siren MLP (:185-203), with what
// autograd derives for them (train.py:181), on the
// bf16 matrix pipe at fp32 accuracy ("bf16x3"): forward with the activation workspace, and the data-gradient chain.
// The NeRF model's forward is the 16-sample-tile kernel of mlp_x3i.hip (same workspace, its own layout of the ReLU
// decisions); this file holds its data-gradient chain and both Siren kernels.
// The default training path stays the fp32 MFMA kernels of mlp_fwd.hip / mlp_bwd.hip; the weight-gradient GEMMs read
// the same workspaces (nerf_mlp_bwd_weights on fp32 MFMA, or nerf_mlp_bwd_weights_x3 with this file's arithmetic).
//
// Arithmetic (the three-term expansion of mlp_bf16.hip): every fp32 operand is the exact sum of three bf16 terms,
// x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); a product of two operands is formed from
// the six term products that reach fp32 precision, w*x ~= w0 x0 + w0 x1 + w0 x2 + w1 x0 + w1 x1 + w2 x0, each exact in
// the MFMA's fp32 accumulator -- six v_mfma_f32_32x32x16_bf16 (32 cycles per 16 k) in place of eight
// v_mfma_f32_32x32x2_f32 (64 cycles per 2 k): 0.375 of the matrix-pipe time, the rounding of an fp32 accumulation
// and nothing else.
//
// Structure: the fp32 tile kernels' (same 64-sample tile, same wave -> output-column split, same accumulator register
// map, same workspace), so that masks, activations and layer gradients land where nerf_mlp_bwd_weights expects them.
// What differs:
//   * the layer input lives in LDS as THREE bf16 planes [64][264] (one per term) instead of one fp32 tile; the
//     epilogue splits every output value once (13 vector instructions per value pair) and writes the three terms;
//   * weights come from three bf16 images per layer (k-steps of 16: one 16-byte load per lane, plane and k-step);
//   * a wave's two 32-column accumulator tiles interleave their columns (tile ni holds columns ncol0 + 2 r + ni),
//     so that a lane owns two ADJACENT columns of every row: one packed bf16 pair per LDS write, one 8-byte store per
//     row for the fp32 copy of the tile that goes to the workspace straight from the registers (the fp32 kernels
//     copy it from their fp32 LDS tile inside the next MFMA loop; there is no fp32 tile here);
//   * 101 KiB of planes allow one workgroup per CU, so the workgroup is EIGHT waves: waves 0..3 own rows 0..31 of
//     the tile, waves 4..7 rows 32..63 (a 32 x 64 accumulator block each) -- two waves per SIMD, so that one wave's
//     MFMAs cover the other's fragment waits, and the dependent chains of the splitting epilogue interleave.  Waves w
//     and w + 4 fetch the same weight fragments (the second fetch hits the CU's vector cache).  The per-sample phases
//     (encodings, heads: four threads per sample) run on the first 256 threads.
#include "mlp_device.h"

using namespace dev;

// mlp_x3i.hip: the NeRF model's kernels of this mode
size_t nerf_x3i_dgrad_image_bytes();
int nerf_x3i_dgrad_pack(const float* const* params, void* image, void* stream);
int nerf_x3i_dgrad(const void* image, const float* w11, int64_t M, const float* dcolors, const float* ddensity, const float* colors,
                   const float* fwd_workspace, float* bwd_workspace, void* stream);
int nerf_x3i_fwd_train(const void* image, int64_t M, const float* points, const float* dirs, const float* rays_o,
                       const float* rays_d, const float* t, int S, int t_per_ray, double near, double far, float* colors,
                       float* density, float* workspace, void* stream);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

static_assert(TR == 64 && TPS == 4, "the bf16x3 training kernels are written for 64-row tiles");
constexpr int X3_THREADS = 512;

// ---- LDS ----------------------------------------------------------------------------------------------
constexpr int LDPB = 528;                        // bytes per plane row: 256 bf16 + 16 bytes (bank spread of the A reads)
constexpr int PLANE_BYTES = TR * LDPB;
constexpr int PLANES_BYTES = 3 * PLANE_BYTES;    // 101,376
constexpr int X3_XIN_FLOATS = TR * 8;
// table | planes | xin | sraw
constexpr int X3_FWD_LDS_BYTES = SINCOS_TAB_FLOATS * 4 + PLANES_BYTES + X3_XIN_FLOATS * 4 + TR * 4;
constexpr int X3_BWD_LDS_BYTES = PLANES_BYTES + TR * 4;
// fp32 views of the plane area while no layer input lives there: the 128-wide rgb hidden tile [64][132]
constexpr int LDG = 132;
static_assert(TR * LDG * 4 <= PLANES_BYTES, "fp32 view of the plane area");

// ---- bf16x3 images (units: bf16 elements) ---------------------------------------------------------------
// One dense layer, K padded to 16: three planes of [K/16][N][16] bf16; element (ks, pos, h, j) of a plane is term p of
// W[col(pos)][16 ks + 8 h + j].  col(pos) = pos for the 128-wide layer; for 256-wide outputs the two 32-column tiles
// of a wave interleave: position 64 w + 32 ni + r holds column 64 w + 2 r + ni.
struct X3Job {
    int tensor, src_off, sn, sk;   // W element (n, k) = src[tensor][src_off + n*sn + k*sk]
    int N, K, Kp;
    int neg, interleave;
    size_t dst;                    // bf16 offset of the layer's first plane
};
constexpr int kMaxX3Jobs = 24;
struct X3PackArgs {
    const float* src[NERF_NUM_PARAM_TENSORS];
    __bf16* dst;
    int njobs;
    X3Job jobs[kMaxX3Jobs];
};

__device__ __forceinline__ __bf16 term_of(float x, int plane) {
    __bf16 t = (__bf16)x;
    for (int i = 0; i < plane; ++i) {
        x -= (float)t;      // exact
        t = (__bf16)x;
    }
    return t;
}

__global__ __launch_bounds__(256) void pack_x3_kernel(X3PackArgs a) {
    const X3Job j = a.jobs[blockIdx.x];
    const float* __restrict__ src = a.src[j.tensor] + j.src_off;
    const int pieces_per_plane = j.Kp * j.N / 8;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < 3 * pieces_per_plane; i += gridDim.y * 256) {
        const int p = i / pieces_per_plane, q = i - p * pieces_per_plane;
        const int h = q & 1, pos = (q >> 1) % j.N, ks = (q >> 1) / j.N;
        const int n = j.interleave ? (pos & ~63) + 2 * (pos & 31) + ((pos >> 5) & 1) : pos;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * h + e;
            float w = (k < j.K) ? src[(size_t)n * j.sn + (size_t)k * j.sk] : 0.0f;
            if (j.neg) w = 0.0f - w;     // ReLU layers run negated (fwd_epilogue of mlp_device.h); terms of -w are minus the terms of w
            v[e] = term_of(w, p);
        }
        *reinterpret_cast<bf16x8*>(a.dst + j.dst + (size_t)p * j.Kp * j.N + (size_t)q * 8) = v;
    }
}

void x3_fwd_job(X3PackArgs& a, int tensor, int ld, int col0, int N, int K, size_t dst, bool neg, bool inter) {
    a.jobs[a.njobs++] = X3Job{tensor, col0, ld, 1, N, K, (K + 15) / 16 * 16, neg, inter, dst};
}
void x3_bwd_job(X3PackArgs& a, int tensor, int ld, int Nin, int Kout, size_t dst) {
    // transposed: reduce over the Kout output rows, image columns = the layer's first Nin input columns
    a.jobs[a.njobs++] = X3Job{tensor, 0, 1, ld, Nin, Kout, (Kout + 15) / 16 * 16, 0, 1, dst};
}

// ---- K loop ---------------------------------------------------------------------------------------------
// acc[mi][ni] (+)= sum over KS k-steps of the six term products of planes (LDS, rows 32 mi + r) and the three image
// planes (global, position ncol0 + 32 ni + r).  Fully unrolled (KS is 2, 4, 8 or 16): fragment sets rotate by
// compile-time index; the weight fragments of k-step ks + 2 and the activation fragments of ks + 1 are requested before
// the 24 (NT = 2) MFMAs of k-step ks -- with one wave per SIMD nothing else hides the L2 latency.
// What a K loop needs from global memory before its first MFMA: the weight fragments of its first PFB - 1 k-steps and the
// bias of the wave's columns.  Requested by the caller BEFORE the previous layer's epilogue (x3_prefetch): the L2 round
// trip then runs under the epilogue, and -- this device has one counter for loads and stores -- the wait for these
// fragments does not also wait for the epilogue's workspace stores, which are younger.
#ifndef X3_PFB
#define X3_PFB 3
#endif
constexpr int PFB = X3_PFB;                      // weight fragments are requested PFB - 1 k-steps ahead
template <int NT>
struct X3Pre {
    bf16x8 b[PFB - 1][NT][3];
    float bias[NT];
};
template <int NT>
__device__ __forceinline__ void x3_load_b(__amdgpu_buffer_rsrc_t rsrc, int b_voff, int plane_elems, int N, int ks,
                                          bf16x8 (&b)[NT][3]) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            b[ni][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsrc, b_voff + ni * 32 * 32, (p * plane_elems + ks * N * 16) * 2, 0));
}
template <int NT, int KS, int INIT>
__device__ __forceinline__ void x3_prefetch(const __bf16* __restrict__ img, int N, int ncol0, int lane,
                                            const float* __restrict__ bias, X3Pre<NT>& pre) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        // the accumulator's column: interleaved for the two-tile layers (see the header of this file)
        const int col = NT == 2 ? ncol0 + 2 * r + ni : ncol0 + r;
        pre.bias[ni] = (INIT == ACC_BIAS) ? bias[col] : 0.0f;
    }
    const int plane_elems = KS * 16 * N;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(img, (unsigned)(3 * plane_elems * 2));
    const int b_voff = ((ncol0 + r) * 16 + 8 * h) * 2;
#pragma unroll
    for (int i = 0; i < PFB - 1; ++i)
        if (i < KS) x3_load_b<NT>(rsrc, b_voff, plane_elems, N, i, pre.b[i]);
}

template <int NT, int KS, int INIT>
__device__ __forceinline__ void mma_layer_x3(const unsigned char* planes, const __bf16* __restrict__ img, int N, int ncol0,
                                             int wrow0, int lane, f32x16 (&acc)[NT], const X3Pre<NT>& pre) {
    const int r = lane & 31, h = lane >> 5;
    const int plane_elems = KS * 16 * N;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(img, (unsigned)(3 * plane_elems * 2));
    const int b_voff = ((ncol0 + r) * 16 + 8 * h) * 2;
    const unsigned char* a_ptr = planes + (wrow0 + r) * LDPB + 16 * h;
    bf16x8 A[2][3], B[PFB][NT][3];
    auto load_a = [&](int ks, bf16x8 (&a)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(a_ptr + p * PLANE_BYTES + ks * 32);
    };
#pragma unroll
    for (int i = 0; i < PFB - 1; ++i)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int p = 0; p < 3; ++p) B[i][ni][p] = pre.b[i][ni][p];
    load_a(0, A[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + PFB - 1 < KS) x3_load_b<NT>(rsrc, b_voff, plane_elems, N, ks + PFB - 1, B[(ks + PFB - 1) % PFB]);
        if (ks + 1 < KS) load_a(ks + 1, A[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&a)[3] = A[ks & 1];
        const bf16x8 (&b)[NT][3] = B[ks % PFB];
        f32x16 c[NT];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            if (ks == 0 && INIT == ACC_ZERO) c[ni] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            else if (ks == 0 && INIT == ACC_BIAS) c[ni] = splat16(pre.bias[ni]);
            else c[ni] = acc[ni];
        }
        // the six term products, alternating between the wave's accumulators (every product is exact, the sum rounds
        // like an fp32 accumulation whatever the order)
        constexpr int TA[6] = {0, 0, 0, 1, 1, 2}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) c[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[ni][TB[t]], c[ni], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[ni] = c[ni];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- splitting ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
// the three bf16 terms of a pair of floats, each as a packed pair (low half = first value)
__device__ __forceinline__ void split2(float v0, float v1, unsigned (&w)[3]) {
#pragma clang fp contract(off)
    w[0] = pack2(v0, v1);
#pragma unroll
    for (int p = 1; p < 3; ++p) {
        v0 -= __builtin_bit_cast(float, w[p - 1] << 16);           // exact
        v1 -= __builtin_bit_cast(float, w[p - 1] & 0xffff0000u);
        w[p] = pack2(v0, v1);
    }
}
__device__ __forceinline__ void store_terms(unsigned char* planes, int byte_off, const unsigned (&w)[3]) {
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(planes + p * PLANE_BYTES + byte_off) = w[p];
}
// one value's three terms into (row, col) of the planes (the encodings, whose threads own scattered columns)
__device__ __forceinline__ void store_value(unsigned char* planes, int row, int col, float v) {
    unsigned w[3];
    split2(v, 0.0f, w);
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<unsigned short*>(planes + p * PLANE_BYTES + row * LDPB + col * 2) = (unsigned short)w[p];
}

// Epilogue of a 256-wide layer (two interleaved tiles per wave): the fp32 tile to the workspace, the three terms to the
// planes.
//   MODE 0: forward, no activation          MODE 4: backward, nothing
// gout: this tile's [64][256] fp32 rows in the workspace (NULL: not stored).  TO_PLANES false: the last layer of a chain.
template <int MODE, bool TO_PLANES>
__device__ __forceinline__ void x3_epilogue(unsigned char* planes, f32x16 (&acc)[2], int ncol0, int wrow0, int lane,
                                            float* __restrict__ gout) {
    const int r = lane & 31, h = lane >> 5;
    const int col = ncol0 + 2 * r;
    const __amdgpu_buffer_rsrc_t g_rsrc = make_rsrc(gout, (unsigned)(TR * lay::H * 4));
    const int g_voff = ((wrow0 + 4 * h) * lay::H + col) * 4;
    unsigned char* pl = planes + (wrow0 + 4 * h) * LDPB + col * 2;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int row0 = (g & 3) + 8 * (g >> 2);      // + wrow0 + 4 h
        const float v[2] = {acc[0][g], acc[1][g]};
        if (gout) {
            // 8 bytes per lane, 256 contiguous bytes per row and store (a 64-bit store is exempt from the wide-store
            // data hazard that keeps the 16-byte stores of the fp32 kernels on immediate offsets)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, (f32x2){v[0], v[1]}), g_rsrc, g_voff,
                                                  row0 * lay::H * 4, 2);
        }
        if constexpr (TO_PLANES) {
            unsigned w[3];
            split2(v[0], v[1], w);
            store_terms(pl, row0 * LDPB, w);
        }
    }
}

// ---- forward --------------------------------------------------------------------------------------------
struct X3FwdArgs {
    const float* pk;       // fp32 image (thin vectors, biases)
    const __bf16* x3;      // the bf16x3 planes
    int64_t M;
    size_t Mp;
    const float* points;
    const float* dirs;
    const float* rays_o;
    const float* rays_d;
    const float* t;
    int S;
    int t_per_ray;
    float nearf, spanf;
    float* colors;
    float* density;
    float* ws;
};

__device__ __forceinline__ float norm_coord_x3(float p, float nearf, float spanf) {
#pragma clang fp contract(off)  // each operation rounded on its own, as the reference's tensor ops are
    return sub_rn(div_rn(mul_rn(2.0f, sub_rn(p, nearf)), spanf), 1.0f);
}

// xin[s][0..2] = normalised position, xin[s][3..5] = direction (zeros beyond M); xin[6], xin[7] (padding of row 0): the
// tile's largest |position| / |direction| component (range check of the table-form encodings)
__device__ __forceinline__ void x3_load_inputs(const X3FwdArgs& a, size_t m0, float* xin, int tid) {
#pragma clang fp contract(off)  // o + t*d must not become an FMA (bit-exact positions)
    if (tid < TR) {
        const size_t m = m0 + tid;
        float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
        if (m < (size_t)a.M) {
            if (a.points) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { x[c] = a.points[m * 3 + c]; d[c] = a.dirs[m * 3 + c]; }
            } else {
                const size_t ray = m / (size_t)a.S;
                const float ts = a.t_per_ray ? a.t[m] : a.t[m - ray * (size_t)a.S];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    d[c] = a.rays_d[ray * 3 + c];
                    x[c] = norm_coord_x3(add_rn(a.rays_o[ray * 3 + c], mul_rn(ts, d[c])), a.nearf, a.spanf);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { xin[tid * 8 + c] = x[c]; xin[tid * 8 + 3 + c] = d[c]; }
        float big = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fabsf(x[2]));
        float bigd = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fabsf(d[2]));
#pragma unroll
        for (int off = TR / 2; off >= 1; off >>= 1) {
            big = fmaxf(big, __shfl_xor(big, off, TR));
            bigd = fmaxf(bigd, __shfl_xor(bigd, off, TR));
        }
        if (tid == 0) { xin[6] = big; xin[7] = bigd; }
    }
}

// The encoding of one sample row: values into the planes (three terms each) and, when `stage` is given, as fp32 into a
// staging row (for the workspace copy).  Table form (EncCache) or, for tiles whose arguments leave its range, the
// full-range sine/cosine -- value for value what the fp32 kernels compute.
template <int L>
__device__ __forceinline__ void x3_enc_put(unsigned char* planes, float* stage, int row, int col, float v) {
    store_value(planes, row, col, v);
    if (stage) stage[col] = v;
}
template <int L>
__device__ __forceinline__ void x3_enc_store(unsigned char* planes, float* stage, int row, int q, const EncCache<L>& c) {
    constexpr int R = EncCache<L>::ROUNDS;
    const bool extra = q == 3;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (enc_round_active<L>(q, i)) {
            const int base = extra ? 48 + i : 12 * i + q;
            x3_enc_put<L>(planes, stage, row, base + 3, c.sn[i][0]);
            x3_enc_put<L>(planes, stage, row, base + 6, c.cs[i][0]);
            x3_enc_put<L>(planes, stage, row, base + 9, c.sn[i][1]);
            x3_enc_put<L>(planes, stage, row, base + 12, c.cs[i][1]);
        }
    }
    if (!extra) x3_enc_put<L>(planes, stage, row, q, c.x);
    else {
#pragma unroll
        for (int col = 3 + 6 * L; col < ((3 + 6 * L + 15) & ~15); ++col) x3_enc_put<L>(planes, stage, row, col, 0.0f);
    }
}
template <int L>
__device__ __forceinline__ void x3_enc_slow(unsigned char* planes, float* stage, int row, int q, const float* xs) {
    if (q < 3) {
        const float x = xs[q];
        x3_enc_put<L>(planes, stage, row, q, x);
        float scale = 1.0f;
#pragma unroll 1
        for (int j = 0; j < L; ++j) {
            float s, c;
            nerf_sincosf(mul_rn(scale, x), &s, &c);
            x3_enc_put<L>(planes, stage, row, 3 + 6 * j + q, s);
            x3_enc_put<L>(planes, stage, row, 3 + 6 * j + 3 + q, c);
            scale *= 2.0f;
        }
    } else {
#pragma unroll
        for (int col = 3 + 6 * L; col < ((3 + 6 * L + 15) & ~15); ++col) x3_enc_put<L>(planes, stage, row, col, 0.0f);
    }
}

// copy a [64][WIDTH] fp32 staging tile (row pitch LDPB bytes) to global rows of WIDTH floats
template <int WIDTH>
__device__ __forceinline__ void x3_copy_stage(const unsigned char* stage0, float* __restrict__ g, int tid) {
    constexpr int C4 = WIDTH / 4;
#pragma unroll
    for (int i = tid; i < TR * C4; i += X3_THREADS) {
        const int row = i / C4, c4 = i % C4;
        *reinterpret_cast<f32x4*>(g + (size_t)row * WIDTH + 4 * c4) = *reinterpret_cast<const f32x4*>(stage0 + row * LDPB + 16 * c4);
    }
}

// ---- data gradient ----------------------------------------------------------------------------------------
struct X3BwdArgs {
    const float* pk;
    const __bf16* x3;
    int64_t M;
    size_t Mp;
    const float* dcolors;
    const float* ddensity;
    const float* colors;
    const float* ws;
    float* bw;
};


// =====================================================================================================================
// Siren (nerf/models.py:185-203) in the same mode: sine layers instead of ReLU layers.  The derivative w0 cos(w0 z) of
// a sine layer has no bits to hide in: like the fp32 kernels, the forward stores it per layer as a tile in the
// accumulator order of THESE kernels (piece ni*4 + g/4 of a lane = its registers 4 (g/4) .. + 3 of tile ni, 16 bytes, at
// float offset (piece * 512 + tid) * 4 of the layer's [64][256] region), which the data-gradient kernel reads back as
// eight coalesced 16-byte loads per lane.
// =====================================================================================================================
namespace x3s {
using namespace lay;
constexpr size_t elems(int Kp, int N) { return (size_t)3 * Kp * N; }
constexpr size_t F_L1 = 0;                                   // L1..L7
constexpr size_t F_FEAT = F_L1 + 7 * elems(H, H);
constexpr size_t F_RGBA = F_FEAT + elems(H, H);
constexpr size_t F_RGBB = F_RGBA + elems(H, HH);
constexpr size_t D_RGB = F_RGBB + elems(DE_P, HH);
constexpr size_t D_FEAT = D_RGB + elems(HH, H);
constexpr size_t D_L7 = D_FEAT + elems(H, H);
constexpr size_t TOTAL = D_L7 + 7 * elems(H, H);
constexpr size_t d_img(int layer) { return D_L7 + (size_t)(7 - layer) * elems(H, H); }   // layer 1..7
static_assert(TOTAL % 8 == 0 && lay::siren::PACKED_FLOATS % 4 == 0, "16-byte pieces");
constexpr size_t TOTAL_FLOATS = TOTAL / 2;
}  // namespace x3s

__device__ __forceinline__ float* x3_cos_piece(float* region, int piece, int tid) { return region + ((size_t)piece * X3_THREADS + tid) * 4; }

// Epilogue of a hidden sine layer: h = sin(z) (table form: the arguments of sine layers fed by sines are bounded) as
// fp32 to the workspace and as three terms to the planes, cos(z) to the layer's derivative tile.
__device__ __forceinline__ void x3_sine_epilogue(unsigned char* planes, f32x16 (&acc)[2], int ncol0, int wrow0, int lane, int tid,
                                                 float* __restrict__ gout, float* __restrict__ gcos) {
    const int r = lane & 31, h = lane >> 5;
    const int col = ncol0 + 2 * r;
    const __amdgpu_buffer_rsrc_t g_rsrc = make_rsrc(gout, (unsigned)(TR * lay::H * 4));
    const int g_voff = ((wrow0 + 4 * h) * lay::H + col) * 4;
    const __amdgpu_buffer_rsrc_t c_rsrc = make_rsrc(gcos, (unsigned)(TR * lay::H * 4));
    unsigned char* pl = planes + (wrow0 + 4 * h) * LDPB + col * 2;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        SinCosPair pr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sincos_tab_issue((f32x2){acc[0][4 * q4 + j], acc[1][4 * q4 + j]}, pr[j]);
        f32x4 c0, c1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row0 = j + 8 * q4;      // (g & 3) + 8 (g >> 2), + wrow0 + 4 h
            f32x2 sn, cs;
            sincos_tab_finish(pr[j], sn, cs);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, sn), g_rsrc, g_voff, row0 * lay::H * 4, 2);
            unsigned w[3];
            split2(sn[0], sn[1], w);
            store_terms(pl, row0 * LDPB, w);
            c0[j] = cs[0];
            c1[j] = cs[1];
        }
        // 16-byte stores with the piece offset in the vector offset and a constant scalar offset (mlp_device.h, sine_epilogue,
        // explains why wide stores keep their offset out of a scalar register)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, c0), c_rsrc, (q4 * X3_THREADS + tid) * 16, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, c1), c_rsrc, ((4 + q4) * X3_THREADS + tid) * 16, 0, 2);
    }
}

// Backward epilogue of a sine layer: dz = (acc [+ dsig[row] * sigw[col]]) * stored derivative
template <bool RANK1, bool TO_PLANES>
__device__ __forceinline__ void x3_sine_bwd_epilogue(unsigned char* planes, f32x16 (&acc)[2], int ncol0, int wrow0, int lane,
                                                     float* __restrict__ gout, const f32x4 (&dv)[8], const float* dsig_lds,
                                                     const float* __restrict__ sigw) {
    const int r = lane & 31, h = lane >> 5;
    const int col = ncol0 + 2 * r;
    const __amdgpu_buffer_rsrc_t g_rsrc = make_rsrc(gout, (unsigned)(TR * lay::H * 4));
    const int g_voff = ((wrow0 + 4 * h) * lay::H + col) * 4;
    unsigned char* pl = planes + (wrow0 + 4 * h) * LDPB + col * 2;
    f32x2 sw = {0.f, 0.f};
    if constexpr (RANK1) sw = *reinterpret_cast<const f32x2*>(sigw + col);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int row0 = (g & 3) + 8 * (g >> 2);
        float v[2] = {acc[0][g], acc[1][g]};
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            if constexpr (RANK1) v[ni] = fmaf(dsig_lds[wrow0 + row0 + 4 * h], sw[ni], v[ni]);
            v[ni] *= dv[ni * 4 + (g >> 2)][g & 3];
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, (f32x2){v[0], v[1]}), g_rsrc, g_voff, row0 * lay::H * 4, 2);
        if constexpr (TO_PLANES) {
            unsigned w[3];
            split2(v[0], v[1], w);
            store_terms(pl, row0 * LDPB, w);
        }
    }
}

__global__ __launch_bounds__(X3_THREADS, 1) void siren_fwd_x3_kernel(X3FwdArgs a) {
    using namespace lay;
    using namespace lay::siren;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
    float* tab = reinterpret_cast<float*>(smem_x3);
    unsigned char* planes = smem_x3 + SINCOS_TAB_FLOATS * 4;
    float* xin = reinterpret_cast<float*>(planes + PLANES_BYTES);
    float* sraw_s = xin + X3_XIN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow0 = (wave >> 2) * 32, wcol = wave & 3;
    const bool per_sample = tid < 256;
    const size_t m0 = (size_t)blockIdx.x * TR;
    const float* __restrict__ pk = a.pk;
    const __bf16* __restrict__ xi = a.x3;
    float* ws = a.ws;
    const size_t Mp = a.Mp;
    auto wsp = [&](size_t off, int width) -> float* { return ws + off * Mp + m0 * width; };
    const int smp = (tid & 255) / TPS, q = tid % TPS;
    constexpr size_t EHH = x3s::elems(H, H);

    if (per_sample) load_sincos_table(tab, tid);
    x3_load_inputs(a, m0, xin, tid);
    const int ncol0 = wcol * 64;
    X3Pre<2> pre;
    x3_prefetch<2, H / 16, ACC_BIAS>(xi + x3s::F_L1, H, ncol0, lane, pk + BIAS + H, pre);
    __syncthreads();
    if (tid < TR) {
        f32x4 v = {xin[tid * 8], xin[tid * 8 + 1], xin[tid * 8 + 2], 0.0f};
        *reinterpret_cast<f32x4*>(ws + WS_X * Mp + (m0 + tid) * 4) = v;
    }
    // layer 0 (3 -> 256, sin(30 z)) on the vector ALU: thread = (column pair, half of the samples), four samples (one value
    // pair each: the thread's two columns) in flight at a time.  Table form for |30 z| <= 1e5, decided per thread
    // from the tile's largest |coordinate| (xin[6]); the library's full-range path otherwise.
    if (per_sample) {
        const int c = 2 * (tid & 127), s_lo = (tid >> 7) * 32;
        const float wa0 = pk[W0 + c * 3], wa1 = pk[W0 + c * 3 + 1], wa2 = pk[W0 + c * 3 + 2], ba = pk[BIAS + c];
        const float wb0 = pk[W0 + c * 3 + 3], wb1 = pk[W0 + c * 3 + 4], wb2 = pk[W0 + c * 3 + 5], bb = pk[BIAS + c + 1];
        const float bound = 30.0f * (fmaxf(fabsf(wa0) + fabsf(wa1) + fabsf(wa2), fabsf(wb0) + fabsf(wb1) + fabsf(wb2)) * xin[6]
                                     + fmaxf(fabsf(ba), fabsf(bb)));
        const bool fast = bound <= 9.9e4f;
        float* h0 = wsp(WS_H, H);
        float* c0 = wsp(WS_C, H);
        // owner of the derivative piece of rows s .. s+3 (s a multiple of 4), columns c / c + 1, in the accumulator order
        const int owner_lane = (c & 63) >> 1, owner_wcol = c >> 6;
#pragma unroll 1
        for (int s0 = s_lo; s0 < s_lo + 32; s0 += 4) {
            f32x2 sn[4], cs[4];
            if (fast) {
                SinCosPair pr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* x = xin + (s0 + j) * 8;
                    const float za = mul_rn(30.0f, fmaf(x[2], wa2, fmaf(x[1], wa1, fmaf(x[0], wa0, ba))));
                    const float zb = mul_rn(30.0f, fmaf(x[2], wb2, fmaf(x[1], wb1, fmaf(x[0], wb0, bb))));
                    sincos_tab_issue((f32x2){za, zb}, pr[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) sincos_tab_finish(pr[j], sn[j], cs[j]);
            } else {
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    const float* x = xin + (s0 + j) * 8;
                    const float za = mul_rn(30.0f, fmaf(x[2], wa2, fmaf(x[1], wa1, fmaf(x[0], wa0, ba))));
                    const float zb = mul_rn(30.0f, fmaf(x[2], wb2, fmaf(x[1], wb1, fmaf(x[0], wb0, bb))));
                    float s_, c_;
                    sincosf(za, &s_, &c_);
                    sn[j][0] = s_; cs[j][0] = c_;
                    sincosf(zb, &s_, &c_);
                    sn[j][1] = s_; cs[j][1] = c_;
                }
            }
            f32x4 d0, d1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = s0 + j;
                *reinterpret_cast<f32x2*>(h0 + (size_t)s * H + c) = sn[j];
                unsigned w[3];
                split2(sn[j][0], sn[j][1], w);
                store_terms(planes, s * LDPB + c * 2, w);
                d0[j] = 30.0f * cs[j][0];
                d1[j] = 30.0f * cs[j][1];
            }
            const int owner = ((s0 >> 5) * 4 + owner_wcol) * 64 + 32 * ((s0 >> 2) & 1) + owner_lane;
            const int q4 = (s0 & 31) >> 3;
            *reinterpret_cast<f32x4*>(x3_cos_piece(c0, q4, owner)) = d0;
            *reinterpret_cast<f32x4*>(x3_cos_piece(c0, 4 + q4, owner)) = d1;
        }
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll 1
    for (int l = 1; l <= 7; ++l) {
        mma_layer_x3<2, H / 16, ACC_BIAS>(planes, xi + x3s::F_L1 + (size_t)(l - 1) * EHH, H, ncol0, wrow0, lane, acc, pre);
        x3_prefetch<2, H / 16, ACC_BIAS>(xi + x3s::F_L1 + (size_t)l * EHH, H, ncol0, lane, pk + BIAS + (l + 1) * H, pre);   // feature_remap after L7
        __syncthreads();
        x3_sine_epilogue(planes, acc, ncol0, wrow0, lane, tid, wsp(WS_H + (size_t)l * H, H), wsp(WS_C + (size_t)l * H, H));
        __syncthreads();
    }
    // feature_remap on MFMA; the density branch is a dot product of h7, taken from its fp32 copy in the workspace
    mma_layer_x3<2, H / 16, ACC_BIAS>(planes, xi + x3s::F_FEAT, H, ncol0, wrow0, lane, acc, pre);
    const int nc = wcol * 32;
    X3Pre<1> pre1;
    x3_prefetch<1, H / 16, ACC_BIAS>(xi + x3s::F_RGBA, HH, nc, lane, pk + BIAS_RGB, pre1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (per_sample) {
        const float* h7 = wsp(WS_H + 7 * (size_t)H, H) + (size_t)smp * H;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = 4 * (TPS * i + q);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(h7 + k);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(pk + SIGW + k);
            s += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += pk[BIAS_SIG];
        if (q == 0) {
            sraw_s[smp] = s;
            ws[WS_SRAW * Mp + m0 + smp] = s;
        }
    }
    x3_epilogue<0, true>(planes, acc, ncol0, wrow0, lane, wsp(WS_F, H));
    __syncthreads();
    // rgb sine layer: [f | de] -> 128
    {
        f32x16 acc1[1];
        mma_layer_x3<1, H / 16, ACC_BIAS>(planes, xi + x3s::F_RGBA, HH, nc, wrow0, lane, acc1, pre1);
        x3_prefetch<1, DE_P / 16, ACC_KEEP>(xi + x3s::F_RGBB, HH, nc, lane, nullptr, pre1);
        __syncthreads();
        unsigned char* stage_de = planes + 128;
        float* st = reinterpret_cast<float*>(stage_de + smp * LDPB);
        const auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
        if (per_sample) {
            if (uniform(xin[7]) * 8.0f <= 9.9e4f) {
                EncCache<4> dec;
                enc_compute<4>(xin + smp * 8 + 3, q, dec);
                x3_enc_store<4>(planes, st, smp, q, dec);
            } else {
                x3_enc_slow<4>(planes, st, smp, q, xin + smp * 8 + 3);
            }
        }
        __syncthreads();
        x3_copy_stage<DE_P>(stage_de, wsp(WS_DE, DE_P), tid);
        mma_layer_x3<1, DE_P / 16, ACC_KEEP>(planes, xi + x3s::F_RGBB, HH, nc, wrow0, lane, acc1, pre1);
        __syncthreads();
        float* gt = reinterpret_cast<float*>(planes);
        float* gg = wsp(WS_G, HH);
        float* gc = wsp(WS_CG, HH);
        const int col = nc + (lane & 31);
#pragma unroll
        for (int g8 = 0; g8 < 16; g8 += 8) {
            SinCosPair pr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sincos_tab_issue((f32x2){acc1[0][g8 + 2 * j], acc1[0][g8 + 2 * j + 1]}, pr[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 sn, cs;
                sincos_tab_finish(pr[j], sn, cs);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int g = g8 + 2 * j + e;
                    const int row = wrow0 + (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
                    gt[row * LDG + col] = sn[e];
                    gg[(size_t)row * HH + col] = sn[e];
                    gc[(size_t)row * HH + col] = cs[e];
                }
            }
        }
        __syncthreads();
        if (per_sample) {
            float z[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 4 * (TPS * i + q);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gt + smp * LDG + k);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(pk + WOUT + c * HH + k);
                    z[c] += gv[0] * wv[0] + gv[1] * wv[1] + gv[2] * wv[2] + gv[3] * wv[3];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                z[c] += __shfl_xor(z[c], 1);
                z[c] += __shfl_xor(z[c], 2);
            }
            const size_t m = m0 + smp;
            if (q == 0 && m < (size_t)a.M) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.colors[m * 3 + c] = sigmoidf_(z[c] + pk[BIAS_OUT + c]);
                a.density[m] = fmaxf(sraw_s[smp], 0.0f) * 10.0f;
            }
        }
    }
}

__global__ __launch_bounds__(X3_THREADS, 1) void siren_dgrad_x3_kernel(X3BwdArgs a) {
    using namespace lay;
    using namespace lay::siren;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
    unsigned char* planes = smem_x3;
    float* dsig_s = reinterpret_cast<float*>(planes + PLANES_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow0 = (wave >> 2) * 32, wcol = wave & 3;
    const size_t m0 = (size_t)blockIdx.x * TR;
    const size_t Mp = a.Mp;
    const float* __restrict__ pk = a.pk;
    const __bf16* __restrict__ xi = a.x3;
    const float* ws = a.ws;
    float* bw = a.bw;
    const int ncol0 = wcol * 64;
    auto dzp = [&](int l) { return bw + (BW_DZ + (size_t)l * H) * Mp + m0 * H; };
    // this lane's eight derivative pieces of sine layer l
    auto load_deriv = [&](int l, f32x4 (&dv)[8]) {
        const f32x4* region = reinterpret_cast<const f32x4*>(ws + (WS_C + (size_t)l * H) * Mp + m0 * H);
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) dv[pc] = region[pc * X3_THREADS + tid];
    };
    X3Pre<2> pre;
    x3_prefetch<2, HH / 16, ACC_ZERO>(xi + x3s::D_RGB, H, ncol0, lane, nullptr, pre);
    // gradient of the 128 -> 3 sigmoid head through the rgb sine layer's stored cosine: dzr [64][128] into the planes
    if (tid < 256) {
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
            ds = ws[WS_SRAW * Mp + m] > 0.0f ? a.ddensity[m] * 10.0f : 0.0f;
        }
        if (q == 0) {
            f32x4 v = {dz[0], dz[1], dz[2], 0.0f};
            *reinterpret_cast<f32x4*>(bw + BW_DZO * Mp + m * 4) = v;
            bw[BW_DSIG * Mp + m] = ds;
            dsig_s[smp] = ds;
        }
        const float* wout = pk + WOUT;
        const float* gsrc = ws + WS_CG * Mp;
        float* bw_dzr = bw + BW_DZR * Mp;
#pragma unroll
        for (int i = 0; i < 32 / TPS; ++i) {
            const int k = 4 * (TPS * i + q);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wout + k);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wout + HH + k);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wout + 2 * HH + k);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gsrc + m * HH + k);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (dz[0] * w0[e] + dz[1] * w1[e] + dz[2] * w2[e]) * gv[e];
            *reinterpret_cast<f32x4*>(bw_dzr + m * HH + k) = v;
            unsigned wa[3], wb[3];
            split2(v[0], v[1], wa);
            split2(v[2], v[3], wb);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<u32x2_t*>(planes + p * PLANE_BYTES + smp * LDPB + k * 2) = u32x2_t{wa[p], wb[p]};
        }
    }
    __syncthreads();
    f32x16 acc[2];
    f32x4 dv[8];
    // rgb sine layer (feature part) -> gradient of feature_remap's output (dzf)
    mma_layer_x3<2, HH / 16, ACC_ZERO>(planes, xi + x3s::D_RGB, H, ncol0, wrow0, lane, acc, pre);
    x3_prefetch<2, H / 16, ACC_ZERO>(xi + x3s::D_FEAT, H, ncol0, lane, nullptr, pre);
    __syncthreads();
    x3_epilogue<4, true>(planes, acc, ncol0, wrow0, lane, bw + BW_DZF * Mp + m0 * H);
    __syncthreads();
    // feature_remap + density branch (rank-1) -> dh7, times cos -> dz7
    load_deriv(7, dv);
    mma_layer_x3<2, H / 16, ACC_ZERO>(planes, xi + x3s::D_FEAT, H, ncol0, wrow0, lane, acc, pre);
    x3_prefetch<2, H / 16, ACC_ZERO>(xi + x3s::d_img(7), H, ncol0, lane, nullptr, pre);
    __syncthreads();
    x3_sine_bwd_epilogue<true, true>(planes, acc, ncol0, wrow0, lane, dzp(7), dv, dsig_s, pk + SIGW);
    __syncthreads();
#pragma unroll 1
    for (int l = 7; l >= 2; --l) {
        load_deriv(l - 1, dv);
        mma_layer_x3<2, H / 16, ACC_ZERO>(planes, xi + x3s::d_img(l), H, ncol0, wrow0, lane, acc, pre);
        x3_prefetch<2, H / 16, ACC_ZERO>(xi + x3s::d_img(l - 1), H, ncol0, lane, nullptr, pre);
        __syncthreads();
        x3_sine_bwd_epilogue<false, true>(planes, acc, ncol0, wrow0, lane, dzp(l - 1), dv, nullptr, nullptr);
        __syncthreads();
    }
    load_deriv(0, dv);
    mma_layer_x3<2, H / 16, ACC_ZERO>(planes, xi + x3s::d_img(1), H, ncol0, wrow0, lane, acc, pre);
    x3_sine_bwd_epilogue<false, false>(planes, acc, ncol0, wrow0, lane, dzp(0), dv, nullptr, nullptr);
}

}  // namespace

extern "C" size_t nerf_mlp_packed_x3_floats(int model) {
    return model == NERF_MODEL_NERF ? lay::nerf::PACKED_FLOATS + (nerf_mlp_packed_bf16x3_bytes(model) + nerf_x3i_dgrad_image_bytes()) / 4
         : model == NERF_MODEL_SIREN ? lay::siren::PACKED_FLOATS + x3s::TOTAL_FLOATS : 0;
}

extern "C" int nerf_mlp_pack_x3(int model, const float* const* params, float* packed, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_pack_x3: unknown model %d", model);
    NERF_REQUIRE_PTR(params);
    NERF_REQUIRE_PTR(packed);
    int rc = nerf_mlp_pack(model, params, packed, stream);   // the fp32 image comes first (thin vectors, biases, inference)
    if (rc) return rc;
    using namespace lay;
    if (model == NERF_MODEL_NERF) {
        // both chains run on the ring kernels of mlp_x3i.hip: the forward's image, then the data-gradient chain's
        float* fwd_img = packed + lay::nerf::PACKED_FLOATS;
        if ((rc = nerf_mlp_pack_bf16x3(model, params, fwd_img, stream))) return rc;
        return nerf_x3i_dgrad_pack(params, fwd_img + nerf_mlp_packed_bf16x3_bytes(model) / 4, stream);
    }
    X3PackArgs a;
    for (int i = 0; i < NERF_NUM_PARAM_TENSORS; ++i) a.src[i] = params[i];
    a.njobs = 0;
    a.dst = reinterpret_cast<__bf16*>(packed + lay::siren::PACKED_FLOATS);
    for (int l = 1; l <= 7; ++l) x3_fwd_job(a, 2 * l, H, 0, H, H, x3s::F_L1 + (size_t)(l - 1) * x3s::elems(H, H), false, true);
    x3_fwd_job(a, 18, H, 0, H, H, x3s::F_FEAT, false, true);
    x3_fwd_job(a, 20, H + DE, 0, HH, H, x3s::F_RGBA, false, false);
    x3_fwd_job(a, 20, H + DE, H, HH, DE, x3s::F_RGBB, false, false);
    x3_bwd_job(a, 20, H + DE, H, HH, x3s::D_RGB);
    x3_bwd_job(a, 18, H, H, H, x3s::D_FEAT);
    for (int l = 7; l >= 1; --l) x3_bwd_job(a, 2 * l, H, H, H, x3s::d_img(l));
    pack_x3_kernel<<<dim3(a.njobs, 64), 256, 0, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_pack_x3");
}

extern "C" int nerf_mlp_fwd_x3(int model, const float* packed, int64_t M, const float* points, const float* dirs,
                               const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near,
                               double far, float* colors, float* density, float* workspace, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_fwd_x3: unknown model %d", model);
    NERF_REQUIRE(M >= 0, "nerf_mlp_fwd_x3: M=%lld < 0", (long long)M);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE_PTR(packed);
    NERF_REQUIRE_PTR(colors);
    NERF_REQUIRE_PTR(density);
    NERF_REQUIRE_PTR(workspace);
    if (points) {
        NERF_REQUIRE_PTR(dirs);
    } else {
        NERF_REQUIRE_PTR(rays_o);
        NERF_REQUIRE_PTR(rays_d);
        NERF_REQUIRE_PTR(t);
        NERF_REQUIRE(S >= 1 && M % S == 0, "nerf_mlp_fwd_x3: ray mode needs M (%lld) to be a multiple of S (%d)", (long long)M, S);
    }
    const size_t mp = lay::padded_samples(M);
    const size_t tiles = mp / TR;
    NERF_REQUIRE(tiles < (1ull << 31), "nerf_mlp_fwd_x3: too many samples");
    const size_t fp32_floats = model == NERF_MODEL_NERF ? lay::nerf::PACKED_FLOATS : lay::siren::PACKED_FLOATS;
    X3FwdArgs a{packed, reinterpret_cast<const __bf16*>(packed + fp32_floats), M, mp, points, dirs, rays_o, rays_d,
                t, S, t_per_ray, (float)near, (float)(far - near), colors, density, workspace};
    if (model == NERF_MODEL_NERF)   // 16-sample tiles, weights through an LDS ring (mlp_x3i.hip)
        return nerf_x3i_fwd_train(packed + lay::nerf::PACKED_FLOATS, M, points, dirs, rays_o, rays_d, t, S,
                                  t_per_ray, near, far, colors, density, workspace, stream);
    if (int rc = nerf_set_lds(siren_fwd_x3_kernel, X3_FWD_LDS_BYTES)) return rc;
    siren_fwd_x3_kernel<<<(unsigned)tiles, X3_THREADS, X3_FWD_LDS_BYTES, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_fwd_x3");
}

extern "C" int nerf_mlp_bwd_data_x3(int model, const float* packed, int64_t M, const float* dcolors, const float* ddensity,
                                    const float* colors, const float* fwd_workspace, float* bwd_workspace, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_bwd_data_x3: unknown model %d", model);
    NERF_REQUIRE(M >= 1, "nerf_mlp_bwd_data_x3: M=%lld < 1", (long long)M);
    NERF_REQUIRE_PTR(packed);
    NERF_REQUIRE_PTR(dcolors);
    NERF_REQUIRE_PTR(ddensity);
    NERF_REQUIRE_PTR(colors);
    NERF_REQUIRE_PTR(fwd_workspace);
    NERF_REQUIRE_PTR(bwd_workspace);
    const size_t Mp = lay::padded_samples(M);
    NERF_REQUIRE(Mp / TR < (1ull << 31), "nerf_mlp_bwd_data_x3: too many samples");
    const size_t fp32_floats = model == NERF_MODEL_NERF ? lay::nerf::PACKED_FLOATS : lay::siren::PACKED_FLOATS;
    X3BwdArgs a{packed, reinterpret_cast<const __bf16*>(packed + fp32_floats), M, Mp, dcolors, ddensity, colors,
                fwd_workspace, bwd_workspace};
    if (model == NERF_MODEL_NERF)
        return nerf_x3i_dgrad(packed + fp32_floats + nerf_mlp_packed_bf16x3_bytes(model) / 4, packed + lay::nerf::W11, M, dcolors,
                              ddensity, colors, fwd_workspace, bwd_workspace, stream);
    if (int rc = nerf_set_lds(siren_dgrad_x3_kernel, X3_BWD_LDS_BYTES)) return rc;
    siren_dgrad_x3_kernel<<<(unsigned)(Mp / TR), X3_THREADS, X3_BWD_LDS_BYTES, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_bwd_data_x3");
}
