// Opt-in inference forwards on the bf16 matrix pipe.  One-term mode, NeRF MLP (nerf/models.py:62-75): bf16 operands, fp32
// accumulation on v_mfma_f32_32x32x16_bf16.  The fp32 kernels in mlp_fwd.hip stay the default and
// the parity path; this one trades accuracy (bf16 rounding of every weight and activation,
// tolerance stated by its tests) for matrix-core throughput.
//
// Orientation.  A layer is computed transposed, D[out feature][sample] = sum_k W[out][k] X[k][sample]:
// the weights are the MFMA A operand, the activations the B operand.  A 32x32 accumulator tile then
// holds 32 features (registers x lane half) of 32 samples (lane & 31), which is exactly the
// B-operand layout of the next layer's MFMAs: bias and ReLU are applied in registers, pairs of
// registers are rounded to bf16 and the next layer starts -- activations never leave the VGPRs,
// there is no LDS round trip and no barrier on their account.
//   accumulator register g of lane (r, h):   feature (g&3) + 8*(g>>2) + 4*h of the tile, sample r
//   B fragment of k-step s (s = 0, 1 per 32-feature tile): registers 8s..8s+7, i.e. element j is
//   feature 16*s + 8*(j>>2) + 4*h + (j&3); the weight image stores each A fragment in that k order.
//
// A wave owns 32 samples, a workgroup of 8 waves 256 samples.  All waves need the same weights in
// the same order, so the packed image is streamed once per workgroup through a ring of four 32 KiB
// LDS buffers with buffer_load ... lds (no staging registers); one barrier per 32 KiB chunk both
// publishes chunk c and releases the buffer of chunk c-1 for chunk c+3.  The image is stored in
// consumption order, one 1 KiB unit per MFMA: unit[lane][8 bf16] is read back with one
// conflict-free ds_read_b128 per lane.
//
// Three-term variant (template parameter P = 3, "bf16x3").  Every fp32 operand is written as the sum of three bf16 terms,
// x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (the subtractions are exact), and a
// product of two operands is formed from the six term products that reach fp32 precision:
//     w*x ~= w0*x0 + w0*x1 + w0*x2 + w1*x0 + w1*x1 + w2*x0      (what is dropped is below 2^-24 of |w*x|)
// -- six bf16 MFMAs (32 cycles each for 16 k) in place of eight fp32 ones (64 cycles each for 2 k).  Every term product
// is exact in the fp32 accumulator, so the result carries the rounding of an fp32 accumulation and nothing else: this
// mode holds the fp32 parity bar, unlike the one-term mode.  The weight image stores the three terms of an A fragment
// as three consecutive units; a wave keeps three B fragments per k-step (192 registers of activations), which is why a
// workgroup is 4 waves of 512 registers here instead of 8 waves of 256; encodings use the fp32 path's sincos.
// This variant also covers the Siren model (nerf/models.py:185-203): its sine activations are applied to the fp32
// accumulators before the split, which the one-term mode could not do usefully (sin(30 z) of bf16-rounded inputs).
#include "mlp_device.h"

#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int UNIT_BYTES = 1024;     // one A fragment for a whole wave
constexpr int CHUNK_UNITS = 32;
constexpr int CHUNK_BYTES = CHUNK_UNITS * UNIT_BYTES;
constexpr int NBUF = 4;

// ---- layer table -------------------------------------------------------------------------
enum Enc { ENC_NONE = 0, ENC_PE = 1, ENC_DE = 2, ENC_XYZ = 3 };
struct LayerInfo {
    int n_ot;      // 32-row output tiles
    int nks_acc;   // k-steps fed by the previous layer's accumulators (16 features each)
    int enc;       // extra k-steps fed by an encoding: ENC_PE 4, ENC_DE 2, ENC_XYZ 1 (the raw coordinates)
    int n_out;     // valid output rows
    int n_in;      // valid accumulator-fed inputs (also the column where the encoding part starts)
    int ld;        // row pitch of the [out, in] weight tensor
    int wtensor;   // state_dict index of the weight; the bias follows it
    int last_row_tensor;   // >= 0: output row n_out-1 is row 0 of this [1, ld] weight (its bias follows it)
};
constexpr int NL_MAX = 12;
template <int MODEL> struct Net;
// nerf/models.py:62-75.  block2.8 (257 rows: 256 features + the density row) is one layer of nine tiles.
template <> struct Net<NERF_MODEL_NERF> {
    static constexpr int NL = 12;
    static constexpr LayerInfo LAYERS[NL_MAX] = {
        {8, 0, ENC_PE, 256, 0, 63, 0, -1},      {8, 16, ENC_NONE, 256, 256, 256, 2, -1},  {8, 16, ENC_NONE, 256, 256, 256, 4, -1},
        {8, 16, ENC_NONE, 256, 256, 256, 6, -1}, {8, 16, ENC_NONE, 256, 256, 256, 8, -1},  {8, 16, ENC_PE, 256, 256, 319, 10, -1},
        {8, 16, ENC_NONE, 256, 256, 256, 12, -1}, {8, 16, ENC_NONE, 256, 256, 256, 14, -1}, {8, 16, ENC_NONE, 256, 256, 256, 16, -1},
        {9, 16, ENC_NONE, 257, 256, 256, 18, -1}, {4, 16, ENC_DE, 128, 256, 283, 20, -1},   {1, 8, ENC_NONE, 3, 128, 128, 22, -1},
    };
};
// nerf/models.py:185-203.  feature_remap (256 rows) and density_branch (1 row) both read the last sine layer: one
// layer of nine tiles whose 257th row comes from the density branch's tensors.
template <> struct Net<NERF_MODEL_SIREN> {
    static constexpr int NL = 11;
    static constexpr LayerInfo LAYERS[NL_MAX] = {
        {8, 0, ENC_XYZ, 256, 0, 3, 0, -1},       {8, 16, ENC_NONE, 256, 256, 256, 2, -1},  {8, 16, ENC_NONE, 256, 256, 256, 4, -1},
        {8, 16, ENC_NONE, 256, 256, 256, 6, -1}, {8, 16, ENC_NONE, 256, 256, 256, 8, -1},  {8, 16, ENC_NONE, 256, 256, 256, 10, -1},
        {8, 16, ENC_NONE, 256, 256, 256, 12, -1}, {8, 16, ENC_NONE, 256, 256, 256, 14, -1}, {9, 16, ENC_NONE, 257, 256, 256, 18, 16},
        {4, 16, ENC_DE, 128, 256, 283, 20, -1},   {1, 8, ENC_NONE, 3, 128, 128, 22, -1},    {0, 0, ENC_NONE, 0, 0, 0, 0, -1},
    };
};
constexpr int enc_ks(int enc) { return enc == ENC_PE ? 4 : enc == ENC_DE ? 2 : enc == ENC_XYZ ? 1 : 0; }
constexpr int NBIAS_TILES_MAX = 86;                        // LDS room for the bias image of either model

// Geometry of the P-term kernel of a model: P units (one per term) for every A fragment, every layer starts a new chunk.
constexpr int wg_threads(int P) { return P == 1 ? 512 : 256; }
constexpr int per_chunk(int P) { return CHUNK_BYTES / (16 * wg_threads(P)); }   // LDS-DMA instructions per thread and chunk
template <int P, int MODEL>
struct Geo {
    using N = Net<MODEL>;
    static constexpr int NL = N::NL;
    static constexpr int WG_THREADS = wg_threads(P);
    static constexpr int WG_SAMPLES = WG_THREADS / 2;                        // a wave owns 32 samples
    static constexpr int PER_CHUNK = per_chunk(P);
    static constexpr int layer_ks(int l) { return N::LAYERS[l].nks_acc + enc_ks(N::LAYERS[l].enc); }
    static constexpr int layer_units(int l) { return P * N::LAYERS[l].n_ot * layer_ks(l); }
    static constexpr int layer_chunks(int l) { return (layer_units(l) + CHUNK_UNITS - 1) / CHUNK_UNITS; }
    static constexpr int chunk0(int l) {
        int c = 0;
        for (int i = 0; i < l; ++i) c += layer_chunks(i);
        return c;
    }
    static constexpr int bias_tile0(int l) {
        int t = 0;
        for (int i = 0; i < l; ++i) t += N::LAYERS[i].n_ot;
        return t;
    }
    static constexpr int NCHUNK = chunk0(NL);
    static constexpr int NBIAS_TILES = bias_tile0(NL);
    static constexpr size_t BIAS_FLOATS = (size_t)NBIAS_TILES * 32;
    static constexpr size_t IMG_BYTES = (size_t)NCHUNK * CHUNK_BYTES;
    static constexpr size_t PACKED_BYTES = IMG_BYTES + BIAS_FLOATS * 4;
};
static_assert(Geo<1, NERF_MODEL_NERF>::NCHUNK == 43 && Geo<3, NERF_MODEL_NERF>::NCHUNK == 124 &&
                  Geo<1, NERF_MODEL_NERF>::NBIAS_TILES == 86 && Geo<3, NERF_MODEL_SIREN>::NBIAS_TILES == 78 &&
                  Geo<3, NERF_MODEL_SIREN>::NBIAS_TILES <= NBIAS_TILES_MAX,
              "bf16 image layout");

// term `plane` of the three-term bf16 expansion of x (term 0 alone is the plain bf16 rounding)
__host__ __device__ inline __bf16 bf16_term(float x, int plane) {
    __bf16 t = (__bf16)x;
    for (int i = 0; i < plane; ++i) {
        x -= (float)t;      // exact: the difference of a float and its bf16 rounding fits a float
        t = (__bf16)x;
    }
    return t;
}

// input column of weight element (k-step ks, lane half h, fragment element j) of layer L, or -1
__host__ __device__ inline int k_column(const LayerInfo& L, int ks, int h, int j) {
    if (ks < L.nks_acc) {
        const int f = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
        return f < L.n_in ? f : -1;
    }
    // encoding slots: lane half h owns half of the frequencies, so both halves run the same code
    const int e = 8 * (ks - L.nks_acc) + j;
    if (L.enc == ENC_XYZ) return (h == 0 && e < 3) ? L.n_in + e : -1;
    if (L.enc == ENC_PE) {  // nerf/encoding.py column order: x | sin(2^f x) cos(2^f x) blocks of 3
        if (e < 30) return L.n_in + 3 + 6 * (5 * h + e / 6) + e % 6;
        if (e == 30) return L.n_in + (h ? 2 : 0);
        return h ? -1 : L.n_in + 1;
    }
    if (e < 12) return L.n_in + 3 + 6 * (2 * h + e / 6) + e % 6;
    if (e < 15) return h ? -1 : L.n_in + (e - 12);
    return -1;
}

struct PackBf16Args {
    const float* src[NERF_NUM_PARAM_TENSORS];
    unsigned char* dst;
    LayerInfo layers[NL_MAX];
    int chunk0[NL_MAX + 1];
    int bias_tile0[NL_MAX + 1];
};

template <int P, int MODEL>
__global__ __launch_bounds__(256) void pack_bf16_kernel(PackBf16Args a) {
    constexpr int NCHUNK = Geo<P, MODEL>::NCHUNK, NBIAS_TILES = Geo<P, MODEL>::NBIAS_TILES;
    constexpr size_t IMG_BYTES = Geo<P, MODEL>::IMG_BYTES;
    const int p = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece: unit, lane
    if (p < NCHUNK * CHUNK_UNITS * 64) {
        const int chunk = p / (CHUNK_UNITS * 64), unit = (p / 64) % CHUNK_UNITS, lane = p & 63;
        int l = 0;
        while (chunk >= a.chunk0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int nks = L.nks_acc + enc_ks(L.enc);
        const int up = (chunk - a.chunk0[l]) * CHUNK_UNITS + unit;
        const int plane = up % P, u = up / P;       // the P terms of a fragment are consecutive units
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
        if (u < L.n_ot * nks) {
            const int ks = u / L.n_ot, ot = u % L.n_ot;
            const int out = 32 * ot + (lane & 31), h = lane >> 5;
            if (out < L.n_out) {
                const float* __restrict__ w = (L.last_row_tensor >= 0 && out == L.n_out - 1)
                                                  ? a.src[L.last_row_tensor] : a.src[L.wtensor] + (size_t)out * L.ld;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = k_column(L, ks, h, j);
                    if (col >= 0) v[j] = bf16_term(w[col], plane);
                }
            }
        }
        *reinterpret_cast<bf16x8*>(a.dst + (size_t)p * 16) = v;
    }
    // bias image: tile, lane half, register -> the accumulator layout
    if (p < NBIAS_TILES * 32) {
        const int tile = p >> 5, h = (p >> 4) & 1, g = p & 15;
        int l = 0;
        while (tile >= a.bias_tile0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int out = 32 * (tile - a.bias_tile0[l]) + (g & 3) + 8 * (g >> 2) + 4 * h;
        float b = 0.0f;
        if (out < L.n_out) b = (L.last_row_tensor >= 0 && out == L.n_out - 1) ? a.src[L.last_row_tensor + 1][0] : a.src[L.wtensor + 1][out];
        reinterpret_cast<float*>(a.dst + IMG_BYTES)[p] = b;
    }
}

// ---- forward -----------------------------------------------------------------------------
struct Bf16Args {
    const unsigned char* img;
    int64_t M;
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
#ifdef NERF_DIAG
    unsigned long long* dbg;   // [workgroup][32] shader-clock stamps
#endif
};

#ifdef NERF_DIAG
#define DIAG_STAMP(i)                                                                          \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define DIAG_STAMP(i) ((void)0)
#endif

constexpr int BF16_LDS_BYTES = NBUF * CHUNK_BYTES + NBIAS_TILES_MAX * 32 * 4;

struct Ring {
    const unsigned char* img;
    unsigned char* lds;
    int next;        // chunk the next begin_chunk() hands out
    unsigned voff;   // this thread's byte offset inside a slice of a chunk (16 * tid)
    int wave;
    int nchunk, img_bytes, per_chunk, wg_threads;   // Geo<P, MODEL> of the kernel (compile-time constants after inlining)
};

// piece i (16 bytes per thread) of a chunk: global memory -> LDS without staging registers
__device__ __forceinline__ void ring_issue_piece(const Ring& r, int chunk, int i) {
    if (chunk < r.nchunk) {
        // buffer form of the LDS-DMA: image base and size in a scalar resource, the chunk and the slice in the scalar
        // offset, this thread's 16 * tid in the one vector offset that never changes -- no vector address arithmetic
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(r.img), 0, r.img_bytes, 0x00020000);
        unsigned char* dst = r.lds + (chunk & (NBUF - 1)) * CHUNK_BYTES + 16 * 64 * r.wave;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + i * 16 * r.wg_threads), 16,
                                                 (int)r.voff, chunk * CHUNK_BYTES + i * 16 * r.wg_threads, 0, 0);
    }
}
__device__ __forceinline__ void ring_issue(const Ring& r, int chunk) {
#pragma unroll
    for (int i = 0; i < r.per_chunk; ++i) ring_issue_piece(r, chunk, i);
}

// Ring protocol.  While the waves read chunk c (already published), chunk c+1 is published a few
// units into c: every wave waits for its own pieces of c+1 (counted vmcnt: c+2 may stay in flight),
// all waves meet at one barrier, and the buffer of chunk c-1 -- which no wave can still be reading,
// each having consumed at least one unit of c -- is refilled with chunk c+3, one piece per following unit: an
// LDS-DMA instruction holds the issuing wave for some tens of cycles, which one MFMA covers and eight in a row do
// not (the three-term kernel has no second wave on the SIMD to fill the gap).  The fragment reads of
// chunk c that are in flight at the barrier are untouched by it, so the MFMA stream never drains.
// REM = chunks of the whole stream that follow the one being read (2 = two or more)
template <int P, int REM>
__device__ __forceinline__ void ring_publish_next(Ring& r) {
    if constexpr (REM >= 1) {
        if constexpr (REM >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(per_chunk(P)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

template <int NOT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[9], const float* bias_lds, int tile0, int h) {
    // (hiding this base from the compiler so that the 4*NOT reads share one address register and immediate offsets --
    // the bias image sits above the 64 KiB an LDS offset field reaches -- saves 33 vector additions per layer and
    // measures 7 % SLOWER: the separately computed addresses let the reads issue earlier)
    const float* base = bias_lds + tile0 * 32 + h * 16;
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) {
        const f32x4* b = reinterpret_cast<const f32x4*>(base + ot * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = b[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ot][4 * q + i] = v[i];
        }
    }
}

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): indices are constants by
// construction, whatever the loop unroller's size limits
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// acc[ot] += W tile (ot, ks) * B(ks) for all ks, consuming the layer's units in image order; k-steps
// below NKS_A take their B fragment from XA, the rest from XE (an encoding).  A fragments are read
// from LDS PF units ahead of the MFMAs that use them.  With P = 3 terms, unit 3g+p is term p of fragment g and
// meets the terms 0..2-p of the B fragment (3 + 2 + 1 MFMAs per fragment, consecutive ones on one accumulator
// run at full rate).
// TAIL = chunks that follow this layer's last chunk in the whole stream (2 = two or more).
// FIRST_OT: output tiles below it are not computed (the density-only pass needs the ninth tile of its last layer alone).
template <int P, int NOT, int NKS_A, int NKS_E, int TAIL, int FIRST_OT = 0>
__device__ __forceinline__ void mma_layer16(Ring& r, int lane, f32x16 (&acc)[9], const bf16x8 (&XA)[P][16],
                                            const bf16x8 (&XE)[P][4]) {
    constexpr int NKS = NKS_A + NKS_E, UNITS = P * NOT * NKS, CHUNKS = (UNITS + CHUNK_UNITS - 1) / CHUNK_UNITS;
    constexpr int PF = 4;            // fragment reads in flight (8 measure the same in both kernels)
    constexpr int PUBLISH_AT = PF;   // unit of a chunk at whose fetch the next chunk is published
    constexpr int PER_CHUNK = per_chunk(P);
    static_assert(UNITS >= PF && (UNITS % CHUNK_UNITS == 0 || UNITS % CHUNK_UNITS >= PUBLISH_AT + PER_CHUNK), "chunk too short");
    const unsigned char* buf = nullptr;
    bf16x8 a[PF];
    auto fetch = [&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (u % CHUNK_UNITS == 0) {
            // one address register per chunk (its offset hidden from the compiler), the units at immediate offsets (< 32 KiB):
            // folded into one constant, offsets past 64 KiB would cost a vector addition per unit
            unsigned off = (unsigned)(r.next & (NBUF - 1)) * CHUNK_BYTES + 16u * (unsigned)lane;
            asm volatile("" : "+v"(off));
            ++r.next;
            buf = r.lds + off;
        }
        constexpr int rem = (CHUNKS - 1 - u / CHUNK_UNITS) + TAIL;
        if constexpr (u % CHUNK_UNITS == PUBLISH_AT) ring_publish_next<P, (rem >= 2 ? 2 : rem)>(r);
        if constexpr (rem >= 1 && u % CHUNK_UNITS >= PUBLISH_AT && u % CHUNK_UNITS < PUBLISH_AT + PER_CHUNK)
            ring_issue_piece(r, r.next + 2, u % CHUNK_UNITS - PUBLISH_AT);   // r.next is the chunk just published
        a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u % CHUNK_UNITS) * UNIT_BYTES);
    };
    static_for<PF>(fetch);
    static_for<UNITS>([&](auto U) {
        constexpr int u = decltype(U)::value, p = u % P, ks = (u / P) / NOT, ot = (u / P) % NOT;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ot >= FIRST_OT)      // (FIRST_OT > 0: the units of the skipped tiles still pass through the ring)
            static_for<P - p>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                bf16x8 b;
                if constexpr (ks < NKS_A) b = XA[q][ks];
                else b = XE[q][ks - NKS_A];
                acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % PF], b, acc[ot], 0, 0, 0);
            });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + PF < UNITS) fetch(std::integral_constant<int, u + PF>{});
    });
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// the P bf16 terms of a pair of floats, each as a packed pair
template <int P>
__device__ __forceinline__ void split2(float v0, float v1, unsigned (&w)[P]) {
    w[0] = pack2(v0, v1);
#pragma unroll
    for (int p = 1; p < P; ++p) {
#pragma clang fp contract(off)
        v0 -= __builtin_bit_cast(float, w[p - 1] << 16);           // exact
        v1 -= __builtin_bit_cast(float, w[p - 1] & 0xffff0000u);
        w[p] = pack2(v0, v1);
    }
}

// accumulator tiles -> B fragments of the next layer (two k-steps per 32-feature tile), through the layer's activation:
// FR_SIN is sin(z) of a hidden sine layer (nerf/models.py:89-90 with w0 = 1; the argument is bounded like the fp32 kernel's),
// FR_SIN30 the first sine layer's sin(30 z).  The sine variants exist for the three-term kernel only.
enum FragAct { FR_NONE = 0, FR_RELU = 1, FR_SIN = 2, FR_SIN30 = 3 };
template <int P, int NFT, int ACT>
__device__ __forceinline__ void to_frags(const f32x16 (&acc)[9], bf16x8 (&X)[P][16]) {
    constexpr bool RELU = ACT == FR_RELU;
    static_assert(P > 1 || ACT == FR_NONE || ACT == FR_RELU, "sine layers need the three-term kernel");
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w[P];
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                float v0 = acc[ft][8 * s + 2 * jp], v1 = acc[ft][8 * s + 2 * jp + 1];
                if constexpr (P == 1) {
                    // ReLU on the packed pair: as 16-bit integers, negative bf16 values are negative
                    const unsigned pk = pack2(v0, v1);
                    w[0][jp] = RELU ? __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), s16x2{0, 0}))
                                    : pk;
                } else {
                    if (RELU) {   // (fmaxf puts a canonicalising v_max v, v, v in front of the v_max)
                        asm("v_max_f32 %0, 0, %1" : "=v"(v0) : "v"(v0));
                        asm("v_max_f32 %0, 0, %1" : "=v"(v1) : "v"(v1));
                    }
                    if constexpr (ACT == FR_SIN || ACT == FR_SIN30) {
                        float c;
                        if constexpr (ACT == FR_SIN30) {
                            // the reduction without the library's large-argument path (128 inlined copies of it otherwise):
                            // exact to 1e5 rad, i.e. for first-layer pre-activations up to 3300 -- coordinates are in [-1, 1]
                            nerf_sincosf_bounded(mul_rn(30.0f, v0), &v0, &c);
                            nerf_sincosf_bounded(mul_rn(30.0f, v1), &v1, &c);
                        } else {
                            nerf_sincosf_bounded(v0, &v0, &c);
                            nerf_sincosf_bounded(v1, &v1, &c);
                        }
                    }
                    unsigned t[P];
                    split2<P>(v0, v1, t);
#pragma unroll
                    for (int p = 0; p < P; ++p) w[p][jp] = t[p];
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) X[p][2 * ft + s] = __builtin_bit_cast(bf16x8, w[p]);
        }
}

template <int P, int N>
__device__ __forceinline__ void vals_to_frags(const float (&v)[N], bf16x8 (&X)[P][4]) {
#pragma unroll
    for (int ks = 0; ks < N / 8; ++ks) {
        u32x4 w[P];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            unsigned t[P];
            split2<P>(v[8 * ks + 2 * jp], v[8 * ks + 2 * jp + 1], t);
#pragma unroll
            for (int p = 0; p < P; ++p) w[p][jp] = t[p];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) X[p][ks] = __builtin_bit_cast(bf16x8, w[p]);
    }
}

// sin/cos of s0*2^f*x for f = 0..NF-1.  One term: one accurate sincos at the base frequency, then angle
// doubling (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a); the absolute error doubles per step:
// about 1e-6 after four steps, three orders below half a bf16 ulp of the values it feeds.  Three terms: the
// fp32 path's sincos of the exactly scaled argument at every frequency (nerf/encoding.py:14-18).
template <int P, int NF>
__device__ __forceinline__ void sincos_octaves(float base_arg, float (&sn)[NF], float (&cs)[NF]) {
    nerf_sincosf(base_arg, &sn[0], &cs[0]);
#pragma unroll
    for (int f = 1; f < NF; ++f) {
        if constexpr (P == 1) {
            sn[f] = 2.0f * sn[f - 1] * cs[f - 1];
            cs[f] = fmaf(-2.0f * sn[f - 1], sn[f - 1], 1.0f);
        } else {
            nerf_sincosf(mul_rn((float)(1 << f), base_arg), &sn[f], &cs[f]);
        }
    }
}

__device__ __forceinline__ float norm_coord16(float p, float nearf, float spanf) {
#pragma clang fp contract(off)  // same operation order and rounding as the fp32 path
    return sub_rn(div_rn(mul_rn(2.0f, sub_rn(p, nearf)), spanf), 1.0f);
}

// DENSITY_ONLY (colors == NULL): the stream ends with the layer that holds the density row -- the coarse pass of a
// hierarchical render without gradients feeds only the re-sampling.
template <int P, int MODEL, bool DENSITY_ONLY = false>
__global__ __launch_bounds__(wg_threads(P), 1) void fwd_bf16_kernel(Bf16Args a) {
    using G = Geo<P, MODEL>;
    constexpr int WG_THREADS = G::WG_THREADS, WG_SAMPLES = G::WG_SAMPLES, PER_CHUNK = G::PER_CHUNK;
    constexpr auto bias_tile0 = [](int l) { return G::bias_tile0(l); };
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float* bias_lds = reinterpret_cast<float*>(smem + NBUF * CHUNK_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    DIAG_STAMP(0);
    // Ordinary loads first (inputs, bias image -> LDS): once LDS-DMA loads are in flight the compiler
    // waits for ALL of them at the first use of an ordinary load's result.
    const float* bias_g = reinterpret_cast<const float*>(a.img + G::IMG_BYTES);
    for (int i = tid; i < (int)G::BIAS_FLOATS; i += WG_THREADS) bias_lds[i] = bias_g[i];

    // both lane halves load the sample of lane & 31
    const size_t m = (size_t)blockIdx.x * WG_SAMPLES + wave * 32 + (lane & 31);
    const bool valid = m < (size_t)a.M;
    float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma clang fp contract(off)  // o + t*d must not become an FMA (same positions as the fp32 path)
        if (a.points) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                x[c] = a.points[m * 3 + c];
                d[c] = a.dirs[m * 3 + c];
            }
        } else {
            const size_t ray = m / (size_t)a.S;
            const float ts = a.t_per_ray ? a.t[m] : a.t[m - ray * (size_t)a.S];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d[c] = a.rays_d[ray * 3 + c];
                x[c] = norm_coord16(add_rn(a.rays_o[ray * 3 + c], mul_rn(ts, d[c])), a.nearf, a.spanf);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing ordinary is outstanding from here on
    __builtin_amdgcn_sched_barrier(0);
    constexpr int DENSITY_LAYER = MODEL == NERF_MODEL_NERF ? 9 : 8;
    // chunks of the stream this variant reads (nothing may be in flight towards LDS when the workgroup ends)
    constexpr int NCHUNK_RUN = DENSITY_ONLY ? G::chunk0(DENSITY_LAYER + 1) : G::NCHUNK;
    Ring ring{a.img, smem, 0, 16u * (unsigned)tid, wave, NCHUNK_RUN, (int)G::IMG_BYTES, PER_CHUNK, WG_THREADS};
    ring_issue(ring, 0);   // these land while the encodings below are computed
    ring_issue(ring, 1);
    ring_issue(ring, 2);
    __builtin_amdgcn_sched_barrier(0);

    // First layer's encoding fragments in the slot order k_column() gives the weight image.  NeRF: lane half h owns
    // frequencies 5h..5h+4 of the position, the raw coordinates fill the tail.  Siren: the raw coordinates, half 0.
    bf16x8 E0[P][4];
    if constexpr (MODEL == NERF_MODEL_NERF) {
        float v[32];
        const float s0 = h ? 32.0f : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn[5], cs[5];
            sincos_octaves<P, 5>(mul_rn(s0, x[c]), sn, cs);
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                v[6 * f + c] = sn[f];
                v[6 * f + 3 + c] = cs[f];
            }
        }
        v[30] = h ? x[2] : x[0];
        v[31] = h ? 0.0f : x[1];
        vals_to_frags<P, 32>(v, E0);
    } else {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = h ? 0.0f : x[c];
        vals_to_frags<P, 8>(v, E0);
    }
    // publish chunk 0 and the bias image: own pieces of chunk 0 landed (chunks 1, 2 stay in flight),
    // own LDS writes done, then the barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER_CHUNK) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DIAG_STAMP(1);

    f32x16 acc[9];
    bf16x8 X[P][16];
    // direction encoding (both models: nerf/encoding.py with 4 frequencies): half h owns frequencies 2h..2h+1, the raw
    // direction sits behind them in half 0.  Only computed in front of the layer that reads it: its registers would
    // otherwise be held through the whole trunk.
    auto direction_frags = [&](bf16x8 (&DE)[P][4]) {
        float v[16];
        const float s0 = h ? 4.0f : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn[2], cs[2];
            sincos_octaves<P, 2>(mul_rn(s0, d[c]), sn, cs);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                v[6 * f + c] = sn[f];
                v[6 * f + 3 + c] = cs[f];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[12 + c] = h ? 0.0f : d[c];
        v[15] = 0.0f;
        vals_to_frags<P, 16>(v, DE);
    };
    float dens;

    if constexpr (MODEL == NERF_MODEL_NERF) {
        init_bias<8>(acc, bias_lds, bias_tile0(0), h);
        mma_layer16<P, 8, 0, 4, 2>(ring, lane, acc, X, E0);
        to_frags<P, 8, FR_RELU>(acc, X);
        DIAG_STAMP(2);
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            init_bias<8>(acc, bias_lds, 8 * l, h);
            DIAG_STAMP(10 + 3 * (l - 1));
            mma_layer16<P, 8, 16, 0, 2>(ring, lane, acc, X, E0);
            DIAG_STAMP(11 + 3 * (l - 1));
            to_frags<P, 8, FR_RELU>(acc, X);
            DIAG_STAMP(12 + 3 * (l - 1));
        }
        DIAG_STAMP(3);
        // skip layer: [features | position encoding]
        init_bias<8>(acc, bias_lds, bias_tile0(5), h);
        mma_layer16<P, 8, 16, 4, 2>(ring, lane, acc, X, E0);
        to_frags<P, 8, FR_RELU>(acc, X);
#pragma unroll 1
        for (int l = 6; l <= 8; ++l) {
            init_bias<8>(acc, bias_lds, 8 * l, h);
            mma_layer16<P, 8, 16, 0, 2>(ring, lane, acc, X, E0);
            to_frags<P, 8, FR_RELU>(acc, X);
        }
        // block2.8: 256 feature rows (no activation) + the density row (first row of the ninth tile)
        init_bias<9>(acc, bias_lds, bias_tile0(9), h);
        if constexpr (DENSITY_ONLY) {
            mma_layer16<P, 9, 16, 0, 0, 8>(ring, lane, acc, X, E0);
            if (valid && h == 0) a.density[m] = fmaxf(acc[8][0], 0.0f);
            return;
        }
        mma_layer16<P, 9, 16, 0, 2>(ring, lane, acc, X, E0);
        dens = fmaxf(acc[8][0], 0.0f);
        to_frags<P, 8, FR_NONE>(acc, X);
        // rgb head: [features | direction encoding] -> 128, ReLU
        bf16x8 DE[P][4];
        direction_frags(DE);
        init_bias<4>(acc, bias_lds, bias_tile0(10), h);
        mma_layer16<P, 4, 16, 2, 1>(ring, lane, acc, X, DE);
        to_frags<P, 4, FR_RELU>(acc, X);
        init_bias<1>(acc, bias_lds, bias_tile0(11), h);
        mma_layer16<P, 1, 8, 0, 0>(ring, lane, acc, X, DE);
    } else {
        // nerf/models.py:185-203: sin(30 (W0 x + b0)), seven sin(W h + b) layers, then feature_remap + density_branch
        // as one nine-tile layer, the 283 -> 128 sine layer of the rgb head and its 128 -> 3 output layer
        init_bias<8>(acc, bias_lds, bias_tile0(0), h);
        mma_layer16<P, 8, 0, 1, 2>(ring, lane, acc, X, E0);
        to_frags<P, 8, FR_SIN30>(acc, X);
        DIAG_STAMP(2);
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            init_bias<8>(acc, bias_lds, 8 * l, h);
            mma_layer16<P, 8, 16, 0, 2>(ring, lane, acc, X, E0);
            to_frags<P, 8, FR_SIN>(acc, X);
        }
        DIAG_STAMP(3);
        init_bias<9>(acc, bias_lds, bias_tile0(8), h);
        if constexpr (DENSITY_ONLY) {
            mma_layer16<P, 9, 16, 0, 0, 8>(ring, lane, acc, X, E0);
            if (valid && h == 0) a.density[m] = fmaxf(acc[8][0], 0.0f) * 10.0f;
            return;
        }
        mma_layer16<P, 9, 16, 0, 2>(ring, lane, acc, X, E0);
        dens = fmaxf(acc[8][0], 0.0f) * 10.0f;           // relu(sigma) * sigma_mul (models.py:193)
        to_frags<P, 8, FR_NONE>(acc, X);
        bf16x8 DE[P][4];
        direction_frags(DE);
        init_bias<4>(acc, bias_lds, bias_tile0(9), h);
        mma_layer16<P, 4, 16, 2, 1>(ring, lane, acc, X, DE);
        to_frags<P, 4, FR_SIN>(acc, X);
        init_bias<1>(acc, bias_lds, bias_tile0(10), h);
        mma_layer16<P, 1, 8, 0, 0>(ring, lane, acc, X, DE);
    }

    DIAG_STAMP(4);
    if (valid && h == 0) {  // rows 0..2 of the last tile and the density row sit in lane half 0
#pragma unroll
        for (int c = 0; c < 3; ++c) a.colors[m * 3 + c] = dev::sigmoidf_(acc[0][c]);
        a.density[m] = dens;
    }
    DIAG_STAMP(5);
}

template <int P, int MODEL>
int pack_image(const float* const* params, void* packed, void* stream, const char* who) {
    using G = Geo<P, MODEL>;
    PackBf16Args a;
    for (int i = 0; i < NERF_NUM_PARAM_TENSORS; ++i) {
        if (!params[i]) return nerf_fail(NERF_E_NULL, "%s: params[%d] is NULL", who, i);
        a.src[i] = params[i];
    }
    a.dst = static_cast<unsigned char*>(packed);
    for (int l = 0; l < NL_MAX; ++l) a.layers[l] = Net<MODEL>::LAYERS[l];
    for (int l = 0; l <= NL_MAX; ++l) {
        a.chunk0[l] = G::chunk0(l < G::NL ? l : G::NL);
        a.bias_tile0[l] = G::bias_tile0(l < G::NL ? l : G::NL);
    }
    const int pieces = G::NCHUNK * CHUNK_UNITS * 64;
    pack_bf16_kernel<P, MODEL><<<(pieces + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    return nerf_check_launch(who);
}

template <int P, int MODEL>
int launch_fwd(const Bf16Args& a, int64_t M, void* stream, const char* who) {
    const int64_t groups = ceil_div64(M, Geo<P, MODEL>::WG_SAMPLES);
    if (groups >= (1ll << 31)) return nerf_fail(NERF_E_ARG, "%s: too many samples", who);
    if (a.colors) {
        if (int rc = nerf_set_lds(fwd_bf16_kernel<P, MODEL, false>, BF16_LDS_BYTES)) return rc;
        fwd_bf16_kernel<P, MODEL, false><<<(unsigned)groups, wg_threads(P), BF16_LDS_BYTES, (hipStream_t)stream>>>(a);
    } else {
        if (int rc = nerf_set_lds(fwd_bf16_kernel<P, MODEL, true>, BF16_LDS_BYTES)) return rc;
        fwd_bf16_kernel<P, MODEL, true><<<(unsigned)groups, wg_threads(P), BF16_LDS_BYTES, (hipStream_t)stream>>>(a);
    }
    return nerf_check_launch(who);
}

#ifdef NERF_DIAG
unsigned long long* g_bf16_dbg = nullptr;
#endif

// P = 1: the NeRF model only (sin(30 z) of bf16-rounded inputs is no usable Siren); P = 3: both models
template <int P>
int fwd_terms(const char* who, int model, const void* packed, int64_t M, const float* points, const float* dirs,
              const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near, double far,
              float* colors, float* density, void* stream) {
    if (!(model == NERF_MODEL_NERF || (P == 3 && model == NERF_MODEL_SIREN)))
        return nerf_fail(NERF_E_ARG, P == 1 ? "%s: the bf16 mode covers the NeRF model only (model %d)" : "%s: unknown model %d", who, model);
    if (M < 0) return nerf_fail(NERF_E_ARG, "%s: M=%lld < 0", who, (long long)M);
    if (M == 0) return NERF_OK;
    if (!packed || !density) return nerf_fail(NERF_E_NULL, "%s: packed and density must not be NULL", who);
    if (points) {
        if (!dirs) return nerf_fail(NERF_E_NULL, "%s: dirs is NULL", who);
    } else {
        if (!rays_o || !rays_d || !t) return nerf_fail(NERF_E_NULL, "%s: rays_o, rays_d and t must not be NULL without points", who);
        if (S < 1 || M % S != 0) return nerf_fail(NERF_E_ARG, "%s: M=%lld is not a multiple of S=%d", who, (long long)M, S);
        if (far == near) return nerf_fail(NERF_E_ARG, "%s: near == far", who);
    }
    Bf16Args a{static_cast<const unsigned char*>(packed), M, points, dirs, rays_o, rays_d, t, S, t_per_ray,
               (float)near, (float)(far - near), colors, density
#ifdef NERF_DIAG
               , g_bf16_dbg
#endif
    };
    if constexpr (P == 3) {
        if (model == NERF_MODEL_SIREN) return launch_fwd<3, NERF_MODEL_SIREN>(a, M, stream, who);
    }
    return launch_fwd<P, NERF_MODEL_NERF>(a, M, stream, who);
}

}  // namespace

#ifdef NERF_DIAG
extern "C" void nerf_diag_bf16_stamps(unsigned long long* device_buffer) { g_bf16_dbg = device_buffer; }
#endif

extern "C" size_t nerf_mlp_packed_bf16_bytes(int model) { return model == NERF_MODEL_NERF ? Geo<1, NERF_MODEL_NERF>::PACKED_BYTES : 0; }
extern "C" int nerf_mlp_pack_bf16(int model, const float* const* params, void* packed, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF, "nerf_mlp_pack_bf16: the bf16 mode covers the NeRF model only (model %d)", model);
    NERF_REQUIRE_PTR(params);
    NERF_REQUIRE_PTR(packed);
    return pack_image<1, NERF_MODEL_NERF>(params, packed, stream, "nerf_mlp_pack_bf16");
}

extern "C" int nerf_mlp_fwd_bf16(int model, const void* packed, int64_t M, const float* points, const float* dirs,
                                 const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray,
                                 double near, double far, float* colors, float* density, void* stream) {
    return fwd_terms<1>("nerf_mlp_fwd_bf16", model, packed, M, points, dirs, rays_o, rays_d, t, S, t_per_ray, near, far, colors,
                        density, stream);
}

