// Opt-in bf16 inference forward of the NeRF MLP (nerf/models.py:62-75): bf16 operands, fp32
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
// LDS buffers with global_load_lds (no staging registers); one barrier per 32 KiB chunk both
// publishes chunk c and releases the buffer of chunk c-1 for chunk c+3.  The image is stored in
// consumption order, one 1 KiB unit per MFMA: unit[lane][8 bf16] is read back with one
// conflict-free ds_read_b128 per lane.
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
constexpr int WG_THREADS = 512;
constexpr int WG_SAMPLES = 256;

// ---- layer table -------------------------------------------------------------------------
enum Enc { ENC_NONE = 0, ENC_PE = 1, ENC_DE = 2 };
struct LayerInfo {
    int n_ot;      // 32-row output tiles
    int nks_acc;   // k-steps fed by the previous layer's accumulators (16 features each)
    int enc;       // extra k-steps fed by an encoding: ENC_PE 4, ENC_DE 2
    int n_out;     // valid output rows
    int n_in;      // valid accumulator-fed inputs (also the column where the encoding part starts)
    int ld;        // row pitch of the [out, in] weight tensor
    int wtensor;   // state_dict index of the weight; the bias follows it
};
constexpr int NL = 12;
constexpr LayerInfo LAYERS[NL] = {
    {8, 0, ENC_PE, 256, 0, 63, 0},      {8, 16, ENC_NONE, 256, 256, 256, 2},  {8, 16, ENC_NONE, 256, 256, 256, 4},
    {8, 16, ENC_NONE, 256, 256, 256, 6}, {8, 16, ENC_NONE, 256, 256, 256, 8},  {8, 16, ENC_PE, 256, 256, 319, 10},
    {8, 16, ENC_NONE, 256, 256, 256, 12}, {8, 16, ENC_NONE, 256, 256, 256, 14}, {8, 16, ENC_NONE, 256, 256, 256, 16},
    {9, 16, ENC_NONE, 257, 256, 256, 18}, {4, 16, ENC_DE, 128, 256, 283, 20},   {1, 8, ENC_NONE, 3, 128, 128, 22},
};
constexpr int enc_ks(int enc) { return enc == ENC_PE ? 4 : enc == ENC_DE ? 2 : 0; }
constexpr int layer_ks(int l) { return LAYERS[l].nks_acc + enc_ks(LAYERS[l].enc); }
constexpr int layer_units(int l) { return LAYERS[l].n_ot * layer_ks(l); }
constexpr int layer_chunks(int l) { return (layer_units(l) + CHUNK_UNITS - 1) / CHUNK_UNITS; }
constexpr int chunk0(int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += layer_chunks(i);
    return c;
}
constexpr int bias_tile0(int l) {
    int t = 0;
    for (int i = 0; i < l; ++i) t += LAYERS[i].n_ot;
    return t;
}
constexpr int NCHUNK = chunk0(NL);
constexpr int NBIAS_TILES = bias_tile0(NL);
constexpr size_t IMG_BYTES = (size_t)NCHUNK * CHUNK_BYTES;
constexpr size_t BIAS_FLOATS = (size_t)NBIAS_TILES * 32;
constexpr size_t PACKED_BYTES = IMG_BYTES + BIAS_FLOATS * 4;
static_assert(NCHUNK == 43 && NBIAS_TILES == 86, "bf16 image layout");

// input column of weight element (k-step ks, lane half h, fragment element j) of layer L, or -1
__host__ __device__ inline int k_column(const LayerInfo& L, int ks, int h, int j) {
    if (ks < L.nks_acc) {
        const int f = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
        return f < L.n_in ? f : -1;
    }
    // encoding slots: lane half h owns half of the frequencies, so both halves run the same code
    const int e = 8 * (ks - L.nks_acc) + j;
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
    LayerInfo layers[NL];
    int chunk0[NL + 1];
    int bias_tile0[NL + 1];
};

__global__ __launch_bounds__(256) void pack_bf16_kernel(PackBf16Args a) {
    const int p = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece: unit, lane
    if (p < NCHUNK * CHUNK_UNITS * 64) {
        const int chunk = p / (CHUNK_UNITS * 64), unit = (p / 64) % CHUNK_UNITS, lane = p & 63;
        int l = 0;
        while (chunk >= a.chunk0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int nks = L.nks_acc + (L.enc == ENC_PE ? 4 : L.enc == ENC_DE ? 2 : 0);
        const int u = (chunk - a.chunk0[l]) * CHUNK_UNITS + unit;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
        if (u < L.n_ot * nks) {
            const int ks = u / L.n_ot, ot = u % L.n_ot;
            const int out = 32 * ot + (lane & 31), h = lane >> 5;
            if (out < L.n_out) {
                const float* __restrict__ w = a.src[L.wtensor] + (size_t)out * L.ld;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = k_column(L, ks, h, j);
                    if (col >= 0) v[j] = (__bf16)w[col];
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
        reinterpret_cast<float*>(a.dst + IMG_BYTES)[p] = out < L.n_out ? a.src[L.wtensor + 1][out] : 0.0f;
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

constexpr int BF16_LDS_BYTES = NBUF * CHUNK_BYTES + (int)BIAS_FLOATS * 4;

struct Ring {
    const unsigned char* img;
    unsigned char* lds;
    int next;        // chunk the next begin_chunk() hands out
    unsigned voff;   // this thread's byte offset inside a 8 KiB slice of a chunk (16 * tid)
    int wave;
};
constexpr int PER_CHUNK = CHUNK_BYTES / (16 * WG_THREADS);   // global_load_lds instructions per thread and chunk

__device__ __forceinline__ void ring_issue(const Ring& r, int chunk) {
    if (chunk < NCHUNK) {
        // buffer form of the LDS-DMA: image base and size in a scalar resource, the chunk and the slice in the scalar
        // offset, this thread's 16 * tid in the one vector offset that never changes -- no vector address arithmetic
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(r.img), 0, (int)IMG_BYTES, 0x00020000);
        unsigned char* dst = r.lds + (chunk & (NBUF - 1)) * CHUNK_BYTES + 16 * 64 * r.wave;
#pragma unroll
        for (int i = 0; i < PER_CHUNK; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + i * 16 * WG_THREADS), 16,
                                                     (int)r.voff, chunk * CHUNK_BYTES + i * 16 * WG_THREADS, 0, 0);
    }
}

// Ring protocol.  While the waves read chunk c (already published), chunk c+1 is published a few
// units into c: every wave waits for its own pieces of c+1 (counted vmcnt: c+2 may stay in flight),
// all waves meet at one barrier, and the buffer of chunk c-1 -- which no wave can still be reading,
// each having consumed at least one unit of c -- is refilled with chunk c+3.  The fragment reads of
// chunk c that are in flight at the barrier are untouched by it, so the MFMA stream never drains.
__device__ __forceinline__ const unsigned char* ring_begin_chunk(Ring& r) {
    const unsigned char* buf = r.lds + (r.next & (NBUF - 1)) * CHUNK_BYTES;
    ++r.next;
    return buf;
}
// REM = chunks of the whole stream that follow the one being read (2 = two or more)
template <int REM>
__device__ __forceinline__ void ring_publish_next(Ring& r) {
    if constexpr (REM >= 1) {
        if constexpr (REM >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ring_issue(r, r.next + 2);   // r.next is the chunk just published
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

// acc[ot] += W tile (ot, ks) * B(ks) for all ks, consuming the layer's units in image order; k-steps
// below NKS_A take their B fragment from XA, the rest from XE (an encoding).  A fragments are read
// from LDS PF units ahead of the MFMA that uses them.
// TAIL = chunks that follow this layer's last chunk in the whole stream (2 = two or more).
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

template <int NOT, int NKS_A, int NKS_E, int TAIL>
__device__ __forceinline__ void mma_layer16(Ring& r, int lane, f32x16 (&acc)[9], const bf16x8 (&XA)[16],
                                            const bf16x8 (&XE)[4]) {
    constexpr int NKS = NKS_A + NKS_E, UNITS = NOT * NKS, CHUNKS = (UNITS + CHUNK_UNITS - 1) / CHUNK_UNITS;
    constexpr int PF = 4;            // fragment reads in flight
    constexpr int PUBLISH_AT = PF;   // unit of a chunk at whose fetch the next chunk is published
    static_assert(UNITS >= PF && (UNITS % CHUNK_UNITS == 0 || UNITS % CHUNK_UNITS > PUBLISH_AT), "chunk too short");
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
        if constexpr (u % CHUNK_UNITS == PUBLISH_AT) {
            constexpr int rem = (CHUNKS - 1 - u / CHUNK_UNITS) + TAIL;
            ring_publish_next<(rem >= 2 ? 2 : rem)>(r);
        }
        a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u % CHUNK_UNITS) * UNIT_BYTES);
    };
    static_for<PF>(fetch);
    static_for<UNITS>([&](auto U) {
        constexpr int u = decltype(U)::value, ks = u / NOT, ot = u % NOT;
        bf16x8 b;
        if constexpr (ks < NKS_A) b = XA[ks];
        else b = XE[ks - NKS_A];
        __builtin_amdgcn_sched_barrier(0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % PF], b, acc[ot], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + PF < UNITS) fetch(std::integral_constant<int, u + PF>{});
    });
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// accumulator tiles -> B fragments of the next layer (two k-steps per 32-feature tile)
template <int NFT, bool RELU>
__device__ __forceinline__ void to_frags(const f32x16 (&acc)[9], bf16x8* X) {
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                // ReLU on the packed pair: as 16-bit integers, negative bf16 values are negative
                const unsigned pk = pack2(acc[ft][8 * s + 2 * jp], acc[ft][8 * s + 2 * jp + 1]);
                w[jp] = RELU ? __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), s16x2{0, 0}))
                             : pk;
            }
            X[2 * ft + s] = __builtin_bit_cast(bf16x8, w);
        }
}

template <int N>
__device__ __forceinline__ void vals_to_frags(const float (&v)[N], bf16x8* X) {
#pragma unroll
    for (int ks = 0; ks < N / 8; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) w[jp] = pack2(v[8 * ks + 2 * jp], v[8 * ks + 2 * jp + 1]);
        X[ks] = __builtin_bit_cast(bf16x8, w);
    }
}

// sin/cos of s0*2^f*x for f = 0..NF-1: one accurate sincos at the base frequency, then angle
// doubling (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a).  The absolute error doubles per step:
// about 1e-6 after four steps, three orders below half a bf16 ulp of the values it feeds.
template <int NF>
__device__ __forceinline__ void sincos_octaves(float base_arg, float (&sn)[NF], float (&cs)[NF]) {
    nerf_sincosf(base_arg, &sn[0], &cs[0]);
#pragma unroll
    for (int f = 1; f < NF; ++f) {
        sn[f] = 2.0f * sn[f - 1] * cs[f - 1];
        cs[f] = fmaf(-2.0f * sn[f - 1], sn[f - 1], 1.0f);
    }
}

__device__ __forceinline__ float norm_coord16(float p, float nearf, float spanf) {
#pragma clang fp contract(off)  // same operation order and rounding as the fp32 path
    return sub_rn(div_rn(mul_rn(2.0f, sub_rn(p, nearf)), spanf), 1.0f);
}

__global__ __launch_bounds__(WG_THREADS, 1) void nerf_fwd_bf16_kernel(Bf16Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float* bias_lds = reinterpret_cast<float*>(smem + NBUF * CHUNK_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    DIAG_STAMP(0);
    // Ordinary loads first (inputs, bias image -> LDS): once LDS-DMA loads are in flight the compiler
    // waits for ALL of them at the first use of an ordinary load's result.
    const float* bias_g = reinterpret_cast<const float*>(a.img + IMG_BYTES);
    for (int i = tid; i < (int)BIAS_FLOATS; i += WG_THREADS) bias_lds[i] = bias_g[i];

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
    Ring ring{a.img, smem, 0, 16u * (unsigned)tid, wave};
    ring_issue(ring, 0);   // these land while the encodings below are computed
    ring_issue(ring, 1);
    ring_issue(ring, 2);
    __builtin_amdgcn_sched_barrier(0);

    // encodings in the slot order k_column() gives the weight image: half h owns frequencies
    // 5h..5h+4 of the position and 2h..2h+1 of the direction, the raw coordinates fill the tail
    bf16x8 PE[4];
    {
        float v[32];
        const float s0 = h ? 32.0f : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn[5], cs[5];
            sincos_octaves<5>(mul_rn(s0, x[c]), sn, cs);
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                v[6 * f + c] = sn[f];
                v[6 * f + 3 + c] = cs[f];
            }
        }
        v[30] = h ? x[2] : x[0];
        v[31] = h ? 0.0f : x[1];
        vals_to_frags<32>(v, PE);
    }
    // publish chunk 0 and the bias image: own pieces of chunk 0 landed (chunks 1, 2 stay in flight),
    // own LDS writes done, then the barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER_CHUNK) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DIAG_STAMP(1);

    f32x16 acc[9];
    bf16x8 X[16];

    init_bias<8>(acc, bias_lds, bias_tile0(0), h);
    mma_layer16<8, 0, 4, 2>(ring, lane, acc, X, PE);
    to_frags<8, true>(acc, X);
    DIAG_STAMP(2);
#pragma unroll 1
    for (int l = 1; l <= 4; ++l) {
        init_bias<8>(acc, bias_lds, 8 * l, h);
        DIAG_STAMP(10 + 3 * (l - 1));
        mma_layer16<8, 16, 0, 2>(ring, lane, acc, X, PE);
        DIAG_STAMP(11 + 3 * (l - 1));
        to_frags<8, true>(acc, X);
        DIAG_STAMP(12 + 3 * (l - 1));
    }
    DIAG_STAMP(3);
    // skip layer: [features | position encoding]
    init_bias<8>(acc, bias_lds, bias_tile0(5), h);
    mma_layer16<8, 16, 4, 2>(ring, lane, acc, X, PE);
    to_frags<8, true>(acc, X);
#pragma unroll 1
    for (int l = 6; l <= 8; ++l) {
        init_bias<8>(acc, bias_lds, 8 * l, h);
        mma_layer16<8, 16, 0, 2>(ring, lane, acc, X, PE);
        to_frags<8, true>(acc, X);
    }
    // block2.8: 256 feature rows (no activation) + the density row (first row of the ninth tile)
    init_bias<9>(acc, bias_lds, bias_tile0(9), h);
    mma_layer16<9, 16, 0, 2>(ring, lane, acc, X, PE);
    const float dens = fmaxf(acc[8][0], 0.0f);
    to_frags<8, false>(acc, X);
    // rgb head: [features | direction encoding] -> 128, ReLU.  The direction encoding is only
    // computed now: its registers would otherwise be held through the whole trunk.
    bf16x8 DE[4];
    {
        float v[16];
        const float s0 = h ? 4.0f : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn[2], cs[2];
            sincos_octaves<2>(mul_rn(s0, d[c]), sn, cs);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                v[6 * f + c] = sn[f];
                v[6 * f + 3 + c] = cs[f];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[12 + c] = h ? 0.0f : d[c];
        v[15] = 0.0f;
        vals_to_frags<16>(v, DE);
    }
    init_bias<4>(acc, bias_lds, bias_tile0(10), h);
    mma_layer16<4, 16, 2, 1>(ring, lane, acc, X, DE);
    to_frags<4, true>(acc, X);
    init_bias<1>(acc, bias_lds, bias_tile0(11), h);
    mma_layer16<1, 8, 0, 0>(ring, lane, acc, X, DE);

    DIAG_STAMP(4);
    if (valid && h == 0) {  // rows 0..2 of the last tile and the density row sit in lane half 0
#pragma unroll
        for (int c = 0; c < 3; ++c) a.colors[m * 3 + c] = dev::sigmoidf_(acc[0][c]);
        a.density[m] = dens;
    }
    DIAG_STAMP(5);
}

}  // namespace

#ifdef NERF_DIAG
static unsigned long long* g_bf16_dbg = nullptr;
extern "C" void nerf_diag_bf16_stamps(unsigned long long* device_buffer) { g_bf16_dbg = device_buffer; }
#endif

extern "C" size_t nerf_mlp_packed_bf16_bytes(int model) { return model == NERF_MODEL_NERF ? PACKED_BYTES : 0; }

extern "C" int nerf_mlp_pack_bf16(int model, const float* const* params, void* packed, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF, "nerf_mlp_pack_bf16: the bf16 mode covers the NeRF model only (model %d)", model);
    NERF_REQUIRE_PTR(params);
    NERF_REQUIRE_PTR(packed);
    PackBf16Args a;
    for (int i = 0; i < NERF_NUM_PARAM_TENSORS; ++i) {
        if (!params[i]) return nerf_fail(NERF_E_NULL, "nerf_mlp_pack_bf16: params[%d] is NULL", i);
        a.src[i] = params[i];
    }
    a.dst = static_cast<unsigned char*>(packed);
    for (int l = 0; l < NL; ++l) a.layers[l] = LAYERS[l];
    for (int l = 0; l <= NL; ++l) {
        a.chunk0[l] = chunk0(l);
        a.bias_tile0[l] = bias_tile0(l);
    }
    const int pieces = NCHUNK * CHUNK_UNITS * 64;
    pack_bf16_kernel<<<(pieces + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_pack_bf16");
}

extern "C" int nerf_mlp_fwd_bf16(int model, const void* packed, int64_t M, const float* points, const float* dirs,
                                 const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray,
                                 double near, double far, float* colors, float* density, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF, "nerf_mlp_fwd_bf16: the bf16 mode covers the NeRF model only (model %d)", model);
    NERF_REQUIRE(M >= 0, "nerf_mlp_fwd_bf16: M=%lld < 0", (long long)M);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE_PTR(packed);
    NERF_REQUIRE_PTR(colors);
    NERF_REQUIRE_PTR(density);
    if (points) {
        NERF_REQUIRE_PTR(dirs);
    } else {
        NERF_REQUIRE_PTR(rays_o);
        NERF_REQUIRE_PTR(rays_d);
        NERF_REQUIRE_PTR(t);
        NERF_REQUIRE(S >= 1 && M % S == 0, "nerf_mlp_fwd_bf16: M=%lld is not a multiple of S=%d", (long long)M, S);
        NERF_REQUIRE(far != near, "nerf_mlp_fwd_bf16: near == far");
    }
    const int64_t groups = ceil_div64(M, WG_SAMPLES);
    NERF_REQUIRE(groups < (1ll << 31), "nerf_mlp_fwd_bf16: too many samples");
    Bf16Args a{static_cast<const unsigned char*>(packed), M, points, dirs, rays_o, rays_d, t, S, t_per_ray,
               (float)near, (float)(far - near), colors, density
#ifdef NERF_DIAG
               , g_bf16_dbg
#endif
    };
    if (int rc = nerf_set_lds(nerf_fwd_bf16_kernel, BF16_LDS_BYTES)) return rc;
    nerf_fwd_bf16_kernel<<<(unsigned)groups, WG_THREADS, BF16_LDS_BYTES, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_fwd_bf16");
}
