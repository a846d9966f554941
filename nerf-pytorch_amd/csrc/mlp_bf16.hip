// Opt-in inference forward on the bf16 matrix pipe, NeRF MLP (nerf/models.py:62-75): bf16 operands, fp32
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
// The three-term mode ("bf16x3": fp32 accuracy on this pipe, both models) started as a variant of this kernel and now
// has a kernel of its own on 16-sample tiles, mlp_x3i.hip.
#include "bf16_common.h"

using namespace bf16k;

namespace {

constexpr int CHUNK_UNITS = 32;
constexpr int CHUNK_BYTES = CHUNK_UNITS * UNIT_BYTES;
constexpr int NBUF = 4;
constexpr int WG_THREADS = 512;                              // 8 waves of 32 samples
constexpr int WG_SAMPLES = WG_THREADS / 2;
constexpr int PER_CHUNK = CHUNK_BYTES / (16 * WG_THREADS);   // LDS-DMA instructions per thread and chunk

constexpr int enc_ks(int enc) { return enc == ENC_PE ? 4 : enc == ENC_DE ? 2 : 0; }

// Geometry of the weight stream: one unit per A fragment, every layer starts a new chunk.
namespace geo {
using N = Net<NERF_MODEL_NERF>;
constexpr int NL = N::NL;
constexpr int layer_ks(int l) { return N::LAYERS[l].nks_acc + enc_ks(N::LAYERS[l].enc); }
constexpr int layer_units(int l) { return N::LAYERS[l].n_ot * layer_ks(l); }
constexpr int layer_chunks(int l) { return (layer_units(l) + CHUNK_UNITS - 1) / CHUNK_UNITS; }
constexpr int chunk0(int l) {
    int c = 0;
    for (int i = 0; i < l; ++i) c += layer_chunks(i);
    return c;
}
constexpr int bias_tile0(int l) {
    int t = 0;
    for (int i = 0; i < l; ++i) t += N::LAYERS[i].n_ot;
    return t;
}
}  // namespace geo
struct Geo {
    static constexpr int NL = geo::NL;
    static constexpr int chunk0(int l) { return geo::chunk0(l); }
    static constexpr int bias_tile0(int l) { return geo::bias_tile0(l); }
    static constexpr int NCHUNK = geo::chunk0(geo::NL);
    static constexpr int NBIAS_TILES = geo::bias_tile0(geo::NL);
    static constexpr size_t BIAS_FLOATS = (size_t)NBIAS_TILES * 32;
    static constexpr size_t IMG_BYTES = (size_t)NCHUNK * CHUNK_BYTES;
    static constexpr size_t PACKED_BYTES = IMG_BYTES + BIAS_FLOATS * 4;
};
static_assert(Geo::NCHUNK == 43 && Geo::NBIAS_TILES == 86, "bf16 image layout");

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
    LayerInfo layers[NL_MAX];
    int chunk0[NL_MAX + 1];
    int bias_tile0[NL_MAX + 1];
};

__global__ __launch_bounds__(256) void pack_bf16_kernel(PackBf16Args a) {
    const int p = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece: unit, lane
    if (p < Geo::NCHUNK * CHUNK_UNITS * 64) {
        const int chunk = p / (CHUNK_UNITS * 64), unit = (p / 64) % CHUNK_UNITS, lane = p & 63;
        int l = 0;
        while (chunk >= a.chunk0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int nks = L.nks_acc + enc_ks(L.enc);
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
    if (p < Geo::NBIAS_TILES * 32) {
        const int tile = p >> 5, h = (p >> 4) & 1, g = p & 15;
        int l = 0;
        while (tile >= a.bias_tile0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int out = 32 * (tile - a.bias_tile0[l]) + (g & 3) + 8 * (g >> 2) + 4 * h;
        reinterpret_cast<float*>(a.dst + Geo::IMG_BYTES)[p] = out < L.n_out ? a.src[L.wtensor + 1][out] : 0.0f;
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
};

constexpr int BF16_LDS_BYTES = NBUF * CHUNK_BYTES + Geo::NBIAS_TILES * 32 * 4;

struct Ring {
    const unsigned char* img;
    unsigned char* lds;
    int next;        // chunk the next begin_chunk() hands out
    unsigned voff;   // this thread's byte offset inside a slice of a chunk (16 * tid)
    int wave;
    int nchunk;      // chunks of the stream the kernel variant reads
};

// piece i (16 bytes per thread) of a chunk: global memory -> LDS without staging registers
__device__ __forceinline__ void ring_issue_piece(const Ring& r, int chunk, int i) {
    if (chunk < r.nchunk) {
        // buffer form of the LDS-DMA: image base and size in a scalar resource, the chunk and the slice in the scalar
        // offset, this thread's 16 * tid in the one vector offset that never changes -- no vector address arithmetic
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(r.img), 0, (int)Geo::IMG_BYTES, 0x00020000);
        unsigned char* dst = r.lds + (chunk & (NBUF - 1)) * CHUNK_BYTES + 16 * 64 * r.wave;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + i * 16 * WG_THREADS), 16,
                                                 (int)r.voff, chunk * CHUNK_BYTES + i * 16 * WG_THREADS, 0, 0);
    }
}
__device__ __forceinline__ void ring_issue(const Ring& r, int chunk) {
#pragma unroll
    for (int i = 0; i < PER_CHUNK; ++i) ring_issue_piece(r, chunk, i);
}

// Ring protocol.  While the waves read chunk c (already published), chunk c+1 is published a few
// units into c: every wave waits for its own pieces of c+1 (counted vmcnt: c+2 may stay in flight),
// all waves meet at one barrier, and the buffer of chunk c-1 -- which no wave can still be reading,
// each having consumed at least one unit of c -- is refilled with chunk c+3, one piece per following unit: an
// LDS-DMA instruction holds the issuing wave for some tens of cycles, which one MFMA covers and eight in a row do
// not.  The fragment reads of chunk c that are in flight at the barrier are untouched by it, so the MFMA stream never drains.
// REM = chunks of the whole stream that follow the one being read (2 = two or more)
template <int REM>
__device__ __forceinline__ void ring_publish_next() {
    if constexpr (REM >= 1) {
        if constexpr (REM >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK) : "memory");
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

// acc[ot] += W tile (ot, ks) * B(ks) for all ks, consuming the layer's units in image order; k-steps
// below NKS_A take their B fragment from XA, the rest from XE (an encoding).  A fragments are read
// from LDS PF units ahead of the MFMAs that use them.
// TAIL = chunks that follow this layer's last chunk in the whole stream (2 = two or more).
// FIRST_OT: output tiles below it are not computed (the density-only pass needs the ninth tile of its last layer alone).
template <int NOT, int NKS_A, int NKS_E, int TAIL, int FIRST_OT = 0>
__device__ __forceinline__ void mma_layer16(Ring& r, int lane, f32x16 (&acc)[9], const bf16x8 (&XA)[16], const bf16x8 (&XE)[4]) {
    constexpr int NKS = NKS_A + NKS_E, UNITS = NOT * NKS, CHUNKS = (UNITS + CHUNK_UNITS - 1) / CHUNK_UNITS;
    constexpr int PF = 4;            // fragment reads in flight (8 measure the same)
    constexpr int PUBLISH_AT = PF;   // unit of a chunk at whose fetch the next chunk is published
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
        if constexpr (u % CHUNK_UNITS == PUBLISH_AT) ring_publish_next<(rem >= 2 ? 2 : rem)>();
        if constexpr (rem >= 1 && u % CHUNK_UNITS >= PUBLISH_AT && u % CHUNK_UNITS < PUBLISH_AT + PER_CHUNK)
            ring_issue_piece(r, r.next + 2, u % CHUNK_UNITS - PUBLISH_AT);   // r.next is the chunk just published
        a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u % CHUNK_UNITS) * UNIT_BYTES);
    };
    static_for<PF>(fetch);
    static_for<UNITS>([&](auto U) {
        constexpr int u = decltype(U)::value, ks = u / NOT, ot = u % NOT;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ot >= FIRST_OT) {      // (FIRST_OT > 0: the units of the skipped tiles still pass through the ring)
            if constexpr (ks < NKS_A) acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % PF], XA[ks], acc[ot], 0, 0, 0);
            else acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % PF], XE[ks - NKS_A], acc[ot], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + PF < UNITS) fetch(std::integral_constant<int, u + PF>{});
    });
}

// accumulator tiles -> B fragments of the next layer (two k-steps per 32-feature tile), through the layer's activation
template <int NFT, bool RELU>
__device__ __forceinline__ void to_frags(const f32x16 (&acc)[9], bf16x8 (&X)[16]) {
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w;
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                // ReLU on the packed pair: as 16-bit integers, negative bf16 values are negative
                const unsigned pk = pack2(acc[ft][8 * s + 2 * jp], acc[ft][8 * s + 2 * jp + 1]);
                w[jp] = RELU ? __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), s16x2{0, 0})) : pk;
            }
            X[2 * ft + s] = __builtin_bit_cast(bf16x8, w);
        }
}

template <int N>
__device__ __forceinline__ void vals_to_frags(const float (&v)[N], bf16x8 (&X)[4]) {
#pragma unroll
    for (int ks = 0; ks < N / 8; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) w[jp] = pack2(v[8 * ks + 2 * jp], v[8 * ks + 2 * jp + 1]);
        X[ks] = __builtin_bit_cast(bf16x8, w);
    }
}

// sin/cos of s0*2^f*x for f = 0..NF-1: one accurate sincos at the base frequency, then angle
// doubling (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a); the absolute error doubles per step:
// about 1e-6 after four steps, three orders below half a bf16 ulp of the values it feeds (nerf/encoding.py:14-18).
template <int NF>
__device__ __forceinline__ void sincos_octaves(float base_arg, float (&sn)[NF], float (&cs)[NF]) {
    nerf_sincosf(base_arg, &sn[0], &cs[0]);
#pragma unroll
    for (int f = 1; f < NF; ++f) {
        sn[f] = 2.0f * sn[f - 1] * cs[f - 1];
        cs[f] = fmaf(-2.0f * sn[f - 1], sn[f - 1], 1.0f);
    }
}

// DENSITY_ONLY (colors == NULL): the stream ends with the layer that holds the density row -- the coarse pass of a
// hierarchical render without gradients feeds only the re-sampling.
template <bool DENSITY_ONLY>
__global__ __launch_bounds__(WG_THREADS, 1) void fwd_bf16_kernel(Bf16Args a) {
    using G = Geo;
    constexpr auto bias_tile0 = [](int l) { return G::bias_tile0(l); };
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float* bias_lds = reinterpret_cast<float*>(smem + NBUF * CHUNK_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

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
    constexpr int DENSITY_LAYER = 9;
    // chunks of the stream this variant reads (nothing may be in flight towards LDS when the workgroup ends)
    constexpr int NCHUNK_RUN = DENSITY_ONLY ? G::chunk0(DENSITY_LAYER + 1) : G::NCHUNK;
    Ring ring{a.img, smem, 0, 16u * (unsigned)tid, wave, NCHUNK_RUN};
    ring_issue(ring, 0);   // these land while the encodings below are computed
    ring_issue(ring, 1);
    ring_issue(ring, 2);
    __builtin_amdgcn_sched_barrier(0);

    // First layer's encoding fragments in the slot order k_column() gives the weight image: lane half h owns
    // frequencies 5h..5h+4 of the position, the raw coordinates fill the tail.
    bf16x8 E0[4];
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
        vals_to_frags<32>(v, E0);
    }
    // publish chunk 0 and the bias image: own pieces of chunk 0 landed (chunks 1, 2 stay in flight),
    // own LDS writes done, then the barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER_CHUNK) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x16 acc[9];
    bf16x8 X[16];
    init_bias<8>(acc, bias_lds, bias_tile0(0), h);
    mma_layer16<8, 0, 4, 2>(ring, lane, acc, X, E0);
    to_frags<8, true>(acc, X);
#pragma unroll 1
    for (int l = 1; l <= 4; ++l) {
        init_bias<8>(acc, bias_lds, 8 * l, h);
        mma_layer16<8, 16, 0, 2>(ring, lane, acc, X, E0);
        to_frags<8, true>(acc, X);
    }
    // skip layer: [features | position encoding]
    init_bias<8>(acc, bias_lds, bias_tile0(5), h);
    mma_layer16<8, 16, 4, 2>(ring, lane, acc, X, E0);
    to_frags<8, true>(acc, X);
#pragma unroll 1
    for (int l = 6; l <= 8; ++l) {
        init_bias<8>(acc, bias_lds, 8 * l, h);
        mma_layer16<8, 16, 0, 2>(ring, lane, acc, X, E0);
        to_frags<8, true>(acc, X);
    }
    // block2.8: 256 feature rows (no activation) + the density row (first row of the ninth tile)
    init_bias<9>(acc, bias_lds, bias_tile0(9), h);
    if constexpr (DENSITY_ONLY) {
        mma_layer16<9, 16, 0, 0, 8>(ring, lane, acc, X, E0);
        if (valid && h == 0) a.density[m] = fmaxf(acc[8][0], 0.0f);
        return;
    }
    mma_layer16<9, 16, 0, 2>(ring, lane, acc, X, E0);
    const float dens = fmaxf(acc[8][0], 0.0f);
    to_frags<8, false>(acc, X);
    // rgb head: [features | direction encoding] -> 128, ReLU.  Direction encoding (nerf/encoding.py with 4 frequencies):
    // half h owns frequencies 2h..2h+1, the raw direction sits behind them in half 0.  Only computed here, in front of the
    // layer that reads it: its registers would otherwise be held through the whole trunk.
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

    if (valid && h == 0) {  // rows 0..2 of the last tile and the density row sit in lane half 0
#pragma unroll
        for (int c = 0; c < 3; ++c) a.colors[m * 3 + c] = dev::sigmoidf_(acc[0][c]);
        a.density[m] = dens;
    }
}

}  // namespace

extern "C" size_t nerf_mlp_packed_bf16_bytes(int model) { return model == NERF_MODEL_NERF ? Geo::PACKED_BYTES : 0; }

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
    for (int l = 0; l < NL_MAX; ++l) a.layers[l] = Net<NERF_MODEL_NERF>::LAYERS[l];
    for (int l = 0; l <= NL_MAX; ++l) {
        a.chunk0[l] = Geo::chunk0(l < Geo::NL ? l : Geo::NL);
        a.bias_tile0[l] = Geo::bias_tile0(l < Geo::NL ? l : Geo::NL);
    }
    const int pieces = Geo::NCHUNK * CHUNK_UNITS * 64;
    pack_bf16_kernel<<<(pieces + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_pack_bf16");
}

extern "C" int nerf_mlp_fwd_bf16(int model, const void* packed, int64_t M, const float* points, const float* dirs,
                                 const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray,
                                 double near, double far, float* colors, float* density, void* stream) {
    const char* who = "nerf_mlp_fwd_bf16";
    if (model != NERF_MODEL_NERF) return nerf_fail(NERF_E_ARG, "%s: the bf16 mode covers the NeRF model only (model %d)", who, model);
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
               (float)near, (float)(far - near), colors, density};
    const int64_t groups = ceil_div64(M, WG_SAMPLES);
    if (groups >= (1ll << 31)) return nerf_fail(NERF_E_ARG, "%s: too many samples", who);
    if (colors) {
        if (int rc = nerf_set_lds(fwd_bf16_kernel<false>, BF16_LDS_BYTES)) return rc;
        fwd_bf16_kernel<false><<<(unsigned)groups, WG_THREADS, BF16_LDS_BYTES, (hipStream_t)stream>>>(a);
    } else {
        if (int rc = nerf_set_lds(fwd_bf16_kernel<true>, BF16_LDS_BYTES)) return rc;
        fwd_bf16_kernel<true><<<(unsigned)groups, WG_THREADS, BF16_LDS_BYTES, (hipStream_t)stream>>>(a);
    }
    return nerf_check_launch(who);
}
