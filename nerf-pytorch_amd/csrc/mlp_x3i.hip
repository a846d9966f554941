// Three-term ("bf16x3") kernels on 16-sample tiles: v_mfma_f32_16x16x32_bf16, two waves per SIMD.  The inference
// forward of both models, and -- the same kernel with workspace outputs, and the same structure run over the
// transposed layers -- the forward and data-gradient chain of the opt-in bf16x3 training mode (entry points: mlp_x3.hip).
//
// Arithmetic: every fp32 operand is the exact sum of three bf16 terms, x = x0 + x1 + x2 with x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (the subtractions are exact), and a product is formed from the six term
// products that reach fp32 precision,
//     w*x ~= w0*x0 + w0*x1 + w0*x2 + w1*x0 + w1*x1 + w2*x0      (what is dropped is below 2^-24 of |w*x|),
// each exact in the MFMA's fp32 accumulator: the results carry the rounding of an fp32 accumulation and nothing else
// (nerf/models.py:62-75, :185-203 at the fp32 parity bar).
//
// Orientation (as in mlp_bf16.hip): a layer is computed transposed, D[out feature][sample] = sum_k W[out][k] X[k][sample],
// weights = A operand (streamed through an LDS ring, one 1 KiB unit = one fragment of 16 rows x 32 k; the three terms of
// a fragment are three consecutive units), activations = B operand, chained from layer to layer in registers.  Operand
// maps of the 16x16x32 MFMA, lane l = (c = l & 15, q = l >> 4):
//   A: row c, k = 8q .. 8q+7        B: column (sample) c, k = 8q .. 8q+7        D register g: row 4q + g, column c.
// Two 16-feature output tiles (2s, 2s+1) of a layer give the B fragment of the next layer's k-step s: element j of lane
// (c, q) is feature 32s + 4q + j for j < 4 and 32s + 16 + 4q + (j - 4) for j >= 4, and the weight image stores its
// fragments in that k order.
//
// Why this shape.  On 32-sample tiles (the kernel this one replaced) a wave holds 192 registers of activation terms + 144
// of accumulators: one wave per SIMD, and nothing runs on the matrix pipe while that wave splits its outputs into
// terms (11 vector instructions per value pair) or issues its LDS-DMA pieces.  Here a wave owns 16 samples (96 + 68
// registers) and two waves share a SIMD.  Every 1 KiB unit then feeds 16 instead of 32 samples per read: 128 bytes of
// LDS reads per cycle and CU at the full MFMA rate, half of what ds_read_b128 delivers.  The chip also holds a higher
// clock under this MFMA shape than under 32x32x16.
//
// Ring protocol.  The image is streamed once per workgroup through NBUF buffers of one chunk (24 units: every layer of
// both models but the last is a whole number of them) each; every thread moves its own 16-byte pieces of every chunk
// (buffer_load ... lds, no staging registers).  Four units into chunk c every wave waits for its pieces of chunk c+1
// (counted vmcnt: chunk c+2 may stay in flight), all waves meet at one barrier -- which publishes chunk c+1 and, each
// wave having read at least one unit of chunk c, frees the buffer of chunk c-1 -- and chunk c+3 is requested into that
// buffer, one piece per following unit.  The fragment reads in flight at the barrier are untouched by it.
//
// Measured and not kept (all with results bit for bit unchanged; cycles = GRBM_GUI_ACTIVE of the launch):
//   * waves 4-7 trailing waves 0-3 by one or two chunks, so that the splitting epilogues of the two waves of a SIMD do
//     not coincide: a barrier per chunk keeps the wave groups within one chunk of each other whatever the lag; 3 % slower;
//   * the split of finished tiles placed between the MFMAs of the layer's last k-step: same cycles;
//   * the plain 256 -> 256 layers tile-major (all k-steps of output tile 0, then of tile 1, ...; two accumulator tiles
//     instead of 16, a second set of fragment registers, 256 registers without spills) with the previous tile's split
//     spread over the next tile's 48 MFMAs, one or two vector instructions after every second unit: + 6-11 % cycles;
//     the same loop with those vector instructions removed (wrong results) takes the cycles of the k-step-major
//     kernel WITH its epilogues.  In tests/micro/coexec16.hip two vector instructions per MFMA issue in its shadow;
//     in this loop (a fragment read and a counted wait per unit beside the MFMAs) they are paid in full, and packed
//     fp32 ones (v_pk_add_f32: 4.7 cycles alone, a whole 16-cycle MFMA slot beside MFMAs) more than that;
//   * LDS-DMA pieces of the two waves of a SIMD at different units; request distance 4 and 5 chunks; 8 fragment reads in
//     flight; the waits for landed pieces removed (still correct: the pieces are never late).
// Where the cycles go (counters, 1,048,576 samples per launch): the matrix pipe is busy 0.76 of them.  Without the
// LDS-DMA instructions (stale weights, timing only) the kernel is 12 % faster, without the splits 7 %.
//
// Training forward (fwd_x3i_kernel<NeRF, false, TRAIN>, entered through nerf_mlp_fwd_x3): the same kernel also fills the
// activation workspace of the gradient kernels, every value from the registers that hold it at the end of its layer
// (16-byte stores, 64 contiguous bytes per sample and tile), and records the ReLU decisions as the sign bits of the
// negated accumulators.  1,048,576 samples: 5.9 ms against 4.7 ms without the workspace (5.0 ms with the stores alone
// removed) and 7.45 ms of the 64-sample-tile kernel it replaced, which kept the layer input in LDS planes, had every
// wave fetch its own weight fragments through the vector cache (twice the L2 traffic per sample) and two workgroup
// barriers per layer.  Streaming stores (nt) are 3 % slower, a request distance of five chunks changes nothing.
#include "bf16_common.h"

// Data-gradient chain of the NeRF model as a "network" of this kernel family (nerf/models.py:62-75 backwards, what
// autograd derives for train.py:181): rgb_head.0's feature part, block2.8 (256 feature rows; the density row enters as a
// rank-1 term), block2.6 .. block2.0 (the skip layer's feature part), block1.8 .. block1.2.  Input gradients of the first
// layer are not part of the chain.
namespace bf16k {
constexpr int NERF_CHAIN_DGRAD = 16;
template <> struct Net<NERF_CHAIN_DGRAD> {
    static constexpr int NL = 10;
    static constexpr LayerInfo LAYERS[NL_MAX] = {
        {8, 8, ENC_NONE, 256, 128, 283, 20, -1, 1, -1, 0},  {8, 16, ENC_NONE, 256, 256, 256, 18, -1, 1, 18, 256},
        {8, 16, ENC_NONE, 256, 256, 256, 16, -1, 1, -1, 0}, {8, 16, ENC_NONE, 256, 256, 256, 14, -1, 1, -1, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 12, -1, 1, -1, 0}, {8, 16, ENC_NONE, 256, 256, 319, 10, -1, 1, -1, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 8, -1, 1, -1, 0},  {8, 16, ENC_NONE, 256, 256, 256, 6, -1, 1, -1, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 4, -1, 1, -1, 0},  {8, 16, ENC_NONE, 256, 256, 256, 2, -1, 1, -1, 0},
        {0, 0, ENC_NONE, 0, 0, 0, 0, -1, 0, -1, 0},         {0, 0, ENC_NONE, 0, 0, 0, 0, -1, 0, -1, 0},
    };
};
// Siren (nerf/models.py:185-203 backwards): rgb_head.0's feature part, feature_remap (the density branch enters as a
// rank-1 term), block1.7 .. block1.1.
constexpr int SIREN_CHAIN_DGRAD = 17;
template <> struct Net<SIREN_CHAIN_DGRAD> {
    static constexpr int NL = 9;
    static constexpr LayerInfo LAYERS[NL_MAX] = {
        {8, 8, ENC_NONE, 256, 128, 283, 20, -1, 1, -1, 0},  {8, 16, ENC_NONE, 256, 256, 256, 18, -1, 1, 16, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 14, -1, 1, -1, 0}, {8, 16, ENC_NONE, 256, 256, 256, 12, -1, 1, -1, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 10, -1, 1, -1, 0}, {8, 16, ENC_NONE, 256, 256, 256, 8, -1, 1, -1, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 6, -1, 1, -1, 0},  {8, 16, ENC_NONE, 256, 256, 256, 4, -1, 1, -1, 0},
        {8, 16, ENC_NONE, 256, 256, 256, 2, -1, 1, -1, 0},  {0, 0, ENC_NONE, 0, 0, 0, 0, -1, 0, -1, 0},
        {0, 0, ENC_NONE, 0, 0, 0, 0, -1, 0, -1, 0},         {0, 0, ENC_NONE, 0, 0, 0, 0, -1, 0, -1, 0},
    };
};
}  // namespace bf16k

using namespace bf16k;

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int NBUF = 4;
constexpr int AHEAD = NBUF - 1;      // a chunk is requested three chunks before it is read

constexpr int WAVES = 8;
constexpr int WG_THREADS = 64 * WAVES;
constexpr int WG_SAMPLES = 16 * WAVES;
constexpr int CHUNK_UNITS = 24;
constexpr int CHUNK_BYTES = CHUNK_UNITS * UNIT_BYTES;
constexpr int PER_CHUNK = CHUNK_BYTES / (16 * WG_THREADS);   // LDS-DMA instructions per thread and chunk
static_assert(PER_CHUNK * 16 * WG_THREADS == CHUNK_BYTES, "whole pieces");

constexpr int enc_ks32(int enc) { return enc == ENC_PE ? 2 : (enc == ENC_DE || enc == ENC_XYZ) ? 1 : 0; }

template <int MODEL>
struct Geo {
    using N = Net<MODEL>;
    static constexpr int NL = N::NL;
    static constexpr int n_ot(int l) { return (N::LAYERS[l].n_out + 15) / 16; }
    static constexpr int nks_acc(int l) { return N::LAYERS[l].nks_acc / 2; }
    static constexpr int nks(int l) { return nks_acc(l) + enc_ks32(N::LAYERS[l].enc); }
    static constexpr int layer_units(int l) { return 3 * n_ot(l) * nks(l); }
    static constexpr int layer_chunks(int l) { return (layer_units(l) + CHUNK_UNITS - 1) / CHUNK_UNITS; }
    static constexpr int chunk0(int l) {
        int c = 0;
        for (int i = 0; i < l; ++i) c += layer_chunks(i);
        return c;
    }
    static constexpr int bias_tile0(int l) {
        int t = 0;
        for (int i = 0; i < l; ++i) t += n_ot(i);
        return t;
    }
    // Layers followed by a ReLU are stored NEGATED (weights and bias): the accumulators then hold -z, the activation is
    // max(0, -acc) with the negation in the instruction's source modifier, and the accumulator's sign bit IS the decision
    // z > 0 that the training forward records for the data gradient (exactly: z = +0 has the bit clear).
    static constexpr bool negated(int l) { return MODEL == NERF_MODEL_NERF && (l <= 8 || l == 10); }
    static constexpr unsigned neg_mask() {
        unsigned m = 0;
        for (int l = 0; l < NL; ++l) m |= negated(l) ? 1u << l : 0u;
        return m;
    }
    static constexpr int NCHUNK = chunk0(NL);
    static constexpr int NBIAS_TILES = bias_tile0(NL);
    static constexpr size_t BIAS_FLOATS = (size_t)NBIAS_TILES * 16;
    static constexpr size_t IMG_BYTES = (size_t)NCHUNK * CHUNK_BYTES;
    static constexpr size_t PACKED_BYTES = IMG_BYTES + BIAS_FLOATS * 4;
};
constexpr int NBIAS_TILES_MAX = 170;                       // LDS room for the bias image of either model
static_assert(Geo<NERF_MODEL_NERF>::NBIAS_TILES == 170 && Geo<NERF_MODEL_SIREN>::NBIAS_TILES <= NBIAS_TILES_MAX, "bias image");
// LDS: the 256-entry sine/cosine table of the sine activations (first 2 KiB: its reads address it through the
// instruction's offset alone), the ring, the bias image
constexpr int TAB_BYTES = dev::SINCOS_TAB_FLOATS * 4;
constexpr int LDS_BYTES = TAB_BYTES + NBUF * CHUNK_BYTES + NBIAS_TILES_MAX * 16 * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");

// Input column of weight element (k-step ks, lane group q, fragment element j) of layer L, or -1.
// Encoding slots: every lane group owns the same NUMBER of frequencies, so that all lanes run the same code:
//   position encoding (nerf/encoding.py:14-18, 10 frequencies; 2 k-steps = 16 slots per group): slots 0..11 the frequencies
//     2q and 2q+1 (sin xyz, cos xyz each), slots 12..14 sin (q even) or cos (q odd) of frequency 8 + q/2, slot 15 the raw
//     coordinate q (none for q = 3);
//   direction encoding (4 frequencies; 1 k-step = 8 slots per group): slots 0..5 frequency q, slot 6 the raw component q;
//   raw coordinates of the Siren model: slots 0..2 of group 0.
__host__ __device__ inline int k_column16(const LayerInfo& L, int ks, int q, int j) {
    const int nka = L.nks_acc / 2;
    if (ks < nka) {
        const int f = 32 * ks + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
        return f < L.n_in ? f : -1;
    }
    const int e = 8 * (ks - nka) + j;
    if (L.enc == ENC_XYZ) return (q == 0 && e < 3) ? L.n_in + e : -1;
    if (L.enc == ENC_PE) {
        if (e < 12) return L.n_in + 3 + 6 * (2 * q + e / 6) + e % 6;
        if (e < 15) return L.n_in + 3 + 6 * (8 + (q >> 1)) + 3 * (q & 1) + (e - 12);
        return q < 3 ? L.n_in + q : -1;
    }
    if (e < 6) return L.n_in + 3 + 6 * q + e;
    if (e == 6) return q < 3 ? L.n_in + q : -1;
    return -1;
}

struct PackArgs {
    const float* src[NERF_NUM_PARAM_TENSORS];
    unsigned char* dst;
    LayerInfo layers[NL_MAX];
    int chunk0[NL_MAX + 1];
    int bias_tile0[NL_MAX + 1];
    unsigned neg_mask;   // bit l: layer l is stored negated
};

template <int MODEL>
__global__ __launch_bounds__(256) void pack_x3i_kernel(PackArgs a) {
    using G = Geo<MODEL>;
    const int p = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece: unit, lane
    if (p < G::NCHUNK * CHUNK_UNITS * 64) {
        const int chunk = p / (CHUNK_UNITS * 64), unit = (p / 64) % CHUNK_UNITS, lane = p & 63;
        int l = 0;
        while (chunk >= a.chunk0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int n_ot = (L.n_out + 15) / 16, nks = L.nks_acc / 2 + enc_ks32(L.enc);
        const int up = (chunk - a.chunk0[l]) * CHUNK_UNITS + unit;
        const int plane = up % 3, u = up / 3;       // the three terms of a fragment are consecutive units
        const bool neg = (a.neg_mask >> l) & 1u;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.0f;
        if (u < n_ot * nks) {
            const int ks = u / n_ot, ot = u % n_ot;
            const int out = 16 * ot + (lane & 15), q = lane >> 4;
            if (out < L.n_out) {
                const float* __restrict__ w = (L.last_row_tensor >= 0 && out == L.n_out - 1)
                                                  ? a.src[L.last_row_tensor] : a.src[L.wtensor] + (size_t)out * L.ld;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = k_column16(L, ks, q, j);
                    if (col >= 0) {
                        const float e = L.transposed ? a.src[L.wtensor][(size_t)col * L.ld + out] : w[col];
                        v[j] = bf16_term(neg ? -e : e, plane);
                    }
                }
            }
        }
        *reinterpret_cast<bf16x8*>(a.dst + (size_t)p * 16) = v;
    }
    // bias image: 16 floats per output tile
    if (p < G::NBIAS_TILES * 16) {
        const int tile = p >> 4, e = p & 15;
        int l = 0;
        while (tile >= a.bias_tile0[l + 1]) ++l;
        const LayerInfo L = a.layers[l];
        const int out = 16 * (tile - a.bias_tile0[l]) + e;
        float b = 0.0f;
        if (L.transposed) {
            if (out < L.n_out && L.rank1_tensor >= 0) b = a.src[L.rank1_tensor][(size_t)L.rank1_row * L.ld + out];
        } else if (out < L.n_out) {
            b = (L.last_row_tensor >= 0 && out == L.n_out - 1) ? a.src[L.last_row_tensor + 1][0] : a.src[L.wtensor + 1][out];
        }
        reinterpret_cast<float*>(a.dst + G::IMG_BYTES)[p] = ((a.neg_mask >> l) & 1u) ? -b : b;
    }
}

// ---- forward -----------------------------------------------------------------------------
struct FwdArgs {
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
    float* ws;    // training forward: the activation workspace (mlp_layout.h), rows of the padded sample count Mp
    size_t Mp;
};

struct Ring {
    const unsigned char* img;
    unsigned char* lds;
    int next;        // chunk the next buffer switch hands out
    unsigned voff;   // this thread's byte offset inside a chunk: its wave's slice, 16 bytes per lane
    int wave;
    int nchunk, img_bytes;
};

// piece i (16 bytes per thread) of a chunk: global memory -> LDS without staging registers.  Buffer form of the LDS-DMA:
// image base and size in a scalar resource, the chunk in the scalar offset, the thread's place (its wave's slice of
// PER_CHUNK KiB, 16 bytes per lane) in the one vector offset, the piece in the instruction's immediate offset -- which
// the hardware adds to the LDS address as well, so that the LDS base (M0) changes once per chunk, not once per piece.
__device__ __forceinline__ void ring_issue_piece(const Ring& r, int chunk, int i) {
    if (chunk < r.nchunk) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(r.img), 0, r.img_bytes, 0x00020000);
        unsigned char* dst = r.lds + (chunk % NBUF) * CHUNK_BYTES + PER_CHUNK * 1024 * r.wave;
        auto lds = (__attribute__((address_space(3))) void*)dst;
        // (the immediate must be a literal for the builtin)
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, (int)r.voff, chunk * CHUNK_BYTES, 0, 0);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, (int)r.voff, chunk * CHUNK_BYTES, 1024, 0);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, (int)r.voff, chunk * CHUNK_BYTES, 2048, 0);
    }
}
static_assert(PER_CHUNK == 3, "ring_issue_piece spells out its immediates");
__device__ __forceinline__ void ring_issue(const Ring& r, int chunk) {
#pragma unroll
    for (int i = 0; i < PER_CHUNK; ++i) ring_issue_piece(r, chunk, i);
}

// publication of the next chunk (header comment).  REM = chunks of the stream that follow the one being read: the
// requests of up to AHEAD - 2 of them, behind the next one, may stay in flight.
// YOUNGER = vector-memory instructions other than the next-but-one chunk's pieces that were issued after the pieces
// waited for (the training forward's workspace stores; the counter is shared and retires in issue order): they may stay in
// flight too.  A value below the true number only waits longer; one above it would publish pieces that have not landed.
template <int REM, int YOUNGER = 0>
__device__ __forceinline__ void ring_publish_next() {
    if constexpr (REM >= 1) {
        constexpr int IN_FLIGHT = ((REM < AHEAD - 1 ? REM : AHEAD - 1) - 1) * PER_CHUNK;   // requested chunks behind the next one
        static_assert(IN_FLIGHT + YOUNGER < 64, "vmcnt is a 6-bit field");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_FLIGHT + YOUNGER) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

constexpr int NOT_MAX = 17;

// The 3, 2 or 1 MFMAs of a unit (term p of the weights meets terms 0 .. 2-p of the activations), accumulating IN PLACE.
// Through the builtin the compiler treats the result as a new value and, accumulator tiles and A fragments both being
// four registers wide, rotates them through each other: results land in the registers of the fragment just consumed
// and the next fragment is loaded over the old accumulator, every such load behind a software wait of 5-7 cycles for the
// MFMA's read of its C operand -- and with two waves per SIMD (the partner's MFMAs delay this wave's) those waits were
// not long enough: wrong sums.  The asm form pins destination = C operand; what the compiler no longer sees is the
// MFMA's result latency, which settle() covers.
template <int N>
__device__ __forceinline__ void mfma16_unit(f32x4& acc, const u32x4& a, const u32x4& b0, const u32x4& b1, const u32x4& b2) {
    if constexpr (N == 3)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0"
                     : "+v"(acc) : "v"(a), "v"(b0), "v"(b1), "v"(b2));
    else if constexpr (N == 2)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %3, %0" : "+v"(acc) : "v"(a), "v"(b0), "v"(b1));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b0));
}
// Wait states between the last MFMAs of a layer and the first vector instruction that reads their accumulators (once
// per layer; the hardware does not interlock this).  The tiles are operands so that the compiler cannot move a read of
// them in front of the wait; tiles finished earlier in the layer are dozens of MFMAs old.
__device__ __forceinline__ void settle(f32x4& last, f32x4& before_last) {
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(last), "+v"(before_last));
}
__device__ __forceinline__ void settle(f32x4& last) { asm volatile("s_nop 15\n\ts_nop 7" : "+v"(last)); }

template <int NOT>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[NOT_MAX], const float* bias_lds, int tile0, int q) {
    const float* base = bias_lds + tile0 * 16 + q * 4;
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) acc[ot] = *reinterpret_cast<const f32x4*>(base + ot * 16);
}

// acc[ot] += W tile (ot, ks) * B(ks) for all ks, consuming the layer's units in image order; k-steps below NKS_A take
// their B fragment from X, the rest from XE (an encoding).  A fragments are read from LDS PF units ahead of the MFMAs
// that use them.  Unit 3g+p is term p of fragment g.
// TAIL = chunks that follow this layer's last chunk in the stream the kernel reads (AHEAD = that many or more).
// FIRST_OT: output tiles below it are not computed (their units still pass through the ring).
// NST: workspace stores issued between the previous layer's last unit and this layer's first one.  They are younger
// than the pieces of this layer's second .. AHEAD-th chunk, whose publications therefore let NST more instructions stay
// in flight (ring_publish_next).
template <int NOT, int NKS_A, int NKS_E, int TAIL, int FIRST_OT = 0, int NST = 0>
__device__ __forceinline__ void mma_layer(Ring& r, int lane, f32x4 (&acc)[NOT_MAX], const u32x4 (&X)[3][8], const u32x4 (&XE)[3][2]) {
    constexpr int CU = CHUNK_UNITS;
    constexpr int NKS = NKS_A + NKS_E, UNITS = 3 * NOT * NKS, CHUNKS = (UNITS + CU - 1) / CU;
    constexpr int PF = 4;            // fragment reads in flight
    constexpr int PUBLISH_AT = PF;   // unit of a chunk at whose fetch the next chunk is published
    static_assert(UNITS >= PF && (UNITS % CU == 0 || UNITS % CU >= PUBLISH_AT + PER_CHUNK), "chunk too short");
    static_assert(PUBLISH_AT + PER_CHUNK <= CU, "pieces of a chunk are requested inside one chunk");
    const unsigned char* buf = nullptr;
    u32x4 a[PF];
    auto fetch = [&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (u % CU == 0) {
            // one address register per chunk (its offset hidden from the compiler), the units at immediate offsets
            unsigned off = (unsigned)(r.next % NBUF) * CHUNK_BYTES + 16u * (unsigned)lane;
            asm volatile("" : "+v"(off));
            ++r.next;
            buf = r.lds + off;
        }
        constexpr int rem = (CHUNKS - 1 - u / CU) + TAIL;
        if constexpr (u % CU == PUBLISH_AT) ring_publish_next<rem, (u / CU < AHEAD - 1 ? NST : 0)>();
        if constexpr (rem >= 1 && u % CU >= PUBLISH_AT && u % CU < PUBLISH_AT + PER_CHUNK)
            ring_issue_piece(r, r.next - 1 + AHEAD, u % CU - PUBLISH_AT);   // r.next - 1 is the chunk being read
        a[u % PF] = *reinterpret_cast<const u32x4*>(buf + (u % CU) * UNIT_BYTES);
    };
    static_for<PF>(fetch);
    static_for<UNITS>([&](auto U) {
        constexpr int u = decltype(U)::value, p = u % 3, ks = (u / 3) / NOT, ot = (u / 3) % NOT;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ot >= FIRST_OT) {
            if constexpr (ks < NKS_A) mfma16_unit<3 - p>(acc[ot], a[u % PF], X[0][ks], X[1][ks], X[2][ks]);
            else mfma16_unit<3 - p>(acc[ot], a[u % PF], XE[0][ks - NKS_A], XE[1][ks - NKS_A], XE[2][ks - NKS_A]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u + PF < UNITS) fetch(std::integral_constant<int, u + PF>{});
    });
    if constexpr (NOT >= 2) settle(acc[NOT - 1], acc[NOT - 2]);
    else settle(acc[0]);
}

// accumulator tiles -> B fragments of the next layer (one k-step per PAIR of 16-feature tiles), through the layer's
// activation: FR_SIN is sin(z) of a hidden sine layer (nerf/models.py:89-90 with w0 = 1), FR_SIN30 the first sine layer's
// sin(30 z).  The sines take the table form of the fp32 kernels' sine epilogues (csrc/sincos.h: 6.2e-8 absolute for
// |x| <= 1e5; pre-activations of sine layers fed by sines are bounded by sum|w| + |b|, first-layer ones by 30 times
// that with coordinates in [-1, 1]): 8 vector instructions and a table read per value where the polynomial form has
// 25 -- the epilogues run while no MFMA does, so every one of them is exposed.
// Data gradient: FR_MASK multiplies by the ReLU decisions in mw, FR_DERIV by the sine layer's derivative tile, read from
// rsrc2 at the places of the values themselves.
enum FragAct { FR_NONE = 0, FR_RELU = 1, FR_SIN = 2, FR_SIN30 = 3, FR_MASK = 4, FR_DERIV = 5 };
// OUT (training forward): FO_STORE also writes the activated tiles as fp32 to the workspace rows of this workgroup (rsrc;
// voff = this lane's row and its 16-byte column group; tile ot at byte 64 ot of the row: 16 bytes per lane and tile, 64
// contiguous bytes per sample); FO_MASK also collects the decisions z > 0 of a ReLU layer in mw: tile ot, register g at bit
// 31 - (4 (ot & 7) + g) of mw[ot >> 3]; FO_DERIV also writes a sine layer's derivative w0 cos(w0 z) to rsrc2, value by
// value where the activation goes in rsrc.
enum FragOut { FO_NONE = 0, FO_STORE = 1, FO_MASK = 2, FO_DERIV = 3 };
// FR_DERIV: the derivative tiles of a layer travel through four slots of two tiles (one tile pair each).  The first
// DERIV_AHEAD pairs are requested before the layer's MFMA loop (deriv_prefetch), the next two at the start of the
// epilogue, pair s + 4 when pair s has been used.
constexpr int DERIV_AHEAD = 2;
struct DerivTiles {
    f32x4 t[4][2];
};
__device__ __forceinline__ void deriv_load(DerivTiles& d, int s, __amdgpu_buffer_rsrc_t rsrc2, int voff) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
        d.t[s & 3][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc2, voff + (2 * s + t) * 64, 0, 0));
}
__device__ __forceinline__ void deriv_prefetch(DerivTiles& d, __amdgpu_buffer_rsrc_t rsrc2, int voff) {
#pragma unroll
    for (int s = 0; s < DERIV_AHEAD; ++s) deriv_load(d, s, rsrc2, voff);
}
template <int NPAIR, int ACT, int OUT>
__device__ __forceinline__ void to_frags_out(const f32x4 (&acc)[NOT_MAX], u32x4 (&X)[3][8], __amdgpu_buffer_rsrc_t rsrc, int voff,
                                             unsigned (&mw)[2], __amdgpu_buffer_rsrc_t rsrc2, DerivTiles& dv) {
    static_assert(OUT != FO_MASK || ACT == FR_RELU, "mask bits are the signs of negated ReLU pre-activations");
    static_assert(OUT != FO_DERIV || ACT == FR_SIN || ACT == FR_SIN30, "derivative of a sine layer");
    if constexpr (ACT == FR_DERIV) {
#pragma unroll
        for (int s = DERIV_AHEAD; s < 4 && s < NPAIR; ++s) deriv_load(dv, s, rsrc2, voff);
    }
#pragma unroll
    for (int s = 0; s < NPAIR; ++s) {
        f32x2 v[4];
        [[maybe_unused]] f32x2 cs[4];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) v[jp] = f32x2{acc[2 * s + (jp >> 1)][2 * (jp & 1)], acc[2 * s + (jp >> 1)][2 * (jp & 1) + 1]};
        if constexpr (ACT == FR_SIN || ACT == FR_SIN30) {
            dev::SinCosPair pr[4];   // the table reads of the four pairs in flight together
#pragma unroll
            for (int jp = 0; jp < 4; ++jp)
                dev::sincos_tab_issue(ACT == FR_SIN30 ? f32x2{mul_rn(30.0f, v[jp][0]), mul_rn(30.0f, v[jp][1])} : v[jp], pr[jp]);
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                dev::sincos_tab_finish(pr[jp], v[jp], cs[jp]);
                if constexpr (ACT == FR_SIN30) cs[jp] = f32x2{30.0f * cs[jp][0], 30.0f * cs[jp][1]};
            }
        }
        float h[8];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            float v0 = v[jp][0], v1 = v[jp][1];
            if constexpr (ACT == FR_RELU) {
                if constexpr (OUT == FO_MASK) {
                    mw[s >> 2] = __builtin_amdgcn_alignbit(mw[s >> 2], __builtin_bit_cast(unsigned, v0), 31);
                    mw[s >> 2] = __builtin_amdgcn_alignbit(mw[s >> 2], __builtin_bit_cast(unsigned, v1), 31);
                }
                // the layer's image is negated (Geo::negated): relu(z) = max(0, -acc)
                asm("v_max_f32_e64 %0, 0, -%1" : "=v"(v0) : "v"(v0));
                asm("v_max_f32_e64 %0, 0, -%1" : "=v"(v1) : "v"(v1));
            }
            if constexpr (ACT == FR_MASK) {   // the decisions as the forward recorded them (FO_MASK): 0 or all ones
                const int idx = 4 * ((2 * s + (jp >> 1)) & 7) + 2 * (jp & 1);
                const int k0 = __builtin_amdgcn_sbfe((int)mw[s >> 2], 31 - idx, 1), k1 = __builtin_amdgcn_sbfe((int)mw[s >> 2], 30 - idx, 1);
                v0 = __builtin_bit_cast(float, __builtin_bit_cast(int, v0) & k0);
                v1 = __builtin_bit_cast(float, __builtin_bit_cast(int, v1) & k1);
            }
            if constexpr (ACT == FR_DERIV) {
                v0 *= dv.t[s & 3][jp >> 1][2 * (jp & 1)];
                v1 *= dv.t[s & 3][jp >> 1][2 * (jp & 1) + 1];
            }
            h[2 * jp] = v0;
            h[2 * jp + 1] = v1;
            unsigned tm[3];
            split2<3>(v0, v1, tm);
#pragma unroll
            for (int p = 0; p < 3; ++p) X[p][s][jp] = tm[p];
        }
        if constexpr (OUT != FO_NONE) {
            // constant scalar offset 0: with it the compiler keeps the next writer of the data registers one slot away
            // (a store of this width reads them a cycle late); the tile's place in the row becomes the immediate offset
#pragma unroll
            for (int t = 0; t < 2; ++t)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{h[4 * t], h[4 * t + 1], h[4 * t + 2], h[4 * t + 3]}),
                                                       rsrc, voff + (2 * s + t) * 64, 0, 0);
        }
        if constexpr (OUT == FO_DERIV) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{cs[2 * t][0], cs[2 * t][1], cs[2 * t + 1][0], cs[2 * t + 1][1]}),
                                                       rsrc2, voff + (2 * s + t) * 64, 0, 0);
        }
        if constexpr (ACT == FR_DERIV)
            if (s + 4 < NPAIR) deriv_load(dv, s + 4, rsrc2, voff);
    }
}
template <int NPAIR, int ACT, int OUT>
__device__ __forceinline__ void to_frags_out(const f32x4 (&acc)[NOT_MAX], u32x4 (&X)[3][8], __amdgpu_buffer_rsrc_t rsrc, int voff,
                                             unsigned (&mw)[2], __amdgpu_buffer_rsrc_t rsrc2) {
    static_assert(ACT != FR_DERIV, "the data gradient passes its derivative tiles");
    DerivTiles none;
    to_frags_out<NPAIR, ACT, OUT>(acc, X, rsrc, voff, mw, rsrc2, none);
}
template <int NPAIR, int ACT, int OUT>
__device__ __forceinline__ void to_frags_out(const f32x4 (&acc)[NOT_MAX], u32x4 (&X)[3][8], __amdgpu_buffer_rsrc_t rsrc, int voff,
                                             unsigned (&mw)[2]) {
    to_frags_out<NPAIR, ACT, OUT>(acc, X, rsrc, voff, mw, rsrc);
}
template <int NPAIR, int ACT>
__device__ __forceinline__ void to_frags(const f32x4 (&acc)[NOT_MAX], u32x4 (&X)[3][8]) {
    unsigned none[2] = {0u, 0u};
    to_frags_out<NPAIR, ACT, FO_NONE>(acc, X, __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0), 0, none);
}

template <int N, int NKS>
__device__ __forceinline__ void vals_to_frags(const float (&v)[N], u32x4 (&X)[3][NKS]) {
    static_assert(N % 8 == 0 && N / 8 <= NKS, "whole k-steps");
#pragma unroll
    for (int ks = 0; ks < N / 8; ++ks) {
        u32x4 w[3];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            unsigned tm[3];
            split2<3>(v[8 * ks + 2 * jp], v[8 * ks + 2 * jp + 1], tm);
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p][jp] = tm[p];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) X[p][ks] = w[p];
    }
}

// DENSITY_ONLY (colors == NULL): the stream ends with the layer that holds the density row -- the coarse pass of a
// hierarchical render without gradients feeds only the re-sampling.
// TRAIN: also fills the activation workspace that the data-gradient and weight-gradient kernels of the bf16x3 training
// mode read (mlp_layout.h; NeRF: pe, de, h0..h8, f, g, the density pre-activation, the ReLU decisions of h0..h8; Siren: x,
// de, h0..h7 and their derivatives w0 cos(w0 z), f, g and its derivative, the density pre-activation), every value
// straight from the registers that hold it.  ReLU decisions: one 8-byte word per lane and layer -- sample c of the
// wave's 16, lane group q: bit 31 - (4 (ot & 7) + g) of word ot >> 3 is the decision of feature 16 ot + 4 q + g -- at
// byte 32 m + 8 q of the layer's mask field (32 bytes per sample m).
template <int MODEL, bool DENSITY_ONLY, bool TRAIN = false>
__global__ __launch_bounds__(WG_THREADS, 2) void fwd_x3i_kernel(FwdArgs a) {
    static_assert(!TRAIN || !DENSITY_ONLY, "training forward: all outputs");
    using G = Geo<MODEL>;
    constexpr auto bias_tile0 = [](int l) { return G::bias_tile0(l); };
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* ring_lds = smem + TAB_BYTES;
    float* bias_lds = reinterpret_cast<float*>(ring_lds + NBUF * CHUNK_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;

    // Ordinary loads first (inputs, bias image -> LDS): once LDS-DMA loads are in flight the compiler
    // waits for ALL of them at the first use of an ordinary load's result.
    const float* bias_g = reinterpret_cast<const float*>(a.img + G::IMG_BYTES);
    for (int i = tid; i < (int)G::BIAS_FLOATS; i += WG_THREADS) bias_lds[i] = bias_g[i];
    if constexpr (MODEL == NERF_MODEL_SIREN)
        if (tid < NERF_SINCOS_TABLE_ENTRIES) dev::load_sincos_table(reinterpret_cast<float*>(smem), tid);

    // all four lane groups load the sample of lane & 15
    const size_t m = (size_t)blockIdx.x * WG_SAMPLES + wave * 16 + (lane & 15);
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
    // workspace rows of this workgroup: a field of `width` floats per sample as a buffer resource that ends with the
    // padded sample count (stores to rows behind it are dropped by the bounds check), this lane's row in row_local
    const size_t m0g = (size_t)blockIdx.x * WG_SAMPLES;
    const int row_local = wave * 16 + (lane & 15);
    [[maybe_unused]] const auto field = [&](size_t off, int width) {
        const size_t rows = a.Mp - m0g < (size_t)WG_SAMPLES ? a.Mp - m0g : (size_t)WG_SAMPLES;
        return __builtin_amdgcn_make_buffer_rsrc(a.ws + off * a.Mp + m0g * width, 0, (int)rows * width * 4, 0x00020000);
    };
    [[maybe_unused]] const auto put = [](float v, __amdgpu_buffer_rsrc_t rsrc, int voff) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, 0, 0);
    };
    constexpr int DENSITY_LAYER = MODEL == NERF_MODEL_NERF ? 9 : 8;
    // chunks of the stream this variant reads (nothing may be in flight towards LDS when the workgroup ends)
    constexpr int NCHUNK_RUN = DENSITY_ONLY ? G::chunk0(DENSITY_LAYER + 1) : G::NCHUNK;
    // chunks behind layer l's last one, as mma_layer's TAIL
    constexpr auto tail = [](int l) { return NCHUNK_RUN - G::chunk0(l + 1) < AHEAD ? NCHUNK_RUN - G::chunk0(l + 1) : AHEAD; };
    Ring ring{a.img, ring_lds, 0, (unsigned)(PER_CHUNK * 1024 * wave + 16 * lane), wave, NCHUNK_RUN, (int)G::IMG_BYTES};
#pragma unroll
    for (int c = 0; c < AHEAD; ++c) ring_issue(ring, c);   // these land while the encodings below are computed
    __builtin_amdgcn_sched_barrier(0);

    // First layer's encoding fragments in the slot order k_column16() gives the weight image.
    u32x4 E0[3][2];
    if constexpr (MODEL == NERF_MODEL_NERF) {
        float v[16];
        const float sa = (float)(1 << (2 * q)), sc = (float)(256 << (q >> 1));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            nerf_sincosf(mul_rn(sa, x[c]), &sn, &cs);
            v[c] = sn;
            v[3 + c] = cs;
            nerf_sincosf(mul_rn(2.0f * sa, x[c]), &sn, &cs);
            v[6 + c] = sn;
            v[9 + c] = cs;
            nerf_sincosf(mul_rn(sc, x[c]), &sn, &cs);
            v[12 + c] = (q & 1) ? cs : sn;
        }
        v[15] = q == 0 ? x[0] : q == 1 ? x[1] : q == 2 ? x[2] : 0.0f;
        vals_to_frags<16, 2>(v, E0);
        if constexpr (TRAIN) {
            // pe in the column order of nerf/encoding.py:14-18 (x, then sin xyz, cos xyz per frequency), column 63 zero
            const __amdgpu_buffer_rsrc_t pe = field(lay::nerf::WS_PE, lay::PE_P);
            const int row = row_local * lay::PE_P * 4;
#pragma unroll
            for (int e = 0; e < 12; ++e) put(v[e], pe, row + 48 * q + (3 + e) * 4);                       // frequencies 2q, 2q+1
#pragma unroll
            for (int c = 0; c < 3; ++c) put(v[12 + c], pe, row + (51 + 6 * (q >> 1) + 3 * (q & 1)) * 4 + c * 4);   // 8 + q/2
            put(v[15], pe, row + (q < 3 ? q : 63) * 4);
        }
    } else {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = q == 0 ? x[c] : 0.0f;
        vals_to_frags<8, 2>(v, E0);
        if constexpr (TRAIN)
            if (q == 0)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{x[0], x[1], x[2], 0.0f}), field(lay::siren::WS_X, 4),
                                                       row_local * 16, 0, 0);
    }
    // publish chunk 0 and the bias image: own pieces of chunk 0 landed (chunks 1, 2 stay in flight),
    // own LDS writes done, then the barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * PER_CHUNK) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x4 acc[NOT_MAX];
    u32x4 X[3][8];
    // direction encoding (both models: nerf/encoding.py with 4 frequencies): lane group q owns frequency q and the raw
    // component q.  Only computed in front of the layer that reads it.
    auto direction_frags = [&](u32x4 (&DE)[3][2]) {
        float v[8];
        const float s0 = (float)(1 << q);
#pragma unroll
        for (int c = 0; c < 3; ++c) nerf_sincosf(mul_rn(s0, d[c]), &v[c], &v[3 + c]);
        v[6] = q == 0 ? d[0] : q == 1 ? d[1] : q == 2 ? d[2] : 0.0f;
        v[7] = 0.0f;
        vals_to_frags<8, 2>(v, DE);
        if constexpr (TRAIN) {
            // de in the reference's column order; columns 27..31 (padding of the 32-wide field) zero
            const __amdgpu_buffer_rsrc_t de = field(MODEL == NERF_MODEL_NERF ? lay::nerf::WS_DE : lay::siren::WS_DE, lay::DE_P);
            const int row = row_local * lay::DE_P * 4;
#pragma unroll
            for (int e = 0; e < 6; ++e) put(v[e], de, row + 24 * q + (3 + e) * 4);
            put(v[6], de, row + (q < 3 ? q : 27) * 4);
            put(0.0f, de, row + (28 + q) * 4);
        }
    };
    constexpr int N_PE_STORES = TRAIN ? 16 : 0, N_DE_STORES = TRAIN ? 8 : 0, N_H_STORES = TRAIN ? 17 : 0;
    // a 256-wide ReLU layer's epilogue: fragments of the next layer, and in training the layer's workspace row and masks
    [[maybe_unused]] const auto relu_layer_out = [&](int l) {
        if constexpr (TRAIN) {
            unsigned mw[2] = {0u, 0u};
            to_frags_out<8, FR_RELU, FO_MASK>(acc, X, field(lay::nerf::WS_H + (size_t)l * lay::H, lay::H), row_local * lay::H * 4 + q * 16, mw);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{mw[0], mw[1]}, field(lay::nerf::WS_MASK + 8 * (size_t)l, 8), row_local * 32 + q * 8, 0, 0);
        } else {
            to_frags<8, FR_RELU>(acc, X);
        }
    };
    float dens;

    if constexpr (MODEL == NERF_MODEL_NERF) {
        init_bias<16>(acc, bias_lds, bias_tile0(0), q);
        mma_layer<16, 0, 2, tail(0), 0, N_PE_STORES>(ring, lane, acc, X, E0);
        relu_layer_out(0);
#pragma unroll 1
        for (int l = 1; l <= 4; ++l) {
            init_bias<16>(acc, bias_lds, 16 * l, q);
            mma_layer<16, 8, 0, tail(4), 0, N_H_STORES>(ring, lane, acc, X, E0);
            relu_layer_out(l);
        }
        // skip layer: [features | position encoding]
        init_bias<16>(acc, bias_lds, bias_tile0(5), q);
        mma_layer<16, 8, 2, tail(5), 0, N_H_STORES>(ring, lane, acc, X, E0);
        relu_layer_out(5);
#pragma unroll 1
        for (int l = 6; l <= 8; ++l) {
            init_bias<16>(acc, bias_lds, 16 * l, q);
            mma_layer<16, 8, 0, tail(8), 0, N_H_STORES>(ring, lane, acc, X, E0);
            relu_layer_out(l);
        }
        // block2.8: 256 feature rows (no activation) + the density row (first row of the 17th tile)
        init_bias<17>(acc, bias_lds, bias_tile0(9), q);
        if constexpr (DENSITY_ONLY) {
            mma_layer<17, 8, 0, 0, 16>(ring, lane, acc, X, E0);
            if (valid && q == 0) a.density[m] = fmaxf(acc[16][0], 0.0f);
            return;
        }
        mma_layer<17, 8, 0, tail(9), 0, N_H_STORES>(ring, lane, acc, X, E0);
        dens = fmaxf(acc[16][0], 0.0f);
        if constexpr (TRAIN) {
            if (q == 0) put(acc[16][0], field(lay::nerf::WS_SRAW, 1), row_local * 4);
            unsigned none[2];
            to_frags_out<8, FR_NONE, FO_STORE>(acc, X, field(lay::nerf::WS_F, lay::H), row_local * lay::H * 4 + q * 16, none);
        } else {
            to_frags<8, FR_NONE>(acc, X);
        }
        // rgb head: [features | direction encoding] -> 128, ReLU
        u32x4 DE[3][2];
        direction_frags(DE);
        init_bias<8>(acc, bias_lds, bias_tile0(10), q);
        mma_layer<8, 8, 1, tail(10), 0, (TRAIN ? 17 + N_DE_STORES : 0)>(ring, lane, acc, X, DE);
        if constexpr (TRAIN) {
            unsigned none[2];
            to_frags_out<4, FR_RELU, FO_STORE>(acc, X, field(lay::nerf::WS_G, lay::HH), row_local * lay::HH * 4 + q * 16, none);
        } else {
            to_frags<4, FR_RELU>(acc, X);
        }
        init_bias<1>(acc, bias_lds, bias_tile0(11), q);
        mma_layer<1, 4, 0, 0>(ring, lane, acc, X, DE);
    } else {
        // nerf/models.py:185-203: sin(30 (W0 x + b0)), seven sin(W h + b) layers, then feature_remap + density_branch
        // as one 17-tile layer, the 283 -> 128 sine layer of the rgb head and its 128 -> 3 output layer
        constexpr int N_SIN_STORES = TRAIN ? 32 : 0;   // activation and derivative
        [[maybe_unused]] unsigned none[2];
        [[maybe_unused]] const int voff_h = row_local * lay::H * 4 + q * 16;
        init_bias<16>(acc, bias_lds, bias_tile0(0), q);
        mma_layer<16, 0, 1, tail(0), 0, (TRAIN ? 1 : 0)>(ring, lane, acc, X, E0);
        if constexpr (TRAIN) to_frags_out<8, FR_SIN30, FO_DERIV>(acc, X, field(lay::siren::WS_H, lay::H), voff_h, none, field(lay::siren::WS_C, lay::H));
        else to_frags<8, FR_SIN30>(acc, X);
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            init_bias<16>(acc, bias_lds, 16 * l, q);
            mma_layer<16, 8, 0, tail(7), 0, N_SIN_STORES>(ring, lane, acc, X, E0);
            if constexpr (TRAIN)
                to_frags_out<8, FR_SIN, FO_DERIV>(acc, X, field(lay::siren::WS_H + (size_t)l * lay::H, lay::H), voff_h, none,
                                                  field(lay::siren::WS_C + (size_t)l * lay::H, lay::H));
            else to_frags<8, FR_SIN>(acc, X);
        }
        init_bias<17>(acc, bias_lds, bias_tile0(8), q);
        if constexpr (DENSITY_ONLY) {
            mma_layer<17, 8, 0, 0, 16>(ring, lane, acc, X, E0);
            if (valid && q == 0) a.density[m] = fmaxf(acc[16][0], 0.0f) * 10.0f;
            return;
        }
        mma_layer<17, 8, 0, tail(8), 0, N_SIN_STORES>(ring, lane, acc, X, E0);
        dens = fmaxf(acc[16][0], 0.0f) * 10.0f;           // relu(sigma) * sigma_mul (models.py:193)
        if constexpr (TRAIN) {
            if (q == 0) put(acc[16][0], field(lay::siren::WS_SRAW, 1), row_local * 4);
            to_frags_out<8, FR_NONE, FO_STORE>(acc, X, field(lay::siren::WS_F, lay::H), voff_h, none);
        } else {
            to_frags<8, FR_NONE>(acc, X);
        }
        u32x4 DE[3][2];
        direction_frags(DE);
        init_bias<8>(acc, bias_lds, bias_tile0(9), q);
        mma_layer<8, 8, 1, tail(9), 0, (TRAIN ? 17 + N_DE_STORES : 0)>(ring, lane, acc, X, DE);
        if constexpr (TRAIN)
            to_frags_out<4, FR_SIN, FO_DERIV>(acc, X, field(lay::siren::WS_G, lay::HH), row_local * lay::HH * 4 + q * 16, none,
                                              field(lay::siren::WS_CG, lay::HH));
        else to_frags<4, FR_SIN>(acc, X);
        init_bias<1>(acc, bias_lds, bias_tile0(10), q);
        mma_layer<1, 4, 0, 0>(ring, lane, acc, X, DE);
    }

    if (valid && q == 0) {  // rows 0..2 of the last tile and the density row sit in lane group 0
#pragma unroll
        for (int c = 0; c < 3; ++c) a.colors[m * 3 + c] = dev::sigmoidf_(acc[0][c]);
        a.density[m] = dens;
    }
}


// ---- data gradient -----------------------------------------------------------------------------------
struct DgradArgs {
    const unsigned char* img;   // ring image of the model's data-gradient chain
    const float* wout;          // the output layer's weight [3][128], fp32
    int64_t M;
    size_t Mp;
    const float* dcolors;
    const float* ddensity;
    const float* colors;
    const float* ws;            // the forward's workspace
    float* bw;                  // layer gradients (mlp_layout.h)
};

// The chain d(rgb hidden) -> d(features) -> d(last hidden layer) -> ... -> dz0 on the forward kernel's structure: dZ^T of a
// layer is the B operand, chained in registers, the transposed weights stream through the ring.  Every layer gradient
// goes to the backward workspace from the registers, as the forward's activations do.  NeRF: the ReLU decisions are the
// words the forward stored for exactly this lane (same samples, same tile and register order), all nine fetched before
// the first chunk is requested.  Siren: the derivative tiles are read in the epilogues, where the forward wrote them.
template <int MODEL>
__global__ __launch_bounds__(WG_THREADS, 2) void dgrad_x3i_kernel(DgradArgs a) {
    constexpr bool IS_NERF = MODEL == NERF_MODEL_NERF;
    using G = Geo<IS_NERF ? NERF_CHAIN_DGRAD : SIREN_CHAIN_DGRAD>;
    using namespace lay;
    constexpr size_t WS_G = IS_NERF ? nerf::WS_G : siren::WS_G, WS_SRAW = IS_NERF ? nerf::WS_SRAW : siren::WS_SRAW;
    constexpr size_t BW_DZ = 0, BW_DZF = IS_NERF ? nerf::BW_DZ9 : siren::BW_DZF, BW_DZR = IS_NERF ? nerf::BW_DZ10 : siren::BW_DZR;
    constexpr size_t BW_DZO = IS_NERF ? nerf::BW_DZ11 : siren::BW_DZO, BW_DSIG = IS_NERF ? nerf::BW_DSIG : siren::BW_DSIG;
    constexpr int LAST_HIDDEN = IS_NERF ? 8 : 7;            // the layer whose output feeds the features and the density
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* ring_lds = smem + TAB_BYTES;
    float* bias_lds = reinterpret_cast<float*>(ring_lds + NBUF * CHUNK_BYTES);   // 16 tiles: the density row's weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;
    const size_t m0g = (size_t)blockIdx.x * WG_SAMPLES;
    const int row_local = wave * 16 + (lane & 15);
    const size_t m = m0g + row_local;
    const bool valid = m < (size_t)a.M;
    const size_t rows = a.Mp - m0g < (size_t)WG_SAMPLES ? a.Mp - m0g : (size_t)WG_SAMPLES;
    const auto field_of = [&](const float* base, size_t off, int width) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + off * a.Mp + m0g * width, 0, (int)rows * width * 4, 0x00020000);
    };

    // ---- ordinary loads first (the compiler waits for everything in flight at the first use of one of them)
    const float* bias_g = reinterpret_cast<const float*>(a.img + G::IMG_BYTES) + G::bias_tile0(1) * 16;
    if (tid < 256) bias_lds[tid] = bias_g[tid];
    float dzo[3] = {0.f, 0.f, 0.f}, ds = 0.f;
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float col = a.colors[m * 3 + c];
            dzo[c] = a.dcolors[m * 3 + c] * col * (1.0f - col);            // sigmoid'
        }
        // relu' of the density output (Siren: times sigma_mul, models.py:193)
        ds = a.ws[WS_SRAW * a.Mp + m] > 0.0f ? a.ddensity[m] * (IS_NERF ? 1.0f : 10.0f) : 0.0f;
    }
    // ReLU decisions of h0..h8 (rows behind the padded sample count read as 0)
    [[maybe_unused]] u32x2 mk[9];
    if constexpr (IS_NERF) {
#pragma unroll
        for (int l = 0; l < 9; ++l)
            mk[l] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(field_of(a.ws, nerf::WS_MASK + 8 * (size_t)l, 8), row_local * 32 + q * 8, 0, 0));
    }
    // the 128 outputs of the rgb hidden layer (NeRF: g, whose sign decides; Siren: the derivative cos z) and the output
    // layer's weights at this lane's fragment features: k-step s, element j = feature 32 s + 4 q + j (j < 4),
    // 32 s + 16 + 4 q + (j - 4)
    f32x4 gv[4][2], wv[3][4][2];
    {
        const __amdgpu_buffer_rsrc_t g_rsrc = field_of(a.ws, IS_NERF ? WS_G : siren::WS_CG, HH);
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wout), 0, 3 * HH * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                gv[s][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_rsrc, row_local * HH * 4 + q * 16 + (32 * s + 16 * t) * 4, 0, 0));
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    wv[c][s][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, q * 16 + (c * HH + 32 * s + 16 * t) * 4, 0, 0));
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    constexpr auto tail = [](int l) { return G::NCHUNK - G::chunk0(l + 1) < AHEAD ? G::NCHUNK - G::chunk0(l + 1) : AHEAD; };
    Ring ring{a.img, ring_lds, 0, (unsigned)(PER_CHUNK * 1024 * wave + 16 * lane), wave, G::NCHUNK, (int)G::IMG_BYTES};
#pragma unroll
    for (int c = 0; c < AHEAD; ++c) ring_issue(ring, c);
    __builtin_amdgcn_sched_barrier(0);

    // ---- head (head_bwd of mlp_bwd.hip): gradient of the sigmoid outputs, of the density, and of the rgb hidden layer's
    // pre-activations -- as fragments and to the workspace
    const auto put = [](float v, __amdgpu_buffer_rsrc_t rsrc, int voff) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, 0, 0);
    };
    if (q == 0) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{dzo[0], dzo[1], dzo[2], 0.0f}), field_of(a.bw, BW_DZO, 4),
                                               row_local * 16, 0, 0);
        put(ds, field_of(a.bw, BW_DSIG, 1), row_local * 4);
    }
    u32x4 X[3][8];
    {
        const __amdgpu_buffer_rsrc_t dzr = field_of(a.bw, BW_DZR, HH);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = dzo[0] * wv[0][s][t][e] + dzo[1] * wv[1][s][t][e] + dzo[2] * wv[2][s][t][e];
                    d[e] = IS_NERF ? (gv[s][t][e] > 0.0f ? x : 0.0f) : x * gv[s][t][e];
                    v[4 * t + e] = d[e];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), dzr, row_local * HH * 4 + q * 16 + (32 * s + 16 * t) * 4, 0, 0);
            }
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                unsigned tm[3];
                split2<3>(v[2 * jp], v[2 * jp + 1], tm);
#pragma unroll
                for (int p = 0; p < 3; ++p) X[p][s][jp] = tm[p];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * PER_CHUNK) : "memory");   // own pieces of chunk 0, own LDS writes
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 acc[NOT_MAX];
    u32x4 XE[3][2];   // (no encoding k-steps in these chains)
    const auto zero_acc = [&]() {
#pragma unroll
        for (int ot = 0; ot < 16; ++ot) acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    constexpr int N_HEAD_STORES = 10, N_DZ_STORES = 16;
    const int voff_h = row_local * H * 4 + q * 16;
    unsigned none[2] = {0u, 0u};
    // Siren: the first derivative tiles of layer l, requested in front of the MFMA loop whose epilogue uses them
    [[maybe_unused]] DerivTiles dv;
    const auto layer_in = [&](int l) {
        if constexpr (!IS_NERF) deriv_prefetch(dv, field_of(a.ws, siren::WS_C + (size_t)l * H, H), voff_h);
    };
    constexpr int N_PREFETCH = IS_NERF ? 0 : 2 * DERIV_AHEAD;
    // dz(l) = d(h(l)) * activation', to the workspace and as the next layer's fragments
    const auto layer_out = [&](int l) {
        if constexpr (IS_NERF) {
            // the decisions of the layer below move to the front of mk
            unsigned mw[2] = {mk[8][0], mk[8][1]};
#pragma unroll
            for (int i = 8; i >= 1; --i) mk[i] = mk[i - 1];
            to_frags_out<8, FR_MASK, FO_STORE>(acc, X, field_of(a.bw, BW_DZ + (size_t)l * H, H), voff_h, mw);
        } else {
            to_frags_out<8, FR_DERIV, FO_STORE>(acc, X, field_of(a.bw, BW_DZ + (size_t)l * H, H), voff_h, none,
                                                field_of(a.ws, siren::WS_C + (size_t)l * H, H), dv);
        }
    };
    // the rgb hidden layer's feature part: gradient of the features (their layer has no activation)
    zero_acc();
    mma_layer<16, 4, 0, tail(0), 0, N_HEAD_STORES>(ring, lane, acc, X, XE);
    to_frags_out<8, FR_NONE, FO_STORE>(acc, X, field_of(a.bw, BW_DZF, H), voff_h, none);
    // the feature layer on the matrix pipe, the density row / branch as the rank-1 term dsig[sample] * w[feature]
#pragma unroll
    for (int ot = 0; ot < 16; ++ot) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(bias_lds + ot * 16 + q * 4);
        acc[ot] = f32x4{ds * w[0], ds * w[1], ds * w[2], ds * w[3]};
    }
    layer_in(LAST_HIDDEN);
    mma_layer<16, 8, 0, tail(1), 0, N_DZ_STORES + N_PREFETCH>(ring, lane, acc, X, XE);
    layer_out(LAST_HIDDEN);
    // the hidden layers down to the second: dz(l-1) = d(h(l-1)) * activation'
#pragma unroll 1
    for (int l = LAST_HIDDEN; l >= 2; --l) {
        zero_acc();
        layer_in(l - 1);
        mma_layer<16, 8, 0, tail(2), 0, N_DZ_STORES + N_PREFETCH>(ring, lane, acc, X, XE);
        layer_out(l - 1);
    }
    // the second layer ends the chain with dz0 (its fragments are not needed)
    zero_acc();
    layer_in(0);
    mma_layer<16, 8, 0, 0, 0, N_DZ_STORES + N_PREFETCH>(ring, lane, acc, X, XE);
    layer_out(0);
    static_assert(G::NL == LAST_HIDDEN + 2 && tail(G::NL - 1) == 0 && tail(G::NL - 2) == AHEAD, "the chain's last layer ends the stream");
}

template <int MODEL>
int pack_image(const float* const* params, void* packed, void* stream, const char* who) {
    using G = Geo<MODEL>;
    PackArgs a;
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
    a.neg_mask = G::neg_mask();
    const int pieces = G::NCHUNK * CHUNK_UNITS * 64;
    pack_x3i_kernel<MODEL><<<(pieces + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    return nerf_check_launch(who);
}

template <int MODEL>
int launch_fwd(const FwdArgs& a, int64_t M, void* stream, const char* who) {
    const int64_t groups = ceil_div64(M, WG_SAMPLES);
    if (groups >= (1ll << 31)) return nerf_fail(NERF_E_ARG, "%s: too many samples", who);
    if (a.colors) {
        if (int rc = nerf_set_lds(fwd_x3i_kernel<MODEL, false>, LDS_BYTES)) return rc;
        fwd_x3i_kernel<MODEL, false><<<(unsigned)groups, WG_THREADS, LDS_BYTES, (hipStream_t)stream>>>(a);
    } else {
        if (int rc = nerf_set_lds(fwd_x3i_kernel<MODEL, true>, LDS_BYTES)) return rc;
        fwd_x3i_kernel<MODEL, true><<<(unsigned)groups, WG_THREADS, LDS_BYTES, (hipStream_t)stream>>>(a);
    }
    return nerf_check_launch(who);
}

}  // namespace

// Training forward of the bf16x3 mode, called by nerf_mlp_fwd_x3 with arguments it has checked: `image` is what
// nerf_mlp_pack_bf16x3 wrote, `workspace` the activation workspace of nerf_mlp_fwd_workspace_floats.
template <int MODEL>
static int launch_fwd_train(const FwdArgs& a, int64_t groups, void* stream, const char* who) {
    if (int rc = nerf_set_lds(fwd_x3i_kernel<MODEL, false, true>, LDS_BYTES)) return rc;
    fwd_x3i_kernel<MODEL, false, true><<<(unsigned)groups, WG_THREADS, LDS_BYTES, (hipStream_t)stream>>>(a);
    return nerf_check_launch(who);
}
__attribute__((visibility("hidden"))) int nerf_x3i_fwd_train(int model, const void* image, int64_t M, const float* points, const float* dirs,
                                                             const float* rays_o, const float* rays_d, const float* t, int S,
                                                             int t_per_ray, double near, double far, float* colors, float* density,
                                                             float* workspace, void* stream) {
    const char* who = "nerf_mlp_fwd_x3";
    const int64_t groups = ceil_div64(M, WG_SAMPLES);
    if (groups >= (1ll << 31)) return nerf_fail(NERF_E_ARG, "%s: too many samples", who);
    FwdArgs a{static_cast<const unsigned char*>(image), M, points, dirs, rays_o, rays_d, t, S, t_per_ray,
              (float)near, (float)(far - near), colors, density, workspace, lay::padded_samples(M)};
    return model == NERF_MODEL_NERF ? launch_fwd_train<NERF_MODEL_NERF>(a, groups, stream, who)
                                    : launch_fwd_train<NERF_MODEL_SIREN>(a, groups, stream, who);
}

// Data-gradient chains of the bf16x3 training mode, called by nerf_mlp_pack_x3 / nerf_mlp_bwd_data_x3 with checked arguments.
__attribute__((visibility("hidden"))) size_t nerf_x3i_dgrad_image_bytes(int model) {
    return model == NERF_MODEL_NERF ? Geo<NERF_CHAIN_DGRAD>::PACKED_BYTES : Geo<SIREN_CHAIN_DGRAD>::PACKED_BYTES;
}
__attribute__((visibility("hidden"))) int nerf_x3i_dgrad_pack(int model, const float* const* params, void* image, void* stream) {
    return model == NERF_MODEL_NERF ? pack_image<NERF_CHAIN_DGRAD>(params, image, stream, "nerf_mlp_pack_x3")
                                    : pack_image<SIREN_CHAIN_DGRAD>(params, image, stream, "nerf_mlp_pack_x3");
}
template <int MODEL>
static int launch_dgrad(const DgradArgs& a, int64_t groups, void* stream, const char* who) {
    if (int rc = nerf_set_lds(dgrad_x3i_kernel<MODEL>, LDS_BYTES)) return rc;
    dgrad_x3i_kernel<MODEL><<<(unsigned)groups, WG_THREADS, LDS_BYTES, (hipStream_t)stream>>>(a);
    return nerf_check_launch(who);
}
__attribute__((visibility("hidden"))) int nerf_x3i_dgrad(int model, const void* image, const float* wout, int64_t M, const float* dcolors,
                                                         const float* ddensity, const float* colors, const float* fwd_workspace,
                                                         float* bwd_workspace, void* stream) {
    const char* who = "nerf_mlp_bwd_data_x3";
    const int64_t groups = ceil_div64(M, WG_SAMPLES);
    if (groups >= (1ll << 31)) return nerf_fail(NERF_E_ARG, "%s: too many samples", who);
    DgradArgs a{static_cast<const unsigned char*>(image), wout, M, lay::padded_samples(M), dcolors, ddensity, colors, fwd_workspace, bwd_workspace};
    return model == NERF_MODEL_NERF ? launch_dgrad<NERF_MODEL_NERF>(a, groups, stream, who) : launch_dgrad<NERF_MODEL_SIREN>(a, groups, stream, who);
}

extern "C" size_t nerf_mlp_packed_bf16x3_bytes(int model) {
    return model == NERF_MODEL_NERF ? Geo<NERF_MODEL_NERF>::PACKED_BYTES : model == NERF_MODEL_SIREN ? Geo<NERF_MODEL_SIREN>::PACKED_BYTES : 0;
}

extern "C" int nerf_mlp_pack_bf16x3(int model, const float* const* params, void* packed, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_pack_bf16x3: unknown model %d", model);
    NERF_REQUIRE_PTR(params);
    NERF_REQUIRE_PTR(packed);
    return model == NERF_MODEL_NERF ? pack_image<NERF_MODEL_NERF>(params, packed, stream, "nerf_mlp_pack_bf16x3")
                                    : pack_image<NERF_MODEL_SIREN>(params, packed, stream, "nerf_mlp_pack_bf16x3");
}

extern "C" int nerf_mlp_fwd_bf16x3(int model, const void* packed, int64_t M, const float* points, const float* dirs,
                                   const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray,
                                   double near, double far, float* colors, float* density, void* stream) {
    const char* who = "nerf_mlp_fwd_bf16x3";
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "%s: unknown model %d", who, model);
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
    FwdArgs a{static_cast<const unsigned char*>(packed), M, points, dirs, rays_o, rays_d, t, S, t_per_ray,
              (float)near, (float)(far - near), colors, density, nullptr, 0};
    return model == NERF_MODEL_SIREN ? launch_fwd<NERF_MODEL_SIREN>(a, M, stream, who) : launch_fwd<NERF_MODEL_NERF>(a, M, stream, who);
}
