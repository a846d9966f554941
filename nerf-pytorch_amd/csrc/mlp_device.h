// Device building blocks of the fused MLP kernels (gfx950, wave64, fp32 MFMA).
//
// Tile geometry: a workgroup of 4 waves owns TR samples (NERF_TILE_ROWS: 64, two workgroups per CU,
// or 32, four per CU).  The current layer input lives in LDS as act[TR][LDA] (row = sample, LDA = 260
// floats so that the 16-byte A fragment reads of 32 consecutive rows fall on distinct banks).  Wave w
// computes all TR rows x columns [w*32*NT, (w+1)*32*NT) of the layer output with
// v_mfma_f32_32x32x2_f32: acc[mi][ni] is the 32x32 tile (rows 32*mi.., cols ncol0+32*ni..), mi < MT = TR/32.
//
// MFMA operand maps (32x32x2 f32): lane l supplies A[row l&31][k l>>5] and
// B[k l>>5][col l&31]; accumulator register g of lane l is
// D[row (g&3) + 8*(g>>2) + 4*(l>>5)][col l&31].
// One 16-byte fragment load gives a lane the values k = 8*kb + 4*(l>>5) + j, j = 0..3;
// MFMA step j therefore contracts k = 8*kb + j and 8*kb + 4 + j -- the order inside a K
// block is permuted identically for A and B, which leaves the sum unchanged.
#pragma once
#include "common.h"
#include "mlp_layout.h"
#include "sincos.h"
#include "sincos_table.h"

#include <type_traits>

namespace dev {

#ifndef NERF_TILE_ROWS
#define NERF_TILE_ROWS 64
#endif
constexpr int TR = NERF_TILE_ROWS;          // samples per workgroup tile
constexpr int MT = TR / 32;                 // 32-row MFMA tiles per wave
constexpr int TPS = 256 / TR;               // threads per sample in the per-sample phases (encodings, heads)
constexpr int WG_PER_CU = TR == 64 ? 2 : 4; // LDS and register budget of the tile kernels
static_assert((TR == 32 || TR == 64) && lay::TILE % TR == 0, "tile height");
// (h > 0) bits of one lane's accumulators of a 256-wide layer: MT*2 tiles x 16 registers
typedef std::conditional<MT == 2, unsigned long long, unsigned>::type mask_t;

constexpr int LDA = 260;
constexpr int ACT_FLOATS = TR * LDA;

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIN = 2 };

// (sin, cos) of 2*pi*k/256 for the table form of the sine/cosine (sincos.h, nerf_sincosf_tab).  Kernels with sine
// epilogues copy it into the FIRST 2 KiB of their LDS once per workgroup, de-interleaved (sin[256], then cos[256]), and
// read it from there: LDS reads cost the matrix pipe nothing, vector-ALU instructions do.
constexpr int SINCOS_TAB_FLOATS = 2 * NERF_SINCOS_TABLE_ENTRIES;
static __device__ const float g_sincos_table[SINCOS_TAB_FLOATS] = {NERF_SINCOS_TABLE_VALUES};
typedef float f32x2 __attribute__((ext_vector_type(2)));

// `lds0` must be the start of the kernel's LDS allocation (byte offset 0: the table reads below address it through
// the instruction's immediate offset alone); traps otherwise.
__device__ __forceinline__ void load_sincos_table(float* lds0, int tid) {
    static_assert(NERF_SINCOS_TABLE_ENTRIES == 256, "one entry per thread of a 256-thread workgroup");
    if ((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds0 != 0u) __builtin_trap();
    const f32x2 e = *reinterpret_cast<const f32x2*>(g_sincos_table + 2 * tid);
    lds0[tid] = e[0];
    lds0[NERF_SINCOS_TABLE_ENTRIES + tid] = e[1];
}

// nerf_sincosf_tab for two values at once, split in two halves so that the table reads of several pairs can be in
// flight together.  Same operations in the same order as the scalar form (bitwise the same results), written on
// 2-vectors so that they become packed instructions (v_pk_fma_f32 ...): 13 per pair, plus one SDWA shift per value
// for the table offset (byte 0 of the rounded sum's bit pattern = k mod 256, times 4 = the LDS byte address, the
// table sitting at LDS offset 0).  The four table reads of a pair are VOLATILE loads: ordinary ones are merged into
// two-address reads (one value's sin and cos), whose result pairs then have to be shuffled into the (value a,
// value b) pairs the packed instructions want -- two moves per value; volatile ones stay single and land where the
// packed instructions read them.  The compiler still tracks them (it places the waits), unlike reads issued from
// inline assembly.
struct SinCosPair {
    f32x2 r;      // reduced arguments
    f32x2 S, C;   // table entries of the two values
};
typedef const volatile __attribute__((address_space(3))) float* lds_table_ptr;
__device__ __forceinline__ void sincos_tab_issue(f32x2 x, SinCosPair& p) {
#pragma clang fp contract(off)   // the operations as written: the sequence the host sweep has checked
    const f32x2 magic = {12582912.0f, 12582912.0f};
    const f32x2 tm = __builtin_elementwise_fma(x, (f32x2){40.743665431525205956834243423364f, 40.743665431525205956834243423364f}, magic);
    const f32x2 kf = tm - magic;
    const f32x2 r = __builtin_elementwise_fma(-kf, (f32x2){0.024543693289160728f, 0.024543693289160728f}, x);
    p.r = __builtin_elementwise_fma(-kf, (f32x2){-6.8299044198028014e-10f, -6.8299044198028014e-10f}, r);
    // (the elements go through plain floats: __builtin_bit_cast applied directly to an element of an ext_vector
    // reads element 0 whatever the index with this compiler)
    const float tm0 = tm[0], tm1 = tm[1];
    unsigned o0, o1;
    asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                 : "=v"(o0) : "v"(2u), "v"(tm0));
    asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                 : "=v"(o1) : "v"(2u), "v"(tm1));
    p.S[0] = *(lds_table_ptr)(size_t)o0;
    p.S[1] = *(lds_table_ptr)(size_t)o1;
    p.C[0] = *(lds_table_ptr)(size_t)(o0 + 4 * NERF_SINCOS_TABLE_ENTRIES);
    p.C[1] = *(lds_table_ptr)(size_t)(o1 + 4 * NERF_SINCOS_TABLE_ENTRIES);
}
__device__ __forceinline__ void sincos_tab_finish(const SinCosPair& p, f32x2& sn, f32x2& cs) {
#pragma clang fp contract(off)
    const f32x2 z = p.r * p.r;
    const f32x2 w = __builtin_elementwise_fma(z, (f32x2){-0.16666667f, -0.16666667f}, (f32x2){1.0f, 1.0f});
    const f32x2 hr = p.r * (f32x2){0.5f, 0.5f};
    sn = __builtin_elementwise_fma(p.r, __builtin_elementwise_fma(-p.S, hr, p.C * w), p.S);
    cs = __builtin_elementwise_fma(-p.r, __builtin_elementwise_fma(p.C, hr, p.S * w), p.C);
}

#define LAYER_SYNC() __syncthreads()

__device__ __forceinline__ f32x16 splat16(float v) {
    f32x16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = v;
    return r;
}

// One K block of MFMAs for the wave's TR x (32*NT) tile: 4 steps x MT x NT instructions.
template <int NT>
__device__ __forceinline__ void mma_block(const f32x4 (&a)[MT], const f32x4 (&b)[NT], f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
}

// How a K loop finds its accumulators: ACC_KEEP continues the sums of an earlier pass; ACC_ZERO / ACC_BIAS start
// them in the first MFMA of every tile, whose C operand is the constant 0 or a 16-register tile holding the lane's
// bias (one tile per column block, shared by the row blocks), instead of 16 register moves per accumulator tile
// beforehand.  VALU instructions do not overlap the fp32 MFMAs of any wave on the same SIMD, so every move saved
// is matrix-pipe time.
enum AccInit { ACC_KEEP = 0, ACC_ZERO = 1, ACC_BIAS = 2 };

template <int NT, int INIT>
__device__ __forceinline__ void mma_block_first(const f32x4 (&a)[MT], const f32x4 (&b)[NT], f32x16 (&acc)[MT][NT],
                                                const float (&bias)[NT]) {
    static_assert(INIT == ACC_ZERO || INIT == ACC_BIAS, "fresh accumulators only");
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        if constexpr (INIT == ACC_ZERO) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][0], b[ni][0], zero, 0, 0, 0);
        } else {
            const f32x16 btile = splat16(bias[ni]);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][0], b[ni][0], btile, 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
}

// Global operands of the K loops go through buffer instructions: the resource (base, size) lives in
// SGPRs, the per-lane part of the address is ONE 32-bit VGPR that never changes, and the K block
// advances in the scalar offset.  Measured on the K loop alone: two global_load_dwordx4 (64-bit
// address VGPR pairs) per 16 MFMAs cost 5 % of the MFMA rate, two buffer_load_dwordx4 nothing.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int NT>
__device__ __forceinline__ void load_b(__amdgpu_buffer_rsrc_t img, int voff, int kb, int kb_bytes, f32x4 (&b)[NT]) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
        b[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, voff + ni * 1024, kb * kb_bytes, 0));
}
__device__ __forceinline__ void load_a(const float* a_ptr, int kb, f32x4 (&a)[MT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(a_ptr + mi * 32 * LDA + kb * 8);
}

// acc[mi][ni] (+)= act[rows 32*mi.., k < 8*kb_count] * img, img = packed image with `n_img`
// columns per K block; this wave's first column is ncol0.  kb_count must be even.
//
// Software-pipelined by hand with two named fragment sets, pinned with sched_barrier: the loads
// of K block kb+1 are issued BEFORE the 16 MFMAs (about 1000 cycles) of block kb.  Left to itself
// the compiler sinks each load to just before its first use and pays the L2 latency of the weight
// fetch on every block.  (A three-block-deep ring was measured and is not faster.)  The loop is
// deliberately not unrolled further: twelve fully unrolled layers are ~90 KB of code.
//
// COPY_COLS != 0 (= 8*kb_count): the loop also copies the tile it reads, act[TR][COPY_COLS], to
// global rows of COPY_COLS floats at gcopy -- one 16-byte piece per thread and storing iteration
// (TR*COPY_COLS/4 pieces = 256 threads * kb_count/2 iterations for 64 rows; with 32 rows every second
// iteration stores).  The piece is read from LDS one iteration ahead and stored right after the weight
// loads of the odd block: vmcnt retires in order, so the first wait that covers the store is the one
// for the weight loads issued a whole MFMA block later, and no wave ever sits on a store acknowledgement.
template <int NT, int COPY_COLS = 0, int INIT = ACC_KEEP>
__device__ __forceinline__ void mma_layer(const float* act, const float* __restrict__ img, int n_img, int kb_count,
                                          int ncol0, int lane, f32x16 (&acc)[MT][NT], float* __restrict__ gcopy = nullptr,
                                          int tid = 0, const float* __restrict__ bias = nullptr) {
    float bias_v[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) bias_v[ni] = (INIT == ACC_BIAS) ? bias[ncol0 + 32 * ni + (lane & 31)] : 0.0f;
    const int r = lane & 31, h = lane >> 5;
    const float* a_ptr = act + r * LDA + 4 * h;
    const int kb_bytes = n_img * 32;                                        // one K block of the image
    const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(img, (unsigned)(kb_count * kb_bytes));
    const int b_voff = ((ncol0 + r) * 8 + 4 * h) * 4;
    const int last = kb_count - 1;
    f32x4 a0[MT], a1[MT], b0[NT], b1[NT];
    constexpr int C4N = COPY_COLS ? COPY_COLS / 4 : 1;       // 16-byte pieces per tile row
    constexpr int RSTEP = 256 / C4N;                         // tile rows covered per storing iteration
    constexpr int SE = 64 / TR;                              // every SE-th iteration stores
    const float* csrc = nullptr;
    __amdgpu_buffer_rsrc_t c_rsrc = b_rsrc;
    int c_voff = 0, c_soff = 0;
    f32x4 piece = {0.f, 0.f, 0.f, 0.f};
    if constexpr (COPY_COLS != 0) {
        const int row = tid / C4N, c4 = tid % C4N;
        csrc = act + row * LDA + 4 * c4;
        c_rsrc = make_rsrc(gcopy, TR * COPY_COLS * 4);
        c_voff = (row * COPY_COLS + 4 * c4) * 4;
        piece = *reinterpret_cast<const f32x4*>(csrc);
    }
    load_b<NT>(b_rsrc, b_voff, 0, kb_bytes, b0);
    load_a(a_ptr, 0, a0);
    // VALU instructions between a wave's MFMAs cost 10-14 cycles of matrix-pipe time each (LDS, buffer and scalar
    // instructions cost none), so the loop keeps ONE address register for the A fragments, advanced once per trip with the
    // other blocks at immediate offsets; the last trip reads 8 columns past the K range (inside the LDS allocation,
    // never used).  Two trips are unrolled so that the copy's LDS address also moves once per two trips.
    const float* a_cur = a_ptr;
#pragma unroll 2
    for (int kb = 0; kb < kb_count; kb += 2) {
        load_b<NT>(b_rsrc, b_voff, kb + 1, kb_bytes, b1);
        if constexpr (COPY_COLS != 0) {
            if (SE == 1 || ((kb >> 1) & (SE - 1)) == 0) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, piece), c_rsrc, c_voff, c_soff, 2);  // 2 = nt
                c_soff += RSTEP * COPY_COLS * 4;
                csrc += RSTEP * LDA;
            }
        }
        load_a(a_cur, 1, a1);
        if constexpr (COPY_COLS != 0) {
            if ((SE == 1 || (((kb >> 1) + 1) & (SE - 1)) == 0) && kb + 2 < kb_count)
                piece = *reinterpret_cast<const f32x4*>(csrc);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (INIT != ACC_KEEP) {
            if (kb == 0) mma_block_first<NT, INIT>(a0, b0, acc, bias_v);
            else mma_block<NT>(a0, b0, acc);
        } else {
            mma_block<NT>(a0, b0, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int kn = min(kb + 2, last);  // clamped: the tail prefetch re-reads the last block of the image
        load_b<NT>(b_rsrc, b_voff, kn, kb_bytes, b0);
        load_a(a_cur, 2, a0);
        a_cur += 16;
        __builtin_amdgcn_sched_barrier(0);
        mma_block<NT>(a1, b1, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulators start at the bias of their column
template <int NT>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[MT][NT], const float* __restrict__ bias, int ncol0, int lane) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        const float b = bias ? bias[ncol0 + 32 * ni + (lane & 31)] : 0.0f;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi][ni] = splat16(b);
    }
}

__device__ __forceinline__ int acc_row(int mi, int g, int lane) { return 32 * mi + (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5); }

// Forward epilogue: activation, write the tile back to LDS as the next layer's input.
// The post-activation tile itself goes to the workspace later, from LDS, inside the next MFMA loop
// that reads it (mma_layer's COPY_COLS); for sine layers (training) the derivative cos(z) has no LDS
// home and is stored from the accumulator registers: gcos = this tile's [TR][ldg] region of the workspace.
//   NT == 1 (the 128-wide rgb sine layer, whose derivative the head gradient reads by rows): row-major, one
//           4-byte store per element;
//   NT == 2 (hidden sine layers): ACCUMULATOR ORDER -- the region holds, for every group of four accumulator
//           registers q = (mi*NT + ni)*4 + g/4, one 16-byte piece per thread at float offset (q*256 + tid)*4,
//           so a thread stores (and the data-gradient kernel, whose tiles have the same register map, loads)
//           16 fully coalesced 16-byte pieces instead of 64 strided 4-byte ones.
// relu_mask (may be NULL): receives this lane's MT*32 (v > 0) bits, accumulator order.
// Sine layers need the sine/cosine table in the first 2 KiB of LDS (load_sincos_table).
// Sine layers: v = sin(z) to LDS, cos(z) to the workspace (training), through the table form of the sine/cosine:
// per 16-register accumulator tile the table reads of all eight value pairs are issued first, then finished.
template <int NT, bool STORE_COS>
__device__ __forceinline__ void sine_epilogue(float* act, f32x16 (&acc)[MT][NT], int ncol0, int lane,
                                              float* __restrict__ gcos, int ldg) {
    __amdgpu_buffer_rsrc_t cos_rsrc;
    int cos_voff = 0;
    if constexpr (STORE_COS) {
        cos_rsrc = make_rsrc(gcos, (unsigned)(TR * ldg * 4));
        if constexpr (NT == 1) cos_voff = (4 * (lane >> 5) * ldg + ncol0 + (lane & 31)) * 4;
        else cos_voff = ((ncol0 / (32 * NT)) * 64 + lane) * 16;   // tid * 16 bytes
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int col = ncol0 + 32 * ni + (lane & 31);
            SinCosPair pr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) sincos_tab_issue((f32x2){acc[mi][ni][2 * q], acc[mi][ni][2 * q + 1]}, pr[q]);
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                f32x2 sa, ca, sb, cb;
                sincos_tab_finish(pr[q], sa, ca);
                sincos_tab_finish(pr[q + 1], sb, cb);
                const int g = 2 * q;   // registers g .. g+3: rows (g&3) + 8*(g>>2) + {0,1,2,3} of the tile
                act[acc_row(mi, g, lane) * LDA + col] = sa[0];
                act[acc_row(mi, g + 1, lane) * LDA + col] = sa[1];
                act[acc_row(mi, g + 2, lane) * LDA + col] = sb[0];
                act[acc_row(mi, g + 3, lane) * LDA + col] = sb[1];
                if constexpr (STORE_COS) {
                    if constexpr (NT == 1) {
                        const float cv[4] = {ca[0], ca[1], cb[0], cb[1]};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cv[e]), cos_rsrc, cos_voff,
                                                                  ((32 * mi + ((g + e) & 3) + 8 * ((g + e) >> 2)) * ldg + 32 * ni) * 4, 0);
                    } else {
                        // The piece offset goes into the VGPR offset and the scalar offset stays the constant 0 on
                        // purpose.  With the offset in an SGPR, hipcc lets a VALU instruction overwrite the four data
                        // registers in the very next slot (its hazard model exempts stores with a register soffset), and
                        // on gfx950 the store then writes a mix of old and new values for some 4-lane groups once several
                        // workgroups keep the memory pipeline busy (seen as run-to-run differences of the derivative
                        // tiles from 1024 tiles on).  With an immediate soffset the compiler inserts the wait state.
                        const f32x4 cpiece = {ca[0], ca[1], cb[0], cb[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, cpiece), cos_rsrc,
                                                               cos_voff + ((mi * NT + ni) * 4 + (g >> 2)) * 256 * 16, 0, 2);
                    }
                }
            }
        }
}

template <int NT, int ACT, bool STORE_COS>
__device__ __forceinline__ void fwd_epilogue(float* act, f32x16 (&acc)[MT][NT], int ncol0, int lane,
                                             float* __restrict__ gcos, int ldg,
                                             mask_t* __restrict__ relu_mask = nullptr) {
    if constexpr (ACT == ACT_SIN) {
        sine_epilogue<NT, STORE_COS>(act, acc, ncol0, lane, gcos, ldg);
        return;
    }
    unsigned mlo = 0u, mhi = 0u;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int col = ncol0 + 32 * ni + (lane & 31);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = acc_row(mi, g, lane);
                float v = acc[mi][ni][g];
                if constexpr (ACT == ACT_RELU) {
                    // ReLU layers arrive NEGATED: their weight image and bias are packed as 0 - w (nerf_mlp_pack), so
                    // the accumulator holds n = -(W x + b) exactly (rounding is symmetric) and the decision bit
                    // "pre-activation > 0" IS n's sign bit -- n = -0 cannot arise from a sum that starts at a bias packed
                    // as 0 - b, and an exactly cancelling sum gives +0 -- shifted into the mask word as it is: element
                    // e = (mi*NT + ni)*16 + g ends up at bit 31 - (e & 31) of word e >> 5.  One instruction per element
                    // for the mask and one for max(-n, 0) (written as the instruction: fmaxf, and every intrinsic the
                    // compiler folds back into it, puts a canonicalising v_max_f32 v, v, v in front); vector instructions
                    // are matrix-pipe time here.
                    if constexpr (NT == 2) {
                        const int bit = (mi * NT + ni) * 16 + g;
                        const unsigned t = __builtin_bit_cast(unsigned, v);
                        if (bit < 32) mlo = __builtin_amdgcn_alignbit(mlo, t, 31);
                        else mhi = __builtin_amdgcn_alignbit(mhi, t, 31);
                    }
                    asm("v_max_f32_e64 %0, 0, -%1" : "=v"(v) : "v"(v));
                }
                act[row * LDA + col] = v;
            }
        }
    if constexpr (ACT == ACT_RELU && NT == 2) {
        if (relu_mask) {
            if constexpr (MT == 2) *relu_mask = ((unsigned long long)mhi << 32) | mlo;
            else *relu_mask = mlo;
        }
    }
}

// Backward epilogue: dX = acc (+ rank-1 term) times the activation derivative, written to
// LDS (next layer's dY); the tile is copied to the backward workspace (for the weight gradient)
// by the next MFMA loop, which reads it (mma_layer's COPY_COLS).
//   MASK 1: derivative = (hsrc > 0)         (ReLU; hsrc = the stored post-activation input)
//   MASK 2: derivative = hsrc               (sine; hsrc = stored w0*cos(w0 z))
//   MASK 3: derivative = bit of `bits`      (ReLU; the forward's per-lane mask word)
//   MASK 0: none
template <int NT, int MASK, bool RANK1>
__device__ __forceinline__ void bwd_epilogue(float* act, f32x16 (&acc)[MT][NT], int ncol0, int lane,
                                             const float* __restrict__ hsrc, int ldg,
                                             const float* dsig_lds, const float* __restrict__ sigw,
                                             mask_t bits = 0) {
    const unsigned blo = (unsigned)bits, bhi = (unsigned)((unsigned long long)bits >> 32);
    __amdgpu_buffer_rsrc_t h_rsrc;
    int h_voff = 0;
    if constexpr (MASK == 1 || MASK == 2) {
        h_rsrc = make_rsrc(hsrc, (unsigned)(TR * ldg * 4));
        h_voff = (4 * (lane >> 5) * ldg + ncol0 + (lane & 31)) * 4;
    }
    auto stored = [&](int mi, int ni, int g) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                             h_rsrc, h_voff, ((32 * mi + (g & 3) + 8 * (g >> 2)) * ldg + 32 * ni) * 4, 0));
    };
    // MASK 2: the stored derivative tile is in accumulator order (see fwd_epilogue): 16-byte pieces
    const int d_voff = ((ncol0 / (32 * NT)) * 64 + lane) * 16;
    auto stored4 = [&](int mi, int ni, int g4) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, d_voff,
                                                                               ((mi * NT + ni) * 4 + g4) * 256 * 16, 0));
    };
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            const int col = ncol0 + 32 * ni + (lane & 31);
            float sw = 0.f;
            if constexpr (RANK1) sw = sigw[col];
            f32x4 dpiece[4];
            if constexpr (MASK == 2) {
                static_assert(MASK != 2 || NT == 2, "accumulator-order derivative tiles exist for the 256-wide layers");
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) dpiece[g4] = stored4(mi, ni, g4);
            }
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = acc_row(mi, g, lane);
                float v = acc[mi][ni][g];
                if constexpr (RANK1) v = fmaf(dsig_lds[row], sw, v);
                if constexpr (MASK == 1) v = stored(mi, ni, g) > 0.0f ? v : 0.0f;
                if constexpr (MASK == 2) v = v * dpiece[g >> 2][g & 3];
                if constexpr (MASK == 3) {
                    const int bit = (mi * NT + ni) * 16 + g;       // at bit 31 - (bit & 31) of its word (fwd_epilogue)
                    const int keep = __builtin_amdgcn_sbfe((int)(bit < 32 ? blo : bhi), 31 - (bit & 31), 1);   // 0 or all ones
                    v = __builtin_bit_cast(float, __builtin_bit_cast(int, v) & keep);
                }
                act[row * LDA + col] = v;
            }
        }
}

// positional encoding of one coordinate into a tile row: frequencies j0 <= j < j1 of the sin/cos
// pairs, plus x itself when j0 == 0.  `dst` points at the row start; column layout of nerf/encoding.py
// (blocks of 3).
__device__ __forceinline__ void encode_axis(float* dst, int axis, float x, int j0, int j1) {
    if (j0 == 0) dst[axis] = x;
    float scale = (float)(1 << j0);
#pragma unroll 1   // one copy of the full-range sincos (its large-argument path is ~100 instructions)
    for (int j = j0; j < j1; ++j) {
        float s, c;
        nerf_sincosf(mul_rn(scale, x), &s, &c);
        dst[3 + 6 * j + axis] = s;
        dst[3 + 6 * j + 3 + axis] = c;
        scale *= 2.0f;
    }
}

// Table form of the same encoding for 64-row tiles (four threads per sample), used when every argument of the tile
// stays inside the table form's range (|2^(L-1) x| <= 9.9e4, decided once per tile by the caller).  Threads 0..2 of a
// sample own one axis each and the frequency pairs (2i, 2i+1), i < ROUNDS; for L = 10 thread 3 owns the fifth pair
// (frequencies 8, 9) of all three axes, one axis per round -- 4 rounds of one value pair per thread instead of 10
// full-range evaluations on three threads out of four.  Per value pair: 13 packed operations + 2 table-offset shifts
// (sincos_tab_issue / sincos_tab_finish) against ~50 vector instructions per frequency of the full-range form, and
// vector instructions are matrix-pipe time in these kernels.  The results stay in registers (EncCache) so that the
// second use of the position encoding (the skip layer) only writes them to LDS again.
// The sine/cosine table must sit in the first 2 KiB of LDS (load_sincos_table).
template <int L>
struct EncCache {
    static constexpr int ROUNDS = L == 10 ? 4 : L / 2;
    f32x2 sn[ROUNDS], cs[ROUNDS];
    float x;   // the thread's own coordinate (threads 0..2)
};
template <int L>
__device__ __forceinline__ bool enc_round_active(int q, int i) { return q < 3 || (L == 10 && i < 3); }

// xs: the sample's three coordinates in LDS (xin row + 0 for positions, + 3 for directions)
template <int L>
__device__ __forceinline__ void enc_compute(const float* xs, int q, EncCache<L>& c) {
    static_assert(TPS == 4 && (L == 10 || L == 4), "four threads per sample");
    constexpr int R = EncCache<L>::ROUNDS;
    const bool extra = q == 3;
    c.x = xs[extra ? 0 : q];
    SinCosPair pr[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        // powers of two: the products are exact, as mul_rn(2^j, x) in the full-range form
        const float xv = (L == 10 && extra) ? xs[i < 3 ? i : 0] : c.x;
        const float scale = (L == 10 && extra) ? 256.0f : (float)(1 << (2 * i));
        const float a0 = mul_rn(scale, xv);
        sincos_tab_issue((f32x2){a0, mul_rn(2.0f, a0)}, pr[i]);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) sincos_tab_finish(pr[i], c.sn[i], c.cs[i]);
}

// writes the cached encoding into the sample's tile row (column layout of nerf/encoding.py: x, then per frequency
// j the three sines at 3 + 6j and the three cosines at 6 + 6j); thread 3 also zeroes the padding columns
template <int L>
__device__ __forceinline__ void enc_store(float* row, int q, const EncCache<L>& c) {
    constexpr int R = EncCache<L>::ROUNDS;
    const bool extra = q == 3;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (enc_round_active<L>(q, i)) {
            // frequency pair p = i (threads 0..2, axis q) or p = 4 (thread 3, axis i): column 3 + 12 p + axis
            float* d = row + (extra ? 48 + i : 12 * i + q);
            d[3] = c.sn[i][0];
            d[6] = c.cs[i][0];
            d[9] = c.sn[i][1];
            d[12] = c.cs[i][1];
        }
    }
    if (!extra) row[q] = c.x;
    else {
#pragma unroll
        for (int col = 3 + 6 * L; col < ((3 + 6 * L + 7) & ~7); ++col) row[col] = 0.0f;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// copy `cols` (multiple of 4) columns of the TR-row LDS tile to global rows of `ldg` floats
__device__ __forceinline__ void store_tile(const float* act, int cols, float* __restrict__ g, int ldg, int tid) {
    const int c4n = cols >> 2;
    for (int i = tid; i < TR * c4n; i += 256) {
        const int row = i / c4n, c4 = i - row * c4n;
        *reinterpret_cast<f32x4*>(g + (size_t)row * ldg + 4 * c4) = *reinterpret_cast<const f32x4*>(act + row * LDA + 4 * c4);
    }
}

}  // namespace dev
