// Small pieces shared by the inference kernels on the bf16 matrix pipe (mlp_bf16.hip: 32-sample tiles;
// mlp_x3i.hip: three-term kernel on 16-sample tiles): vector types, compile-time loops, the three-term
// split of fp32 values into bf16 terms, and the layer tables of the two models.
#pragma once
#include "mlp_device.h"

#include <utility>

namespace bf16k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int UNIT_BYTES = 1024;     // one A fragment for a whole wave: 64 lanes x 8 bf16

// ---- layer tables ------------------------------------------------------------------------
enum Enc { ENC_NONE = 0, ENC_PE = 1, ENC_DE = 2, ENC_XYZ = 3 };
struct LayerInfo {
    int n_ot;      // 32-row output tiles
    int nks_acc;   // k-steps fed by the previous layer's accumulators (16 features each)
    int enc;       // extra k-steps fed by an encoding (ENC_PE, ENC_DE) or the raw coordinates (ENC_XYZ)
    int n_out;     // valid output rows
    int n_in;      // valid accumulator-fed inputs (also the column where the encoding part starts)
    int ld;        // row pitch of the [out, in] weight tensor
    int wtensor;   // state_dict index of the weight; the bias follows it
    int last_row_tensor;   // >= 0: output row n_out-1 is row 0 of this [1, ld] weight (its bias follows it)
    // Layers of a data-gradient chain (mlp_x3i.hip): the layer computes dX = dZ W, its A operand is W transposed --
    // "output row" i is input column i of the [n_in, ld] weight, k runs over the weight's rows.  rank1_tensor >= 0: the
    // layer's bias image holds row rank1_row of that tensor (the density row, whose gradient enters as a rank-1 term).
    int transposed;
    int rank1_tensor, rank1_row;
};
constexpr int NL_MAX = 12;
template <int MODEL> struct Net;
// nerf/models.py:62-75.  block2.8 (257 rows: 256 features + the density row) is one layer.
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
// layer whose 257th row comes from the density branch's tensors.
template <> struct Net<NERF_MODEL_SIREN> {
    static constexpr int NL = 11;
    static constexpr LayerInfo LAYERS[NL_MAX] = {
        {8, 0, ENC_XYZ, 256, 0, 3, 0, -1},       {8, 16, ENC_NONE, 256, 256, 256, 2, -1},  {8, 16, ENC_NONE, 256, 256, 256, 4, -1},
        {8, 16, ENC_NONE, 256, 256, 256, 6, -1}, {8, 16, ENC_NONE, 256, 256, 256, 8, -1},  {8, 16, ENC_NONE, 256, 256, 256, 10, -1},
        {8, 16, ENC_NONE, 256, 256, 256, 12, -1}, {8, 16, ENC_NONE, 256, 256, 256, 14, -1}, {9, 16, ENC_NONE, 257, 256, 256, 18, 16},
        {4, 16, ENC_DE, 128, 256, 283, 20, -1},   {1, 8, ENC_NONE, 3, 128, 128, 22, -1},    {0, 0, ENC_NONE, 0, 0, 0, 0, -1},
    };
};

// ---- compile-time loops --------------------------------------------------------------------
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

// ---- three-term split ------------------------------------------------------------------------
// term `plane` of the three-term bf16 expansion of x (term 0 alone is the plain bf16 rounding)
__host__ __device__ inline __bf16 bf16_term(float x, int plane) {
    __bf16 t = (__bf16)x;
    for (int i = 0; i < plane; ++i) {
        x -= (float)t;      // exact: the difference of a float and its bf16 rounding fits a float
        t = (__bf16)x;
    }
    return t;
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

__device__ __forceinline__ float norm_coord16(float p, float nearf, float spanf) {
#pragma clang fp contract(off)  // same operation order and rounding as the fp32 path
    return sub_rn(div_rn(mul_rn(2.0f, sub_rn(p, nearf)), spanf), 1.0f);
}

}  // namespace bf16k
