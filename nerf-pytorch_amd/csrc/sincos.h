// fp32 sine/cosine pair with a three-constant Cody-Waite reduction, accurate to about
// 1 ulp for |x| <= 1e5 (positional-encoding arguments reach 2^9 * |x| ~ 1e3 rad, the first
// Siren layer ~1e2 rad).  Larger arguments take the math library's full-range path.
// Host-compilable (plain C++) so that tests can sweep it against double precision.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define NERF_HD __host__ __device__ __forceinline__
#else
#define NERF_HD inline
#endif

// The reduction + polynomials alone.  For |x| <= 1e5 this IS nerf_sincosf; beyond that the error grows
// smoothly (kf*P1 stops being exact) instead of switching to the library's Payne-Hanek path, whose
// inlined code is ~100 instructions per call site.  For call sites whose argument is bounded by
// construction: pre-activations of sine layers fed by sines (|z| <= sum|w| + |b|).
NERF_HD void nerf_sincosf_bounded(float x, float* sn, float* cs) {
    // nearest multiple of pi/2: adding 1.5 * 2^23 rounds to an integer (to nearest even, as rintf does) for
    // |x * 2/pi| < 2^22 and leaves that integer in the low bits of the sum -- two additions that pair up into packed
    // instructions, where a multiplication, rintf and a float -> int conversion are three, two of them unpackable
    const float tm = fmaf(x, 0.63661977236758134f, 12582912.0f);   // one rounding, in every translation unit
    const float kf = tm - 12582912.0f;
    // pi/2 = P1 + P2 + P3; P1 has 8 significant bits, so kf*P1 is exact for |kf| < 2^16
    float r = fmaf(-kf, 1.5703125f, x);
    r = fmaf(-kf, 4.837512969970703125e-4f, r);
    r = fmaf(-kf, 7.54978995489188e-8f, r);
    const float z = r * r;
    // minimax polynomials on [-pi/4, pi/4]
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    pc = fmaf(pc, z * z, fmaf(-0.5f, z, 1.0f));
    // quadrant q = kf mod 4: odd q swaps the two, sin changes sign for q in {2, 3}, cos for q in {1, 2}
    // (bit 1 of q, and bit 1 xor bit 0, moved to the sign position and xor-ed in: no compares beyond the swap's)
    const unsigned q = __builtin_bit_cast(unsigned, tm);   // low bits = the integer
    const unsigned b1 = q << 30, b0 = q << 31;             // bit 1 (and bit 0 below it) / bit 0 alone at the sign position
    const bool odd = (int)b0 < 0;
    const float s0 = odd ? pc : ps;
    const float c0 = odd ? ps : pc;
#if defined(__HIP_DEVICE_COMPILE__)
    // three-input bit operations: a ^ (b & c) and (a & b) ^ c
    *sn = __builtin_bit_cast(float, __builtin_amdgcn_bitop3_b32(__builtin_bit_cast(unsigned, s0), b1, 0x80000000u, 0x78));
    *cs = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c0) ^ __builtin_amdgcn_bitop3_b32(b1, 0x80000000u, b0, 0x6A));
#else
    *sn = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, s0) ^ (b1 & 0x80000000u));
    *cs = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c0) ^ ((b1 & 0x80000000u) ^ b0));
#endif
}

// Table form: x = k*(2 pi/256) + r with |r| <= pi/256, (S, C) = (sin, cos) of k*(2 pi/256) from a 256-entry table
// (sincos_table.h; in LDS on the device), and with w = 1 - r^2/6 (sin r = r w, next term < 3e-12) and
// cos r - 1 = -r^2/2 (next term < 1e-9):
//     sin x = S + r (C w - S r/2),   cos x = C - r (S w + C r/2).
// No quadrant logic, no selects: 13 floating-point operations that pair up into packed instructions, one integer
// operation for the table index and one table read per value, against 15 + 8 unpackable ones of the polynomial
// form above -- in the sine epilogues of the fused kernels every vector instruction is matrix-pipe time.
// Reduction: 2 pi/256 = P1 + P2 with P1 the nearest fp32.  x - k*P1 comes out of ONE fused multiply-add exactly (it
// is a multiple of 2^-29 below 2^-5 whatever the size of k), so two constants are enough and the error stays at
// 6.3e-8 absolute up to |x| = 1e5, where the rounding by magic-number addition runs out of integer bits
// (tests/host/sincos_sweep.cpp sweeps both forms against double precision).
// `tab` holds NERF_SINCOS_TABLE_ENTRIES (sin, cos) pairs.
// SIN_AT(k), COS_AT(k): where the table keeps its entries (interleaved pairs on the host, two arrays in LDS).
template <int STRIDE, int COS_OFFSET>
NERF_HD void nerf_sincosf_tab_impl(float x, const float* tab, float* sn, float* cs) {
#if defined(__clang__)
#pragma clang fp contract(off)   // the operations as written (the sequence the host sweep checks), on every compiler
#endif
    const float tm = fmaf(x, 40.743665431525205956834243423364f, 12582912.0f);   // x * 256/(2 pi), rounded to an integer
    const float kf = tm - 12582912.0f;
    float r = fmaf(-kf, 0.024543693289160728f, x);        // exact
    r = fmaf(-kf, -6.8299044198028014e-10f, r);
    const unsigned k = __builtin_bit_cast(unsigned, tm) & 255u;   // low mantissa bits = the integer (two's complement)
    const float S = tab[STRIDE * k], C = tab[STRIDE * k + COS_OFFSET];
    const float z = r * r;
    const float w = fmaf(z, -0.16666667f, 1.0f);
    const float hr = r * 0.5f;
    *sn = fmaf(r, fmaf(-S, hr, C * w), S);
    *cs = fmaf(-r, fmaf(C, hr, S * w), C);
}
// table of 256 interleaved (sin, cos) pairs (sincos_table.h as it stands)
NERF_HD void nerf_sincosf_tab(float x, const float* tab, float* sn, float* cs) {
    nerf_sincosf_tab_impl<2, 1>(x, tab, sn, cs);
}
// table as two arrays, sin[256] then cos[256] (the LDS copy of the fused kernels)
NERF_HD void nerf_sincosf_tab_soa(float x, const float* tab, float* sn, float* cs) {
    nerf_sincosf_tab_impl<1, 256>(x, tab, sn, cs);
}

NERF_HD void nerf_sincosf(float x, float* sn, float* cs) {
    if (!(fabsf(x) <= 1.0e5f)) {  // also catches NaN
        sincosf(x, sn, cs);
        return;
    }
    nerf_sincosf_bounded(x, sn, cs);
}
