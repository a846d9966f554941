// Memory layouts shared by the MLP kernels: the packed weight image, the forward
// activation workspace and the backward workspace, for the two architectures the
// reference's callers build (nerf/models.py NeRF() and Siren() with default arguments).
//
// Weight image.  A dense layer y = x W^T (W [N,K] row-major, the nn.Linear layout) is
// streamed by the MFMA loop as 8-deep K blocks: img[(kb*N + n)*8 + j] = W[n][8*kb + j]
// (zero beyond K).  A lane (n, h) of a wave then fetches its four B values of one K block
// with ONE 16-byte load at img + (kb*N + n)*8 + 4*h, and a wave fetches 1 KiB contiguous.
// The data-gradient pass needs dX = dY W, i.e. the same loop with the roles of the two
// indices exchanged, so a second, transposed image timg[(ob*K + k)*8 + j] = W[8*ob + j][k]
// is kept for every layer whose input gradient is needed.
#pragma once
#include <stddef.h>

namespace lay {

constexpr int H = 256;        // hidden width
constexpr int PE = 63;        // 3 + 6*10 position encoding width
constexpr int PE_P = 64;      // padded to a multiple of 8
constexpr int DE = 27;        // 3 + 6*4 direction encoding width
constexpr int DE_P = 32;
constexpr int HH = 128;       // rgb-head hidden width
constexpr int TILE = 64;      // samples per workgroup tile

constexpr size_t img(int K_padded, int N) { return (size_t)K_padded * N; }

// ----------------------------------------------------------------------------- NeRF
namespace nerf {
// parameter tensor t of the state_dict: layer = t/2, weight if even, bias if odd.
// layers 0-4 block1, 5-9 block2, 10 rgb_head.0, 11 rgb_head.2
constexpr int NLAYER = 12;
constexpr int OUT[NLAYER] = {H, H, H, H, H, H, H, H, H, H + 1, HH, 3};
constexpr int IN[NLAYER] = {PE, H, H, H, H, H + PE, H, H, H, H, H + DE, HH};
constexpr size_t param_off(int tensor) {
    size_t o = 0;
    for (int t = 0; t < tensor; ++t) o += (t & 1) ? (size_t)OUT[t / 2] : (size_t)OUT[t / 2] * IN[t / 2];
    return o;
}
constexpr size_t PARAM_FLOATS = param_off(24);
static_assert(PARAM_FLOATS == 661636, "NeRF parameter count");

// ---- packed image offsets (floats)
constexpr size_t F_L0 = 0;                                   // K 64, N 256
constexpr size_t F_L1 = F_L0 + img(PE_P, H);                 // L1..L4: K 256, N 256 each
constexpr size_t F_L5A = F_L1 + 4 * img(H, H);               // features part of the skip layer
constexpr size_t F_L5B = F_L5A + img(H, H);                  // encoding part (K 64)
constexpr size_t F_L6 = F_L5B + img(PE_P, H);                // L6..L8
constexpr size_t F_L9 = F_L6 + 3 * img(H, H);                // rows 0..255 of block2.8
constexpr size_t F_L10A = F_L9 + img(H, H);                  // rgb_head.0, feature part: K 256, N 128
constexpr size_t F_L10B = F_L10A + img(H, HH);               // direction part: K 32, N 128
constexpr size_t SIGW = F_L10B + img(DE_P, HH);              // row 256 of block2.8 (density), 256 floats
constexpr size_t W11 = SIGW + H;                             // rgb_head.2 weight [3][128]
constexpr size_t BIAS = W11 + 3 * HH;                        // b0..b8 (9*256), b9[0:256], then:
constexpr size_t BIAS_SIG = BIAS + 10 * H;                   // b9[256] (padded to 4)
constexpr size_t BIAS10 = BIAS_SIG + 4;                      // 128
constexpr size_t BIAS11 = BIAS10 + HH;                       // 3 (padded to 4)
constexpr size_t D10 = BIAS11 + 4;                           // transposed: reduce over 128 outs, N 256
constexpr size_t D9 = D10 + img(HH, H);                      // D9, D8, D7, D6, D5, D4, D3, D2, D1
constexpr size_t PACKED_FLOATS = D9 + 9 * img(H, H);
constexpr size_t d_img(int layer) { return D9 + (size_t)(9 - layer) * img(H, H); }  // layer 1..9

// ---- forward workspace (floats per padded sample), rows of the padded sample count Mp
// pe [64] | de [32] | h0..h8 [9][256] | f [256] | g [128] | sraw [1] | relu masks [9][8]
// The masks are the (h > 0) bits of h0..h8 in the accumulator layout of the tile kernels: per
// layer and 64-sample tile 256 lanes x one 64-bit word (bit (2*mi+ni)*16+g of lane l = element
// acc[mi][ni][g] of that lane), i.e. 8 floats' worth per sample.  The data-gradient kernel reads
// them with ONE 8-byte load per lane and layer instead of re-reading h (1 KB per sample).
constexpr size_t WS_PE = 0, WS_DE = 64, WS_H = 96, WS_F = WS_H + 9 * H, WS_G = WS_F + H, WS_SRAW = WS_G + HH,
                 WS_MASK = WS_SRAW + 1;
constexpr size_t WS_PER_SAMPLE = WS_MASK + 9 * 8;
// ---- backward workspace: dz0..dz8 [9][256] | dz9 [256] | dz10 [128] | dz11 [4] | dsig [1]
constexpr size_t BW_DZ = 0, BW_DZ9 = 9 * H, BW_DZ10 = BW_DZ9 + H, BW_DZ11 = BW_DZ10 + HH, BW_DSIG = BW_DZ11 + 4;
constexpr size_t BW_PER_SAMPLE = BW_DSIG + 1;
}  // namespace nerf

// ----------------------------------------------------------------------------- Siren
namespace siren {
// layers 0-7 block1.i.layer, 8 density_branch.0, 9 feature_remap.0, 10 rgb_head.0.layer, 11 rgb_head.1
constexpr int NLAYER = 12;
constexpr int OUT[NLAYER] = {H, H, H, H, H, H, H, H, 1, H, HH, 3};
constexpr int IN[NLAYER] = {3, H, H, H, H, H, H, H, H, H, H + DE, HH};
constexpr size_t param_off(int tensor) {
    size_t o = 0;
    for (int t = 0; t < tensor; ++t) o += (t & 1) ? (size_t)OUT[t / 2] : (size_t)OUT[t / 2] * IN[t / 2];
    return o;
}
constexpr size_t PARAM_FLOATS = param_off(24);
static_assert(PARAM_FLOATS == 564356, "Siren parameter count");

constexpr size_t W0 = 0;                                     // [256][3] plain (padded to 1024)
constexpr size_t F_L1 = 1024;                                // L1..L7: K 256, N 256 each
constexpr size_t F_FEAT = F_L1 + 7 * img(H, H);              // feature_remap
constexpr size_t F_RGBA = F_FEAT + img(H, H);                // rgb sine layer, feature part: K 256, N 128
constexpr size_t F_RGBB = F_RGBA + img(H, HH);               // direction part: K 32, N 128
constexpr size_t SIGW = F_RGBB + img(DE_P, HH);              // density_branch weight, 256
constexpr size_t WOUT = SIGW + H;                            // rgb_head.1 weight [3][128]
constexpr size_t BIAS = WOUT + 3 * HH;                       // b0..b7 (8*256), bfeat (256)
constexpr size_t BIAS_SIG = BIAS + 9 * H;                    // 1 (padded to 4)
constexpr size_t BIAS_RGB = BIAS_SIG + 4;                    // 128
constexpr size_t BIAS_OUT = BIAS_RGB + HH;                   // 3 (padded to 4)
constexpr size_t D_RGB = BIAS_OUT + 4;                       // transposed rgb layer: reduce over 128, N 256
constexpr size_t D_FEAT = D_RGB + img(HH, H);                // transposed feature_remap
constexpr size_t D_L7 = D_FEAT + img(H, H);                  // transposed L7, L6, ..., L1
constexpr size_t PACKED_FLOATS = D_L7 + 7 * img(H, H);
constexpr size_t d_img(int layer) { return D_L7 + (size_t)(7 - layer) * img(H, H); }  // layer 1..7

// forward workspace per sample: xn [4] | de [32] | h0..h7 [8][256] | c0..c7 [8][256] (w0*cos) |
// f [256] | g [128] | cg [128] | sraw [1]
// c0..c7 are not row-major: each tile's region of a layer holds the derivative in the accumulator order of
// the tile kernels (16-byte pieces per thread, see fwd_epilogue); only the data-gradient kernel reads them.
constexpr size_t WS_X = 0, WS_DE = 4, WS_H = 36, WS_C = WS_H + 8 * H, WS_F = WS_C + 8 * H, WS_G = WS_F + H,
                 WS_CG = WS_G + HH, WS_SRAW = WS_CG + HH;
constexpr size_t WS_PER_SAMPLE = WS_SRAW + 1;
// backward workspace: dz0..dz7 [8][256] | dzf [256] | dzr [128] | dzo [4] | dsig [1]
constexpr size_t BW_DZ = 0, BW_DZF = 8 * H, BW_DZR = BW_DZF + H, BW_DZO = BW_DZR + HH, BW_DSIG = BW_DZO + 4;
constexpr size_t BW_PER_SAMPLE = BW_DSIG + 1;
}  // namespace siren

// Weight-gradient partial slabs.  Every per-layer launch owns a region [G][outputs] of the slab area (G <=
// WG_MAX workgroups for the MFMA launches, <= THIN_MAX for the thin VALU products), so that ONE reduction launch
// can add all of them at the end.  Per G the MFMA launches of the larger network (NeRF) write
// 2*(256*64) + 9*(256*256) + 128*256 + 128*32 weight partials and 10*256 + 128 + 256 + 4 bias / rank-1
// partials = 662,404 floats; the thin products at most 3*256 + 256 + 4 floats.
constexpr int WG_MAX = 256;
constexpr int THIN_MAX = 1024;   // 512 left the thin products at 3.5 TB/s, 2048+ only lengthens the reduction
constexpr size_t SLAB_FLOATS = (size_t)WG_MAX * 664000 + (size_t)THIN_MAX * 1100;

inline size_t padded_samples(long long M) { return (size_t)((M + TILE - 1) / TILE) * TILE; }

}  // namespace lay
