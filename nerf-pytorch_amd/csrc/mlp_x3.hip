// Entry points of the opt-in TRAINING mode of the NeRF MLP (nerf/models.py:62-75) and the Siren MLP (:185-203), with
// what autograd derives for them (train.py:181), on the bf16 matrix pipe at fp32 accuracy ("bf16x3"): forward with the
// activation workspace, and the data-gradient chain.  The default training path stays the fp32 MFMA kernels of
// mlp_fwd.hip / mlp_bwd.hip; the weight-gradient GEMMs read the same workspaces (nerf_mlp_bwd_weights on fp32 MFMA, or
// nerf_mlp_bwd_weights_x3 with this mode's arithmetic).
//
// Arithmetic: every fp32 operand is the exact sum of three bf16 terms, x = x0 + x1 + x2 with x0 = bf16(x),
// x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); a product of two operands is formed from the six term products that reach
// fp32 precision, w*x ~= w0 x0 + w0 x1 + w0 x2 + w1 x0 + w1 x1 + w2 x0, each exact in the MFMA's fp32 accumulator: 0.375
// of the matrix-pipe time of fp32 MFMA, the rounding of an fp32 accumulation and nothing else.
//
// Kernels: mlp_x3i.hip.  Both chains of both models run on the structure of the three-term inference kernel --
// 16 samples per wave, 128 per workgroup, the layer input (activations forward, layer gradients backward) chained in
// registers as the B operand, the weights (transposed for the data gradient) streamed once per workgroup through an LDS
// ring -- with the workspace rows written from the registers at the end of each layer.  The kernels this replaced kept
// 64-sample tiles with the layer input in LDS as three bf16 planes, every wave fetching its own weight fragments through
// the vector cache, and two workgroup barriers per layer: forward 7.45 ms and data gradient 6.2 ms per 1,048,576 samples
// of the NeRF model against 5.9 and 5.45 ms.
//
// Image of nerf_mlp_pack_x3: the fp32 image of nerf_mlp_pack (thin vectors for the weight gradient's callers, and so
// that the same pointer serves a pass without a workspace), the forward's ring image (that of nerf_mlp_pack_bf16x3),
// the data-gradient chain's ring image.
#include "common.h"
#include "mlp_layout.h"

// mlp_x3i.hip
size_t nerf_x3i_dgrad_image_bytes(int model);
int nerf_x3i_dgrad_pack(int model, const float* const* params, void* image, void* stream);
int nerf_x3i_dgrad(int model, const void* image, const float* wout, int64_t M, const float* dcolors, const float* ddensity,
                   const float* colors, const float* fwd_workspace, float* bwd_workspace, void* stream);
int nerf_x3i_fwd_train(int model, const void* image, int64_t M, const float* points, const float* dirs, const float* rays_o,
                       const float* rays_d, const float* t, int S, int t_per_ray, double near, double far, float* colors,
                       float* density, float* workspace, void* stream);

namespace {
size_t fp32_floats(int model) { return model == NERF_MODEL_NERF ? lay::nerf::PACKED_FLOATS : lay::siren::PACKED_FLOATS; }
}  // namespace

extern "C" size_t nerf_mlp_packed_x3_floats(int model) {
    if (model != NERF_MODEL_NERF && model != NERF_MODEL_SIREN) return 0;
    static_assert(lay::nerf::PACKED_FLOATS % 4 == 0 && lay::siren::PACKED_FLOATS % 4 == 0, "16-byte pieces of the ring images");
    return fp32_floats(model) + (nerf_mlp_packed_bf16x3_bytes(model) + nerf_x3i_dgrad_image_bytes(model)) / 4;
}

extern "C" int nerf_mlp_pack_x3(int model, const float* const* params, float* packed, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_pack_x3: unknown model %d", model);
    NERF_REQUIRE_PTR(params);
    NERF_REQUIRE_PTR(packed);
    int rc = nerf_mlp_pack(model, params, packed, stream);
    if (rc) return rc;
    float* fwd_img = packed + fp32_floats(model);
    if ((rc = nerf_mlp_pack_bf16x3(model, params, fwd_img, stream))) return rc;
    return nerf_x3i_dgrad_pack(model, params, fwd_img + nerf_mlp_packed_bf16x3_bytes(model) / 4, stream);
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
    return nerf_x3i_fwd_train(model, packed + fp32_floats(model), M, points, dirs, rays_o, rays_d, t, S, t_per_ray, near, far,
                              colors, density, workspace, stream);
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
    // the output layer's weight [3][128] in the fp32 image
    const float* wout = packed + (model == NERF_MODEL_NERF ? lay::nerf::W11 : lay::siren::WOUT);
    return nerf_x3i_dgrad(model, packed + fp32_floats(model) + nerf_mlp_packed_bf16x3_bytes(model) / 4, wout, M, dcolors, ddensity,
                          colors, fwd_workspace, bwd_workspace, stream);
}
