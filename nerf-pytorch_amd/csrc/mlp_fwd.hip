// Weight packing and the fused encode + MLP forward kernels.
// Replaces nerf/models.py NeRF.forward (:62-75) and Siren.forward (:185-203), with the
// sample-position generation and normalisation of nerf/rendering.py:59-62,82 folded in when
// the caller passes rays instead of points.
#include "mlp_device.h"

using namespace dev;

// =====================================================================================
// packing
// =====================================================================================
namespace {

struct PackJob {
    int tensor;        // index into the 24 parameter pointers
    int src_off;       // element offset inside the tensor
    int sn, sk;        // strides of the image's N and K index in the source
    int N, K, KB;      // KB == 0: plain copy of N floats
    unsigned dst_off;  // float offset inside the packed image
    int neg;           // store 0 - value: the layers whose epilogue is a ReLU run negated (fwd_epilogue)
};
constexpr int kMaxPackJobs = 48;
struct PackArgs {
    const float* src[NERF_NUM_PARAM_TENSORS];
    float* dst;
    int njobs;
    PackJob jobs[kMaxPackJobs];
};

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    const PackJob j = a.jobs[blockIdx.x];
    const float* __restrict__ src = a.src[j.tensor] + j.src_off;
    float* __restrict__ dst = a.dst + j.dst_off;
    if (j.KB == 0) {
        for (int i = blockIdx.y * 256 + threadIdx.x; i < j.N; i += gridDim.y * 256) dst[i] = j.neg ? 0.0f - src[i] : src[i];
        return;
    }
    const int total = j.KB * j.N * 8;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < total; i += gridDim.y * 256) {
        const int jj = i & 7, n = (i >> 3) % j.N, kb = (i >> 3) / j.N;
        const int k = kb * 8 + jj;
        const float v = (k < j.K) ? src[(size_t)n * j.sn + (size_t)k * j.sk] : 0.0f;
        dst[i] = j.neg ? 0.0f - v : v;     // (0 - v, not -v: zero weights, biases and the padding stay +0)
    }
}

void add_fwd(PackArgs& a, int tensor, int ld, int col0, int row0, int N, int K, size_t dst, bool neg = false) {
    // forward image of rows row0..row0+N, columns col0..col0+K of a [*, ld] weight
    a.jobs[a.njobs++] = PackJob{tensor, row0 * ld + col0, ld, 1, N, K, (K + 7) / 8, (unsigned)dst, neg};
}
void add_bwd(PackArgs& a, int tensor, int ld, int col0, int row0, int Nin, int Kout, size_t dst) {
    // transposed image: reduce over rows row0..row0+Kout, image columns = weight columns col0..col0+Nin
    a.jobs[a.njobs++] = PackJob{tensor, row0 * ld + col0, 1, ld, Nin, Kout, (Kout + 7) / 8, (unsigned)dst, 0};
}
void add_copy(PackArgs& a, int tensor, int off, int count, size_t dst, bool neg = false) {
    a.jobs[a.njobs++] = PackJob{tensor, off, 0, 0, count, 0, 0, (unsigned)dst, neg};
}

void build_nerf_jobs(PackArgs& a) {
    using namespace lay;
    using namespace lay::nerf;
    // the ten ReLU layers (L0..L8 and rgb_head.0) are packed negated, weights and biases: see fwd_epilogue
    add_fwd(a, 0, PE, 0, 0, H, PE, F_L0, true);
    for (int l = 1; l <= 4; ++l) add_fwd(a, 2 * l, H, 0, 0, H, H, F_L1 + (size_t)(l - 1) * img(H, H), true);
    add_fwd(a, 10, H + PE, 0, 0, H, H, F_L5A, true);
    add_fwd(a, 10, H + PE, H, 0, H, PE, F_L5B, true);
    for (int l = 6; l <= 8; ++l) add_fwd(a, 2 * l, H, 0, 0, H, H, F_L6 + (size_t)(l - 6) * img(H, H), true);
    add_fwd(a, 18, H, 0, 0, H, H, F_L9);
    add_fwd(a, 20, H + DE, 0, 0, HH, H, F_L10A, true);
    add_fwd(a, 20, H + DE, H, 0, HH, DE, F_L10B, true);
    add_copy(a, 18, H * H, H, SIGW);
    add_copy(a, 22, 0, 3 * HH, W11);
    for (int l = 0; l <= 8; ++l) add_copy(a, 2 * l + 1, 0, H, BIAS + (size_t)l * H, true);
    add_copy(a, 19, 0, H, BIAS + 9 * (size_t)H);
    add_copy(a, 19, H, 1, BIAS_SIG);
    add_copy(a, 21, 0, HH, BIAS10, true);
    add_copy(a, 23, 0, 3, BIAS11);
    add_bwd(a, 20, H + DE, 0, 0, H, HH, D10);
    add_bwd(a, 18, H, 0, 0, H, H, d_img(9));
    for (int l = 8; l >= 6; --l) add_bwd(a, 2 * l, H, 0, 0, H, H, d_img(l));
    add_bwd(a, 10, H + PE, 0, 0, H, H, d_img(5));
    for (int l = 4; l >= 1; --l) add_bwd(a, 2 * l, H, 0, 0, H, H, d_img(l));
}

void build_siren_jobs(PackArgs& a) {
    using namespace lay;
    using namespace lay::siren;
    add_copy(a, 0, 0, H * 3, W0);
    for (int l = 1; l <= 7; ++l) add_fwd(a, 2 * l, H, 0, 0, H, H, F_L1 + (size_t)(l - 1) * img(H, H));
    add_fwd(a, 18, H, 0, 0, H, H, F_FEAT);
    add_fwd(a, 20, H + DE, 0, 0, HH, H, F_RGBA);
    add_fwd(a, 20, H + DE, H, 0, HH, DE, F_RGBB);
    add_copy(a, 16, 0, H, SIGW);
    add_copy(a, 22, 0, 3 * HH, WOUT);
    for (int l = 0; l <= 7; ++l) add_copy(a, 2 * l + 1, 0, H, BIAS + (size_t)l * H);
    add_copy(a, 19, 0, H, BIAS + 8 * (size_t)H);
    add_copy(a, 17, 0, 1, BIAS_SIG);
    add_copy(a, 21, 0, HH, BIAS_RGB);
    add_copy(a, 23, 0, 3, BIAS_OUT);
    add_bwd(a, 20, H + DE, 0, 0, H, HH, D_RGB);
    add_bwd(a, 18, H, 0, 0, H, H, D_FEAT);
    for (int l = 7; l >= 1; --l) add_bwd(a, 2 * l, H, 0, 0, H, H, d_img(l));
}

}  // namespace

extern "C" size_t nerf_mlp_param_floats(int model) {
    return model == NERF_MODEL_NERF ? lay::nerf::PARAM_FLOATS : model == NERF_MODEL_SIREN ? lay::siren::PARAM_FLOATS : 0;
}
extern "C" size_t nerf_mlp_packed_floats(int model) {
    return model == NERF_MODEL_NERF ? lay::nerf::PACKED_FLOATS : model == NERF_MODEL_SIREN ? lay::siren::PACKED_FLOATS : 0;
}
extern "C" size_t nerf_mlp_fwd_workspace_floats(int model, int64_t M) {
    if (M < 0) return 0;
    const size_t mp = lay::padded_samples(M);
    return model == NERF_MODEL_NERF ? mp * lay::nerf::WS_PER_SAMPLE : model == NERF_MODEL_SIREN ? mp * lay::siren::WS_PER_SAMPLE : 0;
}
extern "C" size_t nerf_mlp_bwd_workspace_floats(int model, int64_t M) {
    if (M < 0) return 0;
    const size_t mp = lay::padded_samples(M);
    const size_t per = model == NERF_MODEL_NERF ? lay::nerf::BW_PER_SAMPLE : model == NERF_MODEL_SIREN ? lay::siren::BW_PER_SAMPLE : 0;
    return per ? mp * per + lay::SLAB_FLOATS : 0;
}

extern "C" int nerf_mlp_pack(int model, const float* const* params, float* packed, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_pack: unknown model %d", model);
    NERF_REQUIRE_PTR(params);
    NERF_REQUIRE_PTR(packed);
    PackArgs a;
    for (int i = 0; i < NERF_NUM_PARAM_TENSORS; ++i) {
        if (!params[i]) return nerf_fail(NERF_E_NULL, "nerf_mlp_pack: params[%d] is NULL", i);
        a.src[i] = params[i];
    }
    a.dst = packed;
    a.njobs = 0;
    if (model == NERF_MODEL_NERF) build_nerf_jobs(a); else build_siren_jobs(a);
    pack_kernel<<<dim3(a.njobs, 32), 256, 0, (hipStream_t)stream>>>(a);
    return nerf_check_launch("nerf_mlp_pack");
}

// =====================================================================================
// fused forward
// =====================================================================================
namespace {

struct FwdArgs {
    const float* pk;
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
#ifdef NERF_DIAG
    unsigned long long* dbg;   // [tile][32]: 100 MHz real-time stamps of wave 0; [30] = XCC id, [31] = HW_ID
#endif
};

#ifdef NERF_DIAG
#define DIAG_STAMP(i)                                                                                   \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (a.dbg && threadIdx.x == 0) a.dbg[(size_t)blockIdx.x * 32 + (i)] = wall_clock64();           \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#else
#define DIAG_STAMP(i) ((void)0)
#endif

constexpr int XIN_FLOATS = TR * 8;
constexpr int FWD_LDS_BYTES = (ACT_FLOATS + XIN_FLOATS + TR) * 4;
constexpr int TAB_FWD_LDS_BYTES = FWD_LDS_BYTES + SINCOS_TAB_FLOATS * 4;     // + the sine/cosine table (both networks)
constexpr int SIREN_FWD_LDS_BYTES = TAB_FWD_LDS_BYTES;

__device__ __forceinline__ float norm_coord(float p, float nearf, float spanf) {
#pragma clang fp contract(off)  // each operation rounded on its own, as the reference's tensor ops are
    return sub_rn(div_rn(mul_rn(2.0f, sub_rn(p, nearf)), spanf), 1.0f);
}

// xin[s][0..2] = normalised position, xin[s][3..5] = direction (zeros beyond M)
__device__ __forceinline__ void load_inputs(const FwdArgs& a, size_t m0, float* xin, int tid) {
#pragma clang fp contract(off)  // o + t*d must not become an FMA (bit-exact positions)
    if (tid < TR) {
        const size_t m = m0 + tid;
        float x[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
        if (m < (size_t)a.M) {
            if (a.points) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { x[c] = a.points[m * 3 + c]; d[c] = a.dirs[m * 3 + c]; }
            } else {
                const size_t r = m / (size_t)a.S;
                const float ts = a.t_per_ray ? a.t[m] : a.t[m - r * (size_t)a.S];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    d[c] = a.rays_d[r * 3 + c];
                    x[c] = norm_coord(add_rn(a.rays_o[r * 3 + c], mul_rn(ts, d[c])), a.nearf, a.spanf);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { xin[tid * 8 + c] = x[c]; xin[tid * 8 + 3 + c] = d[c]; }
        // xin[6] (a padding column of row 0) = the tile's largest |position coordinate|: lets the first Siren layer
        // bound its sine arguments once per tile.  Infinities survive the maximum; a NaN coordinate does not, and
        // does not need to (NaN in, NaN out on either sine path).
        float big = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fabsf(x[2]));
#pragma unroll
        for (int off = TR / 2; off >= 1; off >>= 1) big = fmaxf(big, __shfl_xor(big, off, TR));
        // xin[7]: the same for the direction components (the table form of the direction encoding has a range too)
        float bigd = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fabsf(d[2]));
#pragma unroll
        for (int off = TR / 2; off >= 1; off >>= 1) bigd = fmaxf(bigd, __shfl_xor(bigd, off, TR));
        if (tid == 0) { xin[6] = big; xin[7] = bigd; }
    }
}

// TPS threads per sample: one per axis (64-row tiles), or two per axis that split the frequencies
// (32-row tiles); the next thread zeroes the padding columns
constexpr int ENC_SPLIT = TPS / 4;   // 1 or 2 threads per axis
__device__ __forceinline__ void encode_pe(float* act, const float* xin, int tid) {
    const int smp = tid / TPS, q = tid % TPS;
    if (q < 3 * ENC_SPLIT) {
        const int axis = q % 3, part = q / 3;
        encode_axis(act + smp * LDA, axis, xin[smp * 8 + axis], part * (10 / ENC_SPLIT), (part + 1) * (10 / ENC_SPLIT));
    } else if (q == 3 * ENC_SPLIT) act[smp * LDA + 63] = 0.0f;
}
__device__ __forceinline__ void encode_de(float* act, const float* xin, int tid) {
    const int smp = tid / TPS, q = tid % TPS;
    if (q < 3 * ENC_SPLIT) {
        const int axis = q % 3, part = q / 3;
        encode_axis(act + smp * LDA, axis, xin[smp * 8 + 3 + axis], part * (4 / ENC_SPLIT), (part + 1) * (4 / ENC_SPLIT));
    } else if (q == 3 * ENC_SPLIT) {
#pragma unroll
        for (int c = lay::DE; c < lay::DE_P; ++c) act[smp * LDA + c] = 0.0f;
    }
}

// Position / direction encoding of the tile: the table form when the tile's arguments allow it (tile-uniform decision from
// the largest |coordinate| that load_inputs left in xin[6] / xin[7]; NaN and infinite coordinates fail the comparison or,
// swallowed by the maximum, give NaN on either path), the full-range form otherwise.  `fast` is wave-uniform.
__device__ __forceinline__ bool enc_fast(const float* xin, int slot, float top_scale) {
    const float big = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xin[slot])));
    return TPS == 4 && big * top_scale <= 9.9e4f;
}

// dot of each tile row (first 256 columns) with a 256-vector; TPS threads per sample
__device__ __forceinline__ float row_dot256(const float* act, const float* __restrict__ w, int tid) {
    const int smp = tid / TPS, q = tid % TPS;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64 / TPS; ++i) {
        const int k = 4 * (TPS * i + q);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(act + smp * LDA + k);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
        s += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if constexpr (TPS == 8) s += __shfl_xor(s, 4);
    return s;
}

// 128 -> 3 output layer + sigmoid; TPS threads per sample over act[:, 0:128]
__device__ __forceinline__ void head_out(const float* act, const float* __restrict__ w, const float* __restrict__ b,
                                         float rgb_mul, float dens, size_t m0, int64_t M, float* __restrict__ colors,
                                         float* __restrict__ density, int tid) {
    const int smp = tid / TPS, q = tid % TPS;
    float z[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32 / TPS; ++i) {
        const int k = 4 * (TPS * i + q);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(act + smp * LDA + k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c * lay::HH + k);
            z[c] += gv[0] * wv[0] + gv[1] * wv[1] + gv[2] * wv[2] + gv[3] * wv[3];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        z[c] += __shfl_xor(z[c], 1);
        z[c] += __shfl_xor(z[c], 2);
        if constexpr (TPS == 8) z[c] += __shfl_xor(z[c], 4);
    }
    const size_t m = m0 + smp;
    if (q == 0 && m < (size_t)M) {
#pragma unroll
        for (int c = 0; c < 3; ++c) colors[m * 3 + c] = sigmoidf_((z[c] + b[c]) * rgb_mul);
        density[m] = dens;
    }
}


// one IN_COLS -> 256 layer on the tile: bias, MFMA loop, barrier, epilogue, barrier.
// Training: the INPUT tile of the layer (the previous layer's output, IN_COLS columns) is copied to
// the workspace by the MFMA loop that consumes it (mma_layer's COPY_COLS), piece by piece between
// the MFMA blocks.
template <int ACT, bool TRAIN, int IN_COLS>
__device__ __forceinline__ void hidden_layer(float* act, const float* __restrict__ img,
                                             const float* __restrict__ bias, int ncol0, int tid, float* gin,
                                             float* gcos, mask_t* gmask_in = nullptr,
                                             mask_t* mask = nullptr, unsigned long long* stamps = nullptr) {
    const int lane = tid & 63;
    f32x16 acc[MT][2];
#ifdef NERF_DIAG
#define LAYER_STAMP(i)                                                         \
    do {                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                     \
        if (stamps && tid == 0) stamps[i] = wall_clock64();                    \
        __builtin_amdgcn_sched_barrier(0);                                     \
    } while (0)
#else
#define LAYER_STAMP(i) ((void)0)
#endif
    LAYER_STAMP(0);
    mma_layer<2, TRAIN ? IN_COLS : 0, ACC_BIAS>(act, img, lay::H, IN_COLS / 8, ncol0, lane, acc, gin, tid, bias);
    if constexpr (TRAIN) {
        if (gmask_in) *gmask_in = *mask;   // ReLU mask word of the input tile's layer, kept in registers until now
    }
    LAYER_STAMP(1);
    LAYER_SYNC();
    LAYER_STAMP(2);
    fwd_epilogue<2, ACT, TRAIN>(act, acc, ncol0, lane, gcos, lay::H, TRAIN ? mask : nullptr);
    LAYER_STAMP(3);
    LAYER_SYNC();
    LAYER_STAMP(4);
}

// DENSITY_ONLY (inference): stops after the density row of block2.8 -- the coarse pass of a hierarchical render only feeds
// the re-sampling, which needs no colors (the feature layer, the 288 -> 128 layer and the head are 16 % of the network)
template <bool TRAIN, bool DENSITY_ONLY = false>
__global__ __launch_bounds__(256, WG_PER_CU) void nerf_fwd_kernel(FwdArgs a) {
    static_assert(!(TRAIN && DENSITY_ONLY), "the density-only pass is an inference variant");
    using namespace lay;
    using namespace lay::nerf;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem + SINCOS_TAB_FLOATS;   // the sine/cosine table of the encodings occupies the first 2 KiB
    float* xin = act + ACT_FLOATS;
    float* sraw_s = xin + XIN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t m0 = (size_t)blockIdx.x * TR;
    const float* __restrict__ pk = a.pk;
    float* ws = a.ws;
    const size_t Mp = a.Mp;
    auto wsp = [&](size_t off, int width) -> float* { return TRAIN ? ws + off * Mp + m0 * width : nullptr; };
    // this lane's ReLU mask word of layer l (training only)
    auto mskp = [&](int l) -> mask_t* {
        return TRAIN ? reinterpret_cast<mask_t*>(ws + (WS_MASK + 8 * (size_t)l) * Mp + m0 * 8) + tid : nullptr;
    };

#ifdef NERF_DIAG
    if (a.dbg && tid == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        a.dbg[(size_t)blockIdx.x * 32 + 30] = xcc & 0xf;
        a.dbg[(size_t)blockIdx.x * 32 + 31] = hwid;
    }
#endif
    DIAG_STAMP(0);
    load_sincos_table(smem, tid);
    load_inputs(a, m0, xin, tid);
    __syncthreads();
    DIAG_STAMP(1);
    // position encoding: computed once, kept in registers for its second use in the skip layer
    const bool pe_fast = enc_fast(xin, 6, 512.0f);
    EncCache<10> pec;
    if (pe_fast) {
        enc_compute<10>(xin + (tid / TPS) * 8, tid % TPS, pec);
        enc_store<10>(act + (tid / TPS) * LDA, tid % TPS, pec);
    } else {
        encode_pe(act, xin, tid);
    }
    __syncthreads();
    DIAG_STAMP(2);

    const int ncol0 = wave * 64;
    // L0: K 64 (input tile = the position encoding)
    mask_t mask = 0;   // ReLU bits of the layer just finished; stored after the next MFMA loop
    hidden_layer<ACT_RELU, TRAIN, PE_P>(act, pk + F_L0, pk + BIAS, ncol0, tid, wsp(WS_PE, PE_P), nullptr, nullptr, &mask);
    DIAG_STAMP(3);
    // L1..L4 (input tile = h(l-1))
#pragma unroll 1   // one copy of the layer body: unrolled, the kernel outgrows the instruction cache
    for (int l = 1; l <= 4; ++l) {
#ifdef NERF_DIAG
        unsigned long long* layer_stamps = (a.dbg && l == 2) ? a.dbg + (size_t)blockIdx.x * 32 + 20 : nullptr;
#else
        unsigned long long* layer_stamps = nullptr;
#endif
        hidden_layer<ACT_RELU, TRAIN, H>(act, pk + F_L1 + (size_t)(l - 1) * img(H, H), pk + BIAS + l * H, ncol0, tid,
                                         wsp(WS_H + (size_t)(l - 1) * H, H), nullptr, mskp(l - 1), &mask, layer_stamps);
        DIAG_STAMP(3 + l);
    }
    // L5: [h4 | pe] in two passes over the same accumulators
    {
        f32x16 acc[MT][2];
        mma_layer<2, TRAIN ? H : 0, ACC_BIAS>(act, pk + F_L5A, H, H / 8, ncol0, lane, acc, wsp(WS_H + 4 * (size_t)H, H), tid,
                                              pk + BIAS + 5 * H);
        if constexpr (TRAIN) *mskp(4) = mask;
        __syncthreads();
        DIAG_STAMP(8);
        if (pe_fast) enc_store<10>(act + (tid / TPS) * LDA, tid % TPS, pec);
        else encode_pe(act, xin, tid);
        __syncthreads();
        DIAG_STAMP(9);
        mma_layer<2>(act, pk + F_L5B, H, PE_P / 8, ncol0, lane, acc);
        __syncthreads();
        fwd_epilogue<2, ACT_RELU, false>(act, acc, ncol0, lane, nullptr, H, TRAIN ? &mask : nullptr);
        __syncthreads();
        DIAG_STAMP(10);
    }
    // L6..L8
#pragma unroll 1   // one copy of the layer body: unrolled, the kernel outgrows the instruction cache
    for (int l = 6; l <= 8; ++l) {
        hidden_layer<ACT_RELU, TRAIN, H>(act, pk + F_L6 + (size_t)(l - 6) * img(H, H), pk + BIAS + l * H, ncol0, tid,
                                         wsp(WS_H + (size_t)(l - 1) * H, H), nullptr, mskp(l - 1), &mask);
        DIAG_STAMP(5 + l);
    }
    if constexpr (DENSITY_ONLY) {
        const float s = row_dot256(act, pk + SIGW, tid) + pk[BIAS_SIG];
        const size_t m = m0 + tid / TPS;
        if (tid % TPS == 0 && m < (size_t)a.M) a.density[m] = fmaxf(s, 0.0f);
        return;
    }
    // L9: 256 feature columns on MFMA, density column on the VALU
    {
        f32x16 acc[MT][2];
        mma_layer<2, TRAIN ? H : 0, ACC_BIAS>(act, pk + F_L9, H, H / 8, ncol0, lane, acc, wsp(WS_H + 8 * (size_t)H, H), tid,
                                              pk + BIAS + 9 * H);
        const float s = row_dot256(act, pk + SIGW, tid) + pk[BIAS_SIG];
        if constexpr (TRAIN) *mskp(8) = mask;
        if (tid % TPS == 0) {
            sraw_s[tid / TPS] = s;
            if constexpr (TRAIN) ws[WS_SRAW * Mp + m0 + tid / TPS] = s;
        }
        __syncthreads();
        fwd_epilogue<2, ACT_NONE, false>(act, acc, ncol0, lane, nullptr, H);
        __syncthreads();
        DIAG_STAMP(14);
    }
    // L10: [f | de] -> 128, two passes
    {
        const int nc = wave * 32;
        f32x16 acc[MT][1];
        mma_layer<1, TRAIN ? H : 0, ACC_BIAS>(act, pk + F_L10A, HH, H / 8, nc, lane, acc, wsp(WS_F, H), tid, pk + BIAS10);
        __syncthreads();
        DIAG_STAMP(15);
        if (enc_fast(xin, 7, 8.0f)) {
            EncCache<4> dec;
            enc_compute<4>(xin + (tid / TPS) * 8 + 3, tid % TPS, dec);
            enc_store<4>(act + (tid / TPS) * LDA, tid % TPS, dec);
        } else {
            encode_de(act, xin, tid);
        }
        __syncthreads();
        DIAG_STAMP(16);
        mma_layer<1, TRAIN ? DE_P : 0>(act, pk + F_L10B, HH, DE_P / 8, nc, lane, acc, wsp(WS_DE, DE_P), tid);
        __syncthreads();
        fwd_epilogue<1, ACT_RELU, false>(act, acc, nc, lane, nullptr, HH);
        __syncthreads();
        DIAG_STAMP(17);
    }
    // L11 + sigmoid, density = relu(raw)
    head_out(act, pk + W11, pk + BIAS11, 1.0f, fmaxf(sraw_s[tid / TPS], 0.0f), m0, a.M, a.colors, a.density, tid);
    if constexpr (TRAIN) store_tile(act, HH, wsp(WS_G, HH), HH, tid);
    DIAG_STAMP(18);
}

template <bool TRAIN, bool DENSITY_ONLY = false>
__global__ __launch_bounds__(256, WG_PER_CU) void siren_fwd_kernel(FwdArgs a) {
    static_assert(!(TRAIN && DENSITY_ONLY), "the density-only pass is an inference variant");
    using namespace lay;
    using namespace lay::siren;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem + SINCOS_TAB_FLOATS;   // the sine/cosine table occupies the first 2 KiB (load_sincos_table)
    float* xin = act + ACT_FLOATS;
    float* sraw_s = xin + XIN_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t m0 = (size_t)blockIdx.x * TR;
    const float* __restrict__ pk = a.pk;
    float* ws = a.ws;
    const size_t Mp = a.Mp;
    auto wsp = [&](size_t off, int width) -> float* { return TRAIN ? ws + off * Mp + m0 * width : nullptr; };

    load_sincos_table(smem, tid);
    load_inputs(a, m0, xin, tid);
    __syncthreads();
    if constexpr (TRAIN) {
        if (tid < TR) {
            f32x4 v = {xin[tid * 8], xin[tid * 8 + 1], xin[tid * 8 + 2], 0.0f};
            *reinterpret_cast<f32x4*>(ws + WS_X * Mp + (m0 + tid) * 4) = v;
        }
    }
    // layer 0 (3 -> 256, sin(30 z)) on the VALU: thread = output column, 16 samples (8 value pairs of the table-form
    // sine/cosine) in flight at a time.  The table form holds for |30 z| <= 1e5; whether this thread's column can
    // leave that range is decided once per tile from the largest |coordinate| of the tile (xin[6], written above),
    // not per value: |30 z| <= 30 ((|w0| + |w1| + |w2|) max|x| + |b|).
    {
        const float w0 = pk[W0 + tid * 3], w1 = pk[W0 + tid * 3 + 1], w2 = pk[W0 + tid * 3 + 2], b = pk[BIAS + tid];
        float* gc = wsp(WS_C, H);
        const float bound = 30.0f * ((fabsf(w0) + fabsf(w1) + fabsf(w2)) * xin[6] + fabsf(b));
        // the derivative tile in the accumulator order the data-gradient kernel reads (sine_epilogue): element
        // (row s, column tid) is register g of lane (tid & 31) + 32*h of wave tid / 64; four consecutive rows
        // s = 4j .. 4j+3 are the four floats of one 16-byte piece
        auto piece_of = [&](int s) {
            const int mi = s >> 5, rr = s & 31, hh = (rr >> 2) & 1, g = 4 * (rr >> 3);
            const int owner = (tid >> 6) * 64 + (tid & 31) + 32 * hh, ni = (tid >> 5) & 1;
            return gc + (((mi * 2 + ni) * 4 + (g >> 2)) * 256 + owner) * 4;
        };
        auto pre = [&](int s) { return mul_rn(30.0f, fmaf(xin[s * 8 + 2], w2, fmaf(xin[s * 8 + 1], w1, fmaf(xin[s * 8], w0, b)))); };
        if (bound <= 9.9e4f) {
#pragma unroll 1
            for (int s0 = 0; s0 < TR; s0 += 16) {
                SinCosPair pr[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) sincos_tab_issue((f32x2){pre(s0 + 2 * q), pre(s0 + 2 * q + 1)}, pr[q]);
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    f32x2 sa, ca, sb, cb;
                    sincos_tab_finish(pr[q], sa, ca);
                    sincos_tab_finish(pr[q + 1], sb, cb);
                    const int s = s0 + 2 * q;
                    act[s * LDA + tid] = sa[0];
                    act[(s + 1) * LDA + tid] = sa[1];
                    act[(s + 2) * LDA + tid] = sb[0];
                    act[(s + 3) * LDA + tid] = sb[1];
                    if constexpr (TRAIN)
                        *reinterpret_cast<f32x4*>(piece_of(s)) = f32x4{30.0f * ca[0], 30.0f * ca[1], 30.0f * cb[0], 30.0f * cb[1]};
                }
            }
        } else {   // (also NaN bounds) value by value through the library's full-range path where needed
#pragma unroll 1
            for (int s = 0; s < TR; ++s) {
                const float arg = pre(s);
                float sn, cs;
                if (fabsf(arg) <= 1.0e5f) nerf_sincosf_tab_soa(arg, smem, &sn, &cs);
                else sincosf(arg, &sn, &cs);
                act[s * LDA + tid] = sn;
                if constexpr (TRAIN) piece_of(s & ~3)[s & 3] = 30.0f * cs;
            }
        }
    }
    __syncthreads();
    const int ncol0 = wave * 64;
#pragma unroll 1   // one copy of the layer body: unrolled, the kernel outgrows the instruction cache
    for (int l = 1; l <= 7; ++l)
        hidden_layer<ACT_SIN, TRAIN, H>(act, pk + F_L1 + (size_t)(l - 1) * img(H, H), pk + BIAS + l * H, ncol0, tid,
                                        wsp(WS_H + (size_t)(l - 1) * H, H), wsp(WS_C + (size_t)l * H, H));
    if constexpr (DENSITY_ONLY) {      // density branch alone: no feature_remap, no rgb layers
        const float s = row_dot256(act, pk + SIGW, tid) + pk[BIAS_SIG];
        const size_t m = m0 + tid / TPS;
        if (tid % TPS == 0 && m < (size_t)a.M) a.density[m] = fmaxf(s, 0.0f) * 10.0f;
        return;
    }
    // feature_remap on MFMA, density branch on the VALU (both read h7)
    {
        f32x16 acc[MT][2];
        mma_layer<2, TRAIN ? H : 0, ACC_BIAS>(act, pk + F_FEAT, H, H / 8, ncol0, lane, acc, wsp(WS_H + 7 * (size_t)H, H), tid,
                                              pk + BIAS + 8 * H);
        const float s = row_dot256(act, pk + SIGW, tid) + pk[BIAS_SIG];
        if (tid % TPS == 0) {
            sraw_s[tid / TPS] = s;
            if constexpr (TRAIN) ws[WS_SRAW * Mp + m0 + tid / TPS] = s;
        }
        __syncthreads();
        fwd_epilogue<2, ACT_NONE, false>(act, acc, ncol0, lane, nullptr, H);
        __syncthreads();
    }
    {
        const int nc = wave * 32;
        f32x16 acc[MT][1];
        mma_layer<1, TRAIN ? H : 0, ACC_BIAS>(act, pk + F_RGBA, HH, H / 8, nc, lane, acc, wsp(WS_F, H), tid, pk + BIAS_RGB);
        __syncthreads();
        encode_de(act, xin, tid);
        __syncthreads();
        mma_layer<1, TRAIN ? DE_P : 0>(act, pk + F_RGBB, HH, DE_P / 8, nc, lane, acc, wsp(WS_DE, DE_P), tid);
        __syncthreads();
        fwd_epilogue<1, ACT_SIN, TRAIN>(act, acc, nc, lane, wsp(WS_CG, HH), HH);
        __syncthreads();
    }
    head_out(act, pk + WOUT, pk + BIAS_OUT, 1.0f, fmaxf(sraw_s[tid / TPS], 0.0f) * 10.0f, m0, a.M, a.colors, a.density,
             tid);
    if constexpr (TRAIN) store_tile(act, HH, wsp(WS_G, HH), HH, tid);
}

#ifdef NERF_DIAG
unsigned long long* g_fwd_dbg = nullptr;
#endif

}  // namespace

#ifdef NERF_DIAG
extern "C" void nerf_diag_fwd_stamps(unsigned long long* device_buffer) { g_fwd_dbg = device_buffer; }
#endif

extern "C" int nerf_mlp_fwd(int model, const float* packed, int64_t M, const float* points, const float* dirs,
                            const float* rays_o, const float* rays_d, const float* t, int S, int t_per_ray, double near,
                            double far, float* colors, float* density, float* workspace, void* stream) {
    NERF_REQUIRE(model == NERF_MODEL_NERF || model == NERF_MODEL_SIREN, "nerf_mlp_fwd: unknown model %d", model);
    NERF_REQUIRE(M >= 0, "nerf_mlp_fwd: M=%lld < 0", (long long)M);
    if (M == 0) return NERF_OK;
    NERF_REQUIRE_PTR(packed);
    NERF_REQUIRE_PTR(density);
    NERF_REQUIRE(colors != nullptr || workspace == nullptr,
                 "nerf_mlp_fwd: colors may be NULL (density only) without a training workspace only");
    if (points) {
        NERF_REQUIRE_PTR(dirs);
    } else {
        NERF_REQUIRE_PTR(rays_o);
        NERF_REQUIRE_PTR(rays_d);
        NERF_REQUIRE_PTR(t);
        NERF_REQUIRE(S >= 1 && M % S == 0, "nerf_mlp_fwd: ray mode needs M (%lld) to be a multiple of S (%d)",
                     (long long)M, S);
    }
    const size_t mp = lay::padded_samples(M);
    const size_t tiles = mp / TR;
    NERF_REQUIRE(tiles < (1ull << 31), "nerf_mlp_fwd: too many samples");
    FwdArgs a{packed, M, mp, points, dirs, rays_o, rays_d, t, S, t_per_ray, (float)near, (float)(far - near), colors, density, workspace};
#ifdef NERF_DIAG
    a.dbg = g_fwd_dbg;
#endif
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (model == NERF_MODEL_NERF) {
        if (workspace) {
            if ((rc = nerf_set_lds(nerf_fwd_kernel<true>, TAB_FWD_LDS_BYTES))) return rc;
            nerf_fwd_kernel<true><<<(unsigned)tiles, 256, TAB_FWD_LDS_BYTES, st>>>(a);
        } else if (colors) {
            if ((rc = nerf_set_lds(nerf_fwd_kernel<false>, TAB_FWD_LDS_BYTES))) return rc;
            nerf_fwd_kernel<false><<<(unsigned)tiles, 256, TAB_FWD_LDS_BYTES, st>>>(a);
        } else {
            if ((rc = nerf_set_lds(nerf_fwd_kernel<false, true>, TAB_FWD_LDS_BYTES))) return rc;
            nerf_fwd_kernel<false, true><<<(unsigned)tiles, 256, TAB_FWD_LDS_BYTES, st>>>(a);
        }
    } else {
        if (workspace) {
            if ((rc = nerf_set_lds(siren_fwd_kernel<true>, SIREN_FWD_LDS_BYTES))) return rc;
            siren_fwd_kernel<true><<<(unsigned)tiles, 256, SIREN_FWD_LDS_BYTES, st>>>(a);
        } else if (colors) {
            if ((rc = nerf_set_lds(siren_fwd_kernel<false>, SIREN_FWD_LDS_BYTES))) return rc;
            siren_fwd_kernel<false><<<(unsigned)tiles, 256, SIREN_FWD_LDS_BYTES, st>>>(a);
        } else {
            if ((rc = nerf_set_lds(siren_fwd_kernel<false, true>, SIREN_FWD_LDS_BYTES))) return rc;
            siren_fwd_kernel<false, true><<<(unsigned)tiles, 256, SIREN_FWD_LDS_BYTES, st>>>(a);
        }
    }
    return nerf_check_launch("nerf_mlp_fwd");
}

