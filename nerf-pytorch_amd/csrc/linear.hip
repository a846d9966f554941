// Layer-wise dense path for architectures that have no fused kernel (any hidden width, depth or
// encoding size the reference's constructors accept: nerf/models.py:18-60, :148-183).
//   forward   y = post * f(pre * (x W^T + b)),   f in {identity, relu, sin, sigmoid}
//   backward  dz = dy * d y/d z;  dx = dz W;  dW = dz^T x;  db = column sums of dz
// One tiled fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32) serves the three products through two
// transposition flags; operands are staged k-major through LDS with bounds checks, so any M, N, K
// and leading dimension works.  The weight gradient reduces over the samples: it is split over
// grid.z into partial slabs that a second kernel adds in a fixed order (no atomics, reproducible).
// This path trades speed for generality; the default architectures never take it.
#include "common.h"
#include "sincos.h"

#include <mutex>

// Scratch of nerf_linear_bwd_noalloc: the workspace of nerf_linear_bwd (activation-gradient matrix,
// weight-gradient partial slabs, bias partials) as a file-scope device array, so that the entry point
// needs no allocation and no workspace argument.  Every GPU the library runs on holds its own copy at
// its own address until the process ends.  In host code the NAME `scratch` is only a handle for
// hipGetSymbolAddress; kernels and memory routines take the per-device address kept in g_scratch_ptr.
static __device__ __align__(256) float scratch[NERF_NOALLOC_SCRATCH_FLOATS];

namespace {

enum { LIN_NONE = 0, LIN_RELU = 1, LIN_SIN = 2, LIN_SIGMOID = 3 };

constexpr int BM = 64, BN = 64, BK = 16, LDT = 68;   // LDT: padded tile row (floats)

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int64_t M;          // rows of C
    int N;              // columns of C
    int64_t Kred;       // reduction length
    int64_t kchunk;     // reduction elements per grid.z slice (multiple of BK)
    int64_t c_slice;    // floats between the C of consecutive slices (0: single slice)
    const float* bias;  // [N] or NULL
    int act; float pre, post;
    float* aux;         // layout of C: sin stores pre*cos(pre z); NULL otherwise
};

__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }

// C[m][n] = sum_k opA(m,k) * opB(k,n):  opA(m,k) = TA ? A[k*lda+m] : A[m*lda+k],
//                                        opB(k,n) = TB ? B[n*ldb+k] : B[k*ldb+n]
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    __shared__ float As[BK * LDT], Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int64_t k_begin = (int64_t)blockIdx.z * a.kchunk;
    const int64_t k_end = k_begin + a.kchunk < a.Kred ? k_begin + a.kchunk : a.Kred;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    float ra[4], rb[4];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            {
                const int mm = TA ? (idx & 63) : (idx >> 4), kk = TA ? (idx >> 6) : (idx & 15);
                const int64_t m = m0 + mm, k = k0 + kk;
                ra[i] = (m < a.M && k < k_end) ? (TA ? a.A[k * a.lda + m] : a.A[m * a.lda + k]) : 0.0f;
            }
            {
                const int nn = TB ? (idx >> 4) : (idx & 63), kk = TB ? (idx & 15) : (idx >> 6);
                const int n = n0 + nn;
                const int64_t k = k0 + kk;
                rb[i] = (n < a.N && k < k_end) ? (TB ? a.B[(int64_t)n * a.ldb + k] : a.B[k * a.ldb + n]) : 0.0f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int mm = TA ? (idx & 63) : (idx >> 4), ka = TA ? (idx >> 6) : (idx & 15);
            As[ka * LDT + mm] = ra[i];
            const int nn = TB ? (idx >> 4) : (idx & 63), kb = TB ? (idx & 15) : (idx >> 6);
            Bs[kb * LDT + nn] = rb[i];
        }
    };

    if (k_begin < k_end) gload(k_begin);
    for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();   // the previous step's fragment reads are done
        lstore();
        __syncthreads();
        if (k0 + BK < k_end) gload(k0 + BK);   // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float av = As[(kk + h) * LDT + wm * 32 + r];
            const float bv = Bs[(kk + h) * LDT + wn * 32 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }

    float* C = a.C + (int64_t)blockIdx.z * a.c_slice;
    const int col = n0 + wn * 32 + r;
    if (col >= a.N) return;
    const float bias = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int64_t row = m0 + wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (row >= a.M) continue;
        const float z = acc[g] + bias;
        float y = z;
        if (a.act == LIN_RELU) y = a.post * fmaxf(z, 0.0f);
        else if (a.act == LIN_SIN) {
            float sn, cs;
            nerf_sincosf(a.pre * z, &sn, &cs);
            y = a.post * sn;
            if (a.aux) a.aux[row * a.ldc + col] = a.post * a.pre * cs;
        } else if (a.act == LIN_SIGMOID) y = a.post * sigmoid_(a.pre * z);
        C[row * a.ldc + col] = y;
    }
}

// dz = dy * (d y / d z) from what the forward kept: relu -> y, sin -> aux, sigmoid -> y
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int64_t ldy,
                                                     const float* __restrict__ y, const float* __restrict__ aux,
                                                     int64_t ldo, int64_t M, int N, int act, float pre, float post,
                                                     float* __restrict__ dz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t m = i / N;
    const int n = (int)(i - m * N);
    const float g = dy[m * ldy + n];
    float d = g;
    if (act == LIN_RELU) d = y[m * ldo + n] > 0.0f ? g * post : 0.0f;
    else if (act == LIN_SIN) d = g * aux[m * ldo + n];
    else if (act == LIN_SIGMOID) {
        const float s = y[m * ldo + n] / post;
        d = g * post * pre * s * (1.0f - s);
    }
    dz[i] = d;
}

// part[z][n] = sum over the slice's rows of dz[m][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dz, int64_t M, int N, int64_t rows_per,
                                                    float* __restrict__ part) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    float s0 = 0.f, s1 = 0.f;
    int64_t m = r0;
    for (; m + 2 <= r1; m += 2) {
        s0 += dz[m * N + n];
        s1 += dz[(m + 1) * N + n];
    }
    if (m < r1) s0 += dz[m * N + n];
    part[(int64_t)blockIdx.y * N + n] = s0 + s1;
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, int Z, int64_t count,
                                                      float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int z = 0; z < Z; ++z) s += slab[(int64_t)z * count + i];
    out[i] = s;
}

// Sample slices of the weight gradient: about 2048 samples each up to 64 slices, then longer slices; beyond 4096
// samples per slice the slice count grows again (up to 256).  A slice is summed sequentially in one fp32 accumulator,
// so its length sets the rounding error of the gradient: at 16384 samples per slice (the old cap of 64 slices at
// 1,048,576 samples) a 63-input layer's gradient was 1e-4 off the float64 value, at 4096 it is about 1e-6.
int splits_for(int64_t M) {
    int64_t z = M / 2048;
    if (z > 64) {
        z = M / 4096;
        z = z < 64 ? 64 : z > 256 ? 256 : z;
    }
    return (int)(z < 1 ? 1 : z);
}

}  // namespace

extern "C" size_t nerf_linear_bwd_workspace_floats(int64_t M, int K, int N) {
    if (M < 0 || K < 1 || N < 1) return 0;
    const int Z = splits_for(M);
    return (size_t)M * N + (size_t)Z * ((size_t)N * K + N);
}

extern "C" int nerf_linear_fwd(const float* x, int64_t ldx, const float* W, const float* b, int64_t M, int K, int N,
                               int act, double pre, double post, float* y, int64_t ldy, float* aux, void* stream) {
    NERF_REQUIRE(M >= 0 && K >= 1 && N >= 1 && ldx >= K && ldy >= N, "nerf_linear_fwd: bad shape M=%lld K=%d N=%d",
                 (long long)M, K, N);
    NERF_REQUIRE(act >= LIN_NONE && act <= LIN_SIGMOID, "nerf_linear_fwd: unknown activation %d", act);
    NERF_REQUIRE(ceil_div64(M, BM) <= 65535 * 1024ll, "nerf_linear_fwd: too many rows");
    if (M == 0) return NERF_OK;
    NERF_REQUIRE_PTR(x);
    NERF_REQUIRE_PTR(W);
    NERF_REQUIRE_PTR(y);
    if (act == LIN_SIN) NERF_REQUIRE_PTR(aux);
    // grid.y is limited to 65535 blocks: walk the rows in slabs of 65535 * 64
    const int64_t rows_per_launch = 65535ll * BM;
    for (int64_t r0 = 0; r0 < M; r0 += rows_per_launch) {
        const int64_t rows = M - r0 < rows_per_launch ? M - r0 : rows_per_launch;
        GemmArgs a{x + r0 * ldx, ldx, W, K, y + r0 * ldy, ldy, rows, N, K, (int64_t)((K + BK - 1) / BK) * BK, 0, b, act,
                   (float)pre, (float)post, aux ? aux + r0 * ldy : nullptr};
        dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)ceil_div64(rows, BM), 1);
        gemm_kernel<false, true><<<grid, 256, 0, (hipStream_t)stream>>>(a);
        int rc = nerf_check_launch("nerf_linear_fwd");
        if (rc) return rc;
    }
    return NERF_OK;
}

static int linear_bwd_impl(const char* fn, const float* dy, int64_t lddy, const float* y, const float* aux, int64_t ldy, int act,
                               double pre, double post, const float* x, int64_t ldx, const float* W, int64_t M, int K,
                               int N, float* workspace, float* dx, int64_t lddx, float* dW, float* db, void* stream) {
    NERF_REQUIRE(M >= 1 && K >= 1 && N >= 1, "%s: bad shape M=%lld K=%d N=%d", fn, (long long)M, K, N);
    NERF_REQUIRE(act >= LIN_NONE && act <= LIN_SIGMOID, "%s: unknown activation %d", fn, act);
    NERF_REQUIRE(ceil_div64(M, BM) <= 65535, "%s: more than 65535*64 rows per call", fn);
    NERF_REQUIRE_PTR(dy);
    NERF_REQUIRE_PTR(x);
    NERF_REQUIRE_PTR(W);
    NERF_REQUIRE_PTR(workspace);
    NERF_REQUIRE_PTR(dW);
    if (act == LIN_RELU || act == LIN_SIGMOID) NERF_REQUIRE_PTR(y);
    if (act == LIN_SIN) NERF_REQUIRE_PTR(aux);
    hipStream_t st = (hipStream_t)stream;
    const int Z = splits_for(M);
    float* dz = workspace;
    float* slab = workspace + (size_t)M * N;
    float* bpart = slab + (size_t)Z * N * K;
    int rc;
    act_bwd_kernel<<<(unsigned)ceil_div64(M * N, 256), 256, 0, st>>>(dy, lddy, y, aux, ldy, M, N, act, (float)pre,
                                                                       (float)post, dz);
    if ((rc = nerf_check_launch("nerf_linear_bwd (activation)"))) return rc;
    if (dx) {  // dx[M,K] = dz[M,N] W[N,K]
        GemmArgs a{dz, N, W, K, dx, lddx, M, K, N, (int64_t)((N + BK - 1) / BK) * BK, 0, nullptr, LIN_NONE, 1.f, 1.f, nullptr};
        dim3 grid((unsigned)((K + BN - 1) / BN), (unsigned)ceil_div64(M, BM), 1);
        gemm_kernel<false, false><<<grid, 256, 0, st>>>(a);
        if ((rc = nerf_check_launch("nerf_linear_bwd (data)"))) return rc;
    }
    {  // slab[z][N,K] = dz[slice]^T x[slice]
        int64_t per = ceil_div64(ceil_div64(M, Z), BK) * BK;
        GemmArgs a{dz, N, x, ldx, slab, K, N, K, M, per, (int64_t)N * K, nullptr, LIN_NONE, 1.f, 1.f, nullptr};
        dim3 grid((unsigned)((K + BN - 1) / BN), (unsigned)((N + BM - 1) / BM), (unsigned)Z);
        gemm_kernel<true, false><<<grid, 256, 0, st>>>(a);
        if ((rc = nerf_check_launch("nerf_linear_bwd (weights)"))) return rc;
        slab_sum_kernel<<<(unsigned)ceil_div64((int64_t)N * K, 256), 256, 0, st>>>(slab, Z, (int64_t)N * K, dW);
        if ((rc = nerf_check_launch("nerf_linear_bwd (weight reduce)"))) return rc;
    }
    if (db) {
        const int64_t rows_per = ceil_div64(M, Z);
        colsum_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)Z), 256, 0, st>>>(dz, M, N, rows_per, bpart);
        if ((rc = nerf_check_launch("nerf_linear_bwd (bias)"))) return rc;
        slab_sum_kernel<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(bpart, Z, N, db);
        if ((rc = nerf_check_launch("nerf_linear_bwd (bias reduce)"))) return rc;
    }
    return NERF_OK;
}

extern "C" int nerf_linear_bwd(const float* dy, int64_t lddy, const float* y, const float* aux, int64_t ldy, int act,
                               double pre, double post, const float* x, int64_t ldx, const float* W, int64_t M, int K,
                               int N, float* workspace, float* dx, int64_t lddx, float* dW, float* db, void* stream) {
    return linear_bwd_impl("nerf_linear_bwd", dy, lddy, y, aux, ldy, act, pre, post, x, ldx, W, M, K, N, workspace, dx, lddx,
                           dW, db, stream);
}

// ---- the allocation-free variant ----------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 64;
float* g_scratch_ptr[kMaxDevices] = {};   // device address of `scratch` on each GPU, resolved on first use
std::mutex g_scratch_mutex;

// Address of this device's copy of `scratch` (resolved once per device, with that device current).
int scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return nerf_fail(NERF_E_LAUNCH, "nerf_linear_bwd_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return nerf_fail(NERF_E_ARG, "nerf_linear_bwd_noalloc: device %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr)
            return nerf_fail(NERF_E_LAUNCH, "nerf_linear_bwd_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev,
                             hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return NERF_OK;
}
}  // namespace

extern "C" int nerf_linear_bwd_noalloc(const float* dy, int64_t lddy, const float* y, const float* aux, int64_t ldy,
                                       int act, double pre, double post, const float* x, int64_t ldx, const float* W,
                                       int64_t M, int K, int N, float* dx, int64_t lddx, float* dW, float* db,
                                       void* stream) {
    NERF_REQUIRE(M >= 1 && K >= 1 && N >= 1, "nerf_linear_bwd_noalloc: bad shape M=%lld K=%d N=%d", (long long)M, K, N);
    const size_t need = nerf_linear_bwd_workspace_floats(M, K, N);
    NERF_REQUIRE(need <= (size_t)NERF_NOALLOC_SCRATCH_FLOATS,
                 "nerf_linear_bwd_noalloc: M=%lld K=%d N=%d needs %zu floats of scratch, the built-in buffer holds %zu "
                 "(use nerf_linear_bwd with a workspace for larger layers)",
                 (long long)M, K, N, need, (size_t)NERF_NOALLOC_SCRATCH_FLOATS);
    float* ws = nullptr;
    int rc = scratch_for_current_device(&ws);
    if (rc) return rc;
    // every scratch element the kernels read is written earlier in the same call (activation gradients,
    // all Z partial slabs, all Z bias partials), so nothing has to be cleared between calls
    return linear_bwd_impl("nerf_linear_bwd_noalloc", dy, lddy, y, aux, ldy, act, pre, post, x, ldx, W, M, K, N, ws, dx,
                           lddx, dW, db, stream);
}
