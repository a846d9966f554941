// The questions of coexec.hip asked of v_mfma_f32_32x32x16_bf16 (8 passes, 32 cycles): do back-to-back MFMAs that
// accumulate into the SAME tile run at full rate, do VALU instructions of the same wave or of a co-resident wave overlap
// with them?  (Design input for a kernel that forms an fp32 product from bf16 pieces: 6 MFMAs per tile and k-step, three of
// them consecutive on one accumulator, and a splitting epilogue of ~9 VALU per two values.)
//   hipcc -O3 --offload-arch=gfx950 -o coexec_bf16 coexec_bf16.hip && ./coexec_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one wave per SIMD; RUN consecutive MFMAs on one accumulator before moving to the next of 8; NV FMAs after every MFMA
template <int RUN, int NV>
__global__ __launch_bounds__(256, 1) void chain(float* out, int iters, float x) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x * (threadIdx.x + i);
    const u32x4 w = {0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, w), b = a;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 48; ++u) {
            acc[(u / RUN) & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(u / RUN) & 7], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], x, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
        }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 8 waves per workgroup, one workgroup per CU: waves 0-3 run MFMAs when ROLE & 1, waves 4-7 FMA chains when ROLE & 2
template <int ROLE>
__global__ __launch_bounds__(512, 1) void roles(float* out, int mfma_iters, int valu_iters, float x) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (ROLE & 1) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
            const u32x4 w = {0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u};
            const bf16x8 a = __builtin_bit_cast(bf16x8, w), b = a;
            for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
        }
    } else if (ROLE & 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = x * (threadIdx.x + i);
        for (int it = 0; it < valu_iters; ++it)
#pragma unroll
            for (int u = 0; u < 64; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], x, 1.0f);
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// The three-term K loop in isolation: one wave per SIMD, every unit is one ds_read_b128 (issued PF units ahead) followed
// by 3, 2 or 1 MFMAs on one accumulator (unit u of a fragment meets 3 - u terms of the other operand).
// ORDER 0: read after the unit's MFMAs, scheduling barriers around them (the kernel's arrangement); 1: read in front of
// the unit's MFMAs; 2: as 0 without scheduling barriers; 3: read after the first MFMA of the unit
template <int PF, int ORDER = 0>
__global__ __launch_bounds__(256, 1) void ldsfed(float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    const u32x4 w = {0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, w), b1 = b0, b2 = b0;
    const unsigned char* buf = smem + 16 * lane;
    bf16x8 a[PF];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < PF; ++u) a[u] = *reinterpret_cast<const bf16x8*>(buf + u * 1024);
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            if (ORDER != 2) __builtin_amdgcn_sched_barrier(0);
            const int t = (u / 3) & 7;
            const bf16x8 au = a[u % PF];
            if (ORDER == 1) {
                if (u + PF < 24) a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u + PF) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(au, b0, acc[t], 0, 0, 0);
            if (ORDER == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (u + PF < 24) a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u + PF) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u % 3 < 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(au, b1, acc[t], 0, 0, 0);
            if (u % 3 < 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(au, b2, acc[t], 0, 0, 0);
            if (ORDER != 2) __builtin_amdgcn_sched_barrier(0);
            if (ORDER == 0 || ORDER == 2)
                if (u + PF < 24) a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u + PF) * 1024);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int g = 0; g < 16; ++g) s += acc[t][g];
    out[blockIdx.x * 256 + tid] = s;
}

// The same loop on 16-sample tiles (v_mfma_f32_16x16x32_bf16, 4 passes): 64 + 96 registers instead of 128 + 192, so two
// waves fit a SIMD -- but every 1 KiB unit then feeds 16 samples instead of 32: with 8 waves per CU the LDS must deliver
// 128 bytes per cycle, its peak.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PF>
__global__ __launch_bounds__(512, 1) void ldsfed16(float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) for (int g = 0; g < 4; ++g) acc[t][g] = 0.f;
    const u32x4 w = {0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, w), b1 = b0, b2 = b0;
    const unsigned char* buf = smem + 16 * lane;
    bf16x8 a[PF];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < PF; ++u) a[u] = *reinterpret_cast<const bf16x8*>(buf + u * 1024);
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            const int t = (u / 3) & 15;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % PF], b0, acc[t], 0, 0, 0);
            if (u % 3 < 2) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % PF], b1, acc[t], 0, 0, 0);
            if (u % 3 < 1) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u % PF], b2, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (u + PF < 24) a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u + PF) * 1024);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int g = 0; g < 4; ++g) s += acc[t][g];
    out[blockIdx.x * 512 + tid] = s;
}

template <typename F>
float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    const int it = 10000;   // 480,000 MFMAs per wave = 15.4 M cycles at 32 each
#define ROW(RUN, NV) printf("same wave: runs of %d on one accumulator, %2d FMA per MFMA: %.3f ms\n", RUN, NV, timed([&] { chain<RUN, NV><<<256, 256>>>(out, it, 1e-3f); }))
    ROW(1, 0); ROW(2, 0); ROW(3, 0); ROW(6, 0); ROW(48, 0);
    ROW(1, 1); ROW(1, 2); ROW(1, 4); ROW(1, 6); ROW(1, 8); ROW(3, 4);
    // 48 MFMAs per iteration, as the chain kernel above
    printf("K loop fed from LDS (3+2+1 MFMAs per unit): PF 2 %.3f ms, PF 4 %.3f ms, PF 8 %.3f ms\n",
           timed([&] { ldsfed<2><<<256, 256>>>(out, it); }), timed([&] { ldsfed<4><<<256, 256>>>(out, it); }),
           timed([&] { ldsfed<8><<<256, 256>>>(out, it); }));
    printf("K loop fed from LDS, PF 4: read in front %.3f ms, no scheduling barriers %.3f ms, read after the first MFMA %.3f ms\n",
           timed([&] { ldsfed<4, 1><<<256, 256>>>(out, it); }), timed([&] { ldsfed<4, 2><<<256, 256>>>(out, it); }),
           timed([&] { ldsfed<4, 3><<<256, 256>>>(out, it); }));
    // 8 waves x 480,000 MFMAs of 16 cycles: the same matrix-pipe time per SIMD as the 4-wave loops above (6.4 ms at 2.4 GHz)
    printf("16-sample tiles, two waves per SIMD, K loop fed from LDS: PF 4 %.3f ms, PF 8 %.3f ms\n",
           timed([&] { ldsfed16<4><<<256, 512>>>(out, it); }), timed([&] { ldsfed16<8><<<256, 512>>>(out, it); }));
    const int mi = 30000, vi = 10000;
    const float tm = timed([&] { roles<1><<<256, 512>>>(out, mi, vi, 1e-3f); });
    const float tv = timed([&] { roles<2><<<256, 512>>>(out, mi, vi, 1e-3f); });
    const float tb = timed([&] { roles<3><<<256, 512>>>(out, mi, vi, 1e-3f); });
    printf("MFMA waves alone %.3f ms, VALU waves alone %.3f ms, both on every SIMD %.3f ms (sum %.3f, max %.3f)\n", tm, tv, tb, tm + tv,
           tm > tv ? tm : tv);
    return 0;
}
