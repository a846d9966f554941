// Do VALU instructions of one wave overlap with fp32 MFMAs of another wave on the same SIMD, and how many VALU
// instructions can a wave put between its own MFMAs for free?  (Diagnostic for the epilogue / encoding phases of the
// fused MLP kernels, which run beside the co-resident workgroup's MFMA loops.)
//   hipcc -O3 --offload-arch=gfx950 -o coexec coexec.hip && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) run MFMAs when ROLE & 1, waves 4-7 run
// independent FMA chains when ROLE & 2.
template <int ROLE>
__global__ __launch_bounds__(512, 1) void roles(float* out, int mfma_iters, int valu_iters, float x) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (ROLE & 1) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
            const float a = x * threadIdx.x, b = x + threadIdx.x;
            for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
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

// one wave per SIMD; NV independent FMAs after every MFMA of the same wave
template <int NV>
__global__ __launch_bounds__(256, 1) void interleaved(float* out, int iters, float x) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x * (threadIdx.x + i);
    const float a = x * threadIdx.x, b = x + threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], x, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
        }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one wave per SIMD; N instructions of one kind after every MFMA of the same wave
//   KIND 0: v_mov_b32 (VALU)   1: s_add_u32 (SALU)   2: ds_read_b32 (LDS)   3: ds_read_b128   4: ds_write_b32
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void between(float* out, int iters, float x) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 + 64];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) lds[i] = x * i;
    __syncthreads();
    const float a = x * threadIdx.x, b = x + threadIdx.x;
    const unsigned lds_addr = (unsigned)(size_t)(lds + threadIdx.x * 4);
    unsigned sreg = 1;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) asm volatile("v_mov_b32 v200, %0" :: "v"(a) : "v200");
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
                if (KIND == 2) asm volatile("ds_read_b32 v200, %0" :: "v"(lds_addr) : "v200");
                if (KIND == 3) asm volatile("ds_read_b128 v[200:203], %0" :: "v"(lds_addr) : "v200", "v201", "v202", "v203");
                if (KIND == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(lds_addr), "v"(a));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = (float)sreg;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// waves 4-7 issue 16 independent instructions of one kind per iteration (fixed registers v200..v231); waves 0-3 run
// MFMAs when WITH_MFMA.   KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_mov_b32   3: v_alignbit_b32   4: v_max_f32
#define R16(OP) OP(200) OP(202) OP(204) OP(206) OP(208) OP(210) OP(212) OP(214) OP(216) OP(218) OP(220) OP(222) OP(224) OP(226) OP(228) OP(230)
#define OP_FMA(r) "v_fma_f32 v" #r ", %0, %0, v" #r "\n\t"
#define OP_PKFMA(r) "v_pk_fma_f32 v[" #r ":" #r "+1], %1, %1, v[" #r ":" #r "+1]\n\t"
#define OP_MOV(r) "v_mov_b32 v" #r ", %0\n\t"
#define OP_ALIGN(r) "v_alignbit_b32 v" #r ", v" #r ", %0, 31\n\t"
#define OP_MAX(r) "v_max_f32 v" #r ", 0, %0\n\t"
#define CLOB "v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231"
template <int KIND, bool WITH_MFMA>
__global__ __launch_bounds__(512, 1) void beside(float* out, int mfma_iters, int valu_iters, float x) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (WITH_MFMA) {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
            const float a = x * threadIdx.x, b = x + threadIdx.x;
            for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
        }
    } else {
        const float a = x * threadIdx.x;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 a2 = {a, a};
        for (int it = 0; it < valu_iters; ++it) {
            if (KIND == 0) asm volatile(R16(OP_FMA) :: "v"(a), "v"(a2) : CLOB);
            if (KIND == 1) asm volatile(R16(OP_PKFMA) :: "v"(a), "v"(a2) : CLOB);
            if (KIND == 2) asm volatile(R16(OP_MOV) :: "v"(a), "v"(a2) : CLOB);
            if (KIND == 3) asm volatile(R16(OP_ALIGN) :: "v"(a), "v"(a2) : CLOB);
            if (KIND == 4) asm volatile(R16(OP_MAX) :: "v"(a), "v"(a2) : CLOB);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// one wave per SIMD: RUN consecutive MFMAs on one accumulator before moving to the next of 4 (does the order of the 16
// MFMAs of a K block matter?)
template <int RUN>
__global__ __launch_bounds__(256, 1) void runs(float* out, int iters, float x) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    const float a = x * threadIdx.x, b = x + threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[(u / RUN) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(u / RUN) & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
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
    printf("fp32 MFMA, runs on one accumulator: 1 %.3f ms, 2 %.3f ms, 4 %.3f ms, 16 %.3f ms (320,000 MFMAs per wave)\n",
           timed([&] { runs<1><<<256, 256>>>(out, 20000, 1e-3f); }), timed([&] { runs<2><<<256, 256>>>(out, 20000, 1e-3f); }),
           timed([&] { runs<4><<<256, 256>>>(out, 20000, 1e-3f); }), timed([&] { runs<16><<<256, 256>>>(out, 20000, 1e-3f); }));
    const int mi = 20000, vi = 20000;   // 320,000 MFMAs (64 cycles each) / 1,280,000 FMAs (4 cycles each) per wave: 20.5 M cycles each
    const float tm = timed([&] { roles<1><<<256, 512>>>(out, mi, vi, 1e-3f); });
    const float tv = timed([&] { roles<2><<<256, 512>>>(out, mi, vi, 1e-3f); });
    const float tb = timed([&] { roles<3><<<256, 512>>>(out, mi, vi, 1e-3f); });
    printf("MFMA waves alone %.3f ms, VALU waves alone %.3f ms, both on every SIMD %.3f ms (sum %.3f, max %.3f)\n", tm, tv, tb, tm + tv,
           tm > tv ? tm : tv);
    const float tb2 = timed([&] { roles<3><<<256, 512>>>(out, mi, vi / 4, 1e-3f); });
    printf("both, a quarter of the VALU work: %.3f ms\n", tb2);
    const int it = 20000;
    const float t0 = timed([&] { interleaved<0><<<256, 256>>>(out, it, 1e-3f); });
    printf("same wave: MFMA only %.3f ms\n", t0);
    printf("same wave: + 2 FMA per MFMA %.3f ms\n", timed([&] { interleaved<2><<<256, 256>>>(out, it, 1e-3f); }));
    printf("same wave: + 4 FMA per MFMA %.3f ms\n", timed([&] { interleaved<4><<<256, 256>>>(out, it, 1e-3f); }));
    printf("same wave: + 8 FMA per MFMA %.3f ms\n", timed([&] { interleaved<8><<<256, 256>>>(out, it, 1e-3f); }));
    printf("same wave: + 12 FMA per MFMA %.3f ms\n", timed([&] { interleaved<12><<<256, 256>>>(out, it, 1e-3f); }));
    printf("same wave: + 16 FMA per MFMA %.3f ms\n", timed([&] { interleaved<16><<<256, 256>>>(out, it, 1e-3f); }));
    {
        const char* names[5] = {"v_fma_f32", "v_pk_fma_f32", "v_mov_b32", "v_alignbit_b32", "v_max_f32"};
        const int vit = 40000;     // 640,000 instructions per wave
        const float m0 = timed([&] { beside<0, true><<<256, 512>>>(out, mi, 0, 1e-3f); });
#define BROW(K)                                                                                                        \
        {                                                                                                              \
            const float alone = timed([&] { beside<K, false><<<256, 512>>>(out, mi, vit, 1e-3f); });                   \
            const float both = timed([&] { beside<K, true><<<256, 512>>>(out, mi, vit, 1e-3f); });                     \
            printf("%-15s alone %.3f ms = %.1f cycles each; beside MFMA waves (%.3f ms alone): %.3f ms = +%.1f cycles each\n", names[K], alone, \
                   alone * 2.4e6 / (16.0 * vit), m0, both, (both - m0) * 2.4e6 / (16.0 * vit));                         \
        }
        BROW(0) BROW(1) BROW(2) BROW(3) BROW(4)
    }
    const char* kinds[5] = {"v_mov_b32", "s_add_u32", "ds_read_b32", "ds_read_b128", "ds_write_b32"};
#define ROW(K)                                                                                              \
    printf("same wave, %-12s per MFMA: x1 %.3f  x2 %.3f  x4 %.3f ms (MFMA only %.3f)\n", kinds[K],           \
           timed([&] { between<K, 1><<<256, 256>>>(out, it, 1e-3f); }),                                      \
           timed([&] { between<K, 2><<<256, 256>>>(out, it, 1e-3f); }),                                      \
           timed([&] { between<K, 4><<<256, 256>>>(out, it, 1e-3f); }), t0)
    ROW(0); ROW(1); ROW(2); ROW(3); ROW(4);
    return 0;
}
