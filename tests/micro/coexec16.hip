// v_mfma_f32_16x16x32_bf16 (16 cycles per SIMD) beside vector-ALU instructions, two waves per SIMD: do the vector
// instructions of a wave (placed between its own MFMAs) issue in the shadow of the MFMAs, its own or the partner's?
// KIND 0: v_fma_f32, 1: v_cvt_pk_bf16_f32, 2: v_pk_add_f32, 3: v_and_b32 / v_lshlrev_b32 pair, 4: s_nop 0
//   hipcc -O3 --offload-arch=gfx950 -o coexec16 coexec16.hip && ./coexec16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__device__ __forceinline__ void filler(float (&v)[8], int j, float x) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(x));
    if constexpr (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j & 7]) : "v"(x));
    if constexpr (KIND == 2) {
        f32x2 t = {v[(2 * j) & 7], v[(2 * j + 1) & 7]};
        asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(t));
        v[(2 * j) & 7] = t[0];
        v[(2 * j + 1) & 7] = t[1];
    }
    if constexpr (KIND == 3) {
        if (j & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[j & 7]));
        else asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(v[j & 7]));
    }
    if constexpr (KIND == 4) asm volatile("s_nop 0");
}

// WAVES waves per workgroup (4: one per SIMD, 8: two), one workgroup per CU.  Per iteration 48 MFMAs in runs of RUN on one
// accumulator of 16, NV fillers after every MFMA.
template <int WAVES, int RUN, int NV, int KIND, bool MFMA = true>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void chain(float* out, int iters, float x) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x * (threadIdx.x + i);
    const u32x4 a = {0x3f803f80u, 0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u}, b = a;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 48; ++u) {
            if constexpr (MFMA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(u / RUN) & 15]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) filler<KIND>(v, j, x);
        }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) s += acc[i][k];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
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
    const int it = 10000;   // 480,000 MFMAs per wave = 7.7 M cycles at 16 each (3.2 ms at 2.4 GHz); x 2 with two waves per SIMD
#define ROW(W, RUN, NV, KIND) printf("%d waves/SIMD, runs of %d, %d fillers of kind %d per MFMA: %.3f ms\n", W / 4, RUN, NV, KIND, timed([&] { chain<W, RUN, NV, KIND><<<256, 64 * W>>>(out, it / (W / 4), 1e-3f); }))
    // the same number of MFMAs per SIMD in every row (two waves: half the iterations each)
    ROW(4, 3, 0, 0); ROW(8, 3, 0, 0); ROW(8, 1, 0, 0); ROW(8, 6, 0, 0); ROW(4, 48, 0, 0); ROW(8, 48, 0, 0); ROW(8, 48, 1, 0); ROW(8, 3, 1, 0);
    ROW(4, 3, 1, 0); ROW(4, 3, 2, 0); ROW(4, 3, 4, 0);
    ROW(8, 3, 1, 0); ROW(8, 3, 2, 0); ROW(8, 3, 4, 0); ROW(8, 3, 6, 0);
    ROW(8, 3, 2, 1); ROW(8, 3, 4, 1);
    ROW(8, 3, 1, 2); ROW(8, 3, 2, 2);
    ROW(8, 3, 2, 3); ROW(8, 3, 4, 3);
    ROW(8, 3, 1, 4); ROW(8, 3, 2, 4);
    // the fillers alone (no MFMAs): cycles per instruction = ms * clock / (480,000 * NV) per SIMD
#define VROW(W, NV, KIND) printf("%d waves/SIMD, no MFMAs, %d fillers of kind %d per slot: %.3f ms\n", W / 4, NV, KIND, timed([&] { chain<W, 3, NV, KIND, false><<<256, 64 * W>>>(out, it / (W / 4), 1e-3f); }))
    VROW(4, 4, 0); VROW(8, 4, 0); VROW(8, 4, 1); VROW(4, 2, 2); VROW(8, 2, 2); VROW(8, 4, 3);
    return 0;
}
