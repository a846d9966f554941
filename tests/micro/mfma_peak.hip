// Practical fp32-MFMA ceiling on this device: register-only loop vs a loop shaped like the MLP
// kernel's K block (2 LDS fragment reads + 2 global fragment loads per 16 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void reg_only(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256, 2) void shaped(float* out, const float* __restrict__ w, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 260; i += 256) lds[i] = (float)(i % 97) * 1e-3f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    const float* ap = lds + (lane & 31) * 260 + 4 * (lane >> 5);
    const float* bp = w + (size_t)(wave * 64 + (lane & 31)) * 8 + 4 * (lane >> 5);
    f32x4 a0[2], b0[2], a1[2], b1[2];
    auto ld = [&](int kb, f32x4 (&a)[2], f32x4 (&b)[2]) {
        for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(bp + (size_t)kb * 2048 + ni * 256);
        for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(ap + mi * 32 * 260 + kb * 8);
    };
    auto mm = [&](f32x4 (&a)[2], f32x4 (&b)[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
    };
    for (int it = 0; it < iters; ++it) {
        ld(0, a0, b0);
        for (int kb = 0; kb < 32; kb += 2) {
            ld(kb + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            ld(kb + 2 < 32 ? kb + 2 : kb, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// Variants of the K loop.  V bit0: A fragments from LDS; bit1: B fragments from global; bit2: the four
// loads of the next block are spread between the MFMAs of the current one (sched_group_barrier)
// instead of being issued in one burst at the block boundary.
template <int V>
__global__ __launch_bounds__(256, 2) void shaped_v(float* out, const float* __restrict__ w, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 260; i += 256) lds[i] = (float)(i % 97) * 1e-3f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    const float* ap = lds + (lane & 31) * 260 + 4 * (lane >> 5);
    const float* bp = w + (size_t)(wave * 64 + (lane & 31)) * 8 + 4 * (lane >> 5);
    f32x4 a0[2], b0[2], a1[2], b1[2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 4; ++k) { a0[i][k] = a1[i][k] = 1e-3f * lane; b0[i][k] = b1[i][k] = 2e-3f * lane; }
    // V bit4: B fragments through buffer loads (SGPR resource + one 32-bit offset VGPR, K block in the scalar offset)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 12 * 65536 * 4, 0x00020000);
    const int voff = ((wave * 64 + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4;
    auto ld = [&](int kb, f32x4 (&a)[2], f32x4 (&b)[2]) {
        if (V & 16) {
            for (int ni = 0; ni < 2; ++ni)
                b[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ni * 1024, kb * 8192, 0));
        } else if (V & 2) {
            for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(bp + (size_t)kb * 2048 + ni * 256);
        }
        if (V & 1) for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(ap + mi * 32 * 260 + kb * 8);
    };
    auto mm = [&](f32x4 (&a)[2], f32x4 (&b)[2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if (V & 8)   // accumulators pinned to the AGPR half of the register file
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[mi][ni]) : "v"(a[mi][j]), "v"(b[ni][j]));
                    else
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
                }
    };
    auto spread = [&]() {
        // MFMA x2, DS read, MFMA x2, DS read, MFMA x2, VMEM read, MFMA x2, VMEM read, MFMA x8
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (V & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (V & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (V & 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        if (V & 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    for (int it = 0; it < iters; ++it) {
        ld(0, a0, b0);
#pragma unroll 1
        for (int kb = 0; kb < 32; kb += 2) {
            if (V & 4) {
                ld(kb + 1, a1, b1);
                mm(a0, b0);
                spread();
                __builtin_amdgcn_sched_barrier(0);
                ld(kb + 2 < 32 ? kb + 2 : kb, a0, b0);
                mm(a1, b1);
                spread();
                __builtin_amdgcn_sched_barrier(0);
            } else {
                ld(kb + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mm(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                ld(kb + 2 < 32 ? kb + 2 : kb, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mm(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (V & 8) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results of the asm form are not tracked by the compiler
    float s = 0; for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
int run_shaped(const char* name, float* out, const float* w, hipEvent_t e0, hipEvent_t e1, int grid) {
    int iters = 400;
    CK(hipFuncSetAttribute((const void*)shaped_v<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 66560));
    shaped_v<V><<<grid, 256, 66560>>>(out, w, 20);
    CK(hipEventRecord(e0)); shaped_v<V><<<grid, 256, 66560>>>(out, w, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, (double)grid * 4 * iters * 32 * 16 * 4096.0 / ms / 1e9);
    return 0;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void layered(float* out, const float* __restrict__ w, int layers, int nimg) {
    // MODE bit0: barriers around an LDS-writing epilogue; bit1: different weight image per layer (nimg images);
    // bit2: bias load into the accumulators at layer start
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 260; i += 256) lds[i] = (float)(i % 97) * 1e-3f;
    __syncthreads();
    const float* ap = lds + (lane & 31) * 260 + 4 * (lane >> 5);
    f32x4 a0[2], b0[2], a1[2], b1[2];
    float s = 0;
    for (int l = 0; l < layers; ++l) {
        const float* img = w + ((MODE & 2) ? (size_t)((l + blockIdx.x) % nimg) * 65536 : 0);
        const float* bp = img + (size_t)(wave * 64 + (lane & 31)) * 8 + 4 * (lane >> 5);
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            float bias = (MODE & 4) ? img[wave * 64 + 32 * j + (lane & 31)] : 0.f;
            for (int k = 0; k < 16; ++k) acc[i][j][k] = bias;
        }
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, 65536 * 4, 0x00020000);
        const int voff = ((wave * 64 + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4;
        auto ld = [&](int kb, f32x4 (&a)[2], f32x4 (&b)[2]) {
            for (int ni = 0; ni < 2; ++ni) {
                if (MODE & 8) b[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ni * 1024, kb * 8192, 0));
                else b[ni] = *reinterpret_cast<const f32x4*>(bp + (size_t)kb * 2048 + ni * 256);
            }
            for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(ap + mi * 32 * 260 + kb * 8);
        };
        auto mm = [&](f32x4 (&a)[2], f32x4 (&b)[2]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        };
        ld(0, a0, b0);
#pragma unroll 1
        for (int kb = 0; kb < 32; kb += 2) {
            ld(kb + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            ld(kb + 2 < 32 ? kb + 2 : kb, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 1) {
            __syncthreads();
            if (MODE & 16) {        // the same 64 values as 16 ds_write_b128 (what a transposed accumulator tile would allow)
                for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g4 = 0; g4 < 4; ++g4) {
                    int row = 32 * mi + (lane & 31), col = wave * 64 + 32 * ni + 8 * g4 + 4 * (lane >> 5);
                    f32x4 v;
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[mi][ni][4 * g4 + k], 0.f) * 1e-3f;
                    *reinterpret_cast<f32x4*>(lds + row * 260 + col) = v;
                }
            } else if (MODE & 32) { // the VALU work without the LDS writes
                for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 16; ++g)
                    s += fmaxf(acc[mi][ni][g], 0.f) * 1e-3f;
            } else {
                for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 16; ++g) {
                    int row = 32 * mi + (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5), col = wave * 64 + 32 * ni + (lane & 31);
                    lds[row * 260 + col] = fmaxf(acc[mi][ni][g], 0.f) * 1e-3f;
                }
            }
            __syncthreads();
        } else {
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

// The layered kernel on 32-sample tiles: wave = 32 rows x 64 columns (2 accumulator tiles), 33 KB of LDS,
// four workgroups per CU (4 waves per SIMD) instead of two 64-sample ones.  B fragments through buffer loads.
template <int MODE>
__global__ __launch_bounds__(256, 4) void layered32(float* out, const float* __restrict__ w, int layers, int nimg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32 * 260; i += 256) lds[i] = (float)(i % 97) * 1e-3f;
    __syncthreads();
    const float* ap = lds + (lane & 31) * 260 + 4 * (lane >> 5);
    const int voff = ((wave * 64 + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4;
    f32x4 a0, b0[2], a1, b1[2];
    float s = 0;
    for (int l = 0; l < layers; ++l) {
        const float* img = w + ((MODE & 2) ? (size_t)((l + blockIdx.x) % nimg) * 65536 : 0);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, 65536 * 4, 0x00020000);
        f32x16 acc[2];
        for (int j = 0; j < 2; ++j) {
            float bias = (MODE & 4) ? img[wave * 64 + 32 * j + (lane & 31)] : 0.f;
            for (int k = 0; k < 16; ++k) acc[j][k] = bias;
        }
        auto ld = [&](int kb, f32x4& a, f32x4 (&b)[2]) {
            for (int ni = 0; ni < 2; ++ni)
                b[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ni * 1024, kb * 8192, 0));
            a = *reinterpret_cast<const f32x4*>(ap + kb * 8);
        };
        auto mm = [&](f32x4& a, f32x4 (&b)[2]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[ni][j], acc[ni], 0, 0, 0);
        };
        ld(0, a0, b0);
#pragma unroll 1
        for (int kb = 0; kb < 32; kb += 2) {
            ld(kb + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            ld(kb + 2 < 32 ? kb + 2 : kb, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 1) {
            __syncthreads();
            for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 16; ++g) {
                int row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5), col = wave * 64 + 32 * ni + (lane & 31);
                lds[row * 260 + col] = fmaxf(acc[ni][g], 0.f) * 1e-3f;
            }
            __syncthreads();
        } else {
            for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[j][k];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int MODE>
int run_layered32(const char* name, float* out, const float* w, hipEvent_t e0, hipEvent_t e1, int grid) {
    int layers = 400;
    CK(hipFuncSetAttribute((const void*)layered32<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 33280));
    layered32<MODE><<<grid, 256, 33280>>>(out, w, 20, 12);
    CK(hipEventRecord(e0)); layered32<MODE><<<grid, 256, 33280>>>(out, w, layers, 12); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, (double)grid * 4 * layers * 32 * 8 * 4096.0 / ms / 1e9);
    return 0;
}

template <int MODE>
int run_layered(const char* name, float* out, const float* w, hipEvent_t e0, hipEvent_t e1, int grid) {
    int layers = 400;
    CK(hipFuncSetAttribute((const void*)layered<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 66560));
    layered<MODE><<<grid, 256, 66560>>>(out, w, 20, 12);
    CK(hipEventRecord(e0)); layered<MODE><<<grid, 256, 66560>>>(out, w, layers, 12); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, (double)grid * 4 * layers * 32 * 16 * 4096.0 / ms / 1e9);
    return 0;
}

// Wave-tile shapes of the K loop with the kernel's per-block ingredients: MT row tiles x NT column tiles of 32x32
// (MT*NT == 4: 64 accumulator registers), MT LDS A-fragment reads + NT buffer-load B fragments per 16 MFMAs;
// COPY adds the training forward's workspace copy (one 16-byte LDS read and one non-temporal 16-byte buffer store
// per thread and K block).
template <int MT_, int NT_, int COPY>
__global__ __launch_bounds__(256, 2) void shaped_t(float* out, const float* __restrict__ w, float* __restrict__ ws, int iters) {
    static_assert(MT_ * NT_ == 4, "64 accumulator registers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 260; i += 256) lds[i] = (float)(i % 97) * 1e-3f;
    __syncthreads();
    f32x16 acc[MT_][NT_];
    for (int i = 0; i < MT_; ++i) for (int j = 0; j < NT_; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    // waves tile the 64 x 256 output: (2 / MT_) row groups x (4 * MT_ / 2 ... ) column groups
    constexpr int ROWG = 2 / MT_;                  // row groups of 32*MT_ rows
    const int rg = wave % ROWG, cg = wave / ROWG;  // column group of 32*NT_ columns
    const float* ap = lds + (rg * 32 * MT_ + (lane & 31)) * 260 + 4 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 12 * 65536 * 4, 0x00020000);
    const int voff = ((cg * 32 * NT_ + (lane & 31)) * 8 + 4 * (lane >> 5)) * 4;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blockIdx.x * 64 * 256, 0, 64 * 256 * 4, 0x00020000);
    f32x4 a0[MT_], b0[NT_], a1[MT_], b1[NT_];
    for (int i = 0; i < MT_; ++i) for (int k = 0; k < 4; ++k) a0[i][k] = a1[i][k] = 1e-3f * lane;
    for (int i = 0; i < NT_; ++i) for (int k = 0; k < 4; ++k) b0[i][k] = b1[i][k] = 2e-3f * lane;
    auto ld = [&](int kb, f32x4 (&a)[MT_], f32x4 (&b)[NT_]) {
#pragma unroll
        for (int ni = 0; ni < NT_; ++ni)
            b[ni] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ni * 1024, kb * 8192, 0));
#pragma unroll
        for (int mi = 0; mi < MT_; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(ap + mi * 32 * 260 + kb * 8);
    };
    auto mm = [&](f32x4 (&a)[MT_], f32x4 (&b)[NT_]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MT_; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT_; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
    };
    f32x4 cp = {0.f, 0.f, 0.f, 0.f};
    // COPY 1: store + read with the first burst; 2: with the second burst; 3: store in the first, read in the second
    auto copy_store = [&](int kb) {  // 512 pieces of 16 bytes per K-block pair: thread t stores the piece it read one block earlier
        const int piece = (kb >> 1) * 256 + threadIdx.x;          // 4096 pieces = 64 rows x 64 pieces
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, cp), wsrc, piece * 16, 0, 2);
    };
    auto copy_read = [&](int kb) {
        const int nxt = ((kb >> 1) + 1) % 16 * 256 + threadIdx.x;
        cp = *reinterpret_cast<const f32x4*>(lds + (nxt >> 6) * 260 + (nxt & 63) * 4);
    };
    for (int it = 0; it < iters; ++it) {
        ld(0, a0, b0);
#pragma unroll 1
        for (int kb = 0; kb < 32; kb += 2) {
            ld(kb + 1, a1, b1);
            if constexpr (COPY == 1 || COPY == 3) copy_store(kb);
            if constexpr (COPY == 1) copy_read(kb);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            ld(kb + 2 < 32 ? kb + 2 : kb, a0, b0);
            if constexpr (COPY == 2) copy_store(kb);
            if constexpr (COPY == 2 || COPY == 3) copy_read(kb);
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int i = 0; i < MT_; ++i) for (int j = 0; j < NT_; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MT_, int NT_, int COPY>
int run_shaped_t(const char* name, float* out, const float* w, float* ws, hipEvent_t e0, hipEvent_t e1, int grid) {
    int iters = 400;
    CK(hipFuncSetAttribute((const void*)shaped_t<MT_, NT_, COPY>, hipFuncAttributeMaxDynamicSharedMemorySize, 66560));
    shaped_t<MT_, NT_, COPY><<<grid, 256, 66560>>>(out, w, ws, 20);
    CK(hipEventRecord(e0)); shaped_t<MT_, NT_, COPY><<<grid, 256, 66560>>>(out, w, ws, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, (double)grid * 4 * iters * 32 * 16 * 4096.0 / ms / 1e9);
    return 0;
}

int main() {
    float *out, *w;
    CK(hipMalloc(&out, 2048 * 256 * 4));
    CK(hipMalloc(&w, 12 * 65536 * 4));
    CK(hipMemset(w, 0, 12 * 65536 * 4));
    std::vector<float> hw(65536);
    for (int i = 0; i < 65536; ++i) hw[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(w, hw.data(), 65536 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* name, double flop, float ms) { printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms / 1e9); };
    for (int rep = 0; rep < 2; ++rep) {
        {   // one wave per SIMD: 256 blocks of 256 threads
            int iters = 20000;
            CK(hipEventRecord(e0)); reg_only<1><<<256, 256>>>(out, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report("reg-only 1 wave/SIMD", 256.0 * 4 * iters * 16 * 4096.0, ms);
        }
        {   // two waves per SIMD
            int iters = 10000;
            CK(hipEventRecord(e0)); reg_only<2><<<512, 256>>>(out, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report("reg-only 2 waves/SIMD", 512.0 * 4 * iters * 16 * 4096.0, ms);
        }
        {
            int iters = 400;
            CK(hipFuncSetAttribute((const void*)shaped, hipFuncAttributeMaxDynamicSharedMemorySize, 66560));
            CK(hipEventRecord(e0)); shaped<<<512, 256, 66560>>>(out, w, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            report("MLP-shaped K loop 2 WG/CU", 512.0 * 4 * iters * 32 * 16 * 4096.0, ms);
        }
    }
    run_shaped<0>("K loop: no loads, 2 WG/CU", out, w, e0, e1, 512);
    run_shaped<1>("K loop: LDS A only", out, w, e0, e1, 512);
    run_shaped<2>("K loop: global B only", out, w, e0, e1, 512);
    run_shaped<3>("K loop: A + B, burst at block boundary", out, w, e0, e1, 512);
    run_shaped<7>("K loop: A + B, spread between MFMAs", out, w, e0, e1, 512);
    run_shaped<8>("K loop: no loads, AGPR accumulators", out, w, e0, e1, 512);
    run_shaped<9>("K loop: LDS A only, AGPR acc", out, w, e0, e1, 512);
    run_shaped<10>("K loop: global B only, AGPR acc", out, w, e0, e1, 512);
    run_shaped<11>("K loop: A + B burst, AGPR acc", out, w, e0, e1, 512);
    run_shaped<16>("K loop: buffer-load B only", out, w, e0, e1, 512);
    run_shaped<17>("K loop: A + buffer-load B, burst", out, w, e0, e1, 512);
    run_shaped<3>("K loop: A + B burst, 1 WG/CU", out, w, e0, e1, 256);
    run_shaped<7>("K loop: A + B spread, 1 WG/CU", out, w, e0, e1, 256);
    run_shaped<0>("K loop: no loads, 1 WG/CU", out, w, e0, e1, 256);
    {
        float* ws;
        CK(hipMalloc(&ws, (size_t)512 * 64 * 256 * 4));
        for (int rep = 0; rep < 2; ++rep) {
            run_shaped_t<2, 2, 0>("tile 64x64  (2 A + 2 B per block)", out, w, ws, e0, e1, 512);
            run_shaped_t<1, 4, 0>("tile 32x128 (1 A + 4 B per block)", out, w, ws, e0, e1, 512);
            run_shaped_t<2, 2, 1>("tile 64x64  + copy in the first burst", out, w, ws, e0, e1, 512);
            run_shaped_t<2, 2, 2>("tile 64x64  + copy in the second burst", out, w, ws, e0, e1, 512);
            run_shaped_t<2, 2, 3>("tile 64x64  + store first, read second", out, w, ws, e0, e1, 512);
            run_shaped_t<1, 4, 1>("tile 32x128 + copy in the first burst", out, w, ws, e0, e1, 512);
        }
        CK(hipFree(ws));
    }
    for (int rep = 0; rep < 2; ++rep) {
        run_layered<15>("layered: 64 ds_write_b32 epilogue", out, w, e0, e1, 512);
        run_layered<15 + 16>("layered: 16 ds_write_b128 epilogue", out, w, e0, e1, 512);
        run_layered<15 + 32>("layered: epilogue VALU only, no LDS writes", out, w, e0, e1, 512);
        run_layered<14>("layered: no epilogue, no barriers", out, w, e0, e1, 512);
    }
    run_layered<15>("layered 64-row: all, buffer loads, 2 WG/CU", out, w, e0, e1, 512);
    run_layered<8>("layered 64-row: plain, buffer loads", out, w, e0, e1, 512);
    run_layered32<7>("layered 32-row tiles: all, 4 WG/CU", out, w, e0, e1, 1024);
    run_layered32<7>("layered 32-row tiles: all, 2 WG/CU", out, w, e0, e1, 512);
    run_layered32<0>("layered 32-row tiles: plain, 4 WG/CU", out, w, e0, e1, 1024);
    run_layered<0>("layered: plain (no epilogue)", out, w, e0, e1, 512);
    run_layered<1>("layered: + barriers + LDS epilogue", out, w, e0, e1, 512);
    run_layered<3>("layered: + 12 weight images", out, w, e0, e1, 512);
    run_layered<7>("layered: + bias loads", out, w, e0, e1, 512);
    run_layered<7>("layered: all, grid 256", out, w, e0, e1, 256);
    run_layered<0>("layered: plain, grid 256", out, w, e0, e1, 256);
    return 0;
}
