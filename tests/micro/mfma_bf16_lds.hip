// Microbenchmark: rate of v_mfma_f32_32x32x16_bf16 when every MFMA takes its A fragment from LDS
// (one ds_read_b128 per MFMA per wave, read PF units ahead), 8 accumulator tiles per wave,
// 8 waves per workgroup, one workgroup per CU -- the inner loop of the bf16 MLP kernel without its
// barriers, LDS-DMA and epilogues.   hipcc -O3 --offload-arch=gfx950 mfma_bf16_lds.hip -o mfma_bf16_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(std::make_integer_sequence<int, N>{}, f); }

template <int PF, bool USE_LDS>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    u32x4 bw = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u};
    bf16x8 b = __builtin_bit_cast(bf16x8, bw);
    const unsigned char* buf = smem + 16 * lane;
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[PF];
        auto fetch = [&](auto U) {
            constexpr int u = decltype(U)::value;
            if (USE_LDS) a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u % 32) * 1024);
            else a[u % PF] = b;
        };
        sfor<PF>(fetch);
        sfor<128>([&](auto U) {
            constexpr int u = decltype(U)::value;
            __builtin_amdgcn_sched_barrier(0);
            acc[u % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % PF], b, acc[u % 8], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u + PF < 128) fetch(std::integral_constant<int, u + PF>{});
        });
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int g = 0; g < 16; ++g) s += acc[t][g];
    out[blockIdx.x * 512 + tid] = s;
}

// MODE bit0: every wave issues 4 LDS-DMA loads of 1 KiB (scalar base + 32-bit lane offset) per 32 MFMAs, as the
// kernel's weight ring does; bit1: one workgroup barrier per 32 MFMAs (counted wait for the wave's own DMA first);
// bit2: the register epilogue of a layer (128 values: pack to bf16 pairs, ReLU as 16-bit max) after every 128 MFMAs.
template <int MODE>
__global__ __launch_bounds__(512, 1) void k2(float* out, const unsigned char* __restrict__ img, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int PF = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 32768 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    u32x4 bw = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u};
    bf16x8 b = __builtin_bit_cast(bf16x8, bw);
    bf16x8 X[16];
    for (int i = 0; i < 16; ++i) X[i] = b;
    int chunk = 0;
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[PF];
        const unsigned char* buf = smem + (chunk & 3) * 32768 + 16 * lane;
        auto fetch = [&](auto U) {
            constexpr int u = decltype(U)::value;
            if constexpr (u % 32 == 0) { buf = smem + (chunk & 3) * 32768 + 16 * lane; ++chunk; }
            if constexpr (u % 32 == PF) {
                if (MODE & 2) {
                    if (MODE & 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                if (MODE & 1) {
                    const unsigned char* src = img + (size_t)((chunk + 2) % 40) * 32768;
                    unsigned char* dst = smem + ((chunk + 2) & 3) * 32768 + 1024 * wave;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 8192 + 16 * tid),
                                                         (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, 0, 0);
                }
            }
            a[u % PF] = *reinterpret_cast<const bf16x8*>(buf + (u % 32) * 1024);
        };
        sfor<PF>(fetch);
        sfor<128>([&](auto U) {
            constexpr int u = decltype(U)::value;
            __builtin_amdgcn_sched_barrier(0);
            acc[u % 8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % PF], X[u / 8], acc[u % 8], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (u + PF < 128) fetch(std::integral_constant<int, u + PF>{});
        });
        if (MODE & 4) {
#pragma unroll
            for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 w;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        typedef short s16x2 __attribute__((ext_vector_type(2)));
                        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(
                            f32x2{acc[ft][8 * s2 + 2 * jp], acc[ft][8 * s2 + 2 * jp + 1]}, bf16x2));
                        w[jp] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), s16x2{0, 0}));
                    }
                    X[2 * ft + s2] = __builtin_bit_cast(bf16x8, w);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int g = 0; g < 16; ++g) s += acc[t][g];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run2(const char* name, float* out, const unsigned char* img) {
    const int iters = 64, blocks = 256 * 8;
    auto kern = k2<MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 512, 140 * 1024>>>(out, img, iters);
    hipEventRecord(e0);
    kern<<<blocks, 512, 140 * 1024>>>(out, img, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 8 * iters * 128 * 32768.0;
    printf("%s: %.3f ms  %.1f TFLOP/s  (%s)\n", name, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

template <typename K>
void run(const char* name, K kern, float* out) {
    const int iters = 64, blocks = 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 512, 140 * 1024>>>(out, iters);
    hipEventRecord(e0);
    kern<<<blocks, 512, 140 * 1024>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 8 * iters * 128 * 32768.0;
    printf("%s: %.3f ms  %.1f TFLOP/s  (%s)\n", name, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    run("reg operands          ", k<4, false>, out);
    run("lds operands, PF 4    ", k<4, true>, out);
    run("lds operands, PF 8    ", k<8, true>, out);
    run("lds operands, PF 2    ", k<2, true>, out);
    unsigned char* img; hipMalloc(&img, 40 * 32768 + 65536); hipMemset(img, 0x3c, 40 * 32768 + 65536);
    run2<0>("ring loop, plain                  ", out, img);
    run2<1>("ring loop + LDS-DMA               ", out, img);
    run2<2>("ring loop + barrier per 32 MFMAs  ", out, img);
    run2<3>("ring loop + DMA + barrier         ", out, img);
    run2<4>("ring loop + register epilogue     ", out, img);
    run2<7>("ring loop + DMA + barrier + epilogue", out, img);
    return 0;
}
