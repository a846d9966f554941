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
    return 0;
}
