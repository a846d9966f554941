// Diagnostic: one wave that samples the shader clock counter against the 100 MHz real-time counter
// while other kernels run, to read the core clock the device sustains under their load.
// Build:  hipcc -O2 --offload-arch=gfx950 -shared -fPIC tests/diag/clock_probe.hip -o tests/diag/libclockprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

// out: [gridDim.x][2*n + 2] -- per workgroup: XCC id, CU/SE id word, then n (cycles, ticks) pairs
__global__ void clock_probe_kernel(uint64_t* out, int n, int spin_ticks) {
    if (threadIdx.x != 0) return;
    out += (size_t)blockIdx.x * (2 * (size_t)n + 2);
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[0] = xcc & 0xf;
    out[1] = hwid;
    out += 2;
    for (int i = 0; i < n; ++i) {
        const uint64_t c = clock64();        // shader clock cycles
        const uint64_t w = wall_clock64();   // 100 MHz ticks
        out[2 * i] = c;
        out[2 * i + 1] = w;
        const uint64_t until = w + (uint64_t)spin_ticks;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);   // bounded: the counter always advances
    }
}

extern "C" int clock_probe_launch(uint64_t* out, int probes, int n, int spin_ticks, void* stream) {
    if (!out || probes < 1 || probes > 64 || n <= 0 || n > (1 << 20) || spin_ticks < 0 || spin_ticks > 100000) return -1;
    clock_probe_kernel<<<probes, 64, 0, (hipStream_t)stream>>>(out, n, spin_ticks);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
