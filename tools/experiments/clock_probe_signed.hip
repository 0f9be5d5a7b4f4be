// What clock does the chip hold under an all-integer v_mad_i64_i32 load (signed variant of clock_probe.hip)?  Every wave runs a long stream of independent
// 32x32+64 multiply-adds and reads the shader clock (s_memtime, clock64) and the constant 100 MHz timer (wall_clock64)
// around it: cycles / wall time = effective shader clock; mads / cycles = issue interval per SIMD.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/experiments/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe(uint64_t* out, uint32_t iters, uint32_t seed) {
    uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const uint32_t x = (uint32_t)a0 | 1u, y = x * 2654435761u;
    const uint64_t t0 = clock64(), w0 = wall_clock64();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {   // 64 independent-enough mads per iteration (8 chains)
            a0 = (uint64_t)((int64_t)(int32_t)a0 * (int32_t)x) + a1; a1 = (uint64_t)((int64_t)(int32_t)a1 * (int32_t)y) + a2; a2 = (uint64_t)((int64_t)(int32_t)a2 * (int32_t)x) + a3;
            a3 = (uint64_t)((int64_t)(int32_t)a3 * (int32_t)y) + a4; a4 = (uint64_t)((int64_t)(int32_t)a4 * (int32_t)x) + a5; a5 = (uint64_t)((int64_t)(int32_t)a5 * (int32_t)y) + a6;
            a6 = (uint64_t)((int64_t)(int32_t)a6 * (int32_t)x) + a7; a7 = (uint64_t)((int64_t)(int32_t)a7 * (int32_t)y) + a0;
        }
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[3 * wave + 0] = t1 - t0;
        out[3 * wave + 1] = w1 - w0;
        out[3 * wave + 2] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    }
}

int main() {
    const uint32_t iters = 20000;                  // 64 mads per iteration per lane
    for (int waves_per_simd : {1, 2, 3, 4, 8}) {
        const int n_waves = 256 * 4 * waves_per_simd;
        const int blocks = n_waves / 4;
        uint64_t* d;
        hipMalloc(&d, (size_t)n_waves * 3 * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        probe<<<blocks, 256>>>(d, 100, 1);       // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<blocks, 256>>>(d, iters, 2);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h((size_t)n_waves * 3);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int w = 0; w < n_waves; ++w) {
            cyc += (double)h[3 * w];
            wall += (double)h[3 * w + 1];
        }
        cyc /= n_waves;
        wall /= n_waves;
        const double mads_per_wave = 64.0 * iters;
        printf("%d waves/SIMD: kernel %.3f ms; per wave %.0f shader cycles in %.3f ms of the 100 MHz timer -> %.0f MHz effective; "
               "%.2f cycles per mad per wave, %.2f per SIMD; chip rate %.3e lane-mads/s\n",
               waves_per_simd, ms, cyc, wall / 100e3, cyc / (wall / 100.0), cyc / mads_per_wave, cyc / mads_per_wave / waves_per_simd,
               mads_per_wave * 64.0 * n_waves / (ms * 1e-3));
        hipFree(d);
    }
    return 0;
}
