// Probe: rate of random 128-byte row gathers (one row per lane, 8 x dwordx4 -- the access shape of msm_accumulate's
// table reads) as a function of the table size.  Question behind it (profiles/r03/r03_notes.md): would a window table with a
// row for EVERY bit position (256 rows instead of 16: 34 GB per 2^20 BLS12-381 points) still feed the kernel, or do
// TLB misses over tens of GB throttle the gathers?  msm_accumulate needs ~7e9 rows/s (0.9 TB/s).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/bin/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

__device__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// every lane: `iters` dependent-free gathers of one 128-byte row at a pseudo-random row index, next row prefetched
// one step ahead as in the real kernel (two loads in flight per lane)
__global__ void __launch_bounds__(128) gather(const uint4* table, uint64_t rows, uint32_t iters, uint32_t* out) {
    extern __shared__ uint32_t occupancy_limiter[];      // 40 KiB per 128 lanes = 2 waves per SIMD, msm_accumulate's occupancy
    if (iters == 0xffffffffu) occupancy_limiter[threadIdx.x] = 1;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    uint64_t h = mix(t * 0x9E3779B97F4A7C15ull + 1);
    uint4 nxt[8];
    {
        const uint4* p = table + (h % rows) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) nxt[k] = p[k];
    }
    for (uint32_t i = 0; i < iters; ++i) {
        uint4 cur[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
        h = mix(h + i);
        const uint4* p = table + (h % rows) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) nxt[k] = p[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc.x ^= cur[k].x;
            acc.y += cur[k].y;
            acc.z ^= cur[k].z;
            acc.w += cur[k].w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

__global__ void fill(uint4* table, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)mix(i);
        table[i] = make_uint4(v, v ^ 1, v ^ 2, v ^ 3);
    }
}

int main(int argc, char** argv) {
    const double sizes_gb[] = {0.25, 2.0, 8.0, 34.0, 68.0, 137.0};
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 4));
    for (double gb : sizes_gb) {
        const uint64_t rows = (uint64_t)(gb * (1ull << 30)) / 128;
        uint4* tab = nullptr;
        if (hipMalloc(&tab, rows * 128) != hipSuccess) {
            printf("%6.2f GiB: allocation failed\n", gb);
            (void)hipGetLastError();
            continue;
        }
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, tab, rows * 8);
        CHECK(hipDeviceSynchronize());
        const uint32_t lanes = 262144, iters = 64;     // msm_accumulate's grid at 2^20: 2^18 lanes x 64 references
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        for (size_t lds : {(size_t)40 << 10, (size_t)0}) {
            CHECK(hipFuncSetAttribute((const void*)gather, hipFuncAttributeMaxDynamicSharedMemorySize, 40 << 10));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(a));
                for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(gather, dim3(lanes / 128), dim3(128), lds, 0, tab, rows, iters, d_out);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
            }
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double n_rows = 4.0 * lanes * (iters + 1);
            printf("%6.2f GiB table, %s: %.3f ms per launch of 2^24 row gathers, %.2f G rows/s, %.2f TB/s\n", gb,
                   lds ? "2 waves/SIMD" : "full occupancy", ms / 4, n_rows / (ms * 1e-3) / 1e9, n_rows * 128 / (ms * 1e-3) / 1e12);
            fflush(stdout);
        }
        CHECK(hipFree(tab));
    }
    return 0;
}
