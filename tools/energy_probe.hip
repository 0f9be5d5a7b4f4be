// Energy per vector instruction under the card's sustained power limit (profiles/r04/r04_notes.md: the prover's step is energy-bound).
// Runs ONE instruction stream -- nothing but independent chains of one instruction -- on every SIMD of the chip at a chosen occupancy
// for a few seconds and prints how many lane-instructions it issued in how long; tools/power_probe.py around it samples the socket
// power and the shader clock meanwhile:  python tools/power_probe.py out.txt -- tools/bin/energy_probe mad 2 3.0
//   joules per lane-instruction = mean watts x seconds / lane-instructions          (static power included: what the card really pays)
// Streams: mad  v_mad_i64_i32 (the limb product of fields.cuh)      add  v_add_u32      shift  v_lshrrev_b64      fma64  v_fma_f64
//          mix  the mixed addition's own ratio: 3055 multiply-adds to 1299 other vector instructions (adds / shifts / ands)
// Build: hipcc --offload-arch=gfx950 -O3 tools/energy_probe.hip -o tools/bin/energy_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHAINS 8
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(64) stream(uint64_t* out, uint32_t seed, int iters) {
    int32_t a = (int32_t)(seed * (threadIdx.x + 1)) | 1, b = (int32_t)(seed ^ (threadIdx.x * 2654435761u));
    int64_t acc[CHAINS];
    uint32_t r[CHAINS];
    double d[CHAINS];
    for (int i = 0; i < CHAINS; ++i) {
        acc[i] = a + i;
        r[i] = b + i;
        d[i] = 1.0 + i;
    }
    const double da = 1.0000001, db = 0.9999999;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == 0) {
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
                REP8(X)
#undef X
            } else {
                // 8 instructions in the mixed addition's ratio 3055 : 1299 ~ 5.6 : 2.4 -> over four rounds 23 multiply-adds, 9 others
#define M(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
#define A(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define S(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
                if (u == 0) { M(0) M(1) M(2) M(3) M(4) M(5) A(6) S(7) }
                else if (u == 1) { M(0) M(1) M(2) M(3) M(4) M(5) A(6) A(7) }
                else if (u == 2) { M(0) M(1) M(2) M(3) M(4) M(5) S(6) A(7) }
                else { M(0) M(1) M(2) M(3) M(4) A(5) S(6) A(7) }
#undef M
#undef A
#undef S
            }
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < CHAINS; ++i) s += (uint64_t)acc[i] + r[i] + (uint64_t)d[i];
    if (s == 0x1234567) out[0] = s;      // keeps the chains alive
}

int main(int argc, char** argv) {
    const char* op = argc > 1 ? argv[1] : "mad";
    const int wps = argc > 2 ? atoi(argv[2]) : 2;              // wavefronts per SIMD
    const double seconds = argc > 3 ? atof(argv[3]) : 3.0;
    const char* names[] = {"mad", "add", "shift", "fma64", "mix"};
    int which = -1;
    for (int i = 0; i < 5; ++i)
        if (!strcmp(op, names[i])) which = i;
    if (which < 0 || wps < 1 || wps > 8) {
        fprintf(stderr, "usage: energy_probe mad|add|shift|fma64|mix [waves per SIMD 1..8] [seconds]\n");
        return 2;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const unsigned blocks = (unsigned)(cus * 4 * wps);         // one 64-lane workgroup per (SIMD, wave slot)
    uint64_t* d_out = nullptr;
    if (hipMalloc(&d_out, 64) != hipSuccess) return 1;
    const int iters = 40000;                                   // x 32 instructions: ~10 ms per launch at 2 waves per SIMD
    void (*k)(uint64_t*, uint32_t, int) = which == 0 ? stream<0> : which == 1 ? stream<1> : which == 2 ? stream<2> : which == 3 ? stream<3> : stream<4>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, 1000);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, 12345u + (uint32_t)launches, iters);
        launches += 8;
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double lane_instr = (double)launches * blocks * 64.0 * iters * 32.0;
    printf("energy_probe %s: %d CUs, %d waves/SIMD, %ld launches, %.3f s, %.4e lane-instructions, %.4e per second\n", op, cus, wps, launches, el,
           lane_instr, lane_instr / el);
    (void)hipFree(d_out);
    return 0;
}
