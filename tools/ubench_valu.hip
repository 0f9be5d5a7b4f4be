// Micro-benchmark: issue cost of the integer VALU instructions the Montgomery products are made of.
// SURVEY.md section 7 "hard part (i)": v_mad_u64_u32 / v_mul_*_u32 rates are not in the MI355X guide.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
// Output: cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define ITERS 2000
#define UNROLL 16

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void bench(uint64_t* out, uint32_t seed) {
    uint32_t a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (threadIdx.x * 2654435761u);
    uint64_t acc[UNROLL];
    uint32_t r[UNROLL];
    double d[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { acc[i] = a + i; r[i] = b + i; d[i] = 1.0 + i; }
    double da = 1.0000001, db = 0.9999999;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            REP16(X)
#undef X
        } else if (OP == 1) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (OP == 2) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (OP == 3) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (OP == 4) {
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (OP == 5) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc");
            REP16(X)
#undef X
        } else if (OP == 6) {
#define X(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
            REP16(X)
#undef X
        } else if (OP == 7) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (OP == 8) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL]));
            REP16(X)
#undef X
        } else if (OP == 9) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
            REP16(X)
#undef X
        } else if (OP == 10) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (OP == 11) {
#define X(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[i]));
            REP16(X)
#undef X
        } else if (OP == 12) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
            REP16(X)
#undef X
        } else if (OP == 13) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (OP == 14) {
            // the CIOS inner step as the compiler-independent pattern: mad + addc
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
            REP16(X)
#undef X
        } else if (OP == 15) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        } else if (OP == 16) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (OP == 17) {
#define X(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s10", "s11");
            REP16(X)
#undef X
        } else if (OP == 18) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
            REP16(X)
#undef X
        } else if (OP == 19) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t s = 0;
    for (int i = 0; i < UNROLL; ++i) s += acc[i] + r[i] + (uint64_t)d[i];
    if (s == 0x1234567) out[0] = s;  // keep live
    if (threadIdx.x % 64 == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name, uint64_t* d_out, std::vector<uint64_t>& h) {
    printf("%-28s", name);
    for (int wps : {1, 2, 4}) {
        int threads = 64 * 4 * wps;  // one workgroup per CU, wps waves per SIMD
        int blocks = 256;
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u);
        hipDeviceSynchronize();
        int nw = blocks * threads / 64;
        hipMemcpy(h.data(), d_out, (nw + 1) * 8, hipMemcpyDeviceToHost);
        std::vector<uint64_t> t(h.begin() + 1, h.begin() + 1 + nw);
        std::sort(t.begin(), t.end());
        double med = (double)t[nw / 2];
        // cycles per wave-instruction on one SIMD = elapsed / (instr per wave * waves per SIMD)
        double per = med / ((double)ITERS * UNROLL * wps);
        printf("  wps=%d: %6.2f cyc/instr/SIMD (wave %.0f)", wps, per, med / ((double)ITERS * UNROLL));
    }
    printf("\n");
}

int main() {
    uint64_t* d_out;
    hipMalloc(&d_out, 8 * (1 + 256 * 16 + 16));
    std::vector<uint64_t> h(1 + 256 * 16 + 16);
    run<19>("v_fma_f32", d_out, h);
    run<13>("v_add_u32", d_out, h);
    run<10>("v_mov_b32", d_out, h);
    run<7>("v_add3_u32", d_out, h);
    run<5>("v_add_co_u32", d_out, h);
    run<6>("v_addc_co_u32", d_out, h);
    run<12>("v_cndmask_b32", d_out, h);
    run<8>("v_lshl_add_u64", d_out, h);
    run<11>("v_lshrrev_b64", d_out, h);
    run<0>("v_mad_u64_u32 (vcc)", d_out, h);
    run<17>("v_mad_u64_u32 (sgpr)", d_out, h);
    run<14>("v_mad_u64_u32+v_addc (pair)", d_out, h);
    run<1>("v_mul_lo_u32", d_out, h);
    run<2>("v_mul_hi_u32", d_out, h);
    run<3>("v_mad_u32_u24", d_out, h);
    run<16>("v_mul_u32_u24", d_out, h);
    run<4>("v_mul_hi_u32_u24", d_out, h);
    run<15>("v_dot4_u32_u8", d_out, h);
    run<9>("v_fma_f64", d_out, h);
    run<18>("v_mul_f64", d_out, h);
    return 0;
}
