// Would pairwise AFFINE additions with a shared inversion (Montgomery's trick) beat msm_accumulate's XYZZ mixed additions on this card?
// A measurement, not a product path (nothing under ark_plonk_amd/ uses it).  profiles/r04/r04_notes.md holds the numbers and the verdict.
//
// The idea: the references of a bucket are sorted next to each other, so a level of a pairwise reduction tree is a list of INDEPENDENT
// additions P + Q.  In affine coordinates one of them costs lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2,
// y3 = lambda (x1 - x3) - y1: 2 products + 1 square + the inverse of x2 - x1.  A lane that walks k pairs shares ONE inversion among
// them: a running product of the denominators on the way forth (1 product each, the prefix kept), the inverse of the total, and on
// the way back 2 products per pair to peel its own inverse off.  5 products + 1 square (1950 limb products) against the mixed
// addition's 3055 -- plus the inversion's share, plus what the XYZZ form never pays: the prefix goes out to HBM and comes back
// (k is in the hundreds: nothing on chip holds it), the operands are fetched twice, and every level writes its sums out.
//
// What runs here, on random field elements (the chord formulas never use the curve equation, so any pairs with distinct x do):
//   xyzz   k mixed additions per lane into one XYZZ accumulator, points gathered from a table by random references: the shape of
//          msm_accumulate (its references are sorted by bucket, which does not make the table rows any less scattered)
//   level0 k pairs per lane, both operands gathered from the table                 (first level of the tree)
//   level1 k pairs per lane, operands = neighbours in the previous level's output  (every later level)
//   inv    nothing but the inversions (Fermat, as fields.cuh has it), to price their share
//   check  level0's sums against from_affine + madd + to_affine, canonical words compared
// Usage: affine_probe <mode: xyzz|level0|level1|inv|check> <lanes> <k> [table log2, default 22] [reps, default 5]
// Build: hipcc --offload-arch=gfx950 -O3 -I ark_plonk_amd/csrc tools/affine_probe.hip -o tools/bin/affine_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "curve_params.h"
#include "ecu.cuh"

typedef Fs<FqBls12_381SParams> F;
constexpr int U4 = 4;     // a coordinate: 13 limbs in 16 words, as msm.hip stores them

ZK_D F ld_f(const uint4* q) {
    F r;
#pragma unroll
    for (int i = 0; i < U4; ++i) {
        const uint4 a = q[i];
        if (4 * i + 0 < F::NL) r.v[4 * i + 0] = a.x;
        if (4 * i + 1 < F::NL) r.v[4 * i + 1] = a.y;
        if (4 * i + 2 < F::NL) r.v[4 * i + 2] = a.z;
        if (4 * i + 3 < F::NL) r.v[4 * i + 3] = a.w;
    }
    return r;
}
ZK_D void st_f(uint4* q, const F& r) {
#pragma unroll
    for (int i = 0; i < U4; ++i) {
        uint4 a;
        a.x = 4 * i + 0 < F::NL ? r.v[4 * i + 0] : 0u;
        a.y = 4 * i + 1 < F::NL ? r.v[4 * i + 1] : 0u;
        a.z = 4 * i + 2 < F::NL ? r.v[4 * i + 2] : 0u;
        a.w = 4 * i + 3 < F::NL ? r.v[4 * i + 3] : 0u;
        q[i] = a;
    }
}
// prefix products, word-interleaved over the lanes: entry i of lane t is four 16-byte pieces at [(4 i + j) * n_lanes + t]
ZK_D F ld_pref(const uint4* s, uint64_t i, uint32_t t, uint32_t n_lanes) {
    F r;
#pragma unroll
    for (int j = 0; j < U4; ++j) {
        const uint4 a = s[(4 * i + j) * n_lanes + t];
        if (4 * j + 0 < F::NL) r.v[4 * j + 0] = a.x;
        if (4 * j + 1 < F::NL) r.v[4 * j + 1] = a.y;
        if (4 * j + 2 < F::NL) r.v[4 * j + 2] = a.z;
        if (4 * j + 3 < F::NL) r.v[4 * j + 3] = a.w;
    }
    return r;
}
ZK_D void st_pref(uint4* s, uint64_t i, uint32_t t, uint32_t n_lanes, const F& r) {
#pragma unroll
    for (int j = 0; j < U4; ++j) {
        uint4 a;
        a.x = 4 * j + 0 < F::NL ? r.v[4 * j + 0] : 0u;
        a.y = 4 * j + 1 < F::NL ? r.v[4 * j + 1] : 0u;
        a.z = 4 * j + 2 < F::NL ? r.v[4 * j + 2] : 0u;
        a.w = 4 * j + 3 < F::NL ? r.v[4 * j + 3] : 0u;
        s[(4 * i + j) * n_lanes + t] = a;
    }
}

// |v| < 8p with almost-balanced limbs -> the same residue in about (-p/2, p/2), strict limbs: the affine sums feed the next level's
// differences, so unlike an XYZZ coordinate (always a sum of fresh products) they would grow by a few p per level
ZK_D F weak_reduce(const F& v) {
    const float top = (float)(int32_t)v.v[F::NL - 1];
    const int32_t q = (int32_t)rintf(top * (1.0f / (float)FqBls12_381SParams::MOD(F::NL - 1)));
    F r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < F::NL - 1; ++i) {
        const int64_t t = (int64_t)(int32_t)v.v[i] - (int64_t)q * FqBls12_381SParams::MOD(i) + carry;
        const int32_t lo = F::sx((uint32_t)t);
        r.v[i] = (uint32_t)lo;
        carry = (t - lo) >> 30;
    }
    r.v[F::NL - 1] = (uint32_t)((int64_t)(int32_t)v.v[F::NL - 1] - (int64_t)q * FqBls12_381SParams::MOD(F::NL - 1) + carry);
    return r;
}

// ------------------------------------------------------------------------------------------------------------------ kernels
__global__ void __launch_bounds__(256) fill_table(uint4* table, uint64_t n_points, uint64_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_points) return;
    uint64_t s = seed + i * 0x9e3779b97f4a7c15ull;
    auto next = [&]() {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    for (int c = 0; c < 2; ++c) {
        F v;
        for (int l = 0; l < F::NL; ++l) v.v[l] = (uint32_t)F::sx((uint32_t)next());
        v.v[F::NL - 1] = (uint32_t)((int32_t)(next() % 3000000u) - 1500000);      // |value| < p: the top limb of p is 1704210
        st_f(table + i * 2 * U4 + c * U4, F::mul(v, F::one()));                   // a strict residue in (-p, p), as the table holds them
    }
}
__global__ void __launch_bounds__(256) fill_refs(uint32_t* refs, uint64_t n, uint32_t mask, uint64_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + i * 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    uint32_t r = (uint32_t)(z ^ (z >> 31)) & mask;
    if (i & 1) {      // the two operands of a pair are different table rows, as two references of one bucket are
        uint64_t y = seed + (i - 1) * 0x9e3779b97f4a7c15ull;
        y = (y ^ (y >> 30)) * 0xbf58476d1ce4e5b9ull;
        y = (y ^ (y >> 27)) * 0x94d049bb133111ebull;
        if (((uint32_t)(y ^ (y >> 31)) & mask) == r) r ^= 1u;
    }
    refs[i] = r;
}

// the shape of msm_accumulate: a chain of k mixed additions per lane, the next point requested before the current addition
__global__ void __launch_bounds__(128) k_xyzz(const uint32_t* refs, const uint4* table, uint4* out, uint32_t k, uint32_t n_lanes) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const uint32_t* r = refs + (uint64_t)t * k;
    XYZZu<F> acc = XYZZu<F>::infinity();
    AffineU<F> pn;
    pn.x = ld_f(table + (uint64_t)r[0] * 2 * U4);
    pn.y = ld_f(table + (uint64_t)r[0] * 2 * U4 + U4);
    for (uint32_t i = 0; i < k; ++i) {
        const AffineU<F> p = pn;
        if (i + 1 < k) {
            pn.x = ld_f(table + (uint64_t)r[i + 1] * 2 * U4);
            pn.y = ld_f(table + (uint64_t)r[i + 1] * 2 * U4 + U4);
        }
        acc = XYZZu<F>::madd(acc, p);
    }
    uint4* o = out + (uint64_t)t * 4 * U4;
    st_f(o, acc.x);
    st_f(o + U4, acc.y);
    st_f(o + 2 * U4, acc.zz);
    st_f(o + 3 * U4, acc.zzz);
}

// one level of the tree: pair i of lane t is number i * n_lanes + t (neighbouring lanes take neighbouring pairs)
template <bool GATHER>
__global__ void __launch_bounds__(128) k_level(const uint32_t* refs, const uint4* src, uint4* dst, uint4* scratch, uint32_t k, uint32_t n_lanes,
                                               uint32_t* flag) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    auto idx = [&](uint64_t pair, int which) -> uint64_t { return GATHER ? (uint64_t)refs[2 * pair + which] : 2 * pair + which; };
    // forth: running product of the denominators
    F run = F::one();
    {
        uint64_t a = idx(t, 0), b = idx(t, 1);
        F x1 = ld_f(src + a * 2 * U4), x2 = ld_f(src + b * 2 * U4);
        for (uint32_t i = 0; i < k; ++i) {
            const F d = F::sub(x2, x1);
            if (i + 1 < k) {
                const uint64_t pair = (uint64_t)(i + 1) * n_lanes + t;
                a = idx(pair, 0);
                b = idx(pair, 1);
                x1 = ld_f(src + a * 2 * U4);
                x2 = ld_f(src + b * 2 * U4);
            }
            run = F::mul(run, d);
            st_pref(scratch, i, t, n_lanes, run);
        }
    }
    if (run.is_zero_mod_reduced()) atomicOr(flag, 1u);      // two operands with one x: the product path would fall back to XYZZ
    F inv = F::inverse(run);
    // back: peel one inverse per pair, finish the addition
    for (int64_t i = (int64_t)k - 1; i >= 0; --i) {
        const uint64_t pair = (uint64_t)i * n_lanes + t;
        const uint64_t a = idx(pair, 0), b = idx(pair, 1);
        const F prev = i > 0 ? ld_pref(scratch, (uint64_t)i - 1, t, n_lanes) : F::one();
        const F x1 = ld_f(src + a * 2 * U4), y1 = ld_f(src + a * 2 * U4 + U4);
        const F x2 = ld_f(src + b * 2 * U4), y2 = ld_f(src + b * 2 * U4 + U4);
        const F d = F::sub(x2, x1);
        const F inv_d = F::mul(inv, prev);
        inv = F::mul(inv, d);
        const F lam = F::mul(F::sub(y2, y1), inv_d);
        const F x3 = weak_reduce(F::sub_sum3(F::sqr(lam), x1, x2, F::zero()));
        const F y3 = weak_reduce(F::sub(F::mul(lam, F::sub(x1, x3)), y1));
        st_f(dst + pair * 2 * U4, x3);
        st_f(dst + pair * 2 * U4 + U4, y3);
    }
}

__global__ void __launch_bounds__(128) k_inv(const uint4* table, uint4* out, uint32_t n_lanes) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    st_f(out + (uint64_t)t * U4, F::inverse(ld_f(table + (uint64_t)t * 2 * U4)));
}

// level0's sums against the XYZZ law, as canonical words
__global__ void __launch_bounds__(128) k_check(const uint32_t* refs, const uint4* table, const uint4* dst, uint64_t n_pairs, uint32_t* bad) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    AffineU<F> a, b, s;
    a.x = ld_f(table + (uint64_t)refs[2 * p] * 2 * U4);
    a.y = ld_f(table + (uint64_t)refs[2 * p] * 2 * U4 + U4);
    b.x = ld_f(table + (uint64_t)refs[2 * p + 1] * 2 * U4);
    b.y = ld_f(table + (uint64_t)refs[2 * p + 1] * 2 * U4 + U4);
    XYZZu<F>::madd(XYZZu<F>::from_affine(a), b).to_affine(s);
    uint32_t w0[12], w1[12];
    s.x.to_sat(w0);
    ld_f(dst + p * 2 * U4).to_sat(w1);
    bool ok = true;
    for (int i = 0; i < 12; ++i) ok = ok && w0[i] == w1[i];
    s.y.to_sat(w0);
    ld_f(dst + p * 2 * U4 + U4).to_sat(w1);
    for (int i = 0; i < 12; ++i) ok = ok && w0[i] == w1[i];
    if (!ok) atomicAdd(bad, 1u);
}

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: affine_probe xyzz|level0|level1|inv|check <lanes> <k> [table log2 = 22] [reps = 5]\n");
        return 2;
    }
    const char* mode = argv[1];
    const uint32_t lanes = (uint32_t)atol(argv[2]);
    const uint32_t k = (uint32_t)atol(argv[3]);
    const int tlog = argc > 4 ? atoi(argv[4]) : 22;
    const int reps = argc > 5 ? atoi(argv[5]) : 5;
    if (lanes == 0 || lanes % 128 || k == 0 || tlog < 10 || tlog > 24 || (uint64_t)lanes * k > (1ull << 27)) {
        fprintf(stderr, "lanes: a multiple of 128; lanes * k <= 2^27; table log2 in 10..24\n");
        return 2;
    }
    const uint64_t T = 1ull << tlog, units = (uint64_t)lanes * k;
    const bool lvl0 = !strcmp(mode, "level0") || !strcmp(mode, "check"), lvl1 = !strcmp(mode, "level1");
    const bool xyzz = !strcmp(mode, "xyzz"), inv = !strcmp(mode, "inv");
    if (!(lvl0 || lvl1 || xyzz || inv)) return 2;
    uint4 *table = nullptr, *src1 = nullptr, *dst = nullptr, *scratch = nullptr;
    uint32_t *refs = nullptr, *flag = nullptr;
    CK(hipMalloc(&table, T * 128));
    CK(hipMalloc(&flag, 8));
    CK(hipMemset(flag, 0, 8));
    hipLaunchKernelGGL(fill_table, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, 0, table, T, 0x5eedull);
    const uint64_t n_refs = xyzz ? units : 2 * units;
    CK(hipMalloc(&refs, n_refs * 4));
    hipLaunchKernelGGL(fill_refs, dim3((unsigned)((n_refs + 255) / 256)), dim3(256), 0, 0, refs, n_refs, (uint32_t)(T - 1), 0xabcdefull);
    CK(hipMalloc(&dst, (xyzz ? (uint64_t)lanes * 256 : units * 128) + 4096));
    if (lvl0 || lvl1) CK(hipMalloc(&scratch, units * 64));
    if (lvl1) {      // operands: 2 * units points laid out as a previous level would leave them
        CK(hipMalloc(&src1, 2 * units * 128));
        hipLaunchKernelGGL(fill_table, dim3((unsigned)((2 * units + 255) / 256)), dim3(256), 0, 0, src1, 2 * units, 0x1234ull);
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&]() {
        const dim3 g(lanes / 128), b(128);
        if (xyzz) hipLaunchKernelGGL(k_xyzz, g, b, 0, 0, refs, table, dst, k, lanes);
        else if (lvl0) hipLaunchKernelGGL(k_level<true>, g, b, 0, 0, refs, table, dst, scratch, k, lanes, flag);
        else if (lvl1) hipLaunchKernelGGL(k_level<false>, g, b, 0, 0, nullptr, src1, dst, scratch, k, lanes, flag);
        else hipLaunchKernelGGL(k_inv, g, b, 0, 0, table, dst, lanes);
    };
    launch();
    CK(hipDeviceSynchronize());
    const auto w0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    uint32_t h_flag[2] = {0, 0};
    if (!strcmp(mode, "check")) {
        hipLaunchKernelGGL(k_check, dim3((unsigned)((units + 127) / 128)), dim3(128), 0, 0, refs, table, dst, units, flag + 1);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h_flag, flag, 8, hipMemcpyDeviceToHost));
    if (inv)
        printf("affine_probe inv: %u lanes, %.3f ms per launch, %.1f us of one lane's time per inversion at this occupancy\n", lanes, ms, ms * 1e3);
    else
        printf("affine_probe %s: %u lanes x %u = %llu additions, table 2^%d, %.3f ms per launch, %.3f G additions/s (wall %.2f s for %d)%s\n", mode,
               lanes, k, (unsigned long long)units, tlog, ms, units / (ms * 1e6), wall, reps, h_flag[0] ? "  [zero denominator seen]" : "");
    if (!strcmp(mode, "check")) {
        printf("check: %u of %llu sums differ from the XYZZ law\n", h_flag[1], (unsigned long long)units);
        return h_flag[1] ? 1 : 0;
    }
    return 0;
}
