#include <hip/hip_runtime.h>
#include "curve_params.h"
#include "field.cuh"
#include "fieldu.cuh"
#include "ecu.cuh"
typedef Fu<FqBls12_381UParams> FqU;
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef WAVES
#define WAVES 2
#endif
#define SB()

// mixed addition with the steps in the order that keeps the fewest values live, fenced so that the scheduler does not
// interleave independent products (which is what takes the inlined madd to 220 registers)
// fast mixed addition: p finite, p != +-q assumed; returns false (p unspecified) when that assumption fails (ZZ3 == 0 mod p)
template <class F>
ZK_D bool madd_fast(XYZZu<F>& p, const AffineU<F>& q) {
    F s2 = F::mul(q.y, p.zzz);
    F r_ = F::sub16(s2, p.y);
    SB();
    F u2 = F::mul(q.x, p.zz);
    F pp_ = F::sub16(u2, p.x);
    SB();
    F pp = F::sqr(pp_);
    SB();
    p.zz = F::mul(p.zz, pp);
    const bool ok = !p.zz.is_zero_mod_reduced();
    SB();
    F rr = F::sqr(r_);
    SB();
    F ppp = F::mul(pp_, pp);
    SB();
    F qq = F::mul(p.x, pp);
    SB();
    p.zzz = F::mul(p.zzz, ppp);
    SB();
    p.x = F::sub8(rr, F::add3(ppp, qq, qq));
    SB();
    p.y = F::dot2(r_, F::sub16(qq, p.x), F::neg16(p.y), ppp);
    SB();
    return ok;
}

template <class F>
ZK_D void madd_tight(XYZZu<F>& p, const AffineU<F>& q) {
    if (VARIANT != 4 && p.is_inf()) { p = XYZZu<F>::from_affine(q); return; }
    F s2 = F::mul(q.y, p.zzz);
    F r_ = F::sub16(s2, p.y);
    SB();
    F u2 = F::mul(q.x, p.zz);
    F pp_ = F::sub16(u2, p.x);
    SB();
    F pp = F::sqr(pp_);
    SB();
    F zz3 = F::mul(p.zz, pp);
    SB();
    F rr = F::sqr(r_);
    if (VARIANT != 4 && zz3.is_zero_mod_reduced()) {
        if (rr.is_zero_mod_reduced()) p = XYZZu<F>::dbl_affine(q);
        else p = XYZZu<F>::infinity();
        return;
    }
    p.zz = zz3;
    SB();
    F ppp = F::mul(pp_, pp);
    SB();
    F qq = F::mul(p.x, pp);
    SB();
    p.zzz = F::mul(p.zzz, ppp);
    SB();
    p.x = F::sub8(rr, F::add3(ppp, qq, qq));
    SB();
    p.y = F::dot2(r_, F::sub16(qq, p.x), F::neg16(p.y), ppp);
    SB();
}
namespace {
// ---- device storage of an Fu: NL limbs padded to a multiple of 4 words (16-byte vector access)
template <class F>
struct Store {
    static constexpr int U4 = (F::NL + 3) / 4;      // uint4 per field element
    static constexpr int WORDS = 4 * U4;
};
template <class F>
ZK_D F ld_fu(const uint4* q) {
    F r;
#pragma unroll
    for (int i = 0; i < Store<F>::U4; ++i) {
        uint4 a = q[i];
        if (4 * i + 0 < F::NL) r.v[4 * i + 0] = a.x;
        if (4 * i + 1 < F::NL) r.v[4 * i + 1] = a.y;
        if (4 * i + 2 < F::NL) r.v[4 * i + 2] = a.z;
        if (4 * i + 3 < F::NL) r.v[4 * i + 3] = a.w;
    }
    return r;
}
template <class F>
ZK_D void st_fu(uint4* q, const F& r) {
#pragma unroll
    for (int i = 0; i < Store<F>::U4; ++i) {
        uint4 a;
        a.x = 4 * i + 0 < F::NL ? r.v[4 * i + 0] : 0u;
        a.y = 4 * i + 1 < F::NL ? r.v[4 * i + 1] : 0u;
        a.z = 4 * i + 2 < F::NL ? r.v[4 * i + 2] : 0u;
        a.w = 4 * i + 3 < F::NL ? r.v[4 * i + 3] : 0u;
        q[i] = a;
    }
}
template <class F>
ZK_D AffineU<F> ld_affine(const void* bases, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(bases) + idx * (2 * Store<F>::U4);
    AffineU<F> p;
    p.x = ld_fu<F>(q);
    p.y = ld_fu<F>(q + Store<F>::U4);
    return p;
}
template <class F>
ZK_D XYZZu<F> ld_xyzz(const void* arr, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(arr) + idx * (4 * Store<F>::U4);
    XYZZu<F> p;
    p.x = ld_fu<F>(q);
    p.y = ld_fu<F>(q + Store<F>::U4);
    p.zz = ld_fu<F>(q + 2 * Store<F>::U4);
    p.zzz = ld_fu<F>(q + 3 * Store<F>::U4);
    return p;
}
template <class F>
ZK_D void st_xyzz(void* arr, uint64_t idx, const XYZZu<F>& p) {
    uint4* q = reinterpret_cast<uint4*>(arr) + idx * (4 * Store<F>::U4);
    st_fu<F>(q, p.x);
    st_fu<F>(q + Store<F>::U4, p.y);
    st_fu<F>(q + 2 * Store<F>::U4, p.zz);
    st_fu<F>(q + 3 * Store<F>::U4, p.zzz);
}

// one coordinate (role 0..3 = X, Y, ZZ, ZZZ) of a stored XYZZ point: the quad-cooperative kernels (ecq.cuh)
template <class F>
ZK_D F ld_coord(const void* arr, uint64_t idx, uint32_t role) {
    return ld_fu<F>(reinterpret_cast<const uint4*>(arr) + idx * (4 * Store<F>::U4) + role * Store<F>::U4);
}
template <class F>
ZK_D void st_coord(void* arr, uint64_t idx, uint32_t role, const F& c) {
    st_fu<F>(reinterpret_cast<uint4*>(arr) + idx * (4 * Store<F>::U4) + role * Store<F>::U4, c);
}



}

// Every lane sums entries [t*L, (t+1)*L) of the bucket-sorted reference list.
// PRE: references carry a window number and `bases` is the window-multiples table [W][n_srs]
// (row w holds 2^(c w) P_i); tab_stride = n_srs, tab_off = base_offset.
template <class F, bool PRE>
__global__ void __launch_bounds__(128, WAVES) msm_accumulate(const uint32_t* entries, const uint32_t* offsets, uint32_t nb, const void* bases,
                                                       void* buckets, void* part_pt, uint32_t L, uint32_t n_lanes, uint64_t tab_stride,
                                                       uint64_t tab_off) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    const uint32_t E = offsets[nb];
    const uint64_t e0 = (uint64_t)t * L;
    if (e0 >= E) return;
    const uint32_t e1 = (uint32_t)min((uint64_t)E, e0 + L);
    // largest b with offsets[b] <= e0
    uint32_t lo = 0, hi = nb - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= (uint32_t)e0) lo = mid; else hi = mid - 1;
    }
    uint32_t b = lo;
    uint32_t bend = offsets[b + 1];
    const bool head_partial = offsets[b] < (uint32_t)e0;
    bool first_run = true;
    bool fresh = true, bad = false;
    XYZZu<F> acc = XYZZu<F>::infinity();
    // software pipeline: the reference and the 128-byte point of iteration e+1 are requested before the
    // mixed addition of iteration e (two dependent HBM/L2 round trips otherwise sit in front of every add)
    auto point_index = [&](uint32_t ref) -> uint64_t {
        return PRE ? (uint64_t)((ref >> 26) & 31u) * tab_stride + tab_off + (ref & 0x3ffffffu) : (uint64_t)(ref & 0x7fffffffu);
    };
#if VARIANT >= 2 && VARIANT != 6
    // The prefetched point of iteration e+1 is parked in LDS by direct global->LDS loads (no VGPRs held across the addition):
    // 8 pieces of 16 B per lane, piece k of lane l at pbuf[wave][k][l].
    __shared__ uint4 pbuf[2][8][64];
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    auto prefetch = [&](uint32_t ref) {
        const uint4* src = reinterpret_cast<const uint4*>(bases) + point_index(ref) * (2 * Store<F>::U4);
#pragma unroll
        for (int k = 0; k < 2 * Store<F>::U4; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k),
                                             (__attribute__((address_space(3))) void*)&pbuf[wv][k][0], 16, 0, 0);
    };
    auto take = [&]() -> AffineU<F> {
        AffineU<F> p;
        uint4 w[2 * Store<F>::U4];
#pragma unroll
        for (int k = 0; k < 2 * Store<F>::U4; ++k) w[k] = pbuf[wv][k][ln];
#pragma unroll
        for (int i = 0; i < Store<F>::U4; ++i) {
            if (4 * i + 0 < F::NL) { p.x.v[4 * i + 0] = w[i].x; p.y.v[4 * i + 0] = w[Store<F>::U4 + i].x; }
            if (4 * i + 1 < F::NL) { p.x.v[4 * i + 1] = w[i].y; p.y.v[4 * i + 1] = w[Store<F>::U4 + i].y; }
            if (4 * i + 2 < F::NL) { p.x.v[4 * i + 2] = w[i].z; p.y.v[4 * i + 2] = w[Store<F>::U4 + i].z; }
            if (4 * i + 3 < F::NL) { p.x.v[4 * i + 3] = w[i].w; p.y.v[4 * i + 3] = w[Store<F>::U4 + i].w; }
        }
        return p;
    };
    uint32_t ref_n = entries[(uint32_t)e0];
    prefetch(ref_n);
    for (uint32_t e = (uint32_t)e0; e < e1; ++e) {
        const uint32_t ref = ref_n;
        AffineU<F> p = take();                         // waits for the DMA of the previous iteration
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS reads are done before the buffer is overwritten
        if (e + 1 < e1) {
            ref_n = entries[e + 1];
            prefetch(ref_n);
        }
#else
    uint32_t ref_n = entries[(uint32_t)e0];
    AffineU<F> p_n = ld_affine<F>(bases, point_index(ref_n));
    for (uint32_t e = (uint32_t)e0; e < e1; ++e) {
        const uint32_t ref = ref_n;
        AffineU<F> p = p_n;
        if (e + 1 < e1) {
            ref_n = entries[e + 1];
            p_n = ld_affine<F>(bases, point_index(ref_n));
        }
#endif
        if (e == bend) {
            if (first_run && head_partial) st_xyzz<F>(part_pt, 2ull * t, acc);
            else st_xyzz<F>(buckets, b, acc);
            first_run = false;
            acc = XYZZu<F>::infinity();
#if VARIANT == 5 || VARIANT == 6
            fresh = true;
#endif
            do {
                ++b;
                bend = offsets[b + 1];
            } while (bend <= e);
        }
        if (p.is_null()) continue;
        if (ref >> 31) p.y = F::neg_canonical(p.y);
#if VARIANT == 5 || VARIANT == 6
        if (fresh) {
            acc = XYZZu<F>::from_affine(p);
            fresh = false;
        } else if (!madd_fast<F>(acc, p)) {
            bad = true;
            break;
        }
#else
        if (VARIANT == 0 || VARIANT == 2) acc = XYZZu<F>::madd(acc, p); else madd_tight<F>(acc, p);
#endif
    }
    // Slot convention (msm_combine relies on it): a run that is the FIRST run of its chunk and is
    // not a whole bucket goes to slot 2t, a trailing incomplete run that is not the first goes to 2t+1.
    if (bad) { ((uint32_t*)part_pt)[0] = t; return; }   // stand-in for the redo queue
    const bool tail_complete = (e1 == bend);
    if (first_run) {
        if (head_partial || !tail_complete) st_xyzz<F>(part_pt, 2ull * t, acc);
        else st_xyzz<F>(buckets, b, acc);
    } else {
        if (tail_complete) st_xyzz<F>(buckets, b, acc);
        else st_xyzz<F>(part_pt, 2ull * t + 1, acc);
    }
}


template __global__ void msm_accumulate<FqU, true>(const uint32_t*, const uint32_t*, uint32_t, const void*, void*, void*, uint32_t, uint32_t, uint64_t, uint64_t);
