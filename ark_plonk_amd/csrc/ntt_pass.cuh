// NTT pass kernels (templates).  Instantiated once per (curve, radix exponent S) in
// ntt_pass_inst.hip so the heavy kernels build in parallel; see ntt.hip for the algorithm notes.
#pragma once
#include <hip/hip_runtime.h>
#include "curve_params.h"
#include "field.cuh"


// ------------------------------------------------------------------------------------------------
template <class Fr>
ZK_D Fr ld_fr(const void* base, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + 2 * idx;
    uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <class Fr>
ZK_D void st_fr(void* base, uint64_t idx, const Fr& r) {
    uint4* q = reinterpret_cast<uint4*>(base) + 2 * idx;
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

ZK_D uint32_t lds_pad(uint32_t slot) { return slot + ((slot >> 4) << 1); }

template <class Fr>
ZK_D void lds_put(uint4* lo, uint4* hi, uint32_t slot, const Fr& r) {
    uint32_t p = lds_pad(slot);
    lo[p] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    hi[p] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}
template <class Fr>
ZK_D Fr lds_get(const uint4* lo, const uint4* hi, uint32_t slot) {
    uint32_t p = lds_pad(slot);
    uint4 a = lo[p], b = hi[p];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}

struct NttPassArgs {
    const void* in;
    void* out;
    const void* tw_inner;   // L/2 entries w_L^j (or inverse)
    const void* tw_pass;    // inter-pass table [k][col], nullptr on the last pass
    const void* pre_mul;    // g^j table (first pass of coset_fft) or nullptr
    const void* post_mul;   // g^-j table (last pass of coset_ifft) or nullptr
    uint32_t scale[8];      // 1/N (single-pass inverse only)
    int has_scale;
    uint64_t in_len;        // valid elements of `in` (first pass); N otherwise
    uint32_t log_n;
    uint32_t logc;          // log2 of tile columns (non-final) / gathered blocks (final)
    uint32_t log_m;         // non-final: log2 of the row stride M
    uint32_t log_mprev;     // non-final: log2 of the block this pass transforms (M * L)
    uint32_t s1;            // final: size (bits) of the most significant digit of the block index
};

__host__ __device__ inline uint32_t bitrev32(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> (32 - bits);
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}

// One window of up to three DIF stages on the 8 registers of a lane.
//   B0   : lowest row bit covered by the window; element e <-> row bits [B0, B0+3)
//   NST  : number of stages performed (row bits B0+NST-1 .. B0), 1..3
template <class Fr, int S, int B0, int NST>
ZK_D void dif_window(Fr (&x)[8], uint32_t v, const uint4* tw_lo, const uint4* tw_hi) {
    const uint32_t vlow = v & ((1u << B0) - 1u);
#pragma unroll
    for (int lb = NST - 1; lb >= 0; --lb) {
        constexpr int dummy = 0;
        (void)dummy;
        const int beta = B0 + lb;                 // row bit paired by this stage
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e & (1 << lb)) continue;
            const int eo = e | (1 << lb);
            // twiddle exponent: (row mod 2^beta) * 2^(S-1-beta)
            const uint32_t rowlow = ((uint32_t)(e & ((1 << lb) - 1)) << B0) | vlow;
            const uint32_t j = rowlow << (S - 1 - beta);
            Fr a = x[e], b = x[eo];
            x[e] = Fr::add(a, b);
            Fr d = Fr::sub(a, b);
            uint4 wl = tw_lo[j], wh = tw_hi[j];
            Fr w;
            w.v[0] = wl.x; w.v[1] = wl.y; w.v[2] = wl.z; w.v[3] = wl.w;
            w.v[4] = wh.x; w.v[5] = wh.y; w.v[6] = wh.z; w.v[7] = wh.w;
            x[eo] = Fr::mul(d, w);
        }
    }
}

template <int B0>
ZK_D uint32_t window_row(uint32_t v, uint32_t e) {
    return ((v >> B0) << (B0 + 3)) | (e << B0) | (v & ((1u << B0) - 1u));
}

// Run all DIF stages of a 2^S-point transform.  On entry x[e] holds row e*(L/8)+v (window
// B0 = S-3); on exit x[e] holds position row = 8*v+e of the bit-reversed-order result.
// SLOT(row) maps a row of this lane's column to an LDS slot.
template <class Fr, int S, int REM, class SlotFn>
ZK_D void dif_all(Fr (&x)[8], uint32_t v, uint4* d_lo, uint4* d_hi, const uint4* tw_lo, const uint4* tw_hi, SlotFn slot) {
    if constexpr (REM >= 3) {
        constexpr int B0 = REM - 3;
        dif_window<Fr, S, B0, 3>(x, v, tw_lo, tw_hi);
        if constexpr (B0 > 0) {
            constexpr int NB0 = (B0 >= 3) ? B0 - 3 : 0;
            // exchange: write rows of this window, read rows of the next
#pragma unroll
            for (int e = 0; e < 8; ++e) lds_put<Fr>(d_lo, d_hi, slot(window_row<B0>(v, e)), x[e]);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = lds_get<Fr>(d_lo, d_hi, slot(window_row<NB0>(v, e)));
            dif_all<Fr, S, B0, SlotFn>(x, v, d_lo, d_hi, tw_lo, tw_hi, slot);
        }
    } else if constexpr (REM > 0) {
        dif_window<Fr, S, 0, REM>(x, v, tw_lo, tw_hi);
    }
}

// ---------------------------------------------------------------------------------- non-final pass
template <class Fr, int S>
__global__ void ntt_pass_mid(NttPassArgs a) {
    constexpr uint32_t L = 1u << S;
    extern __shared__ uint4 smem[];
    const uint32_t logc = a.logc;
    const uint32_t C = 1u << logc;
    const uint32_t nslots = lds_pad(L * C) + 2;
    uint4* d_lo = smem;
    uint4* d_hi = smem + nslots;
    uint4* tw_lo = smem + 2 * nslots;
    uint4* tw_hi = tw_lo + L / 2;

    const uint32_t tid = threadIdx.x;
    const uint32_t T = blockDim.x;
    for (uint32_t j = tid; j < L / 2; j += T) {
        const uint4* q = reinterpret_cast<const uint4*>(a.tw_inner) + 2 * j;
        tw_lo[j] = q[0];
        tw_hi[j] = q[1];
    }
    const uint32_t c = tid & (C - 1);
    const uint32_t v = tid >> logc;
    const uint64_t tile = blockIdx.x;
    const uint32_t lcg = a.log_m - logc;                    // log2(column groups per block)
    const uint64_t blk = tile >> lcg;
    const uint64_t cg = tile & ((1ull << lcg) - 1);
    const uint64_t col = (cg << logc) + c;
    const uint64_t base = (blk << a.log_mprev) + col;

    Fr x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t row = (uint64_t)e * (L / 8) + v;
        const uint64_t idx = base + (row << a.log_m);
        if (idx < a.in_len) {
            x[e] = ld_fr<Fr>(a.in, idx);
            if (a.pre_mul) x[e] = Fr::mul(x[e], ld_fr<Fr>(a.pre_mul, idx));
        } else {
            x[e] = Fr::zero();
        }
    }
    __syncthreads();  // inner twiddles staged
    auto slot = [=](uint32_t row) -> uint32_t { return (row << logc) | c; };
    dif_all<Fr, S, S>(x, v, d_lo, d_hi, tw_lo, tw_hi, slot);

#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t row = (v << 3) | e;
        const uint64_t k = bitrev32(row, S);
        Fr w = ld_fr<Fr>(a.tw_pass, (k << a.log_m) + col);
        st_fr<Fr>(a.out, base + (k << a.log_m), Fr::mul(x[e], w));
    }
}

// -------------------------------------------------------------------------------------- final pass
template <class Fr, int S>
__global__ void ntt_pass_final(NttPassArgs a) {
    constexpr uint32_t L = 1u << S;
    extern __shared__ uint4 smem[];
    const uint32_t logc = a.logc;
    const uint32_t C = 1u << logc;
    const uint32_t nslots = lds_pad(L * C) + 2;
    uint4* d_lo = smem;
    uint4* d_hi = smem + nslots;
    uint4* tw_lo = smem + 2 * nslots;
    uint4* tw_hi = tw_lo + L / 2;

    const uint32_t tid = threadIdx.x;
    const uint32_t T = blockDim.x;
    for (uint32_t j = tid; j < L / 2; j += T) {
        const uint4* q = reinterpret_cast<const uint4*>(a.tw_inner) + 2 * j;
        tw_lo[j] = q[0];
        tw_hi[j] = q[1];
    }
    // block index digits: b = k1 * 2^log_rest + rho ; this tile gathers C consecutive k1
    const uint32_t log_nb = a.log_n - S;
    const uint32_t log_rest = log_nb - a.s1;
    const uint64_t tile = blockIdx.x;
    const uint32_t lg = a.s1 - logc;
    const uint64_t rho = tile >> lg;
    const uint64_t g = tile & ((1ull << lg) - 1);

    // load mapping: lanes run along the (contiguous) row axis
    const uint32_t v = tid & (L / 8 - 1);
    const uint32_t c = tid >> (S - 3);
    const uint64_t k1 = (g << logc) + c;
    const uint64_t b = (k1 << log_rest) | rho;
    Fr x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t row = (uint64_t)e * (L / 8) + v;
        const uint64_t idx = (b << S) + row;
        if (idx < a.in_len) {
            x[e] = ld_fr<Fr>(a.in, idx);
            if (a.pre_mul) x[e] = Fr::mul(x[e], ld_fr<Fr>(a.pre_mul, idx));
        } else {
            x[e] = Fr::zero();
        }
    }
    __syncthreads();
    auto slot = [=](uint32_t row) -> uint32_t { return (c << S) | row; };
    dif_all<Fr, S, S>(x, v, d_lo, d_hi, tw_lo, tw_hi, slot);

    // transpose through LDS so that stores run along the gathered-digit axis
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_put<Fr>(d_lo, d_hi, slot((v << 3) | e), x[e]);
    __syncthreads();
    const uint32_t c2 = tid & (C - 1);
    const uint32_t j2 = tid >> logc;
    const uint64_t k1o = (g << logc) + c2;
    const uint64_t obase = k1o + (rho << a.s1);   // digit-reversed block index
    Fr sc;
#pragma unroll
    for (int i = 0; i < 8; ++i) sc.v[i] = a.scale[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t k = j2 + (uint32_t)i * (L / 8);
        const uint32_t row = bitrev32(k, S);
        Fr y = lds_get<Fr>(d_lo, d_hi, (c2 << S) | row);
        const uint64_t oidx = obase + ((uint64_t)k << log_nb);
        if (a.has_scale) y = Fr::mul(y, sc);
        if (a.post_mul) y = Fr::mul(y, ld_fr<Fr>(a.post_mul, oidx));
        st_fr<Fr>(a.out, oidx, y);
    }
}

// tiny transforms (N = 1, 2, 4): one lane, straight from the definition
template <class Fr>
__global__ void ntt_tiny(NttPassArgs a, Fr w /* w_N */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n = 1u << a.log_n;
    Fr x[4], y[4];
    for (uint32_t i = 0; i < n; ++i) {
        x[i] = i < a.in_len ? ld_fr<Fr>(a.in, i) : Fr::zero();
        if (a.pre_mul && i < a.in_len) x[i] = Fr::mul(x[i], ld_fr<Fr>(a.pre_mul, i));
    }
    Fr sc;
    for (int i = 0; i < 8; ++i) sc.v[i] = a.scale[i];
    Fr wi = Fr::one();  // w^i
    for (uint32_t i = 0; i < n; ++i) {
        Fr acc = Fr::zero();
        Fr wij = Fr::one();
        for (uint32_t j = 0; j < n; ++j) {
            acc = Fr::add(acc, Fr::mul(x[j], wij));
            wij = Fr::mul(wij, wi);
        }
        if (a.has_scale) acc = Fr::mul(acc, sc);
        if (a.post_mul) acc = Fr::mul(acc, ld_fr<Fr>(a.post_mul, i));
        y[i] = acc;
        wi = Fr::mul(wi, w);
    }
    for (uint32_t i = 0; i < n; ++i) st_fr<Fr>(a.out, i, y[i]);
}

