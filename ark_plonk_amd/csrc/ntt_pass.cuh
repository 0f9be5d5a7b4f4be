// NTT pass kernels (templates).  Instantiated once per (curve, radix exponent S) in
// ntt_pass_inst.hip so the heavy kernels build in parallel; see ntt.hip for the algorithm notes.
//
// One WAVEFRONT owns a tile of 512 elements = 2^S rows x 2^(9-S) columns and runs the 2^S-point
// decimation-in-time transform of each column entirely in its own registers: 8 elements per lane,
// three radix-2 stages per window, and between windows a lane<->register transpose through a 2 KiB
// wave-private LDS scratch (one 29-bit limb plane at a time -- no workgroup barrier, no LDS-resident
// tile, so occupancy is bounded by VGPRs only).  Arithmetic is the unsaturated 29-bit-limb Montgomery
// field (fieldu.cuh): data stay in the arkworks Montgomery domain (R = 2^256) because every twiddle
// table holds w * 2^261 mod r, and DIT butterflies (t = w*b; a + t, a - t + 2r) grow the lazily
// reduced values only additively, so no reduction is needed inside a pass (a Montgomery product accepts
// a * b < 69 r^2; the data stay below 47r and every twiddle below r).
// The butterflies of the first window whose twiddle is w^0 = 1 (all of stage 0, half of stage 1, a quarter
// of stage 2) are done without a product.
// Round 5: the butterflies are LAZILY NORMALISED too.  a + t and a - t + 3r are limb-wise operations with no carry sweep
// (fieldu.cuh `add_raw` / `sub_raw3`: the multiple of r is held with limbs in [2^29, 2^30), so no limb of the difference goes
// below zero); limbs grow by at most two "units" (2^29) per stage, a product takes a data operand of up to 6 units, a limb holds
// 8: ONE sweep of the eight registers every third stage (`ntt_sweep_at`) instead of two per butterfly -- ~36 of the ~315 vector
// instructions of a butterfly.  Passes other than the last store their products as they are (below 2r, not canonical).
// The products are Fu::mul_fenced: the same product with every column's chain started from the carry below it (the optimiser would
// otherwise build the columns as independent chains and add the carries afterwards): -1..-2.5 % vector instructions and 186 -> 144
// VGPRs in the mid pass (three waves per SIMD instead of two), 8.55 -> 8.43 ms per proof.
#pragma once
#include <hip/hip_runtime.h>
#include "curve_params.h"
#include "field.cuh"
#include "fieldu.cuh"

typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;

struct NttPassArgs {
    const void* in;
    void* out;
    const void* tw_inner;   // 2^S / 2 entries w_L^j (R'-form), already split into 29-bit limbs: 48 B per entry (ld_limbs)
    const void* tw_pass;    // inter-pass table [k][col] (R'-form), nullptr on the last pass
    const void* pre_mul;    // g^j table (first pass of coset_fft) or nullptr
    const void* post_mul;   // g^-j table (last pass of coset_ifft) or nullptr
    uint32_t scale[8];      // last pass: 1 or 1/N, R'-form packed (applied to every output)
    uint64_t in_len;        // valid elements of `in` (first pass); N otherwise
    uint32_t log_n;
    uint32_t logc;          // log2 of the columns (non-final) / gathered blocks (final) actually used, <= 9 - S
    uint32_t log_m;         // non-final: log2 of the row stride M
    uint32_t log_mprev;     // non-final: log2 of the block this pass transforms (M * L)
    uint32_t s1;            // final: size (bits) of the most significant digit of the block index
    uint32_t s2;            // final, four passes (log N >= 28): bits of the block index's second digit (0: at most three passes)
    uint32_t n_tiles;
    uint32_t quarter;       // first pass of a multi-pass transform whose input fills at most N/4 (coset_fft of n coefficients on
                            // the 4n domain): only rows < 2^S / 4 are non-zero, so the first two stages are plain copies
    // batch of independent transforms of one kind and size in ONE launch (blockIdx.y = polynomial): the four wire iffts of
    // prover.rs:196-203, the four sigma ffts of permutation/mod.rs:671-674, the coset ffts of quotient_poly.rs:72-120.  A single
    // 2^20 transform is only 2048 wavefront-tiles -- two per SIMD, short of the three the kernel's registers allow.
    uint32_t n_batch;       // 0: one transform (in / out / in_len above)
    const void* ins[16];
    void* outs[16];
    uint64_t in_lens[16];
};

__host__ __device__ inline uint32_t bitrev32(uint32_t x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? __brev(x) >> (32 - bits) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
#endif
}

// 32-byte packed integer -> limbs (no domain change)
template <class F>
ZK_D F ld_u(const void* base, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + 2 * idx;
    uint4 a = q[0], b = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return F::split_words(w);
}
// value < 2r -> canonical -> 32-byte store
template <class F>
ZK_D void st_u(void* base, uint64_t idx, const F& x) {
    uint32_t w[8];
    F::canonical_lt2p(x).pack_words(w);
    uint4* q = reinterpret_cast<uint4*>(base) + 2 * idx;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// swept value below 2^256 -> 32-byte store with no reduction (between the passes of one transform)
template <class F>
ZK_D void st_u_raw(void* base, uint64_t idx, const F& x) {
    uint32_t w[8];
    x.pack_words(w);
    uint4* q = reinterpret_cast<uint4*>(base) + 2 * idx;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// inner twiddles are stored as the 29-bit limbs the products consume (NL <= 12 words, padded to 48 B): the table is a few KiB
// and L1-resident, and every butterfly of a pass reads one entry -- splitting 32-byte words into limbs at each use was ~27 of
// the ~315 vector instructions of a butterfly
constexpr int TW_INNER_U4 = 3;
template <class F>
ZK_D F ld_limbs(const void* base, uint64_t idx) {
    static_assert(F::NL <= 4 * TW_INNER_U4, "inner twiddle entry too small");
    const uint4* q = reinterpret_cast<const uint4*>(base) + TW_INNER_U4 * idx;
    F r;
#pragma unroll
    for (int i = 0; i < TW_INNER_U4; ++i) {
        if (4 * i >= F::NL) break;
        const uint4 a = q[i];
        if (4 * i + 0 < F::NL) r.v[4 * i + 0] = a.x;
        if (4 * i + 1 < F::NL) r.v[4 * i + 1] = a.y;
        if (4 * i + 2 < F::NL) r.v[4 * i + 2] = a.z;
        if (4 * i + 3 < F::NL) r.v[4 * i + 3] = a.w;
    }
    return r;
}

// limb units (fieldu.cuh) of the eight registers BEFORE stage t of a pass: fresh loads / products have 1; a stage adds 2; a sweep
// (taken when a data operand would exceed the product's 6 units) brings them back to 1
constexpr int ntt_units_before(int t) {
    int b = 1;
    for (int s = 0; s < t; ++s) {
        if (b > 6) b = 1;
        b += 2;
    }
    return b;
}
constexpr bool ntt_sweep_at(int t) { return ntt_units_before(t) > 6; }
static_assert(ntt_units_before(1) == 3 && ntt_units_before(2) == 5 && ntt_units_before(3) == 7 && ntt_units_before(4) == 3, "two units per stage");
static_assert(!ntt_sweep_at(0) && !ntt_sweep_at(2) && ntt_sweep_at(3) && !ntt_sweep_at(5) && ntt_sweep_at(6) && ntt_sweep_at(9), "a sweep every third stage");

template <int B0>
ZK_D uint32_t window_pos(uint32_t v, uint32_t e) {
    return ((v >> B0) << (B0 + 3)) | (e << B0) | (v & ((1u << B0) - 1u));
}

// lane <-> register transpose through the wave's scratch plane, one limb at a time
template <class F>
ZK_D void wave_exchange(F (&x)[8], lds_u32* sc, const uint32_t (&widx)[8], const uint32_t (&ridx)[8]) {
#pragma unroll
    for (int l = 0; l < F::NL; ++l) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sc[widx[e]] = x[e].v[l];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e].v[l] = sc[ridx[e]];
        __builtin_amdgcn_wave_barrier();
    }
}

// DIT stages for the position bits B0+LB_LO .. B0+LB_HI-1 on the 8 registers of a lane
// (register e <-> position bits [B0, B0+3) ; v = the other position bits)
template <class F, int S, int B0, int LB_LO, int LB_HI>
ZK_D void dit_window(F (&x)[8], uint32_t v, const void* tw, bool quarter) {
    const uint32_t vlow = v & ((1u << B0) - 1u);
#pragma unroll
    for (int lb = LB_LO; lb < LB_HI; ++lb) {
        const int t = B0 + lb;   // position bit paired by this stage; twiddle w_{2^(t+1)}^(p mod 2^t)
        if (ntt_sweep_at(t)) {   // compile-time after unrolling: every third stage
#pragma unroll
            for (int e = 0; e < 8; ++e) F::sweep(x[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e & (1 << lb)) continue;
            const int eo = e | (1 << lb);
            if constexpr (B0 == 0 && LB_LO == 0) {
                // first stage of a pass (t = 0): every twiddle is w^0 = 1 and the operands are swept and < 2r
                // (fresh loads / products), so the butterfly is a + b, a - b + 3r with no product at all
                if (lb < 2 && quarter) {   // the partner is zero (or a copy of a zero-partnered value): a + w*0, a - w*0
                    x[eo] = x[e];
                    continue;
                }
                if (lb == 0) {
                    const F b = x[eo];
                    x[eo] = F::sub_raw3(x[e], b);
                    x[e] = F::add_raw(x[e], b);
                    continue;
                }
                // stages 1 and 2 of the first window: the twiddle index is (e mod 2^lb) << (S-1-lb), i.e. w^0 = 1
                // for the pairs with e mod 2^lb == 0.  Their subtrahend is no longer a product: below 5r after stage 0, below
                // 13r after stage 1 -- swept here, and taken from 8r / 16r; the pass still ends below 47r < 69r.
                if ((e & ((1 << lb) - 1)) == 0) {
                    F b = x[eo];
                    F::sweep(b);
                    x[eo] = lb == 1 ? F::sub_raw8(x[e], b) : F::sub_raw16(x[e], b);
                    x[e] = F::add_raw(x[e], b);
                    continue;
                }
            }
            const uint32_t plow = ((uint32_t)(e & ((1 << lb) - 1)) << B0) | vlow;
            const uint32_t j = plow << (S - 1 - t);
            F w = ld_limbs<F>(tw, j);
            F m = F::mul_fenced(x[eo], w);            // < 2r, swept
            x[eo] = F::sub_raw3(x[e], m);
            x[e] = F::add_raw(x[e], m);
        }
    }
}

// all windows of a 2^S-point DIT; on entry x[e] holds position 8v+e (window B0 = 0), on exit the
// top-window layout: position (e << (S-3)) | v  (S >= 3)
template <class F, int S, int B0>
ZK_D void dit_all(F (&x)[8], uint32_t v, uint32_t c, uint32_t LC, const void* tw, lds_u32* sc, bool quarter = false) {
    constexpr int REM = S - B0;          // position bits not yet processed
    if constexpr (REM >= 3) {
        dit_window<F, S, B0, 0, 3>(x, v, tw, quarter);
        if constexpr (REM > 3) {
            constexpr int NB0 = (REM - 3 >= 3) ? B0 + 3 : S - 3;
            uint32_t wi[8], ri[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                wi[e] = (window_pos<B0>(v, e) << LC) | c;
                ri[e] = (window_pos<NB0>(v, e) << LC) | c;
            }
            wave_exchange<F>(x, sc, wi, ri);
            if constexpr (REM - 3 >= 3) {
                dit_all<F, S, B0 + 3>(x, v, c, LC, tw, sc);
            } else {
                // last, partial window: registers cover position bits [S-3, S); only the top REM-3 are new
                dit_window<F, S, S - 3, 3 - (REM - 3), 3>(x, v, tw, false);
            }
        }
    }
}

// ---------------------------------------------------------------------------------- non-final pass
template <class F, int S>
__global__ void __launch_bounds__(256) ntt_pass_mid(NttPassArgs a) {
    constexpr uint32_t LC = 9 - S;
    __shared__ uint32_t scratch[4][512];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= a.n_tiles) return;
    const void* in = a.in;
    void* out = a.out;
    uint64_t in_len = a.in_len;
    if (a.n_batch) {
        in = a.ins[blockIdx.y];
        out = a.outs[blockIdx.y];
        in_len = a.in_lens[blockIdx.y];
    }
    lds_u32* sc = (lds_u32*)scratch[wv];
    const uint32_t c = lane & ((1u << LC) - 1u);
    const uint32_t v = lane >> LC;
    const bool active = c < (1u << a.logc);
    const uint32_t lcg = a.log_m - a.logc;                   // log2(column groups per block)
    const uint64_t blk = (uint64_t)tile >> lcg;
    const uint64_t cg = (uint64_t)tile & ((1ull << lcg) - 1);
    const uint64_t col = (cg << a.logc) + c;
    const uint64_t base = (blk << a.log_mprev) + col;

    F x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t row = bitrev32((v << 3) | e, S);      // DIT consumes bit-reversed rows
        const uint64_t idx = base + (row << a.log_m);
        if (active && idx < in_len) {
            x[e] = ld_u<F>(in, idx);
            if (a.pre_mul) x[e] = F::mul_fenced(x[e], ld_u<F>(a.pre_mul, idx));
        } else {
            x[e] = F::zero();
        }
    }
    dit_all<F, S, 0>(x, v, c, LC, a.tw_inner, sc, a.quarter != 0);
    if (!active) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t k = S >= 3 ? (((uint64_t)e << (S - 3)) | v) : e;
        F w = ld_u<F>(a.tw_pass, (k << a.log_m) + col);
        if (ntt_units_before(S) > 6) F::sweep(x[e]);
        // the next pass takes any representative below 2r: the product is stored as it is (swept limbs, 32 bytes)
        st_u_raw<F>(out, base + (k << a.log_m), F::mul_fenced(x[e], w));
    }
}

// -------------------------------------------------------------------------------------- final pass
template <class F, int S>
__global__ void __launch_bounds__(256) ntt_pass_final(NttPassArgs a) {
    constexpr uint32_t L = 1u << S;
    constexpr uint32_t LC = 9 - S;
    __shared__ uint32_t scratch[4][512];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + wv;
    if (tile >= a.n_tiles) return;
    const void* in = a.in;
    void* out = a.out;
    uint64_t in_len = a.in_len;
    if (a.n_batch) {
        in = a.ins[blockIdx.y];
        out = a.outs[blockIdx.y];
        in_len = a.in_lens[blockIdx.y];
    }
    lds_u32* sc = (lds_u32*)scratch[wv];
    // block index digits: b = k1 * 2^log_rest + rho ; this tile gathers 2^logc consecutive k1
    const uint32_t log_nb = a.log_n - S;
    const uint32_t log_rest = log_nb - a.s1;
    const uint32_t lg = a.s1 - a.logc;
    const uint64_t rho = (uint64_t)tile >> lg;
    const uint64_t g = (uint64_t)tile & ((1ull << lg) - 1);

    // coalesced load: element q = e*64 + lane -> (gathered block q >> S, row q & (L-1))
    F x[8];
    uint32_t wi[8], ri[8];
    const uint32_t c = lane & ((1u << LC) - 1u);
    const uint32_t v = lane >> LC;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t q = (uint32_t)e * 64u + lane;
        const uint32_t cb = q >> S, row = q & (L - 1);
        const uint64_t b = ((((uint64_t)g << a.logc) + cb) << log_rest) | rho;
        const uint64_t idx = (b << S) + row;
        if (cb < (1u << a.logc) && idx < in_len) {
            x[e] = ld_u<F>(in, idx);
            if (a.pre_mul) x[e] = F::mul_fenced(x[e], ld_u<F>(a.pre_mul, idx));
        } else {
            x[e] = F::zero();
        }
        wi[e] = (bitrev32(row, S) << LC) | cb;               // DIT position of this row
        ri[e] = (((v << 3) | (uint32_t)e) << LC) | c;        // window-0 layout
    }
    wave_exchange<F>(x, sc, wi, ri);
    dit_all<F, S, 0>(x, v, c, LC, a.tw_inner, sc);
    if (c >= (1u << a.logc)) return;
    const uint64_t k1o = ((uint64_t)g << a.logc) + c;
    // digit-reversed block index.  Block b = (k1, k2[, k3]) from the most significant digit down; the output index counts k1 as the
    // LEAST significant digit, then k2, then k3.  With three passes rho = k2 is one digit; with four, rho = (k2, k3) is swapped too.
    uint64_t rho_rev = rho;
    if (a.s2) {
        const uint32_t s3 = log_rest - a.s2;
        rho_rev = (rho >> s3) | ((rho & ((1ull << s3) - 1ull)) << a.s2);
    }
    const uint64_t obase = k1o + (rho_rev << a.s1);
    F sc_mul = F::split_words(a.scale);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t k = S >= 3 ? (((uint64_t)e << (S - 3)) | v) : e;
        const uint64_t oidx = obase + (k << log_nb);
        // every output passes one Montgomery product: it carries 1/N or g^-j/N where needed and
        // brings the lazily reduced value (< 47 r) back under 2r for the canonical store
        if (ntt_units_before(S) > 6) F::sweep(x[e]);
        F y = F::mul_fenced(x[e], sc_mul);
        if (a.post_mul) y = F::mul_fenced(y, ld_u<F>(a.post_mul, oidx));
        st_u<F>(out, oidx, y);
    }
}
