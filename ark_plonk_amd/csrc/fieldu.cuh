// Unsaturated (29-bit limb) Montgomery arithmetic for the scalar field on gfx950: the NTT butterflies, the KZG witness /
// evaluation kernels, the quotient and grand-product kernels.  (The MSM's base field uses signed 30-bit limbs: fields.cuh.)
//
// Why: measured on MI355X (profiles/r01/r01_ubench_valu.txt) v_mad_u64_u32 issues in 4 cycles per
// wave-instruction -- the same as one v_add_co/v_addc -- so in a saturated 32-bit-limb CIOS the carry
// handling costs more than the multiplies.  With 29-bit limbs a 64-bit accumulator absorbs a whole
// product-scanning column (2*NL products < 2^58 each) with no carries: one v_mad_u64_u32 per limb
// product, one v_and + one v_lshrrev_b64 per column.  Additions are limb-wise with one carry sweep
// and no conditional subtraction: values stay lazily reduced (bounded multiples of p) and the
// Montgomery product brings them back below 2p.
//
// Representation: value = sum v[i] * 2^(29 i), NL limbs, Montgomery radix R' = 2^(29 NL).
// "Normalised" = v[i] < 2^29 for i < NL-1.  Every routine here returns normalised limbs.
// sub8(a,b) = a - b + 8p needs b < 8p; sub16 needs b < 16p; the callers state their bounds.
//
// Stands behind ark-ff 0.3 Fp256 arithmetic in the NTT butterflies (prover.rs:196-203) and the polynomial
// kernels; arkworks-format values cross in/out through from_sat/to_sat.
#pragma once
#include "zk_common.h"

template <class P>
struct Fu {
    static constexpr int NL = P::NL;
    static constexpr int SAT = P::SAT_WORDS;
    static constexpr uint32_t M = (1u << 29) - 1u;
    uint32_t v[NL];

    ZK_HD static Fu zero() {
        Fu r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fu one() {
        Fu r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.v[i] = P::ONE(i);
        return r;
    }
    ZK_HD bool limbs_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) o |= v[i];
        return o == 0;
    }

    // carry sweep; limbs are treated as signed so a - b + k*p may pass through negative limbs
    ZK_HD static void normalize(Fu& t) {
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) {
            uint32_t c = (uint32_t)((int32_t)t.v[i] >> 29);
            t.v[i] &= M;
            t.v[i + 1] += c;
        }
    }

    ZK_HD static Fu add(const Fu& a, const Fu& b) {
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + b.v[i];
        normalize(t);
        return t;
    }
    ZK_HD static Fu dbl(const Fu& a) {
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] << 1;
        normalize(t);
        return t;
    }
    // a + b + c without intermediate sweeps
    ZK_HD static Fu add3(const Fu& a, const Fu& b, const Fu& c) {
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + b.v[i] + c.v[i];
        normalize(t);
        return t;
    }
    ZK_HD static Fu sub2(const Fu& a, const Fu& b) {   // a - b + 2p, b < 2p (b a product)
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] - b.v[i] + P::Z2(i);
        normalize(t);
        return t;
    }
    ZK_HD static Fu sub8(const Fu& a, const Fu& b) {   // a - b + 8p, b < 8p
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] - b.v[i] + P::Z8(i);
        normalize(t);
        return t;
    }
    ZK_HD static Fu sub16(const Fu& a, const Fu& b) {  // a - b + 16p, b < 16p
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] - b.v[i] + P::Z16(i);
        normalize(t);
        return t;
    }
    ZK_HD static Fu neg16(const Fu& a) {                 // 16p - a, a < 16p
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = P::Z16(i) - a.v[i];
        normalize(t);
        return t;
    }
    ZK_HD static Fu neg_canonical(const Fu& a) {        // p - a for a <= p (affine input coordinates)
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = P::ZP(i) - a.v[i];
        normalize(t);
        return t;
    }

    // ---- lazily normalised arithmetic (the NTT butterflies, ntt_pass.cuh) -------------------------------------------------
    // Limbs are plain NON-NEGATIVE 32-bit numbers that are not swept after every operation; "k units" = every limb below the top
    // one is < k * 2^29 (a swept value has 1 unit).  A Montgomery product takes a first operand of up to 6 units against a swept
    // second one: a column is at most 9 * (6 * 2^29) * 2^29 + 9 * 2^58 + carry < 2^64.  A limb holds up to 8 units.
    // sweep: the carry pass for non-negative limbs (logical shifts; `normalize` above reads limbs as signed).
    ZK_HD static void sweep(Fu& t) {
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) {
            const uint32_t c = t.v[i] >> 29;
            t.v[i] &= M;
            t.v[i + 1] += c;
        }
    }
    ZK_HD static Fu add_raw(const Fu& a, const Fu& b) {      // units(a) + units(b)
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + b.v[i];
        return t;
    }
    // a - b + k p with no limb ever below zero: b SWEPT (1 unit) and below 2p / 8p / 16p; the multiple of p has every limb under the top
    // one in [2^29, 2^30) and a top limb that covers b's.  units(a) + 2.
    ZK_HD static Fu sub_raw3(const Fu& a, const Fu& b) {     // b < 2p (a product)
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + P::ZB3(i) - b.v[i];
        return t;
    }
    ZK_HD static Fu sub_raw8(const Fu& a, const Fu& b) {     // b < 7p
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + P::ZB8(i) - b.v[i];
        return t;
    }
    ZK_HD static Fu sub_raw16(const Fu& a, const Fu& b) {    // b < 15p
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + P::ZB16(i) - b.v[i];
        return t;
    }

    // Montgomery product a*b/R' : interleaved product scanning.  Each column keeps the a*b products and
    // the m*p products in SEPARATE 64-bit accumulators: a dependent v_mad_u64_u32 chain issues only every
    // ~14 cycles per wave (measured: 7.0 cycles/mad/SIMD at 2 waves), two or three independent chains
    // reach the single-wave issue limit of ~9.5.  Column total < 2*NL*2^58 + carry < 2^63.
    ZK_HD static Fu mul(const Fu& a, const Fu& b) {
        uint32_t m[NL];
        Fu r;
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            uint64_t a0 = 0, a1 = 0, am = 0;
#pragma unroll
            for (int i = 0; i <= k; ++i) {
                if (i & 1) a1 += (uint64_t)a.v[i] * b.v[k - i];
                else a0 += (uint64_t)a.v[i] * b.v[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; ++i) am += (uint64_t)m[i] * P::MOD(k - i);
            uint64_t t = a0 + a1 + am + carry;
            m[k] = ((uint32_t)t * P::PINV) & M;
            t += (uint64_t)m[k] * P::MOD(0);
            carry = t >> 29;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) {
            uint64_t a0 = 0, a1 = 0, am = 0;
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) {
                if (i & 1) a1 += (uint64_t)a.v[i] * b.v[k - i];
                else a0 += (uint64_t)a.v[i] * b.v[k - i];
                am += (uint64_t)m[i] * P::MOD(k - i);
            }
            uint64_t t = a0 + a1 + am + carry;
            r.v[k - NL] = (uint32_t)t & M;
            carry = t >> 29;
        }
        r.v[NL - 1] = (uint32_t)carry;
        return r;
    }
    // `mul` with every column's chain STARTED from the carry of the column below: the first partial sum (carry + first product) is
    // passed through an empty, non-volatile asm, which the optimiser cannot look through -- so it cannot compute the columns as
    // independent chains and add the carries afterwards (one v_lshl_add_u64 and most of a v_mov per column in `mul`'s code).
    // Costs nothing itself; the chains of the several products a caller has in flight still interleave.
    ZK_HD static uint64_t fence64(uint64_t t) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(t));
#endif
        return t;
    }
    ZK_HD static Fu mul_fenced(const Fu& a, const Fu& b) {
        uint32_t m[NL];
        Fu r;
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            uint64_t t = fence64(carry + (uint64_t)a.v[0] * b.v[k]);
#pragma unroll
            for (int i = 1; i <= k; ++i) t += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) t += (uint64_t)m[i] * P::MOD(k - i);
            m[k] = ((uint32_t)t * P::PINV) & M;
            t += (uint64_t)m[k] * P::MOD(0);
            carry = t >> 29;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) {
            uint64_t t = fence64(carry + (uint64_t)a.v[k - NL + 1] * b.v[NL - 1]);
#pragma unroll
            for (int i = k - NL + 2; i < NL; ++i) t += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) t += (uint64_t)m[i] * P::MOD(k - i);
            r.v[k - NL] = (uint32_t)t & M;
            carry = t >> 29;
        }
        r.v[NL - 1] = (uint32_t)carry;
        return r;
    }
    // a*a/R' : cross products once, against pre-doubled limbs; same accumulator split
    ZK_HD static Fu sqr(const Fu& a) {
        uint32_t m[NL], a2[NL];
        Fu r;
#pragma unroll
        for (int i = 0; i < NL; ++i) a2[i] = a.v[i] << 1;
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            uint64_t aa = 0, am = 0;
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) aa += (uint64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) aa += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) am += (uint64_t)m[i] * P::MOD(k - i);
            uint64_t t = aa + am + carry;
            m[k] = ((uint32_t)t * P::PINV) & M;
            t += (uint64_t)m[k] * P::MOD(0);
            carry = t >> 29;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) {
            uint64_t aa = 0, am = 0;
#pragma unroll
            for (int i = k - NL + 1; 2 * i < k; ++i) aa += (uint64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) aa += (uint64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) am += (uint64_t)m[i] * P::MOD(k - i);
            uint64_t t = aa + am + carry;
            r.v[k - NL] = (uint32_t)t & M;
            carry = t >> 29;
        }
        r.v[NL - 1] = (uint32_t)carry;
        return r;
    }

    // exact "== 0 mod p" for a mul/sqr output (value < 2p, normalised): value is 0 or p
    ZK_HD bool is_zero_mod_reduced() const {
        if (v[0] != 0 && v[0] != P::MOD(0)) return false;
        uint32_t dz = 0, dp = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            dz |= v[i];
            dp |= v[i] ^ P::MOD(i);
        }
        return dz == 0 || dp == 0;
    }
    // exact zero test for any lazily reduced value
    ZK_HD bool is_zero_mod() const { return mul(*this, one()).is_zero_mod_reduced(); }

    // fully reduce a value < 2p into [0, p)
    ZK_HD static Fu canonical_lt2p(const Fu& a) {
        Fu d;
#pragma unroll
        for (int i = 0; i < NL; ++i) d.v[i] = a.v[i] - P::MOD(i);
        normalize(d);
        const bool neg = ((int32_t)d.v[NL - 1]) < 0;
        Fu r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.v[i] = neg ? a.v[i] : d.v[i];
        return r;
    }

    // ---- arkworks layout <-> this representation -------------------------------------------
    // w: SAT little-endian 32-bit words of x*R mod p (R = 2^(32 SAT)), canonical.  Returns x*R'.
    ZK_HD static Fu from_sat(const uint32_t* w) {
        Fu t = split_words(w);
        Fu c;
#pragma unroll
        for (int i = 0; i < NL; ++i) c.v[i] = P::C_IN(i);
        return mul(t, c);
    }
    // inverse of from_sat: writes the canonical x*R mod p words
    ZK_HD void to_sat(uint32_t* w) const {
        Fu c;
#pragma unroll
        for (int i = 0; i < NL; ++i) c.v[i] = P::C_OUT(i);
        Fu t = canonical_lt2p(mul(*this, c));
        t.pack_words(w);
    }
    // plain integer (no Montgomery factor change): 32-bit words -> 29-bit limbs
    ZK_HD static Fu split_words(const uint32_t* w) {
        Fu t;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int bit = 29 * i;
            const int wi = bit >> 5, off = bit & 31;
            uint64_t lo = wi < SAT ? w[wi] : 0u;
            uint64_t hi = (wi + 1) < SAT ? w[wi + 1] : 0u;
            t.v[i] = (uint32_t)(((hi << 32) | lo) >> off) & M;
        }
        return t;
    }
    ZK_HD void pack_words(uint32_t* w) const {
#pragma unroll
        for (int j = 0; j < SAT; ++j) {
            // word j holds bits [32j, 32j+32): pieces of up to three limbs
            const int lo_bit = 32 * j;
            uint64_t acc = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int lb = 29 * i;
                if (lb + 29 <= lo_bit || lb >= lo_bit + 32) continue;
                if (lb >= lo_bit) acc |= (uint64_t)v[i] << (lb - lo_bit);
                else acc |= (uint64_t)v[i] >> (lo_bit - lb);
            }
            w[j] = (uint32_t)acc;
        }
    }

    // a^e, e little-endian 32-bit words (square-and-multiply)
    ZK_HD static Fu pow_words(const Fu& a, const uint32_t* e, int n) {
        Fu r = one();
        bool started = false;
        for (int i = n - 1; i >= 0; --i)
            for (int b = 31; b >= 0; --b) {
                if (started) r = sqr(r);
                if ((e[i] >> b) & 1u) {
                    r = mul(r, a);
                    started = true;
                }
            }
        return r;
    }
    // Fermat inverse (inputs < 64p); inverse(0) = 0
    ZK_HD static Fu inverse(const Fu& a) {
        // exponent p - 2 as 32-bit words
        Fu pm;
#pragma unroll
        for (int i = 0; i < NL; ++i) pm.v[i] = P::MOD(i);
        uint32_t e[SAT];
        pm.pack_words(e);
        // p is odd and > 2: subtracting 2 only touches the low word unless it underflows
        uint64_t t = (uint64_t)e[0] - 2u;
        e[0] = (uint32_t)t;
        uint32_t borrow = (uint32_t)(t >> 32) & 1u;
        for (int i = 1; i < SAT && borrow; ++i) {
            t = (uint64_t)e[i] - borrow;
            e[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        return pow_words(a, e, SAT);
    }
};
