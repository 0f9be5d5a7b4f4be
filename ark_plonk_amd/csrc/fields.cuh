// Signed 30-bit-limb Montgomery arithmetic for the MSM base field (the XYZZ group law of ecu.cuh / ecq.cuh).
//
// Why a second unsaturated form next to fieldu.cuh: msm_accumulate is bound by the issue rate of the 64-bit
// multiply-add, so the only way to make a mixed addition cheaper is fewer limb products.  With SIGNED limbs in
// [-2^29, 2^29) a product is below 2^58 in magnitude, a product-scanning column of 2*NL products (3*NL in dot2)
// still fits a signed 64-bit accumulator (checked per modulus by tools/gen_constants.py), and a limb carries 30
// bits instead of 29: BLS12-381 Fq takes 13 limbs instead of 14 (338 v_mad_i64_i32 per product instead of 392),
// BN254 Fq 9 instead of 10 (162 instead of 200).  Signed limbs also make subtraction limb-wise with no multiple
// of p added, and the Montgomery quotient digits balanced, so a product of operands below 12p in magnitude
// comes out in (-p, p): |a||b|/R' + p/2 with R' >= 512 p.
//
// Representation: value = sum v[i] * 2^(30 i), v[i] two's-complement in a uint32_t, Montgomery radix R' = 2^(30 NL).
//   "strict"  : v[i] in [-2^29, 2^29) for i < NL-1 (the top limb takes the rest) -- unique for a given integer;
//               what mul / sqr / dot2 / from_sat / canonical_lt2p return;
//   "almost"  : |v[i]| <= 2^29 + 2 -- what add / sub / dbl / add3 return (one carry step, no ripple).
// Inputs of mul / sqr / dot2: almost-balanced limbs, |value| < 12p.
// The interface (names included) is the one of Fu<>, so the group law templates take either.
//
// Stands behind ark-ff 0.3 Fp384 / Fp256 arithmetic inside VariableBaseMSM (commitment.rs:45); arkworks-format
// values cross in / out through from_sat / to_sat.
#pragma once
#include "zk_common.h"

template <class P>
struct Fs {
    static constexpr int NL = P::NL;
    static constexpr int SAT = P::SAT_WORDS;
    static constexpr uint32_t M = (1u << 30) - 1u;
    static constexpr int32_t H = 1 << 29;
    uint32_t v[NL];

    ZK_HD static int32_t sx(uint32_t x) { return (int32_t)(x << 2) >> 2; }   // low 30 bits, sign-extended
    // The bias 2^29 that starts every high-half column.  On the device it is hidden from the optimiser in a scalar register
    // pair (an empty asm): as a visible constant it is moved to the end of the column's sum and costs a 64-bit add per
    // column; as a variable it can be the addend of the column's first multiply-add, whose addend slot is otherwise 0
    // (the compiler takes that form in about half of the columns; fencing the first partial sum to force it everywhere
    // cost more registers than it saved instructions).
    ZK_HD static int64_t column_bias() {
        int64_t h = H;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(h));
#endif
        return h;
    }

    ZK_HD static Fs zero() {
        Fs r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fs one() {
        Fs r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.v[i] = (uint32_t)P::ONE(i);
        return r;
    }
    ZK_HD bool limbs_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) o |= v[i];
        return o == 0;
    }

    // one carry step, every limb at once: a limb keeps its balanced low 30 bits and takes the carry of the limb
    // below (computed from that limb's value BEFORE the step).  |in| < 2^31 - 2^29 (sums and differences of TWO operands)
    // ->  |out| <= 2^29 + 2.
    ZK_HD static void normalize(Fs& t) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) {
            const int32_t x = (int32_t)t.v[i];
            const int32_t r = sx((uint32_t)x);
            t.v[i] = (uint32_t)(r + c);
            c = (x - r) >> 30;
        }
        t.v[NL - 1] += (uint32_t)c;
    }
    // the same step for limbs anywhere in int32 (sums of three or four operands): x - sx(x) can leave 32 bits there, so the
    // carry is taken as floor(x / 2^30) + bit 29 of x -- one instruction more per limb
    ZK_HD static void normalize_wide(Fs& t) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) {
            const int32_t x = (int32_t)t.v[i];
            t.v[i] = (uint32_t)(sx((uint32_t)x) + c);
            c = (x >> 30) + (int32_t)(((uint32_t)x >> 29) & 1u);
        }
        t.v[NL - 1] += (uint32_t)c;
    }
    // full ripple: strict limbs (unique representation)
    ZK_HD static void normalize_strict(Fs& t) {
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) {
            const int32_t x = (int32_t)t.v[i];
            const int32_t r = sx((uint32_t)x);
            t.v[i] = (uint32_t)r;
            t.v[i + 1] += (uint32_t)((x - r) >> 30);
        }
    }

    ZK_HD static Fs add(const Fs& a, const Fs& b) {
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + b.v[i];
        normalize(t);
        return t;
    }
    ZK_HD static Fs dbl(const Fs& a) {
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] << 1;
        normalize(t);
        return t;
    }
    ZK_HD static Fs add3(const Fs& a, const Fs& b, const Fs& c) {
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] + b.v[i] + c.v[i];
        normalize_wide(t);
        return t;
    }
    ZK_HD static Fs sub(const Fs& a, const Fs& b) {
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] - b.v[i];
        normalize(t);
        return t;
    }
    // a - (b + c + d) with one carry step.  Strict operands: a limb of the raw result is in [-2^31 + 3, 2^31 - 1].
    ZK_HD static Fs sub_sum3(const Fs& a, const Fs& b, const Fs& c, const Fs& d) {
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = a.v[i] - b.v[i] - c.v[i] - d.v[i];
        normalize_wide(t);
        return t;
    }
    // the names the group law uses with Fu (there they say which multiple of p keeps the difference positive)
    ZK_HD static Fs sub2(const Fs& a, const Fs& b) { return sub(a, b); }
    ZK_HD static Fs sub8(const Fs& a, const Fs& b) { return sub(a, b); }
    ZK_HD static Fs sub16(const Fs& a, const Fs& b) { return sub(a, b); }
    ZK_HD static Fs neg16(const Fs& a) {                  // limb-wise: stays almost balanced
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) t.v[i] = 0u - a.v[i];
        return t;
    }
    ZK_HD static Fs neg_canonical(const Fs& a) { return neg16(a); }

    // Montgomery product a*b/R': interleaved product scanning, one signed 64-bit accumulator per column (the
    // compiler merges the three partial sums below into one v_mad_i64_i32 chain per column, two columns interleaved).
    // Low half: the balanced quotient digit m_k makes the column a multiple of 2^30 (exact shift).
    // High half: the column starts from the bias 2^29, so the floor shift is the carry of the BALANCED digit
    // (t + 2^29) mod 2^30 - 2^29.
    ZK_HD static Fs mul(const Fs& a, const Fs& b) {
        int32_t m[NL];
        Fs r;
        int64_t carry = 0;
        const int64_t hb = column_bias();
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            int64_t a0 = 0, a1 = 0, am = 0;
#pragma unroll
            for (int i = 0; i <= k; ++i) {
                if (i & 1) a1 += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[k - i];
                else a0 += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; ++i) am += (int64_t)m[i] * P::MOD(k - i);
            int64_t t = a0 + a1 + am + carry;
            m[k] = (int32_t)((uint32_t)t * (P::PINV << 2)) >> 2;      // (t * PINV mod 2^30), sign-extended: the shift rides in the constant
            t += (int64_t)m[k] * P::MOD(0);
            carry = t >> 30;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) {
            int64_t a0 = hb, a1 = 0, am = 0;
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) {
                if (i & 1) a1 += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[k - i];
                else a0 += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[k - i];
                am += (int64_t)m[i] * P::MOD(k - i);
            }
            const int64_t t = a0 + a1 + am + carry;
            r.v[k - NL] = ((uint32_t)t & M) - (uint32_t)H;
            carry = t >> 30;
        }
        r.v[NL - 1] = (uint32_t)carry;
        return r;
    }
    // (a*b + c*d)/R' with ONE Montgomery reduction.  |a||b| + |c||d| < 288 p^2 keeps the result in (-p, p).
    ZK_HD static Fs dot2(const Fs& a, const Fs& b, const Fs& c, const Fs& d) {
        int32_t m[NL];
        Fs r;
        int64_t carry = 0;
        const int64_t hb = column_bias();
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            int64_t a0 = 0, a1 = 0, am = 0;
#pragma unroll
            for (int i = 0; i <= k; ++i) {
                a0 += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[k - i];
                a1 += (int64_t)(int32_t)c.v[i] * (int32_t)d.v[k - i];
            }
#pragma unroll
            for (int i = 0; i < k; ++i) am += (int64_t)m[i] * P::MOD(k - i);
            int64_t t = a0 + a1 + am + carry;
            m[k] = (int32_t)((uint32_t)t * (P::PINV << 2)) >> 2;      // (t * PINV mod 2^30), sign-extended: the shift rides in the constant
            t += (int64_t)m[k] * P::MOD(0);
            carry = t >> 30;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) {
            int64_t a0 = hb, a1 = 0, am = 0;
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) {
                a0 += (int64_t)(int32_t)a.v[i] * (int32_t)b.v[k - i];
                a1 += (int64_t)(int32_t)c.v[i] * (int32_t)d.v[k - i];
                am += (int64_t)m[i] * P::MOD(k - i);
            }
            const int64_t t = a0 + a1 + am + carry;
            r.v[k - NL] = ((uint32_t)t & M) - (uint32_t)H;
            carry = t >> 30;
        }
        r.v[NL - 1] = (uint32_t)carry;
        return r;
    }
    // a*a/R': cross products once, against doubled limbs (|2 a_i| <= 2^30 + 4: the column bound is the one of mul)
    ZK_HD static Fs sqr(const Fs& a) {
        int32_t m[NL], a2[NL];
        Fs r;
#pragma unroll
        for (int i = 0; i < NL; ++i) a2[i] = (int32_t)(a.v[i] << 1);
        int64_t carry = 0;
        const int64_t hb = column_bias();
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            int64_t aa = 0, am = 0;
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) aa += (int64_t)a2[i] * (int32_t)a.v[k - i];
            if ((k & 1) == 0) aa += (int64_t)(int32_t)a.v[k / 2] * (int32_t)a.v[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) am += (int64_t)m[i] * P::MOD(k - i);
            int64_t t = aa + am + carry;
            m[k] = (int32_t)((uint32_t)t * (P::PINV << 2)) >> 2;      // (t * PINV mod 2^30), sign-extended: the shift rides in the constant
            t += (int64_t)m[k] * P::MOD(0);
            carry = t >> 30;
        }
#pragma unroll
        for (int k = NL; k < 2 * NL - 1; ++k) {
            int64_t aa = hb, am = 0;
#pragma unroll
            for (int i = k - NL + 1; 2 * i < k; ++i) aa += (int64_t)a2[i] * (int32_t)a.v[k - i];
            if ((k & 1) == 0) aa += (int64_t)(int32_t)a.v[k / 2] * (int32_t)a.v[k / 2];
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) am += (int64_t)m[i] * P::MOD(k - i);
            const int64_t t = aa + am + carry;
            r.v[k - NL] = ((uint32_t)t & M) - (uint32_t)H;
            carry = t >> 30;
        }
        r.v[NL - 1] = (uint32_t)carry;
        return r;
    }

    // exact "== 0 mod p" for a mul / sqr / dot2 output: strict limbs, value in (-2p, 2p), so it is 0, p or -p
    ZK_HD bool is_zero_mod_reduced() const {
        if (v[0] != 0 && v[0] != (uint32_t)P::MOD(0) && v[0] != (uint32_t)P::NMOD(0)) return false;
        uint32_t dz = 0, dp = 0, dn = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            dz |= v[i];
            dp |= v[i] ^ (uint32_t)P::MOD(i);
            dn |= v[i] ^ (uint32_t)P::NMOD(i);
        }
        return dz == 0 || dp == 0 || dn == 0;
    }
    // exact zero test for any lazily reduced value
    ZK_HD bool is_zero_mod() const { return mul(*this, one()).is_zero_mod_reduced(); }

    // limbs in [0, 2^30) below the top one (floor carries, full ripple): the sign of the value is the sign of the top limb
    ZK_HD static void floor_sweep(Fs& t) {
#pragma unroll
        for (int i = 0; i < NL - 1; ++i) {
            const int32_t x = (int32_t)t.v[i];
            t.v[i] = (uint32_t)x & M;
            t.v[i + 1] += (uint32_t)(x >> 30);
        }
    }
    // the representative in [0, p) of a value in (-2p, 2p), floor limbs
    ZK_HD static Fs canon_floor(const Fs& a) {
        Fs x = a;
        floor_sweep(x);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            Fs y;
#pragma unroll
            for (int i = 0; i < NL; ++i) y.v[i] = x.v[i] + (uint32_t)P::MOD(i);
            floor_sweep(y);
            const bool neg = (int32_t)x.v[NL - 1] < 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) x.v[i] = neg ? y.v[i] : x.v[i];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            Fs y;
#pragma unroll
            for (int i = 0; i < NL; ++i) y.v[i] = x.v[i] - (uint32_t)P::MOD(i);
            floor_sweep(y);
            const bool ge = (int32_t)y.v[NL - 1] >= 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) x.v[i] = ge ? y.v[i] : x.v[i];
        }
        return x;
    }
    // fully reduced, strict limbs (the stored affine coordinates)
    ZK_HD static Fs canonical_lt2p(const Fs& a) {
        Fs x = canon_floor(a);
        normalize_strict(x);
        return x;
    }

    // ---- arkworks layout <-> this representation -------------------------------------------
    // w: SAT little-endian 32-bit words of x*R mod p (R = 2^(32 SAT)), canonical.  Returns x*R' (strict, in (-p, p)).
    ZK_HD static Fs from_sat(const uint32_t* w) {
        Fs t = split_words(w);
        Fs c;
#pragma unroll
        for (int i = 0; i < NL; ++i) c.v[i] = (uint32_t)P::C_IN(i);
        return mul(t, c);
    }
    // inverse of from_sat: writes the canonical x*R mod p words
    ZK_HD void to_sat(uint32_t* w) const {
        Fs c;
#pragma unroll
        for (int i = 0; i < NL; ++i) c.v[i] = (uint32_t)P::C_OUT(i);
        canon_floor(mul(*this, c)).pack_words(w);
    }
    // plain non-negative integer (no Montgomery factor change): 32-bit words -> strict limbs
    ZK_HD static Fs split_words(const uint32_t* w) {
        Fs t;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int bit = 30 * i;
            const int wi = bit >> 5, off = bit & 31;
            uint64_t lo = wi < SAT ? w[wi] : 0u;
            uint64_t hi = (wi + 1) < SAT ? w[wi + 1] : 0u;
            t.v[i] = (uint32_t)(((hi << 32) | lo) >> off) & M;
        }
        normalize_strict(t);
        return t;
    }
    // floor limbs of a non-negative value below 2^(32 SAT) -> words
    ZK_HD void pack_words(uint32_t* w) const {
#pragma unroll
        for (int j = 0; j < SAT; ++j) {
            const int lo_bit = 32 * j;
            uint64_t acc = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int lb = 30 * i;
                if (lb + 30 <= lo_bit || lb >= lo_bit + 32) continue;
                if (lb >= lo_bit) acc |= (uint64_t)v[i] << (lb - lo_bit);
                else acc |= (uint64_t)v[i] >> (lo_bit - lb);
            }
            w[j] = (uint32_t)acc;
        }
    }

    // a^e, e little-endian 32-bit words (square-and-multiply)
    ZK_HD static Fs pow_words(const Fs& a, const uint32_t* e, int n) {
        Fs r = one();
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
    // Fermat inverse; inverse(0) = 0
    ZK_HD static Fs inverse(const Fs& a) {
        uint32_t e[SAT];
#pragma unroll
        for (int i = 0; i < SAT; ++i) e[i] = P::MODW(i);
        uint64_t t = (uint64_t)e[0] - 2u;      // p - 2: p is odd and > 2
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
