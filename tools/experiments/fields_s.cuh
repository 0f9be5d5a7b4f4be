// BLS12-381 base field on SIGNED radix-2^30 digits: 13 per element (390 bits), Montgomery radix 2^390 -- the arithmetic of the
// table-path accumulation kernel (msm.hip: msm_accumulate_s).  Everything else in the library stays on the unsigned 29-bit form of
// fieldu.cuh (14 limbs, radix 2^406); `fs_from_fu` / `fs_to_fu` cross between the two.
//
// Why: v_mad_i64_i32 runs at the rate of v_mad_u64_u32 (profiles/r02_clock_probe_signed.txt), and with |digit| <= 2^29 a signed 64-bit
// column holds 13 + 13 products of <= 2^58 -- so one limb fewer than the unsigned form, whose 58-bit products need 29-bit limbs:
// 169 + 169 multiply-adds per product instead of 196 + 196.  Sums and differences are plain digit-wise operations (negative values
// are representable: no "+ 8p" biases), followed by one carry sweep before the value enters a product.
//
// Invariants.  "Normalised" = digits 0..11 in [-2^29, 2^29), digit 12 holds what is left (|value| < 2^9 q fits).  Products take
// normalised operands with |value| <= 8q and return a normalised value with |value| <= q (A B / 2^390 + q / 2, A B <= 64 q^2 and
// q / 2^390 < 2^-9).  Column bound of `mul` / `sqr`: 13 * 2^58 + 2^29 * sum|P_i| = 2^61.7 + 2^60.63 < 2^63.  `dot2` (two products, one
// reduction) has 26 * 2^58 in a column, which together with the reduction term can pass 2^63 by 1.4 %: its carry is therefore taken
// in two steps (the operand products first, then the reduction term on the 30 low bits), see below.
// The constants are generated (tools/experiments/fs_consts.py) from the modulus.
#pragma once
#include "fieldu.cuh"

struct FsBls {
    static constexpr int NL = 13;
    int32_t v[13];
};

namespace fs {
constexpr int32_t P[13] = {-21845, -402915328, 356515836, -352321620, -252304353, 55215067, 288093811, 316751073, -321428361, 517541167, -375082566, -91332614, 1704210};                 // q, balanced digits
constexpr uint32_t PINV = 1073545213u;                  // -q^-1 mod 2^30
constexpr int32_t ONE[13] = {13762350, 433586176, -192935228, -301937177, 37952645, -425753694, -36732706, 162803105, -437337492, 366579475, 78814996, -442511456, 89578};           // 2^390 mod q (centred)
constexpr int32_t S2U[13] = {61493465, -304872632, 344236609, -420792340, -63920317, -102651361, -331210690, 479769549, 333524341, -265014713, -104291096, 296650812, -446358};           // 2^406 mod q (centred): Fs::mul by it changes the Montgomery radix 2^390 -> 2^406
constexpr uint32_t U2S[14] = {13762350, 330301440, 302000913, 268857142, 70371403, 334525505, 333461350, 437702779, 247622694, 320742703, 175919453, 511508630, 366909799, 0};   // 2^390 mod q in 29-bit limbs: Fu::mul by it changes 2^406 -> 2^390
}  // namespace fs

struct Fs : FsBls {
    ZK_HD static Fs zero() {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fs one() {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = fs::ONE[i];
        return r;
    }
    ZK_HD bool digits_zero() const {            // exact "== 0" for a normalised value with |value| < q (e.g. a product)
        int32_t o = 0;
#pragma unroll
        for (int i = 0; i < 13; ++i) o |= v[i];
        return o == 0;
    }
    // carry sweep to centred digits; the input digits may be sums of up to four normalised values (|digit| < 2^31)
    ZK_HD static void normalize(Fs& t) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int32_t c = ((t.v[i] >> 29) + 1) >> 1;      // round(v / 2^30) without leaving 32 bits
            t.v[i] -= c << 30;
            t.v[i + 1] += c;
        }
    }
    ZK_HD static Fs add(const Fs& a, const Fs& b) {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = a.v[i] + b.v[i];
        return r;
    }
    ZK_HD static Fs sub(const Fs& a, const Fs& b) {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = a.v[i] - b.v[i];
        return r;
    }
    ZK_HD static Fs neg(const Fs& a) {
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) r.v[i] = -a.v[i];
        return r;
    }
    // the 30 low bits as a digit in [-2^29, 2^29).  The empty asm keeps the digit a 32-bit value in the compiler's eyes: without it
    // the shift pair is widened into the 64-bit column arithmetic it came from, the digit travels as a (low word, sign word) pair and
    // every product with it becomes a full 64 x 64-bit multiply (2 mads + 5 fix-ups instead of one v_mad_i64_i32: seen for 26 of the
    // digits of one mixed addition, +20 % vector instructions).
    ZK_HD static int32_t centre30(uint32_t lo) {
        int32_t d = (int32_t)(lo << 2) >> 2;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(d));
#endif
        return d;
    }

    ZK_HD static int32_t top32(int64_t carry) {                     // the last carry is the top digit; same 32-bit fence as centre30
        int32_t d = (int32_t)carry;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(d));
#endif
        return d;
    }
    // a * b / 2^390 mod q
    ZK_HD static Fs mul(const Fs& a, const Fs& b) {
        int32_t m[13];
        Fs r;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            int64_t t = carry;            // ONE accumulator chain per column: every multiply-add adds into it (no 64-bit adds to join chains;
                                          // the kernel runs 3 wavefronts per SIMD, which covers the latency of the dependent chain)
#pragma unroll
            for (int i = 0; i <= k; ++i) t += (int64_t)a.v[i] * b.v[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) t += (int64_t)m[i] * fs::P[k - i];
            m[k] = centre30((uint32_t)t * fs::PINV);
            t += (int64_t)m[k] * fs::P[0];
            carry = t >> 30;                                  // exact: the 30 low bits are zero
        }
#pragma unroll
        for (int k = 13; k < 25; ++k) {
            int64_t t = carry;
#pragma unroll
            for (int i = k - 12; i < 13; ++i) {
                t += (int64_t)a.v[i] * b.v[k - i];
                t += (int64_t)m[i] * fs::P[k - i];
            }
            carry = (t + (1 << 29)) >> 30;
            r.v[k - 13] = centre30((uint32_t)t);
        }
        r.v[12] = top32(carry);
        return r;
    }
    ZK_HD static Fs sqr(const Fs& a) {
        int32_t m[13], a2[13];
        Fs r;
#pragma unroll
        for (int i = 0; i < 13; ++i) a2[i] = a.v[i] * 2;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            int64_t t = carry;
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) t += (int64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) t += (int64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) t += (int64_t)m[i] * fs::P[k - i];
            m[k] = centre30((uint32_t)t * fs::PINV);
            t += (int64_t)m[k] * fs::P[0];
            carry = t >> 30;
        }
#pragma unroll
        for (int k = 13; k < 25; ++k) {
            int64_t t = carry;
#pragma unroll
            for (int i = k - 12; 2 * i < k; ++i) t += (int64_t)a2[i] * a.v[k - i];
            if ((k & 1) == 0) t += (int64_t)a.v[k / 2] * a.v[k / 2];
#pragma unroll
            for (int i = k - 12; i < 13; ++i) t += (int64_t)m[i] * fs::P[k - i];
            carry = (t + (1 << 29)) >> 30;
            r.v[k - 13] = centre30((uint32_t)t);
        }
        r.v[12] = top32(carry);
        return r;
    }
    // (a*b - c*d) / 2^390 mod q with ONE reduction (a difference, so that no operand has to be negated first: the compiler turns
    // sext(-x) * y into a full 64 x 64-bit multiply -- 2 mads + 5 fix-up instructions per product instead of one v_mad_i64_i32).  26 operand products per column: their sum (<= 2^62.7) is split into its 30 low
    // bits and the rest BEFORE the reduction term (<= 2^60.63) is added, so no partial sum passes 2^63.
    ZK_HD static Fs dot2_sub(const Fs& a, const Fs& b, const Fs& c, const Fs& d) {
        int32_t m[13];
        Fs r;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            int64_t a0 = carry, a1 = 0;
#pragma unroll
            for (int i = 0; i <= k; ++i) {
                a0 += (int64_t)a.v[i] * b.v[k - i];
                a1 += (int64_t)c.v[i] * d.v[k - i];
            }
            const int64_t t = a0 - a1;
            int64_t u = (int64_t)((uint32_t)t & 0x3fffffffu);
#pragma unroll
            for (int i = 0; i < k; ++i) u += (int64_t)m[i] * fs::P[k - i];
            m[k] = centre30((uint32_t)u * fs::PINV);
            u += (int64_t)m[k] * fs::P[0];
            carry = (t >> 30) + (u >> 30);
        }
#pragma unroll
        for (int k = 13; k < 25; ++k) {
            int64_t a0 = carry, a1 = 0;
#pragma unroll
            for (int i = k - 12; i < 13; ++i) {
                a0 += (int64_t)a.v[i] * b.v[k - i];
                a1 += (int64_t)c.v[i] * d.v[k - i];
            }
            const int64_t t = a0 - a1;
            int64_t u = (int64_t)((uint32_t)t & 0x3fffffffu);
#pragma unroll
            for (int i = k - 12; i < 13; ++i) u += (int64_t)m[i] * fs::P[k - i];
            carry = (t >> 30) + ((u + (1 << 29)) >> 30);
            r.v[k - 13] = centre30((uint32_t)u);
        }
        r.v[12] = top32(carry);
        return r;
    }
};

// ---- crossing between the two forms (table build and bucket stores only: a product and a re-slicing each way)
// Fu element (any lazily reduced value < 64q, radix 2^406) -> Fs (normalised, 0 <= value < 2q, radix 2^390)
template <class FU>
ZK_HD Fs fs_from_fu(const FU& x) {
    static_assert(FU::NL == 14, "the 14 x 29-bit form of the BLS12-381 base field");
    FU c;
#pragma unroll
    for (int i = 0; i < 14; ++i) c.v[i] = fs::U2S[i];
    const FU y = FU::mul(x, c);                       // integer v * 2^390 mod q, < 2q, 29-bit limbs
    Fs r;
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {                   // bits [30k, 30k + 30) of the integer
        const int bit = 30 * k, i = bit / 29, off = bit % 29;
        uint64_t w = (uint64_t)y.v[i] >> off;
        if (i + 1 < 14) w |= (uint64_t)y.v[i + 1] << (29 - off);
        if (i + 2 < 14) w |= (uint64_t)y.v[i + 2] << (58 - off);
        int32_t dgt = (int32_t)((uint32_t)w & 0x3fffffffu) + carry;
        carry = 0;
        if (k < 12 && dgt >= (1 << 29)) {
            dgt -= 1 << 30;
            carry = 1;
        }
        r.v[k] = dgt;
    }
    return r;
}
// Fs (normalised, |value| < q: a product) -> Fu (0 < value < 2q, radix 2^406, normalised 29-bit limbs)
template <class FU>
ZK_HD FU fs_to_fu(const Fs& x) {
    static_assert(FU::NL == 14, "the 14 x 29-bit form of the BLS12-381 base field");
    Fs c;
#pragma unroll
    for (int i = 0; i < 13; ++i) c.v[i] = fs::S2U[i];
    Fs y = Fs::mul(x, c);                             // v * 2^406 mod q, |y| <= q/2 + q/512
    int32_t dgt[13];
    int32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {                   // + q, then non-negative 30-bit digits
        const int32_t t = y.v[k] + fs::P[k] + carry;
        if (k < 12) {
            carry = t >> 30;
            dgt[k] = t & 0x3fffffff;
        } else {
            dgt[k] = t;
        }
    }
    FU r;
#pragma unroll
    for (int j = 0; j < 14; ++j) {                   // bits [29j, 29j + 29)
        const int bit = 29 * j, i = bit / 30, off = bit % 30;
        uint64_t w = (uint64_t)(uint32_t)dgt[i] >> off;
        if (i + 1 < 13) w |= (uint64_t)(uint32_t)dgt[i + 1] << (30 - off);
        r.v[j] = (uint32_t)w & ((1u << 29) - 1u);
    }
    return r;
}
