// Montgomery prime-field arithmetic on 32-bit limbs, shared by gfx950 device code and the host
// side of the C ABI (final MSM window combine, domain constants).
//
// Replaces (behind the C ABI) ark-ff 0.3 Fp256/Fp384 as used by the reference's hot path
// (plonk-core/src/proof_system/prover.rs:196-203 fft inputs, commitment.rs:33-48 scalars/points).
// Host layout is arkworks': little-endian 64-bit limbs, Montgomery form with R = 2^(64*limbs);
// a little-endian u64 limb array is byte-identical to the u32 limb array used here.
//
// Device mapping: one field element per lane, limbs in VGPRs; 32x32->64 multiply-accumulate
// lowers to v_mad_u64_u32 on gfx950 (no MFMA: there is no dense contraction on this path).
#pragma once
#include "zk_common.h"

template <class P>
struct Fp {
    static constexpr int N = P::N;
    uint32_t v[N];

    ZK_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fp one() {  // Montgomery form of 1
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::R(i);
        return r;
    }
    ZK_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::R2(i);
        return r;
    }
    ZK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= v[i];
        return o == 0;
    }
    ZK_HD bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // r = a - p if a >= p else a   (a < 2p)
    ZK_HD static Fp reduce_once(const Fp& a) {
        Fp d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a.v[i] - P::MOD(i) - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = borrow ? a.v[i] : d.v[i];
        return r;
    }

    ZK_HD static Fp add(const Fp& a, const Fp& b) {
        // 2p < 2^(32N) for every supported modulus (static_assert in curve_params generator)
        Fp s;
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a.v[i] + b.v[i] + carry;
            s.v[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        return reduce_once(s);
    }
    ZK_HD static Fp sub(const Fp& a, const Fp& b) {
        Fp d;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        uint32_t mask = 0u - borrow;
        uint32_t carry = 0;
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)d.v[i] + (P::MOD(i) & mask) + carry;
            r.v[i] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        return r;
    }
    ZK_HD static Fp neg(const Fp& a) {
        // p - a, mapped to 0 when a == 0
        Fp d;
        uint32_t borrow = 0;
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)P::MOD(i) - a.v[i] - borrow;
            d.v[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
            nz |= a.v[i];
        }
        uint32_t mask = nz ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) d.v[i] &= mask;
        return d;
    }
    ZK_HD static Fp dbl(const Fp& a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod p, CIOS over 32-bit limbs.
    ZK_HD static Fp mul(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
        // host pass: same CIOS on 64-bit limbs (the window combine after an MSM runs here)
        {
            constexpr int M = N / 2;
            uint64_t x[M], y[M], p[M], t64[M + 2];
            for (int i = 0; i < M; ++i) {
                x[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
                y[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
                p[i] = (uint64_t)P::MOD(2 * i) | ((uint64_t)P::MOD(2 * i + 1) << 32);
            }
            for (int i = 0; i < M + 2; ++i) t64[i] = 0;
            for (int i = 0; i < M; ++i) {
                unsigned __int128 s;
                uint64_t c = 0;
                for (int j = 0; j < M; ++j) {
                    s = (unsigned __int128)x[j] * y[i] + t64[j] + c;
                    t64[j] = (uint64_t)s;
                    c = (uint64_t)(s >> 64);
                }
                s = (unsigned __int128)t64[M] + c;
                t64[M] = (uint64_t)s;
                t64[M + 1] = (uint64_t)(s >> 64);
                uint64_t m = t64[0] * P::INV64;
                s = (unsigned __int128)m * p[0] + t64[0];
                c = (uint64_t)(s >> 64);
                for (int j = 1; j < M; ++j) {
                    s = (unsigned __int128)m * p[j] + t64[j] + c;
                    t64[j - 1] = (uint64_t)s;
                    c = (uint64_t)(s >> 64);
                }
                s = (unsigned __int128)t64[M] + c;
                t64[M - 1] = (uint64_t)s;
                t64[M] = t64[M + 1] + (uint64_t)(s >> 64);
            }
            Fp r;
            for (int i = 0; i < M; ++i) {
                r.v[2 * i] = (uint32_t)t64[i];
                r.v[2 * i + 1] = (uint32_t)(t64[i] >> 32);
            }
            return reduce_once(r);
        }
#endif
        uint32_t t[N + 2];
#pragma unroll
        for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
                t[j] = (uint32_t)s;
                c = s >> 32;
            }
            uint64_t s = (uint64_t)t[N] + c;
            t[N] = (uint32_t)s;
            t[N + 1] = (uint32_t)(s >> 32);
            uint32_t m = t[0] * P::INV32;
            s = (uint64_t)m * P::MOD(0) + t[0];
            c = s >> 32;
#pragma unroll
            for (int j = 1; j < N; ++j) {
                s = (uint64_t)m * P::MOD(j) + t[j] + c;
                t[j - 1] = (uint32_t)s;
                c = s >> 32;
            }
            s = (uint64_t)t[N] + c;
            t[N - 1] = (uint32_t)s;
            t[N] = t[N + 1] + (uint32_t)(s >> 32);
        }
        // result < 2p (t[N] == 0 because 2p < 2^(32N) and inputs < p)
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = t[i];
        return reduce_once(r);
    }
    ZK_HD static Fp sqr(const Fp& a) { return mul(a, a); }

    ZK_HD Fp operator+(const Fp& b) const { return add(*this, b); }
    ZK_HD Fp operator-(const Fp& b) const { return sub(*this, b); }
    ZK_HD Fp operator*(const Fp& b) const { return mul(*this, b); }

    // canonical integer -> Montgomery form
    ZK_HD static Fp to_mont(const Fp& c) { return mul(c, r2()); }
    // Montgomery form -> canonical integer (ark PrimeField::into_repr)
    ZK_HD static Fp from_mont(const Fp& m) {
        Fp o = zero();
        o.v[0] = 1;
        return mul(m, o);
    }
    ZK_HD static Fp from_u32(uint32_t x) {
        Fp c = zero();
        c.v[0] = x;
        return to_mont(c);
    }
    ZK_HD static Fp from_u64(uint64_t x) {
        Fp c = zero();
        c.v[0] = (uint32_t)x;
        c.v[1] = (uint32_t)(x >> 32);
        return to_mont(c);
    }

    // a^e for a little-endian 32-bit-limb exponent (square-and-multiply, MSB first)
    ZK_HD static Fp pow_limbs(const Fp& a, const uint32_t* e, int n) {
        Fp r = one();
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
    ZK_HD static Fp pow_u64(const Fp& a, uint64_t e) {
        uint32_t l[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
        return pow_limbs(a, l, 2);
    }
    // Fermat inverse a^(p-2); inverse(0) = 0
    ZK_HD static Fp inverse(const Fp& a) {
        uint32_t e[N];
        uint32_t borrow = 2;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)P::MOD(i) - borrow;
            e[i] = (uint32_t)t;
            borrow = (uint32_t)(t >> 32) & 1u;
        }
        return pow_limbs(a, e, N);
    }
};
