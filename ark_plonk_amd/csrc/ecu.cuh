// G1 group law (a = 0) over the signed-limb base field Fs (fields.cuh) -- the device form of ec.cuh.
//
// Same formulas as ec.cuh (madd-2008-s / add-2008-s / dbl-2008-s-1, XYZZ coordinates) with the lazy
// reduction calculus of fields.cuh: limbs are signed, so a difference is just a difference, and
//   * products and squares come out in (-p, p) with strict limbs; stored coordinates are |X| < 4p (a
//     sum of four products, almost-balanced limbs) and Y, ZZ, ZZZ products;
//   * every operand of a product is a product, a stored coordinate or ONE sum / difference of those
//     (|value| < 12p, limbs re-centred by one carry step) -- the bound fields.cuh asks for;
//   * Y3 = R*(Q - X3) - Y1*PPP is ONE double product with one Montgomery reduction (dot2 against -Y1);
//   * P == +-Q is detected on the *result*: ZZ3 = ZZ1*PP is a product, so ZZ3 == 0 mod p iff its strict
//     limbs are those of 0 (or +-p) -- compares on one limb in the common path.
// The names sub8 / sub16 / neg16 are kept from the unsigned form (they said which multiple of p kept a
// difference positive there); over Fs they are all plain limb-wise differences.
// Infinity is ZZ = 0 with all limbs zero (only ever created explicitly).
#pragma once
#include "fields.cuh"

template <class F>
struct AffineU {
    F x, y;  // residues in (-p, p), strict limbs, R' Montgomery domain; x = y = 0 limbs encodes "no point"
    ZK_HD bool is_null() const { return x.limbs_zero() && y.limbs_zero(); }
};

template <class F>
struct XYZZu {
    F x, y, zz, zzz;

    ZK_HD static XYZZu infinity() {
        XYZZu r;
        r.x = F::zero();
        r.y = F::zero();
        r.zz = F::zero();
        r.zzz = F::zero();
        return r;
    }
    ZK_HD bool is_inf() const { return zz.limbs_zero(); }

    ZK_HD static XYZZu from_affine(const AffineU<F>& p) {
        XYZZu r;
        r.x = p.x;
        r.y = p.y;
        r.zz = F::one();
        r.zzz = F::one();
        return r;
    }

    // 2 * (affine p), p not null
    ZK_HD static XYZZu dbl_affine(const AffineU<F>& p) {
        F u = F::dbl(p.y);
        F v = F::sqr(u);
        F w = F::mul(u, v);
        F s = F::mul(p.x, v);
        F xx = F::sqr(p.x);
        F m = F::add3(xx, xx, xx);
        XYZZu r;
        r.x = F::sub8(F::sqr(m), F::dbl(s));
        r.y = F::dot2(m, F::sub16(s, r.x), w, F::neg16(p.y));   // m*(s - x3) - w*y, one reduction
        r.zz = v;
        r.zzz = w;
        return r;
    }

    ZK_HD static XYZZu dbl(const XYZZu& p) {
        if (p.is_inf()) return p;
        F u = F::dbl(p.y);
        F v = F::sqr(u);
        F w = F::mul(u, v);
        F s = F::mul(p.x, v);
        F xx = F::sqr(p.x);
        F m = F::add3(xx, xx, xx);
        XYZZu r;
        r.x = F::sub8(F::sqr(m), F::dbl(s));
        r.y = F::dot2(m, F::sub16(s, r.x), w, F::neg16(p.y));
        r.zz = F::mul(v, p.zz);
        r.zzz = F::mul(w, p.zzz);
        return r;
    }

    // this + affine q (q not null); handles this == infinity, this == +-q
    ZK_HD static XYZZu madd(const XYZZu& p, const AffineU<F>& q) {
        if (p.is_inf()) return from_affine(q);
        F u2 = F::mul(q.x, p.zz);
        F s2 = F::mul(q.y, p.zzz);
        F pp_ = F::sub16(u2, p.x);
        F r_ = F::sub16(s2, p.y);
        F pp = F::sqr(pp_);
        F rr = F::sqr(r_);
        XYZZu o;
        o.zz = F::mul(p.zz, pp);
        if (o.zz.is_zero_mod_reduced()) {  // P == 0  <=>  same x
            if (rr.is_zero_mod_reduced()) return dbl_affine(q);
            return infinity();
        }
        F ppp = F::mul(pp_, pp);
        F qq = F::mul(p.x, pp);
        o.x = F::sub_sum3(rr, ppp, qq, qq);
        o.y = F::dot2(r_, F::sub16(qq, o.x), F::neg16(p.y), ppp);   // r*(qq - x3) - y1*ppp, one reduction
        o.zzz = F::mul(p.zzz, ppp);
        return o;
    }

    // this + q, both XYZZ; handles infinities, doubling and cancellation
    ZK_HD static XYZZu add(const XYZZu& p, const XYZZu& q) {
        if (p.is_inf()) return q;
        if (q.is_inf()) return p;
        F u1 = F::mul(p.x, q.zz);
        F u2 = F::mul(q.x, p.zz);
        F s1 = F::mul(p.y, q.zzz);
        F s2 = F::mul(q.y, p.zzz);
        F pp_ = F::sub8(u2, u1);
        F r_ = F::sub8(s2, s1);
        F pp = F::sqr(pp_);
        F rr = F::sqr(r_);
        XYZZu o;
        o.zz = F::mul(F::mul(p.zz, q.zz), pp);
        if (o.zz.is_zero_mod_reduced()) {
            if (rr.is_zero_mod_reduced()) return dbl(p);
            return infinity();
        }
        F ppp = F::mul(pp_, pp);
        F qq = F::mul(u1, pp);
        o.x = F::sub_sum3(rr, ppp, qq, qq);
        o.y = F::dot2(r_, F::sub16(qq, o.x), F::neg16(s1), ppp);
        o.zzz = F::mul(F::mul(p.zzz, q.zzz), ppp);
        return o;
    }

    // affine normalisation; returns false for infinity.  Output coordinates are products (< 2p).
    ZK_HD bool to_affine(AffineU<F>& out) const {
        if (is_inf()) {
            out.x = F::zero();
            out.y = F::zero();
            return false;
        }
        F i3 = F::inverse(zzz);
        F zi = F::mul(zz, i3);
        F zi2 = F::sqr(zi);
        out.x = F::mul(x, zi2);
        out.y = F::mul(y, i3);
        return true;
    }
};
