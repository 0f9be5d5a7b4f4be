// Short-Weierstrass G1 (a = 0) group law for the MSM path, shared host/device.
//
// Stands behind ark-ec 0.3 `GroupProjective::{add_assign_mixed, add_assign, double_in_place,
// into_affine}` as used by `VariableBaseMSM::multi_scalar_mul` (reference call sites:
// plonk-core/src/commitment.rs:45, every PC::commit in proof_system/prover.rs).  The reference
// accumulates in Jacobian coordinates; here buckets are kept in extended-Jacobian XYZZ
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ = 0) because the mixed addition is
// 8M + 2S with no field inversion and fewer live registers than Jacobian madd.  Results are
// compared after affine normalisation, where the group element has a single representation.
#pragma once
#include "field.cuh"

template <class Fq>
struct Affine {
    Fq x, y;
};

template <class Fq>
struct XYZZ {
    Fq x, y, zz, zzz;

    ZK_HD static XYZZ infinity() {
        XYZZ r;
        r.x = Fq::zero();
        r.y = Fq::zero();
        r.zz = Fq::zero();
        r.zzz = Fq::zero();
        return r;
    }
    ZK_HD bool is_inf() const { return zz.is_zero(); }

    ZK_HD static XYZZ from_affine(const Affine<Fq>& p) {
        XYZZ r;
        r.x = p.x;
        r.y = p.y;
        r.zz = Fq::one();
        r.zzz = Fq::one();
        return r;
    }

    // 2 * (affine p), p != infinity  (mdbl-2008-s-1)
    ZK_HD static XYZZ dbl_affine(const Affine<Fq>& p) {
        Fq u = Fq::dbl(p.y);
        Fq v = Fq::sqr(u);
        Fq w = Fq::mul(u, v);
        Fq s = Fq::mul(p.x, v);
        Fq xx = Fq::sqr(p.x);
        Fq m = Fq::add(Fq::dbl(xx), xx);
        XYZZ r;
        r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
        r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
        r.zz = v;
        r.zzz = w;
        return r;
    }

    // 2 * this  (dbl-2008-s-1, a = 0)
    ZK_HD static XYZZ dbl(const XYZZ& p) {
        if (p.is_inf()) return p;
        Fq u = Fq::dbl(p.y);
        Fq v = Fq::sqr(u);
        Fq w = Fq::mul(u, v);
        Fq s = Fq::mul(p.x, v);
        Fq xx = Fq::sqr(p.x);
        Fq m = Fq::add(Fq::dbl(xx), xx);
        XYZZ r;
        r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
        r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
        r.zz = Fq::mul(v, p.zz);
        r.zzz = Fq::mul(w, p.zzz);
        return r;
    }

    // this + affine q (q != infinity)  (madd-2008-s); handles this == infinity, this == +-q
    ZK_HD static XYZZ madd(const XYZZ& p, const Affine<Fq>& q) {
        if (p.is_inf()) return from_affine(q);
        Fq u2 = Fq::mul(q.x, p.zz);
        Fq s2 = Fq::mul(q.y, p.zzz);
        Fq pp_ = Fq::sub(u2, p.x);
        Fq r_ = Fq::sub(s2, p.y);
        if (pp_.is_zero()) {
            if (r_.is_zero()) return dbl_affine(q);
            return infinity();
        }
        Fq pp = Fq::sqr(pp_);
        Fq ppp = Fq::mul(pp_, pp);
        Fq qq = Fq::mul(p.x, pp);
        XYZZ o;
        o.x = Fq::sub(Fq::sub(Fq::sqr(r_), ppp), Fq::dbl(qq));
        o.y = Fq::sub(Fq::mul(r_, Fq::sub(qq, o.x)), Fq::mul(p.y, ppp));
        o.zz = Fq::mul(p.zz, pp);
        o.zzz = Fq::mul(p.zzz, ppp);
        return o;
    }

    // this + q, both XYZZ  (add-2008-s); handles infinities, doubling and cancellation
    ZK_HD static XYZZ add(const XYZZ& p, const XYZZ& q) {
        if (p.is_inf()) return q;
        if (q.is_inf()) return p;
        Fq u1 = Fq::mul(p.x, q.zz);
        Fq u2 = Fq::mul(q.x, p.zz);
        Fq s1 = Fq::mul(p.y, q.zzz);
        Fq s2 = Fq::mul(q.y, p.zzz);
        Fq pp_ = Fq::sub(u2, u1);
        Fq r_ = Fq::sub(s2, s1);
        if (pp_.is_zero()) {
            if (r_.is_zero()) return dbl(p);
            return infinity();
        }
        Fq pp = Fq::sqr(pp_);
        Fq ppp = Fq::mul(pp_, pp);
        Fq qq = Fq::mul(u1, pp);
        XYZZ o;
        o.x = Fq::sub(Fq::sub(Fq::sqr(r_), ppp), Fq::dbl(qq));
        o.y = Fq::sub(Fq::mul(r_, Fq::sub(qq, o.x)), Fq::mul(s1, ppp));
        o.zz = Fq::mul(Fq::mul(p.zz, q.zz), pp);
        o.zzz = Fq::mul(Fq::mul(p.zzz, q.zzz), ppp);
        return o;
    }

    ZK_HD static XYZZ neg(const XYZZ& p) {
        XYZZ r = p;
        r.y = Fq::neg(p.y);
        return r;
    }

    // affine normalisation (one inversion); returns false for infinity
    ZK_HD bool to_affine(Affine<Fq>& out) const {
        if (is_inf()) {
            out.x = Fq::zero();
            out.y = Fq::zero();
            return false;
        }
        Fq i3 = Fq::inverse(zzz);          // 1/Z^3
        Fq zi = Fq::mul(zz, i3);           // Z^2/Z^3 = 1/Z
        Fq zi2 = Fq::sqr(zi);              // 1/ZZ
        out.x = Fq::mul(x, zi2);
        out.y = Fq::mul(y, i3);
        return true;
    }
};
