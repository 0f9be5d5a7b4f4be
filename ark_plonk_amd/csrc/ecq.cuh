// Quad-cooperative XYZZ arithmetic for the latency-bound bucket-reduction kernels.
//
// A point lives in 4 adjacent lanes of a wavefront; lane role r = lane & 3 holds ONE coordinate:
//   r = 0: X    r = 1: Y    r = 2: ZZ    r = 3: ZZZ          (F = Fs<...>, 13 registers instead of 52)
// The 12 products + 1 double product of an addition have dependency depth 4, so the four lanes run
// them as four rounds of ONE field product each (every lane executes the same instruction stream on
// its own operands; operands move between the lanes of a quad with quad shuffles, no LDS, no
// barriers): 4.5 product-times instead of 13.5 in sequence.  The reduction chains of msm.hip are
// serial in the number of additions and run at one or two wavefronts per SIMD, where a wavefront
// issues a v_mad_u64_u32 only every ~9.5 cycles -- latency, not throughput, is what they pay for.
//
// Formulas and bounds are those of ecu.cuh (add-2008-s / dbl-2008-s-1 with the lazy calculus of
// fields.cuh); the rare P = +-Q case is resolved with the quad doubling / infinity.
#pragma once
#include "ecu.cuh"

template <class F>
ZK_D F quad_xor(const F& v, int mask) {   // value held by lane ^ mask
    F r;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) r.v[i] = __shfl_xor(v.v[i], mask, 64);
    return r;
}
template <class F>
ZK_D F quad_bcast(const F& v, uint32_t src_role) {   // value held by role src_role of this quad
    const int src = (int)((__lane_id() & ~3u) | src_role);
    F r;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) r.v[i] = __shfl(v.v[i], src, 64);
    return r;
}
template <class F>
ZK_D F fsel(bool c, const F& a, const F& b) {
    F r;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
// flag held by role src_role of this quad
ZK_D bool quad_flag(bool f, uint32_t src_role) {
    const unsigned long long b = __ballot(f);
    return (b >> ((__lane_id() & ~3u) | src_role)) & 1ull;
}

// gather the four coordinates of a distributed point into every lane of the quad
template <class F>
ZK_D XYZZu<F> quad_gather(const F& c) {
    XYZZu<F> p;
    p.x = quad_bcast(c, 0);
    p.y = quad_bcast(c, 1);
    p.zz = quad_bcast(c, 2);
    p.zzz = quad_bcast(c, 3);
    return p;
}
template <class F>
ZK_D F quad_pick(const XYZZu<F>& p, uint32_t role) {
    return role == 0 ? p.x : role == 1 ? p.y : role == 2 ? p.zz : p.zzz;
}
template <class F>
ZK_D bool quad_is_inf(const F& c, uint32_t role) {   // infinity = ZZ limbs all zero
    return quad_flag(role == 2 && c.limbs_zero(), 2);
}

template <class F>
ZK_D F qdbl(const F& p, uint32_t role);

// P + Q, both distributed over the quad; every lane must call it (wave shuffles inside)
template <class F>
ZK_D F qadd(const F& p, const F& q, uint32_t role) {
    const bool p_inf = quad_is_inf(p, role), q_inf = quad_is_inf(q, role);
    // round 1: role0 U1 = X1*ZZ2, role1 S1 = Y1*ZZZ2, role2 U2 = ZZ1*X2, role3 S2 = ZZZ1*Y2
    const F m1 = F::mul(p, quad_xor(q, 2));
    // d: role0 P_ = U2 - U1, role1 R_ = S2 - S1 (roles 2, 3: unused)
    const F d = F::sub8(quad_xor(m1, 2), m1);
    // round 2: role0 PP = P_^2, role1 RR = R_^2, role2 ZZ1*ZZ2, role3 ZZZ1*ZZZ2
    const bool lo = role < 2;
    const F m2 = F::mul(fsel(lo, d, p), fsel(lo, d, q));
    // round 3: role0 PPP = P_*PP, role1 QQ = U1*PP, role2 ZZ3 = ZZ1*ZZ2*PP (role3: unused)
    const F bc_u1 = quad_bcast(m1, 0), bc_pp = quad_bcast(m2, 0);
    const F m3 = F::mul(role == 0 ? d : role == 1 ? bc_u1 : m2, fsel(role == 0, m2, bc_pp));
    // round 4: role1 X3 = RR - PPP - 2QQ, Y3 = R_*(QQ - X3) - S1*PPP (one double product); role3 ZZZ3 = ZZZ1*ZZZ2*PPP
    const F bc_ppp = quad_bcast(m3, 0);
    const F x3 = F::sub_sum3(m2, bc_ppp, m3, m3);
    const bool r1 = role == 1;
    const F m4 = F::dot2(fsel(r1, d, m2), fsel(r1, F::sub16(m3, x3), bc_ppp), fsel(r1, F::neg16(m1), F::zero()), bc_ppp);
    const F x3_from_role1 = quad_xor(x3, 1);      // shuffles stay outside role-dependent control flow
    F res = role == 0 ? x3_from_role1 : role == 2 ? m3 : m4;
    // ZZ3 == 0 mod p  <=>  same x.  Then RR == 0 mod p <=> same y: P == Q, the sum is 2P; otherwise P == -Q and the
    // sum is infinity.  Rare, and kept in quad form (the single-lane law inlined here would triple the registers).
    const bool same_x = quad_flag(role == 2 && m3.is_zero_mod_reduced(), 2);
    const bool same_y = quad_flag(role == 1 && m2.is_zero_mod_reduced(), 1);
    const bool slow = same_x && !p_inf && !q_inf;
    if (__any(slow)) {   // wave-uniform branch: qdbl is made of wave shuffles
        const F twice = qdbl<F>(p, role);
        if (slow) res = same_y ? twice : F::zero();
    }
    if (q_inf) res = p;
    if (p_inf) res = q;
    return res;
}

// 2P, distributed; every lane must call it
template <class F>
ZK_D F qdbl(const F& p, uint32_t role) {
    const bool p_inf = quad_is_inf(p, role);
    // round 1: role0 XX = X^2, role1 V = (2Y)^2
    const F a1 = fsel(role == 1, F::dbl(p), p);
    const F m1 = F::mul(a1, a1);
    // round 2: role0 S = X*V, role1 W = (2Y)*V, role2 ZZ3 = ZZ*V
    const F bc_v = quad_bcast(m1, 1);
    const F m2 = F::mul(a1, bc_v);
    // round 3: role0 M^2 (M = 3 XX), role3 ZZZ3 = ZZZ*W
    const F mm = F::add3(m1, m1, m1);
    const F bc_w = quad_bcast(m2, 1);
    const bool r0 = role == 0;
    const F m3 = F::mul(fsel(r0, mm, p), fsel(r0, mm, bc_w));
    // round 4 (role0): X3 = M^2 - 2S, Y3 = M*(S - X3) - W*Y
    const F x3 = F::sub8(m3, F::dbl(m2));
    const F m4 = F::dot2(mm, F::sub16(m2, x3), bc_w, F::neg16(quad_bcast(p, 1)));
    const F y3_from_role0 = quad_xor(m4, 1);
    F res = role == 0 ? x3 : role == 1 ? y3_from_role0 : role == 2 ? m2 : m3;
    if (p_inf) res = p;
    return res;
}
