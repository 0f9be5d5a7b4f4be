// Pointwise quotient kernel: the evaluations of the quotient polynomial over the
// 4n coset, fused into one pass between the 13 coset-NTTs and the coset-iNTT, data left on the device.
//
// Reference: plonk-core/src/proof_system/quotient_poly.rs:34-178 (`compute`) --
//   quotient[i] = (gate_constraints[i] + permutation[i] + lookup[i]) / v_h_coset_4n[i]
// with gate_constraints = arithmetic + pi + range + logic + fixed-base scalar mul + curve addition
// (quotient_poly.rs:182-268; widget/arithmetic.rs:51-63, range.rs:47-74, logic.rs:65-133,
// ecc/fixed_base_scalar_mul.rs:88-156, ecc/curve_addition.rs:62-97), permutation
// (proof_system/permutation.rs:62-153) and lookup (widget/lookup.rs:97-151).
// The reference walks the 4n points three times, single-threaded, and inverts v_h at every point; here:
//   * one lane per point, every column read once, "next row" = index i+4 cyclic (the reference appends
//     e[0..4] to the vectors instead, quotient_poly.rs:75-118);
//   * X over the coset (`linear_evaluations`, preprocess.rs:209-212) is g*w^i, rebuilt per lane;
//   * v_h[i] = g^n (w^n)^i - 1 (preprocess.rs:429-452) takes only 4 values because w^n is a 4th root of
//     unity: 4 inversions on the host instead of 4n;
//   * l1_alpha_sq = coset_fft(alpha^2 L1) (quotient_poly.rs:292-294) is alpha^2 * l1 by linearity.
#include "ctx.h"

namespace {

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

constexpr uint32_t QT = 256;     // lanes per workgroup
constexpr uint32_t QROWS = 4;    // points per lane: i = blk*QT*QROWS + j*QT + lane (coalesced; X steps by w^QT)
constexpr uint32_t QTAB = 48;    // multiples of r held for the load conversion

// ---- arithmetic: the 29-bit-limb Fr type of the NTT (fieldu.cuh), lazily reduced, with the bound carried in the TYPE.
// Z<F, B> holds a value < (B / 10) * r.  A Montgomery product needs a * b < 2^261 * r / r^2 ~ 70 r^2 (169 r^2 on BN254) and
// returns < 2r; sums add their bounds; a difference a - b adds the smallest of 2r / 8r / 16r that covers b.  Every rule is a
// static_assert, so a formula that could overflow does not compile.
template <class F, int B>
struct Z {
    F v;
    ZK_D Z() {}
    ZK_D Z(const F& f) : v(f) {}
    template <int B2>
    ZK_D Z(const Z<F, B2>& o) : v(o.v) {      // widening only
        static_assert(B2 <= B, "bound would shrink");
    }
};
template <class F, int A, int B>
ZK_D Z<F, 20> operator*(const Z<F, A>& a, const Z<F, B>& b) {
    static_assert(A * B <= 6400, "Montgomery product operands too large");
    return {F::mul(a.v, b.v)};
}
template <class F, int A>
ZK_D Z<F, 20> zsqr(const Z<F, A>& a) {
    static_assert(A * A <= 6400, "square operand too large");
    return {F::sqr(a.v)};
}
template <class F, int A, int B>
ZK_D Z<F, A + B> operator+(const Z<F, A>& a, const Z<F, B>& b) {
    static_assert(A + B <= 600, "sum too large for the 261-bit container");
    return {F::add(a.v, b.v)};
}
template <int B>
struct SubK {
    static_assert(B <= 160, "subtrahend above 16r");
    static constexpr int K = B <= 20 ? 20 : B <= 80 ? 80 : 160;
};
template <class F, int A, int B>
ZK_D Z<F, A + SubK<B>::K> operator-(const Z<F, A>& a, const Z<F, B>& b) {
    static_assert(A + SubK<B>::K <= 600, "difference too large for the 261-bit container");
    if constexpr (SubK<B>::K == 20) return {F::sub2(a.v, b.v)};
    else if constexpr (SubK<B>::K == 80) return {F::sub8(a.v, b.v)};
    else return {F::sub16(a.v, b.v)};
}

template <class F>
struct QArgsU {
    const void *w_l, *w_r, *w_o, *w_4, *z, *z2, *f, *table, *h1, *h2, *pi, *l1;
    const void *q_m, *q_l, *q_r, *q_o, *q_4, *q_c, *q_arith, *q_range, *q_logic, *q_fixed, *q_var, *q_lookup;
    const void* sigma[4];
    // constants: canonical (< r) residues in the R' = 2^261 Montgomery form
    F alpha, beta, gamma, delta, eps, zeta, coeff_a, coeff_d, one;
    F alpha_sq, one_plus_delta, eps_opd;
    F rng[4], lgc[5], fxd[4], var[3], lkp[3];   // s, s^2, s^3 (, s^4, s^5) of the five separation challenges (s itself first)
    F bk[4];                                    // beta * K_k
    F c2, c3, c4, c9, c18, c81, c83;
    F g, omega, omega_t;                        // coset generator, generator of the 4n domain, omega^QT
    F inv_vh[4];                                // 2^-5 / (g^n * (omega^n)^k - 1): the 2^-5 takes the result back to the arkworks R = 2^256 form
    uint32_t rtab[QTAB][F::NL];                 // q * r, q < QTAB
    uint32_t ratio_fx;                          // floor(2^BITS / r * 2^10) - 1
    uint32_t top_shift;                         // BITS - 29 * (NL - 1)
};

// arkworks Montgomery value x * 2^256 (canonical, 8 words) -> x * 2^261 mod r, < 1.2 r:
// shift left by 5 bits (32 v < 32 r < 2^260), subtract q2 * r with q2 = floor(floor(32 v / 2^BITS) * (2^BITS / r)) <= 32 v / r
// (leaves < 2.15 r), then r once more if the rest is still >= r.
template <class F>
ZK_D Z<F, 12> ld_rp(const void* base, uint64_t idx, const uint32_t (*rtab)[F::NL], uint32_t ratio_fx, uint32_t top_shift) {
    const uint4* q = reinterpret_cast<const uint4*>(base) + 2 * idx;
    uint4 a = q[0], b = q[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    F l = F::split_words(w);
    F s;
#pragma unroll
    for (int i = F::NL - 1; i >= 1; --i) s.v[i] = ((l.v[i] << 5) | (l.v[i - 1] >> 24)) & (i == F::NL - 1 ? 0xffffffffu : F::M);
    s.v[0] = (l.v[0] << 5) & F::M;
    const uint32_t q2 = ((s.v[F::NL - 1] >> top_shift) * ratio_fx) >> 10;
    F t;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) t.v[i] = s.v[i] - rtab[q2][i];
    F::normalize(t);
    F d;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) d.v[i] = t.v[i] - rtab[1][i];
    F::normalize(d);
    const bool neg = ((int32_t)d.v[F::NL - 1]) < 0;
    F r;
#pragma unroll
    for (int i = 0; i < F::NL; ++i) r.v[i] = neg ? t.v[i] : d.v[i];
    return Z<F, 12>(r);
}

template <class F, int B>
ZK_D Z<F, 20> delta4(const Z<F, B>& f, const Z<F, 10>& one, const Z<F, 10>& c2, const Z<F, 10>& c3) {   // f(f-1)(f-2)(f-3)
    return (f * (f - one)) * ((f - c2) * (f - c3));
}

// The argument block (28 pointers + the field constants + the table of multiples of r) is read through a pointer.
// 256 VGPRs (25 spilled on BLS12-381): 2 waves per SIMD; 8.7 -> 5.0 ms for the 2^22 points of an n = 2^20 proof against the saturated
// 8 x 32-bit version of round 1 (125 products of ~600 instructions against ~110 products + 35 load conversions of ~250 / ~110).
template <class F>
__global__ void __launch_bounds__(QT, 2) quotient_points(const QArgsU<F>* __restrict__ Ap, uint64_t n4, void* out) {
    typedef Z<F, 10> C;      // a constant of the argument block (canonical)
    typedef Z<F, 12> L;      // a loaded column value
    __shared__ uint32_t rtab[QTAB][F::NL];
    const QArgsU<F>& A = *Ap;
    for (uint32_t k = threadIdx.x; k < QTAB * F::NL; k += QT) rtab[k / F::NL][k % F::NL] = A.rtab[k / F::NL][k % F::NL];
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * QT * QROWS + threadIdx.x;
    if (base >= n4) return;
    const uint32_t ratio = A.ratio_fx, tsh = A.top_shift;
    auto ld = [&](const void* p, uint64_t i) { return ld_rp<F>(p, i, rtab, ratio, tsh); };
    auto K = [](const F& c) { return C{c}; };
    const C one = K(A.one), c2 = K(A.c2), c3 = K(A.c3), c4 = K(A.c4);
    const uint32_t e[2] = {(uint32_t)base, (uint32_t)(base >> 32)};
    Z<F, 20> x{F::mul(A.g, F::pow_words(A.omega, e, 2))};
#pragma unroll 1
    for (uint32_t j = 0; j < QROWS; ++j) {
        const uint64_t i = base + (uint64_t)j * QT;
        if (i >= n4) break;
        const uint64_t nx = i + 4 >= n4 ? i + 4 - n4 : i + 4;
        const L a = ld(A.w_l, i), b = ld(A.w_r, i), c = ld(A.w_o, i), d = ld(A.w_4, i);
        const L a_n = ld(A.w_l, nx), b_n = ld(A.w_r, nx), d_n = ld(A.w_4, nx);
        // every block ends in a product, so the running sum grows by 2r per block (< 20r at the end)
        Z<F, 32> acc0;
        {   // arithmetic + pi   (widget/arithmetic.rs:51-63)
            auto t = (a * b) * ld(A.q_m, i) + a * ld(A.q_l, i) + b * ld(A.q_r, i) + c * ld(A.q_o, i) + d * ld(A.q_4, i) + ld(A.q_c, i);
            acc0 = t * ld(A.q_arith, i) + ld(A.pi, i);
        }
        Z<F, 52> acc1;
        {   // range   (widget/range.rs:47-74)
            auto t = delta4(c - c4 * d, one, c2, c3) + delta4(b - c4 * c, one, c2, c3) * K(A.rng[1]) + delta4(a - c4 * b, one, c2, c3) * K(A.rng[2])
                     + delta4(d_n - c4 * a, one, c2, c3) * K(A.rng[3]);
            acc1 = acc0 + (t * K(A.rng[0])) * ld(A.q_range, i);
        }
        Z<F, 72> acc2;
        {   // logic   (widget/logic.rs:65-133)
            const auto la = a_n - c4 * a, lb = b_n - c4 * b, ldd = d_n - c4 * d;
            const auto ab = la + lb;
            // F = w [ w (4w - 18(a+b) + 81) + 18(a^2 + b^2) - 81(a+b) + 83 ]
            const auto in = (c4 * c - K(A.c18) * ab) + K(A.c81);
            const auto F1 = ((c * in + K(A.c18) * (zsqr(la) + zsqr(lb))) - K(A.c81) * ab) + K(A.c83);
            const auto Fw = c * F1;
            const auto E = K(A.c3) * (ab + ldd) - c2 * Fw;
            const auto Bq = ld(A.q_c, i) * (K(A.c9) * ldd - K(A.c3) * ab);
            auto t = delta4(la, one, c2, c3) + delta4(lb, one, c2, c3) * K(A.lgc[1]) + delta4(ldd, one, c2, c3) * K(A.lgc[2])
                     + (c - la * lb) * K(A.lgc[3]) + (Bq + E) * K(A.lgc[4]);
            acc2 = acc1 + (t * K(A.lgc[0])) * ld(A.q_logic, i);
        }
        Z<F, 92> acc3;
        {   // fixed-base scalar multiplication   (widget/ecc/fixed_base_scalar_mul.rs:88-156)
            const L q_l = ld(A.q_l, i), q_r = ld(A.q_r, i), q_c = ld(A.q_c, i);
            const auto bit = (d_n - d) - d;
            const auto bit_cons = (bit * (bit - one)) * (bit + one);
            const auto y_alpha = zsqr(bit) * (q_r - one) + one;
            const auto x_alpha = q_l * bit;
            const auto xy_cons = (bit * q_c - c) * K(A.fxd[1]);
            const auto cabd = ((c * a) * b) * K(A.coeff_d);        // xy_alpha * acc_x * acc_y * D
            const auto x_lhs = a_n + a_n * cabd;
            const auto x_rhs = x_alpha * b + y_alpha * a;
            const auto y_lhs = b_n - b_n * cabd;
            const auto y_rhs = y_alpha * b - (K(A.coeff_a) * x_alpha) * a;
            auto t = bit_cons + (x_lhs - x_rhs) * K(A.fxd[2]) + (y_lhs - y_rhs) * K(A.fxd[3]) + xy_cons;
            acc3 = acc2 + (t * K(A.fxd[0])) * ld(A.q_fixed, i);
        }
        Z<F, 112> acc4;
        {   // curve addition: x1 = a, x3 = a_n, y1 = b, y3 = b_n, x2 = c, y2 = d, x1*y2 = d_n   (widget/ecc/curve_addition.rs:62-97)
            const auto y1x2 = b * c, y1y2 = b * d, x1x2 = a * c;
            const auto xy = a * d - d_n;
            const auto dxy = (K(A.coeff_d) * d_n) * y1x2;
            const auto x3c = (d_n + y1x2) - (a_n + a_n * dxy);
            const auto y3c = (y1y2 - K(A.coeff_a) * x1x2) - (b_n - b_n * dxy);
            auto t = xy + x3c * K(A.var[1]) + y3c * K(A.var[2]);
            acc4 = acc3 + (t * K(A.var[0])) * ld(A.q_var, i);
        }
        const L l1 = ld(A.l1, i);
        Z<F, 192> acc5;
        {   // permutation   (proof_system/permutation.rs:62-153)
            const L z_i = ld(A.z, i), z_n = ld(A.z, nx);
            const C gm = K(A.gamma), be = K(A.beta);
            const auto ag = a + gm, bg = b + gm, cg = c + gm, dg = d + gm;
            const auto id = ((((ag + K(A.bk[0]) * x) * (bg + K(A.bk[1]) * x)) * ((cg + K(A.bk[2]) * x) * (dg + K(A.bk[3]) * x))) * z_i) * K(A.alpha);
            const auto cp = ((((ag + be * ld(A.sigma[0], i)) * (bg + be * ld(A.sigma[1], i)))
                              * ((cg + be * ld(A.sigma[2], i)) * (dg + be * ld(A.sigma[3], i)))) * z_n) * K(A.alpha);
            const auto one_chk = (z_i - one) * (K(A.alpha_sq) * l1);
            acc5 = acc4 + ((id - cp) + one_chk);
        }
        Z<F, 292> acc6;
        {   // lookup   (widget/lookup.rs:97-151)
            const L f = ld(A.f, i), t_i = ld(A.table, i), t_n = ld(A.table, nx);
            const L h1_i = ld(A.h1, i), h1_n = ld(A.h1, nx), h2_i = ld(A.h2, i);
            const L z2_i = ld(A.z2, i), z2_n = ld(A.z2, nx);
            const C zt = K(A.zeta), eo = K(A.eps_opd), de = K(A.delta);
            const auto tup = a + zt * (b + zt * (c + zt * d));                 // lc([a,b,c,d], zeta), util.rs:152-171
            const auto la = (ld(A.q_lookup, i) * (tup - f)) * K(A.lkp[0]);
            const auto lb = (((z2_i * K(A.one_plus_delta)) * (K(A.eps) + f)) * ((eo + t_i) + de * t_n)) * K(A.lkp[1]);
            const auto lc = ((((eo + h1_i) + de * h2_i) * ((eo + h2_i) + de * h1_n)) * z2_n) * K(A.lkp[1]);
            const auto ldd = ((z2_i - one) * l1) * K(A.lkp[2]);
            acc6 = acc5 + (((la + lb) - lc) + ldd);
        }
        // one last product: divides by the vanishing polynomial, takes the result back to the arkworks form (the 2^-5 inside
        // inv_vh) and brings it under 2r for the canonical store
        const F y = F::mul(acc6.v, A.inv_vh[i & 3]);
        {
            uint32_t w[8];
            F::canonical_lt2p(y).pack_words(w);
            uint4* q = reinterpret_cast<uint4*>(out) + 2 * i;
            q[0] = make_uint4(w[0], w[1], w[2], w[3]);
            q[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        x = x * K(A.omega_t);
    }
}

// host: arkworks-form Fr (R = 2^256) -> the canonical R' = 2^261 residue as 29-bit limbs
template <class C>
typename C::FrU to_rp_host(const typename C::Fr& v) {
    typedef typename C::Fr Fr;
    Fr t = v;
    for (int k = 0; k < 5; ++k) t = Fr::add(t, t);
    return C::FrU::split_words(t.v);
}

template <class C>
int quotient_run(zk_ctx* c, uint32_t log_n, const zk_quotient_args* q, void* d_out) {
    typedef typename C::Fr Fr;
    typedef typename C::FrU FU;
    if (log_n + 2 > (uint32_t)C::FrP::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
    const uint64_t n = 1ull << log_n, n4 = 4 * n;
    static QArgsU<FU> A;          // 4 KiB: filled under the ctx lock of the caller ... and a process-wide one here
    static std::mutex a_mu;
    std::lock_guard<std::mutex> lk(a_mu);
    const void* const cols[28] = {q->w_l, q->w_r, q->w_o, q->w_4, q->z, q->z2, q->f, q->table, q->h1, q->h2, q->pi, q->l1,
                                  q->q_m, q->q_l, q->q_r, q->q_o, q->q_4, q->q_c, q->q_arith, q->q_range, q->q_logic,
                                  q->q_fixed_group_add, q->q_variable_group_add, q->q_lookup, q->sigma[0], q->sigma[1], q->sigma[2], q->sigma[3]};
    for (const void* p : cols)
        if (!p) return ZK_ERR_BAD_ARG;
    A.w_l = q->w_l; A.w_r = q->w_r; A.w_o = q->w_o; A.w_4 = q->w_4; A.z = q->z; A.z2 = q->z2; A.f = q->f; A.table = q->table;
    A.h1 = q->h1; A.h2 = q->h2; A.pi = q->pi; A.l1 = q->l1;
    A.q_m = q->q_m; A.q_l = q->q_l; A.q_r = q->q_r; A.q_o = q->q_o; A.q_4 = q->q_4; A.q_c = q->q_c; A.q_arith = q->q_arith;
    A.q_range = q->q_range; A.q_logic = q->q_logic; A.q_fixed = q->q_fixed_group_add; A.q_var = q->q_variable_group_add; A.q_lookup = q->q_lookup;
    for (int k = 0; k < 4; ++k) A.sigma[k] = q->sigma[k];
    auto ldc = [](const uint64_t* src) { Fr v; memcpy(v.v, src, 32); return v; };
    auto U = [](const Fr& v) { return to_rp_host<C>(v); };
    const Fr alpha = ldc(q->alpha), beta = ldc(q->beta), delta = ldc(q->delta), eps = ldc(q->epsilon);
    A.alpha = U(alpha); A.beta = U(beta); A.gamma = U(ldc(q->gamma)); A.delta = U(delta); A.eps = U(eps); A.zeta = U(ldc(q->zeta));
    A.coeff_a = U(ldc(q->coeff_a)); A.coeff_d = U(ldc(q->coeff_d)); A.one = U(Fr::one());
    A.alpha_sq = U(Fr::sqr(alpha));
    const Fr opd = Fr::add(Fr::one(), delta);
    A.one_plus_delta = U(opd);
    A.eps_opd = U(Fr::mul(eps, opd));
    auto powers = [&](const uint64_t* src, FU* dst, int cnt) {
        const Fr sv = ldc(src);
        Fr cur = sv;
        for (int k = 0; k < cnt; ++k) {
            dst[k] = U(cur);
            cur = Fr::mul(cur, sv);
        }
    };
    powers(q->range_challenge, A.rng, 4);        // s, s^2, s^3, s^4: the widget uses kappa = s^2, kappa^2, kappa^3 ... see below
    powers(q->logic_challenge, A.lgc, 5);
    powers(q->fixed_base_challenge, A.fxd, 4);
    powers(q->var_base_challenge, A.var, 3);
    powers(q->lookup_challenge, A.lkp, 3);
    {   // the widgets separate their terms with kappa = s^2 and scale the sum by s: slots [1..] hold kappa^k = s^(2k)
        auto kappas = [&](const uint64_t* src, FU* dst, int cnt) {
            const Fr sv = ldc(src), kp = Fr::sqr(sv);
            Fr cur = kp;
            for (int k = 1; k < cnt; ++k) {
                dst[k] = U(cur);
                cur = Fr::mul(cur, kp);
            }
        };
        kappas(q->range_challenge, A.rng, 4);
        kappas(q->logic_challenge, A.lgc, 5);
        kappas(q->fixed_base_challenge, A.fxd, 4);
        kappas(q->var_base_challenge, A.var, 3);
        // the lookup widget uses plain powers s, s^2, s^3 (widget/lookup.rs:97-151): already in place
    }
    const uint32_t Kp[4] = {1, 7, 13, 17};   // permutation/constants.rs:12-22
    for (int k = 0; k < 4; ++k) A.bk[k] = U(Fr::mul(beta, Fr::from_u32(Kp[k])));
    A.c2 = U(Fr::from_u32(2)); A.c3 = U(Fr::from_u32(3)); A.c4 = U(Fr::from_u32(4)); A.c9 = U(Fr::from_u32(9));
    A.c18 = U(Fr::from_u32(18)); A.c81 = U(Fr::from_u32(81)); A.c83 = U(Fr::from_u32(83));
    Fr root;
    for (int i = 0; i < Fr::N; ++i) root.v[i] = C::FrP::ROOT(i);
    for (uint32_t k = log_n + 2; k < (uint32_t)C::FrP::TWO_ADICITY; ++k) root = Fr::sqr(root);
    A.omega = U(root);
    A.omega_t = U(Fr::pow_u64(root, QT));
    const Fr g = Fr::from_u32(C::FrP::GENERATOR);
    A.g = U(g);
    // v_h over the coset takes 4 values: g^n * (omega^n)^k - 1; the stored inverse carries 2^-5 (R'/R)
    const Fr gn = Fr::pow_u64(g, n), wn = Fr::pow_u64(root, n);
    const Fr inv32 = Fr::inverse(Fr::from_u32(32));
    Fr cur = gn;
    for (int k = 0; k < 4; ++k) {
        Fr vh = Fr::sub(cur, Fr::one());
        if (vh.is_zero()) return ZK_ERR_NOT_INVERTIBLE;
        A.inv_vh[k] = U(Fr::mul(Fr::inverse(vh), inv32));
        cur = Fr::mul(cur, wn);
    }
    {   // q * r as 29-bit limbs, and floor(2^BITS / r * 2^10) - 1
        uint32_t rw[8];
        for (int i = 0; i < 8; ++i) rw[i] = C::FrP::MOD(i);
        FU acc = FU::zero();
        const FU rl = FU::split_words(rw);
        for (uint32_t k = 0; k < QTAB; ++k) {
            for (int i = 0; i < FU::NL; ++i) A.rtab[k][i] = acc.v[i];
            acc = FU::add(acc, rl);
        }
        long double rv = 0;
        for (int i = Fr::N - 1; i >= 0; --i) rv = rv * 4294967296.0L + (long double)C::FrP::MOD(i);
        const long double ratio = ldexpl(1.0L, C::FrP::BITS + 10) / rv;
        A.ratio_fx = (uint32_t)floorl(ratio) - 1;
        A.top_shift = (uint32_t)(C::FrP::BITS - 29 * (FU::NL - 1));
    }
    ProfScope ps(c, "quotient");
    int rc = c->msm_tmp.ensure(sizeof A);
    if (rc) return rc;
    ZK_HIP_TRY(hipMemcpyAsync(c->msm_tmp.p, &A, sizeof A, hipMemcpyHostToDevice, c->stream));
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));   // A is reused by the next call
    const unsigned blocks = (unsigned)((n4 + (uint64_t)QT * QROWS - 1) / ((uint64_t)QT * QROWS));
    hipLaunchKernelGGL(quotient_points<FU>, dim3(blocks), dim3(QT), 0, c->stream, (const QArgsU<FU>*)c->msm_tmp.p, n4, d_out);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

}  // namespace

int quotient_evals_dev(zk_ctx* c, int curve, uint32_t log_n, const zk_quotient_args* q, void* d_out) {
    if (curve == ZK_CURVE_BLS12_381) return quotient_run<CurveBls>(c, log_n, q, d_out);
    if (curve == ZK_CURVE_BN254) return quotient_run<CurveBn>(c, log_n, q, d_out);
    return ZK_ERR_BAD_ARG;
}
