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

template <class Fr>
struct QArgs {
    const void *w_l, *w_r, *w_o, *w_4, *z, *z2, *f, *table, *h1, *h2, *pi, *l1;
    const void *q_m, *q_l, *q_r, *q_o, *q_4, *q_c, *q_arith, *q_range, *q_logic, *q_fixed, *q_var, *q_lookup;
    const void* sigma[4];
    Fr alpha, beta, gamma, delta, eps, zeta, s_range, s_logic, s_fixed, s_var, s_lookup, coeff_a, coeff_d;
    Fr alpha_sq, one_plus_delta, eps_opd;
    Fr bk[4];            // beta * K_k
    Fr c2, c3, c4, c9, c18, c81, c83;
    Fr g, omega, omega_t;   // coset generator, generator of the 4n domain, omega^QT
    Fr inv_vh[4];        // 1 / (g^n * (omega^n)^k - 1)
};

template <class Fr>
ZK_D Fr delta4(const Fr& f, const Fr& one, const Fr& c2, const Fr& c3) {   // f(f-1)(f-2)(f-3)
    return Fr::mul(Fr::mul(f, Fr::sub(f, one)), Fr::mul(Fr::sub(f, c2), Fr::sub(f, c3)));
}

// The argument block (28 pointers + 37 field constants) is read through a pointer: as by-value kernel
// arguments the constants alone need 300 SGPRs and spill; two waves per SIMD need <= 256 VGPRs.
template <class Fr>
__global__ void __launch_bounds__(QT, 2) quotient_points(const QArgs<Fr>* __restrict__ Ap, uint64_t n4, void* out) {
    const QArgs<Fr>& A = *Ap;
    const uint64_t base = (uint64_t)blockIdx.x * QT * QROWS + threadIdx.x;
    if (base >= n4) return;
    const Fr one = Fr::one();
    Fr x = Fr::mul(A.g, Fr::pow_u64(A.omega, base));
    for (uint32_t j = 0; j < QROWS; ++j) {
        const uint64_t i = base + (uint64_t)j * QT;
        if (i >= n4) break;
        const uint64_t nx = i + 4 >= n4 ? i + 4 - n4 : i + 4;
        const Fr a = ld_fr<Fr>(A.w_l, i), b = ld_fr<Fr>(A.w_r, i), c = ld_fr<Fr>(A.w_o, i), d = ld_fr<Fr>(A.w_4, i);
        const Fr a_n = ld_fr<Fr>(A.w_l, nx), b_n = ld_fr<Fr>(A.w_r, nx), d_n = ld_fr<Fr>(A.w_4, nx);
        const Fr q_l = ld_fr<Fr>(A.q_l, i), q_r = ld_fr<Fr>(A.q_r, i), q_c = ld_fr<Fr>(A.q_c, i);
        Fr acc;
        {   // arithmetic + pi
            Fr t = Fr::mul(Fr::mul(a, b), ld_fr<Fr>(A.q_m, i));
            t = Fr::add(t, Fr::mul(a, q_l));
            t = Fr::add(t, Fr::mul(b, q_r));
            t = Fr::add(t, Fr::mul(c, ld_fr<Fr>(A.q_o, i)));
            t = Fr::add(t, Fr::mul(d, ld_fr<Fr>(A.q_4, i)));
            t = Fr::add(t, q_c);
            acc = Fr::add(Fr::mul(t, ld_fr<Fr>(A.q_arith, i)), ld_fr<Fr>(A.pi, i));
        }
        {   // range
            const Fr s = A.s_range, k = Fr::sqr(s), k2 = Fr::sqr(k), k3 = Fr::mul(k2, k);
            Fr t = delta4<Fr>(Fr::sub(c, Fr::mul(A.c4, d)), one, A.c2, A.c3);
            t = Fr::add(t, Fr::mul(delta4<Fr>(Fr::sub(b, Fr::mul(A.c4, c)), one, A.c2, A.c3), k));
            t = Fr::add(t, Fr::mul(delta4<Fr>(Fr::sub(a, Fr::mul(A.c4, b)), one, A.c2, A.c3), k2));
            t = Fr::add(t, Fr::mul(delta4<Fr>(Fr::sub(d_n, Fr::mul(A.c4, a)), one, A.c2, A.c3), k3));
            acc = Fr::add(acc, Fr::mul(Fr::mul(t, s), ld_fr<Fr>(A.q_range, i)));
        }
        {   // logic
            const Fr s = A.s_logic, k = Fr::sqr(s), k2 = Fr::sqr(k), k3 = Fr::mul(k2, k), k4 = Fr::mul(k3, k);
            const Fr la = Fr::sub(a_n, Fr::mul(A.c4, a)), lb = Fr::sub(b_n, Fr::mul(A.c4, b)), ld = Fr::sub(d_n, Fr::mul(A.c4, d));
            const Fr w = c, ab = Fr::add(la, lb);
            // F = w [ w (4w - 18(a+b) + 81) + 18(a^2 + b^2) - 81(a+b) + 83 ]
            Fr in = Fr::add(Fr::sub(Fr::mul(A.c4, w), Fr::mul(A.c18, ab)), A.c81);
            Fr F = Fr::mul(w, in);
            F = Fr::add(F, Fr::mul(A.c18, Fr::add(Fr::sqr(la), Fr::sqr(lb))));
            F = Fr::add(Fr::sub(F, Fr::mul(A.c81, ab)), A.c83);
            F = Fr::mul(w, F);
            const Fr E = Fr::sub(Fr::mul(A.c3, Fr::add(ab, ld)), Fr::mul(A.c2, F));
            const Fr B = Fr::mul(q_c, Fr::sub(Fr::mul(A.c9, ld), Fr::mul(A.c3, ab)));
            Fr t = delta4<Fr>(la, one, A.c2, A.c3);
            t = Fr::add(t, Fr::mul(delta4<Fr>(lb, one, A.c2, A.c3), k));
            t = Fr::add(t, Fr::mul(delta4<Fr>(ld, one, A.c2, A.c3), k2));
            t = Fr::add(t, Fr::mul(Fr::sub(w, Fr::mul(la, lb)), k3));
            t = Fr::add(t, Fr::mul(Fr::add(B, E), k4));
            acc = Fr::add(acc, Fr::mul(Fr::mul(t, s), ld_fr<Fr>(A.q_logic, i)));
        }
        {   // fixed-base scalar multiplication
            const Fr s = A.s_fixed, k = Fr::sqr(s), k2 = Fr::sqr(k), k3 = Fr::mul(k2, k);
            const Fr bit = Fr::sub(Fr::sub(d_n, d), d);
            const Fr bit_cons = Fr::mul(Fr::mul(bit, Fr::sub(bit, one)), Fr::add(bit, one));
            const Fr y_alpha = Fr::add(Fr::mul(Fr::sqr(bit), Fr::sub(q_r, one)), one);
            const Fr x_alpha = Fr::mul(q_l, bit);
            const Fr xy_cons = Fr::mul(Fr::sub(Fr::mul(bit, q_c), c), k);
            const Fr cabd = Fr::mul(Fr::mul(Fr::mul(c, a), b), A.coeff_d);        // xy_alpha * acc_x * acc_y * D
            const Fr x_lhs = Fr::add(a_n, Fr::mul(a_n, cabd));
            const Fr x_rhs = Fr::add(Fr::mul(x_alpha, b), Fr::mul(y_alpha, a));
            const Fr y_lhs = Fr::sub(b_n, Fr::mul(b_n, cabd));
            const Fr y_rhs = Fr::sub(Fr::mul(y_alpha, b), Fr::mul(Fr::mul(A.coeff_a, x_alpha), a));
            Fr t = Fr::add(bit_cons, Fr::mul(Fr::sub(x_lhs, x_rhs), k2));
            t = Fr::add(t, Fr::mul(Fr::sub(y_lhs, y_rhs), k3));
            t = Fr::add(t, xy_cons);
            acc = Fr::add(acc, Fr::mul(Fr::mul(t, s), ld_fr<Fr>(A.q_fixed, i)));
        }
        {   // curve addition: x1 = a, x3 = a_n, y1 = b, y3 = b_n, x2 = c, y2 = d, x1*y2 = d_n
            const Fr s = A.s_var, k = Fr::sqr(s), k2 = Fr::sqr(k);
            const Fr y1x2 = Fr::mul(b, c), y1y2 = Fr::mul(b, d), x1x2 = Fr::mul(a, c);
            const Fr xy = Fr::sub(Fr::mul(a, d), d_n);
            const Fr dxy = Fr::mul(Fr::mul(A.coeff_d, d_n), y1x2);
            const Fr x3c = Fr::sub(Fr::add(d_n, y1x2), Fr::add(a_n, Fr::mul(a_n, dxy)));
            const Fr y3c = Fr::sub(Fr::sub(y1y2, Fr::mul(A.coeff_a, x1x2)), Fr::sub(b_n, Fr::mul(b_n, dxy)));
            Fr t = Fr::add(xy, Fr::mul(x3c, k));
            t = Fr::add(t, Fr::mul(y3c, k2));
            acc = Fr::add(acc, Fr::mul(Fr::mul(t, s), ld_fr<Fr>(A.q_var, i)));
        }
        const Fr l1 = ld_fr<Fr>(A.l1, i);
        {   // permutation
            const Fr z_i = ld_fr<Fr>(A.z, i), z_n = ld_fr<Fr>(A.z, nx);
            const Fr ag = Fr::add(a, A.gamma), bg = Fr::add(b, A.gamma), cg = Fr::add(c, A.gamma), dg = Fr::add(d, A.gamma);
            Fr id = Fr::mul(Fr::add(ag, Fr::mul(A.bk[0], x)), Fr::add(bg, Fr::mul(A.bk[1], x)));
            id = Fr::mul(id, Fr::mul(Fr::add(cg, Fr::mul(A.bk[2], x)), Fr::add(dg, Fr::mul(A.bk[3], x))));
            id = Fr::mul(Fr::mul(id, z_i), A.alpha);
            Fr cp = Fr::mul(Fr::add(ag, Fr::mul(A.beta, ld_fr<Fr>(A.sigma[0], i))), Fr::add(bg, Fr::mul(A.beta, ld_fr<Fr>(A.sigma[1], i))));
            cp = Fr::mul(cp, Fr::mul(Fr::add(cg, Fr::mul(A.beta, ld_fr<Fr>(A.sigma[2], i))), Fr::add(dg, Fr::mul(A.beta, ld_fr<Fr>(A.sigma[3], i)))));
            cp = Fr::mul(Fr::mul(cp, z_n), A.alpha);
            const Fr one_chk = Fr::mul(Fr::sub(z_i, one), Fr::mul(A.alpha_sq, l1));
            acc = Fr::add(acc, Fr::add(Fr::sub(id, cp), one_chk));
        }
        {   // lookup
            const Fr f = ld_fr<Fr>(A.f, i), t_i = ld_fr<Fr>(A.table, i), t_n = ld_fr<Fr>(A.table, nx);
            const Fr h1_i = ld_fr<Fr>(A.h1, i), h1_n = ld_fr<Fr>(A.h1, nx), h2_i = ld_fr<Fr>(A.h2, i);
            const Fr z2_i = ld_fr<Fr>(A.z2, i), z2_n = ld_fr<Fr>(A.z2, nx);
            const Fr ls = A.s_lookup, ls2 = Fr::sqr(ls), ls3 = Fr::mul(ls2, ls);
            Fr tup = Fr::add(c, Fr::mul(A.zeta, d));                          // lc([a,b,c,d], zeta), util.rs:152-171
            tup = Fr::add(b, Fr::mul(A.zeta, tup));
            tup = Fr::add(a, Fr::mul(A.zeta, tup));
            const Fr la = Fr::mul(Fr::mul(ld_fr<Fr>(A.q_lookup, i), Fr::sub(tup, f)), ls);
            Fr lb = Fr::mul(Fr::mul(z2_i, A.one_plus_delta), Fr::add(A.eps, f));
            lb = Fr::mul(Fr::mul(lb, Fr::add(Fr::add(A.eps_opd, t_i), Fr::mul(A.delta, t_n))), ls2);
            Fr lc = Fr::mul(Fr::add(Fr::add(A.eps_opd, h1_i), Fr::mul(A.delta, h2_i)), Fr::add(Fr::add(A.eps_opd, h2_i), Fr::mul(A.delta, h1_n)));
            lc = Fr::mul(Fr::mul(lc, z2_n), ls2);
            const Fr ldd = Fr::mul(Fr::mul(Fr::sub(z2_i, one), l1), ls3);
            acc = Fr::add(acc, Fr::add(Fr::sub(Fr::add(la, lb), lc), ldd));
        }
        st_fr<Fr>(out, i, Fr::mul(acc, A.inv_vh[i & 3]));
        x = Fr::mul(x, A.omega_t);
    }
}

template <class C>
int quotient_run(zk_ctx* c, uint32_t log_n, const zk_quotient_args* q, void* d_out) {
    typedef typename C::Fr Fr;
    if (log_n + 2 > (uint32_t)C::FrP::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
    const uint64_t n = 1ull << log_n, n4 = 4 * n;
    QArgs<Fr> A;
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
    auto ldc = [](Fr& dst, const uint64_t* src) { memcpy(dst.v, src, 32); };
    ldc(A.alpha, q->alpha); ldc(A.beta, q->beta); ldc(A.gamma, q->gamma); ldc(A.delta, q->delta); ldc(A.eps, q->epsilon); ldc(A.zeta, q->zeta);
    ldc(A.s_range, q->range_challenge); ldc(A.s_logic, q->logic_challenge); ldc(A.s_fixed, q->fixed_base_challenge);
    ldc(A.s_var, q->var_base_challenge); ldc(A.s_lookup, q->lookup_challenge); ldc(A.coeff_a, q->coeff_a); ldc(A.coeff_d, q->coeff_d);
    A.alpha_sq = Fr::sqr(A.alpha);
    A.one_plus_delta = Fr::add(Fr::one(), A.delta);
    A.eps_opd = Fr::mul(A.eps, A.one_plus_delta);
    const uint32_t K[4] = {1, 7, 13, 17};   // permutation/constants.rs:12-22
    for (int k = 0; k < 4; ++k) A.bk[k] = Fr::mul(A.beta, Fr::from_u32(K[k]));
    A.c2 = Fr::from_u32(2); A.c3 = Fr::from_u32(3); A.c4 = Fr::from_u32(4); A.c9 = Fr::from_u32(9);
    A.c18 = Fr::from_u32(18); A.c81 = Fr::from_u32(81); A.c83 = Fr::from_u32(83);
    Fr root;
    for (int i = 0; i < Fr::N; ++i) root.v[i] = C::FrP::ROOT(i);
    for (uint32_t k = log_n + 2; k < (uint32_t)C::FrP::TWO_ADICITY; ++k) root = Fr::sqr(root);
    A.omega = root;
    A.omega_t = Fr::pow_u64(root, QT);
    A.g = Fr::from_u32(C::FrP::GENERATOR);
    // v_h over the coset takes 4 values: g^n * (omega^n)^k - 1
    const Fr gn = Fr::pow_u64(A.g, n), wn = Fr::pow_u64(root, n);
    Fr cur = gn;
    for (int k = 0; k < 4; ++k) {
        Fr vh = Fr::sub(cur, Fr::one());
        if (vh.is_zero()) return ZK_ERR_NOT_INVERTIBLE;
        A.inv_vh[k] = Fr::inverse(vh);
        cur = Fr::mul(cur, wn);
    }
    ProfScope ps(c, "quotient");
    int rc = c->msm_tmp.ensure(sizeof A);
    if (rc) return rc;
    ZK_HIP_TRY(hipMemcpyAsync(c->msm_tmp.p, &A, sizeof A, hipMemcpyHostToDevice, c->stream));
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));   // A lives on this stack frame
    const unsigned blocks = (unsigned)((n4 + (uint64_t)QT * QROWS - 1) / ((uint64_t)QT * QROWS));
    hipLaunchKernelGGL(quotient_points<Fr>, dim3(blocks), dim3(QT), 0, c->stream, (const QArgs<Fr>*)c->msm_tmp.p, n4, d_out);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

}  // namespace

int quotient_evals_dev(zk_ctx* c, int curve, uint32_t log_n, const zk_quotient_args* q, void* d_out) {
    if (curve == ZK_CURVE_BLS12_381) return quotient_run<CurveBls>(c, log_n, q, d_out);
    if (curve == ZK_CURVE_BN254) return quotient_run<CurveBn>(c, log_n, q, d_out);
    return ZK_ERR_BAD_ARG;
}
