// Device self-test of the quad-cooperative point arithmetic (ecq.cuh) against the single-lane law
// (ecu.cuh): every quad builds P = a*G and Q = b*G with the single-lane code, runs qadd / qdbl and the
// single-lane add / dbl, and compares the results projectively.  Cases per quad index i % 6:
//   0 generic, 1 Q = P (doubling), 2 Q = -P (cancellation), 3 P = infinity, 4 Q = infinity, 5 both infinity.
#include "ctx.h"
#include "ecq.cuh"

namespace {

template <class F>
ZK_D bool same_point(const XYZZu<F>& a, const XYZZu<F>& b) {
    if (a.is_inf() || b.is_inf()) return a.is_inf() && b.is_inf();
    // x_a/zz_a == x_b/zz_b and y_a/zzz_a == y_b/zzz_b
    const F dx = F::sub8(F::mul(a.x, b.zz), F::mul(b.x, a.zz));
    const F dy = F::sub8(F::mul(a.y, b.zzz), F::mul(b.y, a.zzz));
    return dx.is_zero_mod() && dy.is_zero_mod();
}

template <class Cv>
__global__ void __launch_bounds__(256) quad_selftest(uint32_t n_quads, uint32_t* mismatches) {
    typedef typename Cv::FqU F;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = tid >> 2, role = tid & 3;
    // no early exit: the quad primitives are wave shuffles
    const bool live = i < n_quads;
    uint32_t gw[2 * F::SAT];
#pragma unroll
    for (int k = 0; k < F::SAT; ++k) {
        gw[k] = Cv::FqP::GX(k);
        gw[F::SAT + k] = Cv::FqP::GY(k);
    }
    AffineU<F> G;
    G.x = F::canonical_lt2p(F::from_sat(gw));
    G.y = F::canonical_lt2p(F::from_sat(gw + F::SAT));
    auto mulG = [&](uint32_t k) {
        XYZZu<F> acc = XYZZu<F>::infinity();
        for (int b = 15; b >= 0; --b) {
            acc = XYZZu<F>::dbl(acc);
            if ((k >> b) & 1u) acc = XYZZu<F>::madd(acc, G);
        }
        return acc;
    };
    const uint32_t kind = i % 6;
    XYZZu<F> P = mulG(3 + (i % 1000)), Q = mulG(7 + 2 * (i % 999));
    if (kind == 1) Q = P;
    if (kind == 2) {
        Q = P;
        Q.y = F::neg16(Q.y);
    }
    if (kind == 3 || kind == 5) P = XYZZu<F>::infinity();
    if (kind == 4 || kind == 5) Q = XYZZu<F>::infinity();
    const XYZZu<F> ref_add = XYZZu<F>::add(P, Q), ref_dbl = XYZZu<F>::dbl(P);
    const F qa = qadd<F>(quad_pick(P, role), quad_pick(Q, role), role);
    const F qd = qdbl<F>(quad_pick(P, role), role);
    // a chain: ((P + Q) + Q) doubled, the way the reduction kernels feed results back in
    const F qc = qdbl<F>(qadd<F>(qa, quad_pick(Q, role), role), role);
    const XYZZu<F> ref_c = XYZZu<F>::dbl(XYZZu<F>::add(ref_add, Q));
    const XYZZu<F> ga = quad_gather(qa), gd = quad_gather(qd), gc = quad_gather(qc);
    if (live && role == 0) {
        uint32_t bad = 0;
        if (!same_point(ga, ref_add)) bad |= 1;
        if (!same_point(gd, ref_dbl)) bad |= 2;
        if (!same_point(gc, ref_c)) bad |= 4;
        if (bad) {
            atomicAdd(&mismatches[0], 1u);
            atomicOr(&mismatches[1], bad << (4 * (kind % 6)));
        }
    }
}

template <class Cv>
int run(zk_ctx* c, uint32_t n_quads, uint32_t* h_out) {
    int rc = c->msm_tmp.ensure(64);
    if (rc) return rc;
    uint32_t* d = (uint32_t*)c->msm_tmp.p;
    ZK_HIP_TRY(hipMemsetAsync(d, 0, 8, c->stream));
    const unsigned blocks = (unsigned)(((uint64_t)n_quads * 4 + 255) / 256);
    hipLaunchKernelGGL(quad_selftest<Cv>, dim3(blocks), dim3(256), 0, c->stream, n_quads, d);
    ZK_HIP_TRY(hipGetLastError());
    ZK_HIP_TRY(hipMemcpyAsync(h_out, d, 8, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    return ZK_OK;
}

}  // namespace

int quad_selftest_dev(zk_ctx* c, int curve, uint32_t n_quads, uint32_t* out2) {
    if (curve == ZK_CURVE_BLS12_381) return run<CurveBls>(c, n_quads, out2);
    if (curve == ZK_CURVE_BN254) return run<CurveBn>(c, n_quads, out2);
    return ZK_ERR_BAD_ARG;
}
