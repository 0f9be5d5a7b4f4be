// KZG10 opening helpers: random linear combination of polynomials and the
// witness polynomial (p(X) - p(z)) / (X - z), on the device -- and, with the same pieces, the two O(n) operations of the
// prover's last round (linearisation_poly.rs:164-350): `poly.evaluate(&point)` for the 23 evaluations of a proof
// (zk_poly_evaluate_dev) and the scalar-weighted sum of ~19 polynomials that is the linearisation polynomial
// (zk_poly_lincomb_dev).
//
// Replaces the CPU work ark-poly-commit 0.3 does inside `PolynomialCommitment::open` before its MSM
// (reference call sites: proof_system/prover.rs:582-591 -- 11 polynomials at z -- and :609-618 --
// 7 polynomials at z*w).  SonicKZG10::open without degree bounds or hiding:
//     p(X) = sum_k chi^k p_k(X);   w(X) = (p(X) - p(z)) / (X - z);   proof = commit(w).
// Synthetic division is the recurrence w[i-1] = p[i] + z*w[i]; here it runs as a three-phase
// Horner scan (chunk sums -> workgroup suffix scan with the operator (h,q)o(h',q') = (h+q h', q q')
// -> per-chunk replay), 3 field multiplications per coefficient, output already in canonical
// (into_repr) form for the opening MSM.
#include "ctx.h"
#include "fr_io.cuh"

// Arithmetic: the 29-bit-limb Montgomery type of fieldu.cuh (as in the NTT).  Data stay in the arkworks
// Montgomery domain (R = 2^256): every constant multiplier (chi^k, z, z^CHUNK and the running powers of the
// scan) is held in the R' = 2^261 form, so data * multiplier / R' keeps the factor R; chunk sums and chunk
// carries are kept as lazily reduced limb vectors (48 B) between the phases.

namespace {

constexpr int MAX_POLYS = 32;
struct RlcArgs {
    const void* poly[MAX_POLYS];
    uint64_t len[MAX_POLYS];
    Packed chi_pow[MAX_POLYS];   // chi^k in the R' form
    uint32_t n_polys;
};

// comb[i] = sum_k chi^k * p_k[i]
template <class FU>
__global__ void kzg_rlc(RlcArgs a, uint64_t m, void* comb) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    FU acc = FU::zero();                     // <= 32 terms < 2r each
    for (uint32_t k = 0; k < a.n_polys; ++k) {
        if (i < a.len[k]) acc = FU::add(acc, FU::mul(ld_u<FU>(a.poly[k], i), unpack<FU>(a.chi_pow[k])));
    }
    st_u<FU>(comb, i, FU::mul(acc, FU::one()));   // * 1 (R' form): back under 2r
}

constexpr uint32_t CHUNK = 64;       // coefficients per lane in phases 1 and 3
constexpr uint32_t SCAN_T = 1024;    // lanes of the single scan workgroup

// phase 1: H[t] = sum_{j < CHUNK} c[t*CHUNK + j] z^j          (lazy, < 3r)
template <class FU>
__global__ void kzg_chunk_horner(const void* comb, uint64_t m, Packed zp, void* H, uint64_t n_chunks) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    const FU z = unpack<FU>(zp);
    const uint64_t lo = t * CHUNK;
    const uint64_t hi = lo + CHUNK < m ? lo + CHUNK : m;
    FU acc = FU::zero();
    for (uint64_t j = hi; j-- > lo;) acc = FU::add(ld_u<FU>(comb, j), FU::mul(acc, z));
    st_l<FU>(H, t, acc);
}

// phase 2: A[t] = sum_{t' > t} H[t'] (z^CHUNK)^(t'-t-1), the value flowing into chunk t.  SCAN_T lanes own n_chunks / SCAN_T
// consecutive chunks each.  Three launches: (a) every lane folds its chunks into (h, q), (b) ONE workgroup scans the SCAN_T
// pairs, (c) every lane replays its chunks from the value entering it.  (a) and (c) are the lanes' own loops and run as 16
// workgroups of one wavefront each -- alone on a SIMD a wavefront issues twice as often as it does sharing it with the three
// others of a 1024-lane workgroup on one CU, which is what the whole phase used to be (0.19 ms per opening with the rest of the
// chip idle).  Same operations on the same operands in the same order.
template <class FU>
ZK_D void scan_range(uint32_t u, uint64_t n_chunks, uint64_t& lo, uint64_t& hi) {
    const uint64_t per = (n_chunks + SCAN_T - 1) / SCAN_T;
    lo = (uint64_t)u * per < n_chunks ? (uint64_t)u * per : n_chunks;
    hi = lo + per < n_chunks ? lo + per : n_chunks;
}
template <class FU>
__global__ void __launch_bounds__(64) kzg_scan_local(const void* H, uint64_t n_chunks, Packed zkp /* z^CHUNK, R' form */, void* HQ) {
    const uint32_t u = blockIdx.x * 64 + threadIdx.x;
    const FU zk = unpack<FU>(zkp);
    uint64_t lo, hi;
    scan_range<FU>(u, n_chunks, lo, hi);
    // local: h = sum_{t in [lo,hi)} H[t] zk^(t-lo) (data, < 5r),  q = zk^(hi-lo) (multiplier, R' form, < 2r)
    FU h = FU::zero(), q = FU::one();
    for (uint64_t t = hi; t-- > lo;) {
        h = FU::add(ld_l<FU>(H, t), FU::mul(h, zk));
        q = FU::mul(q, zk);
    }
    st_l<FU>(HQ, 2 * u, h);
    st_l<FU>(HQ, 2 * u + 1, q);
}
template <class FU>
__global__ void __launch_bounds__(SCAN_T) kzg_scan_cross(const void* HQ, void* carry_out) {
    extern __shared__ uint4 sh[];          // (h, q) per lane: 2 x 48 B
    const uint32_t u = threadIdx.x;
    FU h = ld_l<FU>(HQ, 2 * u), q = ld_l<FU>(HQ, 2 * u + 1);
    // exclusive suffix scan of (h, q) with (h1,q1) o (h2,q2) = (h1 + q1 h2, q1 q2): Hillis-Steele.
    // h grows by < 2r per step (< 25r after 10 steps); q1 * h2 < 2r * 25r stays inside the product's range.
    for (uint32_t d = 1; d < SCAN_T; d <<= 1) {
        st_l<FU>(sh, 2 * u, h);
        st_l<FU>(sh, 2 * u + 1, q);
        __syncthreads();
        if (u + d < SCAN_T) {
            const FU oh = ld_l<FU>(sh, 2 * (u + d)), oq = ld_l<FU>(sh, 2 * (u + d) + 1);
            h = FU::add(h, FU::mul(q, oh));
            q = FU::mul(q, oq);
        }
        __syncthreads();
    }
    // h = inclusive suffix value starting at this lane's first chunk; the value entering the lane
    // from above is the inclusive value of lane u+1
    st_l<FU>(sh, 2 * u, h);
    __syncthreads();
    st_l<FU>(carry_out, u, (u + 1 < SCAN_T) ? ld_l<FU>(sh, 2 * (u + 1)) : FU::zero());
}
template <class FU>
__global__ void __launch_bounds__(64) kzg_scan_replay(const void* H, uint64_t n_chunks, Packed zkp, const void* carry_in, void* A) {
    const uint32_t u = blockIdx.x * 64 + threadIdx.x;
    const FU zk = unpack<FU>(zkp);
    uint64_t lo, hi;
    scan_range<FU>(u, n_chunks, lo, hi);
    FU carry = ld_l<FU>(carry_in, u);
    // replay the lane's chunks from the top to hand every chunk its incoming value
    for (uint64_t t = hi; t-- > lo;) {
        st_l<FU>(A, t, carry);
        carry = FU::add(ld_l<FU>(H, t), FU::mul(carry, zk));
    }
}

// phase 3: replay each chunk with its incoming value; w[i-1] = c[i] + z*w[i], written canonical (into_repr)
template <class FU>
__global__ void kzg_witness(const void* comb, uint64_t m, Packed zp, const void* A, void* w_out, uint64_t n_chunks) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    const FU z = unpack<FU>(zp);
    FU unR = FU::zero();      // the plain integer R'/R = 2^5: (x R) * 2^5 / R' = x
    unR.v[0] = 32u;
    const uint64_t lo = t * CHUNK;
    const uint64_t hi = lo + CHUNK < m ? lo + CHUNK : m;
    FU run = ld_l<FU>(A, t);
    for (uint64_t j = hi; j-- > lo;) {
        run = FU::add(ld_u<FU>(comb, j), FU::mul(run, z));   // = w[j-1] (Montgomery, lazy)
        if (j >= 1) st_u<FU>(w_out, j - 1, FU::mul(run, unR));
    }
}

// ---- evaluations: out[y] = p_y(z_y) for n_polys polynomials in one launch pair ----------------------------------------
struct EvalArgs {
    const void* poly[MAX_POLYS];
    uint64_t len[MAX_POLYS];
    Packed z[MAX_POLYS];         // the point of polynomial y, R' form
    Packed zk[MAX_POLYS];        // z^CHUNK, R' form
    uint32_t n_polys;
    uint64_t max_chunks;
};
constexpr uint32_t FOLD_T = 256;

// phase 1 (grid: chunks x polys): H[y][t] = sum_{j < CHUNK} c_y[t*CHUNK + j] z_y^j
template <class FU>
__global__ void poly_chunk_horner(EvalArgs a, void* H) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    const uint64_t m = a.len[y];
    if (t * CHUNK >= m) return;
    const FU z = unpack<FU>(a.z[y]);
    const uint64_t lo = t * CHUNK;
    const uint64_t hi = lo + CHUNK < m ? lo + CHUNK : m;
    FU acc = FU::zero();
    for (uint64_t j = hi; j-- > lo;) acc = FU::add(ld_u<FU>(a.poly[y], j), FU::mul(acc, z));
    st_l<FU>(H, (uint64_t)y * a.max_chunks + t, acc);
}

// phase 2 (one workgroup per polynomial): p(z) = sum_t H[t] (z^CHUNK)^t.  Lane u folds its `per` consecutive chunks by
// Horner, scales the result by (zk^per)^u (square and multiply on the lane index) and the workgroup adds the lanes up.
template <class FU>
__global__ void __launch_bounds__(FOLD_T) poly_chunk_fold(EvalArgs a, const void* H, void* out) {
    __shared__ uint4 sh[FOLD_T * 3];
    const uint32_t u = threadIdx.x, y = blockIdx.x;
    const uint64_t n_chunks = (a.len[y] + CHUNK - 1) / CHUNK;
    const uint64_t per = (n_chunks + FOLD_T - 1) / FOLD_T;
    const uint64_t lo = (uint64_t)u * per < n_chunks ? (uint64_t)u * per : n_chunks;
    const uint64_t hi = lo + per < n_chunks ? lo + per : n_chunks;
    const FU zk = unpack<FU>(a.zk[y]);
    FU h = FU::zero();
    for (uint64_t t = hi; t-- > lo;) h = FU::add(ld_l<FU>(H, (uint64_t)y * a.max_chunks + t), FU::mul(h, zk));
    // Q = zk^per (the same on every lane), then Q^u
    FU Q = FU::one(), b = zk;
    for (uint64_t e = per; e; e >>= 1) {
        if (e & 1) Q = FU::mul(Q, b);
        b = FU::mul(b, b);
    }
    FU s = FU::one();
    b = Q;
    for (uint32_t e = u; e; e >>= 1) {
        if (e & 1) s = FU::mul(s, b);
        b = FU::mul(b, b);
    }
    h = FU::mul(h, s);                      // < 2r
    // tree sum; every level re-normalises (x * 1 in the R' form keeps the value, brings it under 2r)
    st_l<FU>(sh, u, h);
    for (uint32_t d = FOLD_T / 2; d >= 1; d >>= 1) {
        __syncthreads();
        if (u < d) {
            h = FU::mul(FU::add(h, ld_l<FU>(sh, u + d)), FU::one());
            st_l<FU>(sh, u, h);
        }
    }
    if (u == 0) st_u<FU>(out, y, h);
}

template <class C>
int poly_evaluate(zk_ctx* c, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* points_mont, uint64_t* out_mont) {
    typedef typename C::Fr Fr;
    typedef typename C::FrU FU;
    if (n_polys > (uint32_t)MAX_POLYS) return ZK_ERR_UNSUPPORTED;
    if (n_polys == 0) return ZK_OK;
    Fr to_rp;
    FU::one().pack_words(to_rp.v);
    EvalArgs a;
    memset(&a, 0, sizeof a);
    a.n_polys = n_polys;
    uint64_t m = 0;
    for (uint32_t k = 0; k < n_polys; ++k) {
        if (lens[k] && !d_polys[k]) return ZK_ERR_BAD_ARG;
        Fr z;
        memcpy(z.v, points_mont + 4 * k, 32);
        if (Fr::reduce_once(z) != z) return ZK_ERR_BAD_ARG;      // not a reduced field element
        const Fr zr = Fr::mul(z, to_rp), zkr = Fr::mul(Fr::pow_u64(z, CHUNK), to_rp);
        memcpy(a.z[k].w, zr.v, 32);
        memcpy(a.zk[k].w, zkr.v, 32);
        a.poly[k] = d_polys[k];
        a.len[k] = lens[k];
        m = lens[k] > m ? lens[k] : m;
    }
    a.max_chunks = (m + CHUNK - 1) / CHUNK;
    int rc;
    if ((rc = c->io_b.ensure((size_t)n_polys * (a.max_chunks + 1) * 48 + (size_t)n_polys * 32))) return rc;
    void* H = c->io_b.p;
    void* d_out = (char*)c->io_b.p + (size_t)n_polys * (a.max_chunks + 1) * 48;
    hipStream_t st = c->stream;
    {
        ProfScope ps(c, "poly_evaluate");
        const int T = 128;
        if (a.max_chunks)
            hipLaunchKernelGGL(poly_chunk_horner<FU>, dim3((unsigned)((a.max_chunks + T - 1) / T), n_polys), dim3(T), 0, st, a, H);
        hipLaunchKernelGGL(poly_chunk_fold<FU>, dim3(n_polys), dim3(FOLD_T), 0, st, a, (const void*)H, d_out);
        ZK_HIP_TRY(hipGetLastError());
    }
    ZK_HIP_TRY(hipMemcpyAsync(out_mont, d_out, (size_t)n_polys * 32, hipMemcpyDeviceToHost, st));
    ZK_HIP_TRY(hipStreamSynchronize(st));
    c->d2h_bytes += (uint64_t)n_polys * 32;
    return ZK_OK;
}

template <class C>
int poly_lincomb(zk_ctx* c, uint32_t n_terms, const void* const* d_polys, const size_t* lens, const uint64_t* coeffs_mont, void* d_out, size_t out_len) {
    typedef typename C::Fr Fr;
    typedef typename C::FrU FU;
    if (n_terms > (uint32_t)MAX_POLYS) return ZK_ERR_UNSUPPORTED;
    if (out_len == 0) return ZK_OK;
    Fr to_rp;
    FU::one().pack_words(to_rp.v);
    RlcArgs a;
    memset(&a, 0, sizeof a);
    a.n_polys = n_terms;
    for (uint32_t k = 0; k < n_terms; ++k) {
        if (lens[k] && !d_polys[k]) return ZK_ERR_BAD_ARG;
        Fr cf;
        memcpy(cf.v, coeffs_mont + 4 * k, 32);
        if (Fr::reduce_once(cf) != cf) return ZK_ERR_BAD_ARG;
        const Fr r = Fr::mul(cf, to_rp);
        memcpy(a.chi_pow[k].w, r.v, 32);
        a.poly[k] = d_polys[k];
        a.len[k] = lens[k] < out_len ? lens[k] : out_len;
    }
    ProfScope ps(c, "poly_lincomb");
    const int T = 256;
    hipLaunchKernelGGL(kzg_rlc<FU>, dim3((unsigned)((out_len + T - 1) / T)), dim3(T), 0, c->stream, a, (uint64_t)out_len, d_out);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

template <class C>
int open_prepare(zk_ctx* c, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* z_mont,
                 const uint64_t* chal_mont, void** d_w, size_t* wlen) {
    typedef typename C::Fr Fr;
    typedef typename C::FrU FU;
    if (n_polys > (uint32_t)MAX_POLYS) return ZK_ERR_UNSUPPORTED;
    uint64_t m = 0;
    for (uint32_t k = 0; k < n_polys; ++k) m = lens[k] > m ? lens[k] : m;
    *wlen = m > 0 ? m - 1 : 0;
    *d_w = nullptr;
    if (m <= 1) return ZK_OK;
    Fr z, chi;
    memcpy(z.v, z_mont, 32);
    memcpy(chi.v, chal_mont, 32);
    // R' mod r as a plain integer: x R -> x R' by one product in the arkworks domain (as ntt.hip's tables)
    Fr to_rp;
    FU::one().pack_words(to_rp.v);
    auto pack = [&](const Fr& v_mont) {
        Packed p;
        const Fr rp = Fr::mul(v_mont, to_rp);
        memcpy(p.w, rp.v, 32);
        return p;
    };
    RlcArgs a;
    memset(&a, 0, sizeof a);
    a.n_polys = n_polys;
    Fr pw = Fr::one();
    for (uint32_t k = 0; k < n_polys; ++k) {
        a.poly[k] = d_polys[k];
        a.len[k] = lens[k];
        a.chi_pow[k] = pack(pw);
        pw = Fr::mul(pw, chi);
    }
    const uint64_t n_chunks = (m + CHUNK - 1) / CHUNK;
    int rc;
    if ((rc = c->io_a.ensure(m * 32))) return rc;                       // comb
    if ((rc = c->io_b.ensure((n_chunks * 2 + (size_t)SCAN_T * 3) * 48))) return rc;   // H | A | the scan's (h, q) pairs and carries (limb vectors)
    if ((rc = c->witness.ensure(m * 32))) return rc;                      // witness, canonical
    void* comb = c->io_a.p;
    void* H = c->io_b.p;
    void* A = (char*)c->io_b.p + n_chunks * 48;
    const Packed zp = pack(z), zkp = pack(Fr::pow_u64(z, CHUNK));
    hipStream_t st = c->stream;
    ProfScope ps(c, "kzg_open_prep");
    const int T = 256;
    hipLaunchKernelGGL(kzg_rlc<FU>, dim3((unsigned)((m + T - 1) / T)), dim3(T), 0, st, a, m, comb);
    hipLaunchKernelGGL(kzg_chunk_horner<FU>, dim3((unsigned)((n_chunks + T - 1) / T)), dim3(T), 0, st, comb, m, zp, H, n_chunks);
    void* HQ = (char*)c->io_b.p + n_chunks * 2 * 48;
    void* CR = (char*)HQ + (size_t)SCAN_T * 2 * 48;
    size_t shmem = (size_t)SCAN_T * 2 * 48;
    ZK_HIP_TRY(hipFuncSetAttribute((const void*)kzg_scan_cross<FU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kzg_scan_local<FU>, dim3(SCAN_T / 64), dim3(64), 0, st, H, n_chunks, zkp, HQ);
    hipLaunchKernelGGL(kzg_scan_cross<FU>, dim3(1), dim3(SCAN_T), shmem, st, HQ, CR);
    hipLaunchKernelGGL(kzg_scan_replay<FU>, dim3(SCAN_T / 64), dim3(64), 0, st, H, n_chunks, zkp, CR, A);
    hipLaunchKernelGGL(kzg_witness<FU>, dim3((unsigned)((n_chunks + T - 1) / T)), dim3(T), 0, st, comb, m, zp, A, c->witness.p, n_chunks);
    ZK_HIP_TRY(hipGetLastError());
    *d_w = c->witness.p;
    return ZK_OK;
}

}  // namespace

int poly_evaluate_dev(zk_ctx* c, int curve, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* points_mont,
                      uint64_t* out_mont) {
    if (curve == ZK_CURVE_BLS12_381) return poly_evaluate<CurveBls>(c, n_polys, d_polys, lens, points_mont, out_mont);
    if (curve == ZK_CURVE_BN254) return poly_evaluate<CurveBn>(c, n_polys, d_polys, lens, points_mont, out_mont);
    return ZK_ERR_BAD_ARG;
}

int poly_lincomb_dev(zk_ctx* c, int curve, uint32_t n_terms, const void* const* d_polys, const size_t* lens, const uint64_t* coeffs_mont,
                     void* d_out, size_t out_len) {
    if (curve == ZK_CURVE_BLS12_381) return poly_lincomb<CurveBls>(c, n_terms, d_polys, lens, coeffs_mont, d_out, out_len);
    if (curve == ZK_CURVE_BN254) return poly_lincomb<CurveBn>(c, n_terms, d_polys, lens, coeffs_mont, d_out, out_len);
    return ZK_ERR_BAD_ARG;
}

int kzg_open_prepare_dev(zk_ctx* c, int curve, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                         const uint64_t* z_mont, const uint64_t* chal_mont, void** d_witness_canonical, size_t* wlen) {
    if (curve == ZK_CURVE_BLS12_381) return open_prepare<CurveBls>(c, n_polys, d_polys, lens, z_mont, chal_mont, d_witness_canonical, wlen);
    if (curve == ZK_CURVE_BN254) return open_prepare<CurveBn>(c, n_polys, d_polys, lens, z_mont, chal_mont, d_witness_canonical, wlen);
    return ZK_ERR_BAD_ARG;
}
