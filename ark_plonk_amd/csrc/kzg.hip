// KZG10 opening helpers: random linear combination of polynomials and the
// witness polynomial (p(X) - p(z)) / (X - z), on the device.
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

constexpr int MAX_POLYS = 16;
template <class Fr>
struct RlcArgs {
    const void* poly[MAX_POLYS];
    uint64_t len[MAX_POLYS];
    Fr chi_pow[MAX_POLYS];   // chi^k, Montgomery
    uint32_t n_polys;
};

// comb[i] = sum_k chi^k * p_k[i]
template <class Fr>
__global__ void kzg_rlc(RlcArgs<Fr> a, uint64_t m, void* comb) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr acc = Fr::zero();
    for (uint32_t k = 0; k < a.n_polys; ++k) {
        if (i < a.len[k]) acc = Fr::add(acc, Fr::mul(ld_fr<Fr>(a.poly[k], i), a.chi_pow[k]));
    }
    st_fr<Fr>(comb, i, acc);
}

constexpr uint32_t CHUNK = 64;       // coefficients per lane in phases 1 and 3
constexpr uint32_t SCAN_T = 1024;    // lanes of the single scan workgroup

// phase 1: H[t] = sum_{j < CHUNK} c[t*CHUNK + j] z^j
template <class Fr>
__global__ void kzg_chunk_horner(const void* comb, uint64_t m, Fr z, void* H, uint64_t n_chunks) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    const uint64_t lo = t * CHUNK;
    const uint64_t hi = lo + CHUNK < m ? lo + CHUNK : m;
    Fr acc = Fr::zero();
    for (uint64_t j = hi; j-- > lo;) acc = Fr::add(ld_fr<Fr>(comb, j), Fr::mul(acc, z));
    st_fr<Fr>(H, t, acc);
}

// phase 2 (one workgroup): A[t] = sum_{t' > t} H[t'] (z^CHUNK)^(t'-t-1), the value flowing into chunk t
template <class Fr>
__global__ void __launch_bounds__(SCAN_T) kzg_chunk_scan(const void* H, uint64_t n_chunks, Fr zk /* z^CHUNK */, void* A) {
    extern __shared__ uint4 sh[];          // (h, q) per lane: 2 x Fr
    const uint32_t u = threadIdx.x;
    const uint64_t per = (n_chunks + SCAN_T - 1) / SCAN_T;
    const uint64_t lo = (uint64_t)u * per;
    const uint64_t hi = lo + per < n_chunks ? lo + per : n_chunks;
    // local: h = sum_{t in [lo,hi)} H[t] zk^(t-lo),  q = zk^(hi-lo)
    Fr h = Fr::zero(), q = Fr::one();
    for (uint64_t t = hi; t-- > lo;) {
        h = Fr::add(ld_fr<Fr>(H, t), Fr::mul(h, zk));
        q = Fr::mul(q, zk);
    }
    // exclusive suffix scan of (h, q) with (h1,q1) o (h2,q2) = (h1 + q1 h2, q1 q2): Hillis-Steele
    Fr sh_h = h, sh_q = q;
    for (uint32_t d = 1; d < SCAN_T; d <<= 1) {
        st_fr<Fr>(sh, 2 * u, sh_h);
        st_fr<Fr>(sh, 2 * u + 1, sh_q);
        __syncthreads();
        if (u + d < SCAN_T) {
            Fr oh = ld_fr<Fr>(sh, 2 * (u + d)), oq = ld_fr<Fr>(sh, 2 * (u + d) + 1);
            sh_h = Fr::add(sh_h, Fr::mul(sh_q, oh));
            sh_q = Fr::mul(sh_q, oq);
        }
        __syncthreads();
    }
    // sh_h = inclusive suffix value starting at this lane's first chunk; the value entering the lane
    // from above is the inclusive value of lane u+1
    st_fr<Fr>(sh, 2 * u, sh_h);
    __syncthreads();
    Fr carry = (u + 1 < SCAN_T) ? ld_fr<Fr>(sh, 2 * (u + 1)) : Fr::zero();
    // replay the lane's chunks from the top to hand every chunk its incoming value
    for (uint64_t t = hi; t-- > lo;) {
        st_fr<Fr>(A, t, carry);
        carry = Fr::add(ld_fr<Fr>(H, t), Fr::mul(carry, zk));
    }
}

// phase 3: replay each chunk with its incoming value; w[i-1] = c[i] + z*w[i], written canonical
template <class Fr>
__global__ void kzg_witness(const void* comb, uint64_t m, Fr z, const void* A, void* w_out, uint64_t n_chunks) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    const uint64_t lo = t * CHUNK;
    const uint64_t hi = lo + CHUNK < m ? lo + CHUNK : m;
    Fr run = ld_fr<Fr>(A, t);
    for (uint64_t j = hi; j-- > lo;) {
        run = Fr::add(ld_fr<Fr>(comb, j), Fr::mul(run, z));   // = w[j-1]
        if (j >= 1) st_fr<Fr>(w_out, j - 1, Fr::from_mont(run));
    }
}

template <class C>
int open_prepare(zk_ctx* c, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* z_mont,
                 const uint64_t* chal_mont, void** d_w, size_t* wlen) {
    typedef typename C::Fr Fr;
    if (n_polys > (uint32_t)MAX_POLYS) return ZK_ERR_UNSUPPORTED;
    uint64_t m = 0;
    for (uint32_t k = 0; k < n_polys; ++k) m = lens[k] > m ? lens[k] : m;
    *wlen = m > 0 ? m - 1 : 0;
    *d_w = nullptr;
    if (m <= 1) return ZK_OK;
    Fr z, chi;
    memcpy(z.v, z_mont, 32);
    memcpy(chi.v, chal_mont, 32);
    RlcArgs<Fr> a;
    memset(&a, 0, sizeof a);
    a.n_polys = n_polys;
    Fr pw = Fr::one();
    for (uint32_t k = 0; k < n_polys; ++k) {
        a.poly[k] = d_polys[k];
        a.len[k] = lens[k];
        a.chi_pow[k] = pw;
        pw = Fr::mul(pw, chi);
    }
    const uint64_t n_chunks = (m + CHUNK - 1) / CHUNK;
    int rc;
    if ((rc = c->io_a.ensure(m * 32))) return rc;                       // comb
    if ((rc = c->io_b.ensure(n_chunks * 32 * 2))) return rc;            // H | A
    if ((rc = c->mb[0].scalars.ensure(m * 32))) return rc;                // witness, canonical
    void* comb = c->io_a.p;
    void* H = c->io_b.p;
    void* A = (char*)c->io_b.p + n_chunks * 32;
    Fr zk = Fr::pow_u64(z, CHUNK);
    hipStream_t st = c->stream;
    ProfScope ps(c, "kzg_open_prep");
    const int T = 256;
    hipLaunchKernelGGL(kzg_rlc<Fr>, dim3((unsigned)((m + T - 1) / T)), dim3(T), 0, st, a, m, comb);
    hipLaunchKernelGGL(kzg_chunk_horner<Fr>, dim3((unsigned)((n_chunks + T - 1) / T)), dim3(T), 0, st, comb, m, z, H, n_chunks);
    size_t shmem = (size_t)SCAN_T * 2 * 32;
    ZK_HIP_TRY(hipFuncSetAttribute((const void*)kzg_chunk_scan<Fr>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kzg_chunk_scan<Fr>, dim3(1), dim3(SCAN_T), shmem, st, H, n_chunks, zk, A);
    hipLaunchKernelGGL(kzg_witness<Fr>, dim3((unsigned)((n_chunks + T - 1) / T)), dim3(T), 0, st, comb, m, z, A, c->mb[0].scalars.p, n_chunks);
    ZK_HIP_TRY(hipGetLastError());
    *d_w = c->mb[0].scalars.p;
    return ZK_OK;
}

}  // namespace

int kzg_open_prepare_dev(zk_ctx* c, int curve, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                         const uint64_t* z_mont, const uint64_t* chal_mont, void** d_witness_canonical, size_t* wlen) {
    if (curve == ZK_CURVE_BLS12_381) return open_prepare<CurveBls>(c, n_polys, d_polys, lens, z_mont, chal_mont, d_witness_canonical, wlen);
    if (curve == ZK_CURVE_BN254) return open_prepare<CurveBn>(c, n_polys, d_polys, lens, z_mont, chal_mont, d_witness_canonical, wlen);
    return ZK_ERR_BAD_ARG;
}
