// Grand-product builders: the evaluation vectors of the permutation polynomial
// z(X) and the lookup polynomial z2(X), on the device, feeding the iNTT directly.
//
// Reference: plonk-core/src/permutation/mod.rs:652-752 (`compute_permutation_poly`) and :754-822
// (`compute_lookup_permutation_poly` / `lookup_ratio`), both up to -- not including -- their final
// `domain.ifft`.  The reference walks the rows serially, inverting one denominator per row:
//     z[0] = 1,   z[i+1] = z[i] * num_i / den_i,   the (n+1)-th value dropped.
// Here, with PN_i = prod_{j<i} num_j, SD_i = prod_{j>=i} den_j and T = prod_j den_j:
//     z[i] = PN_i / prod_{j<i} den_j = PN_i * SD_i * T^-1
// i.e. two product scans (prefix of num, suffix of den: wave-level scans of 256 positions -> one-workgroup scan of the
// tile totals), ONE field inversion on the host, one elementwise product that also applies the tile carries.
// Field elements are canonical Montgomery residues, so the result is bit-identical with the serial loop.
// A zero denominator makes the reference panic (`inverse().unwrap()`); here it is ZK_ERR_NOT_INVERTIBLE.
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

constexpr uint32_t TERM_T = 256;     // lanes per workgroup of the term kernels
constexpr uint32_t TERM_ROWS = 8;    // rows per lane (row = blk*T*ROWS + j*T + lane: coalesced)

template <class Fr>
struct PermArgs {
    const void* w[4];
    const void* s[4];
    Fr beta, gamma;
    Fr bk[4];        // beta * K_k, K = (1, 7, 13, 17)  (permutation/constants.rs:12-22)
    Fr omega;        // group generator of the size-n domain
    Fr omega_t;      // omega^TERM_T
};

// num_i = prod_k (w_k[i] + beta*K_k*omega^i + gamma),  den_i = prod_k (w_k[i] + beta*sigma_k[i] + gamma)
// (numerator_irreducible / denominator_irreducible, permutation/mod.rs:626-647)
template <class Fr>
__global__ void __launch_bounds__(TERM_T) gp_perm_terms(PermArgs<Fr> a, uint64_t n, void* N, void* D) {
    const uint64_t base = (uint64_t)blockIdx.x * TERM_T * TERM_ROWS + threadIdx.x;
    if (base >= n) return;
    Fr root = Fr::pow_u64(a.omega, base);
    for (uint32_t j = 0; j < TERM_ROWS; ++j) {
        const uint64_t i = base + (uint64_t)j * TERM_T;
        if (i >= n) break;
        Fr num, den;
        for (int k = 0; k < 4; ++k) {
            Fr wg = Fr::add(ld_fr<Fr>(a.w[k], i), a.gamma);
            Fr nk = Fr::add(wg, Fr::mul(a.bk[k], root));
            Fr dk = Fr::add(wg, Fr::mul(a.beta, ld_fr<Fr>(a.s[k], i)));
            num = k == 0 ? nk : Fr::mul(num, nk);
            den = k == 0 ? dk : Fr::mul(den, dk);
        }
        st_fr<Fr>(N, i, num);
        st_fr<Fr>(D, i, den);
        root = Fr::mul(root, a.omega_t);
    }
}

template <class Fr>
struct LookupArgs {
    const void *f, *t, *h1, *h2;
    Fr delta, one_plus_delta, eps, eps_opd;   // eps_opd = epsilon * (1 + delta)
};

// lookup_ratio (permutation/mod.rs:802-822), numerator and denominator kept apart:
//   num_i = (1+d)(e + f_i)(e(1+d) + t_i + d t_{i+1}),  den_i = (e(1+d) + h1_i + d h2_i)(e(1+d) + h2_i + d h1_{i+1})
// with the cyclic "next" of mod.rs:771-772.
template <class Fr>
__global__ void __launch_bounds__(TERM_T) gp_lookup_terms(LookupArgs<Fr> a, uint64_t n, void* N, void* D) {
    const uint64_t i = (uint64_t)blockIdx.x * TERM_T + threadIdx.x;
    if (i >= n) return;
    const uint64_t nx = i + 1 == n ? 0 : i + 1;
    Fr f = ld_fr<Fr>(a.f, i), t = ld_fr<Fr>(a.t, i), tn = ld_fr<Fr>(a.t, nx);
    Fr h1 = ld_fr<Fr>(a.h1, i), h1n = ld_fr<Fr>(a.h1, nx), h2 = ld_fr<Fr>(a.h2, i);
    Fr num = Fr::mul(Fr::mul(a.one_plus_delta, Fr::add(a.eps, f)), Fr::add(Fr::add(a.eps_opd, t), Fr::mul(a.delta, tn)));
    Fr den = Fr::mul(Fr::add(Fr::add(a.eps_opd, h1), Fr::mul(h2, a.delta)), Fr::add(Fr::add(a.eps_opd, h2), Fr::mul(h1n, a.delta)));
    st_fr<Fr>(N, i, num);
    st_fr<Fr>(D, i, den);
}

// Round 2: the scans are latency-bound (a few hundred wavefronts walking dependent products), so the dependent chain is what
// to cut.  Numerators need exclusive PREFIX products, denominators inclusive SUFFIX products; a suffix over i is a prefix over the
// mirrored position n-1-i, so both run the same code, blockIdx.y = 0 numerators, 1 denominators:
//   gp_wave_scan    a wavefront owns 256 logical positions (4 per lane): 3 serial products per lane, a 6-step shuffle scan of
//                   the lane totals, 4 products to place the results -- 13 dependent products -- and its total goes to P[tile]
//   gp_scan_tiles   one workgroup per array scans the tile totals (prefix)
//   gp_combine      z[i] = (carryN * locN_i) * (carryD * locD_i) * T^-1, elementwise
// 13 + ~18 + 4 dependent products instead of the 64 + 42 + 64 of the chunked version of round 1.
constexpr uint32_t WS_PER = 4;                 // positions per lane
constexpr uint32_t WS_TILE = 64 * WS_PER;      // positions per wavefront
constexpr uint32_t SCAN_T = 1024;              // lanes of the tile-total scan workgroup

struct GpPair {
    void* x[2];        // numerators / denominators, scanned in place
    void* p[2];        // tile totals
    void* a[2];        // tile carries (+ total)
};

template <class Fr>
ZK_D Fr shfl_up_fr(const Fr& v, int d) {
    Fr r;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) r.v[i] = __shfl_up(v.v[i], d, 64);
    return r;
}

template <class Fr>
__global__ void __launch_bounds__(256) gp_wave_scan(GpPair g, uint64_t n, uint64_t n_tiles) {
    const uint32_t mirrored = blockIdx.y;                       // 1: logical position q <-> element n-1-q (suffix products)
    void* X = g.x[mirrored];
    void* P = g.p[mirrored];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;                                // the whole wavefront leaves together
    const uint64_t q0 = tile * WS_TILE + (uint64_t)lane * WS_PER;
    const Fr one = Fr::one();
    Fr inc[WS_PER];                                             // inclusive products inside the lane
#pragma unroll
    for (uint32_t k = 0; k < WS_PER; ++k) {
        const uint64_t q = q0 + k;
        const Fr e = q < n ? ld_fr<Fr>(X, mirrored ? n - 1 - q : q) : one;
        inc[k] = k == 0 ? e : Fr::mul(inc[k - 1], e);
    }
    Fr scan = inc[WS_PER - 1];                                  // inclusive scan of the lane totals over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const Fr o = shfl_up_fr<Fr>(scan, d);
        if ((int)lane >= d) scan = Fr::mul(scan, o);
    }
    Fr excl = shfl_up_fr<Fr>(scan, 1);                          // product of the lanes before this one
    if (lane == 0) excl = one;
#pragma unroll
    for (uint32_t k = 0; k < WS_PER; ++k) {
        const uint64_t q = q0 + k;
        if (q >= n) break;
        // numerators: exclusive (positions before q); denominators: inclusive (mirrored positions up to q = elements from i on)
        const Fr v = mirrored ? Fr::mul(excl, inc[k]) : (k == 0 ? excl : Fr::mul(excl, inc[k - 1]));
        st_fr<Fr>(X, mirrored ? n - 1 - q : q, v);
    }
    if (lane == 63) st_fr<Fr>(P, tile, scan);
}

// one workgroup per array: A[c] = product of the tile totals before c;  A[n_tiles] = product of everything
template <class Fr>
__global__ void __launch_bounds__(SCAN_T) gp_scan_tiles(GpPair g, uint64_t n_tiles) {
    extern __shared__ uint4 sh[];
    const void* P = g.p[blockIdx.x];
    void* A = g.a[blockIdx.x];
    const uint32_t u = threadIdx.x;
    const uint64_t per = (n_tiles + SCAN_T - 1) / SCAN_T;
    const uint64_t lo = (uint64_t)u * per < n_tiles ? (uint64_t)u * per : n_tiles;
    const uint64_t hi = lo + per < n_tiles ? lo + per : n_tiles;
    Fr loc = Fr::one();
    for (uint64_t c = lo; c < hi; ++c) loc = Fr::mul(loc, ld_fr<Fr>(P, c));
    Fr inc = loc;                                               // inclusive Hillis-Steele scan of the lane products
    for (uint32_t d = 1; d < SCAN_T; d <<= 1) {
        st_fr<Fr>(sh, u, inc);
        __syncthreads();
        if (u >= d) inc = Fr::mul(inc, ld_fr<Fr>(sh, u - d));
        __syncthreads();
    }
    st_fr<Fr>(sh, u, inc);
    __syncthreads();
    Fr carry = u >= 1 ? ld_fr<Fr>(sh, u - 1) : Fr::one();
    for (uint64_t c = lo; c < hi; ++c) {
        st_fr<Fr>(A, c, carry);
        carry = Fr::mul(carry, ld_fr<Fr>(P, c));
    }
    if (u == SCAN_T - 1) st_fr<Fr>(A, n_tiles, inc);
}

// z[i] = PN_i * SD_i * T^-1 with PN_i = carryN[tile(i)] * locN_i and SD_i = carryD[tile(n-1-i)] * locD_i
template <class Fr>
__global__ void gp_combine(GpPair g, Fr inv_t, uint64_t n, void* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr pn = Fr::mul(ld_fr<Fr>(g.a[0], i / WS_TILE), ld_fr<Fr>(g.x[0], i));
    const Fr sd = Fr::mul(ld_fr<Fr>(g.a[1], (n - 1 - i) / WS_TILE), ld_fr<Fr>(g.x[1], i));
    st_fr<Fr>(out, i, Fr::mul(Fr::mul(pn, sd), inv_t));
}

// N (numerators) and D (denominators) are in c->io_a / c->io_b; finishes the product into d_out
template <class Fr>
int finish_product(zk_ctx* c, uint64_t n, void* d_out, uint64_t* last_mont) {
    const uint64_t n_tiles = (n + WS_TILE - 1) / WS_TILE;
    int rc;
    // tile totals | tile carries (+ total), for N then D
    if ((rc = c->msm_tmp.ensure(4 * (n_tiles + 1) * 32))) return rc;
    char* t = (char*)c->msm_tmp.p;
    GpPair g;
    g.x[0] = c->io_a.p;
    g.x[1] = c->io_b.p;
    g.p[0] = t;
    g.a[0] = t + (n_tiles + 1) * 32;
    g.p[1] = t + 2 * (n_tiles + 1) * 32;
    g.a[1] = t + 3 * (n_tiles + 1) * 32;
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(gp_wave_scan<Fr>, dim3((unsigned)((n_tiles + 3) / 4), 2), dim3(256), 0, st, g, n, n_tiles);
    hipLaunchKernelGGL(gp_scan_tiles<Fr>, dim3(2), dim3(SCAN_T), (size_t)SCAN_T * 32, st, g, n_tiles);
    ZK_HIP_TRY(hipGetLastError());
    Fr tot[2];   // total numerator product, total denominator product
    ZK_HIP_TRY(hipMemcpyAsync(&tot[0], (char*)g.a[0] + n_tiles * 32, 32, hipMemcpyDeviceToHost, st));
    ZK_HIP_TRY(hipMemcpyAsync(&tot[1], (char*)g.a[1] + n_tiles * 32, 32, hipMemcpyDeviceToHost, st));
    ZK_HIP_TRY(hipStreamSynchronize(st));
    if (tot[1].is_zero()) return ZK_ERR_NOT_INVERTIBLE;
    Fr inv_t = Fr::inverse(tot[1]);
    const int T = 256;
    hipLaunchKernelGGL(gp_combine<Fr>, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, st, g, inv_t, n, d_out);
    ZK_HIP_TRY(hipGetLastError());
    if (last_mont) {
        Fr last = Fr::mul(tot[0], inv_t);
        memcpy(last_mont, last.v, 32);
    }
    return ZK_OK;
}

template <class C>
int perm_product(zk_ctx* c, uint32_t log_n, const void* const* d_wires, const void* const* d_sigmas, const uint64_t* beta_mont,
                 const uint64_t* gamma_mont, void* d_out, uint64_t* last_mont) {
    typedef typename C::Fr Fr;
    if (log_n > (uint32_t)C::FrP::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
    const uint64_t n = 1ull << log_n;
    int rc;
    if ((rc = c->io_a.ensure(n * 32))) return rc;
    if ((rc = c->io_b.ensure(n * 32))) return rc;
    PermArgs<Fr> a;
    memcpy(a.beta.v, beta_mont, 32);
    memcpy(a.gamma.v, gamma_mont, 32);
    const uint32_t K[4] = {1, 7, 13, 17};
    for (int k = 0; k < 4; ++k) {
        a.w[k] = d_wires[k];
        a.s[k] = d_sigmas[k];
        a.bk[k] = Fr::mul(a.beta, Fr::from_u32(K[k]));
    }
    Fr root;
    for (int i = 0; i < Fr::N; ++i) root.v[i] = C::FrP::ROOT(i);
    for (uint32_t k = log_n; k < (uint32_t)C::FrP::TWO_ADICITY; ++k) root = Fr::sqr(root);
    a.omega = root;
    a.omega_t = Fr::pow_u64(root, TERM_T);
    ProfScope ps(c, "grand_product");
    const unsigned blocks = (unsigned)((n + (uint64_t)TERM_T * TERM_ROWS - 1) / ((uint64_t)TERM_T * TERM_ROWS));
    hipLaunchKernelGGL(gp_perm_terms<Fr>, dim3(blocks), dim3(TERM_T), 0, c->stream, a, n, c->io_a.p, c->io_b.p);
    ZK_HIP_TRY(hipGetLastError());
    return finish_product<Fr>(c, n, d_out, last_mont);
}

template <class C>
int lookup_product(zk_ctx* c, size_t n, const void* d_f, const void* d_t, const void* d_h1, const void* d_h2, const uint64_t* delta_mont,
                   const uint64_t* eps_mont, void* d_out, uint64_t* last_mont) {
    typedef typename C::Fr Fr;
    int rc;
    if ((rc = c->io_a.ensure(n * 32))) return rc;
    if ((rc = c->io_b.ensure(n * 32))) return rc;
    LookupArgs<Fr> a;
    a.f = d_f;
    a.t = d_t;
    a.h1 = d_h1;
    a.h2 = d_h2;
    memcpy(a.delta.v, delta_mont, 32);
    memcpy(a.eps.v, eps_mont, 32);
    a.one_plus_delta = Fr::add(Fr::one(), a.delta);
    a.eps_opd = Fr::mul(a.eps, a.one_plus_delta);
    ProfScope ps(c, "grand_product");
    hipLaunchKernelGGL(gp_lookup_terms<Fr>, dim3((unsigned)((n + TERM_T - 1) / TERM_T)), dim3(TERM_T), 0, c->stream, a, (uint64_t)n, c->io_a.p,
                       c->io_b.p);
    ZK_HIP_TRY(hipGetLastError());
    return finish_product<Fr>(c, n, d_out, last_mont);
}

}  // namespace

int perm_product_dev(zk_ctx* c, int curve, uint32_t log_n, const void* const* d_wires, const void* const* d_sigmas,
                     const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, uint64_t* last_mont) {
    if (curve == ZK_CURVE_BLS12_381) return perm_product<CurveBls>(c, log_n, d_wires, d_sigmas, beta_mont, gamma_mont, d_out, last_mont);
    if (curve == ZK_CURVE_BN254) return perm_product<CurveBn>(c, log_n, d_wires, d_sigmas, beta_mont, gamma_mont, d_out, last_mont);
    return ZK_ERR_BAD_ARG;
}

int lookup_product_dev(zk_ctx* c, int curve, size_t n, const void* d_f, const void* d_t, const void* d_h1, const void* d_h2,
                       const uint64_t* delta_mont, const uint64_t* eps_mont, void* d_out, uint64_t* last_mont) {
    if (curve == ZK_CURVE_BLS12_381) return lookup_product<CurveBls>(c, n, d_f, d_t, d_h1, d_h2, delta_mont, eps_mont, d_out, last_mont);
    if (curve == ZK_CURVE_BN254) return lookup_product<CurveBn>(c, n, d_f, d_t, d_h1, d_h2, delta_mont, eps_mont, d_out, last_mont);
    return ZK_ERR_BAD_ARG;
}
