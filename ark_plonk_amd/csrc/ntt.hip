// Radix-2 NTT / iNTT / coset-NTT over the scalar field for gfx950 (MI355X).
//
// Replaces, behind the C ABI, ark_poly 0.3 Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}
// _in_place as called by the reference at plonk-core/src/proof_system/prover.rs:196-203,240-242,
// 281-283,302-305, quotient_poly.rs:72-120,175-177,205,294,325, permutation/mod.rs:671-674,751,800.
//
// Algorithm (not ark's): a multi-pass Cooley-Tukey decomposition N = 2^s1 * 2^s2 (* 2^s3 (* 2^s4)), s <= 9
// (one pass to 2^9, two to 2^18, three to 2^27, four above).
// Each pass gives every wavefront a tile of 2^s rows x 2^(9-s) columns = 512 elements, runs the
// 2^s-point DIT transform in registers (8 elements per lane, three radix-2 stages per window, a
// wave-private 2 KiB LDS scratch for the lane<->register transposes between windows -- no workgroup
// barriers), multiplies by the inter-pass twiddle w_M^(col*k) read from a precomputed table laid out
// like the data, and stores in place.  The last pass gathers adjacent output digits per wavefront so
// the natural-order store is contiguous (digit reversal costs no extra pass).  Zero-extension
// (in_len < N), the coset pre-scale g^j, the 1/N scale and the coset post-scale g^-j are fused into
// the first / last pass.  Arithmetic: unsaturated 29-bit-limb Montgomery field (fieldu.cuh) with
// twiddles stored as w * 2^261 mod r so the data never leave the arkworks (R = 2^256) domain.
// HBM-side the transform is 2 * N * 32 B algorithmic bytes; the kernel is integer-VALU bound.
#include "ctx.h"

#include "ntt_pass.cuh"

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

// ------------------------------------------------------------------------------------ table builders
template <class Fr>
struct PowBits {
    Fr p[32];  // base^(2^b)
};
// mode 0: out[i] = mul * base^i
// mode 1: out[i] = mul * base^((col * k) mod 2^log_mprev), i = k * M + col, M = 2^log_m
// Tables are written in the R' = 2^261 Montgomery form the pass kernels multiply with: `to_rp` is the
// plain integer R' mod r, and Montgomery-multiplying the arkworks-form value by it gives w * R' mod r.
template <class Fr>
__global__ void gen_pow_table(void* out, uint64_t n, PowBits<Fr> pb, Fr mul, int mode, uint32_t log_m, uint32_t log_mprev, Fr to_rp) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t e = i;
    if (mode == 1) {
        uint64_t col = i & ((1ull << log_m) - 1);
        uint64_t k = i >> log_m;
        e = (col * k) & ((1ull << log_mprev) - 1);
    }
    Fr r = mul;
    for (int b = 0; b < 32; ++b) {
        if ((e >> b) & 1ull) r = Fr::mul(r, pb.p[b]);
    }
    st_fr<Fr>(out, i, Fr::mul(r, to_rp));
}

// packed 32-byte table entries -> the 29-bit limbs of the pass kernels' field type, 48 B per entry (ntt_pass.cuh: ld_limbs)
template <class FU>
__global__ void split_table(const void* packed, void* out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FU x = ld_u<FU>(packed, i);
    uint32_t w[4 * TW_INNER_U4];
#pragma unroll
    for (int k = 0; k < 4 * TW_INNER_U4; ++k) w[k] = k < FU::NL ? x.v[k] : 0u;
    uint4* q = reinterpret_cast<uint4*>(out) + TW_INNER_U4 * i;
#pragma unroll
    for (int k = 0; k < TW_INNER_U4; ++k) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// tiny transforms (N = 1, 2, 4): one lane, straight from the definition (arkworks-form arithmetic)
template <class Fr>
__global__ void ntt_tiny(const void* in, void* out, uint64_t in_len, uint32_t log_n, Fr w /* w_N or its inverse */, Fr pre_g /* g or 1 */,
                         Fr post_g /* g^-1 or 1 */, Fr scale /* 1/N or 1 */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n = 1u << log_n;
    Fr x[4], y[4];
    Fr gp = Fr::one();
    for (uint32_t i = 0; i < n; ++i) {
        x[i] = i < in_len ? Fr::mul(ld_fr<Fr>(in, i), gp) : Fr::zero();
        gp = Fr::mul(gp, pre_g);
    }
    Fr wi = Fr::one(), pg = scale;
    for (uint32_t i = 0; i < n; ++i) {
        Fr acc = Fr::zero(), wij = Fr::one();
        for (uint32_t j = 0; j < n; ++j) {
            acc = Fr::add(acc, Fr::mul(x[j], wij));
            wij = Fr::mul(wij, wi);
        }
        y[i] = Fr::mul(acc, pg);
        pg = Fr::mul(pg, post_g);
        wi = Fr::mul(wi, w);
    }
    for (uint32_t i = 0; i < n; ++i) st_fr<Fr>(out, i, y[i]);
}

template <class Fr>
__global__ void fr_convert_kernel(const void* in, void* out, uint64_t n, int to_mont) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = ld_fr<Fr>(in, i);
    st_fr<Fr>(out, i, to_mont ? Fr::to_mont(x) : Fr::from_mont(x));
}
template <class Fr>
__global__ void fr_mul_kernel(const void* a, const void* b, void* out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st_fr<Fr>(out, i, Fr::mul(ld_fr<Fr>(a, i), ld_fr<Fr>(b, i)));
}

// ------------------------------------------------------------------------------------ host side
template <class C>
typename C::Fr root_of_unity_host(uint32_t log_n) {
    typedef typename C::Fr Fr;
    Fr r;
    for (int i = 0; i < Fr::N; ++i) r.v[i] = C::FrP::ROOT(i);
    for (uint32_t k = log_n; k < (uint32_t)C::FrP::TWO_ADICITY; ++k) r = Fr::sqr(r);
    return r;
}

template <class Fr>
PowBits<Fr> make_powbits(Fr base) {
    PowBits<Fr> pb;
    Fr cur = base;
    for (int b = 0; b < 32; ++b) {
        pb.p[b] = cur;
        cur = Fr::sqr(cur);
    }
    return pb;
}

// R' mod r as a plain integer in the saturated type's limbs
template <class C>
typename C::Fr rprime_plain() {
    typename C::Fr r;
    C::FrU::one().pack_words(r.v);
    return r;
}

template <class C>
int launch_pow_table(zk_ctx* c, void* out, uint64_t n, typename C::Fr base, typename C::Fr mul, int mode, uint32_t log_m,
                     uint32_t log_mprev) {
    typedef typename C::Fr Fr;
    if (n == 0) return ZK_OK;
    PowBits<Fr> pb = make_powbits(base);
    Fr to_rp = rprime_plain<C>();
    const int T = 256;
    uint64_t blocks = (n + T - 1) / T;
    hipLaunchKernelGGL(gen_pow_table<Fr>, dim3((unsigned)blocks), dim3(T), 0, c->stream, out, n, pb, mul, mode, log_m, log_mprev, to_rp);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// pass decomposition: every pass radix in [3, 9] (a wavefront holds 2^9 elements)
void decompose(uint32_t log_n, int& n_pass, int s[4]) {
    s[0] = s[1] = s[2] = s[3] = 0;
    if (log_n <= 9) {
        n_pass = 1;
        s[0] = (int)log_n;
        return;
    }
    n_pass = (int)((log_n + 8) / 9);
    int base = (int)log_n / n_pass, extra = (int)log_n % n_pass;
    for (int i = 0; i < n_pass; ++i) s[i] = base + (i < extra ? 1 : 0);
}

template <class C>
int get_inner_tw(zk_ctx* c, int s, bool inverse, void** out) {
    typedef typename C::Fr Fr;
    uint64_t key = ((uint64_t)C::ID << 40) | ((uint64_t)(inverse ? 1 : 0) << 32) | (uint64_t)s;
    auto it = c->inner_tw.find(key);
    if (it != c->inner_tw.end()) {
        *out = it->second;
        return ZK_OK;
    }
    uint64_t cnt = s >= 1 ? (1ull << (s - 1)) : 1;
    void* packed = nullptr;
    void* p = nullptr;
    ZK_HIP_TRY(hipMalloc(&packed, cnt * sizeof(Fr)));
    if (hipMalloc(&p, cnt * TW_INNER_U4 * 16) != hipSuccess) {
        (void)hipFree(packed);
        return ZK_ERR_OOM;
    }
    Fr w = root_of_unity_host<C>((uint32_t)s);
    if (inverse) w = Fr::inverse(w);
    int rc = launch_pow_table<C>(c, packed, cnt, w, Fr::one(), 0, 0, 0);
    if (!rc) {
        const int T = 256;
        hipLaunchKernelGGL(split_table<typename C::FrU>, dim3((unsigned)((cnt + T - 1) / T)), dim3(T), 0, c->stream, packed, p, cnt);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = ZK_ERR_HIP;
    }
    (void)hipFree(packed);
    if (rc) {
        (void)hipFree(p);
        return rc;
    }
    c->inner_tw[key] = p;
    *out = p;
    return ZK_OK;
}

template <class C>
int get_plan(zk_ctx* c, uint32_t log_n, bool inverse, NttPlan** out) {
    typedef typename C::Fr Fr;
    uint64_t key = ((uint64_t)C::ID << 40) | ((uint64_t)(inverse ? 1 : 0) << 32) | log_n;
    auto it = c->plans.find(key);
    if (it != c->plans.end()) {
        *out = it->second;
        return ZK_OK;
    }
    NttPlan* pl = new NttPlan();
    pl->curve = C::ID;
    pl->log_n = log_n;
    pl->inverse = inverse;
    decompose(log_n, pl->n_pass, pl->s);
    uint32_t log_mprev = log_n;
    for (int p = 0; p < pl->n_pass; ++p) {
        if (log_n >= 3) {
            int rc = get_inner_tw<C>(c, pl->s[p], inverse, &pl->tw_inner[p]);
            if (rc) { delete pl; return rc; }
        }
        if (p + 1 < pl->n_pass) {
            uint32_t log_m = log_mprev - (uint32_t)pl->s[p];
            uint64_t cnt = 1ull << log_mprev;
            hipError_t e = hipMalloc(&pl->tw_pass[p], cnt * sizeof(Fr));
            if (e != hipSuccess) { delete pl; return ZK_ERR_OOM; }
            Fr w = root_of_unity_host<C>(log_mprev);
            if (inverse) w = Fr::inverse(w);
            int rc = launch_pow_table<C>(c, pl->tw_pass[p], cnt, w, Fr::one(), 1, log_m, log_mprev);
            if (rc) { delete pl; return rc; }
            log_mprev = log_m;
        }
    }
    c->plans[key] = pl;
    *out = pl;
    return ZK_OK;
}

// g^j (forward) or g^-j (inverse) tables, R'-form.
template <class C>
int ensure_coset(zk_ctx* c, bool inv, uint64_t len, void** out) {
    typedef typename C::Fr Fr;
    DevBuf& buf = inv ? c->coset_inv_pow[C::ID] : c->coset_pow[C::ID];
    size_t& have = inv ? c->coset_inv_len[C::ID] : c->coset_len[C::ID];
    if (have < len) {
        // tables are referenced by in-flight kernels of this stream only; growing reallocates,
        // so drain the stream first
        ZK_HIP_TRY(hipStreamSynchronize(c->stream));
        uint64_t want = 1;
        while (want < len) want <<= 1;
        int rc = buf.ensure(want * sizeof(Fr));
        if (rc) return rc;
        Fr g = Fr::from_u32(C::FrP::GENERATOR);
        if (inv) g = Fr::inverse(g);
        rc = launch_pow_table<C>(c, buf.p, want, g, Fr::one(), 0, 0, 0);
        if (rc) return rc;
        have = want;
    }
    *out = buf.p;
    return ZK_OK;
}

// The pass kernels are compiled per (curve, S) in ntt_pass_inst.hip; ntt_pass_table.hip maps
// (curve, S) to the launcher of that object.
typedef int (*NttPassLauncher)(int final_pass, const NttPassArgs* a, hipStream_t st);
extern "C" NttPassLauncher zk_ntt_pass_launcher(int curve, int s);

template <class C>
int dispatch_pass(zk_ctx* c, int s, bool final_pass, const NttPassArgs& a) {
    if (s < 3 || s > 9) return ZK_ERR_UNSUPPORTED;
    NttPassLauncher fn = zk_ntt_pass_launcher(C::ID, s);
    if (!fn) return ZK_ERR_UNSUPPORTED;
    hipError_t e = (hipError_t)fn(final_pass ? 1 : 0, &a, c->stream);
    ZK_HIP_TRY(e);
    return ZK_OK;
}

// n_polys (1 .. 16) independent transforms of one kind and size; n_polys > 1 runs every pass as ONE launch (blockIdx.y)
template <class C>
int ntt_run(zk_ctx* c, int kind, uint32_t log_n, uint32_t n_polys, const void* const* d_ins, const size_t* in_lens, void* const* d_outs) {
    typedef typename C::Fr Fr;
    if (log_n > (uint32_t)C::FrP::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
    if (n_polys == 0 || n_polys > 16) return ZK_ERR_BAD_ARG;
    const uint64_t N = 1ull << log_n;
    size_t in_len = 0;               // the longest input of the batch
    bool all_quarter = true;
    for (uint32_t i = 0; i < n_polys; ++i) {
        if (in_lens[i] > N) return ZK_ERR_BAD_ARG;
        if (in_lens[i] > in_len) in_len = in_lens[i];
        if ((uint64_t)in_lens[i] * 4 > N) all_quarter = false;
    }
    const void* d_in = d_ins[0];
    void* d_out = d_outs[0];
    const bool inverse = (kind == ZK_NTT_IFFT || kind == ZK_NTT_COSET_IFFT);
    int rc;
    Fr n_inv = Fr::inverse(Fr::from_u64(N));
    if (log_n < 3) {
        Fr w = root_of_unity_host<C>(log_n);
        if (inverse) w = Fr::inverse(w);
        Fr g = Fr::from_u32(C::FrP::GENERATOR);
        Fr pre_g = kind == ZK_NTT_COSET_FFT ? g : Fr::one();
        Fr post_g = kind == ZK_NTT_COSET_IFFT ? Fr::inverse(g) : Fr::one();
        ProfScope ps(c, "ntt_pass");
        for (uint32_t i = 0; i < n_polys; ++i)
            hipLaunchKernelGGL(ntt_tiny<Fr>, dim3(1), dim3(64), 0, c->stream, d_ins[i], d_outs[i], (uint64_t)in_lens[i], log_n, w, pre_g, post_g,
                               inverse ? n_inv : Fr::one());
        ZK_HIP_TRY(hipGetLastError());
        return ZK_OK;
    }
    void* pre = nullptr;
    void* post = nullptr;
    if (kind == ZK_NTT_COSET_FFT && in_len > 0) {
        rc = ensure_coset<C>(c, false, in_len, &pre);
        if (rc) return rc;
    }
    if (kind == ZK_NTT_COSET_IFFT) {
        rc = ensure_coset<C>(c, true, N, &post);
        if (rc) return rc;
    }
    NttPlan* pl = nullptr;
    rc = get_plan<C>(c, log_n, inverse, &pl);
    if (rc) return rc;
    // up to four passes of 2^9: log N <= 36 covers every two-adicity the curves have (32 for BLS12-381, 28 for BN254; the reference
    // admits any domain up to it, prover.rs:169-173, error.rs:14-21).  What bounds the size in practice is memory: the vector, the
    // work vector and the first pass boundary's table are N x 32 B each (8 GiB at 2^28), so 2^30 in place is 96 GiB of a 288 GB card
    // and 2^32 does not fit -- that case is ZK_ERR_OOM from the allocations below, not a refusal here.
    if (pl->n_pass > 4) return ZK_ERR_UNSUPPORTED;
    void* work = nullptr;
    if (pl->n_pass > 1) {
        rc = c->ntt_work.ensure((size_t)n_polys * N * sizeof(Fr));     // one intermediate vector per polynomial of the batch
        if (rc) return rc;
        work = c->ntt_work.p;
    }
    // the last pass multiplies every output by 1/N (inverse) or 1 (forward) -- the product that also
    // brings the lazily reduced value back under 2r -- and, for coset_ifft, by g^-j from the table
    const Fr to_rp = rprime_plain<C>();
    uint32_t log_mprev = log_n;
    NttPassArgs a;
    ProfScope ps(c, "ntt_pass");       // one scope per call (all passes of the transform or batch): an event pair costs ~14 us of host time
    for (int p = 0; p < pl->n_pass; ++p) {
        const int s = pl->s[p];
        const bool last = (p + 1 == pl->n_pass);
        memset(&a, 0, sizeof a);
        a.log_n = log_n;
        a.in = (p == 0) ? d_in : work;
        a.out = last ? d_out : work;
        a.in_len = (p == 0) ? in_lens[0] : N;
        if (n_polys > 1) {
            a.n_batch = n_polys;
            for (uint32_t i = 0; i < n_polys; ++i) {
                void* wk = (char*)work + (size_t)i * N * sizeof(Fr);
                a.ins[i] = (p == 0) ? d_ins[i] : wk;
                a.outs[i] = last ? d_outs[i] : wk;
                a.in_lens[i] = (p == 0) ? in_lens[i] : N;
            }
        }
        a.tw_inner = pl->tw_inner[p];
        a.pre_mul = (p == 0) ? pre : nullptr;
        const uint32_t LC = 9u - (uint32_t)s;
        if (!last) {
            uint32_t log_m = log_mprev - (uint32_t)s;
            uint32_t logc = LC < log_m ? LC : log_m;
            a.logc = logc;
            a.log_m = log_m;
            a.log_mprev = log_mprev;
            a.tw_pass = pl->tw_pass[p];
            a.n_tiles = (uint32_t)(N >> ((uint32_t)s + logc));
            a.quarter = (p == 0 && s >= 3 && all_quarter) ? 1u : 0u;
            log_mprev = log_m;
        } else {
            uint32_t s1 = (pl->n_pass > 1) ? (uint32_t)pl->s[0] : 0;
            uint32_t logc = LC < s1 ? LC : s1;
            a.logc = logc;
            a.s1 = s1;
            a.s2 = pl->n_pass == 4 ? (uint32_t)pl->s[1] : 0u;
            a.post_mul = post;
            Fr sc = Fr::mul(inverse ? n_inv : Fr::one(), to_rp);   // R'-form of 1/N or 1
            for (int i = 0; i < 8; ++i) a.scale[i] = sc.v[i];
            a.n_tiles = (uint32_t)(N >> ((uint32_t)s + logc));
        }
        rc = dispatch_pass<C>(c, s, last, a);
        if (rc) return rc;
    }
    return ZK_OK;
}

}  // namespace

int ntt_run_dev(zk_ctx* c, int curve, int kind, uint32_t log_n, const void* d_in, size_t in_len, void* d_out) {
    return ntt_run_batch_dev(c, curve, kind, log_n, 1, &d_in, &in_len, &d_out);
}

// A batch in place (d_ins[i] == d_outs[i]) is fine: every pass but the last writes the ctx's work vectors.  Two DIFFERENT
// polynomials of a batch must not alias each other.
int ntt_run_batch_dev(zk_ctx* c, int curve, int kind, uint32_t log_n, uint32_t n_polys, const void* const* d_ins, const size_t* in_lens,
                      void* const* d_outs) {
    if (kind < 0 || kind > 3) return ZK_ERR_BAD_ARG;
    for (uint32_t base = 0; base < n_polys; base += 16) {
        const uint32_t cnt = n_polys - base < 16 ? n_polys - base : 16;
        int rc;
        if (curve == ZK_CURVE_BLS12_381) rc = ntt_run<CurveBls>(c, kind, log_n, cnt, d_ins + base, in_lens + base, d_outs + base);
        else if (curve == ZK_CURVE_BN254) rc = ntt_run<CurveBn>(c, kind, log_n, cnt, d_ins + base, in_lens + base, d_outs + base);
        else return ZK_ERR_BAD_ARG;
        if (rc) return rc;
    }
    return ZK_OK;
}

int ntt_prepare(zk_ctx* c, int curve, uint32_t log_n) {
    NttPlan* pl;
    if (log_n < 3) return ZK_OK;
    for (int inv = 0; inv < 2; ++inv) {
        int rc;
        if (curve == ZK_CURVE_BLS12_381) {
            if (log_n > (uint32_t)FrBls12_381Params::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
            rc = get_plan<CurveBls>(c, log_n, inv != 0, &pl);
        } else if (curve == ZK_CURVE_BN254) {
            if (log_n > (uint32_t)FrBn254Params::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
            rc = get_plan<CurveBn>(c, log_n, inv != 0, &pl);
        } else {
            return ZK_ERR_BAD_ARG;
        }
        if (rc) return rc;
    }
    return ZK_OK;
}

void ntt_ctx_free(zk_ctx* c) {
    for (auto& kv : c->plans) delete kv.second;
    c->plans.clear();
    for (auto& kv : c->inner_tw) (void)hipFree(kv.second);
    c->inner_tw.clear();
    c->ntt_work.release();
    for (int i = 0; i < 2; ++i) {
        c->coset_pow[i].release();
        c->coset_inv_pow[i].release();
    }
}

static int fr_convert_on(int curve, int to_mont, const void* d_in, size_t n, void* d_out, hipStream_t st) {
    if (n == 0) return ZK_OK;
    const int T = 256;
    unsigned blocks = (unsigned)((n + T - 1) / T);
    if (curve == ZK_CURVE_BLS12_381)
        hipLaunchKernelGGL(fr_convert_kernel<FrBls>, dim3(blocks), dim3(T), 0, st, d_in, d_out, (uint64_t)n, to_mont);
    else if (curve == ZK_CURVE_BN254)
        hipLaunchKernelGGL(fr_convert_kernel<FrBn>, dim3(blocks), dim3(T), 0, st, d_in, d_out, (uint64_t)n, to_mont);
    else
        return ZK_ERR_BAD_ARG;
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}
int fr_convert_dev(zk_ctx* c, int curve, int to_mont, const void* d_in, size_t n, void* d_out) {
    return fr_convert_on(curve, to_mont, d_in, n, d_out, c->stream);
}
// Montgomery -> canonical (into_repr) on a chosen stream
int fr_convert_stream(zk_ctx* c, int curve, const void* d_in, size_t n, void* d_out, hipStream_t st) {
    (void)c;
    return fr_convert_on(curve, 0, d_in, n, d_out, st);
}

int fr_mul_dev(zk_ctx* c, int curve, const void* a, const void* b, size_t n, void* out) {
    if (n == 0) return ZK_OK;
    const int T = 256;
    unsigned blocks = (unsigned)((n + T - 1) / T);
    if (curve == ZK_CURVE_BLS12_381)
        hipLaunchKernelGGL(fr_mul_kernel<FrBls>, dim3(blocks), dim3(T), 0, c->stream, a, b, out, (uint64_t)n);
    else if (curve == ZK_CURVE_BN254)
        hipLaunchKernelGGL(fr_mul_kernel<FrBn>, dim3(blocks), dim3(T), 0, c->stream, a, b, out, (uint64_t)n);
    else
        return ZK_ERR_BAD_ARG;
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}
