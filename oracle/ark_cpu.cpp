// TEST INFRASTRUCTURE ONLY -- CPU restatement ("port") of the arkworks 0.3 algorithms behind
// plonk-core's NTT + MSM hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product (ark_plonk_amd/) never links or calls it.
//
// PARITY UNPINNED BY THE REFERENCE: heliaxdev/ark-plonk has no golden vectors
// for this path and the arithmetic lives in crates.io dependencies absent from /root/reference
// (ark-poly 0.3.0, ark-ec 0.3.0, ark-ff 0.3.0, ark-bls12-381 0.3.0, ark-poly-commit 0.3.0;
// plonk-core/Cargo.toml:51-58, no Cargo.lock).  This file restates their published algorithms:
//   * Fp<N>::mul            -- ark-ff 0.3 Fp256/Fp384 Montgomery multiplication (CIOS, 64-bit limbs)
//   * ntt_*                 -- ark-poly 0.3 Radix2EvaluationDomain: fft = in-order DIF (io_helper,
//                              per-stage twiddle compaction) + derange; ifft = derange + DIT (oi_helper)
//                              with w^-1 then *size_inv; coset_fft = distribute_powers(g) over the
//                              *input length* then fft; coset_ifft = ifft then distribute_powers(g^-1).
//                              Reference call sites: prover.rs:196-203, quotient_poly.rs:72-120,175-177,
//                              permutation/mod.rs:671-674,751.
//   * g1 jacobian formulas  -- ark-ec 0.3 short_weierstrass_jacobian.rs: add_assign_mixed
//                              (madd-2007-bl), add_assign (add-2007-bl), double_in_place (dbl-2009-l, a=0)
//   * msm_pippenger         -- ark-ec 0.3 msm/variable_base.rs VariableBaseMSM::multi_scalar_mul:
//                              c = 3 if n<32 else ceil(log2 n)*69/100+2, windows over MODULUS_BITS,
//                              zero scalars skipped, scalar==1 added in window 0, 2^c-1 buckets,
//                              running-sum reduction, threads over windows (rayon in the reference).
//                              Reference call sites: commitment.rs:45, every PC::commit in prover.rs.
//   * kzg_commit / kzg_open -- ark-poly-commit 0.3 kzg10: skip leading zero coefficients, into_repr,
//                              MSM over powers_of_g[lz..]; witness poly by division by (X - z).
// It is pinned (tests/test_oracle.py) against oracle/bigint_oracle.py (definitional big-int
// arithmetic), against the ark-bls12-381 TWO_ADIC_ROOT_OF_UNITY constant and the published [2]G1.
//
// Build: see oracle/Makefile (g++ -O3 -fopenmp -shared -fPIC).
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

// ------------------------------------------------------------------ field parameters (derived at init)
template <int N> struct FpParams {
    u64 p[N];      // modulus
    u64 inv;       // -p^-1 mod 2^64
    u64 R[N];      // 2^(64N) mod p   (Montgomery one)
    u64 R2[N];     // 2^(128N) mod p
    int bits;      // modulus bit length
};

template <int N> static inline bool geq(const u64* a, const u64* b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
template <int N> static inline u64 sub_n(u64* r, const u64* a, const u64* b) {
    u64 borrow = 0;
    for (int i = 0; i < N; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
template <int N> static inline u64 add_n(u64* r, const u64* a, const u64* b) {
    u64 carry = 0;
    for (int i = 0; i < N; ++i) {
        u128 s = (u128)a[i] + b[i] + carry;
        r[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
    return carry;
}

template <int N> static void derive_params(FpParams<N>& P, const u64* modulus) {
    memcpy(P.p, modulus, sizeof(u64) * N);
    // -p^-1 mod 2^64 by Newton iteration
    u64 x = 1;
    for (int i = 0; i < 6; ++i) x *= 2 - modulus[0] * x;
    P.inv = (u64)0 - x;
    // bit length
    P.bits = 0;
    for (int i = N - 1; i >= 0; --i)
        if (modulus[i]) { P.bits = 64 * i + 64 - __builtin_clzll(modulus[i]); break; }
    // R = 2^(64N) mod p by 64N modular doublings of 1; R2 by 64N more
    u64 v[N];
    memset(v, 0, sizeof v);
    v[0] = 1;
    for (int k = 0; k < 128 * N; ++k) {
        u64 c = add_n<N>(v, v, v);
        if (c || geq<N>(v, P.p)) sub_n<N>(v, v, P.p);
        if (k == 64 * N - 1) memcpy(P.R, v, sizeof v);
    }
    memcpy(P.R2, v, sizeof v);
}

// ------------------------------------------------------------------ Montgomery field element
template <int N, int TAG> struct Fp {
    u64 v[N];
    static FpParams<N> P;

    static Fp zero() { Fp r; memset(r.v, 0, sizeof r.v); return r; }
    static Fp one() { Fp r; memcpy(r.v, P.R, sizeof r.v); return r; }
    bool is_zero() const { u64 o = 0; for (int i = 0; i < N; ++i) o |= v[i]; return o == 0; }
    bool operator==(const Fp& o) const { return memcmp(v, o.v, sizeof v) == 0; }

    Fp operator+(const Fp& o) const {
        Fp r; u64 c = add_n<N>(r.v, v, o.v);
        if (c || geq<N>(r.v, P.p)) sub_n<N>(r.v, r.v, P.p);
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r; u64 b = sub_n<N>(r.v, v, o.v);
        if (b) add_n<N>(r.v, r.v, P.p);
        return r;
    }
    Fp neg() const { return is_zero() ? *this : zero() - *this; }
    Fp dbl() const { return *this + *this; }

    // CIOS Montgomery product (ark-ff 0.3 fp_256.rs / fp_384.rs mul_assign, generic path)
    Fp operator*(const Fp& o) const {
        u64 t[N + 2];
        memset(t, 0, sizeof t);
        for (int i = 0; i < N; ++i) {
            u64 carry = 0;
            for (int j = 0; j < N; ++j) {
                u128 s = (u128)v[j] * o.v[i] + t[j] + carry;
                t[j] = (u64)s; carry = (u64)(s >> 64);
            }
            u128 s = (u128)t[N] + carry;
            t[N] = (u64)s; t[N + 1] = (u64)(s >> 64);
            u64 m = t[0] * P.inv;
            s = (u128)m * P.p[0] + t[0];
            carry = (u64)(s >> 64);
            for (int j = 1; j < N; ++j) {
                s = (u128)m * P.p[j] + t[j] + carry;
                t[j - 1] = (u64)s; carry = (u64)(s >> 64);
            }
            s = (u128)t[N] + carry;
            t[N - 1] = (u64)s;
            t[N] = t[N + 1] + (u64)(s >> 64);
        }
        Fp r;
        if (t[N] || geq<N>(t, P.p)) sub_n<N>(r.v, t, P.p); else memcpy(r.v, t, sizeof r.v);
        return r;
    }
    Fp sqr() const { return *this * *this; }

    Fp pow_limbs(const u64* e, int nlimbs) const {
        Fp r = one();
        bool started = false;
        for (int i = nlimbs - 1; i >= 0; --i)
            for (int b = 63; b >= 0; --b) {
                if (started) r = r.sqr();
                if ((e[i] >> b) & 1) { r = r * *this; started = true; }
            }
        return r;
    }
    Fp pow_u64(u64 e) const { return pow_limbs(&e, 1); }
    Fp inverse() const {  // Fermat: a^(p-2)
        u64 e[N]; u64 two[N]; memset(two, 0, sizeof two); two[0] = 2;
        sub_n<N>(e, P.p, two);
        return pow_limbs(e, N);
    }
    static Fp from_canonical(const u64* c) {  // into Montgomery form: c * R2 * R^-1
        Fp a, r2; memcpy(a.v, c, sizeof a.v); memcpy(r2.v, P.R2, sizeof r2.v);
        return a * r2;
    }
    void to_canonical(u64* out) const {       // into_repr: multiply by 1
        Fp o; memset(o.v, 0, sizeof o.v); o.v[0] = 1;
        Fp r = *this * o; memcpy(out, r.v, sizeof r.v);
    }
    static Fp from_u64(u64 x) { u64 c[N]; memset(c, 0, sizeof c); c[0] = x; return from_canonical(c); }
};
template <int N, int TAG> FpParams<N> Fp<N, TAG>::P;

// ------------------------------------------------------------------ curve descriptions
struct CurveDesc {
    int fr_limbs, fq_limbs, two_adicity; u64 fr_gen; u64 b;
    u64 r[4]; u64 q[6]; u64 gx[6]; u64 gy[6];
};
// moduli / generators only; everything else is derived.  (little-endian 64-bit limbs)
static const CurveDesc CURVES[2] = {
    {4, 6, 32, 7, 4,
     {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
     {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
      0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL},
     {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL,
      0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL},
     {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL,
      0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}},
    {4, 4, 28, 5, 3,
     {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
     {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL, 0, 0},
     {1, 0, 0, 0, 0, 0},
     {2, 0, 0, 0, 0, 0}},
};

typedef Fp<4, 0> FrBls;  typedef Fp<6, 1> FqBls;
typedef Fp<4, 2> FrBn;   typedef Fp<4, 3> FqBn;

static bool g_init = false;
static void ensure_init() {
    if (g_init) return;
    derive_params<4>(FrBls::P, CURVES[0].r);
    derive_params<6>(FqBls::P, CURVES[0].q);
    derive_params<4>(FrBn::P, CURVES[1].r);
    derive_params<4>(FqBn::P, CURVES[1].q);
    g_init = true;
}

// ------------------------------------------------------------------ NTT (ark-poly 0.3 radix2/fft.rs)
template <class F> struct Domain {
    int log_n; size_t n;
    F group_gen, group_gen_inv, size_inv, gen, gen_inv;
    Domain(const CurveDesc& cd, int log_n_) : log_n(log_n_), n((size_t)1 << log_n_) {
        // TWO_ADIC_ROOT_OF_UNITY = GENERATOR^((r-1)/2^two_adicity); group_gen = root^(2^(s-log_n))
        u64 e[4], onev[4] = {1, 0, 0, 0};
        sub_n<4>(e, F::P.p, onev);
        // e >>= two_adicity
        int s = cd.two_adicity;
        for (int k = 0; k < s; ++k) { for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0); }
        gen = F::from_u64(cd.fr_gen);
        F root = gen.pow_limbs(e, 4);
        for (int k = 0; k < s - log_n; ++k) root = root.sqr();
        group_gen = root; group_gen_inv = root.inverse();
        size_inv = F::from_u64((u64)n).inverse();
        gen_inv = gen.inverse();
    }
};

static inline size_t bitrev(size_t a, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (a & 1); a >>= 1; }
    return r;
}
// ark-poly 0.3 `derange`: swap xs[i] <-> xs[bitrev(i)].  Every pair (i, ri) with i < ri is touched by exactly one iteration, so the
// loop parallelises as it stands (the reference's is serial; same permutation).
template <class F> static void derange(F* xs, int log_n) {
    size_t n = (size_t)1 << log_n;
    #pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t i = 1; i < n; ++i) { size_t ri = bitrev(i, log_n); if (i < ri) std::swap(xs[i], xs[ri]); }
}
// roots[i] = root^i, i < n/2.  ark 0.3's parallel `roots_of_unity` (compute_powers): each block starts from root^(block start)
// and runs the recurrence inside the block -- the same field elements as the serial recurrence (a power has one reduced value).
template <class F> static std::vector<F> roots_of_unity(F root, size_t n) {
    size_t h = n / 2 > 0 ? n / 2 : 1;
    std::vector<F> r(h);
    const size_t B = 4096;
    size_t nb = (h + B - 1) / B;
    #pragma omp parallel for schedule(static) if (h >= 4 * B)
    for (size_t b = 0; b < nb; ++b) {
        F cur = root.pow_u64((u64)(b * B));
        size_t e = std::min(h, (b + 1) * B);
        for (size_t i = b * B; i < e; ++i) { r[i] = cur; cur = cur * root; }
    }
    return r;
}
// Speed only, semantics untouched: ark-poly 0.3 runs the log2(n) butterfly stages one after the other over the
// whole vector, parallel inside a stage (`apply_butterfly`), i.e. log2(n) passes over memory -- 22 passes over 128 MiB for a 2^22
// transform.  The butterflies of a stage with gap g stay inside aligned chunks of 2g elements, and so do all the stages with smaller
// gaps: once a chunk fits a core's cache (BLOCK elements = 1 MiB) a thread runs ALL the remaining stages of its chunk before it moves
// on.  Same butterflies, same operands, same results (the order of independent butterflies does not matter), ~log2(n / BLOCK) + 1
// passes over memory.  The twiddle table is indexed with a stride instead of being compacted between stages (same values).
static const size_t NTT_BLOCK = (size_t)1 << 15;
// The twiddles of the stage with gap g are roots[j * (n/2) / g], j < g: read with that stride they are one cache line and -- for the
// small gaps -- one page each.  ark compacts them per stage for the same reason (`compacted_roots`, MIN_NUM_CHUNKS_FOR_COMPACTION);
// here: one table for all the in-block stages (the entries of gap g at [g, 2g)), and a per-stage copy for the stages above.
template <class F> static std::vector<F> block_twiddles(const F* roots, size_t half, size_t max_gap) {
    std::vector<F> t(2 * max_gap > 2 ? 2 * max_gap : 2);
    for (size_t g = 1; g <= max_gap; g *= 2) {
        const size_t stride = half / g;
        #pragma omp parallel for schedule(static) if (g >= 4096)
        for (size_t j = 0; j < g; ++j) t[g + j] = roots[j * stride];
    }
    return t;
}
template <class F> static const F* stage_twiddles(const F* roots, size_t half, size_t gap, std::vector<F>& scratch) {
    const size_t stride = half / gap;
    if (stride == 1) return roots;
    scratch.resize(gap);
    #pragma omp parallel for schedule(static)
    for (size_t j = 0; j < gap; ++j) scratch[j] = roots[j * stride];
    return scratch.data();
}
// one DIF stage (gap) on the chunk xs[lo, hi): in-order -> out-of-order butterflies  (a, b) -> (a + b, (a - b) w_j)
template <class F> static inline void dif_stage(F* xs, size_t lo, size_t hi, size_t gap, const F* tw) {
    for (size_t c = lo; c < hi; c += 2 * gap)
        for (size_t j = 0; j < gap; ++j) {
            F* x = xs + c + j; F* y = x + gap;
            F neg = *x - *y; *x = *x + *y; *y = neg * tw[j];
        }
}
// one DIT stage (gap) on the chunk xs[lo, hi): out-of-order -> in-order butterflies  (a, b) -> (a + w_j b, a - w_j b)
template <class F> static inline void dit_stage(F* xs, size_t lo, size_t hi, size_t gap, const F* tw) {
    for (size_t c = lo; c < hi; c += 2 * gap)
        for (size_t j = 0; j < gap; ++j) {
            F* x = xs + c + j; F* y = x + gap;
            F t = *y * tw[j];
            F neg = *x - t; *x = *x + t; *y = neg;
        }
}
// in-order -> out-of-order, decimation in frequency (Gentleman-Sande): ark-poly 0.3 io_helper
template <class F> static void io_helper(F* xs, size_t n, F root) {
    std::vector<F> roots = roots_of_unity(root, n);
    const size_t half = n / 2;
    if (half == 0) return;
    std::vector<F> scratch;
    size_t gap = half;
    // stages whose chunks are larger than a block: parallel inside the stage (chunk x piece-of-the-half-chunk items)
    for (; 2 * gap > NTT_BLOCK; gap /= 2) {
        const F* tw = stage_twiddles(roots.data(), half, gap, scratch);
        const size_t chunk = 2 * gap, nch = n / chunk, piece = 4096, ppc = gap / piece;     // gap >= NTT_BLOCK / 2 > piece here
        #pragma omp parallel for schedule(static)
        for (size_t t = 0; t < nch * ppc; ++t) {
            const size_t c = (t / ppc) * chunk, j0 = (t % ppc) * piece;
            for (size_t j = j0; j < j0 + piece; ++j) {
                F* x = xs + c + j; F* y = x + gap;
                F neg = *x - *y; *x = *x + *y; *y = neg * tw[j];
            }
        }
    }
    // all remaining stages, one cache-resident chunk at a time
    const std::vector<F> small = block_twiddles(roots.data(), half, gap);
    const size_t chunk = 2 * gap;
    #pragma omp parallel for schedule(static) if (n >= 4096)
    for (size_t c = 0; c < n; c += chunk)
        for (size_t g = gap; g > 0; g /= 2) dif_stage(xs, c, c + chunk, g, small.data() + g);
}
// out-of-order -> in-order, decimation in time (Cooley-Tukey): ark-poly 0.3 oi_helper
template <class F> static void oi_helper(F* xs, size_t n, F root) {
    std::vector<F> roots = roots_of_unity(root, n);
    const size_t half = n / 2;
    if (half == 0) return;
    const size_t chunk = std::min(n, NTT_BLOCK);
    {   // the stages with gap < chunk, one cache-resident chunk at a time
        const std::vector<F> small = block_twiddles(roots.data(), half, chunk / 2);
        #pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t c = 0; c < n; c += chunk)
            for (size_t g = 1; g < chunk; g *= 2) dit_stage(xs, c, c + chunk, g, small.data() + g);
    }
    // the stages whose chunks are larger than a block: parallel inside the stage
    std::vector<F> scratch;
    for (size_t gap = chunk; gap < n; gap *= 2) {
        const F* tw = stage_twiddles(roots.data(), half, gap, scratch);
        const size_t ch = 2 * gap, nch = n / ch, piece = 4096, ppc = gap / piece;
        #pragma omp parallel for schedule(static)
        for (size_t t = 0; t < nch * ppc; ++t) {
            const size_t c = (t / ppc) * ch, j0 = (t % ppc) * piece;
            for (size_t j = j0; j < j0 + piece; ++j) {
                F* x = xs + c + j; F* y = x + gap;
                F tv = *y * tw[j];
                F neg = *x - tv; *x = *x + tv; *y = neg;
            }
        }
    }
}
template <class F> static void distribute_powers(F* xs, size_t len, F g) {
    // serial recurrence in the reference; blocked here (same values)
    const size_t B = 1024;
    size_t nb = (len + B - 1) / B;
    #pragma omp parallel for schedule(static) if (len >= 4096)
    for (size_t b = 0; b < nb; ++b) {
        F p = g.pow_u64((u64)(b * B));
        size_t e = std::min(len, (b + 1) * B);
        for (size_t i = b * B; i < e; ++i) { xs[i] = xs[i] * p; p = p * g; }
    }
}
template <class F> static int ntt_run(const CurveDesc& cd, int kind, int log_n, const u64* in, size_t in_len, u64* out) {
    if (log_n < 0 || log_n > cd.two_adicity) return -2;
    size_t n = (size_t)1 << log_n;
    if (in_len > n) return -1;
    Domain<F> d(cd, log_n);
    std::vector<F> xs(n, F::zero());
    memcpy(xs.data(), in, in_len * sizeof(F));
    switch (kind) {
    case 0: io_helper(xs.data(), n, d.group_gen); derange(xs.data(), log_n); break;
    case 2: distribute_powers(xs.data(), in_len, d.gen);
            io_helper(xs.data(), n, d.group_gen); derange(xs.data(), log_n); break;
    case 1: case 3:
        derange(xs.data(), log_n); oi_helper(xs.data(), n, d.group_gen_inv);
        #pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t i = 0; i < n; ++i) xs[i] = xs[i] * d.size_inv;
        if (kind == 3) distribute_powers(xs.data(), n, d.gen_inv);
        break;
    default: return -1;
    }
    memcpy(out, xs.data(), n * sizeof(F));
    return 0;
}

// ------------------------------------------------------------------ G1 (ark-ec 0.3 short_weierstrass_jacobian.rs, a = 0)
template <class Fq> struct Affine { Fq x, y; bool inf; };
template <class Fq> struct Jac {
    Fq x, y, z;
    static Jac zero() { Jac r; r.x = Fq::one(); r.y = Fq::one(); r.z = Fq::zero(); return r; }
    bool is_zero() const { return z.is_zero(); }

    void double_in_place() {  // dbl-2009-l
        if (is_zero()) return;
        Fq a = x.sqr(), b = y.sqr(), c = b.sqr();
        Fq d = ((x + b).sqr() - a - c).dbl();
        Fq e = a + a.dbl();
        Fq f = e.sqr();
        z = (z * y).dbl();
        x = f - d - d;
        y = (d - x) * e - c.dbl().dbl().dbl();
    }
    void add_assign_mixed(const Affine<Fq>& o) {  // madd-2007-bl
        if (o.inf) return;
        if (is_zero()) { x = o.x; y = o.y; z = Fq::one(); return; }
        Fq z1z1 = z.sqr();
        Fq u2 = o.x * z1z1;
        Fq s2 = (o.y * z) * z1z1;
        if (x == u2 && y == s2) { double_in_place(); return; }
        Fq h = u2 - x;
        Fq hh = h.sqr();
        Fq i = hh.dbl().dbl();
        Fq j = h * i;
        Fq r = (s2 - y).dbl();
        Fq v = x * i;
        Fq nx = r.sqr() - j - v.dbl();
        Fq ny = r * (v - nx) - (y * j).dbl();
        Fq nz = (z + h).sqr() - z1z1 - hh;
        x = nx; y = ny; z = nz;
    }
    void add_assign(const Jac& o) {  // add-2007-bl
        if (is_zero()) { *this = o; return; }
        if (o.is_zero()) return;
        Fq z1z1 = z.sqr(), z2z2 = o.z.sqr();
        Fq u1 = x * z2z2, u2 = o.x * z1z1;
        Fq s1 = y * o.z * z2z2, s2 = o.y * z * z1z1;
        if (u1 == u2 && s1 == s2) { double_in_place(); return; }
        Fq h = u2 - u1;
        Fq i = h.dbl().sqr();
        Fq j = h * i;
        Fq r = (s2 - s1).dbl();
        Fq v = u1 * i;
        Fq nx = r.sqr() - j - v.dbl();
        Fq ny = r * (v - nx) - (s1 * j).dbl();
        Fq nz = ((z + o.z).sqr() - z1z1 - z2z2) * h;
        x = nx; y = ny; z = nz;
    }
    Affine<Fq> into_affine() const {
        Affine<Fq> a;
        if (is_zero()) { a.x = Fq::zero(); a.y = Fq::one(); a.inf = true; return a; }
        Fq zi = z.inverse(), zi2 = zi.sqr();
        a.x = x * zi2; a.y = y * zi2 * zi; a.inf = false;
        return a;
    }
};

static int ark_window(size_t n) {
    if (n < 32) return 3;
    int lg = 0; while (((size_t)1 << lg) < n) ++lg;   // ark_std::log2 = ceil
    return lg * 69 / 100 + 2;
}

// VariableBaseMSM::multi_scalar_mul (threads over windows, as rayon does in the reference)
template <class Fq, class Fr>
static Jac<Fq> msm_pippenger(const Affine<Fq>* bases, const u64* scalars /* n x 4 canonical */, size_t n, int threads) {
    int c = ark_window(n);
    int num_bits = Fr::P.bits;
    std::vector<int> starts;
    for (int w = 0; w < num_bits; w += c) starts.push_back(w);
    int W = (int)starts.size();
    std::vector<Jac<Fq>> sums(W);
    (void)threads;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int wi = 0; wi < W; ++wi) {
        int w_start = starts[wi];
        Jac<Fq> res = Jac<Fq>::zero();
        std::vector<Jac<Fq>> buckets(((size_t)1 << c) - 1, Jac<Fq>::zero());
        for (size_t i = 0; i < n; ++i) {
            const u64* s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
            if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
                if (w_start == 0) res.add_assign_mixed(bases[i]);
                continue;
            }
            int limb = w_start / 64, off = w_start % 64;
            u64 v = s[limb] >> off;
            if (off && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
            u64 d = v & (((u64)1 << c) - 1);
            if (d) buckets[d - 1].add_assign_mixed(bases[i]);
        }
        Jac<Fq> running = Jac<Fq>::zero();
        for (size_t b = buckets.size(); b-- > 0;) { running.add_assign(buckets[b]); res.add_assign(running); }
        sums[wi] = res;
    }
    Jac<Fq> total = Jac<Fq>::zero();
    for (int wi = W - 1; wi >= 1; --wi) {
        total.add_assign(sums[wi]);
        for (int k = 0; k < c; ++k) total.double_in_place();
    }
    Jac<Fq> lowest = sums[0];
    lowest.add_assign(total);
    return lowest;
}

// The same sum with the work cut into (window, range of points) tasks so that every core of a large host is busy -- NOT ark's shape
// (ark 0.3 parallelises over the windows only: at most W = 17 busy threads at 2^20 points), reported by bench.py as `all_cores`.
// Every task accumulates its range's digits of one window into its own buckets and reduces them with the running sum; the window's
// sum is the sum of its tasks' results.  `parts` ranges per window.
template <class Fq, class Fr>
static Jac<Fq> msm_pippenger_chunked(const Affine<Fq>* bases, const u64* scalars, size_t n, int threads, int parts) {
    int c = ark_window(n);
    int num_bits = Fr::P.bits;
    std::vector<int> starts;
    for (int w = 0; w < num_bits; w += c) starts.push_back(w);
    const int W = (int)starts.size();
    if (parts < 1) parts = 1;
    std::vector<Jac<Fq>> part_sums((size_t)W * parts);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int t = 0; t < W * parts; ++t) {
        const int wi = t / parts, pi = t % parts;
        const int w_start = starts[wi];
        const size_t lo = n * (size_t)pi / parts, hi = n * (size_t)(pi + 1) / parts;
        Jac<Fq> res = Jac<Fq>::zero();
        std::vector<Jac<Fq>> buckets(((size_t)1 << c) - 1, Jac<Fq>::zero());
        for (size_t i = lo; i < hi; ++i) {
            const u64* s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
            if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
                if (w_start == 0) res.add_assign_mixed(bases[i]);
                continue;
            }
            int limb = w_start / 64, off = w_start % 64;
            u64 v = s[limb] >> off;
            if (off && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
            u64 d = v & (((u64)1 << c) - 1);
            if (d) buckets[d - 1].add_assign_mixed(bases[i]);
        }
        Jac<Fq> running = Jac<Fq>::zero();
        for (size_t b = buckets.size(); b-- > 0;) { running.add_assign(buckets[b]); res.add_assign(running); }
        part_sums[t] = res;
    }
    Jac<Fq> total = Jac<Fq>::zero();
    for (int wi = W - 1; wi >= 0; --wi) {
        if (wi != W - 1) for (int k = 0; k < c; ++k) total.double_in_place();
        for (int pi = 0; pi < parts; ++pi) total.add_assign(part_sums[(size_t)wi * parts + pi]);
    }
    return total;
}

template <class Fq>
static void load_bases(std::vector<Affine<Fq>>& v, const u64* xy, const uint8_t* inf, size_t n) {
    v.resize(n);
    const int L = sizeof(Fq) / 8;
    for (size_t i = 0; i < n; ++i) {
        memcpy(v[i].x.v, xy + (2 * i) * L, sizeof(Fq));
        memcpy(v[i].y.v, xy + (2 * i + 1) * L, sizeof(Fq));
        v[i].inf = inf ? inf[i] != 0 : false;
    }
}
template <class Fq>
static void store_affine(const Affine<Fq>& a, u64* out_xy, uint8_t* out_inf) {
    memcpy(out_xy, a.x.v, sizeof(Fq));
    memcpy(out_xy + sizeof(Fq) / 8, a.y.v, sizeof(Fq));
    if (out_inf) *out_inf = a.inf ? 1 : 0;
}

template <class Fq, class Fr>
static int msm_run(const u64* xy, const uint8_t* inf, const u64* scalars, size_t n, u64* out_xy, uint8_t* out_inf, int threads) {
    std::vector<Affine<Fq>> bases;
    load_bases<Fq>(bases, xy, inf, n);
    Jac<Fq> r = n ? msm_pippenger<Fq, Fr>(bases.data(), scalars, n, threads) : Jac<Fq>::zero();
    store_affine<Fq>(r.into_affine(), out_xy, out_inf);
    return 0;
}

// scalar multiplication by a canonical scalar (double-and-add), used to build tau^i G style SRS
template <class Fq>
static Jac<Fq> scalar_mul(const Affine<Fq>& base, const u64* k, int nlimbs) {
    Jac<Fq> r = Jac<Fq>::zero();
    for (int i = nlimbs - 1; i >= 0; --i)
        for (int b = 63; b >= 0; --b) {
            r.double_in_place();
            if ((k[i] >> b) & 1) r.add_assign_mixed(base);
        }
    return r;
}

template <class Fq, class Fr>
static int srs_run(const CurveDesc& cd, const u64* tau_canonical, size_t n, u64* out_xy) {
    Affine<Fq> G; G.x = Fq::from_canonical(cd.gx); G.y = Fq::from_canonical(cd.gy); G.inf = false;
    Fr tau = Fr::from_canonical(tau_canonical);
    std::vector<Fr> pw(n);
    Fr cur = Fr::one();
    for (size_t i = 0; i < n; ++i) { pw[i] = cur; cur = cur * tau; }
    const int L = sizeof(Fq) / 8;
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; ++i) {
        u64 k[4]; pw[i].to_canonical(k);
        Affine<Fq> a = scalar_mul<Fq>(G, k, 4).into_affine();
        memcpy(out_xy + 2 * i * L, a.x.v, sizeof(Fq));
        memcpy(out_xy + (2 * i + 1) * L, a.y.v, sizeof(Fq));
    }
    return 0;
}

template <class Fq, class Fr>
static int kzg_commit_run(const u64* powers_xy, size_t n_powers, const u64* coeffs_mont, size_t n, u64* out_xy, uint8_t* out_inf, int threads) {
    // skip_leading_zeros_and_convert_to_bigints + MSM over powers_of_g[lz..]
    size_t lz = 0;
    const Fr* c = (const Fr*)coeffs_mont;
    while (lz < n && c[lz].is_zero()) ++lz;
    size_t m = n - lz;
    if (m > n_powers - std::min(n_powers, lz)) m = n_powers > lz ? n_powers - lz : 0;
    std::vector<u64> sc(4 * (m ? m : 1));
    #pragma omp parallel for schedule(static) if (m >= 4096)
    for (size_t i = 0; i < m; ++i) c[lz + i].to_canonical(&sc[4 * i]);
    const int L = sizeof(Fq) / 8;
    return msm_run<Fq, Fr>(powers_xy + 2 * lz * L, nullptr, sc.data(), m, out_xy, out_inf, threads);
}

// witness polynomial (p(X) - p(z)) / (X - z), Montgomery in/out; out has n-1 coefficients
template <class Fr>
static void witness_run(const u64* coeffs_mont, size_t n, const u64* z_mont, u64* out_mont) {
    if (n <= 1) return;
    const Fr* c = (const Fr*)coeffs_mont; Fr* w = (Fr*)out_mont;
    Fr z; memcpy(z.v, z_mont, sizeof z.v);
    Fr acc = Fr::zero();
    for (size_t i = n - 1; i >= 1; --i) { acc = c[i] + acc * z; w[i - 1] = acc; }
}

// ------------------------------------------------------------------ C entry points (ctypes)
extern "C" {

int ora_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ora_set_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

int ora_ntt(int curve_id, int kind, int log_n, const u64* in, size_t in_len, u64* out) {
    ensure_init();
    if (curve_id == 0) return ntt_run<FrBls>(CURVES[0], kind, log_n, in, in_len, out);
    if (curve_id == 1) return ntt_run<FrBn>(CURVES[1], kind, log_n, in, in_len, out);
    return -1;
}

// all-cores shape (msm_pippenger_chunked): `parts` point ranges per window
int ora_msm_g1_chunked(int curve_id, const u64* bases_xy, const uint8_t* inf, const u64* scalars, size_t n,
                       u64* out_xy, uint8_t* out_inf, int threads, int parts) {
    ensure_init();
    if (curve_id == 0) {
        std::vector<Affine<FqBls>> bases;
        load_bases<FqBls>(bases, bases_xy, inf, n);
        Jac<FqBls> r = n ? msm_pippenger_chunked<FqBls, FrBls>(bases.data(), scalars, n, threads, parts) : Jac<FqBls>::zero();
        store_affine<FqBls>(r.into_affine(), out_xy, out_inf);
        return 0;
    }
    if (curve_id == 1) {
        std::vector<Affine<FqBn>> bases;
        load_bases<FqBn>(bases, bases_xy, inf, n);
        Jac<FqBn> r = n ? msm_pippenger_chunked<FqBn, FrBn>(bases.data(), scalars, n, threads, parts) : Jac<FqBn>::zero();
        store_affine<FqBn>(r.into_affine(), out_xy, out_inf);
        return 0;
    }
    return -1;
}

int ora_msm_g1(int curve_id, const u64* bases_xy, const uint8_t* inf, const u64* scalars, size_t n,
               u64* out_xy, uint8_t* out_inf, int threads) {
    ensure_init();
    if (curve_id == 0) return msm_run<FqBls, FrBls>(bases_xy, inf, scalars, n, out_xy, out_inf, threads);
    if (curve_id == 1) return msm_run<FqBn, FrBn>(bases_xy, inf, scalars, n, out_xy, out_inf, threads);
    return -1;
}

int ora_srs_powers(int curve_id, const u64* tau_canonical, size_t n, u64* out_xy) {
    ensure_init();
    if (curve_id == 0) return srs_run<FqBls, FrBls>(CURVES[0], tau_canonical, n, out_xy);
    if (curve_id == 1) return srs_run<FqBn, FrBn>(CURVES[1], tau_canonical, n, out_xy);
    return -1;
}

int ora_kzg_commit(int curve_id, const u64* powers_xy, size_t n_powers, const u64* coeffs_mont, size_t n,
                   u64* out_xy, uint8_t* out_inf, int threads) {
    ensure_init();
    if (curve_id == 0) return kzg_commit_run<FqBls, FrBls>(powers_xy, n_powers, coeffs_mont, n, out_xy, out_inf, threads);
    if (curve_id == 1) return kzg_commit_run<FqBn, FrBn>(powers_xy, n_powers, coeffs_mont, n, out_xy, out_inf, threads);
    return -1;
}

int ora_kzg_witness(int curve_id, const u64* coeffs_mont, size_t n, const u64* z_mont, u64* out_mont) {
    ensure_init();
    if (curve_id == 0) { witness_run<FrBls>(coeffs_mont, n, z_mont, out_mont); return 0; }
    if (curve_id == 1) { witness_run<FrBn>(coeffs_mont, n, z_mont, out_mont); return 0; }
    return -1;
}

// which: 0 = Fr, 1 = Fq ; dir: 0 = canonical -> Montgomery, 1 = Montgomery -> canonical
int ora_convert(int curve_id, int which, int dir, const u64* in, size_t n, u64* out) {
    ensure_init();
#define CONV(F)                                                                           \
    do {                                                                                  \
        const int L = sizeof(F) / 8;                                                      \
        for (size_t i = 0; i < n; ++i) {                                                  \
            if (dir == 0) { F v = F::from_canonical(in + i * L); memcpy(out + i * L, v.v, sizeof(F)); } \
            else { F v; memcpy(v.v, in + i * L, sizeof(F)); v.to_canonical(out + i * L); } \
        }                                                                                 \
        return 0;                                                                         \
    } while (0)
    if (curve_id == 0 && which == 0) CONV(FrBls);
    if (curve_id == 0 && which == 1) CONV(FqBls);
    if (curve_id == 1 && which == 0) CONV(FrBn);
    if (curve_id == 1 && which == 1) CONV(FqBn);
#undef CONV
    return -1;
}

// Fr pointwise helpers used by property tests (Montgomery in/out): op 0 mul, 1 add, 2 sub
int ora_fr_op(int curve_id, int op, const u64* a, const u64* b, size_t n, u64* out) {
    ensure_init();
#define FROP(F)                                                              \
    do {                                                                     \
        const F* x = (const F*)a; const F* y = (const F*)b; F* o = (F*)out;  \
        for (size_t i = 0; i < n; ++i) o[i] = op == 0 ? x[i] * y[i] : op == 1 ? x[i] + y[i] : x[i] - y[i]; \
        return 0;                                                            \
    } while (0)
    if (curve_id == 0) FROP(FrBls);
    if (curve_id == 1) FROP(FrBn);
#undef FROP
    return -1;
}

int ora_window_size(size_t n) { return ark_window(n); }

}  // extern "C"
