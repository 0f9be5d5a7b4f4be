// MSM unit 2 of 4 (msm_common.cuh): the accumulation of the sorted references into buckets -- the kernel a proof spends 78 % of its
// time in -- and what builds its inputs: the window-multiples table of an SRS, the conversion of arkworks-layout bases
// into the device form, the fixed-base utility.
// Device arithmetic is the signed 30-bit-limb Montgomery field of fields.cuh with the lazy XYZZ group law of ecu.cuh; bases are
// converted once, at SRS registration, into that form.  The group sum is order-independent, so the arbitrary order inside a bucket
// does not change the (canonical, affine) result.  Algorithmic bytes per MSM: N * (32 + 2 * Fq bytes); the kernel is integer-VALU bound.
#include "msm_common.cuh"

namespace {

// Every lane sums entries [t*L, (t+1)*L) of the bucket-sorted reference list.
// PRE: references carry a window number and `bases` is the window-multiples table [W][n_srs]
// (row w holds 2^(c w) P_i); tab_stride = n_srs, tab_off = base_offset.
template <class F, bool PRE>
ZK_D void accumulate_chunk(const uint32_t t, const uint32_t* entries, const uint32_t* offsets, uint32_t nb, const void* bases, void* buckets,
                           void* part_pt, uint32_t L0, uint32_t n_lanes, uint64_t tab_stride, uint64_t tab_off) {
    const uint32_t E = offsets[nb];
    const uint32_t L = chunk_len(E, n_lanes, L0);
    const uint64_t e0 = (uint64_t)t * L;
    if (e0 >= E) return;
    const uint32_t e1 = (uint32_t)min((uint64_t)E, e0 + L);
    // largest b with offsets[b] <= e0
    uint32_t lo = 0, hi = nb - 1;
    while (lo < hi) {
        uint32_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= (uint32_t)e0) lo = mid; else hi = mid - 1;
    }
    uint32_t b = lo;
    uint32_t bend = offsets[b + 1];
    const bool head_partial = offsets[b] < (uint32_t)e0;
    bool first_run = true;
    XYZZu<F> acc = XYZZu<F>::infinity();
    // software pipeline: the reference and the 128-byte point of iteration e+1 are requested before the
    // mixed addition of iteration e (two dependent HBM/L2 round trips otherwise sit in front of every add)
    auto point_index = [&](uint32_t ref) -> uint64_t {
        return PRE ? (uint64_t)((ref >> 26) & 31u) * tab_stride + tab_off + (ref & 0x3ffffffu) : (uint64_t)(ref & 0x7fffffffu);
    };
    uint32_t ref_n = entries[(uint32_t)e0];
    AffineU<F> p_n = ld_affine<F>(bases, point_index(ref_n));
    for (uint32_t e = (uint32_t)e0; e < e1; ++e) {
        const uint32_t ref = ref_n;
        AffineU<F> p = p_n;
        if (e + 1 < e1) {
            ref_n = entries[e + 1];
            p_n = ld_affine<F>(bases, point_index(ref_n));
        }
        if (e == bend) {
            if (first_run && head_partial) st_xyzz<F>(part_pt, 2ull * t, acc);
            else st_xyzz<F>(buckets, b, acc);
            first_run = false;
            acc = XYZZu<F>::infinity();
            do {
                ++b;
                bend = offsets[b + 1];
            } while (bend <= e);
        }
        if (p.is_null()) continue;
        if (ref >> 31) p.y = F::neg_canonical(p.y);
        acc = XYZZu<F>::madd(acc, p);
    }
    // Slot convention (msm_combine relies on it): a run that is the FIRST run of its chunk and is
    // not a whole bucket goes to slot 2t, a trailing incomplete run that is not the first goes to 2t+1.
    const bool tail_complete = (e1 == bend);
    if (first_run) {
        if (head_partial || !tail_complete) st_xyzz<F>(part_pt, 2ull * t, acc);
        else st_xyzz<F>(buckets, b, acc);
    } else {
        if (tail_complete) st_xyzz<F>(buckets, b, acc);
        else st_xyzz<F>(part_pt, 2ull * t + 1, acc);
    }
}

template <class F, bool PRE>
__global__ void __launch_bounds__(128) msm_accumulate(const uint32_t* entries, const uint32_t* offsets, uint32_t nb, const void* bases,
                                                       void* buckets, void* part_pt, uint32_t L, uint32_t n_lanes, uint64_t tab_stride,
                                                       uint64_t tab_off) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_lanes) return;
    accumulate_chunk<F, PRE>(t, entries, offsets, nb, bases, buckets, part_pt, L, n_lanes, tab_stride, tab_off);
}

// The accumulations of a round's jobs as ONE launch over the window table: a launch ends with CUs waiting for their last
// wavefronts (every lane does the same work, so the waves of the last round of resident lanes finish within a fraction of a
// chunk of each other and the chip idles for that fraction), and a launch per job pays that at the end of every job.  Here the
// lanes of job k+1 follow those of job k without a gap: only the LAST job of the launch is cut into short chunks (several rounds
// of resident lanes), the others get one long chunk per resident lane -- a third of the chunk-edge partials for msm_combine*.
template <class F>
__global__ void __launch_bounds__(128) msm_accumulate_batch(AJobs jobs, uint32_t nb, const void* bases, uint64_t tab_stride) {
    uint32_t k = 0;
    while (k + 1 < jobs.n && blockIdx.x >= jobs.j[k + 1].blk0) ++k;       // uniform: scalar registers
    const AJob& J = jobs.j[k];
    const uint32_t t = (blockIdx.x - J.blk0) * blockDim.x + threadIdx.x;
    if (t >= J.n_lanes) return;
    accumulate_chunk<F, true>(t, J.entries, J.offsets, nb, bases, J.buckets, J.part_pt, J.L0, J.n_lanes, tab_stride, J.tab_off);
}

template <class Cv>
__global__ void bases_to_internal(const uint32_t* xy_sat, const uint8_t* inf, uint64_t n, void* out) {
    typedef typename Cv::FqU F;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = xy_sat + i * 2 * F::SAT;
    uint32_t any_x = 0, y_not_zero = 0, y_not_one = 0;
    for (int k = 0; k < F::SAT; ++k) {
        any_x |= w[k];
        y_not_zero |= w[F::SAT + k];
        y_not_one |= w[F::SAT + k] ^ Cv::FqP::R(k);
    }
    uint4* q = reinterpret_cast<uint4*>(out) + i * (2 * Store<F>::U4);
    if ((any_x == 0 && (y_not_zero == 0 || y_not_one == 0)) || (inf && inf[i])) {
        st_fu<F>(q, F::zero());
        st_fu<F>(q + Store<F>::U4, F::zero());
        return;
    }
    st_fu<F>(q, F::canonical_lt2p(F::from_sat(w)));
    st_fu<F>(q + Store<F>::U4, F::canonical_lt2p(F::from_sat(w + F::SAT)));
}

// out[i] = scalars[i] * G  (double-and-add from the top bit; per-lane Fermat inversion to affine),
// written in the arkworks layout; infinity -> x = y = 0 words.
template <class Cv>
__global__ void __launch_bounds__(128) g1_fixed_base(const uint32_t* scalars, uint64_t n, uint32_t* out_xy) {
    typedef typename Cv::FqU F;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t gw[2 * F::SAT];
#pragma unroll
    for (int k = 0; k < F::SAT; ++k) {
        gw[k] = Cv::FqP::GX(k);
        gw[F::SAT + k] = Cv::FqP::GY(k);
    }
    AffineU<F> G;
    G.x = F::canonical_lt2p(F::from_sat(gw));
    G.y = F::canonical_lt2p(F::from_sat(gw + F::SAT));
    const uint32_t* s = scalars + 8 * i;
    XYZZu<F> acc = XYZZu<F>::infinity();
    for (int limb = 7; limb >= 0; --limb) {
        const uint32_t word = s[limb];
        for (int b = 31; b >= 0; --b) {
            acc = XYZZu<F>::dbl(acc);
            if ((word >> b) & 1u) acc = XYZZu<F>::madd(acc, G);
        }
    }
    AffineU<F> o;
    uint32_t* dst = out_xy + i * 2 * F::SAT;
    if (!acc.to_affine(o)) {
        for (int k = 0; k < 2 * F::SAT; ++k) dst[k] = 0;
        return;
    }
    o.x.to_sat(dst);
    o.y.to_sat(dst + F::SAT);
}

// table[w][i] = 2^(c w) * P_i for w = 1 .. W-1 (row 0 = the points themselves), affine internal form.
// One inversion per row and point (the round-1 kernel) made the table build 140 ms per 2^20 points, nearly all of it Fermat
// inversions.  Here the rows are produced RB at a time: c doublings per row in XYZZ (X, Y parked in the table row itself; ZZ, ZZZ
// and the running product of the ZZZ in `scratch`, RB x n x 3 field elements), ONE inversion of the product, and a backward sweep
// that peels off every 1/ZZZ_j (Montgomery's trick along the chain): 6 products per row + 1/RB of an inversion on top of the doublings.
constexpr uint32_t CHAIN_RB = 32;
// rows: table rows this launch computes, written at table rows out0, out0 + 1, ...; the chain starts from the point src[i] and
// takes d0 doublings to the first computed row and dstep between rows.  Whole table: src = row 0 (the SRS itself), out0 = 1,
// d0 = dstep = c.  Window-sharded table (rows first, first + stride, ...): d0 = c * first (row 0 is the copy of the SRS when
// first = 0, so out0 = 1 and d0 = dstep there), dstep = c * stride.
template <class F>
__global__ void __launch_bounds__(128) msm_precompute(void* table, const void* src, uint64_t n, uint32_t d0, uint32_t dstep, uint32_t rows, uint32_t out0,
                                                      void* scratch) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int U4 = Store<F>::U4;
    uint4* base = reinterpret_cast<uint4*>(table);
    uint4* scr = reinterpret_cast<uint4*>(scratch);
    auto row = [&](uint32_t r) { return base + ((uint64_t)(out0 + r) * n + i) * (2 * U4); };
    auto sc = [&](uint32_t j, uint32_t which) { return scr + (((uint64_t)j * n + i) * 3 + which) * U4; };     // 0 ZZ, 1 ZZZ, 2 prefix product
    AffineU<F> p = ld_affine<F>(src, i);
    if (p.is_null()) {
        for (uint32_t r = 0; r < rows; ++r) {
            st_fu<F>(row(r), F::zero());
            st_fu<F>(row(r) + U4, F::zero());
        }
        return;
    }
    XYZZu<F> acc = XYZZu<F>::from_affine(p);
    for (uint32_t r0 = 0; r0 < rows; r0 += CHAIN_RB) {
        const uint32_t m = rows - r0 < CHAIN_RB ? rows - r0 : CHAIN_RB;
        F prefix = F::one();
        for (uint32_t j = 0; j < m; ++j) {
            const uint32_t nd = r0 + j == 0 ? d0 : dstep;
            for (uint32_t k = 0; k < nd; ++k) acc = XYZZu<F>::dbl(acc);   // a point of odd prime order never doubles to infinity
            st_fu<F>(row(r0 + j), acc.x);
            st_fu<F>(row(r0 + j) + U4, acc.y);
            st_fu<F>(sc(j, 0), acc.zz);
            st_fu<F>(sc(j, 1), acc.zzz);
            prefix = F::mul(prefix, acc.zzz);
            st_fu<F>(sc(j, 2), prefix);
        }
        F inv = F::inverse(prefix);             // 1 / (ZZZ_0 ... ZZZ_(m-1))
        for (uint32_t j = m; j-- > 0;) {
            const F zzz = ld_fu<F>(sc(j, 1));
            const F i3 = j ? F::mul(inv, ld_fu<F>(sc(j - 1, 2))) : inv;     // 1 / ZZZ_j
            inv = F::mul(inv, zzz);
            const F zi = F::mul(ld_fu<F>(sc(j, 0)), i3);                    // ZZ / ZZZ = 1 / Z
            const F zi2 = F::sqr(zi);
            const F x = F::canonical_lt2p(F::mul(ld_fu<F>(row(r0 + j)), zi2));
            const F y = F::canonical_lt2p(F::mul(ld_fu<F>(row(r0 + j) + U4), i3));
            st_fu<F>(row(r0 + j), x);
            st_fu<F>(row(r0 + j) + U4, y);
        }
    }
}

template <class Cv>
int msm_precompute_run(zk_ctx* c, zk_srs* s, uint32_t window_bits, uint32_t w0, uint32_t wstep) {
    typedef typename Cv::FqU F;
    // default window: 16 bits (16 rows, 2^15 buckets) below 2^19 points; from there on 17 bits, which scalars folded to
    // k <= (r - 1) / 2 (MsmGeom::neg) cover in 15 windows -- one mixed addition per scalar fewer for twice the buckets to reduce
    // (measured, profiles/r03/r03_notes.md: 2^18 27.1 / 29.2 ms per proof at c = 16 / 17, 2^19 48.1 / 47.3, 2^20 87.4 / 85.6, 2^22 364.6 / 353.2)
    // From 2^22 points on a whole table takes 20 bits (13 rows, 2^19 buckets): the 2 x 2^19 additions of the wide bucket reduction are
    // then fewer than the two additions per scalar they save (measured, profiles/r05/r05_sweep_window.txt: c = 17 / 20 at 2^21 153.9 / 153.3 ms
    // per proof, 2^22 300.6 / 287.3, 2^23 599.4 / 551.5; at 2^20 79.3 / 87.4 -- the other way round).  A window-sharded table keeps 17:
    // its rank accumulates only W / G rows into the same number of buckets.
    if (window_bits == 0) window_bits = (s->n >= (1u << 22) && wstep == 1) ? 20 : s->n >= (1u << 19) ? PRE_C + 1 : PRE_C;
    if (window_bits < PRE_C || window_bits > PRE_C_MAX) return ZK_ERR_BAD_ARG;
    MsmGeom g = make_geom<typename Cv::FrP>(1u << 20, (int)window_bits, PRE_C_MAX);
    if (wstep == 0 || w0 >= wstep || w0 >= g.W) return ZK_ERR_BAD_ARG;
    const uint32_t rows = (g.W - w0 + wstep - 1) / wstep;         // windows w0, w0 + wstep, ... < W
    const bool whole = wstep == 1;
    const size_t pb = s->point_bytes;
    void* tab = nullptr;
    void* scratch = nullptr;
    if (hipMalloc(&tab, (size_t)rows * s->n * pb) != hipSuccess) {
        (void)hipGetLastError();
        return ZK_ERR_OOM;
    }
    // row 0 is a copy of the SRS when the first owned window is window 0; the chain computes the others
    const uint32_t out0 = w0 == 0 ? 1u : 0u, chain_rows = rows - out0;
    const uint32_t rb = chain_rows < CHAIN_RB ? chain_rows : CHAIN_RB;
    if (hipMalloc(&scratch, (size_t)(rb ? rb : 1) * s->n * 3 * (pb / 2)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(tab);
        return ZK_ERR_OOM;
    }
    hipError_t e = hipSuccess;
    if (w0 == 0) e = hipMemcpyAsync(tab, s->d_xy, s->n * pb, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess && chain_rows) {
        const int T = 128;
        unsigned blocks = (unsigned)((s->n + T - 1) / T);
        hipLaunchKernelGGL(msm_precompute<F>, dim3(blocks), dim3(T), 0, c->stream, tab, (const void*)s->d_xy, (uint64_t)s->n,
                           w0 == 0 ? g.c * wstep : g.c * w0, g.c * wstep, chain_rows, out0, scratch);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(scratch);
    if (e != hipSuccess) {
        (void)hipFree(tab);
        zk_note_hip_error(e, "msm_precompute", __FILE__, __LINE__);
        return ZK_ERR_HIP;
    }
    if (whole) {
        (void)hipFree(s->d_xy);
        s->d_xy = tab;      // row 0 of the table is the SRS itself
    } else {
        s->d_pre = tab;     // the rank's rows; d_xy stays the plain SRS (vectors too short for the table path)
    }
    s->pre_c = g.c;
    s->pre_W = g.W;
    s->pre_rows = rows;
    s->pre_w0 = w0;
    s->pre_wstep = wstep;
    return ZK_OK;
}


// per-window path: one launch over the W windows' lists, CHUNK_L references per lane
template <class Cv>
int pw_queue_accumulate(zk_ctx* c, const MsmGeom& g, MsmBufs& mb, const void* d_bases, uint32_t n_lanes, hipStream_t st) {
    typedef typename Cv::FqU F;
    ProfScope ps(c, "msm_accumulate", st);
    const int T = 128;
    unsigned blocks = (n_lanes + T - 1) / T;
    hipLaunchKernelGGL((msm_accumulate<F, false>), dim3(blocks), dim3(T), 0, st, (const uint32_t*)mb.entries.p, (const uint32_t*)mb.offsets.p, g.nb, d_bases,
                       mb.buckets.p, mb.part_pt.p, CHUNK_L, n_lanes, (uint64_t)0, (uint64_t)0);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// the accumulations of n_jobs sorted jobs as ONE launch (msm_accumulate_batch); tab_offs[k] = base_offset of job k
template <class Cv>
int pre_queue_accumulate(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, const size_t* lens, const size_t* tab_offs, uint32_t n_jobs, zk_srs* s,
                         hipStream_t st) {
    typedef typename Cv::FqU F;
    if (n_jobs == 0) return ZK_OK;
    if (n_jobs > (uint32_t)MAX_JOBS) return ZK_ERR_UNSUPPORTED;
    ProfScope ps(c, "msm_accumulate", st);
    const uint32_t T = 128;
    AJobs aj;
    memset(&aj, 0, sizeof aj);
    aj.n = n_jobs;
    uint64_t blocks = 0, points = 0;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        MsmBufs& mb = *mbs[k];
        AJob& J = aj.j[k];
        J.entries = (const uint32_t*)mb.entries.p;
        J.offsets = (const uint32_t*)mb.offsets.p;
        J.buckets = mb.buckets.p;
        J.part_pt = mb.part_pt.p;
        J.tab_off = tab_offs ? tab_offs[k] : 0;
        J.L0 = pls[k].chunk_l;
        J.n_lanes = pls[k].n_lanes;
        mb.acc_chunk_l = pls[k].chunk_l;        // the reductions find the chunk-edge partials through these two
        mb.acc_n_lanes = pls[k].n_lanes;
        J.blk0 = (uint32_t)blocks;
        blocks += (pls[k].n_lanes + T - 1) / T;
        points += lens[k];
    }
    if (blocks >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
    if (c->profiling) {      // units of the scope above: bench.py prices a launch by the points it processed
        c->prof["msm_accumulate_jobs"].launches += n_jobs;
        c->prof["msm_accumulate_points"].launches += points;
    }
    hipLaunchKernelGGL(msm_accumulate_batch<F>, dim3((unsigned)blocks), dim3(T), 0, st, aj, pls[0].g1.nb, s->table(), (uint64_t)s->n);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

}  // namespace

int ZK_SYM(pw_queue_accumulate)(zk_ctx* c, const MsmGeom& g, MsmBufs& mb, const void* d_bases, uint32_t n_lanes, hipStream_t st) {
    return pw_queue_accumulate<CurveSel>(c, g, mb, d_bases, n_lanes, st);
}
int ZK_SYM(pre_queue_accumulate)(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, const size_t* lens, const size_t* tab_offs, uint32_t n_jobs, zk_srs* s,
                                 hipStream_t st) {
    return pre_queue_accumulate<CurveSel>(c, pls, mbs, lens, tab_offs, n_jobs, s, st);
}

int ZK_SYM(msm_fixed_base_dev)(zk_ctx* c, const void* d_scalars, size_t n, void* d_out_xy) {
    if (n == 0) return ZK_OK;
    const int T = 128;
    unsigned blocks = (unsigned)((n + T - 1) / T);
    hipLaunchKernelGGL(g1_fixed_base<CurveSel>, dim3(blocks), dim3(T), 0, c->stream, (const uint32_t*)d_scalars, (uint64_t)n,
                       (uint32_t*)d_out_xy);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

int ZK_SYM(msm_convert_bases_dev)(zk_ctx* c, const void* d_xy_sat, const uint8_t* d_inf, size_t n, void* d_out_internal) {
    if (n == 0) return ZK_OK;
    const int T = 256;
    unsigned blocks = (unsigned)((n + T - 1) / T);
    hipLaunchKernelGGL(bases_to_internal<CurveSel>, dim3(blocks), dim3(T), 0, c->stream, (const uint32_t*)d_xy_sat, d_inf, (uint64_t)n,
                       d_out_internal);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

int ZK_SYM(msm_precompute_dev)(zk_ctx* c, zk_srs* s, uint32_t window_bits, uint32_t w0, uint32_t wstep) {
    return msm_precompute_run<CurveSel>(c, s, window_bits, w0, wstep);
}

size_t ZK_SYM(msm_point_bytes)() { return (size_t)2 * Store<CurveSel::FqU>::WORDS * 4; }
size_t ZK_SYM(msm_partial_dev_bytes)() { return (size_t)4 * Store<CurveSel::FqU>::WORDS * 4; }
