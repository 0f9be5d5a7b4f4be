// Pippenger (bucket method) multi-scalar multiplication over G1 for gfx950 (MI355X).
//
// Replaces, behind the C ABI, ark_ec 0.3 `VariableBaseMSM::multi_scalar_mul` as called by the
// reference at plonk-core/src/commitment.rs:45,83 and through every `PC::commit` / `PC::open`
// (proof_system/prover.rs:213,289-291,312-317,361-363,387-389,459-469,579,582-591,606,609-618).
//
// MSM unit 4 of 4 (msm_common.cuh has the pipeline): the host side -- window geometry, the plan and memory budget of a window-table job,
// the per-window entry point (msm_run), the table entry points (msm_run_pre, the batch and the deferred rounds' begin / reduce / end),
// the host combine of the window sums and the affine normalisation.  No kernel is defined here: the units msm_sort / msm_accumulate /
// msm_reduce queue them (ZK_SYM entry points of msm_common.cuh).
#include "msm_common.cuh"

#include <chrono>

namespace {

int ensure_pinned(zk_ctx* c, size_t bytes);
template <class Fq>
XYZZ<Fq> jac_to_xyzz(const uint64_t* xyz);

template <class Cv>
int msm_run(zk_ctx* c, const void* d_bases, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    typedef typename Cv::Fq Fq;       // host / arkworks-layout arithmetic
    typedef typename Cv::FqU F;       // device arithmetic
    typedef XYZZ<Fq> PH;
    constexpr int L64 = Fq::N / 2;
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;   // bytes of one stored XYZZu
    if (n == 0) {
        // Jacobian zero as arkworks writes it: (1, 1, 0)
        Fq one = Fq::one(), zero = Fq::zero();
        memcpy(out_xyz, one.v, sizeof(uint64_t) * L64);
        memcpy(out_xyz + L64, one.v, sizeof(uint64_t) * L64);
        memcpy(out_xyz + 2 * L64, zero.v, sizeof(uint64_t) * L64);
        return ZK_OK;
    }
    if (n >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
    MsmGeom g = make_geom<typename Cv::FrP>(n, c->msm_window);
    const uint64_t e_max = (uint64_t)n * g.W;
    if (e_max >= (1ull << 32)) return ZK_ERR_UNSUPPORTED;
    const uint32_t n_lanes = (uint32_t)((e_max + CHUNK_L - 1) / CHUNK_L);
    int rc;
    MsmBufs& mb = c->mb[0];
    // slabs of scalars per window: enough workgroups to fill the chip, each with >= ~8K scalars
    uint32_t S = 1;
    while (S < 32 && (uint64_t)S * 16384 < n && (uint64_t)(2 * S) * g.W <= 1024) S <<= 1;
    const unsigned nblk_scan = (g.nb + 1023) / 1024;
    if (nblk_scan > 1024) return ZK_ERR_UNSUPPORTED;
    if ((rc = mb.counts.ensure((size_t)g.W * S * g.B * 4 + 4096))) return rc;       // hist / cursors + block sums
    if ((rc = mb.offsets.ensure((size_t)(g.nb + 1) * 4))) return rc;
    if ((rc = mb.tmp.ensure((size_t)g.W * n * 2))) return rc;                       // int16 digits, window-major
    if ((rc = mb.entries.ensure((size_t)e_max * 4))) return rc;
    if ((rc = mb.buckets.ensure((size_t)g.nb * PT))) return rc;
    if ((rc = mb.part_pt.ensure((size_t)n_lanes * 2 * PT))) return rc;
    if ((rc = mb.part_key.ensure((size_t)(g.nb + 2) * 4))) return rc;   // combine queues: [n_medium, n_large, ids...]
    if ((rc = mb.seg.ensure((size_t)g.W * g.ns * 2 * PT))) return rc;
    if ((rc = mb.win.ensure((size_t)g.W * sizeof(PH)))) return rc;
    hipStream_t st = c->stream;
    if ((rc = ZK_SYM(pw_queue_sort)(c, g, S, mb, d_scalars, n, st))) return rc;
    if ((rc = ZK_SYM(pw_queue_accumulate)(c, g, mb, d_bases, n_lanes, st))) return rc;
    {
        RJobs jobs;
        memset(&jobs, 0, sizeof jobs);
        jobs.part_pt[0] = mb.part_pt.p;
        jobs.offsets[0] = (const uint32_t*)mb.offsets.p;
        jobs.buckets[0] = mb.buckets.p;
        jobs.q[0] = (uint32_t*)mb.part_key.p;
        jobs.seg_run[0] = mb.seg.p;
        jobs.seg_acc[0] = (char*)mb.seg.p + (size_t)g.W * g.ns * PT;
        jobs.win_s[0] = (uint32_t*)mb.win.p;
        jobs.win_t[0] = nullptr;
        jobs.L[0] = CHUNK_L;
        jobs.lanes[0] = n_lanes;
        jobs.nbk[0] = g.nb;
        if ((rc = ZK_SYM(queue_reduce)(c, jobs, 1, g.nb, g, st, false, 0u))) return rc;
    }
    // window sums -> host, Horner (high window first), Jacobian out
    std::vector<PH> win(g.W);
    ZK_HIP_TRY(hipMemcpyAsync(win.data(), mb.win.p, (size_t)g.W * sizeof(PH), hipMemcpyDeviceToHost, st));
    ZK_HIP_TRY(hipStreamSynchronize(st));
    PH total = PH::infinity();
    for (int w = (int)g.W - 1; w >= 0; --w) {
        for (uint32_t k = 0; k < g.c; ++k) total = PH::dbl(total);
        total = PH::add(total, win[w]);
    }
    // XYZZ -> Jacobian (X*ZZ, Y*ZZZ, ZZ):  x = X/ZZ = X*ZZ/ZZ^2, y = Y/ZZZ = Y*ZZZ/ZZ^3
    Fq X = Fq::one(), Y = Fq::one(), Z = Fq::zero();
    if (!total.is_inf()) {
        X = Fq::mul(total.x, total.zz);
        Y = Fq::mul(total.y, total.zzz);
        Z = total.zz;
    }
    memcpy(out_xyz, X.v, sizeof(uint64_t) * L64);
    memcpy(out_xyz + L64, Y.v, sizeof(uint64_t) * L64);
    memcpy(out_xyz + 2 * L64, Z.v, sizeof(uint64_t) * L64);
    return ZK_OK;
}

// ---- MSM over a precomputed SRS -------------------------------------------------------------------
// Every (scalar, window) digit is a reference to table[w][i]; all windows share one set of 2^(c-1)
// buckets.  The work of one MSM is queued in two pieces so that a batch can pipeline them:
//   sort   (digits, LDS counting sort)        -- light kernels, few VGPRs: they co-reside with the
//                                                 accumulate waves of the PREVIOUS MSM (aux stream)
//   heavy  (accumulate, combine, reduction, read-back of the virtual-window sums)   (main stream)
// the reduction's view of the shared bucket set: a function of the table's window (pl.g.B buckets) and the ctx's options only
inline void pre_reduce_geom(const zk_ctx* c, PrePlan& pl) {
    pl.wide_red = pl.g.B > (1u << 16);
    pl.gv = pl.g;                                       // the reduction sees PRE_VW virtual windows (wide: windows of 512 buckets)
    pl.gv.W = pl.wide_red ? pl.g.B / WIDE_VB : PRE_VW;
    pl.gv.B = pl.g.B / pl.gv.W;
    pl.gv.nb = pl.g.B;
    // 64 chains (one wavefront per SIMD) per virtual window: segments of 8 buckets for windows of 512 (c = 16; measured against 4 / 16),
    // of 16 for windows of 1024 (c = 17; 64 x 16 measured against 128 x 8, and against 128 and 32 virtual windows)
    pl.gv.logG = pl.wide_red ? 2 : pl.gv.B >= 1024 ? 4 : 3;
    if (!pl.wide_red) {                                   // tuning options "pre_vw" / "pre_logg" (profiles/r02/r02_notes.md)
        if (c->tune.pre_vw) {
            const uint32_t v = (uint32_t)c->tune.pre_vw;
            if (v >= 8 && v <= 512 && (v & (v - 1)) == 0 && pl.g.B % v == 0) {
                pl.gv.W = v;
                pl.gv.B = pl.g.B / v;
            }
        }
        if (c->tune.pre_logg >= 0) {
            const uint32_t v = (uint32_t)c->tune.pre_logg;
            if (v <= 5 && (pl.gv.B >> v) >= 1) pl.gv.logG = v;
        }
    }
    pl.gv.ns = pl.gv.B >> pl.gv.logG;
    pl.gv.logq = 0;
    while ((256u << pl.gv.logq) < pl.gv.ns) ++pl.gv.logq;
}

// long_chunks: the job is followed by another one inside a merged accumulation launch (msm_accumulate_batch): one round of
// resident lanes with one long chunk each instead of several rounds of short ones.  The buffers are sized for the larger plan.
// pre_plan_geom derives the plan and allocates nothing; pre_sizes / pre_need / pre_ensure are its buffers.
template <class Cv>
int pre_plan_geom(const zk_ctx* c, const zk_srs* s, size_t n, PrePlan& pl, bool long_chunks = false) {
    typedef typename Cv::Fq Fq;
    typedef XYZZ<Fq> PH;
    if (n > zk_pre_max_n(c)) return ZK_ERR_UNSUPPORTED;      // the callers (api.hip) send longer vectors down the per-window path
    pl.g = make_geom<typename Cv::FrP>(n, (int)s->pre_c, PRE_C_MAX);
    if (pl.g.W != s->pre_W || pl.g.W > 32) return ZK_ERR_UNSUPPORTED;
    pl.g.w0 = s->pre_w0;                                // a window-sharded table: this rank's rows only (W of the Wt windows)
    pl.g.wstep = s->pre_wstep;
    pl.g.W = s->pre_rows;
    pl.g.nb = pl.g.W * pl.g.B;
    pl.wide = pl.g.c > 16;
    pl.nf = (uint64_t)n * pl.g.W;                       // flattened (window, scalar) digits
    if (pl.nf >= (1ull << 31)) return ZK_ERR_UNSUPPORTED;
    pl.g1 = pl.g;                                       // the sort sees ONE window of nf digits
    pl.g1.W = 1;
    pl.g1.nb = pl.g.B;
    pre_reduce_geom(c, pl);
    pl.shared_stage = pl.nf >= PRE_BIG_NF;
    // references per lane: as long as possible (fewer chunk-edge partials) while keeping >= 2 rounds of
    // resident lanes (256 CUs x 4 SIMDs x 2 waves x 64 = 131072 at the kernel's VGPR count), so that
    // lanes finishing early are replaced instead of idling through the tail (measured at 2^20:
    // L = 128 / 64 / 32 / 16 -> 111.8 / 110.3 / 110.8 / 162 ms per proof; 16 overloads the combine)
    pl.chunk_l = PRE_CHUNK_L;
    while (pl.chunk_l > 16 && pl.nf / pl.chunk_l < 196608) pl.chunk_l >>= 1;
    bool tuned = false;
    if (c->tune.chunk_l >= 8 && c->tune.chunk_l <= 1024) {          // tuning option "chunk_l" (profiles/r02/r02_notes.md)
        pl.chunk_l = (uint32_t)c->tune.chunk_l;
        tuned = true;
    }
    pl.n_lanes = (uint32_t)((pl.nf + pl.chunk_l - 1) / pl.chunk_l);
    // whole rounds of resident lanes: every lane does the same work, so 1.9 rounds take as long as 2 (15 windows of 2^20 digits
    // at 64 per lane are 245760 lanes): round the lane count up to a multiple of a round and shorten the chunks instead.
    // Two rounds become three where the chunks stay >= 32 references: the end of the launch, where CUs wait for their last
    // wavefronts, shortens with the chunk (2^20, c = 17: 60 -> 40 per lane, msm_accumulate -2.5 % per launch, msm_combine* +7
    // partials per bucket instead of 5, net +0.5 .. 0.8 % proofs/s; 30 and 24 per lane give the accumulation another 1 % and
    // the combine more than that back: profiles/r03/r03_notes.md).  A job that is not the last one of a merged launch has no end of
    // its own: one round (option "long_rounds": tuning hook), a third of the partials.
    pl.max_lanes = pl.n_lanes;
    {
        constexpr uint32_t ROUND = 131072;
        if (pl.n_lanes > ROUND) {
            uint32_t rounds = (pl.n_lanes + ROUND - 1) / ROUND;
            if (!tuned && rounds == 2 && pl.nf / (3ull * ROUND) >= 32) rounds = 3;
            if (!tuned && pl.nf >= PRE_BIG_NF && rounds > PRE_BIG_ROUNDS) rounds = PRE_BIG_ROUNDS;
            // never below 16 references per lane: sizes just above one round (n ~ 1.4e5 at c = 16) would otherwise get 262144 lanes
            // of 9, and chunks that short overload msm_combine* (measured at 2^20: 16 per lane cost 162 ms per proof against 110)
            const uint32_t l_r = (uint32_t)((pl.nf + (uint64_t)rounds * ROUND - 1) / ((uint64_t)rounds * ROUND));
            if (l_r >= 16) {
                pl.n_lanes = rounds * ROUND;
                pl.chunk_l = l_r;
            }
            pl.max_lanes = pl.n_lanes;
            if (long_chunks) {
                const uint32_t long_rounds = (uint32_t)c->tune.long_rounds;
                const uint32_t lr = long_rounds < 1 ? 1u : long_rounds > rounds ? rounds : long_rounds;
                // never more lanes than the plan the buffers were sized for (part_pt holds two partials per lane of max_lanes):
                // where the whole-round rounding above was refused (chunks below 16), lr rounds of lanes can exceed it
                if ((uint64_t)lr * ROUND <= pl.max_lanes) {
                    pl.n_lanes = lr * ROUND;
                    pl.chunk_l = (uint32_t)((pl.nf + pl.n_lanes - 1) / pl.n_lanes);
                }
            }
        }
    }
    pl.win_bytes = pl.wide_red ? (size_t)4 * sizeof(PH) : (size_t)2 * pl.gv.W * sizeof(PH);
    return ZK_OK;
}

// bytes of every work buffer of a job's set (MsmBufs) under a plan; 0 = not used
struct PreSizes {
    size_t counts, offsets, entries, buckets, part_pt, part_key, seg, seg2, seg3, win, stage;
};

template <class Cv>
void pre_sizes(const PrePlan& pl, PreSizes& z) {
    typedef typename Cv::FqU F;
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    memset(&z, 0, sizeof z);
    z.counts = (size_t)256 * PS_SLABS * 4;                       // slab counts of the 256 sort partitions -> cursors
    z.offsets = (size_t)(pl.g.B + 1) * 4;
    // the sorted references.  The job's digits (int16 / int32 per reference) live here first: the digit kernel writes them, the
    // partition scatter reads them into the staging area, and only then the placement kernel overwrites them with the references
    z.entries = (size_t)pl.nf * 4;
    z.buckets = (size_t)pl.g.B * PT;
    z.part_pt = (size_t)pl.max_lanes * 2 * PT;
    z.part_key = (size_t)(PRE_Q_OFF + pl.g.B + 2) * 4;          // partition-sort scratch | combine queues
    if (pl.wide_red) {
        const size_t n1 = pl.g.B >> WIDE_LOGG1, n2 = n1 >> WIDE_LOGK2;
        z.seg = n1 * 2 * PT;                                     // level 1: (run, acc) of the 4-bucket nodes
        z.seg2 = n2 * 2 * PT;                                    // level 2: 16-bucket nodes
        z.win = (size_t)2 * pl.gv.W * PT;                        // level 3: S_v | T_v of the virtual windows, internal form
        z.seg3 = (size_t)4 * 256 * PT;                           // level 4: (run, acc) of <= 256 segments for each of S, T
    } else {
        z.seg = (size_t)pl.gv.W * pl.gv.ns * 2 * PT;
        z.win = pl.win_bytes > (size_t)2 * pl.gv.W * PT ? pl.win_bytes : (size_t)2 * pl.gv.W * PT;   // SAT or internal form
    }
    z.stage = (size_t)pl.nf * (pl.wide ? 6 : 5);                 // references in partition order + their low bucket bits
}

// bytes the device would have to give for this job: what pre_ensure would newly allocate in `mb` (and in the ctx's shared staging area)
template <class Cv>
size_t pre_need(zk_ctx* c, const PrePlan& pl, const MsmBufs& mb) {
    PreSizes z;
    pre_sizes<Cv>(pl, z);
    size_t t = mb.counts.need_for(z.counts) + mb.offsets.need_for(z.offsets) + mb.entries.need_for(z.entries) + mb.buckets.need_for(z.buckets) +
               mb.part_pt.need_for(z.part_pt) + mb.part_key.need_for(z.part_key) + mb.seg.need_for(z.seg) + mb.win.need_for(z.win);
    if (z.seg2) t += mb.seg2.need_for(z.seg2) + mb.seg3.need_for(z.seg3);
    t += pl.shared_stage ? c->stage_shared.need_for(z.stage) : mb.stage.need_for(z.stage);
    return t;
}

template <class Cv>
int pre_ensure(zk_ctx* c, const PrePlan& pl, MsmBufs& mb) {
    PreSizes z;
    pre_sizes<Cv>(pl, z);
    int rc;
    if ((rc = mb.counts.ensure(z.counts))) return rc;
    if ((rc = mb.offsets.ensure(z.offsets))) return rc;
    if ((rc = mb.entries.ensure(z.entries))) return rc;
    if ((rc = mb.buckets.ensure(z.buckets))) return rc;
    if ((rc = mb.part_pt.ensure(z.part_pt))) return rc;
    if ((rc = mb.part_key.ensure(z.part_key))) return rc;
    if ((rc = mb.seg.ensure(z.seg))) return rc;
    if ((rc = mb.win.ensure(z.win))) return rc;
    if (z.seg2 && ((rc = mb.seg2.ensure(z.seg2)) || (rc = mb.seg3.ensure(z.seg3)))) return rc;
    // (growing a buffer frees the old one: hipFree waits for the device, so kernels of earlier jobs still reading it are safe)
    if ((rc = (pl.shared_stage ? c->stage_shared : mb.stage).ensure(z.stage))) return rc;
    return ZK_OK;
}

template <class Cv>
int pre_plan(zk_ctx* c, zk_srs* s, size_t n, MsmBufs& mb, PrePlan& pl, bool long_chunks = false) {
    int rc = pre_plan_geom<Cv>(c, s, n, pl, long_chunks);
    if (rc) return rc;
    return pre_ensure<Cv>(c, pl, mb);
}

template <class Cv>
bool partial_dev_supported(zk_ctx* c, zk_srs* s, uint32_t* vw, uint32_t* vb) {
    if (!s->pre_W || s->pre_rows == 0) return false;
    // the reduction geometry is a function of the table's window and the ctx's options only (pre_reduce_geom), not of a job's length
    PrePlan pl;
    pl.g = make_geom<typename Cv::FrP>(ZK_PRE_MIN_N, (int)s->pre_c, PRE_C_MAX);
    pre_reduce_geom(c, pl);
    if (!pre_partial_dev_ok(pl)) return false;
    if (vw) *vw = pl.gv.W;
    if (vb) *vb = pl.gv.B;
    return true;
}

// a host Jacobian point (X, Y, Z: what the blocking entry points return) in the device partial form, for jobs of a round that
// were computed at submission (vectors too short for the table path)
template <class Cv>
void jacobian_to_partial_host(const uint64_t* xyz, void* out) {
    typedef typename Cv::Fq Fq;
    typedef typename Cv::FqU F;
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    constexpr int L64 = Fq::N / 2;
    memset(out, 0, PT);
    Fq Z;
    memcpy(Z.v, xyz + 2 * L64, sizeof(uint64_t) * L64);
    if (Z.is_zero()) return;                       // infinity: all limbs zero
    const XYZZ<Fq> p = jac_to_xyzz<Fq>(xyz);
    const Fq* co[4] = {&p.x, &p.y, &p.zz, &p.zzz};
    for (int r = 0; r < 4; ++r) {
        const F v = F::canonical_lt2p(F::from_sat((const uint32_t*)co[r]->v));
        uint32_t* w = (uint32_t*)out + (size_t)r * Store<F>::WORDS;
        for (int i = 0; i < F::NL; ++i) w[i] = (uint32_t)v.v[i];
    }
}


// wide reduction: h = [sum_v S_v (as win: unused), sum_v S_v (tot) | K = sum_v (v+1) T_v, sum_v T_v]; buckets per virtual window = 2^log_bv
template <class Cv>
void pre_host_wide(const void* h, uint32_t log_bv, uint64_t* out_xyz) {
    typedef typename Cv::Fq Fq;
    typedef XYZZ<Fq> PH;
    constexpr int L64 = Fq::N / 2;
    const PH* w = (const PH*)h;
    PH d = PH::add(w[2], PH::neg(w[3]));
    for (uint32_t k = 0; k < log_bv; ++k) d = PH::dbl(d);
    PH total = PH::add(w[1], d);
    Fq X = Fq::one(), Y = Fq::one(), Z = Fq::zero();
    if (!total.is_inf()) {
        X = Fq::mul(total.x, total.zz);
        Y = Fq::mul(total.y, total.zzz);
        Z = total.zz;
    }
    memcpy(out_xyz, X.v, sizeof(uint64_t) * L64);
    memcpy(out_xyz + L64, Y.v, sizeof(uint64_t) * L64);
    memcpy(out_xyz + 2 * L64, Z.v, sizeof(uint64_t) * L64);
}

// host: S = sum_v S_v + B_v * sum_v v * T_v   (bucket j of virtual window v has weight v*B_v + local index).
// The sum over v is cut into HOST_CHUNKS ranges that can run on different pool threads:
//   range [lo, hi): s = sum S_v, t = sum T_v, w = sum (v - lo) * T_v     (running sums, 3 additions per window)
//   S = sum_c s_c + B_v * sum_c (w_c + lo_c * t_c),  lo_c = c * (VW / HOST_CHUNKS)
constexpr uint32_t HOST_CHUNKS = 4;
template <class Fq>
struct HostPartial {
    XYZZ<Fq> s, t, w;
};
template <class Cv>
void pre_host_partial(const void* h_win, uint32_t VW, uint32_t lo, uint32_t hi, HostPartial<typename Cv::Fq>& out) {
    typedef XYZZ<typename Cv::Fq> PH;
    const PH* win = (const PH*)h_win;
    PH s = PH::infinity(), run = PH::infinity(), w = PH::infinity();
    for (int v = (int)hi - 1; v >= (int)lo; --v) {
        s = PH::add(s, win[v]);
        if (v > (int)lo) {
            run = PH::add(run, win[VW + v]);
            w = PH::add(w, run);
        }
    }
    out.s = s;
    out.t = PH::add(run, win[VW + lo]);
    out.w = w;
}
template <class Cv>
void pre_host_final(const HostPartial<typename Cv::Fq>* part, uint32_t VW, uint32_t VB, uint64_t* out_xyz) {
    typedef typename Cv::Fq Fq;
    typedef XYZZ<Fq> PH;
    constexpr int L64 = Fq::N / 2;
    // sum_c c * t_c by running sums, then times the chunk length (a power of two), plus the local weights
    PH total = PH::infinity(), run = PH::infinity(), ct = PH::infinity(), wsum = PH::infinity();
    for (int c = (int)HOST_CHUNKS - 1; c >= 0; --c) {
        total = PH::add(total, part[c].s);
        wsum = PH::add(wsum, part[c].w);
        if (c >= 1) {
            run = PH::add(run, part[c].t);
            ct = PH::add(ct, run);
        }
    }
    for (uint32_t k = 0; (1u << k) < VW / HOST_CHUNKS; ++k) ct = PH::dbl(ct);
    wsum = PH::add(wsum, ct);
    for (uint32_t k = 0; (1u << k) < VB; ++k) wsum = PH::dbl(wsum);
    total = PH::add(total, wsum);
    Fq X = Fq::one(), Y = Fq::one(), Z = Fq::zero();
    if (!total.is_inf()) {
        X = Fq::mul(total.x, total.zz);
        Y = Fq::mul(total.y, total.zzz);
        Z = total.zz;
    }
    memcpy(out_xyz, X.v, sizeof(uint64_t) * L64);
    memcpy(out_xyz + L64, Y.v, sizeof(uint64_t) * L64);
    memcpy(out_xyz + 2 * L64, Z.v, sizeof(uint64_t) * L64);
}
template <class Cv>
void pre_host_combine(const void* h_win, uint32_t VW, uint32_t VB, uint64_t* out_xyz) {
    HostPartial<typename Cv::Fq> part[HOST_CHUNKS];
    for (uint32_t c = 0; c < HOST_CHUNKS; ++c) pre_host_partial<Cv>(h_win, VW, c * (VW / HOST_CHUNKS), (c + 1) * (VW / HOST_CHUNKS), part[c]);
    pre_host_final<Cv>(part, VW, VB, out_xyz);
}

int ensure_pinned(zk_ctx* c, size_t bytes) {
    if (c->pinned_cap >= bytes) return ZK_OK;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr;
    c->pinned_cap = 0;
    if (hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault) != hipSuccess) return ZK_ERR_OOM;
    c->pinned_cap = bytes;
    return ZK_OK;
}

// one MSM, everything on the main stream
template <class Cv>
int msm_run_pre(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    MsmBufs& mb = c->mb[0];
    PrePlan pl;
    int rc = pre_plan<Cv>(c, s, n, mb, pl);
    if (rc == ZK_ERR_OOM) {                        // sets of earlier, larger rounds still hold memory: give it back and try once more
        zk_release_free_work(c, 0);
        rc = pre_plan<Cv>(c, s, n, mb, pl);
    }
    if (rc) return rc;
    if ((rc = ensure_pinned(c, pl.win_bytes * MAX_JOBS))) return rc;
    MsmBufs* one = &mb;
    if ((rc = ZK_SYM(pre_queue_digits)(c, pl, mb, d_scalars, n, c->stream, false))) return rc;
    if ((rc = ZK_SYM(pre_queue_sort_rest)(c, &pl, &one, &n, 1, c->stream))) return rc;
    if ((rc = ZK_SYM(pre_queue_accumulate)(c, &pl, &one, &n, &base_offset, 1, s, c->stream))) return rc;
    if ((rc = ZK_SYM(pre_queue_reduce)(c, &pl, &one, 1, c->pinned, c->stream, nullptr))) return rc;
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    if (pl.wide_red) pre_host_wide<Cv>(c->pinned, ilog2_floor(pl.gv.B), out_xyz);
    else pre_host_combine<Cv>(c->pinned, pl.gv.W, pl.gv.B, out_xyz);
    return ZK_OK;
}

template <class Fq>
int jac_to_affine(const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf);

// A batch of commitments over the same SRS (the polynomials of one prover round): Montgomery
// coefficients in, Jacobian results out.  Every job has its own buffer set and every step is ONE launch per kernel
// for all the jobs of the round (job = blockIdx.y, or a block range of the merged accumulation): the sort's placement
// passes, the accumulation, the combine / segmented-reduction steps.  Everything stays on the ctx stream: overlapping
// neighbouring jobs on a second stream was measured to cost more than it hides (profiles/r01/r01_notes.md, r02_notes.md).
//
// The batch comes in pieces so that a round may be OPENED by several calls and closed by one
// (zk_kzg_round_begin_dev / zk_kzg_round_end):
//   begin   per job: the digit kernel -- the only reader of the caller's vector -- into the buffer set c->mb[slot] (stage 1).
//           With a `before_job` hook (the host-pointer batch uploads job k there, so that the upload of job k+1 runs under the
//           accumulation of job k) the job's whole sort and its own accumulation launch follow at once (stage 2).
//   reduce  the placement passes and ONE accumulation launch for every stage-1 job, then the reductions of all jobs, an event.
//   end     waits for that event and finishes on the host.
// Option "msm_merge" = 0 (A/B hook): every job is sorted and accumulated by its own launches at begin, as before round 4.
static bool msm_merge_enabled(const zk_ctx* c) { return c->tune.msm_merge != 0; }

// bytes of device memory the table path may still take: free memory minus the reserve of the ctx's options (transforms, the
// caller's own allocations in flight); SIZE_MAX when the runtime cannot say (hipMalloc then decides)
static size_t pre_mem_available(const zk_ctx* c) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return SIZE_MAX;
    const size_t reserve = (size_t)(c->tune.mem_reserve_mb < 0 ? 0 : c->tune.mem_reserve_mb) << 20;
    return fr > reserve ? fr - reserve : 0;
}

// Memory budget: before job k takes the buffer set of its slot,
//   1. a FREE set (no job living in it) that already holds more of what the job needs is adopted -- its work buffers change places
//      with the slot's -- so that sets released by an early close are reused instead of allocated again;
//   2. if the device then has no room for what is still missing (hipMemGetInfo; or the test hook's limit on the queued sets) and
//      jobs are queued whose close would free their sets, the call stops with ZK_ERR_OOM and *n_begun jobs begun: the caller
//      closes the queued jobs (msm_batch_pre_end), parks their points and calls again.
// With nothing queued the job is always attempted: a hard ZK_ERR_OOM then comes from hipMalloc itself.
template <class Cv>
int msm_batch_pre_begin(zk_ctx* c, zk_srs* s, uint32_t slot0, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens,
                        const uint8_t* kinds /* per job: 0 Montgomery coefficients, 1 canonical scalars; may be null */,
                        const std::function<int(uint32_t)>* before_job /* optional: runs before job k is queued */, uint32_t* n_begun) {
    if (n_begun) *n_begun = 0;
    if (n_polys == 0) return ZK_OK;
    if (slot0 + n_polys > (uint32_t)MAX_JOBS) return ZK_ERR_UNSUPPORTED;
    int rc;
    hipStream_t st = c->stream;
    const bool defer = msm_merge_enabled(c) && !before_job;
    for (uint32_t k = 0; k < n_polys; ++k) {
        MsmBufs& mb = c->mb[slot0 + k];
        PrePlan pl;
        if ((rc = pre_plan_geom<Cv>(c, s, lens[k], pl))) return rc;
        size_t need = pre_need<Cv>(c, pl, mb);
        if (need) {
            int best = -1;
            for (int j = 0; j < MAX_JOBS; ++j) {
                MsmBufs& o = c->mb[j];
                if (&o == &mb || o.stage_of_job != 0 || o.entries.cap == 0) continue;
                const size_t nj = pre_need<Cv>(c, pl, o);
                if (nj < need) {
                    need = nj;
                    best = j;
                }
            }
            if (best >= 0) mb.swap_work(c->mb[best]);
        }
        if (need) {
            size_t queued_bytes = 0;
            uint32_t queued = 0;
            for (int j = 0; j < MAX_JOBS; ++j)
                if (c->mb[j].stage_of_job != 0) {
                    queued_bytes += c->mb[j].work_bytes();
                    ++queued;
                }
            if (queued) {
                const size_t limit = (size_t)(c->tune.round_mem_limit_mb > 0 ? c->tune.round_mem_limit_mb : 0) << 20;
                if (limit && queued_bytes + need > limit) return ZK_ERR_OOM;
                if (need > pre_mem_available(c)) return ZK_ERR_OOM;
            }
        }
        if ((rc = pre_ensure<Cv>(c, pl, mb))) return rc;
        const bool mont = !kinds || kinds[k] == 0;   // a commit: Montgomery coefficients, into_repr fused into the digit kernel
        if (before_job && (rc = (*before_job)(k))) return rc;
        if ((rc = ZK_SYM(pre_queue_digits)(c, pl, mb, d_coeffs[k], lens[k], st, mont))) return rc;
        mb.stage_of_job = 1;
        if (n_begun) *n_begun = k + 1;
        if (defer) continue;
        MsmBufs* one = &mb;
        if ((rc = ZK_SYM(pre_queue_sort_rest)(c, &pl, &one, &lens[k], 1, st))) return rc;
        if ((rc = ZK_SYM(pre_queue_accumulate)(c, &pl, &one, &lens[k], nullptr, 1, s, st))) return rc;
        mb.stage_of_job = 2;
    }
    return ZK_OK;
}

// slots[k]: the buffer set job k was queued into; lens[k]: its length (the plan is a function of the SRS, the length and whether
// the job is followed by another one in the merged accumulation launch).
// The end comes in two steps so that a caller may put other work of the stream (transforms that do not depend on this round's
// results) BEHIND the reductions before it waits: `reduce` queues everything up to the reduction kernels and an event, `end` waits
// for that event only -- the work queued in between runs while the host combines the window sums and normalises.
template <class Cv>
int msm_batch_pre_reduce(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_winsums = nullptr) {
    if (n_jobs == 0) return ZK_OK;
    if (n_jobs > (uint32_t)MAX_JOBS) return ZK_ERR_UNSUPPORTED;
    int rc;
    PrePlan pl[MAX_JOBS];
    MsmBufs* mbs[MAX_JOBS];
    uint32_t last_deferred = n_jobs;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        if (slots[k] >= (uint32_t)MAX_JOBS) return ZK_ERR_BAD_ARG;
        mbs[k] = &c->mb[slots[k]];
        if (mbs[k]->stage_of_job == 0) return ZK_ERR_BAD_ARG;       // never submitted
        if (mbs[k]->stage_of_job == 1) last_deferred = k;
    }
    PrePlan dpl[MAX_JOBS];
    MsmBufs* dmb[MAX_JOBS];
    size_t dlen[MAX_JOBS];
    uint32_t nd = 0;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        const bool deferred = mbs[k]->stage_of_job == 1;
        if ((rc = pre_plan_geom<Cv>(c, s, lens[k], pl[k], deferred && k != last_deferred))) return rc;    // the buffers were taken at begin
        if (pl[k].g1.nb != pl[0].g1.nb || pl[k].gv.ns != pl[0].gv.ns) return ZK_ERR_UNSUPPORTED;
        // a device form the reduction cannot deliver is refused HERE, before the sort and the accumulation of the deferred jobs
        // are queued: the round is then exactly as it was and the host form may still close it
        if (d_winsums && !pre_partial_dev_ok(pl[k])) return ZK_ERR_UNSUPPORTED;
        if (deferred) {
            dpl[nd] = pl[k];
            dmb[nd] = mbs[k];
            dlen[nd] = lens[k];
            ++nd;
        }
    }
    if ((rc = ensure_pinned(c, pl[0].win_bytes * MAX_JOBS))) return rc;
    hipStream_t st = c->stream;
    if (nd) {
        if ((rc = ZK_SYM(pre_queue_sort_rest)(c, dpl, dmb, dlen, nd, st))) return rc;
        if ((rc = ZK_SYM(pre_queue_accumulate)(c, dpl, dmb, dlen, nullptr, nd, s, st))) return rc;
        for (uint32_t k = 0; k < nd; ++k) dmb[k]->stage_of_job = 2;
    }
    if ((rc = ZK_SYM(pre_queue_reduce)(c, pl, mbs, n_jobs, c->pinned, st, d_winsums))) return rc;
    for (uint32_t k = 0; k < n_jobs; ++k) mbs[k]->stage_of_job = 0;
    if (!c->round_ev) ZK_HIP_TRY(hipEventCreateWithFlags(&c->round_ev, hipEventDisableTiming));
    ZK_HIP_TRY(hipEventRecord(c->round_ev, st));
    c->round_reduced = n_jobs;
    return ZK_OK;
}

// the host tail of a round on the shared-bucket path: n_jobs x (VW pairs S_v | T_v in pinned memory, arkworks layout) -> Jacobian (and affine)
template <class Cv>
int pre_host_finish_jobs(zk_ctx* c, const char* h_win, size_t wb, uint32_t n_jobs, uint32_t VW, uint32_t VB, uint64_t* out_xyz /* n_jobs x 3L */,
                         uint64_t* out_xy /* optional */, uint8_t* out_inf /* optional */) {
    typedef typename Cv::Fq Fq;
    constexpr int L64 = Fq::N / 2;
    int rcs[MAX_JOBS] = {0};
    // ~200 point additions + one field inversion per job (measured: 200 us on one host thread, the GPU idle meanwhile): every
    // job's virtual windows are cut into HOST_CHUNKS ranges that go to the pool as separate items, and whichever thread
    // finishes a job's last range also does that job's final sum and affine normalisation -- one wake-up of the pool per round
    HostPartial<Fq> part[MAX_JOBS * HOST_CHUNKS];
    std::atomic<uint32_t> left[MAX_JOBS];
    for (uint32_t k = 0; k < n_jobs; ++k) left[k].store(HOST_CHUNKS);
    c->pool->run(n_jobs * HOST_CHUNKS, [&](uint32_t i) {
        const uint32_t k = i / HOST_CHUNKS, ch = i % HOST_CHUNKS;
        pre_host_partial<Cv>(h_win + (size_t)k * wb, VW, ch * (VW / HOST_CHUNKS), (ch + 1) * (VW / HOST_CHUNKS), part[i]);
        if (left[k].fetch_sub(1, std::memory_order_acq_rel) != 1) return;
        uint64_t* xyz = out_xyz + (size_t)k * 3 * L64;
        pre_host_final<Cv>(part + k * HOST_CHUNKS, VW, VB, xyz);
        if (out_xy) rcs[k] = jac_to_affine<Fq>(xyz, out_xy + (size_t)k * 2 * L64, out_inf ? out_inf + k : nullptr);
    });
    for (uint32_t k = 0; k < n_jobs; ++k)
        if (rcs[k]) return rcs[k];
    return ZK_OK;
}

// the ranks' virtual-window sums (zk_kzg_round_end_winsums_dev on every rank, all-gathered rank-major) -> n_jobs affine commitments:
// one kernel adds them element-wise into the pinned buffer, one wait, then the single-GPU path's own host tail
template <class Cv>
int sum_winsums_dev(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    typedef typename Cv::Fq Fq;
    typedef XYZZ<Fq> PH;
    constexpr int L64 = Fq::N / 2;
    if (n_jobs == 0) return ZK_OK;
    if (ranks == 0 || ranks > 4096 || n_jobs > (uint32_t)MAX_JOBS) return ZK_ERR_BAD_ARG;
    uint32_t VW = 0, VB = 0;
    if (!partial_dev_supported<Cv>(c, s, &VW, &VB)) return ZK_ERR_UNSUPPORTED;
    const size_t wb = (size_t)2 * VW * sizeof(PH);
    int rc = ensure_pinned(c, wb * MAX_JOBS);
    if (rc) return rc;
    const uint32_t n_pts = n_jobs * 2 * VW;
    if ((rc = ZK_SYM(queue_sum_winsums)(c, d_all, (uint32_t)ranks, n_pts, c->pinned, c->stream))) return rc;
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    uint64_t xyz[MAX_JOBS * 3 * L64];
    return pre_host_finish_jobs<Cv>(c, (const char*)c->pinned, wb, n_jobs, VW, VB, xyz, out_xy, out_inf);
}

template <class Cv>
int msm_batch_pre_end(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, uint64_t* out_xyz /* n_jobs x 3L */,
                      uint64_t* out_xy /* optional: n_jobs x 2L affine */, uint8_t* out_inf /* optional flags */) {
    typedef typename Cv::Fq Fq;
    constexpr int L64 = Fq::N / 2;
    if (n_jobs == 0) return ZK_OK;
    if (n_jobs > (uint32_t)MAX_JOBS) return ZK_ERR_UNSUPPORTED;
    int rc;
    if (c->round_reduced != n_jobs && (rc = msm_batch_pre_reduce<Cv>(c, s, n_jobs, slots, lens))) return rc;
    c->round_reduced = 0;
    PrePlan pl[MAX_JOBS];
    for (uint32_t k = 0; k < n_jobs; ++k)
        if ((rc = pre_plan_geom<Cv>(c, s, lens[k], pl[k]))) return rc;
    const size_t wb = pl[0].win_bytes;
    static const bool host_timing = getenv("ZK_HOST_TIMING") != nullptr;      // diagnostic: where the host tail of a round goes
    const auto t0 = std::chrono::steady_clock::now();
    ZK_HIP_TRY(hipEventSynchronize(c->round_ev));
    const auto t1 = std::chrono::steady_clock::now();
    struct TailTimer {
        bool on;
        uint32_t n;
        std::chrono::steady_clock::time_point t0, t1;
        ~TailTimer() {
            if (!on) return;
            const auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[zk host tail] jobs %u: wait for the stream %.1f us, combine + affine %.1f us\n", n,
                    std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
        }
    } tail_timer{host_timing, n_jobs, t0, t1};
    const char* h_win = (const char*)c->pinned;
    if (pl[0].wide_red) {
        int rcs[MAX_JOBS] = {0};
        c->pool->run(n_jobs, [&](uint32_t k) {
            uint64_t* xyz = out_xyz + (size_t)k * 3 * L64;
            pre_host_wide<Cv>(h_win + (size_t)k * wb, ilog2_floor(pl[k].gv.B), xyz);
            if (out_xy) rcs[k] = jac_to_affine<Fq>(xyz, out_xy + (size_t)k * 2 * L64, out_inf ? out_inf + k : nullptr);
        });
        for (uint32_t k = 0; k < n_jobs; ++k)
            if (rcs[k]) return rcs[k];
        return ZK_OK;
    }
    return pre_host_finish_jobs<Cv>(c, h_win, wb, n_jobs, pl[0].gv.W, pl[0].gv.B, out_xyz, out_xy, out_inf);
}

// The blocking form: begin every job, end them together.  Under the memory budget (msm_batch_pre_begin) the call may come in
// pieces -- the jobs begun so far are ended, their sets reused by the rest -- with the same points in the same order.
template <class Cv>
int msm_batch_pre(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens, uint64_t* out_xyz /* n_polys x 3L */,
                  const uint8_t* kinds, uint64_t* out_xy, uint8_t* out_inf, const std::function<int(uint32_t)>* before_job) {
    constexpr int L64 = Cv::Fq::N / 2;
    if (n_polys == 0) return ZK_OK;
    if (n_polys > (uint32_t)MAX_JOBS) return ZK_ERR_UNSUPPORTED;
    for (int k = 0; k < MAX_JOBS; ++k) c->mb[k].stage_of_job = 0;     // no round is open (the callers refuse otherwise): every set is free
    uint32_t done = 0;
    bool released = false;
    while (done < n_polys) {
        uint32_t begun = 0;
        const std::function<int(uint32_t)> shifted = [&](uint32_t k) { return (*before_job)(done + k); };
        int rc = msm_batch_pre_begin<Cv>(c, s, 0, n_polys - done, d_coeffs + done, lens + done, kinds ? kinds + done : nullptr,
                                         before_job ? &shifted : nullptr, &begun);
        if (rc == ZK_ERR_OOM && begun == 0 && !released) {
            zk_release_free_work(c, -1);           // whatever earlier, larger calls left in the sets
            released = true;
            continue;
        }
        if (rc && !(rc == ZK_ERR_OOM && begun > 0)) {
            if (begun) (void)hipStreamSynchronize(c->stream);     // the queued kernels still read the caller's inputs
            for (int k = 0; k < MAX_JOBS; ++k) c->mb[k].stage_of_job = 0;
            return rc;
        }
        const uint32_t m = rc ? begun : n_polys - done;
        if (rc) ++c->round_flushes;
        uint32_t slots[MAX_JOBS];
        for (uint32_t k = 0; k < m; ++k) slots[k] = k;
        rc = msm_batch_pre_end<Cv>(c, s, m, slots, lens + done, out_xyz + (size_t)done * 3 * L64, out_xy ? out_xy + (size_t)done * 2 * L64 : nullptr,
                                   out_inf ? out_inf + done : nullptr);
        if (rc) {
            (void)hipStreamSynchronize(c->stream);
            for (int k = 0; k < MAX_JOBS; ++k) c->mb[k].stage_of_job = 0;
            return rc;
        }
        done += m;
    }
    return ZK_OK;
}

template <class Fq>
int jac_to_affine(const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf) {
    constexpr int L64 = Fq::N / 2;
    Fq X, Y, Z;
    memcpy(X.v, xyz, sizeof(uint64_t) * L64);
    memcpy(Y.v, xyz + L64, sizeof(uint64_t) * L64);
    memcpy(Z.v, xyz + 2 * L64, sizeof(uint64_t) * L64);
    if (Z.is_zero()) {
        // GroupAffine::zero() = (0, 1, infinity = true)
        Fq zero = Fq::zero(), one = Fq::one();
        memcpy(out_xy, zero.v, sizeof(uint64_t) * L64);
        memcpy(out_xy + L64, one.v, sizeof(uint64_t) * L64);
        if (out_inf) *out_inf = 1;
        return ZK_OK;
    }
    Fq zi = Fq::inverse(Z);
    Fq zi2 = Fq::sqr(zi);
    Fq x = Fq::mul(X, zi2);
    Fq y = Fq::mul(Y, Fq::mul(zi2, zi));
    memcpy(out_xy, x.v, sizeof(uint64_t) * L64);
    memcpy(out_xy + L64, y.v, sizeof(uint64_t) * L64);
    if (out_inf) *out_inf = 0;
    return ZK_OK;
}

// Jacobian (X, Y, Z) -> XYZZ (X, Y, Z^2, Z^3)
template <class Fq>
XYZZ<Fq> jac_to_xyzz(const uint64_t* xyz) {
    constexpr int L64 = Fq::N / 2;
    XYZZ<Fq> p;
    Fq Z;
    memcpy(p.x.v, xyz, sizeof(uint64_t) * L64);
    memcpy(p.y.v, xyz + L64, sizeof(uint64_t) * L64);
    memcpy(Z.v, xyz + 2 * L64, sizeof(uint64_t) * L64);
    p.zz = Fq::sqr(Z);
    p.zzz = Fq::mul(p.zz, Z);
    return p;
}

template <class Fq>
int sum_partials(const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    constexpr int L64 = Fq::N / 2;
    XYZZ<Fq> acc = XYZZ<Fq>::infinity();
    for (size_t i = 0; i < count; ++i) acc = XYZZ<Fq>::add(acc, jac_to_xyzz<Fq>(partials + i * 3 * L64));
    Affine<Fq> a;
    bool fin = acc.to_affine(a);
    if (!fin) {
        Fq one = Fq::one();
        memset(out_xy, 0, sizeof(uint64_t) * L64);
        memcpy(out_xy + L64, one.v, sizeof(uint64_t) * L64);
        if (out_inf) *out_inf = 1;
        return ZK_OK;
    }
    memcpy(out_xy, a.x.v, sizeof(uint64_t) * L64);
    memcpy(out_xy + L64, a.y.v, sizeof(uint64_t) * L64);
    if (out_inf) *out_inf = 0;
    return ZK_OK;
}

}  // namespace

int ZK_SYM(msm_run_dev)(zk_ctx* c, const void* d_bases_xy, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    return msm_run<CurveSel>(c, d_bases_xy, d_scalars, n, out_xyz);
}
int ZK_SYM(msm_run_pre_dev)(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    return msm_run_pre<CurveSel>(c, s, base_offset, d_scalars, n, out_xyz);
}
int ZK_SYM(msm_batch_pre_dev)(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens, uint64_t* out_xyz,
                              const uint8_t* kinds, uint64_t* out_xy, uint8_t* out_inf, const std::function<int(uint32_t)>* before_job) {
    return msm_batch_pre<CurveSel>(c, s, n_polys, d_coeffs, lens, out_xyz, kinds, out_xy, out_inf, before_job);
}
int ZK_SYM(msm_batch_pre_begin_dev)(zk_ctx* c, zk_srs* s, uint32_t slot0, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens,
                                    const uint8_t* kinds, const std::function<int(uint32_t)>* before_job) {
    return msm_batch_pre_begin<CurveSel>(c, s, slot0, n_polys, d_coeffs, lens, kinds, before_job, nullptr);
}
int ZK_SYM(msm_batch_pre_reduce_dev)(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_winsums) {
    return msm_batch_pre_reduce<CurveSel>(c, s, n_jobs, slots, lens, d_winsums);
}
bool ZK_SYM(msm_partial_dev_supported)(zk_ctx* c, zk_srs* s, uint32_t* vw, uint32_t* vb) { return partial_dev_supported<CurveSel>(c, s, vw, vb); }
int ZK_SYM(g1_sum_winsums_dev)(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    return sum_winsums_dev<CurveSel>(c, s, d_all, ranks, n_jobs, out_xy, out_inf);
}
void ZK_SYM(g1_jacobian_to_partial_host)(const uint64_t* xyz, void* out) { jacobian_to_partial_host<CurveSel>(xyz, out); }
int ZK_SYM(msm_batch_pre_end_dev)(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, uint64_t* out_xyz,
                                  uint64_t* out_xy, uint8_t* out_inf) {
    return msm_batch_pre_end<CurveSel>(c, s, n_jobs, slots, lens, out_xyz, out_xy, out_inf);
}
int ZK_SYM(g1_jacobian_to_affine_host)(const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf) {
    return jac_to_affine<CurveSel::Fq>(xyz, out_xy, out_inf);
}
int ZK_SYM(g1_sum_partials_host)(const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    return sum_partials<CurveSel::Fq>(partials, count, out_xy, out_inf);
}
