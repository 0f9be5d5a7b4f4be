// Shared by the MSM units (msm_sort.hip, msm_accumulate.hip, msm_reduce.hip, msm_plan.hip): the device storage of field elements and
// points, the geometry and job structs the kernels take, the plan of a window-table job, and the per-curve entry points one unit offers
// the others.  Every unit is built once per curve (-DZK_CURVE_SEL=<0|1>, ark_plonk_amd/build.py): the names below carry the curve
// suffix; msm_dispatch.hip dispatches the library-internal interface of ctx.h on the curve id.
//
// Pipeline (all on the ctx stream; no host round trip until the window sums are read back).  On the window-table path every step is
// ONE launch per kernel for all the MSMs of a prover round (job = blockIdx.y, or a block range of the accumulation):
//   msm_sort.hip        digits (signed c-bit, window-major; into_repr fused for commits) + the sort of the (point, sign) references by
//                       bucket, no global atomics: two-pass partition sort over the one shared bucket set (table path), LDS counting
//                       sort per window (per-window path)
//   msm_accumulate.hip  every lane sums a fixed-length chunk of the sorted list with XYZZ mixed additions (no inversion); runs that
//                       cross a chunk edge are emitted as partials; the jobs of a round follow each other inside one launch.
//                       Also the window-multiples table of an SRS, the base conversion and the fixed-base utility
//   msm_reduce.hip      joins the chunk-edge partials of each bucket, segmented running-sum reduction, window sums (arkworks layout
//                       in pinned host memory, or the internal form on the device for the multi-GPU exchange)
//   msm_plan.hip        host: window geometry, the plan and memory budget of a job, the per-window and table entry points, the
//                       deferred rounds' begin / reduce / end, the host combine and affine normalisation
#pragma once
#include "ctx.h"

#include <cstdio>

#if ZK_CURVE_SEL == 0
typedef CurveBls CurveSel;
#define ZK_SYM(name) name##_c0
#else
typedef CurveBn CurveSel;
#define ZK_SYM(name) name##_c1
#endif

namespace zkmsm {

// ---- device storage of a field element (Fs, fields.cuh): NL limbs padded to a multiple of 4 words (16-byte vector access)
template <class F>
struct Store {
    static constexpr int U4 = (F::NL + 3) / 4;      // uint4 per field element
    static constexpr int WORDS = 4 * U4;
};
template <class F>
ZK_D F ld_fu(const uint4* q) {
    F r;
#pragma unroll
    for (int i = 0; i < Store<F>::U4; ++i) {
        uint4 a = q[i];
        if (4 * i + 0 < F::NL) r.v[4 * i + 0] = a.x;
        if (4 * i + 1 < F::NL) r.v[4 * i + 1] = a.y;
        if (4 * i + 2 < F::NL) r.v[4 * i + 2] = a.z;
        if (4 * i + 3 < F::NL) r.v[4 * i + 3] = a.w;
    }
    return r;
}
template <class F>
ZK_D void st_fu(uint4* q, const F& r) {
#pragma unroll
    for (int i = 0; i < Store<F>::U4; ++i) {
        uint4 a;
        a.x = 4 * i + 0 < F::NL ? r.v[4 * i + 0] : 0u;
        a.y = 4 * i + 1 < F::NL ? r.v[4 * i + 1] : 0u;
        a.z = 4 * i + 2 < F::NL ? r.v[4 * i + 2] : 0u;
        a.w = 4 * i + 3 < F::NL ? r.v[4 * i + 3] : 0u;
        q[i] = a;
    }
}
template <class F>
ZK_D AffineU<F> ld_affine(const void* bases, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(bases) + idx * (2 * Store<F>::U4);
    AffineU<F> p;
    p.x = ld_fu<F>(q);
    p.y = ld_fu<F>(q + Store<F>::U4);
    return p;
}
template <class F>
ZK_D XYZZu<F> ld_xyzz(const void* arr, uint64_t idx) {
    const uint4* q = reinterpret_cast<const uint4*>(arr) + idx * (4 * Store<F>::U4);
    XYZZu<F> p;
    p.x = ld_fu<F>(q);
    p.y = ld_fu<F>(q + Store<F>::U4);
    p.zz = ld_fu<F>(q + 2 * Store<F>::U4);
    p.zzz = ld_fu<F>(q + 3 * Store<F>::U4);
    return p;
}
template <class F>
ZK_D void st_xyzz(void* arr, uint64_t idx, const XYZZu<F>& p) {
    uint4* q = reinterpret_cast<uint4*>(arr) + idx * (4 * Store<F>::U4);
    st_fu<F>(q, p.x);
    st_fu<F>(q + Store<F>::U4, p.y);
    st_fu<F>(q + 2 * Store<F>::U4, p.zz);
    st_fu<F>(q + 3 * Store<F>::U4, p.zzz);
}

// one coordinate (role 0..3 = X, Y, ZZ, ZZZ) of a stored XYZZ point: the quad-cooperative kernels (ecq.cuh)
template <class F>
ZK_D F ld_coord(const void* arr, uint64_t idx, uint32_t role) {
    return ld_fu<F>(reinterpret_cast<const uint4*>(arr) + idx * (4 * Store<F>::U4) + role * Store<F>::U4);
}
template <class F>
ZK_D void st_coord(void* arr, uint64_t idx, uint32_t role, const F& c) {
    st_fu<F>(reinterpret_cast<uint4*>(arr) + idx * (4 * Store<F>::U4) + role * Store<F>::U4, c);
}

struct MsmGeom {
    uint32_t c;        // window bits
    uint32_t W;        // windows
    uint32_t B;        // buckets per window = 2^(c-1)
    uint32_t nb;       // W * B
    uint32_t logG;     // level-1 segment = 2^logG buckets
    uint32_t ns;       // segments per window
    uint32_t logq;     // level-2: 2^logq segments per lane
    // neg: a scalar k > (r - 1) / 2 is replaced by r - k and the signs of its digits are flipped (k P = (r - k)(-P)).  The
    // replaced scalar has one bit less, which saves a whole window where the window size divides the remaining bits well:
    // 255-bit scalars in 17-bit windows need 16 windows (the last one holds nothing but a carry), 254-bit ones exactly 15.
    // make_geom switches it on only when it removes a window; half = (r - 1) / 2 and mod = r, little-endian 32-bit words.
    uint32_t neg = 0;
    uint32_t half[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t mod[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Window-sharded table (one rank of a multi-GPU MSM owns the windows w0, w0 + wstep, ...): the digit kernels walk all Wt windows of
    // the scalar -- the carries run through every window -- and keep the digits of the owned ones as rows 0 .. W-1 of the digit array;
    // everything behind them sees an MSM of W windows.  Whole table: Wt = W, w0 = 0, wstep = 1.
    uint32_t Wt = 0, w0 = 0, wstep = 1;
    ZK_HD bool owns(uint32_t w, uint32_t& row) const {
        if (w < w0) return false;
        const uint32_t d = w - w0;
        row = d / wstep;
        return d == row * wstep && row < W;
    }
};

// The kernels behind a prover round take up to 16 jobs (blockIdx.y, or a block-range table): the MSMs of one round are
// sorted, accumulated and reduced by ONE launch of each kernel, so that launch gaps, partly filled last rounds of wavefronts and
// the latency of the dependent-addition chains are paid once per round instead of once per MSM.
constexpr int MAX_JOBS = 16;
// one job of the batched partition-sort kernels (psort_scan / psort(w)_scatter / psort(w)_final)
struct SJob {
    const void* dig;        // digits [W][n]: int16 (c = 16) or int32 (c > 16)
    uint64_t n;             // scalars
    uint32_t sp, pad;       // scalars per slab
    uint32_t* hist;         // [P][PS_SLABS] slab counts -> cursors
    uint32_t* part_start;   // P + 1 partition starts
    uint32_t* part_total;   // P
    uint32_t* counter;
    uint32_t* stage_ref;    // references in partition order ...
    void* stage_lo;         // ... and their low bucket bits (uint8 / uint16)
    uint32_t* entries;      // references in bucket order
    uint32_t* offsets;      // bucket starts, offsets[nb] = references in the list
};
struct SJobs {
    SJob j[MAX_JOBS];
};
// one job of the merged accumulation launch: workgroups [blk0, next job's blk0) sum this job's list
struct AJob {
    const uint32_t* entries;
    const uint32_t* offsets;
    void* buckets;
    void* part_pt;
    uint64_t tab_off;
    uint32_t L0, n_lanes, blk0, pad;
};
struct AJobs {
    AJob j[MAX_JOBS];
    uint32_t n;
};

// The reduction kernels take up to 16 jobs (blockIdx.y), like the sort and the accumulation (MAX_JOBS above).
constexpr int MAX_RJOBS = 32;     // the last level of the wide reduction reduces two arrays (S_v, T_v) per job
struct RJobs {
    const void* part_pt[MAX_RJOBS];
    const uint32_t* offsets[MAX_RJOBS];     // nullptr: every bucket is present (levels above the first)
    void* buckets[MAX_RJOBS];
    uint32_t* q[MAX_RJOBS];
    void* seg_run[MAX_RJOBS];
    void* seg_acc[MAX_RJOBS];
    uint32_t* win_s[MAX_RJOBS];
    uint32_t* win_t[MAX_RJOBS];
    uint32_t L[MAX_RJOBS];          // references per lane the accumulate launch was sized for ...
    uint32_t lanes[MAX_RJOBS];      // ... its lanes, and the bucket count (offsets[nbk] = references in the list): see chunk_len
    uint32_t nbk[MAX_RJOBS];
};


constexpr uint32_t PS_LOB = 7;        // low bucket bits ordered inside a partition of the two-pass sort (msm_sort.hip)
constexpr uint32_t PS_T = 1024;
constexpr uint32_t PS_SLABS = 1024;   // workgroups of the partition passes
constexpr uint32_t PS_STILE = 8192;   // digits ordered in LDS at a time by the partition scatter: 8 per lane
constexpr uint32_t PS_TILE = 16384;   // references ordered in LDS at a time by the placement kernel (64 KiB): 16 per lane
constexpr uint32_t COMBINE_SMALL = 32;     // buckets spanning <= this many chunks: summed by one lane
constexpr uint32_t COMBINE_MEDIUM = 2048;  // <= this many: one wavefront per bucket; above: one workgroup
// the wide reduction (window tables with c > 16): see msm_reduce.hip
constexpr uint32_t WIDE_LOGG1 = 2;     // buckets per level-1 node
constexpr uint32_t WIDE_LOGK2 = 2;     // level-1 nodes per level-2 node
constexpr uint32_t WIDE_CHAINS = 128;  // level-2 nodes per virtual window
constexpr uint32_t WIDE_VB = WIDE_CHAINS << (WIDE_LOGG1 + WIDE_LOGK2);   // buckets per virtual window (2048)
constexpr uint32_t CHUNK_L = 32;      // references per lane on the per-window path
constexpr uint32_t PRE_C = 16;        // default window of the precomputed table
constexpr uint32_t PRE_C_MAX = 21;    // 2^20 shared buckets: 4096 per partition in the second sort pass (144 KiB of LDS)
constexpr uint32_t PRE_CHUNK_L = 128; // references per lane on the shared-bucket path (buckets hold ~W*n/2^15 each)
constexpr uint32_t PRE_Q_OFF = 1024;  // words of part_key in front of the combine queues (partition starts, totals, counter)
constexpr uint32_t PRE_VW = 64;       // virtual windows for the final bucket reduction: 512 buckets = 64 chains x 4 lanes per workgroup
                                      // (256 registers per lane; with 32 windows the 1024-lane workgroup spilled at 128)


// References per lane actually used.  The launch is sized for nf = L0 * n_lanes references, but the sorted list holds E <= nf (zero
// digits are not in it: sparse or small scalars).  Cutting the E references into n_lanes equal chunks keeps every lane of the launch
// busy -- with fixed chunks of L0 a list 8 % shorter leaves the last round of resident wavefronts 16 % empty and takes exactly as
// long.  msm_accumulate and the msm_combine* kernels derive the same value from the same inputs.
ZK_D uint32_t chunk_len(uint32_t E, uint32_t n_lanes, uint32_t L0) {
    const uint32_t need = (uint32_t)(((uint64_t)E + n_lanes - 1) / n_lanes);
    const uint32_t lo = L0 < 16u ? L0 : 16u;       // never below 16 (or L0): shorter chunks only multiply the chunk-edge partials
    return need < lo ? lo : need;
}

inline uint32_t ilog2_floor(uint64_t x) {
    uint32_t r = 0;
    while (x >>= 1) ++r;
    return r;
}

// top_of(shift) must return (r - 1) >> shift (low 32 bits): the largest value the last window can hold
template <class FrP>
uint32_t modulus_minus_one_bits(uint32_t shift) {
    uint32_t w[FrP::N + 1];
    for (int i = 0; i < FrP::N; ++i) w[i] = FrP::MOD(i);
    w[FrP::N] = 0;
    w[0] -= 1;  // r is odd
    uint32_t limb = shift >> 5, off = shift & 31;
    if (limb >= (uint32_t)FrP::N) return 0;
    uint64_t v = ((uint64_t)w[limb + 1] << 32 | w[limb]) >> off;
    return (uint32_t)v;
}

template <class FrP>
MsmGeom make_geom(uint64_t n, int c_override, uint32_t max_c = 16) {
    const int bits = FrP::BITS;
    MsmGeom g;
    uint32_t c;
    if (c_override > 0) {
        c = (uint32_t)c_override;
    } else {
        // per-window path: measured on MI355X over c = 9 .. 16 at every size (profiles/r02/r02_notes.md, "window of the per-window
        // path"): the bucket reduction (W * 2^(c-1) buckets) is what a large c pays for, and its kernels change shape with c, so
        // the best window is a step function of the size, not lg - 4 all the way (6.35 -> 5.04 ms at 2^20, 3.35 -> 2.23 at 2^18)
        uint32_t lg = ilog2_floor(n ? n : 1);
        if (lg <= 13) c = lg > 4 ? lg - 4 : 0;
        else if (lg == 14) c = 9;
        else if (lg <= 17) c = 10;
        else if (lg <= 20) c = 13;
        else if (lg == 21) c = 15;
        else c = 16;
        if (c < 3) c = 3;
    }
    if (c < 2) c = 2;
    if (c > max_c) c = max_c;   // per-window path: digits are stored as int16 and a window's histogram lives in LDS (16)
    g.c = c;
    g.W = (uint32_t)bits / c + 1;
    // the last window must never produce a carry: its largest raw value (top bits of r-1, plus the
    // incoming carry) has to stay below 2^(c-1); otherwise spend one more window
    if (modulus_minus_one_bits<FrP>((g.W - 1) * c) + 1 >= (1u << (c - 1))) g.W += 1;
    // the same count for scalars folded to k <= (r - 1) / 2 (MsmGeom::neg): used only where it removes a window
    {
        uint32_t Wn = ((uint32_t)bits - 1 + c - 1) / c;
        if (Wn == 0) Wn = 1;
        if (modulus_minus_one_bits<FrP>((Wn - 1) * c + 1) + 1 >= (1u << (c - 1))) Wn += 1;     // ((r - 1) / 2) >> shift = (r - 1) >> (shift + 1)
        // Folded scalars flip the digits' signs, and -(-2^(c-1)) does not fit the int16 digits of the c <= 16 paths at c = 16: the
        // fold is taken only below 16 bits or with the int32 digits of the wide path (c > 16).  (No supported curve asks for it at
        // c = 16 -- 16 windows either way for 254- and 255-bit scalars -- so this only guards a third curve or a changed rule.)
        if (Wn < g.W && FrP::N == 8 && c != 16) {
            g.W = Wn;
            g.neg = 1;
            uint32_t w[9];
            for (int i = 0; i < 8; ++i) w[i] = g.mod[i] = FrP::MOD(i);
            w[8] = 0;
            w[0] -= 1;      // r is odd
            for (int i = 0; i < 8; ++i) g.half[i] = (w[i] >> 1) | (w[i + 1] << 31);
        }
    }
    g.Wt = g.W;
    g.B = 1u << (c - 1);
    g.nb = g.W * g.B;
    g.logG = c - 1 < 4 ? c - 1 : 4;
    g.ns = g.B >> g.logG;
    uint32_t per = (g.ns + 255) / 256;
    g.logq = 0;
    while ((1u << g.logq) < per) ++g.logq;
    return g;
}

struct PrePlan {
    MsmGeom g, g1, gv;
    uint64_t nf;
    uint32_t chunk_l, n_lanes, max_lanes;
    size_t win_bytes;
    bool wide;          // c > 16: int32 digits, 2^(c-9) buckets per sort partition
    bool wide_red;      // more than 2^16 shared buckets: three-level device reduction (up to 2^16 the virtual-window reduction of the
                        // c = 16 table serves, with virtual windows of 1024 buckets)
    bool shared_stage;  // the sort's staging area (5-6 B per reference) is the ctx's, not the job's: the jobs of a round are placed one
                        // after the other (pre_queue_sort_rest) instead of by one launch per kernel
};

// From this many references per job (n = 2^24 at c = 20) the plan trades the last per cent of speed for memory: one staging area
// for all jobs of a round (2.6 GB per job at 2^25) and at most PRE_BIG_ROUNDS rounds of resident lanes (the chunk-edge partials of
// 26 rounds were 1.7 GB per job).  Sixteen deferred jobs of a 2^25 round then hold 41 GB instead of 194.
constexpr uint64_t PRE_BIG_NF = 1ull << 27;
constexpr uint32_t PRE_BIG_ROUNDS = 8;

// the table windows are 16 .. 21 bits: up to 16 bits the int16 partition sort (2^15 buckets = 256 partitions of 128)
inline bool pre_psort16(const PrePlan& pl) { return !pl.wide && pl.g1.nb % (1u << PS_LOB) == 0 && (pl.g1.nb >> PS_LOB) <= 256; }
// the device form of a round's result (its virtual-window sums left on the device) needs the quad-cooperative reduction of at most
// 128 power-of-two virtual windows
inline bool pre_partial_dev_ok(const PrePlan& p) {
    return !p.wide_red && p.gv.logq == 0 && p.gv.ns <= 256 && p.gv.W <= 128 && (p.gv.W & (p.gv.W - 1)) == 0;
}

}  // namespace zkmsm
using namespace zkmsm;

// ---- what one unit offers the others (this curve's build) --------------------------------------------------------------------------
// msm_sort.hip
//   per-window path: digits + LDS counting sort of all W windows (hist | block sums in mb.counts, digits in mb.tmp)
int ZK_SYM(pw_queue_sort)(zk_ctx* c, const MsmGeom& g, uint32_t S, MsmBufs& mb, const void* d_scalars, size_t n, hipStream_t st);
//   table path: the digit kernel of one job (the only reader of the caller's scalars), then the placement passes of a round's jobs
int ZK_SYM(pre_queue_digits)(zk_ctx* c, const PrePlan& pl, MsmBufs& mb, const void* d_scalars, size_t n, hipStream_t st, bool mont);
int ZK_SYM(pre_queue_sort_rest)(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, const size_t* lens, uint32_t n_jobs, hipStream_t st);
// msm_accumulate.hip
int ZK_SYM(pw_queue_accumulate)(zk_ctx* c, const MsmGeom& g, MsmBufs& mb, const void* d_bases, uint32_t n_lanes, hipStream_t st);
int ZK_SYM(pre_queue_accumulate)(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, const size_t* lens, const size_t* tab_offs, uint32_t n_jobs, zk_srs* s,
                                 hipStream_t st);
// msm_reduce.hip
int ZK_SYM(queue_reduce)(zk_ctx* c, const RJobs& jobs, uint32_t n_jobs, uint32_t nb, const MsmGeom& gr, hipStream_t st, bool queues_cleared, uint32_t raw);
//   d_winsums (optional, n_jobs pointers): the jobs' 2 VW virtual-window sums stay on the device, internal form, instead of going to h_win
int ZK_SYM(pre_queue_reduce)(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, uint32_t n_jobs, void* h_win, hipStream_t st, void* const* d_winsums);
//   ranks x n_pts points (internal form, rank-major) added element-wise into h_out (arkworks layout, pinned)
int ZK_SYM(queue_sum_winsums)(zk_ctx* c, const void* d_all, uint32_t ranks, uint32_t n_pts, void* h_out, hipStream_t st);
