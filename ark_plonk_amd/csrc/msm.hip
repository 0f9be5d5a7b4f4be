// Pippenger (bucket method) multi-scalar multiplication over G1 for gfx950 (MI355X).
//
// Replaces, behind the C ABI, ark_ec 0.3 `VariableBaseMSM::multi_scalar_mul` as called by the
// reference at plonk-core/src/commitment.rs:45,83 and through every `PC::commit` / `PC::open`
// (proof_system/prover.rs:213,289-291,312-317,361-363,387-389,459-469,579,582-591,606,609-618).
//
// Pipeline (all on the ctx stream; no host round trip until the window sums are read back).  On the window-table path every step is
// ONE launch per kernel for all the MSMs of a prover round (job = blockIdx.y, or a block range of the accumulation):
//   1. digits         signed c-bit digits of every scalar (window-major; into_repr fused for commits) + the slab counts of the sort's
//                     256 partitions: psort(w)_digits_hist, per job, queued when the job is submitted (the only reader of the input)
//   2. sort           the (point, sign) references by bucket, no global atomics:
//                       window-table path: psort_scan / psort(w)_scatter / psort(w)_final -- two-pass partition sort over the one
//                                          shared bucket set, the jobs of a round batched;
//                       per-window path:   msm_digits / msm_hist / msm_scan1/2/3 / msm_scatter -- LDS counting sort
//   3. msm_accumulate(_batch) every lane sums a fixed-length chunk of the sorted list with XYZZ mixed additions (no inversion); runs
//                     that cross a chunk edge are emitted as partials (load-balanced regardless of the scalar distribution); the
//                     jobs of a round follow each other inside one launch
//   4. msm_combine*   joins the chunk-edge partials of each bucket: lane(s) / wavefront (shuffle tree) /
//                     workgroup per bucket by size class
//   5. msm_seg_reduce segmented running-sum reduction (sum_j j*B_j), level 1
//   6. msm_win_finish LDS suffix-scan + tree reduction per (virtual) window -> window sums (arkworks layout), or -- the multi-GPU
//                     exchange -- one more launch that leaves every job's whole sum on the device (zk_kzg_round_end_partial_dev)
//   host: the few window sums are combined (table path: sum of 64 virtual windows; per-window path:
//         Horner with W*c doublings) and normalised to affine.
// Device arithmetic is the signed 30-bit-limb Montgomery field of fields.cuh with the lazy XYZZ
// group law of ecu.cuh; bases are converted once, at SRS registration, into that form.
// The group sum is order-independent, so the non-deterministic order inside a bucket (atomic
// cursors) does not change the (canonical, affine) result.
// Algorithmic bytes per MSM: N * (32 + 2*Fq bytes); the kernel is integer-VALU bound.
#include "ctx.h"
#include "ecq.cuh"

#include <chrono>
#include <cstdio>

namespace {

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

// c-bit field of a canonical scalar (8 x u32 limbs in registers: selects, no indexing) at bit position pos
ZK_D uint32_t scalar_bits(const uint32_t (&s)[8], uint32_t pos, uint32_t c) {
    const uint32_t limb = pos >> 5, off = pos & 31;
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((uint32_t)k == limb) lo = s[k];
        if ((uint32_t)k == limb + 1) hi = s[k];
    }
    const uint64_t v = ((hi << 32) | lo) >> off;
    return (uint32_t)v & ((1u << c) - 1u);
}

// the negated-scalar rule of MsmGeom::neg: k <- r - k when k > (r - 1) / 2; returns whether it did (the digits' signs flip)
ZK_D bool scalar_fold(uint32_t (&k)[8], const MsmGeom& g) {
    if (!g.neg) return false;
    // the ABI asks for canonical scalars, and the fold needs k < r: an unreduced k < 2^256 is brought below r first (at most 5
    // subtractions for a 254-bit r), so it is still multiplied as k mod r instead of silently as garbage
#pragma unroll 1
    for (int it = 0; it < 6; ++it) {
        bool ge = true, dec = false;
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            if (!dec && k[i] != g.mod[i]) {
                ge = k[i] > g.mod[i];
                dec = true;
            }
        }
        if (!ge) break;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t d = (uint64_t)k[i] - g.mod[i] - br;
            k[i] = (uint32_t)d;
            br = (uint32_t)(d >> 63);
        }
    }
    bool gt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (!decided && k[i] != g.half[i]) {
            gt = k[i] > g.half[i];
            decided = true;
        }
    }
    if (!gt) return false;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t d = (uint64_t)g.mod[i] - k[i] - borrow;
        k[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    return true;
}

// ---- counting sort of the (point, sign) references by (window, bucket), without global atomics ----
// K0  msm_digits : signed c-bit digits of every scalar, stored window-major as int16 (c <= 16)
// K1  msm_hist   : one workgroup per (window, slab of scalars): LDS histogram -> hist[w][slab][bucket]
// K2  msm_scan1/2/3 : exclusive scan in (window, bucket, slab) order -> bucket offsets + per-slab cursors
// K3  msm_scatter: same grid as K1, LDS cursors, writes the references to their sorted position
// digit convention: raw = bits + carry; raw >= 2^(c-1) -> digit raw - 2^c (negative), carry 1.
__global__ void msm_digits(const uint32_t* scalars, uint64_t n, MsmGeom g, int16_t* dig) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * i;
    const uint4 a = q[0], b = q[1];
    uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const bool flip = scalar_fold(s, g);
    uint32_t carry = 0;
    const uint32_t half = 1u << (g.c - 1);
    for (uint32_t w = 0; w < g.Wt; ++w) {
        uint32_t raw = scalar_bits(s, w * g.c, g.c) + carry;
        carry = raw >= half ? 1u : 0u;
        int32_t d = carry ? (int32_t)raw - (int32_t)(1u << g.c) : (int32_t)raw;
        uint32_t row;
        // (the int16 store cannot hold -(-32768): make_geom never folds scalars at c = 16, see the assert there)
        if (g.owns(w, row)) dig[(uint64_t)row * n + i] = (int16_t)(flip ? -d : d);
    }
}

// Two scalars per lane (n even; c = 16, W = 16): 32-byte vector loads, one 4-byte store per window instead of two 2-byte
// ones.  MONT: the input is a Montgomery coefficient (a commit): into_repr is fused here instead of a
// separate conversion pass over the vector.
template <class Fr, bool MONT>
__global__ void msm_digits2(const uint32_t* scalars, uint64_t n, MsmGeom g, int16_t* dig) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t >= n) return;
    uint32_t s[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * (2 * t + h);
        uint4 a = q[0], b = q[1];
        Fr x;
        x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
        x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
        if (MONT) x = Fr::from_mont(x);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[h][i] = x.v[i];
    }
    // 16-bit windows, 16 of them (the window-table geometry): digit w is half-word w of the scalar
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        uint32_t r0 = ((s[0][w >> 1] >> (16 * (w & 1))) & 0xffffu) + c0;
        uint32_t r1 = ((s[1][w >> 1] >> (16 * (w & 1))) & 0xffffu) + c1;
        c0 = r0 >= 0x8000u ? 1u : 0u;     // raw >= 2^15 -> digit raw - 2^16 (its low 16 bits are unchanged), carry 1
        c1 = r1 >= 0x8000u ? 1u : 0u;
        *reinterpret_cast<uint32_t*>(dig + (uint64_t)w * n + 2 * t) = (r0 & 0xffffu) | (r1 << 16);
    }
}

ZK_D void slab_range(uint64_t n, uint32_t S, uint32_t slab, uint64_t& lo, uint64_t& hi) {
    const uint64_t per = (n + S - 1) / S;
    lo = (uint64_t)slab * per;
    hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = n;
}

__global__ void msm_hist(const int16_t* dig, uint64_t n, MsmGeom g, uint32_t S, uint32_t* hist) {
    extern __shared__ uint32_t lh[];
    const uint32_t w = blockIdx.y, slab = blockIdx.x;
    for (uint32_t j = threadIdx.x; j < g.B; j += blockDim.x) lh[j] = 0;
    __syncthreads();
    uint64_t lo, hi;
    slab_range(n, S, slab, lo, hi);
    const int16_t* dw = dig + (uint64_t)w * n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int32_t d = dw[i];
        if (d != 0) atomicAdd(&lh[(d < 0 ? -d : d) - 1], 1u);
    }
    __syncthreads();
    uint32_t* out = hist + ((uint64_t)w * S + slab) * g.B;
    for (uint32_t j = threadIdx.x; j < g.B; j += blockDim.x) out[j] = lh[j];
}

ZK_D uint32_t bucket_total(const uint32_t* hist, MsmGeom g, uint32_t S, uint32_t k) {
    const uint32_t w = k / g.B, j = k % g.B;
    uint32_t t = 0;
    for (uint32_t s = 0; s < S; ++s) t += hist[((uint64_t)w * S + s) * g.B + j];
    return t;
}

// block sums of the per-bucket totals (1024 buckets per block)
__global__ void msm_scan1(const uint32_t* hist, MsmGeom g, uint32_t S, uint32_t* bsum) {
    __shared__ uint32_t red[1024];
    const uint32_t k = blockIdx.x * 1024 + threadIdx.x;
    red[threadIdx.x] = k < g.nb ? bucket_total(hist, g, S, k) : 0u;
    __syncthreads();
    for (uint32_t d = 512; d >= 1; d >>= 1) {
        if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
// exclusive scan of the block sums (<= 1024 of them), in place
__global__ void msm_scan2(uint32_t* bsum, uint32_t nblk) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t v = t < nblk ? bsum[t] : 0u;
    part[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t o = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    if (t < nblk) bsum[t] = part[t] - v;
}
// bucket offsets (offsets[k], offsets[nb] = total) and per-(window, slab, bucket) cursors (in place in hist)
__global__ void msm_scan3(uint32_t* hist, MsmGeom g, uint32_t S, const uint32_t* bsum, uint32_t* offsets) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t k = blockIdx.x * 1024 + t;
    const uint32_t tot = k < g.nb ? bucket_total(hist, g, S, k) : 0u;
    part[t] = tot;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t o = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    if (k >= g.nb) return;
    uint32_t run = bsum[blockIdx.x] + part[t] - tot;
    offsets[k] = run;
    if (k == g.nb - 1) offsets[g.nb] = run + tot;
    const uint32_t w = k / g.B, j = k % g.B;
    for (uint32_t s = 0; s < S; ++s) {
        uint32_t* p = hist + ((uint64_t)w * S + s) * g.B + j;
        const uint32_t cnt = *p;
        *p = run;
        run += cnt;
    }
}

// n_real != 0: dig is the flattened [W][n_real] array sorted as ONE window (shared bucket set); the
// reference written is sign<<31 | window<<26 | index.
__global__ void msm_scatter(const int16_t* dig, uint64_t n, MsmGeom g, uint32_t S, const uint32_t* cursors, uint32_t* entries,
                            uint32_t n_real) {
    extern __shared__ uint32_t lh[];
    const uint32_t w = blockIdx.y, slab = blockIdx.x;
    const uint32_t* cur = cursors + ((uint64_t)w * S + slab) * g.B;
    for (uint32_t j = threadIdx.x; j < g.B; j += blockDim.x) lh[j] = cur[j];
    __syncthreads();
    uint64_t lo, hi;
    slab_range(n, S, slab, lo, hi);
    const int16_t* dw = dig + (uint64_t)w * n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int32_t d = dw[i];
        if (d == 0) continue;
        const uint32_t neg = d < 0 ? 1u : 0u;
        const uint32_t pos = atomicAdd(&lh[(neg ? -d : d) - 1], 1u);
        uint32_t ref = (uint32_t)i;
        if (n_real) {
            const uint32_t wq = (uint32_t)i / n_real;
            ref = (wq << 26) | ((uint32_t)i - wq * n_real);
        }
        entries[pos] = ref | (neg << 31);
    }
}

// ---- two-pass partition sort for the shared-bucket path (one "window" of nf digits, nb buckets) ----
// The single-pass counting sort above leaves every slab only ~4 references per bucket, so its scatter
// writes 16-byte runs at random places (measured: 513 MB leaving L2 per launch for 67 MB of output).
// Here the references first go to P = nb/128 partitions by the high bucket bits -- every (slab, partition)
// run is ~1 KiB contiguous -- and one workgroup per partition then orders its ~nf/P references by the low
// 7 bits out of L2.  psort_digits_hist (or psort_hist) / psort_scan / psort_scatter / psort_final; order
// inside a bucket is arbitrary (the sums are commutative).  Measured at 2^20: 0.23 ms against 0.33 ms for
// msm_hist + msm_scan1/2/3 + msm_scatter; what is left is the ~64 distinct cache lines every wave-store of
// the two placement kernels touches.
constexpr uint32_t PS_LOB = 7;        // low bucket bits ordered inside a partition
constexpr uint32_t PS_T = 1024;
constexpr uint32_t PS_SLABS = 1024;   // workgroups of the partition passes

// A slab of the partition sort = a range of SCALARS with all their W digits (dig[w*n + i], i in the range), so
// that the kernel that produces the digits can count them too.  sp = scalars per slab (even).
ZK_HD uint32_t psort_slab_len(uint64_t n) {
    uint64_t sp = (n + PS_SLABS - 1) / PS_SLABS;
    sp = (sp + 1) & ~1ull;
    return (uint32_t)(sp < 2 ? 2 : sp);
}

// digits of the slab's scalars (as msm_digits2: two scalars per lane, 16-bit windows) + the slab's partition counts
template <class Fr, bool MONT>
__global__ void __launch_bounds__(256) psort_digits_hist(const uint32_t* scalars, uint64_t n, uint32_t sp, int16_t* dig,
                                                         uint32_t* hist /* [256][PS_SLABS] */, uint32_t* scan_counter, uint32_t* combine_q) {
    __shared__ uint32_t lc[256];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scan_counter[0] = 0;
        combine_q[0] = 0;      // counters of the combine queues of this job (msm_combine*)
        combine_q[1] = 0;
    }
    lc[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * sp < n ? (uint64_t)blockIdx.x * sp : n;
    const uint64_t hi = lo + sp < n ? lo + sp : n;      // n even, sp even: the range holds whole pairs
    for (uint64_t i0 = lo + 2 * threadIdx.x; i0 < hi; i0 += 512) {
        uint32_t sc[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * (i0 + h);
            uint4 a = q[0], b = q[1];
            Fr x;
            x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
            x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
            if (MONT) x = Fr::from_mont(x);
#pragma unroll
            for (int k = 0; k < 8; ++k) sc[h][k] = x.v[k];
        }
        uint32_t c0 = 0, c1 = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            uint32_t r0 = ((sc[0][w >> 1] >> (16 * (w & 1))) & 0xffffu) + c0;
            uint32_t r1 = ((sc[1][w >> 1] >> (16 * (w & 1))) & 0xffffu) + c1;
            c0 = r0 >= 0x8000u ? 1u : 0u;
            c1 = r1 >= 0x8000u ? 1u : 0u;
            r0 &= 0xffffu;
            r1 &= 0xffffu;
            *reinterpret_cast<uint32_t*>(dig + (uint64_t)w * n + i0) = r0 | (r1 << 16);
            if (r0) atomicAdd(&lc[((c0 ? 0x10000u - r0 : r0) - 1u) >> PS_LOB], 1u);
            if (r1) atomicAdd(&lc[((c1 ? 0x10000u - r1 : r1) - 1u) >> PS_LOB], 1u);
        }
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * PS_SLABS + blockIdx.x] = lc[threadIdx.x];
}

// the same counts from an existing digit array (lengths the fused kernel does not take)
__global__ void __launch_bounds__(PS_T) psort_hist(const int16_t* dig, uint64_t n, uint32_t W, uint32_t sp, uint32_t P,
                                                   uint32_t* hist /* [P][PS_SLABS] */, uint32_t* scan_counter, uint32_t* combine_q) {
    extern __shared__ uint32_t lc[];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scan_counter[0] = 0;
        combine_q[0] = 0;
        combine_q[1] = 0;
    }
    for (uint32_t j = threadIdx.x; j < P; j += PS_T) lc[j] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * sp < n ? (uint64_t)blockIdx.x * sp : n;
    const uint64_t hi = lo + sp < n ? lo + sp : n;
    const uint32_t len = (uint32_t)(hi - lo);
    for (uint32_t q = threadIdx.x; q < W * len; q += PS_T) {
        const uint32_t w = q / len, ii = q - w * len;
        const int32_t d = dig[(uint64_t)w * n + lo + ii];
        if (d != 0) atomicAdd(&lc[(uint32_t)((d < 0 ? -d : d) - 1) >> PS_LOB], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < P; j += PS_T) hist[(uint64_t)j * PS_SLABS + blockIdx.x] = lc[j];
}

ZK_D uint32_t scan1024_excl(uint32_t v, uint32_t t, uint32_t* tmp);

// per partition: exclusive scan of its PS_SLABS slab counts in place (coalesced); the workgroup that finishes
// last (a counter, no waiting) then scans the P (<= 256) partition totals into part_start[0..P].
// `counter` must be 0 on entry (psort_hist clears it) and is left 0.  Both scans are wave shuffles plus one LDS step
// (scan1024_excl): as twenty-barrier Hillis-Steele loops over LDS this kernel was 13 us of every MSM's sort.
__global__ void __launch_bounds__(PS_SLABS) psort_scan(SJobs jobs, uint32_t P) {
    static_assert(PS_SLABS == 1024, "scan1024_excl");
    uint32_t* hist = jobs.j[blockIdx.y].hist;
    uint32_t* part_total = jobs.j[blockIdx.y].part_total;
    uint32_t* part_start = jobs.j[blockIdx.y].part_start;
    uint32_t* counter = jobs.j[blockIdx.y].counter;
    __shared__ uint32_t tmp[16];
    __shared__ uint32_t last_block;
    const uint32_t t = threadIdx.x;
    uint32_t* row = hist + (uint64_t)blockIdx.x * PS_SLABS;
    const uint32_t v = row[t];
    const uint32_t ex = scan1024_excl(v, t, tmp);
    row[t] = ex;
    if (t == PS_SLABS - 1) {
        part_total[blockIdx.x] = ex + v;
        __threadfence();
        last_block = atomicAdd(counter, 1u) == P - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_block) return;
    const uint32_t tv = t < P ? ((volatile uint32_t*)part_total)[t] : 0u;
    const uint32_t ex2 = scan1024_excl(tv, t, tmp);
    if (t < P) part_start[t] = ex2;
    if (t == PS_SLABS - 1) {
        part_start[P] = ex2 + tv;
        *counter = 0;
    }
}

// exclusive scan of 256 values held by lanes 0..255 of a workgroup (every lane calls it); tmp: 4 LDS words
ZK_D uint32_t scan256_excl(uint32_t v, uint32_t t, uint32_t* tmp) {
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((t & 63) >= d) inc += o;
    }
    if (t < 256 && (t & 63) == 63) tmp[t >> 6] = inc;
    __syncthreads();
    uint32_t add = 0;
    for (uint32_t w = 0; w < (t >> 6) && w < 4; ++w) add += tmp[w];
    __syncthreads();
    return inc + add - v;
}

// references (sign<<31 | window<<26 | index, as msm_scatter writes them) + their low bucket bits -> partition order.
// A tile of PS_STILE digits is ordered by partition in LDS first (packed: position in the tile, sign, low bits,
// partition), so the 8-byte records leave as runs of consecutive addresses, one run per partition and tile.
constexpr uint32_t PS_STILE = 8192;   // 8 digits per lane
__global__ void __launch_bounds__(PS_T) psort_scatter(SJobs jobs, uint32_t W, uint32_t P) {
    constexpr uint32_t PER = PS_STILE / PS_T, LOM = (1u << PS_LOB) - 1u;
    const SJob& J = jobs.j[blockIdx.y];
    const int16_t* dig = (const int16_t*)J.dig;
    const uint64_t n = J.n;
    const uint32_t sp = J.sp;
    const uint32_t* cursors = J.hist;
    const uint32_t* part_start = J.part_start;
    uint32_t* stage_ref = J.stage_ref;
    uint8_t* stage_lo = (uint8_t*)J.stage_lo;
    __shared__ uint32_t cnt[256], toff[257], gcur[256], stmp[4];
    __shared__ uint32_t rec[PS_STILE];       // k (14 bits) | neg << 14 | low bits << 15 | partition << 22
    const uint32_t t = threadIdx.x;
    if (t < 256) gcur[t] = t < P ? part_start[t] + cursors[(uint64_t)t * PS_SLABS + blockIdx.x] : 0u;
    const uint64_t lo = (uint64_t)blockIdx.x * sp < n ? (uint64_t)blockIdx.x * sp : n;
    const uint64_t hi = lo + sp < n ? lo + sp : n;
    const uint32_t len = (uint32_t)(hi - lo);
    const uint32_t total_digits = W * len;           // the slab: W windows x len scalars, visited window-major
    // the digits of the tile after the current one are requested while the current one is counted and placed
    int32_t nd[PER];
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t q = base + k * PS_T + t;
            nd[k] = 0;
            if (q < total_digits) {
                const uint32_t w = q / len, ii = q - w * len;
                nd[k] = dig[(uint64_t)w * n + lo + ii];
            }
        }
    };
    if (total_digits) fetch(0);
    for (uint32_t base = 0; base < total_digits; base += PS_STILE) {
        __syncthreads();
        if (t < 256) cnt[t] = 0;
        __syncthreads();
        int32_t vd[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) vd[k] = nd[k];
        if (base + PS_STILE < total_digits) fetch(base + PS_STILE);
        uint32_t pk[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            pk[k] = 0xffffffffu;
            const int32_t d = vd[k];      // 0 past the end of the slab
            if (d != 0) {
                const uint32_t neg = d < 0 ? 1u : 0u;
                const uint32_t b = (uint32_t)((neg ? -d : d) - 1);
                pk[k] = i | (neg << 14) | ((b & LOM) << 15) | ((b >> PS_LOB) << 22);
                atomicAdd(&cnt[b >> PS_LOB], 1u);
            }
        }
        __syncthreads();
        {
            const uint32_t c = t < 256 ? cnt[t] : 0u;
            const uint32_t ex = scan256_excl(c, t, stmp);
            if (t < 256) toff[t] = ex;
            if (t == 255) toff[256] = ex + c;
        }
        __syncthreads();
        if (t < 256) cnt[t] = toff[t];
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k)
            if (pk[k] != 0xffffffffu) rec[atomicAdd(&cnt[pk[k] >> 22], 1u)] = pk[k];
        __syncthreads();
        const uint32_t total = toff[256];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t qq = k * PS_T + t;   // consecutive lanes -> consecutive records of a partition's run
            if (qq < total) {
                const uint32_t r = rec[qq];
                const uint32_t pp = r >> 22;
                const uint32_t q = base + (r & 0x3fffu), w = q / len, ii = q - w * len;
                const uint32_t ref = (w << 26) | (uint32_t)(lo + ii) | (((r >> 14) & 1u) << 31);
                const uint32_t dst = gcur[pp] + (qq - toff[pp]);
                stage_ref[dst] = ref;
                stage_lo[dst] = (uint8_t)((r >> 15) & LOM);
            }
        }
        __syncthreads();
        if (t < 256) gcur[t] += toff[t + 1] - toff[t];
    }
}

// one workgroup per partition: count the low bits, publish the bucket offsets, then place the references
// tile by tile: a tile of PS_TILE references is ordered in LDS first, so that the global stores are runs
// of consecutive addresses (one run per bucket and tile) instead of 64 different cache lines per wave-store.
// (Wave-private counters were tried for the counting: 16 x 128 write streams per workgroup made it slower.)
constexpr uint32_t PS_TILE = 16384;   // references ordered in LDS at a time (64 KiB) -- 16 per lane

// exclusive scan of 128 values held by lanes 0..127 of a workgroup (every lane calls it); tmp: one LDS word
ZK_D uint32_t scan128_excl(uint32_t v, uint32_t t, uint32_t* tmp) {
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((t & 63) >= d) inc += o;
    }
    if (t == 63) *tmp = inc;
    __syncthreads();
    if (t >= 64 && t < 128) inc += *tmp;
    return inc - v;
}
// counts of the keys key[i], i = first, first + PS_T, ... < end, into the LDS table cnt.  Eight loads in flight per lane: written
// as a plain loop the compiler keeps ONE (load, wait, LDS atomic) per iteration, and the pass over a partition's ~60 keys per
// lane was sixty memory round trips in a row -- most of the kernel's time.
template <class K>
ZK_D void count_keys(const K* key, uint32_t first, uint32_t end, uint32_t* cnt) {
    uint32_t i = first;
    for (; i + 7 * PS_T < end; i += 8 * PS_T) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = key[i + k * PS_T];
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&cnt[v[k]], 1u);
    }
    for (; i < end; i += PS_T) atomicAdd(&cnt[key[i]], 1u);
}
__global__ void __launch_bounds__(PS_T) psort_final(SJobs jobs, uint32_t P) {
    constexpr uint32_t NB = 1u << PS_LOB, PER = PS_TILE / PS_T;
    const SJob& J = jobs.j[blockIdx.y];
    const uint32_t* stage_ref = J.stage_ref;
    const uint8_t* stage_lo = (const uint8_t*)J.stage_lo;
    const uint32_t* part_start = J.part_start;
    uint32_t* entries = J.entries;
    uint32_t* offsets = J.offsets;
    __shared__ uint32_t cnt[NB], cur[NB], toff[NB + 1], stmp;
    __shared__ uint32_t sorted[PS_TILE];
    __shared__ uint8_t skey[PS_TILE];
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t s = part_start[p], e = part_start[p + 1];
    if (t < NB) cnt[t] = 0;
    __syncthreads();
    count_keys(stage_lo, s + t, e, cnt);
    __syncthreads();
    {
        const uint32_t ex = scan128_excl(t < NB ? cnt[t] : 0u, t, &stmp);
        if (t < NB) cur[t] = s + ex;
    }
    __syncthreads();
    if (t < NB) offsets[p * NB + t] = cur[t];
    if (p == P - 1 && t == 0) offsets[P * NB] = e;
    // the tile after the current one is requested while the current one is counted, scanned and placed
    uint2 nv[PER];
    auto fetch = [&](uint32_t base) {
        const uint32_t m = e - base < PS_TILE ? e - base : PS_TILE;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) nv[k] = make_uint2(stage_ref[base + i], stage_lo[base + i]);
        }
    };
    if (s < e) fetch(s);
    for (uint32_t base = s; base < e; base += PS_TILE) {
        const uint32_t m = e - base < PS_TILE ? e - base : PS_TILE;   // references in this tile
        __syncthreads();
        if (t < NB) cnt[t] = 0;
        __syncthreads();
        uint2 v[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) v[k] = nv[k];
        if (base + PS_TILE < e) fetch(base + PS_TILE);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) atomicAdd(&cnt[v[k].y], 1u);
        }
        __syncthreads();
        {
            const uint32_t c = t < NB ? cnt[t] : 0u;
            const uint32_t ex = scan128_excl(c, t, &stmp);
            if (t < NB) toff[t] = ex;
            if (t == NB - 1) toff[NB] = ex + c;
        }
        __syncthreads();
        if (t < NB) cnt[t] = toff[t];      // running position inside the tile
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) {
                const uint32_t q = atomicAdd(&cnt[v[k].y], 1u);
                sorted[q] = v[k].x;
                skey[q] = (uint8_t)v[k].y;
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;   // consecutive lanes -> consecutive positions of a bucket's run
            if (i < m) {
                const uint32_t j = skey[i];
                entries[cur[j] + (i - toff[j])] = sorted[i];
            }
        }
        __syncthreads();
        if (t < NB) cur[t] += toff[t + 1] - toff[t];
    }
}


// ---- the same partition sort for window tables with c > 16 (2^(c-1) shared buckets, c <= 21) ---------------------------
// Still P = 256 partitions by the high 8 bucket bits; the low part grows to lob = c - 9 bits (128 ... 4096 buckets per
// partition), so the digits are int32, the staged low bits uint16 and the LDS tables of the second pass are sized at
// launch.  One scalar per lane in the digit kernel (a 4-byte store per window either way).
template <class Fr, bool MONT>
__global__ void __launch_bounds__(256) psortw_digits_hist(const uint32_t* scalars, uint64_t n, uint32_t sp, MsmGeom g, uint32_t lob, int32_t* dig,
                                                          uint32_t* hist /* [256][PS_SLABS] */, uint32_t* scan_counter, uint32_t* combine_q) {
    __shared__ uint32_t lc[256];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scan_counter[0] = 0;
        combine_q[0] = 0;
        combine_q[1] = 0;
    }
    lc[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * sp < n ? (uint64_t)blockIdx.x * sp : n;
    const uint64_t hi = lo + sp < n ? lo + sp : n;
    const uint32_t half = 1u << (g.c - 1), cmask = (1u << g.c) - 1u;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const uint4* q = reinterpret_cast<const uint4*>(scalars) + 2 * i;
        uint4 a = q[0], b = q[1];
        Fr x;
        x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
        x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
        if (MONT) x = Fr::from_mont(x);
        const bool flip = scalar_fold(x.v, g);
        uint32_t carry = 0;
        for (uint32_t w = 0; w < g.Wt; ++w) {
            const uint32_t raw = (scalar_bits(x.v, w * g.c, g.c) & cmask) + carry;
            carry = raw >= half ? 1u : 0u;
            int32_t d = carry ? (int32_t)raw - (int32_t)(1u << g.c) : (int32_t)raw;
            if (flip) d = -d;
            uint32_t row;
            if (!g.owns(w, row)) continue;
            dig[(uint64_t)row * n + i] = d;
            if (d != 0) atomicAdd(&lc[(uint32_t)((d < 0 ? -d : d) - 1) >> lob], 1u);
        }
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * PS_SLABS + blockIdx.x] = lc[threadIdx.x];
}


__global__ void __launch_bounds__(PS_T) psortw_scatter(SJobs jobs, uint32_t W, uint32_t lob) {
    constexpr uint32_t PER = PS_STILE / PS_T;
    const SJob& J = jobs.j[blockIdx.y];
    const int32_t* dig = (const int32_t*)J.dig;
    const uint64_t n = J.n;
    const uint32_t sp = J.sp;
    const uint32_t* cursors = J.hist;
    const uint32_t* part_start = J.part_start;
    uint32_t* stage_ref = J.stage_ref;
    uint16_t* stage_lo = (uint16_t*)J.stage_lo;
    const uint32_t LOM = (1u << lob) - 1u;
    __shared__ uint32_t cnt[256], toff[257], gcur[256], stmp[4];
    __shared__ uint32_t rec[PS_STILE];       // k (14 bits) | neg << 14 | partition << 15
    __shared__ uint16_t rlo[PS_STILE];       // low bucket bits of the record at the same position
    const uint32_t t = threadIdx.x;
    if (t < 256) gcur[t] = part_start[t] + cursors[(uint64_t)t * PS_SLABS + blockIdx.x];
    const uint64_t lo = (uint64_t)blockIdx.x * sp < n ? (uint64_t)blockIdx.x * sp : n;
    const uint64_t hi = lo + sp < n ? lo + sp : n;
    const uint32_t len = (uint32_t)(hi - lo);
    const uint32_t total_digits = W * len;
    // the digits of the tile after the current one are requested while the current one is counted and placed
    int32_t nd[PER];
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t q = base + k * PS_T + t;
            nd[k] = 0;
            if (q < total_digits) {
                const uint32_t w = q / len, ii = q - w * len;
                nd[k] = dig[(uint64_t)w * n + lo + ii];
            }
        }
    };
    if (total_digits) fetch(0);
    for (uint32_t base = 0; base < total_digits; base += PS_STILE) {
        __syncthreads();
        if (t < 256) cnt[t] = 0;
        __syncthreads();
        int32_t vd[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) vd[k] = nd[k];
        if (base + PS_STILE < total_digits) fetch(base + PS_STILE);
        uint32_t pk[PER];
        uint16_t pl[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            pk[k] = 0xffffffffu;
            pl[k] = 0;
            const int32_t d = vd[k];      // 0 past the end of the slab
            if (d != 0) {
                const uint32_t neg = d < 0 ? 1u : 0u;
                const uint32_t b = (uint32_t)((neg ? -d : d) - 1);
                pk[k] = i | (neg << 14) | ((b >> lob) << 15);
                pl[k] = (uint16_t)(b & LOM);
                atomicAdd(&cnt[b >> lob], 1u);
            }
        }
        __syncthreads();
        {
            const uint32_t c = t < 256 ? cnt[t] : 0u;
            const uint32_t ex = scan256_excl(c, t, stmp);
            if (t < 256) toff[t] = ex;
            if (t == 255) toff[256] = ex + c;
        }
        __syncthreads();
        if (t < 256) cnt[t] = toff[t];
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k)
            if (pk[k] != 0xffffffffu) {
                const uint32_t at = atomicAdd(&cnt[pk[k] >> 15], 1u);
                rec[at] = pk[k];
                rlo[at] = pl[k];
            }
        __syncthreads();
        const uint32_t total = toff[256];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t qq = k * PS_T + t;
            if (qq < total) {
                const uint32_t r = rec[qq];
                const uint32_t pp = r >> 15;
                const uint32_t q = base + (r & 0x3fffu), w = q / len, ii = q - w * len;
                const uint32_t ref = (w << 26) | (uint32_t)(lo + ii) | (((r >> 14) & 1u) << 31);
                const uint32_t dst = gcur[pp] + (qq - toff[pp]);
                stage_ref[dst] = ref;
                stage_lo[dst] = rlo[qq];
            }
        }
        __syncthreads();
        if (t < 256) gcur[t] += toff[t + 1] - toff[t];
    }
}

// exclusive scan of one value per lane over a 1024-lane workgroup; tmp: 16 LDS words
ZK_D uint32_t scan1024_excl(uint32_t v, uint32_t t, uint32_t* tmp) {
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((t & 63) >= d) inc += o;
    }
    __syncthreads();
    if ((t & 63) == 63) tmp[t >> 6] = inc;
    __syncthreads();
    uint32_t add = 0;
    for (uint32_t w = 0; w < (t >> 6); ++w) add += tmp[w];
    return inc + add - v;
}

// one workgroup per partition, NB = 2^lob buckets; dynamic LDS: cnt[NB] | cur[NB] | toff[NB + 1] | tmp[16] | sorted[PS_TILE] | skey[PS_TILE] (u16)
__global__ void __launch_bounds__(PS_T) psortw_final(SJobs jobs, uint32_t P, uint32_t lob) {
    extern __shared__ uint32_t lds[];
    const SJob& J = jobs.j[blockIdx.y];
    const uint32_t* stage_ref = J.stage_ref;
    const uint16_t* stage_lo = (const uint16_t*)J.stage_lo;
    const uint32_t* part_start = J.part_start;
    uint32_t* entries = J.entries;
    uint32_t* offsets = J.offsets;
    constexpr uint32_t PER = PS_TILE / PS_T;
    const uint32_t NB = 1u << lob;
    const uint32_t K = NB > PS_T ? NB / PS_T : 1u;       // counters per lane in the scans
    uint32_t* cnt = lds;
    uint32_t* cur = cnt + NB;
    uint32_t* toff = cur + NB;
    uint32_t* tmp = toff + NB + 1;
    uint32_t* sorted = tmp + 16;
    uint16_t* skey = reinterpret_cast<uint16_t*>(sorted + PS_TILE);
    const uint32_t p = blockIdx.x, t = threadIdx.x;
    const uint32_t s = part_start[p], e = part_start[p + 1];
    for (uint32_t j = t; j < NB; j += PS_T) cnt[j] = 0;
    __syncthreads();
    count_keys(stage_lo, s + t, e, cnt);
    __syncthreads();
    // exclusive scan of cnt[0 .. NB): lane t owns counters [t*K, (t+1)*K)
    auto scan_counts = [&](uint32_t* dst, uint32_t add, bool with_total) {
        uint32_t mine = 0;
        if (t * K < NB)
            for (uint32_t k = 0; k < K; ++k) mine += cnt[t * K + k];
        uint32_t ex = scan1024_excl(mine, t, tmp);
        if (t * K < NB) {
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t c = cnt[t * K + k];
                dst[t * K + k] = add + ex;
                ex += c;
            }
            if (with_total && (t + 1) * K == NB) dst[NB] = add + ex;
        }
        __syncthreads();
    };
    scan_counts(cur, s, false);
    for (uint32_t j = t; j < NB; j += PS_T) offsets[p * NB + j] = cur[j];
    if (p == P - 1 && t == 0) offsets[P * NB] = e;
    // the tile after the current one is requested while the current one is counted, scanned and placed: its 16 references and
    // keys per lane sit in registers across the barriers instead of costing a memory round trip at the top of every tile
    uint32_t nr[PER];
    uint16_t nk[PER];
    auto fetch = [&](uint32_t base) {
        const uint32_t m = e - base < PS_TILE ? e - base : PS_TILE;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) {
                nr[k] = stage_ref[base + i];
                nk[k] = stage_lo[base + i];
            }
        }
    };
    if (s < e) fetch(s);
    for (uint32_t base = s; base < e; base += PS_TILE) {
        const uint32_t m = e - base < PS_TILE ? e - base : PS_TILE;
        __syncthreads();
        for (uint32_t j = t; j < NB; j += PS_T) cnt[j] = 0;
        __syncthreads();
        uint32_t vr[PER];
        uint16_t vk[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            vr[k] = nr[k];
            vk[k] = nk[k];
        }
        if (base + PS_TILE < e) fetch(base + PS_TILE);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) atomicAdd(&cnt[vk[k]], 1u);
        }
        __syncthreads();
        scan_counts(toff, 0u, true);
        for (uint32_t j = t; j < NB; j += PS_T) cnt[j] = toff[j];      // running position inside the tile
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) {
                const uint32_t q = atomicAdd(&cnt[vk[k]], 1u);
                sorted[q] = vr[k];
                skey[q] = vk[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t i = k * PS_T + t;
            if (i < m) {
                const uint32_t j = skey[i];
                entries[cur[j] + (i - toff[j])] = sorted[i];
            }
        }
        __syncthreads();
        for (uint32_t j = t; j < NB; j += PS_T) cur[j] += toff[j + 1] - toff[j];
    }
}

// References per lane actually used.  The launch is sized for nf = L0 * n_lanes references, but the sorted list holds E <= nf (zero
// digits are not in it: sparse or small scalars).  Cutting the E references into n_lanes equal chunks keeps every lane of the launch
// busy -- with fixed chunks of L0 a list 8 % shorter leaves the last round of resident wavefronts 16 % empty and takes exactly as
// long.  msm_accumulate and the msm_combine* kernels derive the same value from the same inputs.
ZK_D uint32_t chunk_len(uint32_t E, uint32_t n_lanes, uint32_t L0) {
    const uint32_t need = (uint32_t)(((uint64_t)E + n_lanes - 1) / n_lanes);
    const uint32_t lo = L0 < 16u ? L0 : 16u;       // never below 16 (or L0): shorter chunks only multiply the chunk-edge partials
    return need < lo ? lo : need;
}

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

constexpr uint32_t COMBINE_SMALL = 32;     // buckets spanning <= this many chunks: summed by one lane
constexpr uint32_t COMBINE_MEDIUM = 2048;  // <= this many: one wavefront per bucket; above: one workgroup

ZK_D uint64_t partial_slot(uint32_t t, uint32_t ta, uint32_t s, uint32_t L) {
    return (t == ta && (s % L) != 0) ? 2ull * t + 1 : 2ull * t;
}

// wave reduction: lane 0 ends with the sum of all 64 lanes (order irrelevant: abelian group)
template <class F>
ZK_D XYZZu<F> wave_sum(XYZZu<F> acc) {
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
        XYZZu<F> o;
#pragma unroll
        for (int i = 0; i < F::NL; ++i) {
            o.x.v[i] = __shfl_down(acc.x.v[i], d, 64);
            o.y.v[i] = __shfl_down(acc.y.v[i], d, 64);
            o.zz.v[i] = __shfl_down(acc.zz.v[i], d, 64);
            o.zzz.v[i] = __shfl_down(acc.zzz.v[i], d, 64);
        }
        acc = XYZZu<F>::add(acc, o);
    }
    return acc;
}

// One lane per bucket: a bucket whose entries span p >= 2 chunks has exactly p partials at slots
// known from the offsets (see msm_accumulate).  Small p is summed here; larger p is queued.
// queues: q[0] = medium count, q[1] = large count, q[2 ..] medium ids (grow up), q[.. 2+nb) large ids (grow down)
// COMBINE_SG lanes cooperate on one small bucket: 4 shortens the dependent chain when the launch is
// latency-bound (1-2 jobs: 0.25 -> 0.19 ms); with more jobs the launch is throughput-bound and the idle
// lanes of the shuffle tree cost more than they save (7 jobs: 1.0 ms at 4 lanes), so 1 is used there.
template <class F, uint32_t COMBINE_SG>
__global__ void __launch_bounds__(128) msm_combine(RJobs jobs, uint32_t nb) {
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = id / COMBINE_SG, sub = id % COMBINE_SG;
    // every lane stays to the end: the sub-group sums below are wave shuffles
    XYZZu<F> acc = XYZZu<F>::infinity();
    bool mine = false;
    if (b < nb) {
        const uint32_t s = offsets[b], e = offsets[b + 1];
        if (e != s) {
            const uint32_t ta = s / L, tb = (e - 1) / L;
            const uint32_t p = tb - ta + 1;   // p == 1: whole bucket inside one chunk, already complete
            if (p > COMBINE_MEDIUM) {
                if (sub == 0) q[2 + nb - 1 - atomicAdd(&q[1], 1u)] = b;
            } else if (p > COMBINE_SMALL) {
                if (sub == 0) q[2 + atomicAdd(&q[0], 1u)] = b;
            } else if (p > 1) {
                mine = true;
#pragma unroll 1
                for (uint32_t t = ta + sub; t <= tb; t += COMBINE_SG)
                    acc = XYZZu<F>::add(acc, ld_xyzz<F>(part_pt, partial_slot(t, ta, s, L)));
            }
        }
    }
#pragma unroll 1
    for (int d = COMBINE_SG / 2; d >= 1; d >>= 1) {
        XYZZu<F> o;
#pragma unroll
        for (int i = 0; i < F::NL; ++i) {
            o.x.v[i] = __shfl_down(acc.x.v[i], d, 64);
            o.y.v[i] = __shfl_down(acc.y.v[i], d, 64);
            o.zz.v[i] = __shfl_down(acc.zz.v[i], d, 64);
            o.zzz.v[i] = __shfl_down(acc.zzz.v[i], d, 64);
        }
        acc = XYZZu<F>::add(acc, o);
    }
    if (mine && sub == 0) st_xyzz<F>(buckets, b, acc);
}

// medium buckets: one wavefront per bucket, lanes stride over its partials, shuffle tree
template <class F>
__global__ void __launch_bounds__(256) msm_combine_wave(RJobs jobs) {
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nm = q[0];
    for (uint32_t h = wave; h < nm; h += n_waves) {
        const uint32_t b = q[2 + h];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t ta = s / L, tb = (e - 1) / L;
        XYZZu<F> acc = XYZZu<F>::infinity();
        for (uint32_t t = ta + lane; t <= tb; t += 64) acc = XYZZu<F>::add(acc, ld_xyzz<F>(part_pt, partial_slot(t, ta, s, L)));
        acc = wave_sum<F>(acc);
        if (lane == 0) st_xyzz<F>(buckets, b, acc);
    }
}

// large buckets (heavily skewed scalars): one 256-lane workgroup per bucket
template <class F>
__global__ void __launch_bounds__(256) msm_combine_block(RJobs jobs, uint32_t nb) {
    extern __shared__ uint4 sh[];
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t u = threadIdx.x;
    const uint32_t nl = q[1];
    for (uint32_t h = blockIdx.x; h < nl; h += gridDim.x) {
        const uint32_t b = q[2 + nb - 1 - h];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t ta = s / L, tb = (e - 1) / L;
        XYZZu<F> acc = XYZZu<F>::infinity();
        for (uint32_t t = ta + u; t <= tb; t += 256) acc = XYZZu<F>::add(acc, ld_xyzz<F>(part_pt, partial_slot(t, ta, s, L)));
        acc = wave_sum<F>(acc);
        __syncthreads();
        if ((u & 63) == 0) st_xyzz<F>(sh, u >> 6, acc);
        __syncthreads();
        if (u == 0) {
            for (uint32_t w = 1; w < 4; ++w) acc = XYZZu<F>::add(acc, ld_xyzz<F>(sh, w));
            st_xyzz<F>(buckets, b, acc);
        }
    }
}

// level 1 of the per-window reduction: segment s of window w covers buckets [s*G, (s+1)*G)
//   run = sum B_i ; acc = sum (i+1) * B_i   (i local index)
template <class F>
__global__ void __launch_bounds__(128) msm_seg_reduce(RJobs jobs, MsmGeom g) {
    const void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* seg_run = jobs.seg_run[blockIdx.y];
    void* seg_acc = jobs.seg_acc[blockIdx.y];
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= g.W * g.ns) return;
    const uint32_t w = id / g.ns, s = id % g.ns;
    const uint32_t G = 1u << g.logG;
    XYZZu<F> run = XYZZu<F>::infinity(), acc = XYZZu<F>::infinity();
    for (int i = (int)G - 1; i >= 0; --i) {
        const uint32_t bi = w * g.B + s * G + (uint32_t)i;
        if (!offsets || offsets[bi + 1] != offsets[bi]) run = XYZZu<F>::add(run, ld_xyzz<F>(buckets, bi));
        acc = XYZZu<F>::add(acc, run);
    }
    st_xyzz<F>(seg_run, id, run);
    st_xyzz<F>(seg_acc, id, acc);
}

// level 2: one 256-lane workgroup per window.
//   S_w = sum_s acc_s + G * sum_s s * run_s
// The window sum leaves the device in the arkworks layout (XYZZ of 4 x SAT words, canonical).
template <class F>
// tot_out (optional): sum of all buckets of the window, same layout (used when one real window is
// reduced as several "virtual" windows to shorten the dependent-addition chain).
__global__ void __launch_bounds__(256) msm_win_finish(RJobs jobs, MsmGeom g) {
    extern __shared__ uint4 sh[];
    const void* seg_run = jobs.seg_run[blockIdx.y];
    const void* seg_acc = jobs.seg_acc[blockIdx.y];
    uint32_t* win_out = jobs.win_s[blockIdx.y];
    uint32_t* tot_out = jobs.win_t[blockIdx.y];
    const uint32_t w = blockIdx.x, u = threadIdx.x;
    const uint32_t q = 1u << g.logq;
    typedef XYZZu<F> P;
    P A = P::infinity(), V = P::infinity(), tsum = P::infinity(), R = P::infinity();
    for (int v = (int)q - 1; v >= 0; --v) {
        const uint32_t s = u * q + (uint32_t)v;
        P x = P::infinity();
        if (s < g.ns) {
            x = ld_xyzz<F>(seg_run, (uint64_t)w * g.ns + s);
            A = P::add(A, ld_xyzz<F>(seg_acc, (uint64_t)w * g.ns + s));
        }
        if (v >= 1) {
            tsum = P::add(tsum, x);
            V = P::add(V, tsum);
        } else {
            R = P::add(tsum, x);
        }
    }
    // Y = A + G * V
    for (uint32_t k = 0; k < g.logG; ++k) V = P::dbl(V);
    P Y = P::add(A, V);
    // suffix sums Q_u = sum_{u' >= u} R_u'  (Hillis-Steele in LDS)
    st_xyzz<F>(sh, u, R);
    for (uint32_t d = 1; d < 256; d <<= 1) {
        __syncthreads();
        P o = P::infinity();
        if (u + d < 256) o = ld_xyzz<F>(sh, u + d);
        __syncthreads();
        R = P::add(R, o);
        st_xyzz<F>(sh, u, R);
    }
    if (tot_out && u == 0) {
        uint32_t* o = tot_out + (size_t)w * 4 * F::SAT;
        if (R.is_inf()) {
            for (int i = 0; i < 4 * F::SAT; ++i) o[i] = 0;
        } else {
            R.x.to_sat(o);
            R.y.to_sat(o + F::SAT);
            R.zz.to_sat(o + 2 * F::SAT);
            R.zzz.to_sat(o + 3 * F::SAT);
        }
    }
    // Z = Y + (G*q) * Q_u   (u >= 1)
    P Z = Y;
    if (u >= 1) {
        P Qm = R;
        for (uint32_t k = 0; k < g.logG + g.logq; ++k) Qm = P::dbl(Qm);
        Z = P::add(Z, Qm);
    }
    __syncthreads();
    st_xyzz<F>(sh, u, Z);
    for (uint32_t d = 128; d >= 1; d >>= 1) {
        __syncthreads();
        if (u < d) {
            Z = P::add(Z, ld_xyzz<F>(sh, u + d));
            st_xyzz<F>(sh, u, Z);
        }
    }
    if (u == 0) {
        uint32_t* o = win_out + (size_t)w * 4 * F::SAT;
        if (Z.is_inf()) {
            for (int i = 0; i < 4 * F::SAT; ++i) o[i] = 0;
        } else {
            Z.x.to_sat(o);
            Z.y.to_sat(o + F::SAT);
            Z.zz.to_sat(o + 2 * F::SAT);
            Z.zzz.to_sat(o + 3 * F::SAT);
        }
    }
}

// ---- quad-cooperative forms of the three reduction kernels (ecq.cuh): four lanes per dependent chain,
// an addition in 4.5 product-times instead of 13.5.  Same inputs, outputs and arithmetic results.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_combine_q(RJobs jobs, uint32_t nb) {
    const void* part_pt = jobs.part_pt[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* buckets = jobs.buckets[blockIdx.y];
    uint32_t* q = jobs.q[blockIdx.y];
    const uint32_t L = chunk_len(offsets[jobs.nbk[blockIdx.y]], jobs.lanes[blockIdx.y], jobs.L[blockIdx.y]);
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = id >> 2, role = id & 3;
    uint32_t s = 0, ta = 0, np = 0;     // np = partials this quad sums itself (0: nothing to do)
    if (b < nb) {
        s = offsets[b];
        const uint32_t e = offsets[b + 1];
        if (e != s) {
            ta = s / L;
            const uint32_t p = (e - 1) / L - ta + 1;
            if (p > COMBINE_MEDIUM) {
                if (role == 0) q[2 + nb - 1 - atomicAdd(&q[1], 1u)] = b;
            } else if (p > COMBINE_SMALL) {
                if (role == 0) q[2 + atomicAdd(&q[0], 1u)] = b;
            } else if (p > 1) {
                np = p;
            }
        }
    }
    // the quads of a wavefront run in lock step to the longest bucket among them; shorter ones add infinity
    uint32_t steps = np;
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) {
        const uint32_t o = __shfl_xor(steps, d, 64);
        steps = o > steps ? o : steps;
    }
    F acc = F::zero();
#pragma unroll 1
    for (uint32_t k = 0; k < steps; ++k) {
        F v = F::zero();
        if (k < np) v = ld_coord<F>(part_pt, partial_slot(ta + k, ta, s, L), role);
        acc = qadd<F>(acc, v, role);
    }
    if (np) st_coord<F>(buckets, b, role, acc);
}

template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_seg_reduce_q(RJobs jobs, MsmGeom g) {
    const void* buckets = jobs.buckets[blockIdx.y];
    const uint32_t* offsets = jobs.offsets[blockIdx.y];
    void* seg_run = jobs.seg_run[blockIdx.y];
    void* seg_acc = jobs.seg_acc[blockIdx.y];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t id = tid >> 2, role = tid & 3;
    const bool live = id < g.W * g.ns;              // no early exit: wave shuffles inside qadd
    const uint32_t w = live ? id / g.ns : 0, sg = live ? id % g.ns : 0;
    const uint32_t G = 1u << g.logG;
    F run = F::zero(), acc = F::zero();
    // the bucket of the NEXT step (its two offsets, then its coordinate: two dependent round trips) is requested before the two
    // additions of the current one; at one to two wavefronts per SIMD nothing else would hide them
    auto bucket = [&](int i) -> F {
        const uint32_t bi = w * g.B + sg * G + (uint32_t)i;
        F v = F::zero();
        if (live && (!offsets || offsets[bi + 1] != offsets[bi])) v = ld_coord<F>(buckets, bi, role);
        return v;
    };
    F nv = bucket((int)G - 1);
#pragma unroll 1
    for (int i = (int)G - 1; i >= 0; --i) {
        const F v = nv;
        if (i > 0) nv = bucket(i - 1);
        run = qadd<F>(run, v, role);
        acc = qadd<F>(acc, run, role);
    }
    if (live) {
        st_coord<F>(seg_run, id, role, run);
        st_coord<F>(seg_acc, id, role, acc);
    }
}

// msm_win_finish for logq == 0 (one segment per chain): 4 * ns lanes per workgroup, ns <= 256
// raw != 0: the two sums stay on the device in the internal point form (the "buckets" of the next reduction level);
// raw == 0: arkworks layout for the host, as msm_win_finish.
// MAXT = lanes per workgroup the instance is built for.  512 (up to 128 chains: every launch of the window-table path) leaves the
// register file to two wavefronts per SIMD and nothing spills; built for 1024 lanes -- four wavefronts per SIMD, 128 registers --
// the same code spilled 162 registers to 332 bytes of scratch per lane (the form every launch used before round 4; only the
// per-window path's 256-chain geometry still needs it).
template <class F, int MAXT>
__global__ void __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(MAXT / 256, MAXT / 256))) msm_win_finish_q(RJobs jobs, MsmGeom g, uint32_t raw) {
    extern __shared__ uint4 sh[];
    const void* seg_run = jobs.seg_run[blockIdx.y];
    const void* seg_acc = jobs.seg_acc[blockIdx.y];
    uint32_t* win_out = jobs.win_s[blockIdx.y];
    uint32_t* tot_out = jobs.win_t[blockIdx.y];
    const uint32_t w = blockIdx.x, u = threadIdx.x >> 2, role = threadIdx.x & 3;
    const uint32_t T = blockDim.x >> 2;             // chains = power of two >= ns
    F R = F::zero(), Y = F::zero();
    if (u < g.ns) {
        R = ld_coord<F>(seg_run, (uint64_t)w * g.ns + u, role);
        Y = ld_coord<F>(seg_acc, (uint64_t)w * g.ns + u, role);
    }
    // suffix sums Q_u = sum_{u' >= u} R_u'  (Hillis-Steele in LDS)
    st_coord<F>(sh, u, role, R);
    for (uint32_t d = 1; d < T; d <<= 1) {
        __syncthreads();
        F o = F::zero();
        if (u + d < T) o = ld_coord<F>(sh, u + d, role);
        __syncthreads();
        R = qadd<F>(R, o, role);
        st_coord<F>(sh, u, role, R);
    }
    const bool r_inf = quad_is_inf(R, role);
    if (tot_out && u == 0) {
        if (raw) {
            st_coord<F>(tot_out, w, role, r_inf ? F::zero() : R);
        } else {
            uint32_t* o = tot_out + (size_t)w * 4 * F::SAT + role * F::SAT;
            if (r_inf) {
                for (int i = 0; i < F::SAT; ++i) o[i] = 0;
            } else {
                R.to_sat(o);
            }
        }
    }
    // Z = Y + G * Q_u   (u >= 1)
    F Qm = R;
    for (uint32_t k = 0; k < g.logG; ++k) Qm = qdbl<F>(Qm, role);
    F Z = qadd<F>(Y, u >= 1 ? Qm : F::zero(), role);
    __syncthreads();
    st_coord<F>(sh, u, role, Z);
    for (uint32_t d = T / 2; d >= 1; d >>= 1) {
        __syncthreads();
        if (u < d) {                                 // quad-uniform: all four lanes of a chain agree
            Z = qadd<F>(Z, ld_coord<F>(sh, u + d, role), role);
            st_coord<F>(sh, u, role, Z);
        }
    }
    const bool z_inf = quad_is_inf(Z, role);
    if (u == 0) {
        if (raw) {
            st_coord<F>(win_out, w, role, z_inf ? F::zero() : Z);
        } else {
            uint32_t* o = win_out + (size_t)w * 4 * F::SAT + role * F::SAT;
            if (z_inf) {
                for (int i = 0; i < F::SAT; ++i) o[i] = 0;
            } else {
                Z.to_sat(o);
            }
        }
    }
}

// Work-efficient level of the wide reduction: a node (run, acc) stands for m = 2^logm consecutive buckets,
//   run = sum B_i,  acc = sum (i + 1) B_i   (i local to the node);
// K = 2^logk neighbouring nodes make one node of K*m buckets:  run' = sum_j run_j,  acc' = sum_j acc_j + m * sum_j j * run_j
// -- 3 additions per child instead of the log2(chains) of the Hillis-Steele scan in msm_win_finish_q, which is what made
// 2^17 level-1 nodes per job (window tables with c = 20) cost more than the level below them.  One quad per output node.
// in: jobs.seg_run / seg_acc (n_out * K nodes);  out: jobs.win_s (run') / jobs.win_t (acc'), internal point form.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) msm_node_reduce_q(RJobs jobs, uint32_t n_out, uint32_t logk, uint32_t logm) {
    const void* in_run = jobs.seg_run[blockIdx.y];
    const void* in_acc = jobs.seg_acc[blockIdx.y];
    void* out_run = jobs.win_s[blockIdx.y];
    void* out_acc = jobs.win_t[blockIdx.y];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t id = tid >> 2, role = tid & 3;
    const bool live = id < n_out;                  // no early exit: wave shuffles inside qadd
    const uint32_t K = 1u << logk;
    const uint64_t first = (uint64_t)(live ? id : 0) * K;
    F run = F::zero(), wsum = F::zero(), asum = F::zero();
#pragma unroll 1
    for (int j = (int)K - 1; j >= 1; --j) {
        run = qadd<F>(run, live ? ld_coord<F>(in_run, first + (uint32_t)j, role) : F::zero(), role);
        wsum = qadd<F>(wsum, run, role);           // after the loop: sum_j j * run_j
        asum = qadd<F>(asum, live ? ld_coord<F>(in_acc, first + (uint32_t)j, role) : F::zero(), role);
    }
    run = qadd<F>(run, live ? ld_coord<F>(in_run, first, role) : F::zero(), role);
    asum = qadd<F>(asum, live ? ld_coord<F>(in_acc, first, role) : F::zero(), role);
    for (uint32_t t = 0; t < logm; ++t) wsum = qdbl<F>(wsum, role);
    asum = qadd<F>(asum, wsum, role);
    if (live) {
        st_coord<F>(out_run, id, role, run);
        st_coord<F>(out_acc, id, role, asum);
    }
}

// arkworks-layout affine (x||y Montgomery words) -> internal points.  Accepted encodings of the point at infinity:
// the flag, x = y = 0, and GroupAffine::zero() = (0, 1) (Montgomery one) -- what this library itself emits for an
// infinite result and what an arkworks caller holds; (0, 1) lies on neither supported curve (b = 4 / b = 3).
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

// ---------------------------------------------------------------------------------------- host side
int ensure_pinned(zk_ctx* c, size_t bytes);
template <class Fq>
XYZZ<Fq> jac_to_xyzz(const uint64_t* xyz);

// msm_win_finish_q with `chains` (a power of two <= 256) chains of four lanes per workgroup
template <class F>
int launch_win_finish_q(dim3 grid, uint32_t chains, hipStream_t st, const RJobs& jobs, const MsmGeom& g, uint32_t raw) {
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    const size_t shmem = (size_t)chains * PT;
    if (chains <= 128) {
        if (shmem > 48 * 1024)
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_win_finish_q<F, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((msm_win_finish_q<F, 512>), grid, dim3(4 * chains), shmem, st, jobs, g, raw);
    } else if (chains <= 256) {
        if (shmem > 48 * 1024)
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_win_finish_q<F, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((msm_win_finish_q<F, 1024>), grid, dim3(4 * chains), shmem, st, jobs, g, raw);
    } else {
        return ZK_ERR_UNSUPPORTED;
    }
    return ZK_OK;
}

// combine + segmented reduction of n_jobs MSMs that share the geometry (nb buckets, reduction geometry gr)
template <class F>
int queue_reduce(zk_ctx* c, const RJobs& jobs, uint32_t n_jobs, uint32_t nb, const MsmGeom& gr, hipStream_t st, bool queues_cleared = false,
                 uint32_t raw = 0 /* 1: the window sums stay in the internal point form (device buffers), quad geometry only */) {
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    ProfScope ps(c, "msm_reduce", st);
    const int T = 128;
    int rc;
    if (!queues_cleared)
        for (uint32_t k = 0; k < n_jobs; ++k) ZK_HIP_TRY(hipMemsetAsync(jobs.q[k], 0, 8, st));
    // quad-cooperative kernels where the geometry allows (one segment per chain, <= 256 chains per window)
    const bool quad = gr.logq == 0 && gr.ns <= 256;
    if (n_jobs <= 2) {
        if (quad) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + 255) / 256);
            hipLaunchKernelGGL(msm_combine_q<F>, dim3(blocks, n_jobs), dim3(256), 0, st, jobs, nb);
        } else {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 4>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        }
    } else {
        // lanes per small bucket when the launch has many jobs.  Option "combine_sg": tuning hook (profiles/r03_notes.md)
        const int sg = c->tune.combine_sg ? c->tune.combine_sg : 1;
        if (sg == 4) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 4>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        } else if (sg == 2) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 2 + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 2>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        } else {
            unsigned blocks = (unsigned)(((uint64_t)nb + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 1>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        }
    }
    hipLaunchKernelGGL(msm_combine_wave<F>, dim3(256, n_jobs), dim3(256), 0, st, jobs);
    hipLaunchKernelGGL(msm_combine_block<F>, dim3(64, n_jobs), dim3(256), 4 * PT, st, jobs, nb);
    if (quad) {
        unsigned sblocks = (unsigned)(((uint64_t)gr.W * gr.ns * 4 + 255) / 256);
        hipLaunchKernelGGL(msm_seg_reduce_q<F>, dim3(sblocks, n_jobs), dim3(256), 0, st, jobs, gr);
        uint32_t chains = 1;
        while (chains < gr.ns) chains <<= 1;
        size_t shmem = (size_t)chains * PT;
        if ((rc = launch_win_finish_q<F>(dim3(gr.W, n_jobs), chains, st, jobs, gr, raw))) return rc;
    } else {
        if (raw) return ZK_ERR_UNSUPPORTED;
        unsigned sblocks = (gr.W * gr.ns + T - 1) / T;
        hipLaunchKernelGGL(msm_seg_reduce<F>, dim3(sblocks, n_jobs), dim3(T), 0, st, jobs, gr);
        size_t shmem = 256 * PT;
        if (shmem > 48 * 1024)
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_win_finish<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(msm_win_finish<F>, dim3(gr.W, n_jobs), dim3(256), shmem, st, jobs, gr);
    }
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// Reduction for window tables with c > 16 (nb = 2^(c-1) >= 2^16 shared buckets), all on the device:
//   level 1  msm_seg_reduce      one LANE per node of 4 buckets: with >= 16 Ki nodes per job the launch is throughput-bound,
//                                where the single-lane group law costs 13.5 product-times per addition against the quad form's 18
//   level 2  msm_node_reduce_q   4 nodes -> one node of 16 buckets, 3 additions per child (quads)
//   level 3  msm_win_finish_q    VW = nb / 2048 virtual windows of 128 nodes -> S_v = sum_l (l+1) B_{v,l},  T_v = sum_l B_{v,l}, kept on
//                                the device in the internal point form
//   level 4  msm_seg_reduce_q + msm_win_finish_q over the two arrays S and T of every job (2 n_jobs "jobs", every element present):
//                                sum_v S_v,  K = sum_v (v+1) T_v,  sum_v T_v  -> pinned host memory, arkworks layout
//   host     total = sum_v S_v + 2048 * (K - sum_v T_v)      (bucket j = 2048 v + l has weight j + 1)
constexpr uint32_t WIDE_LOGG1 = 2;     // buckets per level-1 node
constexpr uint32_t WIDE_LOGK2 = 2;     // level-1 nodes per level-2 node
constexpr uint32_t WIDE_CHAINS = 128;  // level-2 nodes per virtual window
constexpr uint32_t WIDE_VB = WIDE_CHAINS << (WIDE_LOGG1 + WIDE_LOGK2);   // buckets per virtual window (2048)
template <class F>
int queue_reduce_wide(zk_ctx* c, const RJobs& jobs, uint32_t n_jobs, uint32_t nb, void* const* d_vw, void* const* d_seg3, void* const* d_seg2,
                      char* h_out, size_t h_stride, hipStream_t st) {
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    ProfScope ps(c, "msm_reduce", st);
    int rc;
    const uint32_t VW = nb / WIDE_VB;
    if (VW == 0 || VW > 2048) return ZK_ERR_UNSUPPORTED;
    {   // chunk-edge partials -> buckets (queues cleared by the job's sort)
        const int T = 128;
        if (n_jobs <= 2) {
            unsigned blocks = (unsigned)(((uint64_t)nb * 4 + 255) / 256);
            hipLaunchKernelGGL(msm_combine_q<F>, dim3(blocks, n_jobs), dim3(256), 0, st, jobs, nb);
        } else {
            unsigned blocks = (unsigned)(((uint64_t)nb + T - 1) / T);
            hipLaunchKernelGGL((msm_combine<F, 1>), dim3(blocks, n_jobs), dim3(T), 0, st, jobs, nb);
        }
        hipLaunchKernelGGL(msm_combine_wave<F>, dim3(256, n_jobs), dim3(256), 0, st, jobs);
        hipLaunchKernelGGL(msm_combine_block<F>, dim3(64, n_jobs), dim3(256), 4 * PT, st, jobs, nb);
    }
    const uint32_t n1 = nb >> WIDE_LOGG1, n2 = n1 >> WIDE_LOGK2;
    {   // level 1: flat over all buckets (one "window" of nb buckets, nodes of 4)
        MsmGeom g1;
        memset(&g1, 0, sizeof g1);
        g1.W = 1;
        g1.B = nb;
        g1.nb = nb;
        g1.logG = WIDE_LOGG1;
        g1.ns = n1;
        const int T = 128;
        unsigned sblocks = (unsigned)(((uint64_t)n1 + T - 1) / T);
        hipLaunchKernelGGL(msm_seg_reduce<F>, dim3(sblocks, n_jobs), dim3(T), 0, st, jobs, g1);
    }
    RJobs j2 = jobs;      // level 2: seg_run / seg_acc (n1 nodes) -> d_seg2 (n2 nodes: run | acc)
    for (uint32_t k = 0; k < n_jobs; ++k) {
        j2.win_s[k] = (uint32_t*)d_seg2[k];
        j2.win_t[k] = (uint32_t*)((char*)d_seg2[k] + (size_t)n2 * PT);
    }
    {
        unsigned blocks = (unsigned)(((uint64_t)n2 * 4 + 255) / 256);
        hipLaunchKernelGGL(msm_node_reduce_q<F>, dim3(blocks, n_jobs), dim3(256), 0, st, j2, n2, WIDE_LOGK2, WIDE_LOGG1);
    }
    RJobs j3 = jobs;      // level 3: virtual windows of 128 level-2 nodes (16 buckets each)
    MsmGeom gv;
    memset(&gv, 0, sizeof gv);
    gv.W = VW;
    gv.B = WIDE_VB;
    gv.nb = nb;
    gv.logG = WIDE_LOGG1 + WIDE_LOGK2;
    gv.ns = WIDE_CHAINS;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        j3.seg_run[k] = d_seg2[k];
        j3.seg_acc[k] = (char*)d_seg2[k] + (size_t)n2 * PT;
        j3.win_s[k] = (uint32_t*)d_vw[k];
        j3.win_t[k] = (uint32_t*)((char*)d_vw[k] + (size_t)VW * PT);
    }
    {
        if ((rc = launch_win_finish_q<F>(dim3(VW, n_jobs), WIDE_CHAINS, st, j3, gv, 1u))) return rc;
    }
    MsmGeom g4;           // level 4: the VW pairs (S_v, T_v) of every job
    memset(&g4, 0, sizeof g4);
    g4.W = 1;
    g4.B = VW;
    g4.nb = VW;
    g4.logG = VW <= 4 ? 0 : VW <= 1024 ? 2 : 3;
    g4.ns = VW >> g4.logG;
    if (g4.ns == 0 || g4.ns > 256) return ZK_ERR_UNSUPPORTED;
    RJobs j4;
    memset(&j4, 0, sizeof j4);
    const size_t PHB = h_stride / 4;      // bytes of one host point
    for (uint32_t k = 0; k < n_jobs; ++k)
        for (uint32_t a = 0; a < 2; ++a) {
            const uint32_t j = 2 * k + a;
            j4.buckets[j] = (char*)d_vw[k] + (size_t)a * VW * PT;
            j4.offsets[j] = nullptr;
            j4.seg_run[j] = (char*)d_seg3[k] + (size_t)a * 2 * g4.ns * PT;
            j4.seg_acc[j] = (char*)d_seg3[k] + ((size_t)a * 2 + 1) * g4.ns * PT;
            j4.win_s[j] = (uint32_t*)(h_out + (size_t)k * h_stride + (size_t)a * 2 * PHB);
            j4.win_t[j] = (uint32_t*)(h_out + (size_t)k * h_stride + ((size_t)a * 2 + 1) * PHB);
        }
    {
        unsigned sblocks = (unsigned)(((uint64_t)g4.ns * 4 + 255) / 256);
        hipLaunchKernelGGL(msm_seg_reduce_q<F>, dim3(sblocks, 2 * n_jobs), dim3(256), 0, st, j4, g4);
        uint32_t chains = 1;
        while (chains < g4.ns) chains <<= 1;
        if ((rc = launch_win_finish_q<F>(dim3(1, 2 * n_jobs), chains, st, j4, g4, 0u))) return rc;
    }
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

uint32_t ilog2_floor(uint64_t x) {
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
        // per-window path: measured on MI355X over c = 9 .. 16 at every size (profiles/r02_notes.md, "window of the per-window
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

constexpr uint32_t CHUNK_L = 32;

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
    uint32_t* hist = (uint32_t*)mb.counts.p;
    uint32_t* bsum = hist + (size_t)g.W * S * g.B;
    uint32_t* offsets = (uint32_t*)mb.offsets.p;
    int16_t* dig = (int16_t*)mb.tmp.p;
    uint32_t* entries = (uint32_t*)mb.entries.p;
    void* seg_run = mb.seg.p;
    void* seg_acc = (char*)mb.seg.p + (size_t)g.W * g.ns * PT;
    hipStream_t st = c->stream;

    {
        ProfScope ps(c, "msm_sort");
        const int T = 256;
        unsigned blocks = (unsigned)((n + T - 1) / T);
        hipLaunchKernelGGL(msm_digits, dim3(blocks), dim3(T), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, g, dig);
        size_t lds = (size_t)g.B * 4;
        if (lds > 48 * 1024) {
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ZK_HIP_TRY(hipFuncSetAttribute((const void*)msm_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL(msm_hist, dim3(S, g.W), dim3(1024), lds, st, dig, (uint64_t)n, g, S, hist);
        const unsigned nblk = (g.nb + 1023) / 1024;
        hipLaunchKernelGGL(msm_scan1, dim3(nblk), dim3(1024), 0, st, hist, g, S, bsum);
        hipLaunchKernelGGL(msm_scan2, dim3(1), dim3(1024), 0, st, bsum, nblk);
        hipLaunchKernelGGL(msm_scan3, dim3(nblk), dim3(1024), 0, st, hist, g, S, bsum, offsets);
        hipLaunchKernelGGL(msm_scatter, dim3(S, g.W), dim3(1024), lds, st, dig, (uint64_t)n, g, S, hist, entries, 0u);
        ZK_HIP_TRY(hipGetLastError());
    }
    {
        ProfScope ps(c, "msm_accumulate");
        const int T = 128;
        unsigned blocks = (n_lanes + T - 1) / T;
        hipLaunchKernelGGL((msm_accumulate<F, false>), dim3(blocks), dim3(T), 0, st, entries, offsets, g.nb, d_bases, mb.buckets.p,
                           mb.part_pt.p, CHUNK_L, n_lanes, (uint64_t)0, (uint64_t)0);
        ZK_HIP_TRY(hipGetLastError());
    }
    {
        RJobs jobs;
        memset(&jobs, 0, sizeof jobs);
        jobs.part_pt[0] = mb.part_pt.p;
        jobs.offsets[0] = offsets;
        jobs.buckets[0] = mb.buckets.p;
        jobs.q[0] = (uint32_t*)mb.part_key.p;
        jobs.seg_run[0] = seg_run;
        jobs.seg_acc[0] = seg_acc;
        jobs.win_s[0] = (uint32_t*)mb.win.p;
        jobs.win_t[0] = nullptr;
        jobs.L[0] = CHUNK_L;
        jobs.lanes[0] = n_lanes;
        jobs.nbk[0] = g.nb;
        if ((rc = queue_reduce<F>(c, jobs, 1, g.nb, g, st))) return rc;
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

constexpr uint32_t PRE_C = 16;        // default window of the precomputed table
constexpr uint32_t PRE_C_MAX = 21;    // 2^20 shared buckets: 4096 per partition in the second sort pass (144 KiB of LDS)
constexpr uint32_t PRE_CHUNK_L = 128; // references per lane on the shared-bucket path (buckets hold ~W*n/2^15 each)
constexpr uint32_t PRE_Q_OFF = 1024;  // words of part_key in front of the combine queues (partition starts, totals, counter)
constexpr uint32_t PRE_VW = 64;       // virtual windows for the final bucket reduction: 512 buckets = 64 chains x 4 lanes per workgroup
                                      // (256 registers per lane; with 32 windows the 1024-lane workgroup spilled at 128)

template <class Cv>
int msm_precompute_run(zk_ctx* c, zk_srs* s, uint32_t window_bits, uint32_t w0, uint32_t wstep) {
    typedef typename Cv::FqU F;
    // default window: 16 bits (16 rows, 2^15 buckets) below 2^19 points; from there on 17 bits, which scalars folded to
    // k <= (r - 1) / 2 (MsmGeom::neg) cover in 15 windows -- one mixed addition per scalar fewer for twice the buckets to reduce
    // (measured, profiles/r03_notes.md: 2^18 27.1 / 29.2 ms per proof at c = 16 / 17, 2^19 48.1 / 47.3, 2^20 87.4 / 85.6, 2^22 364.6 / 353.2)
    // From 2^22 points on a whole table takes 20 bits (13 rows, 2^19 buckets): the 2 x 2^19 additions of the wide bucket reduction are
    // then fewer than the two additions per scalar they save (measured, profiles/r05_sweep_window.txt: c = 17 / 20 at 2^21 153.9 / 153.3 ms
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

// ---- MSM over a precomputed SRS -------------------------------------------------------------------
// Every (scalar, window) digit is a reference to table[w][i]; all windows share one set of 2^(c-1)
// buckets.  The work of one MSM is queued in two pieces so that a batch can pipeline them:
//   sort   (digits, LDS counting sort)        -- light kernels, few VGPRs: they co-reside with the
//                                                 accumulate waves of the PREVIOUS MSM (aux stream)
//   heavy  (accumulate, combine, reduction, read-back of the virtual-window sums)   (main stream)
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
    if (!pl.wide_red) {                                   // tuning options "pre_vw" / "pre_logg" (profiles/r02_notes.md)
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
    if (c->tune.chunk_l >= 8 && c->tune.chunk_l <= 1024) {          // tuning option "chunk_l" (profiles/r02_notes.md)
        pl.chunk_l = (uint32_t)c->tune.chunk_l;
        tuned = true;
    }
    pl.n_lanes = (uint32_t)((pl.nf + pl.chunk_l - 1) / pl.chunk_l);
    // whole rounds of resident lanes: every lane does the same work, so 1.9 rounds take as long as 2 (15 windows of 2^20 digits
    // at 64 per lane are 245760 lanes): round the lane count up to a multiple of a round and shorten the chunks instead.
    // Two rounds become three where the chunks stay >= 32 references: the end of the launch, where CUs wait for their last
    // wavefronts, shortens with the chunk (2^20, c = 17: 60 -> 40 per lane, msm_accumulate -2.5 % per launch, msm_combine* +7
    // partials per bucket instead of 5, net +0.5 .. 0.8 % proofs/s; 30 and 24 per lane give the accumulation another 1 % and
    // the combine more than that back: profiles/r03_notes.md).  A job that is not the last one of a merged launch has no end of
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

// The sort of a table-path job comes in two pieces.  `pre_queue_digits` is the only kernel that reads the caller's scalars
// (digits + the slab counts of the 256 partitions; into_repr of a commit's Montgomery coefficients fused in): it is queued when
// the job is submitted, so the input vector is consumed in stream order at the call, as before.  `pre_queue_sort_rest` -- scan,
// partition scatter, final placement -- takes the jobs of a round as ONE launch per kernel (blockIdx.y = job).
inline bool pre_psort16(const PrePlan& pl) { return !pl.wide && pl.g1.nb % (1u << PS_LOB) == 0 && (pl.g1.nb >> PS_LOB) <= 256; }

template <class Cv>
int pre_queue_digits(zk_ctx* c, const PrePlan& pl, MsmBufs& mb, const void* d_scalars, size_t n, hipStream_t st, bool mont) {
    ProfScope ps(c, "msm_sort", st);
    typedef typename Cv::Fr FrS;
    const int T = 256;
    const uint32_t sp = psort_slab_len(n);
    int rc;
    if (pl.wide) {
        const uint32_t lob = pl.g.c - 9, P = 256;
        uint32_t* part_start = (uint32_t*)mb.part_key.p;
        uint32_t* part_total = part_start + P + 1;
        uint32_t* scan_counter = part_total + P;
        uint32_t* combine_q = (uint32_t*)mb.part_key.p + PRE_Q_OFF;
        uint32_t* hist = (uint32_t*)mb.counts.p;
        int32_t* dig32 = (int32_t*)mb.entries.p;          // the digits wait in the buffer of the sorted references (pre_sizes)
        if (mont) hipLaunchKernelGGL((psortw_digits_hist<FrS, true>), dim3(PS_SLABS), dim3(256), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, sp, pl.g,
                                     lob, dig32, hist, scan_counter, combine_q);
        else hipLaunchKernelGGL((psortw_digits_hist<FrS, false>), dim3(PS_SLABS), dim3(256), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, sp, pl.g,
                                lob, dig32, hist, scan_counter, combine_q);
        ZK_HIP_TRY(hipGetLastError());
        return ZK_OK;
    }
    if (!pre_psort16(pl)) return ZK_ERR_UNSUPPORTED;     // the table windows are 16 .. 21 bits: 2^15 buckets = 256 partitions of 128
    int16_t* dig = (int16_t*)mb.entries.p;
    const bool pairs = (n & 1) == 0 && pl.g.c == 16 && pl.g.W == 16 && pl.g.Wt == 16 && !pl.g.neg;        // two scalars per lane
    const uint32_t P = pl.g1.nb >> PS_LOB;
    uint32_t* part_start = (uint32_t*)mb.part_key.p;    // P + 1 partition starts | P totals | scan counter
    uint32_t* part_total = part_start + P + 1;
    uint32_t* scan_counter = part_total + P;
    uint32_t* combine_q = (uint32_t*)mb.part_key.p + PRE_Q_OFF;
    uint32_t* hist = (uint32_t*)mb.counts.p;
    if (pairs && P == 256) {
        // digits and the per-slab partition counts in one kernel
        if (mont) hipLaunchKernelGGL((psort_digits_hist<FrS, true>), dim3(PS_SLABS), dim3(256), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, sp,
                                     dig, hist, scan_counter, combine_q);
        else hipLaunchKernelGGL((psort_digits_hist<FrS, false>), dim3(PS_SLABS), dim3(256), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, sp,
                                dig, hist, scan_counter, combine_q);
    } else {
        if (pairs) {
            unsigned b2 = (unsigned)((n / 2 + T - 1) / T);
            if (mont) hipLaunchKernelGGL((msm_digits2<FrS, true>), dim3(b2), dim3(T), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, pl.g, dig);
            else hipLaunchKernelGGL((msm_digits2<FrS, false>), dim3(b2), dim3(T), 0, st, (const uint32_t*)d_scalars, (uint64_t)n, pl.g, dig);
        } else {
            const void* canon = d_scalars;
            if (mont) {   // odd length: separate into_repr pass, then the one-scalar-per-lane kernel
                if ((rc = mb.scalars.ensure(n * 32))) return rc;
                if ((rc = fr_convert_stream(c, Cv::ID, d_scalars, n, mb.scalars.p, st))) return rc;
                canon = mb.scalars.p;
            }
            unsigned blocks = (unsigned)((n + T - 1) / T);
            hipLaunchKernelGGL(msm_digits, dim3(blocks), dim3(T), 0, st, (const uint32_t*)canon, (uint64_t)n, pl.g, dig);
        }
        hipLaunchKernelGGL(psort_hist, dim3(PS_SLABS), dim3(PS_T), P * 4, st, dig, (uint64_t)n, pl.g.W, sp, P, hist, scan_counter, combine_q);
    }
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// jobs of one SRS (same window geometry); lens[k] scalars in job k.  Jobs whose plan shares the ctx's staging area (shared_stage:
// 2^24 scalars and more at c = 20) are placed one after the other -- three launches per job, each tens of milliseconds long.
template <class Cv>
int pre_queue_sort_rest(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, const size_t* lens, uint32_t n_jobs, hipStream_t st) {
    if (n_jobs == 0) return ZK_OK;
    if (n_jobs > (uint32_t)MAX_JOBS) return ZK_ERR_UNSUPPORTED;
    uint32_t n_shared = 0;
    for (uint32_t k = 0; k < n_jobs; ++k) n_shared += pls[k].shared_stage ? 1u : 0u;
    if (n_shared && n_jobs > 1) {
        for (uint32_t k = 0; k < n_jobs; ++k) {
            int rc = pre_queue_sort_rest<Cv>(c, pls + k, mbs + k, lens + k, 1, st);
            if (rc) return rc;
        }
        return ZK_OK;
    }
    ProfScope ps(c, "msm_sort", st);
    const PrePlan& p0 = pls[0];
    const uint32_t P = p0.wide ? 256u : p0.g1.nb >> PS_LOB;
    SJobs sj;
    memset(&sj, 0, sizeof sj);
    for (uint32_t k = 0; k < n_jobs; ++k) {
        MsmBufs& mb = *mbs[k];
        SJob& J = sj.j[k];
        void* stage = pls[k].shared_stage ? c->stage_shared.p : mb.stage.p;
        J.dig = mb.entries.p;           // overwritten by the placement kernel once the scatter has read them
        J.n = lens[k];
        J.sp = psort_slab_len(lens[k]);
        J.hist = (uint32_t*)mb.counts.p;
        J.part_start = (uint32_t*)mb.part_key.p;
        J.part_total = J.part_start + P + 1;
        J.counter = J.part_total + P;
        J.stage_ref = (uint32_t*)stage;
        J.stage_lo = (char*)stage + (size_t)pls[k].nf * 4;
        J.entries = (uint32_t*)mb.entries.p;
        J.offsets = (uint32_t*)mb.offsets.p;
    }
    hipLaunchKernelGGL(psort_scan, dim3(P, n_jobs), dim3(PS_SLABS), 0, st, sj, P);
    if (p0.wide) {
        const uint32_t lob = p0.g.c - 9;
        hipLaunchKernelGGL(psortw_scatter, dim3(PS_SLABS, n_jobs), dim3(PS_T), 0, st, sj, p0.g.W, lob);
        const uint32_t NB = 1u << lob;
        const size_t lds = ((size_t)3 * NB + 1 + 16 + PS_TILE) * 4 + (size_t)PS_TILE * 2;
        ZK_HIP_TRY(hipFuncSetAttribute((const void*)psortw_final, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(psortw_final, dim3(P, n_jobs), dim3(PS_T), lds, st, sj, P, lob);
    } else {
        hipLaunchKernelGGL(psort_scatter, dim3(PS_SLABS, n_jobs), dim3(PS_T), 0, st, sj, p0.g.W, P);
        hipLaunchKernelGGL(psort_final, dim3(P, n_jobs), dim3(PS_T), 0, st, sj, P);
    }
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

// fused reduction of the jobs mbs[0..n_jobs) (same geometry) + read-back of their virtual-window sums
// into h_win (n_jobs x win_bytes)
// d_partials (optional, n_jobs pointers): instead of the virtual-window sums going to the host, every job's whole sum
//   sum_v S_v + B_v sum_v v T_v  is formed on the device (one more msm_win_finish_q launch over the VW pairs of each job) and left
// at d_partials[k] as ONE point in the internal XYZZ form (zk_partial_dev_bytes): the multi-GPU exchange reads it from there.
// the device forms of a round's result need the quad-cooperative reduction of at most 128 power-of-two virtual windows
inline bool pre_partial_dev_ok(const PrePlan& p) {
    return !p.wide_red && p.gv.logq == 0 && p.gv.ns <= 256 && p.gv.W <= 128 && (p.gv.W & (p.gv.W - 1)) == 0;
}

template <class Cv>
int pre_queue_reduce(zk_ctx* c, const PrePlan* pls, MsmBufs* const* mbs, uint32_t n_jobs, void* h_win, hipStream_t st, void* const* d_partials = nullptr,
                     int partial_kind = 1) {
    typedef typename Cv::FqU F;
    constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
    RJobs jobs;
    memset(&jobs, 0, sizeof jobs);
    const PrePlan& p0 = pls[0];
    if (d_partials && !pre_partial_dev_ok(p0)) return ZK_ERR_UNSUPPORTED;     // checked by the callers before anything is queued
    for (uint32_t k = 0; k < n_jobs; ++k) {
        MsmBufs& mb = *mbs[k];
        jobs.part_pt[k] = mb.part_pt.p;
        jobs.offsets[k] = (const uint32_t*)mb.offsets.p;
        jobs.buckets[k] = mb.buckets.p;
        jobs.q[k] = (uint32_t*)mb.part_key.p + PRE_Q_OFF;
        jobs.seg_run[k] = mb.seg.p;
        jobs.seg_acc[k] = (char*)mb.seg.p + (p0.wide_red ? (size_t)(p0.g.B >> WIDE_LOGG1) : (size_t)p0.gv.W * p0.gv.ns) * PT;
        // the window sums (a few KiB per job) are written by the last kernel straight into the pinned host
        // buffer (hipHostMalloc memory is device-visible): no copy launches at the tail of the call
        if (d_partials && partial_kind == 2) {
            jobs.win_s[k] = (uint32_t*)d_partials[k];                                       // S_v | T_v, internal form, straight into the
            jobs.win_t[k] = (uint32_t*)((char*)d_partials[k] + (size_t)p0.gv.W * PT);       // caller's buffer (the collective's send buffer)
        } else if (d_partials) {
            jobs.win_s[k] = (uint32_t*)mb.win.p;                                            // S_v | T_v, internal form, on the device
            jobs.win_t[k] = (uint32_t*)((char*)mb.win.p + (size_t)p0.gv.W * PT);
        } else {
            jobs.win_s[k] = (uint32_t*)((char*)h_win + (size_t)k * p0.win_bytes);
            jobs.win_t[k] = jobs.win_s[k] + (size_t)p0.gv.W * 4 * F::SAT;
        }
        jobs.L[k] = mb.acc_chunk_l;            // the plan the accumulation really ran with (pre_queue_accumulate), not a re-derived one
        jobs.lanes[k] = mb.acc_n_lanes;
        jobs.nbk[k] = p0.g1.nb;
    }
    // the queue counters were cleared by the job's sort (psort_hist / the memset of the fallback sort)
    if (p0.wide_red) {
        void* d_vw[MAX_JOBS];
        void* d_seg3[MAX_JOBS];
        void* d_seg2[MAX_JOBS];
        for (uint32_t k = 0; k < n_jobs; ++k) {
            d_vw[k] = mbs[k]->win.p;
            d_seg3[k] = mbs[k]->seg3.p;
            d_seg2[k] = mbs[k]->seg2.p;
        }
        return queue_reduce_wide<F>(c, jobs, n_jobs, p0.g1.nb, d_vw, d_seg3, d_seg2, (char*)h_win, p0.win_bytes, st);
    }
    if (!d_partials) return queue_reduce<F>(c, jobs, n_jobs, p0.g1.nb, p0.gv, st, true);
    int rc = queue_reduce<F>(c, jobs, n_jobs, p0.g1.nb, p0.gv, st, true, 1u);
    if (rc) return rc;
    if (partial_kind == 2) return ZK_OK;      // the window sums ARE the result: zk_g1_sum_winsums_dev adds the ranks' and the host combines
    // the VW pairs of a job as one "window" of VW chains: R_u = T_u, Y_u = S_u, Z = sum_u S_u + B_v * sum_u u T_u
    const uint32_t chains = p0.gv.W;
    MsmGeom gf;
    memset(&gf, 0, sizeof gf);
    gf.W = 1;
    gf.ns = p0.gv.W;
    gf.logG = ilog2_floor(p0.gv.B);
    RJobs jf;
    memset(&jf, 0, sizeof jf);
    for (uint32_t k = 0; k < n_jobs; ++k) {
        jf.seg_acc[k] = jobs.win_s[k];
        jf.seg_run[k] = jobs.win_t[k];
        jf.win_s[k] = (uint32_t*)d_partials[k];
        jf.win_t[k] = nullptr;
    }
    ProfScope ps(c, "msm_reduce", st);
    if ((rc = launch_win_finish_q<F>(dim3(1, n_jobs), chains, st, jf, gf, 1u))) return rc;
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

// sum over the ranks of every job's device partial (ranks x n_jobs points as the all-gather leaves them) -> n_jobs points,
// arkworks layout, straight into pinned host memory: one quad per job, ranks - 1 dependent additions
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) g1_sum_partials_q(const void* parts, uint32_t ranks, uint32_t n_jobs, uint32_t* out_sat) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = tid >> 2, role = tid & 3;
    const bool live = k < n_jobs;                  // no early exit: wave shuffles inside qadd
    F acc = F::zero();
#pragma unroll 1
    for (uint32_t r = 0; r < ranks; ++r) acc = qadd<F>(acc, live ? ld_coord<F>(parts, (uint64_t)r * n_jobs + k, role) : F::zero(), role);
    const bool inf = quad_is_inf(acc, role);
    if (live) {
        uint32_t* o = out_sat + (size_t)k * 4 * F::SAT + role * F::SAT;
        if (inf) {
            for (int i = 0; i < F::SAT; ++i) o[i] = 0;
        } else {
            acc.to_sat(o);
        }
    }
}

template <class Cv>
int sum_partials_dev(zk_ctx* c, const void* d_parts, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    typedef typename Cv::Fq Fq;
    typedef typename Cv::FqU F;
    typedef XYZZ<Fq> PH;
    constexpr int L64 = Fq::N / 2;
    if (n_jobs == 0) return ZK_OK;
    if (ranks == 0 || ranks > 4096 || n_jobs > 4096) return ZK_ERR_BAD_ARG;
    int rc = ensure_pinned(c, (size_t)n_jobs * sizeof(PH) > (size_t)MAX_JOBS * 4096 ? (size_t)n_jobs * sizeof(PH) : (size_t)MAX_JOBS * 4096);
    if (rc) return rc;
    const unsigned blocks = (n_jobs * 4 + 255) / 256;
    hipLaunchKernelGGL(g1_sum_partials_q<F>, dim3(blocks), dim3(256), 0, c->stream, d_parts, (uint32_t)ranks, n_jobs, (uint32_t*)c->pinned);
    ZK_HIP_TRY(hipGetLastError());
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    const PH* h = (const PH*)c->pinned;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        Affine<Fq> a;
        uint64_t* xy = out_xy + (size_t)k * 2 * L64;
        if (!h[k].to_affine(a)) {
            const Fq one = Fq::one();
            memset(xy, 0, sizeof(uint64_t) * L64);
            memcpy(xy + L64, one.v, sizeof(uint64_t) * L64);        // GroupAffine::zero() = (0, 1, infinity)
            if (out_inf) out_inf[k] = 1;
        } else {
            memcpy(xy, a.x.v, sizeof(uint64_t) * L64);
            memcpy(xy + L64, a.y.v, sizeof(uint64_t) * L64);
            if (out_inf) out_inf[k] = 0;
        }
    }
    return ZK_OK;
}

// element-wise sum over the ranks of every job's 2 VW virtual-window sums (ranks x n_jobs x 2 VW points as the all-gather leaves
// them) -> n_jobs x 2 VW points in the arkworks layout in pinned host memory, where the single-GPU path's last reduction kernel
// puts them.  Q = 2^logq quads share one sum: quad j adds the ranks j, j + Q, ... (ranks / Q - 1 dependent additions), an LDS tree
// adds the Q partial sums (logq more): 3 dependent additions for 8 ranks instead of 7 -- the launch is latency-shaped (2 VW n_jobs
// points, at most a few thousand quads), so the chain is what it costs.
template <class F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) g1_sum_winsums_q(const void* all, uint32_t ranks, uint32_t n_pts /* n_jobs * 2 VW */,
                                                                                                     uint32_t logq, uint32_t* out_sat) {
    extern __shared__ uint4 sh[];
    const uint32_t Q = 1u << logq;
    const uint32_t quad = threadIdx.x >> 2, role = threadIdx.x & 3;
    const uint32_t per_block = (blockDim.x >> 2) >> logq;                      // points per workgroup
    const uint32_t k = blockIdx.x * per_block + (quad >> logq), j = quad & (Q - 1);
    const bool live = k < n_pts;                   // no early exit: wave shuffles inside qadd, barriers below
    F acc = F::zero();
#pragma unroll 1
    for (uint32_t r = j; r < ranks; r += Q) acc = qadd<F>(acc, live ? ld_coord<F>(all, (uint64_t)r * n_pts + k, role) : F::zero(), role);
    for (uint32_t d = Q >> 1; d >= 1; d >>= 1) {
        st_coord<F>(sh, quad, role, acc);
        __syncthreads();
        const F o = j < d ? ld_coord<F>(sh, quad + d, role) : F::zero();
        __syncthreads();
        acc = qadd<F>(acc, o, role);               // quads with j >= d add the point at infinity: uniform control flow
    }
    const bool inf = quad_is_inf(acc, role);
    if (live && j == 0) {
        uint32_t* o = out_sat + (size_t)k * 4 * F::SAT + role * F::SAT;
        if (inf) {
            for (int i = 0; i < F::SAT; ++i) o[i] = 0;
        } else {
            acc.to_sat(o);
        }
    }
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
    if ((rc = pre_queue_digits<Cv>(c, pl, mb, d_scalars, n, c->stream, false))) return rc;
    if ((rc = pre_queue_sort_rest<Cv>(c, &pl, &one, &n, 1, c->stream))) return rc;
    if ((rc = pre_queue_accumulate<Cv>(c, &pl, &one, &n, &base_offset, 1, s, c->stream))) return rc;
    if ((rc = pre_queue_reduce<Cv>(c, &pl, &one, 1, c->pinned, c->stream))) return rc;
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
// neighbouring jobs on a second stream was measured to cost more than it hides (profiles/r01_notes.md, r02_notes.md).
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
        if ((rc = pre_queue_digits<Cv>(c, pl, mb, d_coeffs[k], lens[k], st, mont))) return rc;
        mb.stage_of_job = 1;
        if (n_begun) *n_begun = k + 1;
        if (defer) continue;
        MsmBufs* one = &mb;
        if ((rc = pre_queue_sort_rest<Cv>(c, &pl, &one, &lens[k], 1, st))) return rc;
        if ((rc = pre_queue_accumulate<Cv>(c, &pl, &one, &lens[k], nullptr, 1, s, st))) return rc;
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
int msm_batch_pre_reduce(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_partials = nullptr,
                         int partial_kind = 1) {
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
        if (d_partials && !pre_partial_dev_ok(pl[k])) return ZK_ERR_UNSUPPORTED;
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
        if ((rc = pre_queue_sort_rest<Cv>(c, dpl, dmb, dlen, nd, st))) return rc;
        if ((rc = pre_queue_accumulate<Cv>(c, dpl, dmb, dlen, nullptr, nd, s, st))) return rc;
        for (uint32_t k = 0; k < nd; ++k) dmb[k]->stage_of_job = 2;
    }
    if ((rc = pre_queue_reduce<Cv>(c, pl, mbs, n_jobs, c->pinned, st, d_partials, partial_kind))) return rc;
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
    typedef typename Cv::FqU F;
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
    {
        ProfScope ps(c, "msm_sum_winsums", c->stream);
        constexpr size_t PT = (size_t)4 * Store<F>::WORDS * 4;
        uint32_t logq = 0;                                    // quads per sum: half the ranks (rounded up to a power of two), at most 16
        while (logq < 4 && (2u << logq) < ranks) ++logq;
        const uint32_t per_block = 64u >> logq;
        hipLaunchKernelGGL(g1_sum_winsums_q<F>, dim3((n_pts + per_block - 1) / per_block), dim3(256), 64 * PT, c->stream, d_all, (uint32_t)ranks, n_pts, logq,
                           (uint32_t*)c->pinned);
        ZK_HIP_TRY(hipGetLastError());
    }
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

// Built once per curve (-DZK_CURVE_SEL=<0|1>); msm_dispatch.hip dispatches on the curve id.
#if ZK_CURVE_SEL == 0
typedef CurveBls CurveSel;
#define ZK_SYM(name) name##_c0
#else
typedef CurveBn CurveSel;
#define ZK_SYM(name) name##_c1
#endif

int ZK_SYM(msm_run_dev)(zk_ctx* c, const void* d_bases_xy, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    return msm_run<CurveSel>(c, d_bases_xy, d_scalars, n, out_xyz);
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
int ZK_SYM(msm_batch_pre_reduce_dev)(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_partials,
                                     int partial_kind) {
    return msm_batch_pre_reduce<CurveSel>(c, s, n_jobs, slots, lens, d_partials, partial_kind);
}
bool ZK_SYM(msm_partial_dev_supported)(zk_ctx* c, zk_srs* s, uint32_t* vw, uint32_t* vb) { return partial_dev_supported<CurveSel>(c, s, vw, vb); }
int ZK_SYM(g1_sum_winsums_dev)(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    return sum_winsums_dev<CurveSel>(c, s, d_all, ranks, n_jobs, out_xy, out_inf);
}
size_t ZK_SYM(msm_partial_dev_bytes)() { return (size_t)4 * Store<CurveSel::FqU>::WORDS * 4; }
int ZK_SYM(g1_sum_partials_dev)(zk_ctx* c, const void* d_parts, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    return sum_partials_dev<CurveSel>(c, d_parts, ranks, n_jobs, out_xy, out_inf);
}
void ZK_SYM(g1_jacobian_to_partial_host)(const uint64_t* xyz, void* out) { jacobian_to_partial_host<CurveSel>(xyz, out); }
int ZK_SYM(msm_batch_pre_end_dev)(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, uint64_t* out_xyz,
                                  uint64_t* out_xy, uint8_t* out_inf) {
    return msm_batch_pre_end<CurveSel>(c, s, n_jobs, slots, lens, out_xyz, out_xy, out_inf);
}

size_t ZK_SYM(msm_point_bytes)() { return (size_t)2 * Store<CurveSel::FqU>::WORDS * 4; }

int ZK_SYM(g1_jacobian_to_affine_host)(const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf) {
    return jac_to_affine<CurveSel::Fq>(xyz, out_xy, out_inf);
}

int ZK_SYM(g1_sum_partials_host)(const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    return sum_partials<CurveSel::Fq>(partials, count, out_xy, out_inf);
}
