// Internal context shared by the NTT / MSM translation units.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <list>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <condition_variable>
#include <functional>
#include <memory>
#include <vector>
#include <utility>

#include "../../include/ark_plonk_amd.h"
#include "curve_params.h"
#include "ec.cuh"
#include "fieldu.cuh"
#include "ecu.cuh"

typedef Fp<FrBls12_381Params> FrBls;
typedef Fp<FqBls12_381Params> FqBls;
typedef Fp<FrBn254Params> FrBn;
typedef Fp<FqBn254Params> FqBn;

struct CurveBls {
    typedef FrBls Fr;
    typedef FqBls Fq;
    typedef FrBls12_381Params FrP;
    typedef FqBls12_381Params FqP;
    typedef Fs<FqBls12_381SParams> FqU;   // device representation of the MSM base field: 13 signed 30-bit limbs (fields.cuh)
    typedef Fu<FrBls12_381UParams> FrU;
    static constexpr int ID = ZK_CURVE_BLS12_381;
};
struct CurveBn {
    typedef FrBn Fr;
    typedef FqBn Fq;
    typedef FrBn254Params FrP;
    typedef FqBn254Params FqP;
    typedef Fs<FqBn254SParams> FqU;       // 9 signed 30-bit limbs
    typedef Fu<FrBn254UParams> FrU;
    static constexpr int ID = ZK_CURVE_BN254;
};

#define ZK_HIP_TRY(expr)                                  \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) {                           \
            zk_note_hip_error(_e, #expr, __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? ZK_ERR_OOM : ZK_ERR_HIP; \
        }                                                 \
    } while (0)

void zk_note_hip_error(hipError_t e, const char* what, const char* file, int line);

// device buffer that grows on demand and is reused across calls (no hipMalloc in steady state)
// Persistent helper threads for the host-side tails of a batch (window combine + affine normalisation:
// ~70 us of serial field arithmetic per job while the GPU waits).  Creating a std::thread costs ~35 us on
// these hosts -- as much as half a job -- so the workers are started once and woken per call; the caller
// takes items too.  run() is not re-entrant (one batch at a time per pool; zk_ctx calls hold its mutex).
class HostPool {
  public:
    explicit HostPool(unsigned workers) {
        for (unsigned i = 0; i < workers; ++i) th_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;
    void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || th_.empty()) {
            for (uint32_t k = 0; k < n; ++k) fn(k);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            n_ = n;
            next_ = 0;
            pending_ = n;
            ++epoch_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work() {   // take items until none is left
        for (;;) {
            uint32_t k;
            const std::function<void(uint32_t)>* f;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!fn_ || next_ >= n_) return;
                k = next_++;
                f = fn_;
            }
            (*f)(k);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(uint32_t)>* fn_ = nullptr;
    uint32_t n_ = 0, next_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

// fn(k) for k in [0, n) on the process-wide pool (host-only entry points without a ctx)
void host_parallel_for(uint32_t n, const std::function<void(uint32_t)>& fn);

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // what ensure(bytes) allocates: a little slack so that lengths differing by a few elements (n, n - 1, n + 2: stripped zeros, the
    // openings) reuse the buffer.  The slack is bounded: an eighth of a 2.6 GB sort buffer was 330 MB per job of a 2^25 round.
    static size_t want_for(size_t bytes) {
        const size_t slack = bytes / 8 < ((size_t)8 << 20) ? bytes / 8 : ((size_t)8 << 20);
        return bytes + slack + 256;
    }
    // bytes ensure(bytes) would newly take from the device (0 when the buffer is already large enough)
    size_t need_for(size_t bytes) const { return bytes <= cap ? 0 : want_for(bytes); }
    int ensure(size_t bytes) {
        if (bytes <= cap) return ZK_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = want_for(bytes);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return ZK_ERR_OOM;
        }
        cap = want;
        return ZK_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct MsmBufs {
    DevBuf counts, offsets, entries, buckets, part_pt, part_key, seg, seg2, seg3, win, tmp, scalars, stage, upload;
    int stage_of_job = 0;   // table-path job living in this set: 0 none, 1 digits queued (placement + accumulation wait for the round's
                            // merged launches), 2 accumulated (reductions pending) -- msm_batch_pre_begin / _reduce
    // the chunk plan the job's accumulation launch was sized with (pre_queue_accumulate): the reductions read the chunk-edge partials
    // through exactly these two numbers, whatever a later pre_plan of the same job would derive (a round whose reduction
    // was refused after the accumulation had run was re-planned with the other chunk length)
    uint32_t acc_chunk_l = 0, acc_n_lanes = 0;
    void release() {
        release_work();
        scalars.release();
        upload.release();
    }
    // The WORK buffers of a set: everything a table-path job sorts, accumulates and reduces in.  `scalars` (into_repr output of the
    // single-MSM paths) and `upload` (the host-pointer batch's copy of job k's coefficients) belong to the SLOT, not to the job's work:
    // callers hold pointers into them across the begin of other jobs, so they are neither swapped nor released by the memory budget.
    void work_bufs(DevBuf* (&out)[12]) {
        DevBuf* all[12] = {&counts, &offsets, &entries, &buckets, &part_pt, &part_key, &seg, &seg2, &seg3, &win, &tmp, &stage};
        for (int i = 0; i < 12; ++i) out[i] = all[i];
    }
    size_t work_bytes() {
        DevBuf* w[12];
        work_bufs(w);
        size_t t = 0;
        for (DevBuf* b : w) t += b->cap;
        return t;
    }
    void release_work() {
        DevBuf* w[12];
        work_bufs(w);
        for (DevBuf* b : w) b->release();
    }
    // a free set's work buffers handed to another slot (deferred rounds under a memory budget: msm_batch_pre_begin)
    void swap_work(MsmBufs& o) {
        DevBuf* a[12];
        DevBuf* b[12];
        work_bufs(a);
        o.work_bufs(b);
        for (int i = 0; i < 12; ++i) std::swap(*a[i], *b[i]);
    }
};

struct ProfEntry {
    double total_ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// one NTT plan = twiddle tables of one (curve, log_n, direction)
struct NttPlan {
    int curve = 0;
    uint32_t log_n = 0;
    bool inverse = false;
    int n_pass = 0;
    int s[4] = {0, 0, 0, 0};                 // pass radix exponents, sum = log_n
    void* tw_inner[4] = {nullptr, nullptr, nullptr, nullptr};  // omega_{2^s}^j, j < 2^s/2  (shared, not owned)
    void* tw_pass[4] = {nullptr, nullptr, nullptr, nullptr};   // inter-pass tables (owned)
    ~NttPlan() {
        for (int i = 0; i < 4; ++i)
            if (tw_pass[i]) (void)hipFree(tw_pass[i]);
    }
};

// Tuning options of a ctx (zk_ctx_set_option): plain integers read by the MSM planner.  They replace the ZK_* environment hooks of
// rounds 2-4 (getenv in a library racing a caller's setenv is undefined behaviour, and a number must not depend on ambient variables).
// Every default is "the library decides".  A/B tools set them per ctx; results are identical for every value.
struct ZkTune {
    int msm_merge = 1;       // 1: one sort / accumulation launch per round for all deferred jobs; 0: per job at submission (round 3's shape)
    int pre_vw = 0;          // virtual windows of the shared-bucket reduction (0 = default 64; a power of two 8 .. 512)
    int pre_logg = -1;       // log2 buckets per reduction segment (-1 = default)
    int chunk_l = 0;         // sorted references per accumulation lane (0 = planned)
    int long_rounds = 1;     // rounds of resident lanes for a non-final job of a merged accumulation launch
    int combine_sg = 0;      // lanes per small bucket in msm_combine when a launch has > 2 jobs (0 = default 1; 2; 4)
    int pre_max_log_n = 0;   // vectors longer than 2^this leave the window-table path (0 = the built-in 2^26); test hook, 13 .. 25
    // Memory budget of the table path.  A job's buffer set is taken only if the device has room for it (hipMemGetInfo
    // minus mem_reserve_mb); otherwise the jobs queued so far are closed first -- sorted, accumulated, reduced, their points parked on
    // the host in call order -- and their sets reused.  round_mem_limit_mb (test hook, 0 = off): additionally pretend that the sets of
    // the queued jobs together may hold at most this many MiB, so that the flush can be exercised at small sizes.
    int mem_reserve_mb = 1024;
    int round_mem_limit_mb = 0;
    int host_workers = -1;   // helper threads of the ctx's host pool (-1 = min(15, cores / LOCAL_WORLD_SIZE - 1)); read when the pool starts
};

struct zk_ctx {
    int device = 0;
    ZkTune tune;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::recursive_mutex mu;
    std::unique_ptr<HostPool> pool;   // created with the ctx (15 workers + the calling thread)
    int msm_window = 0;  // 0 = auto
    bool profiling = false;
    int profile_level = 0;   // 1: every scope; 2: msm_accumulate only (an event pair costs ~14 us of host time and a bubble on the stream)
    std::map<std::string, ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;

    // NTT state
    std::map<uint64_t, NttPlan*> plans;          // key = curve<<40 | inverse<<32 | log_n
    std::map<uint64_t, void*> inner_tw;          // key = curve<<40 | inverse<<32 | s
    DevBuf ntt_work;                             // N-element scratch between passes
    DevBuf coset_pow[2];                         // per curve: g^j
    DevBuf coset_inv_pow[2];                     // per curve: g^-j
    size_t coset_len[2] = {0, 0};
    size_t coset_inv_len[2] = {0, 0};
    DevBuf io_a, io_b;                           // staging for the host-buffer entry points

    // MSM state: mb[0] is the working set of a single MSM; a batch (<= 16 jobs) gives every job its own
    // set so that the jobs' bucket reductions can run as one fused launch
    MsmBufs mb[16];
    hipEvent_t ev_job[16] = {};
    void* pinned = nullptr;      // virtual-window sums of up to 16 batched MSMs land here
    size_t pinned_cap = 0;
    void* pinned_small = nullptr;   // 4 KiB: digests of the commitment cache
    DevBuf msm_tmp;       // infinity flags staging (SRS registration)
    DevBuf stage_shared;  // the partition sort's staging area for jobs too large to own one (msm plan: shared_stage): the jobs of a round
                          // are then placed one after the other instead of by one launch per kernel
    DevBuf witness;       // the witness polynomial of zk_kzg_open*: read by the digit kernel of its MSM in stream order, so one per ctx
    uint64_t round_flushes = 0;   // times the memory budget closed the queued jobs of a round early (zk_round_mem_stats)

    // host-pointer entry points (the drop-in boundary): pinned staging ring + a copy stream so that the upload of
    // polynomial k+1 runs under the MSM of polynomial k (hostio.hip)
    hipStream_t copy_stream = nullptr;
    static constexpr int STAGE_SLOTS = 4;
    static constexpr size_t STAGE_BYTES = (size_t)8 << 20;
    void* stage_pin[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t stage_ev[STAGE_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_busy[STAGE_SLOTS] = {false, false, false, false};
    int stage_next = 0;
    hipEvent_t ev_up[16] = {};     // "input of job k is on the device" (copy stream -> main stream)
    uint64_t h2d_bytes = 0, d2h_bytes = 0;   // PCIe volume of the host-pointer entry points (zk_io_stats)
    int staging_mode = 0;                     // 0 = plain hipMemcpyAsync from the caller's (pageable) buffer: measured 56 GB/s on the MI355X
                                              // hosts, the same as pinned memory (tools/pcie_probe.py); 1 = the ctx's pinned staging ring (49 GB/s)

    // N3: opt-in content-addressed commitment cache -- key = (srs id, input kind, length, 256-bit
    // digest of the coefficient vector computed on the device); value = the affine commitment
    struct CommitEntry {
        uint64_t srs_id;
        uint64_t n;
        uint32_t kind;
        uint64_t dig[4];
        uint64_t xy[12];
        uint8_t inf;
    };
    bool commit_cache_on = false;
    size_t commit_cache_cap = 64;
    std::list<CommitEntry> commit_cache;      // most recently used first
    uint64_t cache_hits = 0, cache_misses = 0;
    DevBuf digest_dev;                        // 16 jobs x 4 u64
    uint64_t digest_key[4] = {0, 0, 0, 0};    // zk_process_key mixed with the ctx's address: keys the device digests of this cache

    // Residency cache of the HOST-POINTER entry points (zk_ctx_set_residency_cache; opt-in): device copies of vectors this ctx
    // produced (zk_ntt outputs of at most res_max_vec bytes) or uploaded, keyed by (bytes, keyed 256-bit digest of the HOST bytes).
    // A later zk_ntt / zk_kzg_commit_batch / zk_kzg_open input with the same bytes uses the copy instead of crossing PCIe again
    // (prover.rs:196-213: an `ifft` output goes straight back up as a `commit` input, then into `coset_fft`, `open`, ...).
    struct ResEntry {
        size_t bytes = 0;
        uint64_t dig[4] = {0, 0, 0, 0};
        bool valid = false;          // dig names the buffer's contents (false while an output is still being produced)
        uint64_t epoch = 0;          // the call that last used it: entries of the running call are never evicted
        uint64_t born = 0;           // the call that created it: until that call returns its bytes may still be on their way up
        DevBuf buf;
    };
    bool res_on = false;
    size_t res_cap = (size_t)2 << 30, res_max_vec = (size_t)64 << 20, res_bytes = 0;
    std::list<ResEntry> res;         // most recently used first
    std::vector<DevBuf> res_free;    // buffers of evicted entries, reused before anything is allocated
    uint64_t res_epoch = 0, res_hits = 0, res_misses = 0;
    // option "cache_verify": every hit of the commitment / residency cache is checked against the real thing (the MSM recomputed, the
    // resident bytes compared with the caller's); a mismatch is counted and the computed / uploaded value used (zk_cache_verify_stats)
    bool cache_verify = false;
    bool key_from_os = false;          // zk_process_key delivered OS entropy when the ctx was created: the caches may be switched on
    uint64_t verify_checked = 0, verify_mismatch = 0;
    std::vector<unsigned char> verify_host;

    // open round (zk_kzg_round_begin_dev / zk_kzg_open_begin_dev ... zk_kzg_round_end): jobs whose sort + accumulate are queued
    // on the stream and whose reduction waits for the round to close, in submission order.  Job k lives in buffer set mb[k].
    struct PendingJob {
        size_t n = 0;
        bool queued = false;        // false: computed at begin (no table / short vector / commitment cache) -- result below
        bool have_xyz = false;
        uint64_t xyz[18] = {};
        uint64_t xy[12] = {};
        uint8_t inf = 0;
    };
    uint32_t pend_n = 0;
    bool pend_reduced = false;      // zk_kzg_round_reduce ran: the round takes no further jobs, zk_kzg_round_end only waits
    void* pend_partials = nullptr;  // ... as zk_kzg_round_reduce_winsums_dev: the jobs' virtual-window sums are (being) written there, on the device
    void* pinned_jobs = nullptr;    // 16 x 512 B pinned: partials of jobs computed at submission, on their way to the device (async copies)
    hipEvent_t round_ev = nullptr;  // recorded behind the reduction kernels of a round (msm_batch_pre_reduce)
    uint32_t round_reduced = 0;     // jobs whose reductions are queued behind round_ev (0: none)
    zk_srs* pend_srs = nullptr;
    PendingJob pend[16];
};

// Give back the work buffers of every buffer set no job lives in (and the shared staging area): the last resort of the memory
// budget before a call returns ZK_ERR_OOM.  hipFree waits for the device, so kernels still reading them are safe.  ctx lock held.
inline size_t zk_release_free_work(zk_ctx* c, int keep_slot) {
    size_t freed = 0;
    for (int j = 0; j < 16; ++j) {
        if (j == keep_slot || c->mb[j].stage_of_job != 0) continue;
        freed += c->mb[j].work_bytes();
        c->mb[j].release_work();
    }
    freed += c->stage_shared.cap;
    c->stage_shared.release();
    return freed;
}

// An SRS belongs to a DEVICE, not to a ctx: every zk_ctx of that device may use it (several proof streams share one
// copy of the bases and of the window table).  Readers (the MSM entry points) hold `mu` shared from reading the
// pointers to the end of their kernel enqueue; zk_srs_precompute takes it exclusively, drains the device and swaps
// the bases for the table.  refs counts user handles; a cached SRS (zk_srs_register) outlives refs == 0 until evicted.
struct zk_srs {
    int device = 0;
    uint64_t id = 0;
    std::shared_mutex mu;
    std::atomic<int> refs{1};
    bool cached = false;
    uint64_t digest[4] = {0, 0, 0, 0};
    int curve = 0;
    size_t n = 0;
    void* d_xy = nullptr;   // n points in the device-internal form (2 x Fs, signed 30-bit limbs, padded to 16 B);
                            // "no point" (infinity) is all-zero limbs
    size_t point_bytes = 0;
    // optional table of window multiples 2^(c*w) * P_i, w = 1 .. pre_W-1, window-major (n points each);
    // with it all windows of an MSM share ONE bucket set (no per-window reduction, no host doublings)
    // Window-sharded table (zk_srs_precompute_rows: one rank of a multi-GPU MSM owns the windows first, first + stride, ...): the rows
    // live in d_pre (pre_rows x n points, row j = 2^(c (first + j stride)) P_i) and d_xy stays the plain SRS; with the whole table
    // (stride 1) d_pre is null and d_xy IS the table, row 0 being the SRS itself.
    void* d_pre = nullptr;
    uint32_t pre_c = 0, pre_W = 0;         // window bits, windows of a full-width scalar (all ranks' rows together)
    uint32_t pre_rows = 0, pre_w0 = 0, pre_wstep = 1;
    const void* table() const { return d_pre ? d_pre : d_xy; }
};

// profiling helpers (ctx mutex held by caller)
struct ProfScope {
    zk_ctx* c;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st = nullptr;
    ProfScope(zk_ctx* ctx, const char* nm);
    ProfScope(zk_ctx* ctx, const char* nm, hipStream_t stream);
    ~ProfScope();
};
void zk_prof_collect(zk_ctx* c);

// implemented in ntt.hip / msm.hip
int ntt_run_dev(zk_ctx* c, int curve, int kind, uint32_t log_n, const void* d_in, size_t in_len, void* d_out);
int ntt_run_batch_dev(zk_ctx* c, int curve, int kind, uint32_t log_n, uint32_t n_polys, const void* const* d_ins, const size_t* in_lens,
                      void* const* d_outs);
int ntt_prepare(zk_ctx* c, int curve, uint32_t log_n);
void ntt_ctx_free(zk_ctx* c);
int fr_convert_dev(zk_ctx* c, int curve, int to_mont, const void* d_in, size_t n, void* d_out);
int fr_mul_dev(zk_ctx* c, int curve, const void* a, const void* b, size_t n, void* out);

// MSM over device bases/scalars; writes per-window sums back to host and combines there.
// out_xyz: Jacobian (X,Y,Z) 3L u64 limbs on host.
int msm_run_dev(zk_ctx* c, int curve, const void* d_bases_xy, const void* d_scalars, size_t n, uint64_t* out_xyz);
int msm_fixed_base_dev(zk_ctx* c, int curve, const void* d_scalars, size_t n, void* d_out_xy);
// window-multiples table of an SRS (see zk_srs::d_pre) and the MSM that uses it
int msm_precompute_dev(zk_ctx* c, zk_srs* s, uint32_t window_bits /* 0 = default (16); 16 .. 21 */, uint32_t first_window = 0, uint32_t window_stride = 1);
int msm_run_pre_dev(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz);
// a batch of commitments over one SRS, queued back to back; the host blocks once per result
// out_xy / out_inf (optional): also normalise every result to affine (n_polys x 2L limbs, n_polys flags)
// before_job(k) (optional) is called right before job k's kernels are queued on the ctx stream
int msm_batch_pre_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens, uint64_t* out_xyz,
                      const uint8_t* kinds = nullptr, uint64_t* out_xy = nullptr, uint8_t* out_inf = nullptr,
                      const std::function<int(uint32_t)>* before_job = nullptr);
// the two halves of msm_batch_pre_dev: queue sort + accumulate of n_polys jobs into the buffer sets c->mb[slot0 ..] / reduce the
// jobs in slots[0 .. n_jobs) with one launch per reduction kernel, wait once, combine on the host
int msm_batch_pre_begin_dev(zk_ctx* c, zk_srs* s, uint32_t slot0, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens,
                            const uint8_t* kinds = nullptr, const std::function<int(uint32_t)>* before_job = nullptr);
// d_winsums (optional, n_jobs pointers): the job's 2 VW virtual-window sums are left there, on the device (zk_winsums_dev_bytes each)
int msm_batch_pre_reduce_dev(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_winsums = nullptr);
// whether (and in which geometry) the device-resident forms of the exchange exist for this SRS's table: vw = virtual windows, vb = buckets
// of each; false for tables the reduction finishes on the host side only (window_bits >= 18) or without a table
bool msm_partial_dev_supported(zk_ctx* c, zk_srs* s, uint32_t* vw, uint32_t* vb);
// multi-GPU exchange on the device: bytes of one point in the internal XYZZ form; the 2 VW virtual-window sums of a job (S_v | T_v) are
// added element-wise over the ranks and combined on the host pool as the single-GPU path combines them
size_t msm_partial_dev_bytes(int curve);
int g1_sum_winsums_dev(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf);
int g1_jacobian_to_partial_host(int curve, const uint64_t* xyz, void* out);
int msm_batch_pre_end_dev(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, uint64_t* out_xyz,
                          uint64_t* out_xy = nullptr, uint8_t* out_inf = nullptr);
int fr_convert_stream(zk_ctx* c, int curve, const void* d_in, size_t n, void* d_out, hipStream_t st);
constexpr size_t ZK_PRE_MIN_N = 1u << 13;   // below this the per-window path is used
constexpr size_t ZK_PRE_MAX_N = 1u << 26;   // ... and above this: a sorted reference of the table path is sign | 5 bits of window | 26 bits of point index
// the limit the dispatch uses: ZK_PRE_MAX_N, or 2^pre_max_log_n of the ctx's options when that is set (test hook: the fall-back to
// the per-window path over a table SRS can then be exercised without a 130 GiB table).  Ranks of a window-sharded MSM must agree on it.
inline size_t zk_pre_max_n(const zk_ctx* c) {
    const int v = c->tune.pre_max_log_n;
    if (v >= 13 && v < 26) return (size_t)1 << v;
    return ZK_PRE_MAX_N;
}
// arkworks-layout affine bases (x||y, Montgomery R = 2^(64L)) -> device-internal points
int msm_convert_bases_dev(zk_ctx* c, int curve, const void* d_xy_sat, const uint8_t* d_inf, size_t n, void* d_out_internal);
size_t msm_point_bytes(int curve);
int g1_jacobian_to_affine_host(int curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf);
int g1_sum_partials_host(int curve, const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf);
int perm_product_dev(zk_ctx* c, int curve, uint32_t log_n, const void* const* d_wires, const void* const* d_sigmas,
                     const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, uint64_t* last_mont);
int lookup_product_dev(zk_ctx* c, int curve, size_t n, const void* d_f, const void* d_t, const void* d_h1, const void* d_h2,
                       const uint64_t* delta_mont, const uint64_t* eps_mont, void* d_out, uint64_t* last_mont);
int quad_selftest_dev(zk_ctx* c, int curve, uint32_t n_quads, uint32_t* out2);
int quotient_evals_dev(zk_ctx* c, int curve, uint32_t log_n, const zk_quotient_args* q, void* d_out);
int lookup_query_dev(zk_ctx* c, int curve, size_t n, const void* d_q, size_t q_len, const void* const* d_w, const uint64_t* zeta_mont,
                     const void* d_table, void* d_out);
int lookup_combine_split_dev(zk_ctx* c, const void* d_t, size_t n_t, const void* d_f, size_t n_f, void* d_h1, void* d_h2, size_t* len_h1,
                             size_t* len_h2);
int poly_evaluate_dev(zk_ctx* c, int curve, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* points_mont,
                      uint64_t* out_mont);
int poly_lincomb_dev(zk_ctx* c, int curve, uint32_t n_terms, const void* const* d_polys, const size_t* lens, const uint64_t* coeffs_mont,
                     void* d_out, size_t out_len);
// the witness lands in c->witness: its MSM's digit kernel reads it in stream order, before the next call can overwrite it
int kzg_open_prepare_dev(zk_ctx* c, int curve, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                         const uint64_t* z_mont, const uint64_t* chal_mont, void** d_witness_canonical, size_t* wlen);

// hostio.hip: staged host<->device copies and digests
// Copies run on `st`; h2d returns once the host buffer has been read (the device copy may still be in flight on st),
// d2h returns once the host buffer is filled.
int zk_h2d(zk_ctx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t st);
int zk_d2h(zk_ctx* c, void* h_dst, const void* d_src, size_t bytes, hipStream_t st);
void zk_io_release(zk_ctx* c);
// 256-bit digest of a host buffer (4 lanes; block-parallel on the ctx-less pool)
void host_digest256(const void* p, size_t bytes, uint64_t seed, uint64_t out[4]);
// the same digest for several buffers at once on a given pool (all 1 MiB blocks of all buffers as one batch of work items)
void host_digest256_multi(HostPool* pool, const void* const* ptrs, const size_t* bytes, uint32_t n, uint64_t seed, uint64_t (*out)[4]);
// 256-bit multiset digests of n_jobs device vectors of 32-byte elements -> d_out[job][4] (async on st)
// keyed with `key` (the ctx's digest_key): see hostio.hip
int dev_digest256(const void* const* d_ptrs, const size_t* lens, uint32_t n_jobs, uint64_t* d_out, hipStream_t st, const uint64_t key[4]);
// 256 random bits per process (operating-system entropy), the key of every cache digest; false: the OS gave none (hostio.hip)
bool zk_process_key(uint64_t out[4]);
void zk_process_key_reset_for_tests(const char* path);
