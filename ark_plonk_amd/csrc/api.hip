// C ABI of the MI355X NTT + MSM hot path (include/ark_plonk_amd.h).  Thin: argument checks,
// host<->device staging for the host-buffer entry points, ctx / SRS lifetime, profiling.
#include "ctx.h"

#include <cstdio>
#include <cstring>

static thread_local char g_last_hip[256];

void zk_note_hip_error(hipError_t e, const char* what, const char* file, int line) {
    snprintf(g_last_hip, sizeof g_last_hip, "%s (%s) at %s:%d", hipGetErrorString(e), what, file, line);
    if (getenv("ZK_VERBOSE")) fprintf(stderr, "[ark_plonk_amd] HIP error: %s\n", g_last_hip);
}

// ------------------------------------------------------------------------------------- profiling
ProfScope::ProfScope(zk_ctx* ctx, const char* nm) : ProfScope(ctx, nm, ctx->stream) {}
ProfScope::ProfScope(zk_ctx* ctx, const char* nm, hipStream_t stream) : c(ctx), name(nm), st(stream) {
    if (!c->profiling) return;
    if (c->profile_level == 2 && strcmp(nm, "msm_accumulate") != 0) return;   // level 2: the dominant kernel only
    auto take = [&]() -> hipEvent_t {
        if (!c->event_pool.empty()) {
            hipEvent_t e = c->event_pool.back();
            c->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    };
    a = take();
    b = take();
    (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    c->prof[name].pending.emplace_back(a, b);
}
void zk_prof_collect(zk_ctx* c) {
    for (auto& kv : c->prof) {
        ProfEntry& pe = kv.second;
        for (auto& ev : pe.pending) {
            (void)hipEventSynchronize(ev.second);
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                pe.total_ms += ms;
                pe.launches += 1;
            }
            c->event_pool.push_back(ev.first);
            c->event_pool.push_back(ev.second);
        }
        pe.pending.clear();
    }
}

void host_parallel_for(uint32_t n, const std::function<void(uint32_t)>& fn) {
    static std::mutex mu;                     // HostPool::run is one batch at a time
    static HostPool pool(7);
    std::lock_guard<std::mutex> lk(mu);
    pool.run(n, fn);
}

namespace {

inline int fq_limbs64(int curve) { return curve == ZK_CURVE_BLS12_381 ? 6 : curve == ZK_CURVE_BN254 ? 4 : 0; }
// Montgomery one of the base field, arkworks layout (u64 limbs)
inline void fq_one_sat(int curve, uint64_t* out) {
    if (curve == ZK_CURVE_BLS12_381) {
        const FqBls o = FqBls::one();
        memcpy(out, o.v, 48);
    } else {
        const FqBn o = FqBn::one();
        memcpy(out, o.v, 32);
    }
}

struct Guard {
    zk_ctx* c;
    std::unique_lock<std::recursive_mutex> lk;
    int prev = -1;
    bool ok = true;
    explicit Guard(zk_ctx* ctx) : c(ctx), lk(ctx->mu) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) ok = hipSetDevice(c->device) == hipSuccess;
    }
    ~Guard() {
        if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    }
};

int finish_point(int curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf) {
    return g1_jacobian_to_affine_host(curve, xyz, out_xy, out_inf);
}

// ------------------------------------------------------------------------------------------- a1
template <class C>
static int domain_new(uint64_t num_coeffs, zk_domain_info* out) {
    typedef typename C::Fr Fr;
    uint64_t size = 1;
    uint32_t lg = 0;
    while (size < num_coeffs) {
        size <<= 1;
        ++lg;
        if (lg > 63) return ZK_ERR_DOMAIN_TOO_LARGE;
    }
    if (lg > (uint32_t)C::FrP::TWO_ADICITY) return ZK_ERR_DOMAIN_TOO_LARGE;
    memset(out, 0, sizeof *out);
    out->size = size;
    out->log_size_of_group = lg;
    Fr root;
    for (int i = 0; i < Fr::N; ++i) root.v[i] = C::FrP::ROOT(i);
    for (uint32_t k = lg; k < (uint32_t)C::FrP::TWO_ADICITY; ++k) root = Fr::sqr(root);
    Fr gen = Fr::from_u32(C::FrP::GENERATOR);
    Fr size_inv = Fr::inverse(Fr::from_u64(size));
    Fr root_inv = Fr::inverse(root);
    Fr gen_inv = Fr::inverse(gen);
    memcpy(out->size_inv, size_inv.v, 32);
    memcpy(out->group_gen, root.v, 32);
    memcpy(out->group_gen_inv, root_inv.v, 32);
    memcpy(out->generator, gen.v, 32);
    memcpy(out->generator_inv, gen_inv.v, 32);
    return ZK_OK;
}

// a deferred round (zk_kzg_round_begin_dev ...) holds the ctx's MSM buffer sets until zk_kzg_round_end
inline bool round_open(const zk_ctx* c) { return c->pend_n != 0; }

int ensure_pinned_small(zk_ctx* c) {
    if (c->pinned_small) return ZK_OK;
    if (hipHostMalloc(&c->pinned_small, 4096, hipHostMallocDefault) != hipSuccess) return ZK_ERR_OOM;
    return ZK_OK;
}

constexpr size_t PINNED_JOB_SLOT = 512;
int ensure_pinned_jobs(zk_ctx* c) {
    if (c->pinned_jobs) return ZK_OK;
    if (hipHostMalloc(&c->pinned_jobs, 16 * PINNED_JOB_SLOT, hipHostMallocDefault) != hipSuccess) return ZK_ERR_OOM;
    return ZK_OK;
}

// ---- residency cache of the host-pointer entry points (ctx lock held) ------------------------------------------------
constexpr uint64_t RES_SEED = 0x52455349444Eull;

inline bool res_wants(const zk_ctx* c, size_t bytes) { return c->res_on && bytes >= 4096 && bytes <= c->res_max_vec; }

// every host-pointer call that consults the cache starts a new epoch: what it touches stays until it returns
inline void res_begin_call(zk_ctx* c) { ++c->res_epoch; }

// the buffer of an evicted / dropped entry: kept for reuse (no hipMalloc in the steady state) while the free list stays within an
// eighth of the cache's capacity and eight buffers -- device memory beyond res_cap that no statistic showed
void res_recycle(zk_ctx* c, DevBuf& buf) {
    size_t held = 0;
    for (const DevBuf& b : c->res_free) held += b.cap;
    if (c->res_free.size() < 8 && held + buf.cap <= c->res_cap / 8) c->res_free.push_back(buf);
    else buf.release();
    buf = DevBuf();
}

zk_ctx::ResEntry* res_find(zk_ctx* c, size_t bytes, const uint64_t dig[4]) {
    for (auto it = c->res.begin(); it != c->res.end(); ++it)
        if (it->valid && it->bytes == bytes && memcmp(it->dig, dig, 32) == 0) {
            it->epoch = c->res_epoch;
            if (it != c->res.begin()) c->res.splice(c->res.begin(), c->res, it);
            ++c->res_hits;
            return &c->res.front();
        }
    ++c->res_misses;
    return nullptr;
}

void res_drop(zk_ctx* c, zk_ctx::ResEntry* e);

// option "cache_verify": a hit is believed only after the resident bytes were compared with the caller's (both streams drained, one
// blocking download); a mismatch -- two vectors under one keyed digest, or a bug -- is counted, the entry dropped, the call uploads
bool res_verify_hit(zk_ctx* c, zk_ctx::ResEntry* e, const void* host_bytes) {
    if (!c->cache_verify) return true;
    if (e->born == c->res_epoch) return true;       // created by this very call (the same vector twice in one batch): its upload has not been queued yet
    (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    c->verify_host.resize(e->bytes);
    ++c->verify_checked;
    bool same = hipMemcpy(c->verify_host.data(), e->buf.p, e->bytes, hipMemcpyDeviceToHost) == hipSuccess &&
                memcmp(c->verify_host.data(), host_bytes, e->bytes) == 0;
    if (!same) {
        ++c->verify_mismatch;
        --c->res_hits;
        ++c->res_misses;
        res_drop(c, e);
    }
    return same;
}

// a fresh entry of `bytes` at the front of the list (not yet valid), or nullptr when nothing may be evicted / allocated: the
// caller then takes the uncached path.  Evicted buffers are kept for reuse (hipFree synchronises the device).
zk_ctx::ResEntry* res_new(zk_ctx* c, size_t bytes) {
    while (c->res_bytes + bytes > c->res_cap) {
        auto it = c->res.end();
        bool found = false;
        while (it != c->res.begin()) {
            --it;
            if (it->epoch != c->res_epoch) {
                found = true;
                break;
            }
        }
        if (!found) return nullptr;
        c->res_bytes -= it->bytes;
        res_recycle(c, it->buf);
        c->res.erase(it);
    }
    DevBuf buf;
    for (size_t i = 0; i < c->res_free.size(); ++i)
        if (c->res_free[i].cap >= bytes && (buf.cap == 0 || c->res_free[i].cap < buf.cap)) buf = c->res_free[i];
    if (buf.cap) {
        for (size_t i = 0; i < c->res_free.size(); ++i)
            if (c->res_free[i].p == buf.p) {
                c->res_free.erase(c->res_free.begin() + (long)i);
                break;
            }
    } else if (buf.ensure(bytes) != ZK_OK) {
        return nullptr;
    }
    c->res.emplace_front();
    zk_ctx::ResEntry& e = c->res.front();
    e.bytes = bytes;
    e.buf = buf;
    e.epoch = c->res_epoch;
    e.born = c->res_epoch;
    e.valid = false;
    c->res_bytes += bytes;
    return &e;
}

void res_drop(zk_ctx* c, zk_ctx::ResEntry* e) {        // an entry whose production failed
    for (auto it = c->res.begin(); it != c->res.end(); ++it)
        if (&*it == e) {
            c->res_bytes -= it->bytes;
            res_recycle(c, it->buf);
            c->res.erase(it);
            return;
        }
}

// a failed call: whatever it touched or created may hold bytes that never arrived
void res_fail_call(zk_ctx* c) {
    for (auto it = c->res.begin(); it != c->res.end();) {
        if (it->epoch == c->res_epoch) {
            c->res_bytes -= it->bytes;
            res_recycle(c, it->buf);
            it = c->res.erase(it);
        } else {
            ++it;
        }
    }
}

// Every host-pointer call that consults the cache runs inside one of these.  An exit that was not marked ok() -- a failed allocation,
// a failed copy, a failed kernel launch, at ANY point of the call -- waits for both streams (uploads may still be in flight on the copy
// stream) and drops every entry the call touched or created: their digests may name bytes that never arrived (an early
// `return rc` used to leave such entries valid, and a retry with the same polynomials then committed over garbage).
struct ResCall {
    zk_ctx* c;
    bool active, good = false;
    explicit ResCall(zk_ctx* ctx) : c(ctx), active(ctx->res_on) {
        if (active) res_begin_call(c);
    }
    int done(int rc) {
        good = rc == ZK_OK;
        return rc;
    }
    ~ResCall() {
        if (!active || good) return;
        (void)hipStreamSynchronize(c->stream);
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        res_fail_call(c);
    }
};

void res_clear(zk_ctx* c) {
    for (auto& e : c->res) e.buf.release();
    c->res.clear();
    for (auto& b : c->res_free) b.release();
    c->res_free.clear();
    c->res_bytes = 0;
}

// Inputs of one host-pointer call: for every vector the cache may hold (res_wants), its digest (all of them as ONE batch of work on
// the ctx's host pool) and, on a hit, the device copy.  On a miss a fresh entry takes the upload (the caller copies into *d_ptr and
// the vector is resident from then on); where the cache cannot take it, *d_ptr stays null and the caller uses its own staging buffer.
struct ResInput {
    const void* d_ptr = nullptr;   // device copy to use (hit, or the fresh entry a miss uploads into)
    bool upload = true;            // the bytes still have to go up
};
void res_resolve(zk_ctx* c, uint32_t n, const void* const* h_ptrs, const size_t* bytes, ResInput* out) {
    const void* hp[16];
    size_t hb[16];
    uint32_t idx[16], m = 0;
    for (uint32_t k = 0; k < n && k < 16; ++k) {
        out[k] = ResInput();
        if (res_wants(c, bytes[k]) && h_ptrs[k]) {
            hp[m] = h_ptrs[k];
            hb[m] = bytes[k];
            idx[m++] = k;
        }
    }
    if (!m) return;
    uint64_t dig[16][4];
    host_digest256_multi(c->pool.get(), hp, hb, m, RES_SEED, dig);
    for (uint32_t j = 0; j < m; ++j) {
        const uint32_t k = idx[j];
        zk_ctx::ResEntry* e = res_find(c, hb[j], dig[j]);
        if (e && !res_verify_hit(c, e, hp[j])) e = nullptr;
        if (e) {
            out[k].d_ptr = e->buf.p;
            out[k].upload = false;
        } else if (zk_ctx::ResEntry* f = res_new(c, hb[j])) {
            memcpy(f->dig, dig[j], 32);
            f->valid = true;            // its bytes go up in stream order before anything reads them
            out[k].d_ptr = f->buf.p;
        }
    }
}

}  // namespace

extern "C" {

const char* zk_strerror(int code) {
    switch (code) {
    case ZK_OK: return "ok";
    case ZK_ERR_BAD_ARG: return "bad argument";
    case ZK_ERR_DOMAIN_TOO_LARGE: return "evaluation domain larger than the field's two-adicity";
    case ZK_ERR_HIP: return g_last_hip[0] ? g_last_hip : "HIP runtime error";
    case ZK_ERR_OOM: return "out of device memory";
    case ZK_ERR_NO_DEVICE: return "no usable HIP device";
    case ZK_ERR_UNSUPPORTED: return "size not supported";
    case ZK_ERR_NOT_INVERTIBLE: return "zero denominator in a grand product";
    case ZK_ERR_NOT_INDEXED: return "lookup query value not in the table";
    case ZK_ERR_PENDING: return "a deferred commitment round is open on this ctx (zk_kzg_round_end closes it)";
    default: return "unknown error";
    }
}

const char* zk_build_info(void) { return "ark_plonk_amd gfx950 (CDNA4) hipcc; NTT+MSM hot path"; }

int zk_ctx_create(int device, zk_ctx** out) {
    if (!out) return ZK_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZK_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return ZK_ERR_BAD_ARG;
    int prev = 0;
    (void)hipGetDevice(&prev);
    ZK_HIP_TRY(hipSetDevice(device));
    zk_ctx* c = new zk_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        (void)hipSetDevice(prev);
        return ZK_ERR_HIP;
    }
    c->stream = c->own_stream;
    {
        const unsigned hc = std::thread::hardware_concurrency();
        c->tune.host_workers = hc > 16 ? 15 : hc > 1 ? (int)hc - 1 : 0;     // + the calling thread; option "host_workers" resizes it
        c->pool.reset(new HostPool((unsigned)c->tune.host_workers));
    }
    c->key_from_os = zk_process_key(c->digest_key);
    c->digest_key[0] ^= (uint64_t)(uintptr_t)c * 0x9E3779B97F4A7C15ull;      // caches are per ctx: so are their keys
    for (int i = 0; i < 16 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->ev_job[i], hipEventDisableTiming);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        zk_ctx_destroy(c);
        return ZK_ERR_HIP;
    }
    *out = c;
    return ZK_OK;
}

void zk_ctx_destroy(zk_ctx* c) {
    if (!c) return;
    {
        Guard g(c);
        (void)hipStreamSynchronize(c->stream);
        zk_prof_collect(c);
        for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
        ntt_ctx_free(c);
        DevBuf* bufs[] = {&c->io_a, &c->io_b, &c->msm_tmp, &c->stage_shared, &c->witness};
        for (DevBuf* b : bufs) b->release();
        for (int i = 0; i < 16; ++i) c->mb[i].release();
        res_clear(c);
        for (int i = 0; i < 16; ++i)
            if (c->ev_job[i]) (void)hipEventDestroy(c->ev_job[i]);
        if (c->round_ev) (void)hipEventDestroy(c->round_ev);
        if (c->pinned) (void)hipHostFree(c->pinned);
        if (c->pinned_small) (void)hipHostFree(c->pinned_small);
        if (c->pinned_jobs) (void)hipHostFree(c->pinned_jobs);
        zk_io_release(c);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
}

int zk_ctx_set_stream(zk_ctx* c, void* hip_stream) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)hip_stream;
    return ZK_OK;
}

int zk_ctx_use_own_stream(zk_ctx* c) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return ZK_OK;
}

int zk_ctx_sync(zk_ctx* c) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_ctx_set_msm_window(zk_ctx* c, int w) {
    if (!c || w < 0 || w > 16 || w == 1) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    c->msm_window = w;
    return ZK_OK;
}

// Tuning options (ZkTune, ctx.h).  Refused while a round is open: a job's plan must not change between its accumulation and its reduction.
static int* tune_field(zk_ctx* c, const char* key, int64_t* lo, int64_t* hi) {
    struct Row {
        const char* key;
        int ZkTune::*field;
        int64_t lo, hi;
    };
    static const Row rows[] = {
        {"msm_merge", &ZkTune::msm_merge, 0, 1},         {"pre_vw", &ZkTune::pre_vw, 0, 512},
        {"pre_logg", &ZkTune::pre_logg, -1, 5},          {"chunk_l", &ZkTune::chunk_l, 0, 1024},
        {"long_rounds", &ZkTune::long_rounds, 1, 16},    {"combine_sg", &ZkTune::combine_sg, 0, 4},
        {"pre_max_log_n", &ZkTune::pre_max_log_n, 0, 25}, {"mem_reserve_mb", &ZkTune::mem_reserve_mb, 0, 1 << 20},
        {"round_mem_limit_mb", &ZkTune::round_mem_limit_mb, 0, 1 << 20},
    };
    for (const Row& r : rows)
        if (strcmp(key, r.key) == 0) {
            *lo = r.lo;
            *hi = r.hi;
            return &(c->tune.*(r.field));
        }
    return nullptr;
}

int zk_ctx_set_option(zk_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    int64_t lo = 0, hi = 0;
    if (strcmp(key, "cache_verify") == 0) {
        if (value < 0 || value > 1) return ZK_ERR_BAD_ARG;
        c->cache_verify = value != 0;
        return ZK_OK;
    }
    if (strcmp(key, "host_workers") == 0) {
        if (value < 0 || value > 63) return ZK_ERR_BAD_ARG;
        c->pool.reset(new HostPool((unsigned)value));      // joins the old workers first (no batch is running: the ctx lock is held)
        c->tune.host_workers = (int)value;
        return ZK_OK;
    }
    int* f = tune_field(c, key, &lo, &hi);
    if (!f) return ZK_ERR_UNSUPPORTED;
    if (value < lo || value > hi) return ZK_ERR_BAD_ARG;
    if (f == &c->tune.pre_vw && value && (value < 8 || (value & (value - 1)))) return ZK_ERR_BAD_ARG;
    if (f == &c->tune.chunk_l && value && value < 8) return ZK_ERR_BAD_ARG;
    if (f == &c->tune.combine_sg && value == 3) return ZK_ERR_BAD_ARG;
    if (f == &c->tune.pre_max_log_n && value && value < 13) return ZK_ERR_BAD_ARG;
    *f = (int)value;
    return ZK_OK;
}

int zk_ctx_get_option(zk_ctx* c, const char* key, int64_t* value) {
    if (!c || !key || !value) return ZK_ERR_BAD_ARG;
    Guard g(c);
    int64_t lo = 0, hi = 0;
    if (strcmp(key, "host_workers") == 0) {
        *value = c->tune.host_workers;
        return ZK_OK;
    }
    if (strcmp(key, "cache_verify") == 0) {
        *value = c->cache_verify ? 1 : 0;
        return ZK_OK;
    }
    int* f = tune_field(c, key, &lo, &hi);
    if (!f) return ZK_ERR_UNSUPPORTED;
    *value = *f;
    return ZK_OK;
}

int zk_ctx_set_residency_cache(zk_ctx* c, int enable, size_t capacity_bytes, size_t max_vector_bytes) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    if (enable && !c->key_from_os) return ZK_ERR_UNSUPPORTED;     // as zk_ctx_set_commit_cache
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    if (capacity_bytes) c->res_cap = capacity_bytes;
    if (max_vector_bytes) c->res_max_vec = max_vector_bytes;
    c->res_on = enable != 0;
    if (!c->res_on) res_clear(c);
    return ZK_OK;
}

int zk_residency_cache_stats(zk_ctx* c, uint64_t* hits, uint64_t* misses, uint64_t* entries, uint64_t* bytes) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (hits) *hits = c->res_hits;
    if (misses) *misses = c->res_misses;
    if (entries) *entries = c->res.size();
    if (bytes) *bytes = c->res_bytes;
    return ZK_OK;
}

int zk_cache_verify_stats(zk_ctx* c, uint64_t* checked, uint64_t* mismatches) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (checked) *checked = c->verify_checked;
    if (mismatches) *mismatches = c->verify_mismatch;
    return ZK_OK;
}

int zk_profile_enable(zk_ctx* c, int on) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    c->profiling = on != 0;
    c->profile_level = on;
    return ZK_OK;
}
int zk_profile_reset(zk_ctx* c) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    zk_prof_collect(c);
    for (auto& kv : c->prof) {
        kv.second.total_ms = 0;
        kv.second.launches = 0;
    }
    return ZK_OK;
}
int zk_profile_get(zk_ctx* c, const char* name, double* total_ms, uint64_t* launches) {
    if (!c || !name) return ZK_ERR_BAD_ARG;
    Guard g(c);
    zk_prof_collect(c);
    auto it = c->prof.find(name);
    double t = 0;
    uint64_t n = 0;
    if (it != c->prof.end()) {
        t = it->second.total_ms;
        n = it->second.launches;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = n;
    return ZK_OK;
}

int zk_domain_new(int curve_id, uint64_t num_coeffs, zk_domain_info* out) {
    if (!out) return ZK_ERR_BAD_ARG;
    if (curve_id == ZK_CURVE_BLS12_381) return domain_new<CurveBls>(num_coeffs, out);
    if (curve_id == ZK_CURVE_BN254) return domain_new<CurveBn>(num_coeffs, out);
    return ZK_ERR_BAD_ARG;
}

// ---------------------------------------------------------------------------------------- a2-a5
int zk_ntt_dev(zk_ctx* c, int curve_id, int kind, uint32_t log_n, const void* d_in, size_t in_len, void* d_out) {
    if (!c || !d_out || (!d_in && in_len)) return ZK_ERR_BAD_ARG;
    if (log_n > 63) return ZK_ERR_DOMAIN_TOO_LARGE;
    Guard g(c);
    return ntt_run_dev(c, curve_id, kind, log_n, d_in, in_len, d_out);
}

int zk_ntt_batch_dev(zk_ctx* c, int curve_id, int kind, uint32_t log_n, uint32_t n_polys, const void* const* d_ins,
                     const size_t* in_lens, void* const* d_outs) {
    if (!c || (n_polys && (!d_ins || !in_lens || !d_outs))) return ZK_ERR_BAD_ARG;
    if (log_n > 63) return ZK_ERR_DOMAIN_TOO_LARGE;
    for (uint32_t i = 0; i < n_polys; ++i)
        if (!d_outs[i] || (!d_ins[i] && in_lens[i])) return ZK_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n_polys; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (d_outs[i] == d_outs[j]) return ZK_ERR_BAD_ARG;     // two results in one buffer
    Guard g(c);
    return ntt_run_batch_dev(c, curve_id, kind, log_n, n_polys, d_ins, in_lens, d_outs);
}

int zk_ntt_prepare(zk_ctx* c, int curve_id, uint32_t log_n) {
    if (!c) return ZK_ERR_BAD_ARG;
    if (log_n > 63) return ZK_ERR_DOMAIN_TOO_LARGE;
    Guard g(c);
    return ntt_prepare(c, curve_id, log_n);
}

int zk_ntt(zk_ctx* c, int curve_id, int kind, uint32_t log_n, const uint64_t* in, size_t in_len, uint64_t* out) {
    if (!c || !out || (!in && in_len)) return ZK_ERR_BAD_ARG;
    if (curve_id != ZK_CURVE_BLS12_381 && curve_id != ZK_CURVE_BN254) return ZK_ERR_BAD_ARG;
    if (log_n > 32) return ZK_ERR_DOMAIN_TOO_LARGE;
    Guard g(c);
    const size_t n = (size_t)1 << log_n;
    if (in_len > n) return ZK_ERR_BAD_ARG;
    int rc;
    // residency cache.  Which vectors come back is a property of the transforms' direction: the OUTPUT of an inverse transform is a
    // coefficient vector -- what PC::commit, PC::open and the coset transforms take next (prover.rs:196-213, quotient_poly.rs:72-120)
    // -- so it is produced into a cache entry and named by the digest of the bytes the caller receives; the INPUT of a forward
    // transform is a coefficient vector, so it is looked up (and not inserted on a miss: only commitments insert what they upload).
    // Inputs of inverse transforms and outputs of forward ones are evaluation vectors, made and consumed by host code: never digested.
    // (A policy about time only: a vector that is not looked up is simply uploaded.)
    const bool inverse = kind == ZK_NTT_IFFT || kind == ZK_NTT_COSET_IFFT;
    const void* d_in = nullptr;
    bool need_upload = in_len != 0;
    zk_ctx::ResEntry* out_entry = nullptr;
    ResCall rcall(c);
    if (c->res_on) {
        if (!inverse && res_wants(c, in_len * 32)) {
            const void* hp = in;
            const size_t hb = in_len * 32;
            uint64_t dig[1][4];
            host_digest256_multi(c->pool.get(), &hp, &hb, 1, RES_SEED, dig);
            if (zk_ctx::ResEntry* e = res_find(c, hb, dig[0])) {
                if (res_verify_hit(c, e, in)) {
                    d_in = e->buf.p;
                    need_upload = false;
                }
            }
        }
        if (inverse && res_wants(c, n * 32)) out_entry = res_new(c, n * 32);
    }
    if (!d_in) {
        if ((rc = c->io_a.ensure((in_len ? in_len : 1) * 32))) return rc;
        d_in = c->io_a.p;
    }
    void* d_out = out_entry ? out_entry->buf.p : nullptr;
    if (!d_out) {
        if ((rc = c->io_b.ensure(n * 32))) return rc;
        d_out = c->io_b.p;
    }
    if (need_upload && (rc = zk_h2d(c, const_cast<void*>(d_in), in, in_len * 32, c->stream))) return rc;
    rc = ntt_run_dev(c, curve_id, kind, log_n, d_in, in_len, d_out);
    if (!rc) rc = zk_d2h(c, out, d_out, n * 32, c->stream);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = ZK_ERR_HIP;
    if (rc) return rc;                 // ~ResCall drops out_entry with everything else this call touched
    if (out_entry) {
        const void* hp = out;
        const size_t hb = n * 32;
        uint64_t dig[1][4];
        host_digest256_multi(c->pool.get(), &hp, &hb, 1, RES_SEED, dig);
        memcpy(out_entry->dig, dig[0], 32);
        out_entry->valid = true;
    }
    return rcall.done(ZK_OK);
}

int zk_ntt_batch(zk_ctx* c, int curve_id, int kind, uint32_t log_n, uint32_t n_polys, const uint64_t* const* ins, const size_t* in_lens,
                 uint64_t* const* outs) {
    if (!c || (n_polys && (!ins || !in_lens || !outs))) return ZK_ERR_BAD_ARG;
    Guard g(c);   // one lock for the whole batch; the transforms share the plan
    for (uint32_t i = 0; i < n_polys; ++i) {
        int rc = zk_ntt(c, curve_id, kind, log_n, ins[i], in_lens[i], outs[i]);
        if (rc) return rc;
    }
    return ZK_OK;
}

int zk_fr_from_mont_dev(zk_ctx* c, int curve_id, const void* d_in, size_t n, void* d_out) {
    if (!c || (n && (!d_in || !d_out))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return fr_convert_dev(c, curve_id, 0, d_in, n, d_out);
}
int zk_fr_to_mont_dev(zk_ctx* c, int curve_id, const void* d_in, size_t n, void* d_out) {
    if (!c || (n && (!d_in || !d_out))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return fr_convert_dev(c, curve_id, 1, d_in, n, d_out);
}
int zk_fr_mul_dev(zk_ctx* c, int curve_id, const void* d_a, const void* d_b, size_t n, void* d_out) {
    if (!c || (n && (!d_a || !d_b || !d_out))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return fr_mul_dev(c, curve_id, d_a, d_b, n, d_out);
}

// ------------------------------------------------------------------------------------------ MSM
// ---- SRS registry: device-scoped, refcounted, content-addressed (PC::trim runs on every gen_proof,
// circuit.rs:276, so a second registration of the same powers_of_g must cost a lookup, not a 96 MiB upload + table build)
static std::mutex g_srs_mu;
static std::list<zk_srs*> g_srs_cache;            // cached entries, most recently used first
static std::atomic<uint64_t> g_srs_next_id{1};
static size_t g_srs_idle_limit = (size_t)32 << 30;   // bytes of UNREFERENCED cached SRS (incl. window tables) kept resident
static uint64_t g_srs_hits = 0, g_srs_misses = 0;

typedef std::shared_lock<std::shared_mutex> SrsReadLock;
static size_t srs_bytes(const zk_srs* s) { return s->n * s->point_bytes * (s->d_pre ? 1 + s->pre_rows : s->pre_W ? s->pre_W : 1); }

static void srs_destroy(zk_srs* s) {
    if (s->d_xy) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (prev != s->device) (void)hipSetDevice(s->device);
        (void)hipDeviceSynchronize();     // kernels of any ctx may still read the bases
        (void)hipFree(s->d_xy);
        if (s->d_pre) (void)hipFree(s->d_pre);
        if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
    }
    delete s;
}

// g_srs_mu held: drop least recently used unreferenced entries beyond the idle limit
static void srs_evict_locked() {
    size_t idle = 0;
    for (zk_srs* s : g_srs_cache)
        if (s->refs.load() == 0) idle += srs_bytes(s);
    for (auto it = g_srs_cache.end(); idle > g_srs_idle_limit && it != g_srs_cache.begin();) {
        --it;
        zk_srs* s = *it;
        if (s->refs.load() != 0) continue;
        idle -= srs_bytes(s);
        it = g_srs_cache.erase(it);
        srs_destroy(s);
    }
}

// shared tail of the registration entry points: d_sat = arkworks-layout points on the device
static int srs_build(zk_ctx* c, int curve_id, const void* d_sat, const uint8_t* d_inf, size_t n, zk_srs** out) {
    zk_srs* s = new zk_srs();
    s->device = c->device;
    s->id = g_srs_next_id.fetch_add(1);
    s->curve = curve_id;
    s->n = n;
    s->point_bytes = msm_point_bytes(curve_id);
    if (n) {
        if (hipMalloc(&s->d_xy, n * s->point_bytes) != hipSuccess) {
            delete s;
            return ZK_ERR_OOM;
        }
        int rc = msm_convert_bases_dev(c, curve_id, d_sat, d_inf, n, s->d_xy);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc || e != hipSuccess) {
            (void)hipFree(s->d_xy);
            delete s;
            return rc ? rc : ZK_ERR_HIP;
        }
    }
    *out = s;
    return ZK_OK;
}

int zk_srs_register_dev(zk_ctx* c, int curve_id, const void* d_bases_xy, const uint8_t* d_inf_flags, size_t n, zk_srs** out) {
    if (!c || !out || (n && !d_bases_xy)) return ZK_ERR_BAD_ARG;
    if (!fq_limbs64(curve_id)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return srs_build(c, curve_id, d_bases_xy, d_inf_flags, n, out);
}

// upload + convert; use_cache: look the content digest up first / publish the new handle
static int srs_register_host(zk_ctx* c, int curve_id, const uint64_t* bases_xy, const uint8_t* inf_flags, size_t n, zk_srs** out, bool use_cache) {
    int L = fq_limbs64(curve_id);
    if (!L) return ZK_ERR_BAD_ARG;
    Guard g(c);
    const size_t bytes = n * 2 * L * 8;
    uint64_t dig[4] = {0, 0, 0, 0};
    std::unique_lock<std::mutex> reg(g_srs_mu, std::defer_lock);
    if (!c->key_from_os) use_cache = false;      // no OS entropy behind the digest key: every registration builds its own copy
    if (use_cache) {
        host_digest256(bases_xy, bytes, 0x5125ull ^ ((uint64_t)curve_id << 32) ^ (uint64_t)n, dig);
        if (inf_flags) {
            uint64_t d2[4];
            host_digest256(inf_flags, n, 0xF1A65ull, d2);
            bool any = false;
            for (size_t i = 0; i < n && !any; ++i) any = inf_flags[i] != 0;
            if (any)   // an all-zero flag array is the same SRS as no flag array
                for (int k = 0; k < 4; ++k) dig[k] ^= d2[k];
        }
        reg.lock();   // held across the build: two threads registering the same SRS build it once
        for (auto it = g_srs_cache.begin(); it != g_srs_cache.end(); ++it) {
            zk_srs* s = *it;
            if (s->device == c->device && s->curve == curve_id && s->n == n && !memcmp(s->digest, dig, sizeof dig)) {
                s->refs.fetch_add(1);
                g_srs_cache.splice(g_srs_cache.begin(), g_srs_cache, it);
                ++g_srs_hits;
                *out = s;
                return ZK_OK;
            }
        }
        ++g_srs_misses;
    }
    int rc = c->io_b.ensure(bytes ? bytes : 1);
    if (rc) return rc;
    const uint8_t* d_inf = nullptr;
    if (n) {
        if ((rc = zk_h2d(c, c->io_b.p, bases_xy, bytes, c->stream))) return rc;
        if (inf_flags) {
            rc = c->msm_tmp.ensure(n);
            if (rc) return rc;
            if ((rc = zk_h2d(c, c->msm_tmp.p, inf_flags, n, c->stream))) return rc;
            d_inf = (const uint8_t*)c->msm_tmp.p;
        }
    }
    zk_srs* s = nullptr;
    rc = srs_build(c, curve_id, c->io_b.p, d_inf, n, &s);
    if (rc) return rc;
    if (use_cache) {
        s->cached = true;
        memcpy(s->digest, dig, sizeof dig);
        g_srs_cache.push_front(s);
        srs_evict_locked();
    }
    *out = s;
    return ZK_OK;
}

int zk_srs_register(zk_ctx* c, int curve_id, const uint64_t* bases_xy, const uint8_t* inf_flags, size_t n, zk_srs** out) {
    if (!c || !out || (n && !bases_xy)) return ZK_ERR_BAD_ARG;
    return srs_register_host(c, curve_id, bases_xy, inf_flags, n, out, n != 0);
}

int zk_srs_precompute_rows(zk_ctx* c, zk_srs* s, uint32_t window_bits, uint32_t first_window, uint32_t window_stride) {
    if (!c || !s || s->device != c->device) return ZK_ERR_BAD_ARG;
    if (window_bits != 0 && (window_bits < 16 || window_bits > 21)) return ZK_ERR_BAD_ARG;
    if (window_stride == 0 || first_window >= window_stride) return ZK_ERR_BAD_ARG;
    Guard g(c);
    std::unique_lock<std::shared_mutex> wl(s->mu);   // no MSM of any ctx is reading or enqueueing on this SRS
    if (s->n == 0) return ZK_OK;
    if (s->pre_W)                                    // one table per SRS: the first precompute wins
        return ((window_bits == 0 || window_bits == s->pre_c) && first_window == s->pre_w0 && window_stride == s->pre_wstep) ? ZK_OK : ZK_ERR_UNSUPPORTED;
    ZK_HIP_TRY(hipDeviceSynchronize());               // ... and none it enqueued earlier is still running
    return msm_precompute_dev(c, s, window_bits, first_window, window_stride);
}

int zk_srs_precompute_ex(zk_ctx* c, zk_srs* s, uint32_t window_bits) { return zk_srs_precompute_rows(c, s, window_bits, 0, 1); }

int zk_srs_table_rows(zk_srs* s, uint32_t* first_window, uint32_t* window_stride, uint32_t* rows) {
    if (!s) return ZK_ERR_BAD_ARG;
    SrsReadLock rl(s->mu);
    if (first_window) *first_window = s->pre_w0;
    if (window_stride) *window_stride = s->pre_wstep;
    if (rows) *rows = s->pre_rows;
    return ZK_OK;
}

int zk_srs_precompute(zk_ctx* c, zk_srs* s) { return zk_srs_precompute_ex(c, s, 0); }

int zk_srs_table_info(zk_srs* s, uint32_t* window_bits, uint32_t* windows) {
    if (!s) return ZK_ERR_BAD_ARG;
    SrsReadLock rl(s->mu);
    if (window_bits) *window_bits = s->pre_c;
    if (windows) *windows = s->pre_W;
    return ZK_OK;
}

int zk_srs_retain(zk_srs* s) {
    if (!s) return ZK_ERR_BAD_ARG;
    std::lock_guard<std::mutex> reg(g_srs_mu);
    if (s->refs.load() <= 0) return ZK_ERR_BAD_ARG;     // only a live handle can be shared
    s->refs.fetch_add(1);
    return ZK_OK;
}

void zk_srs_free(zk_srs* s) {
    if (!s) return;
    std::lock_guard<std::mutex> reg(g_srs_mu);
    const int left = s->refs.fetch_sub(1) - 1;
    if (left > 0) return;
    if (!s->cached) {
        srs_destroy(s);
        return;
    }
    srs_evict_locked();    // stays resident for the next PC::trim unless the idle budget is exceeded
}

size_t zk_srs_len(const zk_srs* s) { return s ? s->n : 0; }

int zk_srs_cache_config(size_t max_idle_bytes) {
    std::lock_guard<std::mutex> reg(g_srs_mu);
    g_srs_idle_limit = max_idle_bytes;
    srs_evict_locked();
    return ZK_OK;
}

int zk_srs_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* entries, uint64_t* resident_bytes) {
    std::lock_guard<std::mutex> reg(g_srs_mu);
    if (hits) *hits = g_srs_hits;
    if (misses) *misses = g_srs_misses;
    if (entries) *entries = g_srs_cache.size();
    if (resident_bytes) {
        uint64_t b = 0;
        for (zk_srs* s : g_srs_cache) b += srs_bytes(s);
        *resident_bytes = b;
    }
    return ZK_OK;
}

static int srs_slice(zk_srs* s, size_t base_offset, size_t n, const void** d_bases) {
    if (!s) return ZK_ERR_BAD_ARG;
    if (base_offset > s->n || n > s->n - base_offset) return ZK_ERR_BAD_ARG;
    *d_bases = (const char*)s->d_xy + base_offset * s->point_bytes;
    return ZK_OK;
}

typedef std::shared_lock<std::shared_mutex> SrsRead;

// ctx lock and SRS read lock held
static int msm_partial_locked(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    const void* d_bases = nullptr;
    int rc = srs_slice(s, base_offset, n, &d_bases);
    if (rc) return rc;
    if (s->pre_W && n >= ZK_PRE_MIN_N && n <= zk_pre_max_n(c) && c->msm_window == 0) return msm_run_pre_dev(c, s, base_offset, d_scalars, n, out_xyz);
    if (s->pre_wstep > 1 && s->pre_w0 != 0) {
        // window-sharded table: what the MSM entry points of this SRS return is the rank's PARTIAL, and the ranks' partials add up, so a
        // vector that does not take the table path is computed (whole, per-window path) by the owner of window 0 only; here: infinity
        const int L = fq_limbs64(s->curve);
        uint64_t one[6];
        fq_one_sat(s->curve, one);
        memcpy(out_xyz, one, sizeof(uint64_t) * L);
        memcpy(out_xyz + L, one, sizeof(uint64_t) * L);
        memset(out_xyz + 2 * L, 0, sizeof(uint64_t) * L);
        return ZK_OK;
    }
    return msm_run_dev(c, s->curve, d_bases, d_scalars, n, out_xyz);
}

int zk_msm_g1_srs_partial_dev(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    if (!c || !s || s->device != c->device || !out_xyz || (n && !d_scalars)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    return msm_partial_locked(c, s, base_offset, d_scalars, n, out_xyz);
}

int zk_msm_g1_srs_dev(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!out_xy) return ZK_ERR_BAD_ARG;
    uint64_t xyz[18];
    int rc = zk_msm_g1_srs_partial_dev(c, s, base_offset, d_scalars, n, xyz);
    if (rc) return rc;
    return finish_point(s->curve, xyz, out_xy, out_inf);
}

int zk_msm_g1_srs(zk_ctx* c, zk_srs* s, size_t base_offset, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || !s || s->device != c->device || !out_xy || (n && !scalars)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    {
        int rc = c->mb[0].scalars.ensure((n ? n : 1) * 32);
        if (rc) return rc;
        if (n && (rc = zk_h2d(c, c->mb[0].scalars.p, scalars, n * 32, c->stream))) return rc;
    }
    return zk_msm_g1_srs_dev(c, s, base_offset, c->mb[0].scalars.p, n, out_xy, out_inf);
}

int zk_msm_g1(zk_ctx* c, int curve_id, const uint64_t* bases_xy, const uint8_t* inf_flags, const uint64_t* scalars, size_t n,
              uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || !out_xy || (n && (!bases_xy || !scalars))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    zk_srs* s = nullptr;
    int rc = srs_register_host(c, curve_id, bases_xy, inf_flags, n, &s, false);   // ad-hoc bases: never cached
    if (rc) return rc;
    rc = zk_msm_g1_srs(c, s, 0, scalars, n, out_xy, out_inf);
    zk_srs_free(s);
    return rc;
}

int zk_g1_sum_partials(int curve_id, const uint64_t* partials_xyz, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    if (!out_xy || (count && !partials_xyz)) return ZK_ERR_BAD_ARG;
    return g1_sum_partials_host(curve_id, partials_xyz, count, out_xy, out_inf);
}

int zk_g1_sum_partials_batch(int curve_id, const uint64_t* partials_xyz, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    if (n_jobs == 0) return ZK_OK;
    if (!out_xy || !partials_xyz || ranks == 0) return ZK_ERR_BAD_ARG;
    if (curve_id != ZK_CURVE_BLS12_381 && curve_id != ZK_CURVE_BN254) return ZK_ERR_BAD_ARG;
    const size_t L = (size_t)fq_limbs64(curve_id);
    std::vector<int> rcs(n_jobs, 0);
    auto one = [&](uint32_t k) {
        std::vector<uint64_t> mine(ranks * 3 * L);
        for (size_t r = 0; r < ranks; ++r) memcpy(&mine[r * 3 * L], partials_xyz + (r * n_jobs + k) * 3 * L, 3 * L * sizeof(uint64_t));
        rcs[k] = g1_sum_partials_host(curve_id, mine.data(), ranks, out_xy + (size_t)k * 2 * L, out_inf ? out_inf + k : nullptr);
    };
    for (uint32_t base = 0; base < n_jobs; base += 16) {   // bounded thread count
        const uint32_t cnt = n_jobs - base < 16 ? n_jobs - base : 16;
        host_parallel_for(cnt, [&](uint32_t k) { one(base + k); });
    }
    for (uint32_t k = 0; k < n_jobs; ++k)
        if (rcs[k]) return rcs[k];
    return ZK_OK;
}

// ------------------------------------------------------------------------------------ KZG commit
typedef std::function<int(uint32_t)> BeforeJob;

// one commitment, no table / no batch: into_repr (unless canonical) + the per-window or single-job table MSM
static int commit_one_locked(zk_ctx* c, zk_srs* s, const void* d_in, size_t n, bool canonical, uint64_t* out_xyz) {
    if (n > s->n) return ZK_ERR_BAD_ARG;
    const void* sc = d_in;
    if (!canonical) {
        int rc = c->mb[0].scalars.ensure((n ? n : 1) * 32);
        if (rc) return rc;
        if ((rc = fr_convert_dev(c, s->curve, 0, d_in, n, c->mb[0].scalars.p))) return rc;
        sc = c->mb[0].scalars.p;
    }
    return msm_partial_locked(c, s, 0, sc, n, out_xyz);
}

// the jobs of one PC call, no cache: fused window-table batch when every job qualifies, else one at a time.
// before_job(k) (optional) runs right before job k's kernels are queued (the host-pointer batch uploads job k there).
// Exactly one of out_xyz (Jacobian partials, 3L per job) / out_xy (affine, 2L per job) is non-null.
static int batch_locked(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens, const uint8_t* kinds,
                        uint64_t* out_xyz, uint64_t* out_xy, uint8_t* out_inf, const BeforeJob* before_job) {
    const int L = fq_limbs64(s->curve);
    bool fused = s->pre_W != 0 && c->msm_window == 0;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        if (lens[k] > s->n || (lens[k] && !d_inputs[k])) return ZK_ERR_BAD_ARG;
        if (lens[k] < ZK_PRE_MIN_N || lens[k] > zk_pre_max_n(c)) fused = false;
    }
    if (fused) {
        uint64_t tmp[16 * 18];
        return msm_batch_pre_dev(c, s, n_jobs, d_inputs, lens, out_xyz ? out_xyz : tmp, kinds, out_xy, out_inf, before_job);
    }
    for (uint32_t k = 0; k < n_jobs; ++k) {
        int rc;
        if (before_job && (rc = (*before_job)(k))) return rc;
        uint64_t xyz[18];
        if ((rc = commit_one_locked(c, s, d_inputs[k], lens[k], kinds && kinds[k], out_xyz ? out_xyz + (size_t)k * 3 * L : xyz))) return rc;
        if (out_xy && (rc = finish_point(s->curve, xyz, out_xy + (size_t)k * 2 * L, out_inf ? out_inf + k : nullptr))) return rc;
    }
    return ZK_OK;
}

// N3: the same with the ctx's content-addressed commitment cache in front (affine outputs only).
// Inputs must already be on the device (the digest is computed there).
static int batch_cached_locked(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens, const uint8_t* kinds,
                               uint64_t* out_xy, uint8_t* out_inf) {
    const int L = fq_limbs64(s->curve);
    if (!c->commit_cache_on || n_jobs == 0) return batch_locked(c, s, n_jobs, d_inputs, lens, kinds, nullptr, out_xy, out_inf, nullptr);
    int rc;
    if ((rc = c->digest_dev.ensure(16 * 32))) return rc;
    if ((rc = ensure_pinned_small(c))) return rc;
    if ((rc = dev_digest256(d_inputs, lens, n_jobs, (uint64_t*)c->digest_dev.p, c->stream, c->digest_key))) return rc;
    ZK_HIP_TRY(hipMemcpyAsync(c->pinned_small, c->digest_dev.p, (size_t)n_jobs * 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    const uint64_t* dig = (const uint64_t*)c->pinned_small;
    const void* miss_in[16];
    size_t miss_len[16];
    uint8_t miss_kind[16];
    uint32_t miss_job[16], n_miss = 0;
    int alias[16];     // job k repeats miss alias[k] of this very call
    bool was_hit[16] = {false};      // option "cache_verify": a hit is computed all the same and compared below
    uint64_t hit_xy[16 * 12];
    uint8_t hit_inf[16];
    for (uint32_t k = 0; k < n_jobs; ++k) {
        const uint32_t kind = kinds && kinds[k] ? 1u : 0u;
        alias[k] = -1;
        bool hit = false;
        for (auto it = c->commit_cache.begin(); it != c->commit_cache.end(); ++it) {
            if (it->srs_id == s->id && it->n == lens[k] && it->kind == kind && !memcmp(it->dig, dig + 4 * k, 32)) {
                memcpy(out_xy + (size_t)k * 2 * L, it->xy, sizeof(uint64_t) * 2 * L);
                if (out_inf) out_inf[k] = it->inf;
                c->commit_cache.splice(c->commit_cache.begin(), c->commit_cache, it);
                hit = true;
                break;
            }
        }
        if (hit) {
            ++c->cache_hits;
            if (!c->cache_verify) continue;
            was_hit[k] = true;
            memcpy(hit_xy + (size_t)k * 12, out_xy + (size_t)k * 2 * L, sizeof(uint64_t) * 2 * L);
            hit_inf[k] = c->commit_cache.front().inf;
            miss_in[n_miss] = d_inputs[k];
            miss_len[n_miss] = lens[k];
            miss_kind[n_miss] = (uint8_t)kind;
            miss_job[n_miss] = k;
            ++n_miss;
            continue;
        }
        for (uint32_t m = 0; m < n_miss && alias[k] < 0; ++m) {
            const uint32_t j = miss_job[m];
            if (lens[j] == lens[k] && miss_kind[m] == kind && !memcmp(dig + 4 * j, dig + 4 * k, 32)) alias[k] = (int)m;
        }
        if (alias[k] >= 0) {
            ++c->cache_hits;
            continue;
        }
        ++c->cache_misses;
        miss_in[n_miss] = d_inputs[k];
        miss_len[n_miss] = lens[k];
        miss_kind[n_miss] = (uint8_t)kind;
        miss_job[n_miss] = k;
        ++n_miss;
    }
    uint64_t m_xy[16 * 12];
    uint8_t m_inf[16];
    if (n_miss) {
        if ((rc = batch_locked(c, s, n_miss, miss_in, miss_len, miss_kind, nullptr, m_xy, m_inf, nullptr))) return rc;
        for (uint32_t m = 0; m < n_miss; ++m) {
            const uint32_t k = miss_job[m];
            memcpy(out_xy + (size_t)k * 2 * L, m_xy + (size_t)m * 2 * L, sizeof(uint64_t) * 2 * L);
            if (out_inf) out_inf[k] = m_inf[m];
            if (was_hit[k]) {          // cache_verify: the entry exists; what it held against what was just computed
                ++c->verify_checked;
                if (hit_inf[k] != m_inf[m] || (!m_inf[m] && memcmp(hit_xy + (size_t)k * 12, m_xy + (size_t)m * 2 * L, sizeof(uint64_t) * 2 * L))) {
                    ++c->verify_mismatch;
                    for (auto& ce : c->commit_cache)
                        if (ce.srs_id == s->id && ce.n == lens[k] && ce.kind == miss_kind[m] && !memcmp(ce.dig, dig + 4 * k, 32)) {
                            memcpy(ce.xy, m_xy + (size_t)m * 2 * L, sizeof(uint64_t) * 2 * L);
                            ce.inf = m_inf[m];
                        }
                }
                continue;
            }
            zk_ctx::CommitEntry e;
            memset(&e, 0, sizeof e);
            e.srs_id = s->id;
            e.n = lens[k];
            e.kind = miss_kind[m];
            memcpy(e.dig, dig + 4 * k, 32);
            memcpy(e.xy, m_xy + (size_t)m * 2 * L, sizeof(uint64_t) * 2 * L);
            e.inf = m_inf[m];
            c->commit_cache.push_front(e);
        }
        while (c->commit_cache.size() > c->commit_cache_cap) c->commit_cache.pop_back();
    }
    for (uint32_t k = 0; k < n_jobs; ++k)
        if (alias[k] >= 0) {
            memcpy(out_xy + (size_t)k * 2 * L, m_xy + (size_t)alias[k] * 2 * L, sizeof(uint64_t) * 2 * L);
            if (out_inf) out_inf[k] = m_inf[alias[k]];
        }
    return ZK_OK;
}

int zk_ctx_set_commit_cache(zk_ctx* c, int enable, uint32_t capacity) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    if (enable && !c->key_from_os) return ZK_ERR_UNSUPPORTED;     // the digests that would address it have no OS entropy behind their key
    c->commit_cache_on = enable != 0;
    if (capacity) c->commit_cache_cap = capacity;
    if (!enable) c->commit_cache.clear();
    while (c->commit_cache.size() > c->commit_cache_cap) c->commit_cache.pop_back();
    return ZK_OK;
}

int zk_commit_cache_stats(zk_ctx* c, uint64_t* hits, uint64_t* misses, uint64_t* entries) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (hits) *hits = c->cache_hits;
    if (misses) *misses = c->cache_misses;
    if (entries) *entries = c->commit_cache.size();
    return ZK_OK;
}

int zk_kzg_commit_dev(zk_ctx* c, zk_srs* s, const void* d_coeffs_mont, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || !s || s->device != c->device || !out_xy || (n && !d_coeffs_mont)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    if (n > s->n) return ZK_ERR_BAD_ARG;
    if (c->commit_cache_on && n) return batch_cached_locked(c, s, 1, &d_coeffs_mont, &n, nullptr, out_xy, out_inf);
    uint64_t xyz[18];
    int rc = commit_one_locked(c, s, d_coeffs_mont, n, false, xyz);
    if (rc) return rc;
    return finish_point(s->curve, xyz, out_xy, out_inf);
}

int zk_kzg_commit_batch_partial_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs_mont, const size_t* lens,
                                    uint64_t* out_xyz) {
    return zk_kzg_round_batch_partial_dev(c, s, n_polys, d_coeffs_mont, lens, nullptr, out_xyz);
}

int zk_kzg_round_batch_partial_dev(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens,
                                   const uint8_t* kinds, uint64_t* out_xyz) {
    if (!c || !s || s->device != c->device || (n_jobs && (!d_inputs || !lens || !out_xyz))) return ZK_ERR_BAD_ARG;
    if (n_jobs > 16) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    return batch_locked(c, s, n_jobs, d_inputs, lens, kinds, out_xyz, nullptr, nullptr, nullptr);
}

int zk_kzg_commit_batch_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs_mont, const size_t* lens,
                            uint64_t* out_xy, uint8_t* out_inf) {
    return zk_kzg_round_batch_dev(c, s, n_polys, d_coeffs_mont, lens, nullptr, out_xy, out_inf);
}

int zk_kzg_round_batch_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs_mont, const size_t* lens,
                           const uint8_t* kinds, uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || !s || s->device != c->device || (n_polys && (!d_coeffs_mont || !lens || !out_xy))) return ZK_ERR_BAD_ARG;
    if (n_polys > 16) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    return batch_cached_locked(c, s, n_polys, d_coeffs_mont, lens, kinds, out_xy, out_inf);
}

// ---------------------------------------------------------------------- deferred rounds (begin ... end)
// ctx lock held.  Entry points that would reuse the buffer sets of queued jobs refuse while a round is open (ZK_ERR_PENDING).

static void round_clear(zk_ctx* c) {
    for (int k = 0; k < 16; ++k) c->mb[k].stage_of_job = 0;
    c->pend_n = 0;
    c->pend_srs = nullptr;
    c->pend_reduced = false;
    c->pend_partials = nullptr;
    c->round_reduced = 0;
}

// The memory budget closes the jobs queued so far: sorted, accumulated, reduced and combined NOW, their points parked
// in the pending entries in call order; their buffer sets are free for the jobs that follow.  The round itself stays open and
// zk_kzg_round_end returns the same points.  The caller holds the SRS lock.
static int round_flush_locked(zk_ctx* c) {
    zk_srs* s = c->pend_srs;
    uint32_t slots[16], nq = 0;
    size_t qlens[16];
    for (uint32_t k = 0; k < c->pend_n; ++k)
        if (c->pend[k].queued) {
            slots[nq] = k;
            qlens[nq] = c->pend[k].n;
            ++nq;
        }
    if (!nq) return ZK_OK;
    const int L = fq_limbs64(s->curve);
    uint64_t q_xyz[16 * 18];
    int rc = msm_batch_pre_end_dev(c, s, nq, slots, qlens, q_xyz, nullptr, nullptr);
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    for (uint32_t q = 0; q < nq; ++q) {
        zk_ctx::PendingJob& pj = c->pend[slots[q]];
        pj.queued = false;
        pj.have_xyz = true;
        memcpy(pj.xyz, q_xyz + (size_t)q * 3 * L, sizeof(uint64_t) * 3 * L);
    }
    ++c->round_flushes;
    return ZK_OK;
}

// one table-path job into the buffer set of its slot, under the memory budget: no room -> close what is queued and reuse its sets;
// still no room -> give back every free set's buffers; only then ZK_ERR_OOM
static int round_begin_job_locked(zk_ctx* c, zk_srs* s, uint32_t slot, const void* d_in, size_t len, uint8_t kind) {
    int rc = msm_batch_pre_begin_dev(c, s, slot, 1, &d_in, &len, &kind, nullptr);
    if (rc != ZK_ERR_OOM) return rc;
    if ((rc = round_flush_locked(c))) return rc;
    rc = msm_batch_pre_begin_dev(c, s, slot, 1, &d_in, &len, &kind, nullptr);
    if (rc != ZK_ERR_OOM) return rc;
    zk_release_free_work(c, (int)slot);
    return msm_batch_pre_begin_dev(c, s, slot, 1, &d_in, &len, &kind, nullptr);
}

// jobs [0, n_jobs) appended to the open round; inputs on the device
static int round_append_locked(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens, const uint8_t* kinds) {
    if (c->pend_n && c->pend_srs != s) return ZK_ERR_BAD_ARG;
    if (c->pend_reduced) return ZK_ERR_PENDING;          // zk_kzg_round_reduce closed the round to new jobs: zk_kzg_round_end first
    if (c->pend_n + n_jobs > 16) return ZK_ERR_UNSUPPORTED;
    for (uint32_t k = 0; k < n_jobs; ++k)
        if (lens[k] > s->n || (lens[k] && !d_inputs[k])) return ZK_ERR_BAD_ARG;
    const uint32_t slot0 = c->pend_n;
    int rc;
    if (c->commit_cache_on) {
        // the cache answers (or computes) at once; no job of such a round is ever queued, so the buffer sets are free
        uint64_t xy[16 * 12];
        uint8_t inf[16];
        const int L = fq_limbs64(s->curve);
        if ((rc = batch_cached_locked(c, s, n_jobs, d_inputs, lens, kinds, xy, inf))) return rc;
        for (uint32_t k = 0; k < n_jobs; ++k) {
            zk_ctx::PendingJob& pj = c->pend[slot0 + k];
            pj = zk_ctx::PendingJob();
            pj.n = lens[k];
            memcpy(pj.xy, xy + (size_t)k * 2 * L, sizeof(uint64_t) * 2 * L);
            pj.inf = inf[k];
        }
        c->pend_n += n_jobs;
        c->pend_srs = s;
        return ZK_OK;
    }
    const bool table = s->pre_W != 0 && c->msm_window == 0;
    for (uint32_t k = 0; k < n_jobs; ++k) {
        const uint32_t slot = slot0 + k;
        zk_ctx::PendingJob& pj = c->pend[slot];
        pj = zk_ctx::PendingJob();
        pj.n = lens[k];
        if (table && lens[k] >= ZK_PRE_MIN_N && lens[k] <= zk_pre_max_n(c)) {
            c->pend_srs = s;
            if ((rc = round_begin_job_locked(c, s, slot, d_inputs[k], lens[k], kinds ? kinds[k] : 0))) return rc;
            pj.queued = true;
        } else {
            // short vector / no table: computed now, in this job's own buffer set (set 0 may belong to a queued job)
            auto run_now = [&]() {
                if (slot) std::swap(c->mb[0], c->mb[slot]);
                const int r = commit_one_locked(c, s, d_inputs[k], lens[k], kinds && kinds[k], pj.xyz);
                if (slot) std::swap(c->mb[0], c->mb[slot]);
                return r;
            };
            rc = run_now();
            if (rc == ZK_ERR_OOM) {        // the memory budget, as for a table-path job: close what is queued, give its sets back, once more
                c->pend_srs = s;
                if ((rc = round_flush_locked(c))) return rc;
                zk_release_free_work(c, -1);
                rc = run_now();
            }
            if (rc) return rc;
            pj.have_xyz = true;
        }
        c->pend_n = slot + 1;      // a failure further on leaves the jobs queued so far open: zk_kzg_round_end / _abort settles them
        c->pend_srs = s;
    }
    return ZK_OK;
}

int zk_kzg_round_begin_dev(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens, const uint8_t* kinds) {
    if (!c || !s || s->device != c->device || (n_jobs && (!d_inputs || !lens))) return ZK_ERR_BAD_ARG;
    if (n_jobs > 16) return ZK_ERR_BAD_ARG;
    Guard g(c);
    SrsRead rl(s->mu);
    return round_append_locked(c, s, n_jobs, d_inputs, lens, kinds);
}

int zk_kzg_open_begin_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* z_mont,
                          const uint64_t* challenge_mont) {
    if (!c || !s || s->device != c->device || !z_mont || !challenge_mont || (n_polys && (!d_polys || !lens))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    SrsRead rl(s->mu);
    if (c->pend_reduced) return ZK_ERR_PENDING;
    if (c->pend_n >= 16) return ZK_ERR_UNSUPPORTED;
    if (c->pend_n && c->pend_srs != s) return ZK_ERR_BAD_ARG;
    void* d_w = nullptr;
    size_t wlen = 0;
    int rc = kzg_open_prepare_dev(c, s->curve, n_polys, d_polys, lens, z_mont, challenge_mont, &d_w, &wlen);
    if (rc == ZK_ERR_OOM) {
        // the combination / witness vectors (2 x 32 B per coefficient) found no room: close what is queued, free its sets, once more
        if ((rc = round_flush_locked(c))) return rc;
        zk_release_free_work(c, -1);
        rc = kzg_open_prepare_dev(c, s->curve, n_polys, d_polys, lens, z_mont, challenge_mont, &d_w, &wlen);
    }
    if (rc) return rc;
    if (wlen > s->n) return ZK_ERR_BAD_ARG;
    const void* in = d_w;
    const uint8_t kind = 1;
    return round_append_locked(c, s, 1, &in, &wlen, &kind);
}

// closes the round: one reduction launch per kernel for every queued job, one wait, results in submission order
static int round_end_locked(zk_ctx* c, uint32_t n_expected, uint64_t* out_xyz, uint64_t* out_xy, uint8_t* out_inf) {
    const uint32_t n = c->pend_n;
    zk_srs* s = c->pend_srs;
    if (n != n_expected) return ZK_ERR_BAD_ARG;          // the round stays open
    if (c->pend_partials) return ZK_ERR_PENDING;         // reduced towards the device (zk_kzg_round_reduce_winsums_dev): close it with _end_winsums_dev
    if (n == 0) return ZK_OK;
    const int L = fq_limbs64(s->curve);
    uint32_t slots[16], nq = 0;
    size_t qlens[16];
    for (uint32_t k = 0; k < n; ++k)
        if (c->pend[k].queued) {
            slots[nq] = k;
            qlens[nq] = c->pend[k].n;
            ++nq;
        }
    uint64_t q_xyz[16 * 18], q_xy[16 * 12];
    uint8_t q_inf[16];
    int rc = ZK_OK;
    if (nq) {
        SrsRead rl(s->mu);
        rc = msm_batch_pre_end_dev(c, s, nq, slots, qlens, q_xyz, out_xy ? q_xy : nullptr, q_inf);
        // a failure before the wait (a plan, the pinned buffer) leaves the round's kernels in flight: they still read the caller's
        // inputs and the SRS, which the header lets the caller free once this call has returned
        if (rc) (void)hipStreamSynchronize(c->stream);
    }
    uint32_t q = 0;
    for (uint32_t k = 0; k < n && !rc; ++k) {
        const zk_ctx::PendingJob& pj = c->pend[k];
        if (pj.queued) {
            if (out_xyz) memcpy(out_xyz + (size_t)k * 3 * L, q_xyz + (size_t)q * 3 * L, sizeof(uint64_t) * 3 * L);
            if (out_xy) {
                memcpy(out_xy + (size_t)k * 2 * L, q_xy + (size_t)q * 2 * L, sizeof(uint64_t) * 2 * L);
                if (out_inf) out_inf[k] = q_inf[q];
            }
            ++q;
        } else if (pj.have_xyz) {
            if (out_xyz) memcpy(out_xyz + (size_t)k * 3 * L, pj.xyz, sizeof(uint64_t) * 3 * L);
            if (out_xy) rc = finish_point(s->curve, pj.xyz, out_xy + (size_t)k * 2 * L, out_inf ? out_inf + k : nullptr);
        } else {
            if (out_xyz) rc = ZK_ERR_UNSUPPORTED;         // the commitment cache holds affine points only
            if (out_xy) {
                memcpy(out_xy + (size_t)k * 2 * L, pj.xy, sizeof(uint64_t) * 2 * L);
                if (out_inf) out_inf[k] = pj.inf;
            }
        }
    }
    round_clear(c);
    return rc;
}

int zk_kzg_round_reduce(zk_ctx* c) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (c->pend_n == 0 || c->pend_reduced) return ZK_OK;
    zk_srs* s = c->pend_srs;
    uint32_t slots[16], nq = 0;
    size_t qlens[16];
    for (uint32_t k = 0; k < c->pend_n; ++k)
        if (c->pend[k].queued) {
            slots[nq] = k;
            qlens[nq] = c->pend[k].n;
            ++nq;
        }
    if (nq) {
        SrsRead rl(s->mu);
        int rc = msm_batch_pre_reduce_dev(c, s, nq, slots, qlens);
        if (rc) return rc;
    }
    c->pend_reduced = true;
    return ZK_OK;
}

// ---- the same round closed with its result left ON THE DEVICE (multi-GPU exchange without a host hop): every job's 2 VW
// virtual-window sums S_v | T_v as the last reduction kernel writes them (zk_winsums_dev_bytes: 32 KiB per job at the default geometry).
// The combination sum_v S_v + B_v sum_v v T_v is linear in them, so the ranks add them element-wise after the all-gather
// (zk_g1_sum_winsums_dev: one throughput-shaped kernel) and the one combine per job stays on the host pool, as on a single GPU.
// (Round 4's other device form -- ONE point per job, formed by a further dependent quad launch -- measured last of the three
// exchanges on one card and was retired in round 6: profiles/design_history_msm.md.)
size_t zk_winsums_dev_bytes(zk_ctx* c, zk_srs* s) {
    if (!c || !s) return 0;
    Guard g(c);
    SrsRead rl(s->mu);
    uint32_t vw = 0, vb = 0;
    if (!msm_partial_dev_supported(c, s, &vw, &vb)) return 0;
    return (size_t)2 * vw * msm_partial_dev_bytes(s->curve);
}

int zk_winsums_geometry(zk_ctx* c, zk_srs* s, uint32_t out[4]) {
    if (!c || !s || !out) return ZK_ERR_BAD_ARG;
    Guard g(c);
    SrsRead rl(s->mu);
    uint32_t vw = 0, vb = 0;
    if (!msm_partial_dev_supported(c, s, &vw, &vb)) return ZK_ERR_UNSUPPORTED;
    out[0] = s->pre_c;
    out[1] = s->pre_W;
    out[2] = vw;
    out[3] = vb;
    return ZK_OK;
}

// ctx lock held: queues everything up to every job's window sums at d_out + k * zk_winsums_dev_bytes, k = submission order.
// Whatever makes the device form impossible is found BEFORE anything is queued, so that ZK_ERR_UNSUPPORTED really leaves the
// round as it was (the accumulation used to have run, with the long-chunk plan, when a c >= 18 table was refused).
static int round_reduce_winsums_dev_locked(zk_ctx* c, void* d_out) {
    zk_srs* s = c->pend_srs;
    const size_t pb = msm_partial_dev_bytes(s->curve);
    if (pb > PINNED_JOB_SLOT) return ZK_ERR_UNSUPPORTED;
    uint32_t slots[16], nq = 0;
    size_t qlens[16];
    void* outs[16];
    uint32_t n_host = 0;
    for (uint32_t k = 0; k < c->pend_n; ++k) {
        const zk_ctx::PendingJob& pj = c->pend[k];
        if (!pj.queued && !pj.have_xyz) return ZK_ERR_UNSUPPORTED;         // the commitment cache holds affine points only
        if (!pj.queued) ++n_host;
    }
    SrsRead rl(s->mu);
    uint32_t vw = 0, vb = 0;
    if (!msm_partial_dev_supported(c, s, &vw, &vb)) return ZK_ERR_UNSUPPORTED;   // no table, or one whose reduction finishes on the host (c >= 18)
    const size_t jb = (size_t)2 * vw * pb;                                  // bytes per job in d_out
    for (uint32_t k = 0; k < c->pend_n; ++k) {
        const zk_ctx::PendingJob& pj = c->pend[k];
        if (!pj.queued) continue;
        slots[nq] = k;
        qlens[nq] = pj.n;
        outs[nq] = (char*)d_out + (size_t)k * jb;
        ++nq;
    }
    int rc;
    if (n_host && (rc = ensure_pinned_jobs(c))) return rc;
    if (nq && (rc = msm_batch_pre_reduce_dev(c, s, nq, slots, qlens, outs))) return rc;
    // jobs computed at submission (vectors too short for the table path) or parked by the memory budget: their host Jacobian, converted,
    // goes up in one small copy each from a pinned slot of its own (truly asynchronous; the slot is reused by the next round of this ctx
    // only, which begins after this one was waited for).  As window sums such a job is S_0 = the point, every other sum the point at infinity.
    for (uint32_t k = 0; k < c->pend_n; ++k) {
        const zk_ctx::PendingJob& pj = c->pend[k];
        if (pj.queued) continue;
        unsigned char* slot = (unsigned char*)c->pinned_jobs + (size_t)k * PINNED_JOB_SLOT;
        if ((rc = g1_jacobian_to_partial_host(s->curve, pj.xyz, slot))) return rc;
        char* dst = (char*)d_out + (size_t)k * jb;
        ZK_HIP_TRY(hipMemsetAsync(dst, 0, jb, c->stream));
        ZK_HIP_TRY(hipMemcpyAsync(dst, slot, pb, hipMemcpyHostToDevice, c->stream));
    }
    c->pend_reduced = true;
    c->pend_partials = d_out;
    return ZK_OK;
}

int zk_kzg_round_reduce_winsums_dev(zk_ctx* c, void* d_out) {
    if (!c || !d_out) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (c->pend_n == 0) return ZK_OK;
    if (c->pend_reduced) return c->pend_partials == d_out ? ZK_OK : ZK_ERR_PENDING;
    return round_reduce_winsums_dev_locked(c, d_out);
}

int zk_kzg_round_end_winsums_dev(zk_ctx* c, uint32_t n_jobs, void* d_out) {
    if (!c || (n_jobs && !d_out)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (c->pend_n != n_jobs) return ZK_ERR_BAD_ARG;           // the round stays open
    if (n_jobs == 0) return ZK_OK;
    int rc = ZK_OK;
    if (!c->pend_reduced) rc = round_reduce_winsums_dev_locked(c, d_out);
    else if (c->pend_partials != d_out) rc = ZK_ERR_PENDING;    // reduced towards the host or another buffer
    if (rc == ZK_ERR_PENDING || rc == ZK_ERR_UNSUPPORTED) return rc;   // nothing was queued by this call: the round stays open for the host form
    if (rc) (void)hipStreamSynchronize(c->stream);          // as round_end_locked: kernels of the round may still read the inputs
    round_clear(c);
    return rc;
}

int zk_g1_sum_winsums_dev(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    if (n_jobs == 0) return ZK_OK;
    if (!c || !s || s->device != c->device || !d_all || !out_xy || ranks == 0) return ZK_ERR_BAD_ARG;
    if (n_jobs > 16) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    return g1_sum_winsums_dev(c, s, d_all, ranks, n_jobs, out_xy, out_inf);
}

int zk_kzg_round_end(zk_ctx* c, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || (n_jobs && !out_xy)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return round_end_locked(c, n_jobs, nullptr, out_xy, out_inf);
}

int zk_kzg_round_end_partial(zk_ctx* c, uint32_t n_jobs, uint64_t* out_xyz) {
    if (!c || (n_jobs && !out_xyz)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return round_end_locked(c, n_jobs, out_xyz, nullptr, nullptr);
}

int zk_kzg_round_pending(zk_ctx* c, uint32_t* n_jobs) {
    if (!c || !n_jobs) return ZK_ERR_BAD_ARG;
    Guard g(c);
    *n_jobs = c->pend_n;
    return ZK_OK;
}

int zk_round_mem_stats(zk_ctx* c, uint64_t* flushes, uint64_t* set_bytes, uint64_t* device_free, uint64_t* device_total) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (flushes) *flushes = c->round_flushes;
    if (set_bytes) {
        uint64_t t = c->stage_shared.cap;
        for (int k = 0; k < 16; ++k) t += c->mb[k].work_bytes();
        *set_bytes = t;
    }
    if (device_free || device_total) {
        size_t fr = 0, tot = 0;
        ZK_HIP_TRY(hipMemGetInfo(&fr, &tot));
        if (device_free) *device_free = fr;
        if (device_total) *device_total = tot;
    }
    return ZK_OK;
}

int zk_kzg_round_abort(zk_ctx* c) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (c->pend_n) (void)hipStreamSynchronize(c->stream);     // the queued kernels still read the caller's inputs
    round_clear(c);
    return ZK_OK;
}

static int ensure_copy_stream(zk_ctx* c) {
    if (!c->copy_stream) ZK_HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 16; ++i)
        if (!c->ev_up[i]) ZK_HIP_TRY(hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
    return ZK_OK;
}

// PC::commit(ck, polys) with the caller's host slices (prover.rs:213 passes 4 polynomials, :579 and :606 seven):
// polynomial k+1 is uploaded (pinned staging ring, copy stream) while polynomial k's MSM runs.
int zk_kzg_commit_batch(zk_ctx* c, zk_srs* s, uint32_t n_polys, const uint64_t* const* coeffs_mont, const size_t* lens, uint64_t* out_xy,
                        uint8_t* out_inf) {
    if (!c || !s || s->device != c->device || (n_polys && (!coeffs_mont || !lens || !out_xy))) return ZK_ERR_BAD_ARG;
    if (n_polys > 16) return ZK_ERR_BAD_ARG;
    for (uint32_t k = 0; k < n_polys; ++k)
        if (lens[k] && !coeffs_mont[k]) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    int rc = ensure_copy_stream(c);
    if (rc) return rc;
    // residency cache: polynomials this ctx produced (zk_ntt outputs) or uploaded before are used where they lie; the others go up
    // into a fresh entry (resident from then on) or, where the cache cannot take them, into the job's staging buffer.  Polynomial k is
    // digested right before job k is queued, i.e. while the GPU runs job k - 1: like the uploads, the digests hide under the MSMs.
    // Every job gets a stable input pointer up front (a fresh entry or its staging buffer); a hit copies nothing into it and points the
    // job at the resident copy instead -- d_in[k] is read when job k is queued, after up(k) has run.
    const void* d_in[16];
    bool cached[16] = {false};
    ResCall rcall(c);
    for (uint32_t k = 0; k < n_polys; ++k) {
        if (lens[k] > s->n) return ZK_ERR_BAD_ARG;
        cached[k] = res_wants(c, lens[k] * 32);
        if ((rc = c->mb[k].upload.ensure((lens[k] ? lens[k] : 1) * 32))) return rc;
        d_in[k] = c->mb[k].upload.p;
    }
    BeforeJob up = [&](uint32_t k) -> int {
        if (lens[k] == 0) return ZK_OK;
        if (cached[k]) {
            const void* hp = coeffs_mont[k];
            const size_t hb = lens[k] * 32;
            uint64_t dig[1][4];
            host_digest256_multi(c->pool.get(), &hp, &hb, 1, RES_SEED, dig);
            if (zk_ctx::ResEntry* e = res_find(c, hb, dig[0])) {
                if (res_verify_hit(c, e, coeffs_mont[k])) {
                    d_in[k] = e->buf.p;                // resident: nothing crosses PCIe, nothing to wait for
                    return ZK_OK;
                }
            }
            if (zk_ctx::ResEntry* f = res_new(c, hb)) {
                memcpy(f->dig, dig[0], 32);
                f->valid = true;                       // its bytes go up in stream order before anything reads them
                d_in[k] = f->buf.p;
            }
        }
        int r = zk_h2d(c, const_cast<void*>(d_in[k]), coeffs_mont[k], lens[k] * 32, c->copy_stream);
        if (r) return r;
        ZK_HIP_TRY(hipEventRecord(c->ev_up[k], c->copy_stream));
        ZK_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_up[k], 0));
        return ZK_OK;
    };
    if (c->commit_cache_on) {   // the digests need every input on the device first
        for (uint32_t k = 0; k < n_polys; ++k)
            if ((rc = up(k))) return rc;
        return rcall.done(batch_cached_locked(c, s, n_polys, d_in, lens, nullptr, out_xy, out_inf));
    }
    return rcall.done(batch_locked(c, s, n_polys, d_in, lens, nullptr, nullptr, out_xy, out_inf, &up));
}

int zk_kzg_commit(zk_ctx* c, zk_srs* s, const uint64_t* coeffs_mont, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    return zk_kzg_commit_batch(c, s, 1, &coeffs_mont, &n, out_xy, out_inf);
}

int zk_kzg_open_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* z_mont,
                    const uint64_t* challenge_mont, uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || !s || s->device != c->device || !out_xy || !z_mont || !challenge_mont || (n_polys && (!d_polys || !lens))) return ZK_ERR_BAD_ARG;
    void* d_w = nullptr;
    size_t wlen = 0;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    SrsRead rl(s->mu);
    {
        int rc = kzg_open_prepare_dev(c, s->curve, n_polys, d_polys, lens, z_mont, challenge_mont, &d_w, &wlen);
        if (rc) return rc;
    }
    if (wlen > s->n) return ZK_ERR_BAD_ARG;
    uint64_t xyz[18];
    int rc = msm_partial_locked(c, s, 0, d_w, wlen, xyz);
    if (rc) return rc;
    return finish_point(s->curve, xyz, out_xy, out_inf);
}

// PC::open with the caller's host slices: the polynomials are uploaded (staged), everything else as zk_kzg_open_dev
int zk_kzg_open(zk_ctx* c, zk_srs* s, uint32_t n_polys, const uint64_t* const* polys_mont, const size_t* lens, const uint64_t* z_mont,
                const uint64_t* challenge_mont, uint64_t* out_xy, uint8_t* out_inf) {
    if (!c || !s || s->device != c->device || !out_xy || !z_mont || !challenge_mont || (n_polys && (!polys_mont || !lens))) return ZK_ERR_BAD_ARG;
    if (n_polys > 16) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (round_open(c)) return ZK_ERR_PENDING;
    const void* d_in[16];
    ResInput ri[16];
    for (uint32_t k = 0; k < n_polys; ++k)
        if (lens[k] && !polys_mont[k]) return ZK_ERR_BAD_ARG;
    ResCall rcall(c);
    if (c->res_on) {      // the eleven / seven polynomials of an opening were all transformed or committed before (prover.rs:582-618)
        const void* hp[16];
        size_t hb[16];
        for (uint32_t k = 0; k < n_polys; ++k) {
            hp[k] = polys_mont[k];
            hb[k] = lens[k] * 32;
        }
        res_resolve(c, n_polys, hp, hb, ri);
    }
    for (uint32_t k = 0; k < n_polys; ++k) {
        int rc;
        if (ri[k].d_ptr) {
            d_in[k] = ri[k].d_ptr;
        } else {
            if ((rc = c->mb[k].upload.ensure((lens[k] ? lens[k] : 1) * 32))) return rc;
            d_in[k] = c->mb[k].upload.p;
        }
        if (ri[k].upload && (rc = zk_h2d(c, const_cast<void*>(d_in[k]), polys_mont[k], lens[k] * 32, c->stream))) return rc;
    }
    return rcall.done(zk_kzg_open_dev(c, s, n_polys, d_in, lens, z_mont, challenge_mont, out_xy, out_inf));
}

int zk_kzg_witness_dev(zk_ctx* c, int curve_id, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* z_mont,
                       const uint64_t* challenge_mont, void* d_out, size_t* out_len) {
    if (!c || !z_mont || !challenge_mont || !out_len || (n_polys && (!d_polys || !lens))) return ZK_ERR_BAD_ARG;
    void* d_w = nullptr;
    size_t wlen = 0;
    Guard g(c);
    int rc = kzg_open_prepare_dev(c, curve_id, n_polys, d_polys, lens, z_mont, challenge_mont, &d_w, &wlen);
    if (rc) return rc;
    *out_len = wlen;
    if (wlen) {
        if (!d_out) return ZK_ERR_BAD_ARG;
        ZK_HIP_TRY(hipMemcpyAsync(d_out, d_w, wlen * 32, hipMemcpyDeviceToDevice, c->stream));
    }
    return ZK_OK;
}

int zk_lookup_query_dev(zk_ctx* c, int curve_id, size_t n, const void* d_q_lookup, size_t q_len, const void* const d_wires[4], const uint64_t* zeta_mont,
                        const void* d_table_compressed, void* d_out) {
    if (!c || !zeta_mont) return ZK_ERR_BAD_ARG;
    if (n && (!d_wires || !d_wires[0] || !d_wires[1] || !d_wires[2] || !d_wires[3] || !d_table_compressed || !d_out || (q_len && !d_q_lookup)))
        return ZK_ERR_BAD_ARG;
    Guard g(c);
    return lookup_query_dev(c, curve_id, n, d_q_lookup, q_len, d_wires, zeta_mont, d_table_compressed, d_out);
}

int zk_lookup_combine_split_dev(zk_ctx* c, int curve_id, const void* d_t, size_t n_t, const void* d_f, size_t n_f, void* d_h1, void* d_h2,
                                size_t* len_h1, size_t* len_h2) {
    if (!c || !len_h1 || !len_h2 || (curve_id != ZK_CURVE_BLS12_381 && curve_id != ZK_CURVE_BN254)) return ZK_ERR_BAD_ARG;
    if ((n_t && !d_t) || (n_f && !d_f) || ((n_t + n_f) && (!d_h1 || !d_h2))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return lookup_combine_split_dev(c, d_t, n_t, d_f, n_f, d_h1, d_h2, len_h1, len_h2);
}

int zk_poly_evaluate_dev(zk_ctx* c, int curve_id, uint32_t n_polys, const void* const* d_polys, const size_t* lens, const uint64_t* points_mont,
                         uint64_t* out_mont) {
    if (!c || (n_polys && (!d_polys || !lens || !points_mont || !out_mont))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return poly_evaluate_dev(c, curve_id, n_polys, d_polys, lens, points_mont, out_mont);
}

int zk_poly_lincomb_dev(zk_ctx* c, int curve_id, uint32_t n_terms, const void* const* d_polys, const size_t* lens, const uint64_t* coeffs_mont,
                        void* d_out, size_t out_len) {
    if (!c || (n_terms && (!d_polys || !lens || !coeffs_mont)) || (out_len && !d_out)) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return poly_lincomb_dev(c, curve_id, n_terms, d_polys, lens, coeffs_mont, d_out, out_len);
}

int zk_io_stats(zk_ctx* c, uint64_t* h2d_bytes, uint64_t* d2h_bytes, int reset) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (h2d_bytes) *h2d_bytes = c->h2d_bytes;
    if (d2h_bytes) *d2h_bytes = c->d2h_bytes;
    if (reset) c->h2d_bytes = c->d2h_bytes = 0;
    return ZK_OK;
}

int zk_ctx_set_staging(zk_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 1) return ZK_ERR_BAD_ARG;
    Guard g(c);
    c->staging_mode = mode;
    return ZK_OK;
}

// ------------------------------------------------------------------------------------- utilities
int zk_g1_fixed_base_batch_dev(zk_ctx* c, int curve_id, const void* d_scalars, size_t n, void* d_out_xy) {
    if (!c || (n && (!d_scalars || !d_out_xy))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return msm_fixed_base_dev(c, curve_id, d_scalars, n, d_out_xy);
}

int zk_dev_alloc(zk_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return ZK_ERR_BAD_ARG;
    Guard g(c);
    *d_ptr = nullptr;
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) return ZK_ERR_OOM;
    return ZK_OK;
}
int zk_dev_free(zk_ctx* c, void* d_ptr) {
    if (!c) return ZK_ERR_BAD_ARG;
    Guard g(c);
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    if (d_ptr) ZK_HIP_TRY(hipFree(d_ptr));
    return ZK_OK;
}
int zk_dev_upload(zk_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !h_src))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (bytes) ZK_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    return ZK_OK;
}
int zk_dev_download(zk_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    if (!c || (bytes && (!h_dst || !d_src))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (bytes) ZK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP_TRY(hipStreamSynchronize(c->stream));
    return ZK_OK;
}

int zk_dev_copy(zk_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
    if (!c || (bytes && (!d_dst || !d_src))) return ZK_ERR_BAD_ARG;
    Guard g(c);
    if (bytes) ZK_HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));      // queued, not waited for
    return ZK_OK;
}

// ------------------------------------------------------------------------------ N2: grand products
int zk_perm_product_dev(zk_ctx* c, int curve_id, uint32_t log_n, const void* const* d_wires, const void* const* d_sigmas,
                        const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, uint64_t* last_mont) {
    if (!c || !d_wires || !d_sigmas || !beta_mont || !gamma_mont || !d_out) return ZK_ERR_BAD_ARG;
    if (curve_id != ZK_CURVE_BLS12_381 && curve_id != ZK_CURVE_BN254) return ZK_ERR_BAD_ARG;
    if (log_n > 32) return ZK_ERR_DOMAIN_TOO_LARGE;
    for (int k = 0; k < 4; ++k)
        if (!d_wires[k] || !d_sigmas[k]) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return perm_product_dev(c, curve_id, log_n, d_wires, d_sigmas, beta_mont, gamma_mont, d_out, last_mont);
}

int zk_lookup_product_dev(zk_ctx* c, int curve_id, size_t n, const void* d_f, const void* d_t, const void* d_h1, const void* d_h2,
                          const uint64_t* delta_mont, const uint64_t* epsilon_mont, void* d_out, uint64_t* last_mont) {
    if (!c || !n || !d_f || !d_t || !d_h1 || !d_h2 || !delta_mont || !epsilon_mont || !d_out) return ZK_ERR_BAD_ARG;
    if (curve_id != ZK_CURVE_BLS12_381 && curve_id != ZK_CURVE_BN254) return ZK_ERR_BAD_ARG;
    Guard g(c);
    return lookup_product_dev(c, curve_id, n, d_f, d_t, d_h1, d_h2, delta_mont, epsilon_mont, d_out, last_mont);
}

// ------------------------------------------------------------------------------ N1: quotient
int zk_quotient_evals_dev(zk_ctx* c, int curve_id, uint32_t log_n, const zk_quotient_args* args, void* d_out) {
    if (!c || !args || !d_out) return ZK_ERR_BAD_ARG;
    if (curve_id != ZK_CURVE_BLS12_381 && curve_id != ZK_CURVE_BN254) return ZK_ERR_BAD_ARG;
    if (log_n > 30) return ZK_ERR_DOMAIN_TOO_LARGE;
    Guard g(c);
    return quotient_evals_dev(c, curve_id, log_n, args, d_out);
}

// ------------------------------------------------------------------------------ device self-test
int zk_selftest_quad_dev(zk_ctx* c, int curve_id, uint32_t n_quads, uint32_t* mismatches, uint32_t* case_mask) {
    if (!c || !mismatches) return ZK_ERR_BAD_ARG;
    Guard g(c);
    uint32_t out[2] = {0, 0};
    int rc = quad_selftest_dev(c, curve_id, n_quads, out);
    if (rc) return rc;
    *mismatches = out[0];
    if (case_mask) *case_mask = out[1];
    return ZK_OK;
}

}  // extern "C"
