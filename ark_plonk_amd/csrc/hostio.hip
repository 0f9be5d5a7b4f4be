// Host <-> device transfers of the host-pointer entry points (zk_ntt, zk_kzg_commit(_batch), zk_kzg_open,
// zk_msm_g1_srs: the calls a Rust shim binds) and the content digests of the SRS / commitment caches.
//
// The reference hands over ordinary heap slices (`domain.ifft(&w_l_scalar)` prover.rs:196-203, `PC::commit(ck, polys)`
// prover.rs:213), i.e. pageable memory.  Two ways to move it (zk_ctx_set_staging):
//   0 (default)  hipMemcpyAsync straight from / to the caller's buffer.  Measured on the MI355X hosts (tools/pcie_probe.py,
//                profiles/r02/r02_notes.md): 56 GB/s in both directions, the same as pinned memory -- the runtime's own staging
//                keeps the link busy.
//   1            an explicit ring of pinned 8 MiB slots per ctx, the caller's bytes copied into a slot by the ctx's host pool
//                and sent by DMA while the next slot is being filled: 49 GB/s there, kept for hosts where mode 0 is slow.
// Pinned or hipHostRegister-ed caller buffers are detected and always sent directly.
#include "ctx.h"
#include <sys/random.h>

#include <chrono>
#include <cstdio>
#include <mutex>

namespace {

bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary heap pointer: not an error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int stage_setup(zk_ctx* c) {
    for (int i = 0; i < zk_ctx::STAGE_SLOTS; ++i) {
        if (!c->stage_pin[i]) {
            if (hipHostMalloc(&c->stage_pin[i], zk_ctx::STAGE_BYTES, hipHostMallocDefault) != hipSuccess) return ZK_ERR_OOM;
            ZK_HIP_TRY(hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming));
            c->stage_busy[i] = false;
        }
    }
    return ZK_OK;
}

// copy with the ctx's pool: pieces of 1 MiB
void pooled_memcpy(zk_ctx* c, void* dst, const void* src, size_t bytes) {
    constexpr size_t PIECE = (size_t)1 << 20;
    const uint32_t pieces = (uint32_t)((bytes + PIECE - 1) / PIECE);
    if (pieces <= 1 || !c->pool) {
        memcpy(dst, src, bytes);
        return;
    }
    c->pool->run(pieces, [&](uint32_t k) {
        const size_t off = (size_t)k * PIECE;
        const size_t len = bytes - off < PIECE ? bytes - off : PIECE;
        memcpy((char*)dst + off, (const char*)src + off, len);
    });
}

int take_slot(zk_ctx* c, int* slot) {
    const int s = c->stage_next;
    c->stage_next = (s + 1) % zk_ctx::STAGE_SLOTS;
    if (c->stage_busy[s]) {
        ZK_HIP_TRY(hipEventSynchronize(c->stage_ev[s]));
        c->stage_busy[s] = false;
    }
    *slot = s;
    return ZK_OK;
}

// ---- digests --------------------------------------------------------------------------------------------
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull,
                   P5 = 0x27D4EB2F165667C5ull;

ZK_HD uint64_t fmix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// one block: four xxhash-style lanes over 32-byte stripes.  Lane k reads word k of every stripe, so the lanes are folded into each
// other every eighth stripe (and twice more at the end): a difference confined to one 64-bit column of the vector -- small field
// elements in canonical form differ in their low limb only -- reaches all 256 bits of state within 256 bytes instead of being carried
// by one lane's 64 bits to the final mix.  The lanes start from the 256-bit process key (zk_process_key): the digests
// address caches, and a fixed, public mixing function would let a client who picks the cached bytes search for two inputs with one
// digest offline.
void block_digest(const uint8_t* p, size_t bytes, uint64_t seed, const uint64_t key[4], uint64_t out[4]) {
    uint64_t a[4] = {key[0] ^ (seed + P1 + P2), key[1] ^ (seed + P2), key[2] ^ seed, key[3] ^ (seed - P1)};
    auto fold = [&a]() {
        const uint64_t t0 = a[0], t1 = a[1], t2 = a[2], t3 = a[3];
        a[0] = t0 + (rotl64(t1, 13) ^ rotl64(t2, 29) ^ rotl64(t3, 47));
        a[1] = t1 + (rotl64(t2, 13) ^ rotl64(t3, 29) ^ rotl64(t0, 47));
        a[2] = t2 + (rotl64(t3, 13) ^ rotl64(t0, 29) ^ rotl64(t1, 47));
        a[3] = t3 + (rotl64(t0, 13) ^ rotl64(t1, 29) ^ rotl64(t2, 47));
    };
    size_t i = 0;
    for (; i + 256 <= bytes; i += 256) {
        for (size_t j = 0; j < 256; j += 32) {
            uint64_t w[4];
            memcpy(w, p + i + j, 32);
            for (int k = 0; k < 4; ++k) a[k] = rotl64(a[k] + w[k] * P2, 31) * P1;
        }
        fold();
    }
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) a[k] = rotl64(a[k] + w[k] * P2, 31) * P1;
    }
    if (i < bytes) {
        uint64_t w[4] = {0, 0, 0, 0};
        memcpy(w, p + i, bytes - i);
        for (int k = 0; k < 4; ++k) a[k] = rotl64(a[k] + w[k] * P2, 31) * P1;
    }
    fold();
    const uint64_t len = (uint64_t)bytes;
    for (int r = 0; r < 2; ++r)
        for (int k = 0; k < 4; ++k) a[k] = fmix64(a[k] ^ rotl64(a[(k + 1) & 3], 17) ^ (a[(k + 2) & 3] * P3) ^ (len + P5 * (uint64_t)(k + 1)));
    for (int k = 0; k < 4; ++k) out[k] = a[k];
}

// per element (index i, four 64-bit words): four independently seeded chained mixes; the vector digest is their sum
// over i (a multiset hash: position enters through i, so the launch geometry and reduction order are free)
struct DigJobs {
    const void* p[16];
    uint64_t n[16];
    uint64_t key[4];      // the ctx's secret: the per-element mixes are keyed, so equal sums cannot be searched for without it
};
__global__ void __launch_bounds__(256) digest_kernel(DigJobs jobs, uint64_t* out) {
    const uint32_t job = blockIdx.y;
    const uint64_t* v = (const uint64_t*)jobs.p[job];
    const uint64_t n = jobs.n[job];
    uint64_t acc[4] = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 lo = reinterpret_cast<const ulonglong2*>(v)[2 * i], hi = reinterpret_cast<const ulonglong2*>(v)[2 * i + 1];
        const uint64_t w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t h = fmix64((i * 0x9E3779B97F4A7C15ull + 0xD6E8FEB86659FD93ull * (uint64_t)(k + 1)) ^ jobs.key[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) h = fmix64(h ^ (w[(j + k) & 3] + 0x9E3779B185EBCA87ull * (uint64_t)(j + 1)));
            acc[k] += fmix64(h + jobs.key[(k + 1) & 3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t x = acc[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_down(x, d, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd((unsigned long long*)&out[4 * job + k], (unsigned long long)x);
    }
}

}  // namespace

// 256 bits drawn once per process from the operating system (getrandom, else /dev/urandom).  Digests are only ever compared inside
// the process that computed them, so the key never leaves it.  Returns whether the operating system delivered: without it the key
// falls back to address-space layout + clock -- digests still work as checksums of this process, but the caches whose hits REPLACE a
// computation (commitment cache, residency cache, SRS registry) refuse to switch on (ZK_ERR_UNSUPPORTED / no sharing).
static std::mutex g_key_mu;
static bool g_key_done = false, g_key_ok = false;
static uint64_t g_key[4];
static std::string g_entropy_path = "/dev/urandom";
static bool g_use_getrandom = true;

bool zk_process_key(uint64_t out[4]) {
    std::lock_guard<std::mutex> lk(g_key_mu);
    if (!g_key_done) {
        bool ok = false;
        if (g_use_getrandom) ok = getrandom(g_key, sizeof g_key, 0) == (ssize_t)sizeof g_key;
        if (!ok) {
            if (FILE* f = fopen(g_entropy_path.c_str(), "rb")) {
                ok = fread(g_key, 1, sizeof g_key, f) == sizeof g_key;
                fclose(f);
            }
        }
        if (!ok) {
            uint64_t x = (uint64_t)(uintptr_t)&g_key ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
            for (int k = 0; k < 4; ++k) g_key[k] = x = fmix64(x + P1 * (uint64_t)(k + 1));
        }
        g_key_ok = ok;
        g_key_done = true;
    }
    memcpy(out, g_key, sizeof g_key);
    return g_key_ok;
}

// test hook (tests/sanitize/host_stress.cpp; not part of the C ABI): draw the key again, from `path` only
void zk_process_key_reset_for_tests(const char* path) {
    std::lock_guard<std::mutex> lk(g_key_mu);
    g_key_done = false;
    g_use_getrandom = path == nullptr;
    g_entropy_path = path ? path : "/dev/urandom";
}

void host_digest256(const void* p, size_t bytes, uint64_t seed, uint64_t out[4]) {
    constexpr size_t BLOCK = (size_t)1 << 20;
    const uint32_t nblk = (uint32_t)((bytes + BLOCK - 1) / BLOCK);
    uint64_t key[4];
    zk_process_key(key);
    if (nblk <= 1) {
        block_digest((const uint8_t*)p, bytes, seed, key, out);
        return;
    }
    std::vector<uint64_t> parts((size_t)nblk * 4);
    host_parallel_for(nblk, [&](uint32_t k) {
        const size_t off = (size_t)k * BLOCK;
        const size_t len = bytes - off < BLOCK ? bytes - off : BLOCK;
        block_digest((const uint8_t*)p + off, len, seed ^ (P4 * (uint64_t)(k + 1)), key, &parts[(size_t)k * 4]);
    });
    block_digest((const uint8_t*)parts.data(), parts.size() * 8, seed ^ (uint64_t)bytes, key, out);
}

void host_digest256_multi(HostPool* pool, const void* const* ptrs, const size_t* bytes, uint32_t n, uint64_t seed, uint64_t (*out)[4]) {
    constexpr size_t BLOCK = (size_t)1 << 20;
    uint64_t key[4];
    zk_process_key(key);
    std::vector<uint32_t> first(n + 1, 0);
    for (uint32_t k = 0; k < n; ++k) first[k + 1] = first[k] + (uint32_t)((bytes[k] + BLOCK - 1) / BLOCK);
    std::vector<uint64_t> parts((size_t)first[n] * 4);
    auto item = [&](uint32_t i) {
        uint32_t k = 0;
        while (first[k + 1] <= i) ++k;                       // n <= 16
        const uint32_t b = i - first[k];
        const size_t off = (size_t)b * BLOCK;
        const size_t len = bytes[k] - off < BLOCK ? bytes[k] - off : BLOCK;
        block_digest((const uint8_t*)ptrs[k] + off, len, seed ^ (P4 * (uint64_t)(b + 1)), key, &parts[(size_t)i * 4]);
    };
    if (pool) pool->run(first[n], item);
    else host_parallel_for(first[n], item);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t nb = first[k + 1] - first[k];
        if (nb == 0) {
            block_digest(nullptr, 0, seed, key, out[k]);
            continue;
        }
        block_digest((const uint8_t*)&parts[(size_t)first[k] * 4], (size_t)nb * 32, seed ^ (uint64_t)bytes[k], key, out[k]);
    }
}

int dev_digest256(const void* const* d_ptrs, const size_t* d_lens, uint32_t n_jobs, uint64_t* d_out, hipStream_t st, const uint64_t key[4]) {
    if (n_jobs == 0) return ZK_OK;
    if (n_jobs > 16) return ZK_ERR_BAD_ARG;
    DigJobs jobs;
    memcpy(jobs.key, key, sizeof jobs.key);
    for (uint32_t k = 0; k < 16; ++k) {
        jobs.p[k] = k < n_jobs ? d_ptrs[k] : nullptr;
        jobs.n[k] = k < n_jobs ? d_lens[k] : 0;
    }
    ZK_HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)n_jobs * 32, st));
    hipLaunchKernelGGL(digest_kernel, dim3(512, n_jobs), dim3(256), 0, st, jobs, d_out);
    ZK_HIP_TRY(hipGetLastError());
    return ZK_OK;
}

int zk_h2d(zk_ctx* c, void* d_dst, const void* h_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return ZK_OK;
    c->h2d_bytes += bytes;
    if (c->staging_mode == 0 || is_pinned(h_src)) {
        ZK_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
        return ZK_OK;
    }
    int rc = stage_setup(c);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += zk_ctx::STAGE_BYTES) {
        const size_t len = bytes - off < zk_ctx::STAGE_BYTES ? bytes - off : zk_ctx::STAGE_BYTES;
        int s;
        if ((rc = take_slot(c, &s))) return rc;
        pooled_memcpy(c, c->stage_pin[s], (const char*)h_src + off, len);
        ZK_HIP_TRY(hipMemcpyAsync((char*)d_dst + off, c->stage_pin[s], len, hipMemcpyHostToDevice, st));
        ZK_HIP_TRY(hipEventRecord(c->stage_ev[s], st));
        c->stage_busy[s] = true;
    }
    return ZK_OK;
}

int zk_d2h(zk_ctx* c, void* h_dst, const void* d_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return ZK_OK;
    c->d2h_bytes += bytes;
    if (c->staging_mode == 0 || is_pinned(h_dst)) {
        ZK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        ZK_HIP_TRY(hipStreamSynchronize(st));
        return ZK_OK;
    }
    int rc = stage_setup(c);
    if (rc) return rc;
    // keep up to STAGE_SLOTS DMA chunks in flight; unload the oldest into the caller's buffer while the rest arrive
    struct Pending {
        int slot;
        size_t off, len;
    };
    Pending ring[zk_ctx::STAGE_SLOTS];
    int head = 0, count = 0;
    auto drain_one = [&]() -> int {
        Pending& p = ring[head];
        ZK_HIP_TRY(hipEventSynchronize(c->stage_ev[p.slot]));
        c->stage_busy[p.slot] = false;
        pooled_memcpy(c, (char*)h_dst + p.off, c->stage_pin[p.slot], p.len);
        head = (head + 1) % zk_ctx::STAGE_SLOTS;
        --count;
        return ZK_OK;
    };
    for (size_t off = 0; off < bytes; off += zk_ctx::STAGE_BYTES) {
        const size_t len = bytes - off < zk_ctx::STAGE_BYTES ? bytes - off : zk_ctx::STAGE_BYTES;
        if (count == zk_ctx::STAGE_SLOTS && (rc = drain_one())) return rc;
        int s;
        if ((rc = take_slot(c, &s))) return rc;
        ZK_HIP_TRY(hipMemcpyAsync(c->stage_pin[s], (const char*)d_src + off, len, hipMemcpyDeviceToHost, st));
        ZK_HIP_TRY(hipEventRecord(c->stage_ev[s], st));
        c->stage_busy[s] = true;
        ring[(head + count) % zk_ctx::STAGE_SLOTS] = Pending{s, off, len};
        ++count;
    }
    while (count)
        if ((rc = drain_one())) return rc;
    return ZK_OK;
}

void zk_io_release(zk_ctx* c) {
    for (int i = 0; i < zk_ctx::STAGE_SLOTS; ++i) {
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
        if (c->stage_pin[i]) (void)hipHostFree(c->stage_pin[i]);
        c->stage_ev[i] = nullptr;
        c->stage_pin[i] = nullptr;
    }
    for (int i = 0; i < 16; ++i)
        if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    c->copy_stream = nullptr;
    c->digest_dev.release();
}
