// TEST INFRASTRUCTURE (tests/test_sanitize.py): the host side of the whole library under a sanitizer, without a GPU.
// Linked with fake_hip.cpp (device memory = zeroed host memory, kernel launches do nothing), so every device result is the
// all-zero pattern: the point at infinity / the zero digest.  What is exercised for real is everything the sanitizers can see:
//   * the device-scoped, content-addressed, reference-counted SRS registry (zk_srs_register from several threads with the same and
//     with different bytes, zk_srs_retain / zk_srs_free, eviction through zk_srs_cache_config, zk_srs_precompute under readers);
//   * threads sharing ONE zk_ctx and threads with a ctx each over one SRS -- the pattern of tests/test_concurrency_gpu.py --
//     through the host-pointer entry points (staging buffers, pinned ring, copy stream bookkeeping), blocking batches,
//     deferred rounds (begin / open_begin / reduce / end / abort, the ZK_ERR_PENDING refusals), the commitment cache (LRU list);
//   * the persistent host pool (HostPool::run from several ctxs at once) and the host tails of a round;
//   * with REAL points: zk_g1_sum_partials(_batch) -- Jacobian -> XYZZ, host additions incl. doubling and cancellation, Fermat
//     inversion, affine normalisation -- on the group generator decoded by wire.hip's own deserialiser.
// Exit code 0 and "host_stress ok" = every call returned what it must; the sanitizer's own report fails the test otherwise.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "ark_plonk_amd.h"

extern "C" long fake_hip_launches();
extern "C" long fake_hip_live_allocations();
extern "C" void fake_hip_set_memory(size_t budget, size_t report_extra);
extern "C" size_t fake_hip_live_bytes();
void zk_process_key_reset_for_tests(const char* path);      // csrc/hostio.hip: test hook, not part of the C ABI

#define REQUIRE(cond)                                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            fprintf(stderr, "host_stress: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static const int CURVE = ZK_CURVE_BLS12_381;
static const size_t L = 6;      // u64 limbs of Fq

// the standard compressed encodings of the BLS12-381 G1 generator in ark-serialize's form (x little-endian, flags in the last byte):
// decoded by the library itself, so the harness needs no field arithmetic of its own
static int generator(uint64_t* xy, bool negate) {
    static const uint8_t gx_be[48] = {0x17, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
                                      0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
                                      0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    uint8_t enc[48];
    for (int i = 0; i < 48; ++i) enc[i] = gx_be[47 - i];
    uint8_t inf = 0;
    // the sign bit selects one of the two points with this x; try both and keep the one asked for (y differs, x is equal)
    uint64_t a[12], b[12];
    enc[47] &= 0x3f;
    if (zk_g1_deserialize_compressed(CURVE, enc, a, &inf) != ZK_OK || inf) return 1;
    enc[47] |= 0x80;
    if (zk_g1_deserialize_compressed(CURVE, enc, b, &inf) != ZK_OK || inf) return 1;
    if (memcmp(a, b, 48) != 0 || memcmp(a + 6, b + 6, 48) == 0) return 1;
    memcpy(xy, negate ? b : a, 96);
    return 0;
}

static int real_point_arithmetic() {
    uint64_t g[12], ng[12], one[6], out[12], two_g[12], again[12];
    uint8_t inf = 0;
    REQUIRE(generator(g, false) == 0 && generator(ng, true) == 0);
    // Montgomery one of Fq: the y of GroupAffine::zero() = (0, 1), which is what an empty sum returns
    REQUIRE(zk_g1_sum_partials(CURVE, nullptr, 0, out, &inf) == ZK_OK && inf == 1);
    memcpy(one, out + 6, 48);
    auto jac = [&](const uint64_t* xy, uint64_t* p) {
        memcpy(p, xy, 96);
        memcpy(p + 12, one, 48);
    };
    uint64_t parts[4 * 18];
    jac(g, parts);
    REQUIRE(zk_g1_sum_partials(CURVE, parts, 1, out, &inf) == ZK_OK && !inf && memcmp(out, g, 96) == 0);          // G
    jac(g, parts + 18);
    REQUIRE(zk_g1_sum_partials(CURVE, parts, 2, two_g, &inf) == ZK_OK && !inf && memcmp(two_g, g, 96) != 0);      // G + G: the doubling branch
    jac(ng, parts + 18);
    REQUIRE(zk_g1_sum_partials(CURVE, parts, 2, out, &inf) == ZK_OK && inf == 1);                                 // G - G: infinity
    jac(two_g, parts);
    jac(ng, parts + 18);
    REQUIRE(zk_g1_sum_partials(CURVE, parts, 2, out, &inf) == ZK_OK && !inf && memcmp(out, g, 96) == 0);          // 2G - G = G
    // batch form: 3 ranks x 2 jobs, rank-major; job 0 = G + G - G, job 1 = 2G - G + infinity
    uint64_t batch[3 * 2 * 18];
    memset(batch, 0, sizeof batch);
    jac(g, batch + 0 * 18);
    jac(two_g, batch + 1 * 18);
    jac(g, batch + 2 * 18);
    jac(ng, batch + 3 * 18);
    jac(ng, batch + 4 * 18);
    memcpy(batch + 5 * 18, one, 48);                 // (1, 1, 0): Jacobian zero as arkworks writes it
    memcpy(batch + 5 * 18 + 6, one, 48);
    uint64_t bout[2 * 12];
    uint8_t binf[2] = {9, 9};
    REQUIRE(zk_g1_sum_partials_batch(CURVE, batch, 3, 2, bout, binf) == ZK_OK);
    REQUIRE(!binf[0] && !binf[1] && memcmp(bout, g, 96) == 0 && memcmp(bout + 12, g, 96) == 0);
    // uncompressed / compressed round trips of the results through wire.hip
    uint8_t enc[96];
    REQUIRE(zk_g1_serialize_compressed(CURVE, two_g, 0, enc) == ZK_OK);
    REQUIRE(zk_g1_deserialize_compressed(CURVE, enc, again, &inf) == ZK_OK && !inf && memcmp(again, two_g, 96) == 0);
    return 0;
}

static std::atomic<int> g_fail{0};
#define TCHECK(cond)                                                                          \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            fprintf(stderr, "host_stress thread: %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
            g_fail.fetch_add(1);                                                              \
            return;                                                                           \
        }                                                                                     \
    } while (0)

int main() {
    REQUIRE(real_point_arithmetic() == 0);

    const size_t n = 1 << 13;           // the smallest size that takes the window-table path (ZK_PRE_MIN_N)
    std::vector<uint64_t> srs_a(n * 2 * L), srs_b(n * 2 * L);
    for (size_t i = 0; i < srs_a.size(); ++i) {
        srs_a[i] = 0x1000 + i;          // bytes only: the fake device never looks at them, the registry digests them
        srs_b[i] = 0x9000 + 3 * i;
    }
    std::vector<std::vector<uint64_t>> polys(4, std::vector<uint64_t>(n * 4));
    for (size_t k = 0; k < polys.size(); ++k)
        for (size_t i = 0; i < polys[k].size(); ++i) polys[k][i] = (k + 1) * 1000003ull + i;

    // ---- 1. several threads register the same and different SRS bytes at once, precompute, commit through ONE shared ctx
    zk_ctx* shared = nullptr;
    REQUIRE(zk_ctx_create(0, &shared) == ZK_OK);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 6; ++t)
            th.emplace_back([&, t] {
                for (int it = 0; it < 8; ++it) {
                    zk_srs* s = nullptr;
                    TCHECK(zk_srs_register(shared, CURVE, (t & 1) ? srs_b.data() : srs_a.data(), nullptr, n, &s) == ZK_OK && s);
                    TCHECK(zk_srs_len(s) == n);
                    TCHECK(zk_srs_precompute(shared, s) == ZK_OK);
                    uint32_t c = 0, w = 0;
                    TCHECK(zk_srs_table_info(s, &c, &w) == ZK_OK && c == 16 && w == 16);
                    uint64_t xy[4 * 12];
                    uint8_t inf[4] = {0, 0, 0, 0};
                    const uint64_t* ptrs[4] = {polys[0].data(), polys[1].data(), polys[2].data(), polys[3].data()};
                    size_t lens[4] = {n, n - 1, n, 100};          // table path and a short vector (per-window path)
                    TCHECK(zk_kzg_commit_batch(shared, s, 4, ptrs, lens, xy, inf) == ZK_OK);
                    TCHECK(inf[0] && inf[1] && inf[2] && inf[3]);      // all-zero device results: the point at infinity
                    uint64_t z[4] = {5, 0, 0, 0}, ch[4] = {7, 0, 0, 0};
                    TCHECK(zk_kzg_open(shared, s, 3, ptrs, lens, z, ch, xy, inf) == ZK_OK);
                    std::vector<uint64_t> v(polys[t % 4]);
                    TCHECK(zk_ntt(shared, CURVE, ZK_NTT_IFFT, 13, v.data(), n, v.data()) == ZK_OK);
                    if (it & 1) TCHECK(zk_srs_retain(s) == ZK_OK);
                    zk_srs_free(s);
                    if (it & 1) zk_srs_free(s);
                }
            });
        for (auto& t : th) t.join();
        REQUIRE(g_fail.load() == 0);
        uint64_t hits = 0, misses = 0, entries = 0, bytes = 0;
        REQUIRE(zk_srs_cache_stats(&hits, &misses, &entries, &bytes) == ZK_OK);
        REQUIRE(misses == 2 && hits == 6 * 8 - 2 && entries == 2);
    }

    // ---- 2. a ctx per thread over ONE SRS: deferred rounds, the refusals while a round is open, abort, the commitment cache
    zk_srs* srs = nullptr;
    REQUIRE(zk_srs_register(shared, CURVE, srs_a.data(), nullptr, n, &srs) == ZK_OK);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                zk_ctx* c = nullptr;
                TCHECK(zk_ctx_create(0, &c) == ZK_OK);
                std::vector<void*> d(4, nullptr);
                for (int k = 0; k < 4; ++k) {
                    TCHECK(zk_dev_alloc(c, n * 32, &d[k]) == ZK_OK);
                    TCHECK(zk_dev_upload(c, d[k], polys[k].data(), n * 32) == ZK_OK);
                }
                const void* in[4] = {d[0], d[1], d[2], d[3]};
                size_t lens[4] = {n, n, n - 1, 64};
                uint64_t xy[16 * 12];
                uint8_t inf[16];
                uint64_t z[4] = {5, 0, 0, 0}, ch[4] = {7, 0, 0, 0};
                for (int it = 0; it < 10; ++it) {
                    TCHECK(zk_kzg_round_begin_dev(c, srs, 2, in, lens, nullptr) == ZK_OK);
                    TCHECK(zk_kzg_round_begin_dev(c, srs, 2, in + 2, lens + 2, nullptr) == ZK_OK);      // a short vector joins the round
                    TCHECK(zk_kzg_open_begin_dev(c, srs, 3, in, lens, z, ch) == ZK_OK);
                    uint32_t pend = 0;
                    TCHECK(zk_kzg_round_pending(c, &pend) == ZK_OK && pend == 5);
                    TCHECK(zk_kzg_commit_dev(c, srs, d[0], n, xy, inf) == ZK_ERR_PENDING);               // blocking calls refuse
                    TCHECK(zk_kzg_round_end(c, 4, xy, inf) == ZK_ERR_BAD_ARG);                           // wrong count: stays open
                    if (it % 3 == 2) {
                        TCHECK(zk_kzg_round_abort(c) == ZK_OK);
                    } else {
                        if (it & 1) TCHECK(zk_kzg_round_reduce(c) == ZK_OK);
                        if (it & 1) TCHECK(zk_kzg_round_begin_dev(c, srs, 1, in, lens, nullptr) == ZK_ERR_PENDING);   // closed to new jobs
                        TCHECK(zk_kzg_round_end(c, 5, xy, inf) == ZK_OK);
                    }
                    TCHECK(zk_kzg_round_pending(c, &pend) == ZK_OK && pend == 0);
                    // the device form of a round's result: 2 VW sums per job on the device, summed over the ranks, combined by the host pool
                    {
                        const size_t wb = zk_winsums_dev_bytes(c, srs), pb = 256;
                        uint32_t geom[4] = {0, 0, 0, 0};
                        TCHECK(wb == 2 * 64 * pb && zk_winsums_geometry(c, srs, geom) == ZK_OK && geom[0] == 16 && geom[2] == 64);
                        void* ws = nullptr;
                        TCHECK(zk_dev_alloc(c, 2 * 4 * wb, &ws) == ZK_OK);
                        TCHECK(zk_kzg_round_begin_dev(c, srs, 4, in, lens, nullptr) == ZK_OK);
                        TCHECK(zk_kzg_round_reduce_winsums_dev(c, ws) == ZK_OK);
                        TCHECK(zk_kzg_round_end(c, 4, xy, inf) == ZK_ERR_PENDING);                // reduced towards the device: the host form refuses
                        TCHECK(zk_kzg_round_end_winsums_dev(c, 4, (char*)ws + 64) == ZK_ERR_PENDING);   // ... and so does another buffer
                        TCHECK(zk_kzg_round_end_winsums_dev(c, 4, ws) == ZK_OK);
                        TCHECK(zk_g1_sum_winsums_dev(c, srs, ws, 2, 4, xy, inf) == ZK_OK && inf[0] && inf[3]);
                        TCHECK(zk_dev_free(c, ws) == ZK_OK);
                    }
                    // tuning options: per ctx, refused while a round is open, toggled here between rounds while a second thread of the
                    // SAME ctx reads them (the ctx lock orders the two; getenv / setenv would have raced)
                    {
                        std::thread reader([&] {
                            for (int r = 0; r < 50; ++r) {
                                int64_t v = -7;
                                TCHECK(zk_ctx_get_option(c, "msm_merge", &v) == ZK_OK && (v == 0 || v == 1));
                                TCHECK(zk_ctx_get_option(c, "pre_vw", &v) == ZK_OK && (v == 0 || v == 32));
                            }
                        });
                        TCHECK(zk_ctx_set_option(c, "msm_merge", it & 1) == ZK_OK);
                        TCHECK(zk_ctx_set_option(c, "pre_vw", (it & 2) ? 32 : 0) == ZK_OK);
                        TCHECK(zk_ctx_set_option(c, "no_such_key", 1) == ZK_ERR_UNSUPPORTED && zk_ctx_set_option(c, "pre_vw", 48) == ZK_ERR_BAD_ARG);
                        TCHECK(zk_kzg_round_begin_dev(c, srs, 2, in, lens, nullptr) == ZK_OK);
                        TCHECK(zk_ctx_set_option(c, "chunk_l", 64) == ZK_ERR_PENDING);
                        TCHECK(zk_kzg_round_end(c, 2, xy, inf) == ZK_OK);
                        reader.join();
                        TCHECK(zk_ctx_set_option(c, "msm_merge", 1) == ZK_OK && zk_ctx_set_option(c, "pre_vw", 0) == ZK_OK);
                    }
                    // the residency cache of the host-pointer calls: outputs kept, inputs digested and found again, a cache of two
                    // vectors evicting under a batch of four, off = everything returned
                    {
                        const uint64_t* hp[4] = {polys[0].data(), polys[1].data(), polys[2].data(), polys[3].data()};
                        size_t hl[4] = {n, n, n - 1, 64};
                        TCHECK(zk_ctx_set_residency_cache(c, 1, (it & 1) ? 2 * n * 32 : 0, 0) == ZK_OK);
                        std::vector<uint64_t> v(polys[t % 4]);
                        TCHECK(zk_ntt(c, CURVE, ZK_NTT_IFFT, 13, v.data(), n, v.data()) == ZK_OK);          // in place: digested before it is overwritten
                        TCHECK(zk_kzg_commit_batch(c, srs, 4, hp, hl, xy, inf) == ZK_OK);
                        TCHECK(zk_kzg_commit_batch(c, srs, 4, hp, hl, xy, inf) == ZK_OK);
                        TCHECK(zk_kzg_open(c, srs, 3, hp, hl, z, ch, xy, inf) == ZK_OK);
                        uint64_t rh = 0, rm = 0, re = 0, rb = 0;
                        TCHECK(zk_residency_cache_stats(c, &rh, &rm, &re, &rb) == ZK_OK && rh >= 3 && re >= 1 && rb <= ((it & 1) ? 2 * n * 32 : ((size_t)2 << 30)));
                        TCHECK(zk_kzg_round_begin_dev(c, srs, 1, in, lens, nullptr) == ZK_OK);
                        TCHECK(zk_ctx_set_residency_cache(c, 0, 0, 0) == ZK_ERR_PENDING);
                        TCHECK(zk_kzg_round_abort(c) == ZK_OK);
                        TCHECK(zk_ctx_set_residency_cache(c, 0, 0, 0) == ZK_OK);
                        TCHECK(zk_residency_cache_stats(c, nullptr, nullptr, &re, &rb) == ZK_OK && re == 0 && rb == 0);
                    }
                    // the commitment cache: second batch is all hits
                    TCHECK(zk_ctx_set_commit_cache(c, 1, 8) == ZK_OK);
                    TCHECK(zk_kzg_commit_batch_dev(c, srs, 4, in, lens, xy, inf) == ZK_OK);
                    TCHECK(zk_kzg_commit_batch_dev(c, srs, 4, in, lens, xy, inf) == ZK_OK);
                    uint64_t h = 0, m = 0, e = 0;
                    TCHECK(zk_commit_cache_stats(c, &h, &m, &e) == ZK_OK && h >= 4 && e >= 1);
                    TCHECK(zk_ctx_set_commit_cache(c, 0, 0) == ZK_OK);
                }
                for (int k = 0; k < 4; ++k) TCHECK(zk_dev_free(c, d[k]) == ZK_OK);
                zk_ctx_destroy(c);
            });
        for (auto& t : th) t.join();
        REQUIRE(g_fail.load() == 0);
    }

    // ---- 2b. the memory budget of the deferred rounds (round 6): a begin that finds no room for another job's buffer set closes the
    // queued jobs early and reuses their sets; the option's soft limit, a "device" that really runs out (hipMalloc failing), and a
    // device too small for even one set (ZK_ERR_OOM, the ctx usable afterwards)
    {
        std::vector<void*> d(4, nullptr);
        zk_ctx* c = nullptr;
        REQUIRE(zk_ctx_create(0, &c) == ZK_OK);
        for (int k = 0; k < 4; ++k) {
            REQUIRE(zk_dev_alloc(c, n * 32, &d[k]) == ZK_OK);
            REQUIRE(zk_dev_upload(c, d[k], polys[k].data(), n * 32) == ZK_OK);
        }
        const void* in[8] = {d[0], d[1], d[2], d[3], d[0], d[1], d[2], d[3]};
        size_t lens[8] = {n, n, n - 1, 64, n, n - 2, n, n};
        uint64_t xy[16 * 12], z[4] = {5, 0, 0, 0}, ch[4] = {7, 0, 0, 0};
        uint8_t inf[16];
        uint64_t closes = 0, set_bytes = 0, fr = 0, tot = 0;
        REQUIRE(zk_ctx_set_option(c, "round_mem_limit_mb", 40) == ZK_OK && zk_ctx_set_option(c, "mem_reserve_mb", 0) == ZK_OK);
        for (int it = 0; it < 3; ++it) {
            REQUIRE(zk_kzg_round_begin_dev(c, srs, 8, in, lens, nullptr) == ZK_OK);
            REQUIRE(zk_kzg_open_begin_dev(c, srs, 3, in, lens, z, ch) == ZK_OK);
            REQUIRE(zk_kzg_round_begin_dev(c, srs, 4, in, lens, nullptr) == ZK_OK);
            if (it == 1) REQUIRE(zk_kzg_round_reduce(c) == ZK_OK);
            REQUIRE(zk_kzg_round_end(c, 13, xy, inf) == ZK_OK && inf[0] && inf[12]);
        }
        REQUIRE(zk_round_mem_stats(c, &closes, &set_bytes, &fr, &tot) == ZK_OK && closes >= 3 && set_bytes < ((size_t)110 << 20) && tot > fr);     // two table-path sets of 19 MB (nine without the
                                                                                          // budget); n - 1, n - 2 and 64 are per-window jobs
        REQUIRE(zk_kzg_commit_batch_dev(c, srs, 8, in, lens, xy, inf) == ZK_OK && inf[7]);       // the blocking batch in pieces
        REQUIRE(zk_ctx_set_option(c, "round_mem_limit_mb", 0) == ZK_OK);
        zk_ctx_destroy(c);
        // a device with room for about three sets beyond what is live: hipMemGetInfo says so first, hipMalloc itself when the reserve hides it
        for (int lie = 0; lie < 2; ++lie) {
            REQUIRE(zk_ctx_create(0, &c) == ZK_OK);
            REQUIRE(zk_ctx_set_option(c, "mem_reserve_mb", 0) == ZK_OK);
            fake_hip_set_memory(fake_hip_live_bytes() + ((size_t)56 << 20), lie ? (size_t)1 << 30 : 0);
            REQUIRE(zk_kzg_round_begin_dev(c, srs, 8, in, lens, nullptr) == ZK_OK);
            REQUIRE(zk_kzg_open_begin_dev(c, srs, 3, in, lens, z, ch) == ZK_OK);
            REQUIRE(zk_kzg_round_end(c, 9, xy, inf) == ZK_OK && inf[8]);
            REQUIRE(zk_round_mem_stats(c, &closes, nullptr, &fr, nullptr) == ZK_OK && closes >= 1 && (lie || fr < ((size_t)56 << 20)));
            // too small for a single set: ZK_ERR_OOM, nothing queued, the round can be dropped and the ctx works again with memory back
            fake_hip_set_memory(fake_hip_live_bytes() + ((size_t)1 << 20), 0);
            zk_ctx* c2 = nullptr;
            REQUIRE(zk_ctx_create(0, &c2) == ZK_OK);
            REQUIRE(zk_kzg_round_begin_dev(c2, srs, 2, in, lens, nullptr) == ZK_ERR_OOM);
            REQUIRE(zk_kzg_round_abort(c2) == ZK_OK);
            REQUIRE(zk_kzg_commit_batch_dev(c2, srs, 2, in, lens, xy, inf) == ZK_ERR_OOM);
            fake_hip_set_memory(0, 0);
            REQUIRE(zk_kzg_commit_batch_dev(c2, srs, 2, in, lens, xy, inf) == ZK_OK && inf[1]);
            zk_ctx_destroy(c2);
            zk_ctx_destroy(c);
        }
        REQUIRE(zk_ctx_create(0, &c) == ZK_OK);
        for (int k = 0; k < 4; ++k) REQUIRE(zk_dev_free(c, d[k]) == ZK_OK);
        zk_ctx_destroy(c);
    }

    // ---- 2c. the caches' error paths and guarantees (round 6)
    {
        zk_ctx* c = nullptr;
        REQUIRE(zk_ctx_create(0, &c) == ZK_OK);
        const uint64_t* hp[4] = {polys[0].data(), polys[1].data(), polys[2].data(), polys[3].data()};
        size_t hl[4] = {n, n, n - 1, 64};
        uint64_t xy[16 * 12], z[4] = {5, 0, 0, 0}, ch[4] = {7, 0, 0, 0};
        uint8_t inf[16];
        uint64_t h0 = 0, m0 = 0, e0 = 0, b0 = 0, h1 = 0, m1 = 0, e1 = 0, b1 = 0;
        REQUIRE(zk_ctx_set_residency_cache(c, 1, 0, 0) == ZK_OK);
        REQUIRE(zk_kzg_commit_batch(c, srs, 2, hp, hl, xy, inf) == ZK_OK);                 // two vectors resident
        REQUIRE(zk_residency_cache_stats(c, &h0, &m0, &e0, &b0) == ZK_OK && e0 == 2);
        // a call that fails AFTER it has created entries (a staging buffer of a later polynomial finds no memory): the entries it made or
        // touched are gone -- their digests would name bytes that never arrived -- and the same call succeeds once memory is back
        fake_hip_set_memory(fake_hip_live_bytes() + 4096, 0);
        REQUIRE(zk_kzg_open(c, srs, 4, hp, hl, z, ch, xy, inf) == ZK_ERR_OOM);
        REQUIRE(zk_residency_cache_stats(c, &h1, &m1, &e1, &b1) == ZK_OK && e1 <= e0 && b1 <= b0);
        REQUIRE(zk_ctx_set_commit_cache(c, 1, 8) == ZK_OK);
        zk_ctx* cfresh = nullptr;                                                       // no staging buffers yet: its first upload fails
        REQUIRE(zk_ctx_create(0, &cfresh) == ZK_OK);
        REQUIRE(zk_ctx_set_residency_cache(cfresh, 1, 0, 0) == ZK_OK && zk_ctx_set_commit_cache(cfresh, 1, 8) == ZK_OK);
        REQUIRE(zk_kzg_commit_batch(cfresh, srs, 4, hp, hl, xy, inf) == ZK_ERR_OOM);
        REQUIRE(zk_residency_cache_stats(cfresh, nullptr, nullptr, &e1, &b1) == ZK_OK && e1 == 0 && b1 == 0);
        fake_hip_set_memory(0, 0);
        REQUIRE(zk_kzg_commit_batch(cfresh, srs, 4, hp, hl, xy, inf) == ZK_OK && inf[0] && inf[3]);
        REQUIRE(zk_kzg_open(c, srs, 4, hp, hl, z, ch, xy, inf) == ZK_OK);
        zk_ctx_destroy(cfresh);
        // cache_verify: every hit is checked against the real thing; nothing may be found wrong
        uint64_t checked = 0, wrong = 9;
        REQUIRE(zk_ctx_set_option(c, "cache_verify", 1) == ZK_OK);
        REQUIRE(zk_kzg_commit_batch(c, srs, 4, hp, hl, xy, inf) == ZK_OK);
        REQUIRE(zk_kzg_commit_batch(c, srs, 4, hp, hl, xy, inf) == ZK_OK);
        REQUIRE(zk_kzg_open(c, srs, 3, hp, hl, z, ch, xy, inf) == ZK_OK);
        REQUIRE(zk_cache_verify_stats(c, &checked, &wrong) == ZK_OK && checked >= 6 && wrong == 0);
        REQUIRE(zk_ctx_set_option(c, "cache_verify", 0) == ZK_OK && zk_ctx_set_option(c, "cache_verify", 2) == ZK_ERR_BAD_ARG);
        zk_ctx_destroy(c);
        // no operating-system entropy behind the digest key: the caches whose hits replace a computation refuse to switch on, and the
        // SRS registry shares nothing
        zk_process_key_reset_for_tests("/nonexistent/entropy");
        REQUIRE(zk_ctx_create(0, &c) == ZK_OK);
        REQUIRE(zk_ctx_set_commit_cache(c, 1, 8) == ZK_ERR_UNSUPPORTED && zk_ctx_set_residency_cache(c, 1, 0, 0) == ZK_ERR_UNSUPPORTED);
        REQUIRE(zk_ctx_set_commit_cache(c, 0, 0) == ZK_OK && zk_ctx_set_residency_cache(c, 0, 0, 0) == ZK_OK);
        zk_srs *s1 = nullptr, *s2 = nullptr;
        REQUIRE(zk_srs_register(c, CURVE, srs_a.data(), nullptr, n, &s1) == ZK_OK && zk_srs_register(c, CURVE, srs_a.data(), nullptr, n, &s2) == ZK_OK);
        REQUIRE(s1 != s2 && s1 != srs);
        REQUIRE(zk_kzg_commit_batch(c, s1, 2, hp, hl, xy, inf) == ZK_OK);
        zk_srs_free(s1);
        zk_srs_free(s2);
        zk_ctx_destroy(c);
        zk_process_key_reset_for_tests(nullptr);
    }

    // ---- 3. eviction: drop every unreferenced entry, then the last references
    zk_srs_free(srs);
    REQUIRE(zk_srs_cache_config(0) == ZK_OK);
    uint64_t entries = 99;
    REQUIRE(zk_srs_cache_stats(nullptr, nullptr, &entries, nullptr) == ZK_OK && entries == 0);
    REQUIRE(zk_srs_cache_config((size_t)32 << 30) == ZK_OK);
    zk_ctx_destroy(shared);
    REQUIRE(fake_hip_launches() > 1000);
    REQUIRE(fake_hip_live_allocations() == 0);        // every device / pinned buffer of every ctx and SRS was returned
    printf("host_stress ok: %ld kernel launches (no-ops), all device memory returned\n", fake_hip_launches());
    return 0;
}
