// Smallest C++ caller of the host layer (host/ark_plonk_amd.hpp): a round trip through the transforms and
// the commitments of one "prover round" on device-resident data.  tests/test_host_cpp_gpu.py builds it with
// g++ against libark_plonk_amd.so, runs it on the GPU and compares its output with the Python mirror.
//   usage: example <log_n>        prints  "<label> <hex limbs>"  lines
#include <cstdio>
#include <cstdlib>

#include "ark_plonk_amd.hpp"

static void print(const char* label, const std::vector<uint64_t>& v, size_t limbs) {
    std::printf("%s", label);
    for (size_t i = 0; i < limbs && i < v.size(); ++i) std::printf(" %016llx", (unsigned long long)v[i]);
    std::printf("\n");
}

int main(int argc, char** argv) {
    const uint32_t log_n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 13;
    const size_t n = (size_t)1 << log_n;
    try {
        zk::Context ctx(0);
        auto dom = zk::Radix2EvaluationDomain::create(ctx, n).value();
        // evaluations: splitmix-like residues below 2^254 (valid Montgomery elements of BLS12-381 Fr)
        std::vector<uint64_t> ev(4 * n);
        uint64_t s = 0x5EED0000;
        for (size_t i = 0; i < 4 * n; ++i) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            ev[i] = (i % 4 == 3) ? (z >> 2) : z;
        }
        // host-buffer API: ifft then fft is the identity
        auto coeffs = dom.ifft(ev);
        auto back = dom.fft(coeffs);
        std::printf("roundtrip %s\n", back == ev ? "ok" : "MISMATCH");
        print("coeff0", coeffs, 4);
        // SRS: P_i = k_i G with small scalars (fixed-base kernel), registered + window table
        std::vector<uint64_t> ks(4 * n, 0);
        for (size_t i = 0; i < n; ++i) ks[4 * i] = 3 + 2 * i;
        zk::DeviceVec d_ks(ctx, ks);
        std::vector<uint64_t> srs(12 * n);
        void* d_srs = nullptr;
        zk::check(zk_dev_alloc(ctx.handle(), 96 * n, &d_srs), "zk_dev_alloc");
        zk::check(zk_g1_fixed_base_batch_dev(ctx.handle(), ZK_CURVE_BLS12_381, d_ks.data(), n, d_srs), "zk_g1_fixed_base_batch_dev");
        zk::check(zk_dev_download(ctx.handle(), srs.data(), d_srs, 96 * n), "zk_dev_download");
        zk::check(zk_dev_free(ctx.handle(), d_srs), "zk_dev_free");
        zk::CommitterKey ck(ctx, srs);
        zk::G1Affine single = ck.commit(coeffs);          // per-window path (no table yet)
        ck.precompute();
        zk::DeviceVec d_ev(ctx, ev);
        zk::DeviceVec d_coeffs = dom.transform(ZK_NTT_IFFT, d_ev);
        auto round = ck.commit_round({&d_coeffs, &d_coeffs});
        std::printf("commit_round %s\n", (round[0].xy == single.xy && round[1].xy == single.xy && !single.infinity) ? "ok" : "MISMATCH");
        print("commit_x", single.xy, 6);
        // the same two commitments as two deferred calls collected once (f | h_1 of prover.rs:289-312)
        ck.commit_begin({&d_coeffs});
        zk::DeviceVec d_again = dom.transform(ZK_NTT_IFFT, d_ev);      // a transform queued while the round is open
        ck.commit_begin({&d_again});
        ck.round_reduce();                                               // reductions queued; the next transform runs under the host's part
        zk::DeviceVec d_under = dom.transform(ZK_NTT_FFT, d_coeffs);
        auto deferred = ck.round_end();
        std::printf("deferred_round %s\n", (deferred.size() == 2 && deferred[0].xy == single.xy && deferred[1].xy == single.xy) ? "ok" : "MISMATCH");
        // the multi-GPU exchange's device form with one "rank": the jobs' virtual-window sums (2 VW points each) stay on the device, one kernel
        // adds the ranks' element-wise, the host pool combines -- here with a planner option flipped in between (options never change a result)
        {
            void* d_ws = nullptr;
            zk::check(zk_dev_alloc(ctx.handle(), 2 * ck.winsums_bytes(), &d_ws), "zk_dev_alloc");
            ctx.set_option("msm_merge", 0);
            ck.commit_begin({&d_coeffs});
            ck.commit_begin({&d_again});
            ck.round_end_winsums_dev(d_ws);
            ctx.set_option("msm_merge", 1);
            auto summed = ck.sum_winsums_dev(d_ws, 1, 2);
            zk::check(zk_dev_free(ctx.handle(), d_ws), "zk_dev_free");
            std::printf("device_winsums %s\n", (ctx.get_option("msm_merge") == 1 && summed.size() == 2 && summed[0].xy == single.xy &&
                                                summed[1].xy == single.xy) ? "ok" : "MISMATCH");
        }
        // the unchanged caller's PC::commit(ck, polys): host vectors, one call for the whole slice (prover.rs:213)
        auto host_round = ck.commit({&coeffs, &ev, &coeffs});
        zk::CommitterKey again(ctx, srs);                 // PC::trim on the next gen_proof: the same bytes -> the resident SRS and table
        auto ev_commit = again.commit(ev);
        std::printf("host_batch %s\n", (host_round[0].xy == single.xy && host_round[2].xy == single.xy && host_round[1].xy == ev_commit.xy) ? "ok" : "MISMATCH");
        // the same host batch with the residency cache on: the second call finds its three vectors on the device (two distinct ones
        // uploaded by the first), a rewritten vector is a miss -- same points either way
        {
            ctx.set_residency_cache(true);
            auto first = ck.commit({&coeffs, &ev, &coeffs});
            const auto before = ctx.residency_stats();
            auto second = ck.commit({&coeffs, &ev, &coeffs});
            const auto after = ctx.residency_stats();
            std::vector<uint64_t> changed(coeffs);
            changed[4] ^= 1;
            auto third = ck.commit({&changed});
            ctx.set_residency_cache(false);
            std::printf("residency_cache %s\n", (first[0].xy == single.xy && second[0].xy == single.xy && second[1].xy == ev_commit.xy &&
                                                 second[2].xy == single.xy && after.hits - before.hits == 3 && third[0].xy != single.xy &&
                                                 ctx.residency_stats().entries == 0) ? "ok" : "MISMATCH");
        }
        // round 1 of the prover's transcript: four wire commitments in, zeta out (prover.rs:217-226)
        zk::Transcript pre("example");
        pre.circuit_domain_sep(n);
        zk::Transcript t(pre);
        t.append("w_l", single);
        t.append("w_r", ev_commit);
        t.append("w_o", single);
        t.append("w_4", ev_commit);
        auto zeta = t.challenge_scalar("zeta");
        t.append("zeta", zeta.data());
        print("zeta", zeta, 4);
        auto enc = zk::serialize(single);
        std::printf("commit_ser");
        for (uint8_t b : enc) std::printf(" %02x", b);
        std::printf("\n");
        return (back == ev && round[0].xy == single.xy && host_round[0].xy == single.xy) ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
}
