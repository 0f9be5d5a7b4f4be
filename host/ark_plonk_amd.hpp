// Header-only C++ host layer over the C ABI (include/ark_plonk_amd.h), mirroring the reference
// interfaces of the hot path with the same names and argument meaning:
//   zk::Radix2EvaluationDomain  <- ark_poly::EvaluationDomain (prover.rs:169-173,196-203; quotient_poly.rs:64-120)
//   zk::VariableBaseMSM         <- ark_ec::msm::VariableBaseMSM (commitment.rs:45)
//   zk::CommitterKey::commit    <- KZG10 PC::commit (prover.rs:213 ...); commit_round for the polynomials of one call
//   zk::DeviceVec               -- device-resident Fr vector (what stays on the GPU between fft and commit)
//   zk::permutation_evals / lookup_permutation_evals / quotient_evals
//                               <- permutation/mod.rs:652-822, quotient_poly.rs:34-178
//   zk::evaluate / zk::lincomb  <- DensePolynomial::evaluate and the scalar * polynomial sums of linearisation_poly.rs:203-336
//   zk::Transcript              <- merlin::Transcript + TranscriptProtocol (transcript.rs:16-49); zk::serialize(...) <- ark-serialize
//                                  encodings of Fr / G1Affine
// The reference is Rust and infallible at these call sites; here failures throw zk::Error.
#pragma once
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/ark_plonk_amd.h"

namespace zk {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char* where) : std::runtime_error(std::string(where) + ": " + zk_strerror(c)), code(c) {}
};
inline void check(int rc, const char* where) {
    if (rc != ZK_OK) throw Error(rc, where);
}

using Fr = uint64_t[4];  // Montgomery limbs, arkworks layout

class Context {
  public:
    explicit Context(int device = 0) { check(zk_ctx_create(device, &h_), "zk_ctx_create"); }
    ~Context() { zk_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    zk_ctx* handle() const { return h_; }
    void sync() { check(zk_ctx_sync(h_), "zk_ctx_sync"); }
    // tuning options of the MSM planner (see the header: "msm_merge", "pre_vw", "pre_logg", "chunk_l", "long_rounds", "combine_sg",
    // "pre_max_log_n"); results never depend on them
    void set_option(const char* key, int64_t value) { check(zk_ctx_set_option(h_, key, value), "zk_ctx_set_option"); }
    int64_t get_option(const char* key) const {
        int64_t v = 0;
        check(zk_ctx_get_option(h_, key, &v), "zk_ctx_get_option");
        return v;
    }
    // the host-pointer calls stop re-uploading what the library produced or has seen (an ifft output that comes back as a commit /
    // coset_fft / open input): opt-in, hits on a keyed digest of the caller's current bytes; 0 = leave a size unchanged
    void set_residency_cache(bool on, size_t capacity_bytes = 0, size_t max_vector_bytes = 0) {
        check(zk_ctx_set_residency_cache(h_, on ? 1 : 0, capacity_bytes, max_vector_bytes), "zk_ctx_set_residency_cache");
    }
    struct ResidencyStats {
        uint64_t hits = 0, misses = 0, entries = 0, bytes = 0;
    };
    ResidencyStats residency_stats() const {
        ResidencyStats s;
        check(zk_residency_cache_stats(h_, &s.hits, &s.misses, &s.entries, &s.bytes), "zk_residency_cache_stats");
        return s;
    }

  private:
    zk_ctx* h_ = nullptr;
};

// n Fr elements (4 limbs each) in device memory
class DeviceVec {
  public:
    DeviceVec(Context& ctx, size_t n) : ctx_(&ctx), n_(n) { check(zk_dev_alloc(ctx.handle(), (n ? n : 1) * 32, &p_), "zk_dev_alloc"); }
    DeviceVec(Context& ctx, const std::vector<uint64_t>& host) : DeviceVec(ctx, host.size() / 4) {
        if (n_) check(zk_dev_upload(ctx.handle(), p_, host.data(), n_ * 32), "zk_dev_upload");
    }
    ~DeviceVec() {
        if (p_) zk_dev_free(ctx_->handle(), p_);
    }
    DeviceVec(const DeviceVec&) = delete;
    DeviceVec& operator=(const DeviceVec&) = delete;
    DeviceVec(DeviceVec&& o) noexcept : ctx_(o.ctx_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    void* data() const { return p_; }
    size_t size() const { return n_; }
    void truncate(size_t n) { n_ = n < n_ ? n : n_; }       // keep the allocation, shorten the logical length
    std::vector<uint64_t> to_host() const {
        std::vector<uint64_t> h(4 * n_);
        if (n_) check(zk_dev_download(ctx_->handle(), h.data(), p_, n_ * 32), "zk_dev_download");
        return h;
    }

  private:
    Context* ctx_;
    void* p_ = nullptr;
    size_t n_;
};

// GeneralEvaluationDomain::Radix2 over the scalar field of `curve`
class Radix2EvaluationDomain {
  public:
    // EvaluationDomain::new -> None when num_coeffs exceeds the two-adicity (error.rs:14-21)
    static std::optional<Radix2EvaluationDomain> create(Context& ctx, uint64_t num_coeffs, int curve = ZK_CURVE_BLS12_381) {
        zk_domain_info info;
        int rc = zk_domain_new(curve, num_coeffs, &info);
        if (rc == ZK_ERR_DOMAIN_TOO_LARGE) return std::nullopt;
        check(rc, "zk_domain_new");
        return Radix2EvaluationDomain(ctx, curve, info);
    }
    uint64_t size() const { return info_.size; }
    uint32_t log_size_of_group() const { return info_.log_size_of_group; }
    const uint64_t* size_inv() const { return info_.size_inv; }
    const uint64_t* group_gen() const { return info_.group_gen; }
    const uint64_t* group_gen_inv() const { return info_.group_gen_inv; }
    const uint64_t* generator_inv() const { return info_.generator_inv; }

    // coeffs/evals: 4 limbs per element; the vector is resized to size() like the *_in_place methods
    void fft_in_place(std::vector<uint64_t>& v) const { run(ZK_NTT_FFT, v); }
    void ifft_in_place(std::vector<uint64_t>& v) const { run(ZK_NTT_IFFT, v); }
    void coset_fft_in_place(std::vector<uint64_t>& v) const { run(ZK_NTT_COSET_FFT, v); }
    void coset_ifft_in_place(std::vector<uint64_t>& v) const { run(ZK_NTT_COSET_IFFT, v); }
    std::vector<uint64_t> fft(const std::vector<uint64_t>& c) const { auto v = c; fft_in_place(v); return v; }
    std::vector<uint64_t> ifft(const std::vector<uint64_t>& e) const { auto v = e; ifft_in_place(v); return v; }
    std::vector<uint64_t> coset_fft(const std::vector<uint64_t>& c) const { auto v = c; coset_fft_in_place(v); return v; }
    std::vector<uint64_t> coset_ifft(const std::vector<uint64_t>& e) const { auto v = e; coset_ifft_in_place(v); return v; }
    // device-resident form: `in` (<= size() elements, zero-extended) -> a new vector of size() elements
    DeviceVec transform(int kind, const DeviceVec& in) const {
        DeviceVec out(*ctx_, info_.size);
        check(zk_ntt_dev(ctx_->handle(), curve_, kind, info_.log_size_of_group, in.data(), in.size(), out.data()), "zk_ntt_dev");
        return out;
    }
    Context& context() const { return *ctx_; }
    int curve() const { return curve_; }

  private:
    Radix2EvaluationDomain(Context& ctx, int curve, const zk_domain_info& i) : ctx_(&ctx), curve_(curve), info_(i) {}
    void run(int kind, std::vector<uint64_t>& v) const {
        if (v.size() % 4) throw Error(ZK_ERR_BAD_ARG, "element vector");
        size_t in_len = v.size() / 4;
        if (in_len > info_.size) throw Error(ZK_ERR_BAD_ARG, "input longer than the domain");
        v.resize(4 * info_.size, 0);
        check(zk_ntt(ctx_->handle(), curve_, kind, info_.log_size_of_group, v.data(), in_len, v.data()), "zk_ntt");
    }
    Context* ctx_;
    int curve_;
    zk_domain_info info_;
};

struct G1Affine {
    std::vector<uint64_t> xy;  // x || y, Montgomery limbs
    bool infinity = false;
};

inline int fq_limbs(int curve) { return curve == ZK_CURVE_BLS12_381 ? 6 : 4; }

struct VariableBaseMSM {
    // bases: n x 2L limbs (+ optional infinity flags); scalars: canonical 4-limb integers.
    // Truncates to the shorter slice like the reference.
    static G1Affine multi_scalar_mul(Context& ctx, const std::vector<uint64_t>& bases, const std::vector<uint64_t>& scalars,
                                     int curve = ZK_CURVE_BLS12_381, const std::vector<uint8_t>* infinity = nullptr) {
        const int L = fq_limbs(curve);
        size_t n = std::min(bases.size() / (2 * L), scalars.size() / 4);
        G1Affine out;
        out.xy.assign(2 * L, 0);
        uint8_t inf = 0;
        check(zk_msm_g1(ctx.handle(), curve, bases.data(), infinity ? infinity->data() : nullptr, scalars.data(), n, out.xy.data(), &inf),
              "zk_msm_g1");
        out.infinity = inf != 0;
        return out;
    }
};

// device-resident powers_of_g (PC::trim output) + KZG10::commit
class CommitterKey {
  public:
    // PC::trim: content-addressed -- a second key over the same bases (circuit.rs:276 trims on every gen_proof) shares the
    // resident device copy and window table.  infinity: optional per-point flags; (0, 1) and (0, 0) are infinity too.
    CommitterKey(Context& ctx, const std::vector<uint64_t>& powers_of_g, int curve = ZK_CURVE_BLS12_381,
                 const std::vector<uint8_t>* infinity = nullptr)
        : ctx_(&ctx), curve_(curve) {
        const size_t n = powers_of_g.size() / (2 * fq_limbs(curve));
        if (infinity && infinity->size() != n) throw Error(ZK_ERR_BAD_ARG, "infinity flags length");
        check(zk_srs_register(ctx.handle(), curve, powers_of_g.data(), infinity ? infinity->data() : nullptr, n, &h_), "zk_srs_register");
    }
    ~CommitterKey() { zk_srs_free(h_); }
    CommitterKey(const CommitterKey&) = delete;
    CommitterKey& operator=(const CommitterKey&) = delete;
    size_t size() const { return zk_srs_len(h_); }
    // window-multiples table (once per key, like PC::trim): enables the fused round batches
    void precompute() { check(zk_srs_precompute(ctx_->handle(), h_), "zk_srs_precompute"); }
    // PC::commit over the labeled polynomials of ONE call (prover.rs:213 passes four, :579 seven): one batch
    std::vector<G1Affine> commit_round(const std::vector<const DeviceVec*>& polys) const {
        const int L = fq_limbs(curve_);
        const uint32_t k = (uint32_t)polys.size();
        std::vector<const void*> ptrs(k);
        std::vector<size_t> lens(k);
        for (uint32_t i = 0; i < k; ++i) {
            ptrs[i] = polys[i]->data();
            lens[i] = polys[i]->size();
        }
        std::vector<uint64_t> xy((size_t)k * 2 * L);
        std::vector<uint8_t> inf(k ? k : 1);
        check(zk_kzg_round_batch_dev(ctx_->handle(), h_, k, ptrs.data(), lens.data(), nullptr, xy.data(), inf.data()), "zk_kzg_round_batch_dev");
        std::vector<G1Affine> out(k);
        for (uint32_t i = 0; i < k; ++i) {
            out[i].xy.assign(xy.begin() + (size_t)i * 2 * L, xy.begin() + (size_t)(i + 1) * 2 * L);
            out[i].infinity = inf[i] != 0;
        }
        return out;
    }
    // The deferred form of the same call (zk_kzg_round_begin_dev ... zk_kzg_round_end): commitments whose inputs do not depend on
    // each other's results -- f | h_1 | h_2 (prover.rs:289-317), z | z_2 (prover.rs:361-389), the last round's four calls
    // (prover.rs:579-618) -- are begun call by call and collected once, in submission order.
    void commit_begin(const std::vector<const DeviceVec*>& polys) const {
        const uint32_t k = (uint32_t)polys.size();
        std::vector<const void*> ptrs(k);
        std::vector<size_t> lens(k);
        for (uint32_t i = 0; i < k; ++i) {
            ptrs[i] = polys[i]->data();
            lens[i] = polys[i]->size();
        }
        check(zk_kzg_round_begin_dev(ctx_->handle(), h_, k, ptrs.data(), lens.data(), nullptr), "zk_kzg_round_begin_dev");
    }
    // PC::open as a job of the open round (prover.rs:582-591,609-618)
    void open_begin(const std::vector<const DeviceVec*>& polys, const uint64_t* z_mont, const uint64_t* challenge_mont) const {
        const uint32_t k = (uint32_t)polys.size();
        std::vector<const void*> ptrs(k);
        std::vector<size_t> lens(k);
        for (uint32_t i = 0; i < k; ++i) {
            ptrs[i] = polys[i]->data();
            lens[i] = polys[i]->size();
        }
        check(zk_kzg_open_begin_dev(ctx_->handle(), h_, k, ptrs.data(), lens.data(), z_mont, challenge_mont), "zk_kzg_open_begin_dev");
    }
    // optional: queue the round's reductions now; what the caller launches on the stream until round_end() (transforms that do
    // not depend on this round's results) runs behind them, while round_end() does the host part
    void round_reduce() const { check(zk_kzg_round_reduce(ctx_->handle()), "zk_kzg_round_reduce"); }
    std::vector<G1Affine> round_end() const {
        const int L = fq_limbs(curve_);
        uint32_t k = 0;
        check(zk_kzg_round_pending(ctx_->handle(), &k), "zk_kzg_round_pending");
        std::vector<uint64_t> xy((size_t)(k ? k : 1) * 2 * L);
        std::vector<uint8_t> inf(k ? k : 1);
        check(zk_kzg_round_end(ctx_->handle(), k, xy.data(), inf.data()), "zk_kzg_round_end");
        std::vector<G1Affine> out(k);
        for (uint32_t i = 0; i < k; ++i) {
            out[i].xy.assign(xy.begin() + (size_t)i * 2 * L, xy.begin() + (size_t)(i + 1) * 2 * L);
            out[i].infinity = inf[i] != 0;
        }
        return out;
    }
    // ---- multi-GPU (one process per GPU; the reference is a single process): see INTEGRATION.md section 5
    // rank g of G holds the whole SRS and builds the table rows of the windows g, g + G, ... only: every MSM / commitment over this
    // key is then the rank's PARTIAL (zk_srs_precompute_rows)
    void precompute_rows(uint32_t rank, uint32_t ranks, uint32_t window_bits = 0) {
        check(zk_srs_precompute_rows(ctx_->handle(), h_, window_bits, rank, ranks), "zk_srs_precompute_rows");
    }
    // the exchange without a host hop: every job's 2 VW virtual-window sums, winsums_bytes() per job, left on the device where the
    // single-GPU path's last reduction kernel writes them (the send buffer of the all-gather); all-gathered; added element-wise
    size_t winsums_bytes() const { return zk_winsums_dev_bytes(ctx_->handle(), h_); }
    void round_end_winsums_dev(void* d_out) const {
        uint32_t k = 0;
        check(zk_kzg_round_pending(ctx_->handle(), &k), "zk_kzg_round_pending");
        check(zk_kzg_round_end_winsums_dev(ctx_->handle(), k, d_out), "zk_kzg_round_end_winsums_dev");
    }
    std::vector<G1Affine> sum_winsums_dev(const void* d_all, size_t ranks, uint32_t jobs) const {
        const int L = fq_limbs(curve_);
        std::vector<uint64_t> xy((size_t)(jobs ? jobs : 1) * 2 * L);
        std::vector<uint8_t> inf(jobs ? jobs : 1);
        check(zk_g1_sum_winsums_dev(ctx_->handle(), h_, d_all, ranks, jobs, xy.data(), inf.data()), "zk_g1_sum_winsums_dev");
        std::vector<G1Affine> out(jobs);
        for (uint32_t i = 0; i < jobs; ++i) {
            out[i].xy.assign(xy.begin() + (size_t)i * 2 * L, xy.begin() + (size_t)(i + 1) * 2 * L);
            out[i].infinity = inf[i] != 0;
        }
        return out;
    }
    // PC::commit(ck, polys, None) with host coefficient vectors (prover.rs:213,579,606): uploads overlap the MSMs
    std::vector<G1Affine> commit(const std::vector<const std::vector<uint64_t>*>& polys) const {
        const int L = fq_limbs(curve_);
        const uint32_t k = (uint32_t)polys.size();
        std::vector<const uint64_t*> ptrs(k);
        std::vector<size_t> lens(k);
        for (uint32_t i = 0; i < k; ++i) {
            ptrs[i] = polys[i]->data();
            lens[i] = polys[i]->size() / 4;
        }
        std::vector<uint64_t> xy((size_t)k * 2 * L);
        std::vector<uint8_t> inf(k ? k : 1);
        check(zk_kzg_commit_batch(ctx_->handle(), h_, k, ptrs.data(), lens.data(), xy.data(), inf.data()), "zk_kzg_commit_batch");
        std::vector<G1Affine> out(k);
        for (uint32_t i = 0; i < k; ++i) {
            out[i].xy.assign(xy.begin() + (size_t)i * 2 * L, xy.begin() + (size_t)(i + 1) * 2 * L);
            out[i].infinity = inf[i] != 0;
        }
        return out;
    }
    G1Affine commit(const std::vector<uint64_t>& coeffs_mont) const {
        G1Affine out;
        out.xy.assign(2 * fq_limbs(curve_), 0);
        uint8_t inf = 0;
        check(zk_kzg_commit(ctx_->handle(), h_, coeffs_mont.data(), coeffs_mont.size() / 4, out.xy.data(), &inf), "zk_kzg_commit");
        out.infinity = inf != 0;
        return out;
    }

  private:
    Context* ctx_;
    int curve_;
    zk_srs* h_ = nullptr;
};

// ark-serialize 0.3 CanonicalSerialize (compressed) of a commitment / a scalar
inline std::vector<uint8_t> serialize(const G1Affine& p, int curve = ZK_CURVE_BLS12_381) {
    std::vector<uint8_t> out(zk_g1_compressed_size(curve));
    check(zk_g1_serialize_compressed(curve, p.xy.data(), p.infinity ? 1 : 0, out.data()), "zk_g1_serialize_compressed");
    return out;
}
inline std::vector<uint8_t> serialize_fr(const uint64_t* fr_mont, int curve = ZK_CURVE_BLS12_381) {
    std::vector<uint8_t> out(zk_fr_serialized_size(curve));
    check(zk_fr_serialize(curve, fr_mont, out.data()), "zk_fr_serialize");
    return out;
}

// merlin::Transcript with plonk-core's TranscriptProtocol on top (transcript.rs:16-49)
class Transcript {
  public:
    explicit Transcript(const std::string& label, int curve = ZK_CURVE_BLS12_381)
        : h_(zk_transcript_new((const uint8_t*)label.data(), label.size())), curve_(curve) {
        if (!h_) throw Error(ZK_ERR_BAD_ARG, "zk_transcript_new");
    }
    Transcript(const Transcript& o) : h_(zk_transcript_clone(o.h_)), curve_(o.curve_) {      // prover.rs:179 clones the preprocessed one
        if (!h_) throw Error(ZK_ERR_BAD_ARG, "zk_transcript_clone");
    }
    Transcript& operator=(const Transcript&) = delete;
    ~Transcript() { zk_transcript_free(h_); }
    void append_message(const std::string& label, const std::vector<uint8_t>& msg) {
        check(zk_transcript_append_message(h_, (const uint8_t*)label.data(), label.size(), msg.data(), msg.size()), "zk_transcript_append_message");
    }
    void append(const std::string& label, const G1Affine& commitment) {
        check(zk_transcript_append_g1(h_, curve_, (const uint8_t*)label.data(), label.size(), commitment.xy.data(), commitment.infinity ? 1 : 0),
              "zk_transcript_append_g1");
    }
    void append(const std::string& label, const uint64_t* fr_mont) {
        check(zk_transcript_append_fr(h_, curve_, (const uint8_t*)label.data(), label.size(), fr_mont), "zk_transcript_append_fr");
    }
    // challenge_scalar: 31 challenge bytes read as a little-endian integer, returned in Montgomery form
    std::vector<uint64_t> challenge_scalar(const std::string& label) {
        std::vector<uint64_t> out(4);
        check(zk_transcript_challenge_scalar(h_, curve_, (const uint8_t*)label.data(), label.size(), out.data()), "zk_transcript_challenge_scalar");
        return out;
    }
    void circuit_domain_sep(uint64_t n) { check(zk_transcript_circuit_domain_sep(h_, n), "zk_transcript_circuit_domain_sep"); }

  private:
    zk_transcript* h_;
    int curve_;
};

// Permutation::compute_permutation_poly up to its ifft (permutation/mod.rs:652-747): z over the domain
inline DeviceVec permutation_evals(const Radix2EvaluationDomain& d, const DeviceVec* const wires[4], const DeviceVec* const sigma_evals[4],
                                   const uint64_t* beta_mont, const uint64_t* gamma_mont) {
    const void* w[4];
    const void* s[4];
    for (int k = 0; k < 4; ++k) {
        if (wires[k]->size() != d.size() || sigma_evals[k]->size() != d.size()) throw Error(ZK_ERR_BAD_ARG, "column length");
        w[k] = wires[k]->data();
        s[k] = sigma_evals[k]->data();
    }
    DeviceVec out(d.context(), d.size());
    check(zk_perm_product_dev(d.context().handle(), d.curve(), d.log_size_of_group(), w, s, beta_mont, gamma_mont, out.data(), nullptr),
          "zk_perm_product_dev");
    return out;
}
// compute_lookup_permutation_poly up to its ifft (permutation/mod.rs:754-797)
inline DeviceVec lookup_permutation_evals(Context& ctx, int curve, const DeviceVec& f, const DeviceVec& t, const DeviceVec& h1,
                                          const DeviceVec& h2, const uint64_t* delta_mont, const uint64_t* epsilon_mont) {
    if (t.size() != f.size() || h1.size() != f.size() || h2.size() != f.size()) throw Error(ZK_ERR_BAD_ARG, "column length");   // mod.rs:764-767 asserts
    DeviceVec out(ctx, f.size());
    check(zk_lookup_product_dev(ctx.handle(), curve, f.size(), f.data(), t.data(), h1.data(), h2.data(), delta_mont, epsilon_mont, out.data(),
                                nullptr),
          "zk_lookup_product_dev");
    return out;
}
// quotient_poly::compute between its coset FFTs and its coset iFFT (quotient_poly.rs:122-173); `d` = the size-n domain
inline DeviceVec quotient_evals(const Radix2EvaluationDomain& d, const zk_quotient_args& args) {
    DeviceVec out(d.context(), 4 * d.size());
    check(zk_quotient_evals_dev(d.context().handle(), d.curve(), d.log_size_of_group(), &args, out.data()), "zk_quotient_evals_dev");
    return out;
}
// MultiSet::combine_split (lookup/multiset.rs:131-176): t.combine_split(&f) -> (h_1, h_2); Error::ElementNotIndexed -> zk::Error(ZK_ERR_NOT_INDEXED)
inline std::pair<DeviceVec, DeviceVec> combine_split(Context& ctx, int curve, const DeviceVec& t, const DeviceVec& f) {
    const size_t cap = (t.size() + f.size() + 1) / 2;
    DeviceVec h1(ctx, cap), h2(ctx, cap);
    size_t l1 = 0, l2 = 0;
    check(zk_lookup_combine_split_dev(ctx.handle(), curve, t.data(), t.size(), f.data(), f.size(), h1.data(), h2.data(), &l1, &l2),
          "zk_lookup_combine_split_dev");
    h1.truncate(l1);
    h2.truncate(l2);
    return {std::move(h1), std::move(h2)};
}
// prover.rs:244-279: the compressed query column over wires[0]->size() rows
inline DeviceVec lookup_query(Context& ctx, int curve, const DeviceVec& q_lookup, const DeviceVec* const wires[4], const uint64_t* zeta_mont,
                              const DeviceVec& table_compressed) {
    const size_t n = wires[0]->size();
    const void* w[4];
    for (int k = 0; k < 4; ++k) {
        if (wires[k]->size() != n) throw Error(ZK_ERR_BAD_ARG, "column length");
        w[k] = wires[k]->data();
    }
    if (table_compressed.size() == 0) throw Error(ZK_ERR_BAD_ARG, "empty table");
    DeviceVec out(ctx, n);
    check(zk_lookup_query_dev(ctx.handle(), curve, n, q_lookup.data(), q_lookup.size(), w, zeta_mont, table_compressed.data(), out.data()),
          "zk_lookup_query_dev");
    return out;
}
// `DensePolynomial::evaluate` for a batch (linearisation_poly.rs:203-261: 16 polynomials at z, 7 at z*omega): polys[k] at
// points_mont[4k..4k+4) -> 4 Montgomery limbs per polynomial
inline std::vector<uint64_t> evaluate(Context& ctx, int curve, const std::vector<const DeviceVec*>& polys, const std::vector<uint64_t>& points_mont) {
    if (points_mont.size() != 4 * polys.size()) throw Error(ZK_ERR_BAD_ARG, "one point per polynomial");
    std::vector<const void*> ptrs(polys.size());
    std::vector<size_t> lens(polys.size());
    for (size_t k = 0; k < polys.size(); ++k) {
        ptrs[k] = polys[k]->data();
        lens[k] = polys[k]->size();
    }
    std::vector<uint64_t> out(4 * polys.size());
    check(zk_poly_evaluate_dev(ctx.handle(), curve, (uint32_t)polys.size(), ptrs.data(), lens.data(), points_mont.data(), out.data()),
          "zk_poly_evaluate_dev");
    return out;
}
// sum_k coeffs[k] * polys[k]: the `&poly * scalar` / `+` chains of the linearisation polynomial (linearisation_poly.rs:288-336)
inline DeviceVec lincomb(Context& ctx, int curve, const std::vector<const DeviceVec*>& polys, const std::vector<uint64_t>& coeffs_mont) {
    if (coeffs_mont.size() != 4 * polys.size()) throw Error(ZK_ERR_BAD_ARG, "one coefficient per polynomial");
    std::vector<const void*> ptrs(polys.size());
    std::vector<size_t> lens(polys.size());
    size_t m = 0;
    for (size_t k = 0; k < polys.size(); ++k) {
        ptrs[k] = polys[k]->data();
        lens[k] = polys[k]->size();
        m = lens[k] > m ? lens[k] : m;
    }
    DeviceVec out(ctx, m);
    check(zk_poly_lincomb_dev(ctx.handle(), curve, (uint32_t)polys.size(), ptrs.data(), lens.data(), coeffs_mont.data(), out.data(), m),
          "zk_poly_lincomb_dev");
    return out;
}

}  // namespace zk
