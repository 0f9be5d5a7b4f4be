// Header-only C++ host layer over the C ABI (include/ark_plonk_amd.h), mirroring the reference
// interfaces of the hot path with the same names and argument meaning:
//   zk::Radix2EvaluationDomain  <- ark_poly::EvaluationDomain (prover.rs:169-173,196-203; quotient_poly.rs:64-120)
//   zk::VariableBaseMSM         <- ark_ec::msm::VariableBaseMSM (commitment.rs:45)
//   zk::CommitterKey::commit    <- KZG10 PC::commit (prover.rs:213 ...)
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

  private:
    zk_ctx* h_ = nullptr;
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
    CommitterKey(Context& ctx, const std::vector<uint64_t>& powers_of_g, int curve = ZK_CURVE_BLS12_381) : ctx_(&ctx), curve_(curve) {
        check(zk_srs_register(ctx.handle(), curve, powers_of_g.data(), nullptr, powers_of_g.size() / (2 * fq_limbs(curve)), &h_),
              "zk_srs_register");
    }
    ~CommitterKey() { zk_srs_free(h_); }
    CommitterKey(const CommitterKey&) = delete;
    CommitterKey& operator=(const CommitterKey&) = delete;
    size_t size() const { return zk_srs_len(h_); }
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

}  // namespace zk
