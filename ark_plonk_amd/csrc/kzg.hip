// KZG10 opening helpers (row a7): random linear combination of polynomials and the witness
// polynomial (p(X) - p(z)) / (X - z).  Replaces ark-poly-commit 0.3 `PolynomialCommitment::open`
// glue around the opening MSM (reference call sites: proof_system/prover.rs:582-591,609-618).
#include "ctx.h"

int kzg_open_prepare_dev(zk_ctx* c, int curve, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                         const uint64_t* z_mont, const uint64_t* chal_mont, void** d_witness_canonical, size_t* wlen) {
    (void)c; (void)curve; (void)n_polys; (void)d_polys; (void)lens; (void)z_mont; (void)chal_mont;
    (void)d_witness_canonical; (void)wlen;
    return ZK_ERR_UNSUPPORTED;
}
