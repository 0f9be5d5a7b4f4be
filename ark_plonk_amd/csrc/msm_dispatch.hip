// Curve dispatch for the per-curve MSM objects (the msm_* units built with -DZK_CURVE_SEL=0/1: msm_common.cuh).
#include "ctx.h"

#define DECLS(sfx)                                                                                                    \
    int msm_run_dev##sfx(zk_ctx* c, const void* d_bases_xy, const void* d_scalars, size_t n, uint64_t* out_xyz);       \
    int msm_fixed_base_dev##sfx(zk_ctx* c, const void* d_scalars, size_t n, void* d_out_xy);                           \
    int g1_jacobian_to_affine_host##sfx(const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf);                      \
    int g1_sum_partials_host##sfx(const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf);         \
    int msm_convert_bases_dev##sfx(zk_ctx* c, const void* d_xy_sat, const uint8_t* d_inf, size_t n, void* d_out);      \
    size_t msm_point_bytes##sfx();                                                                                     \
    int msm_precompute_dev##sfx(zk_ctx* c, zk_srs* s, uint32_t window_bits, uint32_t w0, uint32_t wstep);                                  \
    int msm_run_pre_dev##sfx(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz);  \
    int msm_batch_pre_dev##sfx(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens, uint64_t* out_xyz, \
                               const uint8_t* kinds, uint64_t* out_xy, uint8_t* out_inf, const std::function<int(uint32_t)>* before_job); \
    int msm_batch_pre_begin_dev##sfx(zk_ctx* c, zk_srs* s, uint32_t slot0, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens, \
                                     const uint8_t* kinds, const std::function<int(uint32_t)>* before_job);                \
    int msm_batch_pre_reduce_dev##sfx(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_winsums); \
    bool msm_partial_dev_supported##sfx(zk_ctx* c, zk_srs* s, uint32_t* vw, uint32_t* vb);                             \
    int g1_sum_winsums_dev##sfx(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf); \
    size_t msm_partial_dev_bytes##sfx();                                                                               \
    void g1_jacobian_to_partial_host##sfx(const uint64_t* xyz, void* out);                                             \
    int msm_batch_pre_end_dev##sfx(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, uint64_t* out_xyz, \
                                   uint64_t* out_xy, uint8_t* out_inf);
DECLS(_c0)
DECLS(_c1)

int msm_run_dev(zk_ctx* c, int curve, const void* d_bases_xy, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    if (curve == ZK_CURVE_BLS12_381) return msm_run_dev_c0(c, d_bases_xy, d_scalars, n, out_xyz);
    if (curve == ZK_CURVE_BN254) return msm_run_dev_c1(c, d_bases_xy, d_scalars, n, out_xyz);
    return ZK_ERR_BAD_ARG;
}
int msm_fixed_base_dev(zk_ctx* c, int curve, const void* d_scalars, size_t n, void* d_out_xy) {
    if (curve == ZK_CURVE_BLS12_381) return msm_fixed_base_dev_c0(c, d_scalars, n, d_out_xy);
    if (curve == ZK_CURVE_BN254) return msm_fixed_base_dev_c1(c, d_scalars, n, d_out_xy);
    return ZK_ERR_BAD_ARG;
}
int g1_jacobian_to_affine_host(int curve, const uint64_t* xyz, uint64_t* out_xy, uint8_t* out_inf) {
    if (curve == ZK_CURVE_BLS12_381) return g1_jacobian_to_affine_host_c0(xyz, out_xy, out_inf);
    if (curve == ZK_CURVE_BN254) return g1_jacobian_to_affine_host_c1(xyz, out_xy, out_inf);
    return ZK_ERR_BAD_ARG;
}
int g1_sum_partials_host(int curve, const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    if (curve == ZK_CURVE_BLS12_381) return g1_sum_partials_host_c0(partials, count, out_xy, out_inf);
    if (curve == ZK_CURVE_BN254) return g1_sum_partials_host_c1(partials, count, out_xy, out_inf);
    return ZK_ERR_BAD_ARG;
}
int msm_convert_bases_dev(zk_ctx* c, int curve, const void* d_xy_sat, const uint8_t* d_inf, size_t n, void* d_out_internal) {
    if (curve == ZK_CURVE_BLS12_381) return msm_convert_bases_dev_c0(c, d_xy_sat, d_inf, n, d_out_internal);
    if (curve == ZK_CURVE_BN254) return msm_convert_bases_dev_c1(c, d_xy_sat, d_inf, n, d_out_internal);
    return ZK_ERR_BAD_ARG;
}
size_t msm_point_bytes(int curve) {
    if (curve == ZK_CURVE_BLS12_381) return msm_point_bytes_c0();
    if (curve == ZK_CURVE_BN254) return msm_point_bytes_c1();
    return 0;
}
int msm_precompute_dev(zk_ctx* c, zk_srs* s, uint32_t window_bits, uint32_t w0, uint32_t wstep) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_precompute_dev_c0(c, s, window_bits, w0, wstep);
    if (s->curve == ZK_CURVE_BN254) return msm_precompute_dev_c1(c, s, window_bits, w0, wstep);
    return ZK_ERR_BAD_ARG;
}
int msm_run_pre_dev(zk_ctx* c, zk_srs* s, size_t base_offset, const void* d_scalars, size_t n, uint64_t* out_xyz) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_run_pre_dev_c0(c, s, base_offset, d_scalars, n, out_xyz);
    if (s->curve == ZK_CURVE_BN254) return msm_run_pre_dev_c1(c, s, base_offset, d_scalars, n, out_xyz);
    return ZK_ERR_BAD_ARG;
}
int msm_batch_pre_dev(zk_ctx* c, zk_srs* s, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens, uint64_t* out_xyz,
                      const uint8_t* kinds, uint64_t* out_xy, uint8_t* out_inf, const std::function<int(uint32_t)>* before_job) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_batch_pre_dev_c0(c, s, n_polys, d_coeffs, lens, out_xyz, kinds, out_xy, out_inf, before_job);
    if (s->curve == ZK_CURVE_BN254) return msm_batch_pre_dev_c1(c, s, n_polys, d_coeffs, lens, out_xyz, kinds, out_xy, out_inf, before_job);
    return ZK_ERR_BAD_ARG;
}
int msm_batch_pre_begin_dev(zk_ctx* c, zk_srs* s, uint32_t slot0, uint32_t n_polys, const void* const* d_coeffs, const size_t* lens,
                            const uint8_t* kinds, const std::function<int(uint32_t)>* before_job) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_batch_pre_begin_dev_c0(c, s, slot0, n_polys, d_coeffs, lens, kinds, before_job);
    if (s->curve == ZK_CURVE_BN254) return msm_batch_pre_begin_dev_c1(c, s, slot0, n_polys, d_coeffs, lens, kinds, before_job);
    return ZK_ERR_BAD_ARG;
}
int msm_batch_pre_reduce_dev(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, void* const* d_winsums) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_batch_pre_reduce_dev_c0(c, s, n_jobs, slots, lens, d_winsums);
    if (s->curve == ZK_CURVE_BN254) return msm_batch_pre_reduce_dev_c1(c, s, n_jobs, slots, lens, d_winsums);
    return ZK_ERR_BAD_ARG;
}
bool msm_partial_dev_supported(zk_ctx* c, zk_srs* s, uint32_t* vw, uint32_t* vb) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_partial_dev_supported_c0(c, s, vw, vb);
    if (s->curve == ZK_CURVE_BN254) return msm_partial_dev_supported_c1(c, s, vw, vb);
    return false;
}
int g1_sum_winsums_dev(zk_ctx* c, zk_srs* s, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf) {
    if (s->curve == ZK_CURVE_BLS12_381) return g1_sum_winsums_dev_c0(c, s, d_all, ranks, n_jobs, out_xy, out_inf);
    if (s->curve == ZK_CURVE_BN254) return g1_sum_winsums_dev_c1(c, s, d_all, ranks, n_jobs, out_xy, out_inf);
    return ZK_ERR_BAD_ARG;
}
size_t msm_partial_dev_bytes(int curve) {
    if (curve == ZK_CURVE_BLS12_381) return msm_partial_dev_bytes_c0();
    if (curve == ZK_CURVE_BN254) return msm_partial_dev_bytes_c1();
    return 0;
}
int g1_jacobian_to_partial_host(int curve, const uint64_t* xyz, void* out) {
    if (curve == ZK_CURVE_BLS12_381) g1_jacobian_to_partial_host_c0(xyz, out);
    else if (curve == ZK_CURVE_BN254) g1_jacobian_to_partial_host_c1(xyz, out);
    else return ZK_ERR_BAD_ARG;
    return ZK_OK;
}
int msm_batch_pre_end_dev(zk_ctx* c, zk_srs* s, uint32_t n_jobs, const uint32_t* slots, const size_t* lens, uint64_t* out_xyz, uint64_t* out_xy,
                          uint8_t* out_inf) {
    if (s->curve == ZK_CURVE_BLS12_381) return msm_batch_pre_end_dev_c0(c, s, n_jobs, slots, lens, out_xyz, out_xy, out_inf);
    if (s->curve == ZK_CURVE_BN254) return msm_batch_pre_end_dev_c1(c, s, n_jobs, slots, lens, out_xyz, out_xy, out_inf);
    return ZK_ERR_BAD_ARG;
}
