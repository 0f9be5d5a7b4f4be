/* ark_plonk_amd.h -- C ABI of the MI355X-native NTT + MSM prover hot path.
 *
 * Drop-in boundary for heliaxdev/ark-plonk's proof-generation inner loop.
 * Every entry point names the reference interface it replaces (file:line relative to the
 * reference tree) and, where the arithmetic lives in a crates.io dependency, the crate item the
 * reference calls there.  Plain pointers and sizes only; no torch / HIP types in signatures
 * (a HIP stream is passed as an opaque void*).
 *
 * Data formats (identical bytes to arkworks 0.3 in-memory values):
 *   Fr element  : 4 x uint64 little-endian limbs, Montgomery form (value * 2^256 mod r), < r.
 *   Fr scalar   : 4 x uint64 little-endian limbs, canonical integer (PrimeField::into_repr), < r.
 *   Fq element  : L x uint64 limbs (L = 6 BLS12-381, 4 BN254), Montgomery form (R = 2^(64L)).
 *   G1 affine   : x || y (2L limbs, packed) (arkworks' GroupAffine is not repr(C): the shim copies x, y, infinity).
 *                 The point at infinity is accepted in three encodings: inf_flags[i] != 0 (x, y ignored),
 *                 x = y = 0, and GroupAffine::zero() = (0, 1) with 1 in Montgomery form.  (0, 1) is on neither
 *                 supported curve, so no finite point is lost.
 *   Output point: affine x || y Montgomery; infinity is (0, 1) + flag 1, as GroupAffine::zero() -- an output may be
 *                 fed back as a base with or without its flag.
 *
 * Ownership: the caller owns every buffer; nothing is retained after return except the device
 * copy held by a zk_srs handle.  Errors: 0 = success, negative = failure (never aborts, never
 * throws).  Threading: calls on one ctx are serialised internally; one HIP stream per ctx, and the stream is ctx
 * state (zk_ctx_set_stream), so threads that want different streams use one ctx each -- ctxs are cheap and share
 * SRS handles.  One ctx drives one GPU: zk_ctx_create(int device) instead of SURVEY.md 8b's
 * (const int* devices, int n_dev) -- one process per GPU; multi-GPU MSM = one ctx per rank + an all-gather of
 * zk_msm_g1_*_partial outputs, see INTEGRATION.md.
 * Host-pointer entry points (zk_ntt, zk_kzg_commit(_batch), zk_kzg_open, zk_msm_g1(_srs), zk_srs_register) accept
 * ordinary pageable memory (zk_ctx_set_staging selects how it is moved); zk_io_stats reports the bytes moved.
 */
#ifndef ARK_PLONK_AMD_H
#define ARK_PLONK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;
typedef struct zk_srs zk_srs;

/* curve ids */
#define ZK_CURVE_BLS12_381 0
#define ZK_CURVE_BN254 1

/* transform kinds == ark_poly::EvaluationDomain methods */
#define ZK_NTT_FFT 0        /* EvaluationDomain::fft        (permutation/mod.rs:671-674)            */
#define ZK_NTT_IFFT 1       /* EvaluationDomain::ifft       (prover.rs:196-203,240-242,281-283,302-305) */
#define ZK_NTT_COSET_FFT 2  /* EvaluationDomain::coset_fft  (quotient_poly.rs:72-120,205,294)       */
#define ZK_NTT_COSET_IFFT 3 /* EvaluationDomain::coset_ifft (quotient_poly.rs:175-177)              */

/* error codes */
#define ZK_OK 0
#define ZK_ERR_BAD_ARG (-1)
#define ZK_ERR_DOMAIN_TOO_LARGE (-2) /* log_n > two-adicity: plonk-core/src/error.rs:14-21 InvalidEvalDomainSize */
#define ZK_ERR_HIP (-3)
#define ZK_ERR_OOM (-4)
#define ZK_ERR_NO_DEVICE (-5)
#define ZK_ERR_UNSUPPORTED (-6)
#define ZK_ERR_NOT_INVERTIBLE (-7) /* a grand-product denominator is zero (the reference panics: `inverse().unwrap()`) */
#define ZK_ERR_NOT_INDEXED (-8)    /* a lookup query value is not in the table (the reference's Error::ElementNotIndexed) */
#define ZK_ERR_PENDING (-9)        /* a deferred round (zk_kzg_round_begin_dev) is open on this ctx: close it with zk_kzg_round_end first */

const char* zk_strerror(int code);

/* ---- context -------------------------------------------------------------------------------- */
/* Create a context on HIP device `device` (one process per GPU). */
int zk_ctx_create(int device, zk_ctx** out);
void zk_ctx_destroy(zk_ctx* ctx);
/* Run all work of this ctx on an existing HIP stream (hipStream_t as void*).  NULL is HIP's
 * default (null) stream -- e.g. torch's default stream -- not "none". */
int zk_ctx_set_stream(zk_ctx* ctx, void* hip_stream);
/* Go back to the ctx's own non-blocking stream (the state after zk_ctx_create). */
int zk_ctx_use_own_stream(zk_ctx* ctx);
/* Block until all queued work of this ctx is complete. */
int zk_ctx_sync(zk_ctx* ctx);
/* Override the MSM window size c of the per-window path (0 = automatic, else 2..16). Test/tuning hook. */
int zk_ctx_set_msm_window(zk_ctx* ctx, int c);
/* Host<->device bytes moved by the host-pointer entry points of this ctx since creation / the last reset. */
int zk_io_stats(zk_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes, int reset);
/* 0 (default): pageable host buffers are handed to hipMemcpyAsync as they are -- on the MI355X hosts that reaches the same
 * 56 GB/s as pinned memory; 1: through the ctx's pinned staging ring (49 GB/s there; for hosts whose runtime stages pageable
 * copies slowly).  Tuning hook, see tools/pcie_probe.py. */
int zk_ctx_set_staging(zk_ctx* ctx, int mode);
/* Tuning options of the MSM planner, per ctx, integers by name (SURVEY.md 5 "runtime ctx options"; until round 4 these were ZK_*
 * environment variables read inside the hot path).  Results are identical for every value; only launch shapes change.  Keys:
 *   "msm_merge"      1 (default): one sort / accumulation launch per round for all its deferred jobs; 0: per job at submission
 *   "pre_vw"         virtual windows of the shared-bucket reduction: 0 = default (64), else a power of two 8 .. 512
 *   "pre_logg"       log2 buckets per reduction segment: -1 = default, else 0 .. 5
 *   "chunk_l"        sorted references per accumulation lane: 0 = planned, else 8 .. 1024
 *   "long_rounds"    rounds of resident lanes given to a non-final job of a merged accumulation launch: 1 (default) .. 16
 *   "combine_sg"     lanes per small bucket in the combine kernel when a launch has more than two jobs: 0 = default (1), 2, 4
 *   "pre_max_log_n"  vectors longer than 2^value take the per-window path even over a table: 0 = built-in limit (2^26), 13 .. 25
 *                    (test hook; the ranks of a window-sharded MSM must agree on it, as on "pre_vw" / "pre_logg" for the
 *                    window-sum exchange below)
 *   "mem_reserve_mb"      device memory (MiB) the table path leaves alone when it decides whether another deferred job's buffer set
 *                         still fits (hipMemGetInfo): default 1024
 *   "round_mem_limit_mb"  test hook, 0 = off: the buffer sets of a round's QUEUED jobs may together hold at most this many MiB, so that
 *                         the early close described at zk_kzg_round_begin_dev can be exercised at small sizes
 *   "host_workers"        helper threads of the ctx's host pool (window-sum combine, affine normalisation, digests): default
 *                         min(15, hardware threads - 1); a launcher with several ranks per host passes cores / LOCAL_WORLD_SIZE - 1
 *   "cache_verify"        0 (default) | 1: every hit of the commitment cache is recomputed and every hit of the residency cache compared
 *                         byte for byte with the caller's vector before it is believed; mismatches are counted (zk_cache_verify_stats)
 *                         and the computed / uploaded value is used.  A diagnostic mode: it costs what the caches save
 * ZK_ERR_UNSUPPORTED: unknown key; ZK_ERR_BAD_ARG: value out of range; ZK_ERR_PENDING: a deferred round is open (a job's plan must
 * not change between its accumulation and its reduction).  The library reads NO environment variable on a compute path
 * (ZK_VERBOSE and ZK_HOST_TIMING switch diagnostics on stderr only). */
/* Residency cache of the HOST-POINTER entry points (opt-in, per ctx; round 5).  The reference's call structure sends the same
 * vector across PCIe again and again: an `ifft` output goes straight back up as a `PC::commit` input (prover.rs:196-213), then as a
 * `coset_fft` input (quotient_poly.rs:72-120), then into the last round's commitments and openings (prover.rs:569-618) -- about 58 of
 * the 83 vectors an unchanged Prover::prove uploads per proof.  With the cache on, zk_ntt keeps the device copy of every output of at
 * most max_vector_bytes (named by a keyed 256-bit digest of the bytes the caller receives) and zk_ntt / zk_kzg_commit_batch / zk_kzg_open
 * digest every input of at most that size on the ctx's host pool and use the resident copy on a match; a miss uploads into a fresh
 * entry, so a second use of any vector hits too (the prover key's sigma polynomials, proof after proof).  A digest runs at ~190 GB/s
 * against PCIe's 56, so a hit costs a third of an upload and a miss a third more (profiles/r05/r05_notes.md).  WHAT IS GUARANTEED: results
 * are identical to the uncached calls UNLESS two different vectors of one length collide under the keyed 256-bit non-cryptographic
 * digest (per-process key from the operating system) -- a hit is taken on equality of (length, digest) of the caller's CURRENT bytes,
 * the contents are not compared: the trust model of zk_srs_register's registry below.  Option "cache_verify" = 1 compares them (and
 * recomputes every commitment-cache hit) and counts mismatches: zk_cache_verify_stats.  Without operating-system entropy behind the
 * key (getrandom and /dev/urandom both unavailable) enabling either cache returns ZK_ERR_UNSUPPORTED.  capacity_bytes = device memory the entries may hold (0 = leave unchanged; default 2 GiB, least recently used evicted;
 * entries of the running call are never evicted); max_vector_bytes: 0 = leave unchanged (default 64 MiB: at n = 2^20 the n-sized
 * vectors are cached, the 4n-sized coset evaluations -- consumed by host code, never sent back -- are not).  enable = 0 drops every entry. */
int zk_ctx_set_residency_cache(zk_ctx* ctx, int enable, size_t capacity_bytes, size_t max_vector_bytes);
int zk_residency_cache_stats(zk_ctx* ctx, uint64_t* hits, uint64_t* misses, uint64_t* entries, uint64_t* bytes);
int zk_ctx_set_option(zk_ctx* ctx, const char* key, int64_t value);
int zk_ctx_get_option(zk_ctx* ctx, const char* key, int64_t* value);
/* hits checked under option "cache_verify" and how many of them did not hold (either pointer may be NULL) */
int zk_cache_verify_stats(zk_ctx* ctx, uint64_t* checked, uint64_t* mismatches);

/* Per-kernel HIP-event timing (bench.py roofline leg). on = 1: every launch of the hot kernels is bracketed by
 * hipEventRecord on the ctx stream (≈ 200 scopes, ≈ 2 ms per 2^20 proof); on = 2: only the dominant kernel, msm_accumulate
 * (58 scopes per proof) -- what bench.py keeps on inside its timed region; 0: off. */
int zk_profile_enable(zk_ctx* ctx, int on);
int zk_profile_reset(zk_ctx* ctx);
/* name: "ntt_pass", "msm_accumulate", "msm_sort", "msm_reduce", ...; returns total ms and launch count */
int zk_profile_get(zk_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);

/* ---- a1: GeneralEvaluationDomain::new + EvaluationDomainExt (util.rs:24-88) ------------------- */
typedef struct zk_domain_info {
    uint64_t size;               /* 2^log_n                                   */
    uint32_t log_size_of_group;  /* EvaluationDomainExt::log_size_of_group    */
    uint32_t reserved;
    uint64_t size_inv[4];        /* ::size_inv        (Montgomery)            */
    uint64_t group_gen[4];       /* ::group_gen                               */
    uint64_t group_gen_inv[4];   /* ::group_gen_inv                           */
    uint64_t generator[4];       /* F::multiplicative_generator() (coset g)   */
    uint64_t generator_inv[4];   /* ::generator_inv                           */
} zk_domain_info;
/* Replaces GeneralEvaluationDomain::<F>::new(num_coeffs) (prover.rs:169-173,
 * quotient_poly.rs:64-69): size = next_power_of_two(num_coeffs). */
int zk_domain_new(int curve_id, uint64_t num_coeffs, zk_domain_info* out);

/* ---- a2-a5: NTT ------------------------------------------------------------------------------ */
/* Host-buffer transform.  in: in_len (<= 2^log_n) Montgomery Fr elements, zero-extended;
 * out: 2^log_n elements, natural order.  in == out allowed.
 * Replaces ark_poly Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place.
 * Sizes: every log_n the reference admits (prover.rs:169-173, error.rs:14-21: up to the field's two-adicity, 32 for BLS12-381,
 * 28 for BN254; beyond it ZK_ERR_DOMAIN_TOO_LARGE, as Error::InvalidEvalDomainSize) -- one pass to 2^9, two to 2^18, three to 2^27,
 * four above.  What bounds the size in practice is memory: the vector, a work vector and the first pass boundary's twiddle table
 * are 2^log_n x 32 bytes each (8 GiB at 2^28); a size that does not fit the card returns ZK_ERR_OOM. */
int zk_ntt(zk_ctx* ctx, int curve_id, int kind, uint32_t log_n, const uint64_t* in, size_t in_len, uint64_t* out);
/* n_polys host-buffer transforms of one kind and size; ins[i] == outs[i] allowed. */
int zk_ntt_batch(zk_ctx* ctx, int curve_id, int kind, uint32_t log_n, uint32_t n_polys, const uint64_t* const* ins,
                 const size_t* in_lens, uint64_t* const* outs);
/* Same with device-resident buffers (async on the ctx stream).  d_in == d_out allowed. */
int zk_ntt_dev(zk_ctx* ctx, int curve_id, int kind, uint32_t log_n, const void* d_in, size_t in_len, void* d_out);
/* n_polys transforms of one kind/size sharing a plan (the 13 coset_fft of quotient_poly.rs:72-120). */
int zk_ntt_batch_dev(zk_ctx* ctx, int curve_id, int kind, uint32_t log_n, uint32_t n_polys,
                     const void* const* d_ins, const size_t* in_lens, void* const* d_outs);
/* Pre-build (and keep) the twiddle plan for (curve, log_n); otherwise built on first use. */
int zk_ntt_prepare(zk_ctx* ctx, int curve_id, uint32_t log_n);

/* Fr Montgomery <-> canonical on device (PrimeField::into_repr / from_repr; commitment.rs:37-40). */
int zk_fr_from_mont_dev(zk_ctx* ctx, int curve_id, const void* d_in, size_t n, void* d_out);
int zk_fr_to_mont_dev(zk_ctx* ctx, int curve_id, const void* d_in, size_t n, void* d_out);

/* ---- a6/a8: MSM ------------------------------------------------------------------------------ */
/* Replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul(bases, scalars) (commitment.rs:45,83)
 * followed by `.into()` affine.  n = min(len) pairs; inf_flags may be NULL. */
int zk_msm_g1(zk_ctx* ctx, int curve_id, const uint64_t* bases_xy, const uint8_t* inf_flags,
              const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf);

/* Device-resident SRS (CommitterKey::powers_of_g after PC::trim, circuit.rs:236,276).
 * A zk_srs belongs to the DEVICE of the registering ctx: every ctx of that device may use it (several proof streams
 * share one copy of the bases and of the window table).  It may outlive the ctx that registered it.
 * zk_srs_register is content-addressed: the reference trims on every gen_proof (circuit.rs:276), so registering
 * the same (curve, n, bases, flags) again returns the resident handle -- window table included -- after one pass of a
 * 256-bit digest over the host bytes, with no upload.  Handles are reference counted: call zk_srs_free once per
 * successful register.  An unreferenced cached SRS stays resident until zk_srs_cache_config's idle budget (default
 * 32 GiB) is exceeded, least recently used first.  zk_srs_register_dev (bases already on the device; d_inf_flags may
 * be NULL) is never cached.
 *
 * Trust model of the content-addressed caches (this one, zk_ctx_set_commit_cache, zk_ctx_set_residency_cache): a hit is accepted
 * on equality of a 256-bit digest, without comparing contents -- results are identical to the uncached call unless two inputs collide
 * under that digest.  The digests are KEYED with 256 bits drawn from the operating system once per process (the commitment cache
 * mixes in a per-ctx value); they are never exported, so a client that chooses the cached bytes cannot search for two inputs with
 * one digest offline; without operating-system entropy the registry shares nothing and the two ctx caches refuse to switch on.  The
 * mixing functions are fast NON-CRYPTOGRAPHIC ones (four xxhash-style lanes folded into each other every 256 bytes; a sum of
 * per-element keyed mixes on the device): a collision would return another SRS handle / another polynomial's
 * commitment -- an invalid proof that the verifier rejects, never an unsound one.  A service that must not even produce an
 * invalid proof for adversarial witnesses leaves the commitment cache off (the default) and registers its SRS from one
 * trusted source. */
int zk_srs_register(zk_ctx* ctx, int curve_id, const uint64_t* bases_xy, const uint8_t* inf_flags, size_t n, zk_srs** out);
int zk_srs_register_dev(zk_ctx* ctx, int curve_id, const void* d_bases_xy, const uint8_t* d_inf_flags, size_t n, zk_srs** out);
/* Optional: build the table of window multiples 2^(c w) * P_i (w = 1..W-1) for this SRS, W x its
 * size in HBM (1.9 GiB per 2^20 BLS12-381 points; the card has 288 GB).  MSMs over the SRS then use a
 * single bucket set: no per-window reduction and no host-side doublings.  Results are unchanged.  Idempotent.
 * Default window: c = 16 (16 rows) below 2^19 points, c = 17 from there on -- 15 rows for the 255-bit scalars of BLS12-381,
 * because a scalar k > (r - 1) / 2 is treated as -(r - k): one mixed addition per scalar fewer, 2^16 buckets instead of 2^15;
 * from 2^22 points on, c = 20 (13 rows, 2^19 buckets): two more additions per scalar saved outweigh the wider bucket reduction there
 * (the device-resident exchange below -- zk_kzg_round_end_winsums_dev -- needs window_bits <= 17: for an SRS or shard of 2^22 points
 * and more that is meant for it, call zk_srs_precompute_ex(ctx, srs, 17) instead; on a 20-bit table those calls return
 * ZK_ERR_UNSUPPORTED before anything is queued and the host form closes the round). */
int zk_srs_precompute(zk_ctx* ctx, zk_srs* srs);
/* Same with the table's window c chosen: 16 (default; 16 rows, 2^15 shared buckets) .. 21.  A larger window means fewer
 * rows (15 at c = 17, 13 at c = 20: mixed additions per scalar, table rows) but 2^(c-1) buckets to reduce.
 * 0 = default, or whatever table the SRS already holds.  The table belongs to the SRS, and a content-addressed
 * zk_srs_register hands every caller that registers the same bytes the SAME handle: the first precompute wins.  A later call
 * with window_bits = 0 (zk_srs_precompute) is a no-op; one that names a different window returns ZK_ERR_UNSUPPORTED and changes
 * nothing -- read the window in use with zk_srs_table_info.  Results never depend on the window. */
int zk_srs_precompute_ex(zk_ctx* ctx, zk_srs* srs, uint32_t window_bits);
/* window_bits / windows (= rows = mixed additions per scalar) of the SRS's table; both 0 without a table. */
int zk_srs_table_info(zk_srs* srs, uint32_t* window_bits, uint32_t* windows);
/* Multi-GPU, sharded by WINDOWS (BASELINE.json north_star: "the MSM shards its windows/buckets across GPUs"; SURVEY.md 8e's
 * alternative): rank g of G registers the WHOLE SRS and builds only the table rows of the windows first_window, first_window +
 * window_stride, ... (g, g + G, ...: ceil((W - g) / G) of the W rows), i.e. about 1/G of the table.  Every MSM entry point over such
 * an SRS then returns the rank's PARTIAL sum_i sum_{w owned} d_{i,w} 2^(c w) P_i -- all N scalars, the owned digits only, the rank's
 * own bucket set -- and the ranks' partials add up to the MSM (zk_g1_sum_partials*, after the same all-gather as the point-sharded
 * form: zk_kzg_round_end_partial, or the window-sum form on the device).  A vector too short for the table path is computed whole by the owner of window 0 and is
 * the point at infinity on the other ranks.  (first_window, window_stride) = (0, 1) is zk_srs_precompute_ex.  The first precompute
 * of an SRS wins, as above; a later call must name the same rows.  The plain SRS stays resident next to the rows. */
int zk_srs_precompute_rows(zk_ctx* ctx, zk_srs* srs, uint32_t window_bits, uint32_t first_window, uint32_t window_stride);
/* first_window / window_stride / rows held (0 / 1 / windows for a whole table; rows = 0 without a table). */
int zk_srs_table_rows(zk_srs* srs, uint32_t* first_window, uint32_t* window_stride, uint32_t* rows);
/* One more owner of a live handle (a second zk_ctx / thread that keeps using the SRS on its own): pairs with one more
 * zk_srs_free.  ZK_ERR_BAD_ARG for a handle whose last reference is gone. */
int zk_srs_retain(zk_srs* srs);
void zk_srs_free(zk_srs* srs);
size_t zk_srs_len(const zk_srs* srs);
/* SRS cache: bytes of unreferenced entries kept resident (0 = free on last zk_srs_free) / counters. */
int zk_srs_cache_config(size_t max_idle_bytes);
int zk_srs_cache_stats(uint64_t* hits, uint64_t* misses, uint64_t* entries, uint64_t* resident_bytes);

/* MSM over srs[base_offset .. base_offset+n) with host / device canonical scalars.
 * Sizes: the window-table path takes 2^13 <= n <= 2^26 points per MSM (a sorted reference holds 26 bits of point index); shorter and
 * longer vectors run the per-window path over the same SRS inside the same call (identical result, no table rows read), which
 * takes n * windows < 2^32 references: about 2^27 points at its 16-bit window.  Beyond that ZK_ERR_UNSUPPORTED (a 2^28-point SRS
 * is 24 GiB in the ABI form; the reference's largest bench is 2^18, benches/plonk.rs:95-103). */
int zk_msm_g1_srs(zk_ctx* ctx, zk_srs* srs, size_t base_offset, const uint64_t* scalars, size_t n,
                  uint64_t* out_xy, uint8_t* out_inf);
int zk_msm_g1_srs_dev(zk_ctx* ctx, zk_srs* srs, size_t base_offset, const void* d_scalars, size_t n,
                      uint64_t* out_xy, uint8_t* out_inf);
/* Multi-GPU leg: this rank's partial sum as a projective (X, Y, Z) Jacobian triple, 3L limbs,
 * Montgomery (Z = 0 for infinity) -- the 144-byte message all-gathered over RCCL. */
int zk_msm_g1_srs_partial_dev(zk_ctx* ctx, zk_srs* srs, size_t base_offset, const void* d_scalars, size_t n,
                              uint64_t* out_xyz);
/* Host: sum `count` Jacobian partials (count x 3L limbs) and normalise to affine. */
int zk_g1_sum_partials(int curve_id, const uint64_t* partials_xyz, size_t count, uint64_t* out_xy, uint8_t* out_inf);
/* Host: the all-gathered partials of one prover round, laid out [rank][job][3L]; job k's sum -> out_xy[k], out_inf[k]. */
int zk_g1_sum_partials_batch(int curve_id, const uint64_t* partials_xyz, size_t ranks, uint32_t n_jobs, uint64_t* out_xy,
                             uint8_t* out_inf);

/* ---- a6: KZG10 commit (PC::commit, prover.rs:213,289-291,312-317,361-363,387-389,459-469) ----- */
/* d_coeffs_mont: n Montgomery Fr coefficients on device.  into_repr + MSM over
 * srs[0..n) (leading zero coefficients contribute nothing, as kzg10::commit's stripping). */
int zk_kzg_commit_dev(zk_ctx* ctx, zk_srs* srs, const void* d_coeffs_mont, size_t n, uint64_t* out_xy, uint8_t* out_inf);
int zk_kzg_commit(zk_ctx* ctx, zk_srs* srs, const uint64_t* coeffs_mont, size_t n, uint64_t* out_xy, uint8_t* out_inf);
/* PC::commit(ck, polys, None) with the caller's host slices -- the call GpuKZG10::commit forwards its whole
 * `polys` iterator to (prover.rs:213 passes 4 polynomials, :579 and :606 seven; n_polys <= 16).  Polynomial k+1 is
 * uploaded while polynomial k's MSM runs; results equal n_polys calls of zk_kzg_commit. */
int zk_kzg_commit_batch(zk_ctx* ctx, zk_srs* srs, uint32_t n_polys, const uint64_t* const* coeffs_mont, const size_t* lens,
                        uint64_t* out_xy, uint8_t* out_inf);

/* N3: opt-in, per-ctx, content-addressed commitment cache.  The reference commits twelve polynomials
 * a second time in its last round (prover.rs:569-607: aw_polys / saw_polys repeat sigma_1..3, f, h_2, table, z, w_l,
 * w_r, w_4, h_1, z_2).  With the cache on, every zk_kzg_commit* / zk_kzg_*_batch* call that returns affine points
 * first computes a 256-bit digest of each coefficient vector on the device and serves (srs, input kind, length,
 * digest) hits from the cache: an unchanged Prover::prove runs 17 MSMs per proof instead of 29 (20 on the first: the
 * prover key's sigma commitments then stay cached across proofs); outputs identical unless two coefficient vectors collide under the
 * keyed digest (trust model: at zk_srs_register above; option "cache_verify").  capacity = entries kept (0 = leave unchanged; default
 * 64, least recently used dropped).  ZK_ERR_UNSUPPORTED: no operating-system entropy behind the digest key. */
int zk_ctx_set_commit_cache(zk_ctx* ctx, int enable, uint32_t capacity);
int zk_commit_cache_stats(zk_ctx* ctx, uint64_t* hits, uint64_t* misses, uint64_t* entries);

/* The commitments of one prover round (e.g. the 4 wire commits of prover.rs:213, the 7 + 7 of
 * prover.rs:569-607): n_polys (<= 16) device-resident coefficient vectors over one precomputed SRS.
 * Results are identical to n_polys calls of zk_kzg_commit_dev; the batch queues all device work back to
 * back and the host blocks once per result instead of once per call.
 * out_xy: n_polys x 2L limbs, out_inf: n_polys flags. */
int zk_kzg_commit_batch_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_polys, const void* const* d_coeffs_mont, const size_t* lens,
                            uint64_t* out_xy, uint8_t* out_inf);

/* Same with a per-job input kind: kinds[k] = 0 Montgomery coefficients (a commit), 1 canonical scalars
 * (e.g. an opening witness from zk_kzg_witness_dev).  kinds == NULL means all 0.  Lets the 16 independent
 * MSMs of the last prover round (7 commits + opening, 7 commits + opening; prover.rs:569-618) be one batch. */
int zk_kzg_round_batch_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens,
                           const uint8_t* kinds, uint64_t* out_xy, uint8_t* out_inf);

/* Multi-GPU form of the batch: this rank's Jacobian partials (n_polys x 3L limbs) over ITS shard of the
 * SRS, for coefficient slices that already are the rank's [lo, hi) ranges; one all-gather per round. */
int zk_kzg_commit_batch_partial_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_polys, const void* const* d_coeffs_mont,
                                    const size_t* lens, uint64_t* out_xyz);

/* zk_kzg_round_batch_dev for a sharded SRS: per-job input kinds as above, Jacobian partials out
 * (n_jobs x 3L limbs); the opening witnesses of prover.rs:582-618 are passed as this rank's slice. */
int zk_kzg_round_batch_partial_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens,
                                   const uint8_t* kinds, uint64_t* out_xyz);

/* Deferred form of zk_kzg_round_batch_dev: a round OPENED by one or more calls and CLOSED by one.
 * The reference issues commitments whose inputs do not depend on each other's results as separate blocking calls:
 * f | h_1 | h_2 (prover.rs:289-317: no challenge is drawn between them), z | z_2 (prover.rs:361-389: delta and epsilon are
 * drawn before z is committed), and the four calls of the last round (prover.rs:579-618: both opening challenges are drawn
 * with no transcript append in between).  Every blocking call pays the latency of the bucket reduction's dependent-addition
 * chains and a host round trip while the GPU idles.  zk_kzg_round_begin_dev queues sort + bucket accumulation of its jobs on
 * the ctx stream and returns; other work (the next polynomial's NTT) and further begins may follow; zk_kzg_round_end reduces
 * every queued job with ONE launch per reduction kernel, waits once, and returns all results in submission order (n_jobs must
 * equal the number of jobs begun, else ZK_ERR_BAD_ARG and the round stays open).  Results are identical to the blocking
 * calls.  At most 16 jobs per round, one SRS per round; the caller keeps the inputs and the SRS alive until the round ends.
 * While a round is open the blocking MSM / commit / open entry points of the same ctx return ZK_ERR_PENDING (NTTs and the
 * other device builders may run).  Jobs that do not take the window-table path (no table, fewer than 2^13 elements) and,
 * with the commitment cache on, all jobs are computed at begin.  zk_kzg_open_begin_dev is zk_kzg_open_dev as a job of the
 * round (witness polynomial built at once, its MSM deferred).  zk_kzg_round_end_partial returns Jacobian partials (n_jobs x
 * 3L limbs) for a sharded SRS; zk_kzg_round_abort drops an open round (waits for the queued kernels).
 * zk_kzg_round_reduce (optional) queues the round's reduction kernels and returns: the round takes no further jobs, and
 * zk_kzg_round_end then waits for those kernels only -- whatever the caller queues on the stream in between (transforms that
 * do not depend on this round's results: prover.rs would have the sigma ffts of permutation/mod.rs:671-674 behind f | h_1 |
 * h_2, the coset ffts of quotient_poly.rs:72-120 behind z | z_2) runs while the host combines the window sums, normalises
 * and hashes the results into the transcript, instead of the GPU idling through that. */
int zk_kzg_round_begin_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_jobs, const void* const* d_inputs, const size_t* lens,
                           const uint8_t* kinds);
int zk_kzg_open_begin_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                          const uint64_t* z_mont, const uint64_t* challenge_mont);
int zk_kzg_round_reduce(zk_ctx* ctx);
int zk_kzg_round_end(zk_ctx* ctx, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf);
int zk_kzg_round_end_partial(zk_ctx* ctx, uint32_t n_jobs, uint64_t* out_xyz);
int zk_kzg_round_pending(zk_ctx* ctx, uint32_t* n_jobs);
int zk_kzg_round_abort(zk_ctx* ctx);
/* Memory of a deferred round (round 6).  Every queued table-path job lives in a buffer set of its own until the round closes --
 * 4 B per (scalar, window) for its sorted references plus the chunk-edge partials and bucket arrays: 71 MB per job at n = 2^20, 2.5 GB at
 * 2^25 -- and sets are kept for the next round.  A begin that finds no room on the device for another set (hipMemGetInfo
 * minus option "mem_reserve_mb", or hipMalloc failing) CLOSES THE JOBS QUEUED SO FAR -- sort, accumulation, reduction, host combine, the
 * points parked in call order -- and reuses their sets; the round stays open and zk_kzg_round_end* returns exactly the points it would
 * have returned.  ZK_ERR_OOM is left for a size whose blocking form does not fit either (the reference itself has no size limit but
 * the field's two-adicity: plonk-core/src/error.rs:14-21).  zk_round_mem_stats: early closes so far, bytes held by the buffer sets, and
 * the device's free / total memory as the budget sees it (any pointer may be NULL). */
int zk_round_mem_stats(zk_ctx* ctx, uint64_t* early_closes, uint64_t* set_bytes, uint64_t* device_free, uint64_t* device_total);
/* Since round 4 a begin queues only the digit kernel of its jobs -- the one kernel that reads the caller's vectors, so the inputs are
 * consumed in stream order at the call, as before -- and zk_kzg_round_reduce / _end queue the rest for ALL jobs of the round as one
 * launch per kernel: the sort's placement passes, ONE accumulation launch (msm_accumulate_batch), the reductions.
 *
 * The multi-GPU exchange without a host hop (the reference is a single process).  A job's result is
 *   sum_v S_v + B_v * sum_v v * T_v
 * over the 2 VW virtual-window sums S_v | T_v its last reduction kernel writes (VW = 64 by default: 128 points in the library's internal
 * XYZZ limb form, opaque, 32 KiB for BLS12-381; every rank runs this library), and that expression is LINEAR in them.
 * zk_kzg_round_reduce_winsums_dev / zk_kzg_round_end_winsums_dev close the round with those sums -- not their combination -- left ON THE
 * DEVICE at d_out + k * zk_winsums_dev_bytes(ctx, srs), k = submission order, written by the reduction kernel the single-GPU path ends
 * with: no further dependent launch.  d_out is typically the send buffer of the collective (ncclAllGather / torch
 * all_gather_into_tensor on the same stream).  Neither call waits: d_out, the inputs and the SRS stay valid until the stream has passed
 * this point (zk_g1_sum_winsums_dev waits for it).  zk_kzg_round_reduce_winsums_dev is zk_kzg_round_reduce for this form (work queued
 * after it runs behind the reductions); zk_kzg_round_end_winsums_dev closes the round (queues the reductions itself unless reduce ran;
 * d_out must then be the same buffer).  All-zero sums are the point at infinity (a rank with an empty shard); a job computed at
 * submission or parked by the memory budget enters as S_0 = its point.  The ranks all-gather the buffers and zk_g1_sum_winsums_dev adds
 * the ranks' sums element-wise (one kernel of n_jobs * 2 VW independent quads, ranks - 1 additions each), waits once, and leaves the
 * one combine + inversion per job to the ctx's host pool exactly as zk_kzg_round_end does.  Same results.
 * ZK_ERR_UNSUPPORTED (returned before anything is queued: the round stays open, close it with zk_kzg_round_end_partial): no table, a
 * table with window_bits >= 18 -- the DEFAULT of zk_srs_precompute for whole tables of 2^22 points and more: pass window_bits = 17 to
 * zk_srs_precompute_ex / _rows for an SRS meant for this exchange -- and the commitment cache.  Every rank must run the same
 * table window and the same "pre_vw" / "pre_logg" options: zk_winsums_geometry returns {window_bits, windows of a full-width scalar, VW,
 * buckets per virtual window} for the caller to compare across ranks once (the Python schedule all-gathers it at construction and
 * refuses a mismatch); zk_winsums_dev_bytes is 0 where the form does not exist.
 * (Round 4's other device form -- ONE point per job, formed by a further dependent launch: zk_kzg_round_end_partial_dev /
 * zk_g1_sum_partials_dev -- measured last of the three exchanges and was retired in round 6: profiles/design_history_msm.md.) */
size_t zk_winsums_dev_bytes(zk_ctx* ctx, zk_srs* srs);
int zk_winsums_geometry(zk_ctx* ctx, zk_srs* srs, uint32_t out[4]);
int zk_kzg_round_reduce_winsums_dev(zk_ctx* ctx, void* d_out);
int zk_kzg_round_end_winsums_dev(zk_ctx* ctx, uint32_t n_jobs, void* d_out);
int zk_g1_sum_winsums_dev(zk_ctx* ctx, zk_srs* srs, const void* d_all, size_t ranks, uint32_t n_jobs, uint64_t* out_xy, uint8_t* out_inf);

/* ---- a7: KZG10 open (PC::open, prover.rs:582-591,609-618) ------------------------------------- */
/* p = sum_k challenge^k * polys[k]; witness = (p - p(z)) / (X - z); returns commit(witness).
 * polys: n_polys device pointers to Montgomery coefficient vectors of lens[k] elements. */
int zk_kzg_open_dev(zk_ctx* ctx, zk_srs* srs, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                    const uint64_t* z_mont, const uint64_t* challenge_mont, uint64_t* out_xy, uint8_t* out_inf);

/* Same with the caller's host slices (n_polys <= 16), as PC::open receives them. */
int zk_kzg_open(zk_ctx* ctx, zk_srs* srs, uint32_t n_polys, const uint64_t* const* polys_mont, const size_t* lens,
                const uint64_t* z_mont, const uint64_t* challenge_mont, uint64_t* out_xy, uint8_t* out_inf);

/* Only the CPU-side part of PC::open: writes the witness polynomial's max(len)-1 coefficients as
 * canonical scalars into d_out (device, max(len) x 4 limbs) and their count into *out_len; the caller
 * runs the opening MSM itself (e.g. sharded over GPUs with zk_msm_g1_srs_partial_dev). */
int zk_kzg_witness_dev(zk_ctx* ctx, int curve_id, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                       const uint64_t* z_mont, const uint64_t* challenge_mont, void* d_out, size_t* out_len);

/* ---- the O(n) work of the prover's last round (linearisation_poly.rs:164-350), device-resident ------------------ */
/* `DensePolynomial::evaluate(&point)` for a batch: out[k] = polys[k](points[k]), k < n_polys <= 32, one launch pair for
 * the batch.  The reference evaluates 16 polynomials at z and 7 at z*omega per proof (linearisation_poly.rs:203-261: the
 * ProofEvaluations of proof.rs:41-103).  polys: device pointers to Montgomery coefficient vectors of lens[k] elements
 * (lens[k] == 0 -> 0); points_mont: n_polys x 4 limbs (host), reduced; out_mont: n_polys x 4 limbs (host), Montgomery,
 * reduced.  Blocks until the values are in out_mont. */
int zk_poly_evaluate_dev(zk_ctx* ctx, int curve_id, uint32_t n_polys, const void* const* d_polys, const size_t* lens,
                         const uint64_t* points_mont, uint64_t* out_mont);

/* out[i] = sum_k coeffs[k] * polys[k][i] for i < out_len (a polynomial shorter than out_len counts as zero-extended, a longer
 * one is cut): the scalar-times-polynomial sums the reference builds with `&poly * scalar` and `+` -- the linearisation
 * polynomial (linearisation_poly.rs:288-336 with widget/arithmetic.rs:66-82, permutation.rs:156-291, widget/lookup.rs:154-203,
 * widget/mod.rs:96-104; 19 terms), the compressed lookup columns (sum zeta^k col_k).  n_terms <= 32; coeffs_mont: n_terms x
 * 4 limbs (host), reduced (pass the negated scalar for a subtracted term); d_out: device, out_len x 4 limbs, Montgomery,
 * reduced; may alias one of the inputs.  Queued on the ctx stream (no host synchronisation). */
int zk_poly_lincomb_dev(zk_ctx* ctx, int curve_id, uint32_t n_terms, const void* const* d_polys, const size_t* lens,
                        const uint64_t* coeffs_mont, void* d_out, size_t out_len);

/* ---- round 2 of the prover (prover.rs:228-317): the lookup query column and the sorted table/query halves ------------ */
/* prover.rs:244-279: out[i] = w_l[i] + zeta w_r[i] + zeta^2 w_o[i] + zeta^3 w_4[i] (`MultiSet::compress` of the four query
 * columns, util.rs `lc`) where q_lookup[i] != 0, and d_table_compressed[0] -- the first row of the compressed table, the
 * reference's dummy value -- where q_lookup[i] == 0 or i >= q_len (the reference pads q_lookup with zeros to n).
 * All vectors device-resident Montgomery, n rows; the compressed table itself is zk_poly_lincomb_dev over the four table
 * columns with coefficients 1, zeta, zeta^2, zeta^3 (prover.rs:229-237). */
int zk_lookup_query_dev(zk_ctx* ctx, int curve_id, size_t n, const void* d_q_lookup, size_t q_len, const void* const d_wires[4],
                        const uint64_t* zeta_mont, const void* d_table_compressed, void* d_out);

/* `MultiSet::combine_split` (lookup/multiset.rs:131-176), t.combine_split(&f): the values of t and f grouped by value in the
 * order of first appearance in t, every group written half to d_h1 ("evens") and half to d_h2 ("odds"), the odd element of
 * odd-sized groups alternately to h1 / h2 starting with h1.  d_t: n_t (< 2^24) elements, d_f: n_f elements; d_h1 / d_h2: room for
 * (n_t + n_f + 1) / 2 elements each; *len_h1 / *len_h2 receive the lengths (they differ by at most one).
 * ZK_ERR_NOT_INDEXED when f holds a value t does not (Error::ElementNotIndexed; nothing is written).  Values are compared as
 * 32-byte strings: pass reduced elements, as everywhere.  Blocks once (16-byte read-back of the lengths and the error flag). */
int zk_lookup_combine_split_dev(zk_ctx* ctx, int curve_id, const void* d_t, size_t n_t, const void* d_f, size_t n_f, void* d_h1, void* d_h2,
                                size_t* len_h1, size_t* len_h2);

/* ---- N2: grand-product builders, feeding the iNTT on device ------------------------ */
/* Evaluations of the permutation polynomial z over the size-2^log_n domain, i.e. everything
 * `Permutation::compute_permutation_poly` (permutation/mod.rs:652-752) does before its `domain.ifft`:
 *   out[0] = 1,  out[i+1] = out[i] * prod_k (w_k[i] + beta*K_k*omega^i + gamma) / prod_k (w_k[i] + beta*sigma_k[i] + gamma),
 * K = (1, 7, 13, 17) (permutation/constants.rs:12-22), the (n+1)-th value dropped as the reference does.
 * d_wires[k], d_sigmas[k] (k = 0..3): n Montgomery Fr evaluations each (sigma_k = domain.fft(sigma poly k),
 * mod.rs:671-676); beta, gamma: Montgomery Fr, 4 limbs, host.  d_out: n elements, may alias an input.
 * last_mont (optional, host, 4 limbs): the dropped value -- 1 for a satisfied permutation.
 * ZK_ERR_NOT_INVERTIBLE if a denominator is zero (the reference panics there). */
int zk_perm_product_dev(zk_ctx* ctx, int curve_id, uint32_t log_n, const void* const* d_wires, const void* const* d_sigmas,
                        const uint64_t* beta_mont, const uint64_t* gamma_mont, void* d_out, uint64_t* last_mont);
/* Same for the lookup product z2 (`compute_lookup_permutation_poly` + `lookup_ratio`, mod.rs:754-822):
 *   ratio_i = (1+d)(e + f_i)(e(1+d) + t_i + d t_{i+1}) / ((e(1+d) + h1_i + d h2_i)(e(1+d) + h2_i + d h1_{i+1})),
 * indices cyclic; d = delta, e = epsilon.  n need not be a power of two. */
int zk_lookup_product_dev(zk_ctx* ctx, int curve_id, size_t n, const void* d_f, const void* d_t, const void* d_h1, const void* d_h2,
                          const uint64_t* delta_mont, const uint64_t* epsilon_mont, void* d_out, uint64_t* last_mont);

/* ---- N1: pointwise quotient over the 4n coset ------------------------------------- */
/* Everything `quotient_poly::compute` (proof_system/quotient_poly.rs:34-178) does between its coset FFTs and
 * its final coset_ifft: out[i] = (gate_constraints[i] + permutation[i] + lookup[i]) / v_h_coset_4n[i] with all
 * widgets (arithmetic, range, logic, fixed-base scalar mul, curve addition), the permutation argument and the
 * lookup argument.  Every pointer is a device vector of 4n Montgomery Fr evaluations over the coset
 * g*<omega_4n> (the output of zk_ntt_dev kind 2); the "next row" of the reference (index i+4 of vectors it
 * extends by e[0..4], quotient_poly.rs:75-118) is taken cyclically.  Challenges / coefficients: Montgomery Fr. */
typedef struct zk_quotient_args {
    const void *w_l, *w_r, *w_o, *w_4;    /* wire polynomials */
    const void *z, *z2;                    /* permutation and lookup grand products */
    const void *f, *table, *h1, *h2;       /* lookup: query, compressed table, sorted halves */
    const void *pi;                        /* public inputs */
    const void *l1;                        /* first Lagrange polynomial (quotient_poly.rs:68-69) */
    /* prover key evaluations (preprocess.rs:144-212) */
    const void *q_m, *q_l, *q_r, *q_o, *q_4, *q_c, *q_arith;
    const void *q_range, *q_logic, *q_fixed_group_add, *q_variable_group_add, *q_lookup;
    const void* sigma[4];                  /* left, right, out, fourth */
    uint64_t alpha[4], beta[4], gamma[4], delta[4], epsilon[4], zeta[4];
    uint64_t range_challenge[4], logic_challenge[4], fixed_base_challenge[4], var_base_challenge[4], lookup_challenge[4];
    uint64_t coeff_a[4], coeff_d[4];       /* P::COEFF_A, P::COEFF_D of the embedded twisted Edwards curve */
} zk_quotient_args;
/* n = 2^log_n is the circuit domain size; vectors hold 4n elements.  d_out (4n elements) must not alias an input. */
int zk_quotient_evals_dev(zk_ctx* ctx, int curve_id, uint32_t log_n, const zk_quotient_args* args, void* d_out);

/* ---- N4: canonical wire formats and the transcript (host only, no GPU needed) ------------------ */
/* ark-serialize 0.3 CanonicalSerialize, as the reference applies it in transcript.rs:27-33 (`append`) and through the
 * derives on proof.rs:41-103 / linearisation_poly.rs:34-161 / widget/mod.rs:252-278:
 *   Fr        : zk_fr_serialized_size bytes (32), little-endian canonical (non-Montgomery) integer.
 *   G1Affine  : compressed = x as little-endian canonical integer in zk_g1_compressed_size bytes (48 BLS12-381,
 *               32 BN254) with SWFlags in the top bits of the LAST byte: 0x80 = y is the larger of (y, p - y),
 *               0x40 = infinity (x = 0).  Uncompressed = x then y (2 x that size), the infinity flag on y.
 * Deserialisation validates like ark: integer < modulus, point on the curve and in the prime-order subgroup, flag
 * combinations; any violation is ZK_ERR_BAD_ARG.  Field elements cross this API in Montgomery form as everywhere. */
size_t zk_fr_serialized_size(int curve_id);
size_t zk_g1_compressed_size(int curve_id);
int zk_fr_serialize(int curve_id, const uint64_t* fr_mont, uint8_t* out);
int zk_fr_deserialize(int curve_id, const uint8_t* in, uint64_t* fr_mont);
int zk_g1_serialize_compressed(int curve_id, const uint64_t* xy_mont, uint8_t inf, uint8_t* out);
int zk_g1_deserialize_compressed(int curve_id, const uint8_t* in, uint64_t* xy_mont, uint8_t* inf);
int zk_g1_serialize_uncompressed(int curve_id, const uint64_t* xy_mont, uint8_t inf, uint8_t* out);
int zk_g1_deserialize_uncompressed(int curve_id, const uint8_t* in, uint64_t* xy_mont, uint8_t* inf);

/* merlin 3.0 `Transcript` (STROBE-128 / Keccak-f[1600]) -- prover.rs:179 clones one and drives it through the proof. */
typedef struct zk_transcript zk_transcript;
zk_transcript* zk_transcript_new(const uint8_t* label, size_t label_len);            /* Transcript::new(label)   */
zk_transcript* zk_transcript_clone(const zk_transcript* t);                          /* prover.rs:179            */
void zk_transcript_free(zk_transcript* t);
int zk_transcript_append_message(zk_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
int zk_transcript_append_u64(zk_transcript* t, const uint8_t* label, size_t label_len, uint64_t v);
int zk_transcript_challenge_bytes(zk_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len);
/* plonk-core's TranscriptProtocol (transcript.rs:16-49): append(label, item) for the two item types the prover appends
 * (a commitment = compressed G1, prover.rs:217-220,294,320-321,366,472-475; a scalar, prover.rs:226,327-337,399-426,481,
 * 516-553), challenge_scalar (31 challenge bytes as a little-endian integer, transcript.rs:35-46) and
 * circuit_domain_sep (transcript.rs:45-48).  The order of labels over one proof is tabulated in INTEGRATION.md section 7
 * and replayed by ark_plonk_amd/transcript.py::ProverTranscript. */
int zk_transcript_append_fr(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, const uint64_t* fr_mont);
int zk_transcript_append_g1(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, const uint64_t* xy_mont, uint8_t inf);
int zk_transcript_challenge_scalar(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, uint64_t* fr_mont);
int zk_transcript_circuit_domain_sep(zk_transcript* t, uint64_t n);
/* prover.rs:182 `transcript.append(b"pi", self.cs.get_pi())`: PublicInputs = BTreeMap<usize, F> (pi.rs:28-36), i.e. u64 count,
 * then (u64 position, Fr) with strictly ascending positions (anything else is ZK_ERR_BAD_ARG). */
int zk_transcript_append_public_inputs(zk_transcript* t, int curve_id, const uint8_t* label, size_t label_len, const uint64_t* positions,
                                       const uint64_t* values_mont, size_t n);

/* `Proof<F, PC>` for PC = (Sonic)KZG10 (proof.rs:41-103): the serialised form is the fields in declaration order --
 * 13 commitments a b c d z f h_1 h_2 z_2 t_1 t_2 t_3 t_4 (compressed G1 each), aw_opening and saw_opening
 * (kzg10::Proof { w, random_v: None } = compressed G1 + one 0 byte), then ProofEvaluations (linearisation_poly.rs:34-161):
 * a b c d | left_sigma right_sigma out_sigma permutation | q_lookup z2_next h1 h1_next h2 f table table_next as Fr,
 * then custom_evals: u64 count, and per entry u64 label length + label bytes + Fr. */
#define ZK_PROOF_N_EVALS 16
typedef struct zk_proof {
    const uint64_t* commitments;     /* 13 x 2L limbs, Montgomery affine, the order above */
    const uint8_t* commitment_inf;   /* 13 flags */
    const uint64_t* openings;        /* 2 x 2L limbs: aw_opening.w, saw_opening.w */
    const uint8_t* opening_inf;      /* 2 flags */
    const uint64_t* evals;           /* ZK_PROOF_N_EVALS x 4 limbs, Montgomery, the order above */
    uint32_t n_custom_evals;
    const char* const* custom_labels; /* NUL-terminated */
    const uint64_t* custom_evals;    /* n_custom_evals x 4 limbs */
} zk_proof;
size_t zk_proof_serialized_size(int curve_id, uint32_t n_custom_evals, const uint32_t* label_lens);
int zk_proof_serialize(int curve_id, const zk_proof* proof, uint8_t* out, size_t cap, size_t* written);

/* ---- device self-test ------------------------------------------------------------------------------ */
/* Runs the quad-cooperative point arithmetic of the bucket-reduction kernels (csrc/ecq.cuh) against the
 * single-lane group law on n_quads point pairs incl. doubling, cancellation and infinity cases.
 * *mismatches = number of disagreeing pairs (0 expected); *case_mask (optional) = which cases failed. */
int zk_selftest_quad_dev(zk_ctx* ctx, int curve_id, uint32_t n_quads, uint32_t* mismatches, uint32_t* case_mask);

/* ---- utilities (synthetic SRS for tests/bench; stands in for PC::setup, out of scope) --------- */
/* out[i] = scalars[i] * G1 generator, affine Montgomery, device buffers. */
int zk_g1_fixed_base_batch_dev(zk_ctx* ctx, int curve_id, const void* d_scalars, size_t n, void* d_out_xy);
/* out[i] = a[i] * b[i] (Montgomery Fr, device) -- pointwise helper for property tests. */
int zk_fr_mul_dev(zk_ctx* ctx, int curve_id, const void* d_a, const void* d_b, size_t n, void* d_out);

/* Device memory helpers so a non-torch host (the Rust shim, C++ host layer) needs no HIP headers. */
int zk_dev_alloc(zk_ctx* ctx, size_t bytes, void** d_ptr);
int zk_dev_free(zk_ctx* ctx, void* d_ptr);
int zk_dev_upload(zk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int zk_dev_download(zk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* device -> device on the ctx stream, queued in order with the kernels (no host wait): e.g. the four quarters of the quotient
 * polynomial as vectors of their own (prover.rs:107-123 split_tx_poly) for a caller whose vector type has no sub-ranges */
int zk_dev_copy(zk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* Library build info: returns "gfx950" etc. */
const char* zk_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* ARK_PLONK_AMD_H */
