"""ctypes binding of libark_plonk_amd.so (the C ABI in include/ark_plonk_amd.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module raises.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARK_PLONK_AMD_LIB: another build of the SAME HIP library (A/B measurements of kernel variants in one GPU session, tools/ab_bench.sh);
# there is still no CPU fallback behind it
LIB_PATH = os.environ.get("ARK_PLONK_AMD_LIB") or os.path.join(_HERE, "libark_plonk_amd.so")

ZK_OK = 0
ZK_ERR_BAD_ARG = -1
ZK_ERR_DOMAIN_TOO_LARGE = -2
ZK_ERR_HIP = -3
ZK_ERR_OOM = -4
ZK_ERR_NO_DEVICE = -5
ZK_ERR_UNSUPPORTED = -6
ZK_ERR_NOT_INVERTIBLE = -7
ZK_ERR_NOT_INDEXED = -8
ZK_ERR_PENDING = -9

c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
c_int = ctypes.c_int
c_u32 = ctypes.c_uint32
c_u64 = ctypes.c_uint64
u64p = ctypes.POINTER(ctypes.c_uint64)
u8p = ctypes.POINTER(ctypes.c_uint8)


class DomainInfo(ctypes.Structure):
    _fields_ = [
        ("size", c_u64),
        ("log_size_of_group", c_u32),
        ("reserved", c_u32),
        ("size_inv", c_u64 * 4),
        ("group_gen", c_u64 * 4),
        ("group_gen_inv", c_u64 * 4),
        ("generator", c_u64 * 4),
        ("generator_inv", c_u64 * 4),
    ]


ZK_PROOF_N_EVALS = 16


class ZkProof(ctypes.Structure):
    """zk_proof (proof_system/proof.rs:41-103)."""
    _fields_ = [
        ("commitments", c_void_p), ("commitment_inf", c_void_p), ("openings", c_void_p), ("opening_inf", c_void_p),
        ("evals", c_void_p), ("n_custom_evals", c_u32), ("custom_labels", ctypes.POINTER(ctypes.c_char_p)), ("custom_evals", c_void_p),
    ]


# every symbol include/ark_plonk_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "zk_strerror": (ctypes.c_char_p, [c_int]),
    "zk_build_info": (ctypes.c_char_p, []),
    "zk_ctx_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "zk_ctx_destroy": (None, [c_void_p]),
    "zk_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "zk_ctx_use_own_stream": (c_int, [c_void_p]),
    "zk_ctx_sync": (c_int, [c_void_p]),
    "zk_ctx_set_msm_window": (c_int, [c_void_p, c_int]),
    "zk_ctx_set_residency_cache": (c_int, [c_void_p, c_int, c_size_t, c_size_t]),
    "zk_residency_cache_stats": (c_int, [c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                         ctypes.POINTER(ctypes.c_uint64)]),
    "zk_ctx_set_option": (c_int, [c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "zk_ctx_get_option": (c_int, [c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "zk_cache_verify_stats": (c_int, [c_void_p, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "zk_profile_enable": (c_int, [c_void_p, c_int]),
    "zk_profile_reset": (c_int, [c_void_p]),
    "zk_profile_get": (c_int, [c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_u64)]),
    "zk_domain_new": (c_int, [c_int, c_u64, ctypes.POINTER(DomainInfo)]),
    "zk_ntt": (c_int, [c_void_p, c_int, c_int, c_u32, c_void_p, c_size_t, c_void_p]),
    "zk_ntt_dev": (c_int, [c_void_p, c_int, c_int, c_u32, c_void_p, c_size_t, c_void_p]),
    "zk_ntt_batch": (c_int, [c_void_p, c_int, c_int, c_u32, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t),
                             ctypes.POINTER(c_void_p)]),
    "zk_ntt_batch_dev": (c_int, [c_void_p, c_int, c_int, c_u32, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t),
                                 ctypes.POINTER(c_void_p)]),
    "zk_ntt_prepare": (c_int, [c_void_p, c_int, c_u32]),
    "zk_fr_from_mont_dev": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "zk_fr_to_mont_dev": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "zk_msm_g1": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "zk_srs_register": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    "zk_srs_register_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    "zk_srs_cache_config": (c_int, [c_size_t]),
    "zk_srs_cache_stats": (c_int, [ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "zk_io_stats": (c_int, [c_void_p, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), c_int]),
    "zk_ctx_set_staging": (c_int, [c_void_p, c_int]),
    "zk_ctx_set_commit_cache": (c_int, [c_void_p, c_int, c_u32]),
    "zk_commit_cache_stats": (c_int, [c_void_p, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "zk_kzg_commit_batch": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p]),
    "zk_kzg_open": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "zk_srs_precompute": (c_int, [c_void_p, c_void_p]),
    "zk_srs_precompute_ex": (c_int, [c_void_p, c_void_p, c_u32]),
    "zk_srs_table_info": (c_int, [c_void_p, ctypes.POINTER(c_u32), ctypes.POINTER(c_u32)]),
    "zk_srs_precompute_rows": (c_int, [c_void_p, c_void_p, c_u32, c_u32, c_u32]),
    "zk_srs_table_rows": (c_int, [c_void_p, ctypes.POINTER(c_u32), ctypes.POINTER(c_u32), ctypes.POINTER(c_u32)]),
    "zk_srs_retain": (c_int, [c_void_p]),
    "zk_srs_free": (None, [c_void_p]),
    "zk_srs_len": (c_size_t, [c_void_p]),
    "zk_msm_g1_srs": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "zk_msm_g1_srs_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "zk_msm_g1_srs_partial_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "zk_selftest_quad_dev": (c_int, [c_void_p, c_int, c_u32, c_void_p, c_void_p]),
    "zk_quotient_evals_dev": (c_int, [c_void_p, c_int, c_u32, c_void_p, c_void_p]),
    "zk_perm_product_dev": (c_int, [c_void_p, c_int, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "zk_lookup_product_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "zk_g1_sum_partials": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "zk_g1_sum_partials_batch": (c_int, [c_int, c_void_p, c_size_t, c_u32, c_void_p, c_void_p]),
    "zk_kzg_commit_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "zk_kzg_commit": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "zk_kzg_commit_batch_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p]),
    "zk_kzg_round_begin_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p]),
    "zk_kzg_open_begin_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p]),
    "zk_kzg_round_reduce": (c_int, [c_void_p]),
    "zk_kzg_round_end": (c_int, [c_void_p, c_u32, c_void_p, c_void_p]),
    "zk_kzg_round_end_partial": (c_int, [c_void_p, c_u32, c_void_p]),
    "zk_kzg_round_pending": (c_int, [c_void_p, ctypes.POINTER(c_u32)]),
    "zk_winsums_dev_bytes": (c_size_t, [c_void_p, c_void_p]),
    "zk_winsums_geometry": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_u32)]),
    "zk_kzg_round_reduce_winsums_dev": (c_int, [c_void_p, c_void_p]),
    "zk_kzg_round_end_winsums_dev": (c_int, [c_void_p, c_u32, c_void_p]),
    "zk_g1_sum_winsums_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_u32, c_void_p, c_void_p]),
    "zk_kzg_round_abort": (c_int, [c_void_p]),
    "zk_round_mem_stats": (c_int, [c_void_p, ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64), ctypes.POINTER(c_u64)]),
    "zk_kzg_round_batch_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p, c_void_p]),
    "zk_kzg_commit_batch_partial_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p]),
    "zk_kzg_round_batch_partial_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p]),
    "zk_kzg_open_dev": (c_int, [c_void_p, c_void_p, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "zk_kzg_witness_dev": (c_int, [c_void_p, c_int, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p,
                                   c_void_p, ctypes.POINTER(c_size_t)]),
    "zk_lookup_query_dev": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_size_t, ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p]),
    "zk_lookup_combine_split_dev": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p,
                                            ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "zk_poly_evaluate_dev": (c_int, [c_void_p, c_int, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p]),
    "zk_poly_lincomb_dev": (c_int, [c_void_p, c_int, c_u32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p, c_void_p, c_size_t]),
    "zk_g1_fixed_base_batch_dev": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "zk_fr_mul_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "zk_dev_alloc": (c_int, [c_void_p, c_size_t, ctypes.POINTER(c_void_p)]),
    "zk_dev_free": (c_int, [c_void_p, c_void_p]),
    "zk_dev_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "zk_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "zk_dev_copy": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    # N4: wire formats + transcript (host only)
    "zk_fr_serialized_size": (c_size_t, [c_int]),
    "zk_g1_compressed_size": (c_size_t, [c_int]),
    "zk_fr_serialize": (c_int, [c_int, c_void_p, ctypes.c_char_p]),
    "zk_fr_deserialize": (c_int, [c_int, ctypes.c_char_p, c_void_p]),
    "zk_g1_serialize_compressed": (c_int, [c_int, c_void_p, ctypes.c_uint8, ctypes.c_char_p]),
    "zk_g1_deserialize_compressed": (c_int, [c_int, ctypes.c_char_p, c_void_p, ctypes.POINTER(ctypes.c_uint8)]),
    "zk_g1_serialize_uncompressed": (c_int, [c_int, c_void_p, ctypes.c_uint8, ctypes.c_char_p]),
    "zk_g1_deserialize_uncompressed": (c_int, [c_int, ctypes.c_char_p, c_void_p, ctypes.POINTER(ctypes.c_uint8)]),
    "zk_transcript_new": (c_void_p, [ctypes.c_char_p, c_size_t]),
    "zk_transcript_clone": (c_void_p, [c_void_p]),
    "zk_transcript_free": (None, [c_void_p]),
    "zk_transcript_append_message": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, ctypes.c_char_p, c_size_t]),
    "zk_transcript_append_u64": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, c_u64]),
    "zk_transcript_challenge_bytes": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, ctypes.c_char_p, c_size_t]),
    "zk_transcript_append_fr": (c_int, [c_void_p, c_int, ctypes.c_char_p, c_size_t, c_void_p]),
    "zk_transcript_append_g1": (c_int, [c_void_p, c_int, ctypes.c_char_p, c_size_t, c_void_p, ctypes.c_uint8]),
    "zk_transcript_challenge_scalar": (c_int, [c_void_p, c_int, ctypes.c_char_p, c_size_t, c_void_p]),
    "zk_transcript_circuit_domain_sep": (c_int, [c_void_p, c_u64]),
    "zk_transcript_append_public_inputs": (c_int, [c_void_p, c_int, ctypes.c_char_p, c_size_t, c_void_p, c_void_p, c_size_t]),
    "zk_proof_serialized_size": (c_size_t, [c_int, c_u32, ctypes.POINTER(c_u32)]),
    "zk_proof_serialize": (c_int, [c_int, ctypes.POINTER(ZkProof), ctypes.c_char_p, c_size_t, ctypes.POINTER(c_size_t)]),
}

_lib = None


class ZkError(RuntimeError):
    def __init__(self, code: int, where: str = ""):
        self.code = code
        msg = lib().zk_strerror(code).decode() if _lib is not None else str(code)
        super().__init__(f"{where}: {msg} (code {code})" if where else f"{msg} (code {code})")


def lib():
    """Load the HIP library; raise loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m ark_plonk_amd.build` (hipcc, gfx950). "
                "ark_plonk_amd has no CPU fallback."
            )
        # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP
        # runtimes in one process leave the second without devices, so when torch is installed let
        # it load its runtime first; our library then binds to the copy already in the process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int, where: str = ""):
    if code != ZK_OK:
        raise ZkError(code, where)
