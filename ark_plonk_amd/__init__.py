"""ark_plonk_amd -- MI355X-native NTT + MSM hot path of the ark-plonk prover (HIP, gfx950).

Host-side mirror of the reference interfaces for this path:
  Radix2EvaluationDomain / GeneralEvaluationDomain  <- ark_poly::EvaluationDomain
  VariableBaseMSM, CommitterKey (commit / open)     <- ark_ec::msm::VariableBaseMSM, KZG10 PC::commit/open
  permutation / quotient / lookup / linearisation    <- the O(n) bodies of prover rounds 2-5 (permutation/mod.rs, quotient_poly.rs,
                                                       lookup/multiset.rs, linearisation_poly.rs) on device-resident vectors
  transcript, prover                                 <- merlin + TranscriptProtocol, Proof bytes; Prover::prove_with_preprocessed end to end
All compute runs in libark_plonk_amd.so (C ABI: include/ark_plonk_amd.h); there is no CPU fallback.
"""
from .context import Context, default_context  # noqa: F401
from .curves import BLS12_381, BN254, get_curve  # noqa: F401
from .domain import GeneralEvaluationDomain, Radix2EvaluationDomain  # noqa: F401
from . import linearisation, lookup, permutation, prover, quotient, transcript  # noqa: F401
from . import _lib, msm  # noqa: F401
from .msm import (CommitterKey, G1Affine, VariableBaseMSM, kzg_witness, srs_cache_config, srs_cache_stats, sum_partials,  # noqa: F401
                  sum_partials_batch)

__all__ = [
    "Context", "default_context", "BLS12_381", "BN254", "get_curve", "GeneralEvaluationDomain",
    "Radix2EvaluationDomain", "CommitterKey", "G1Affine", "VariableBaseMSM", "kzg_witness", "sum_partials", "sum_partials_batch", "srs_cache_stats", "srs_cache_config", "permutation", "quotient", "lookup", "linearisation", "prover", "transcript",
]
