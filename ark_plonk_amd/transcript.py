"""Host mirror of the reference's transcript and wire formats, over the C ABI.

Mirrors (same names and argument meaning):
  merlin::Transcript::{new, append_message, append_u64, challenge_bytes}        (transcript.rs:12, prover.rs:179)
  TranscriptProtocol::{append, challenge_scalar, circuit_domain_sep}             (plonk-core/src/transcript.rs:16-49)
  CanonicalSerialize for Fr / G1Affine / Proof                                  (proof_system/proof.rs:41-103)
  the order in which Prover::prove_with_preprocessed feeds the transcript        (proof_system/prover.rs:179-594)
so that a non-Rust caller of this library can derive the same challenges and emit the bytes the reference's
`Proof::deserialize` + `verify` accept.  Field elements are 4 x uint64 Montgomery limbs as everywhere in this package;
points are `G1Affine`.  All encoding / hashing runs in libark_plonk_amd.so (csrc/wire.hip), host only.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import check, lib
from .curves import get_curve
from .msm import G1Affine


def _b(x) -> bytes:
    return x if isinstance(x, (bytes, bytearray)) else str(x).encode()


def _fr_ptr(x):
    a = np.ascontiguousarray(x, dtype=np.uint64).reshape(4)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def fr_serialize(x, curve="bls12_381") -> bytes:
    cv = get_curve(curve)
    a, p = _fr_ptr(x)
    out = ctypes.create_string_buffer(lib().zk_fr_serialized_size(cv.curve_id))
    check(lib().zk_fr_serialize(cv.curve_id, p, out), "zk_fr_serialize")
    return out.raw


def fr_deserialize(data: bytes, curve="bls12_381") -> np.ndarray:
    cv = get_curve(curve)
    if len(data) != lib().zk_fr_serialized_size(cv.curve_id):
        raise ValueError("wrong length")
    out = np.zeros(4, dtype=np.uint64)
    check(lib().zk_fr_deserialize(cv.curve_id, bytes(data), out.ctypes.data_as(ctypes.c_void_p)), "zk_fr_deserialize")
    return out


def g1_serialize(pt: G1Affine, curve="bls12_381", compressed: bool = True) -> bytes:
    cv = get_curve(curve)
    xy = np.ascontiguousarray(pt.xy(), dtype=np.uint64)
    n = lib().zk_g1_compressed_size(cv.curve_id)
    out = ctypes.create_string_buffer(n if compressed else 2 * n)
    fn = lib().zk_g1_serialize_compressed if compressed else lib().zk_g1_serialize_uncompressed
    check(fn(cv.curve_id, xy.ctypes.data_as(ctypes.c_void_p), 1 if pt.infinity else 0, out), "zk_g1_serialize")
    return out.raw


def g1_deserialize(data: bytes, curve="bls12_381", compressed: bool = True) -> G1Affine:
    """Validates like ark: reduced x, on the curve, in the subgroup, legal flags; raises ZkError otherwise."""
    cv = get_curve(curve)
    n = lib().zk_g1_compressed_size(cv.curve_id)
    if len(data) != (n if compressed else 2 * n):
        raise ValueError("wrong length")
    L = cv.fq_limbs
    xy = np.zeros(2 * L, dtype=np.uint64)
    inf = ctypes.c_uint8(0)
    fn = lib().zk_g1_deserialize_compressed if compressed else lib().zk_g1_deserialize_uncompressed
    check(fn(cv.curve_id, bytes(data), xy.ctypes.data_as(ctypes.c_void_p), ctypes.byref(inf)), "zk_g1_deserialize")
    return G1Affine(xy[:L].copy(), xy[L:].copy(), inf.value != 0, cv.name)


class Transcript:
    """`merlin::Transcript` + plonk-core's `TranscriptProtocol`."""

    def __init__(self, label, curve="bls12_381", _handle=None):
        self.curve = get_curve(curve)
        if _handle is not None:
            self._h = _handle
            return
        lb = _b(label)
        self._h = lib().zk_transcript_new(lb, len(lb))
        if not self._h:
            raise MemoryError("zk_transcript_new")

    def clone(self) -> "Transcript":           # prover.rs:179: `self.preprocessed_transcript.clone()`
        h = lib().zk_transcript_clone(self._h)
        if not h:
            raise MemoryError("zk_transcript_clone")
        return Transcript(None, self.curve, _handle=h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().zk_transcript_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- merlin
    def append_message(self, label, message: bytes):
        lb, m = _b(label), bytes(message)
        check(lib().zk_transcript_append_message(self._h, lb, len(lb), m, len(m)), "zk_transcript_append_message")

    def append_u64(self, label, x: int):
        lb = _b(label)
        check(lib().zk_transcript_append_u64(self._h, lb, len(lb), int(x)), "zk_transcript_append_u64")

    def challenge_bytes(self, label, n: int) -> bytes:
        lb = _b(label)
        out = ctypes.create_string_buffer(n)
        check(lib().zk_transcript_challenge_bytes(self._h, lb, len(lb), out, n), "zk_transcript_challenge_bytes")
        return out.raw

    # -- TranscriptProtocol (transcript.rs:27-49)
    def append(self, label, item):
        """item: a G1Affine (a commitment) or 4 Montgomery limbs (a scalar)."""
        lb = _b(label)
        cid = self.curve.curve_id
        if isinstance(item, G1Affine):
            xy = np.ascontiguousarray(item.xy(), dtype=np.uint64)
            check(lib().zk_transcript_append_g1(self._h, cid, lb, len(lb), xy.ctypes.data_as(ctypes.c_void_p), 1 if item.infinity else 0),
                  "zk_transcript_append_g1")
        else:
            a, p = _fr_ptr(item)
            check(lib().zk_transcript_append_fr(self._h, cid, lb, len(lb), p), "zk_transcript_append_fr")

    def append_public_inputs(self, label, pi: dict):
        """`PublicInputs` = BTreeMap<usize, F> (pi.rs:28-36): {position: 4 Montgomery limbs}."""
        lb = _b(label)
        pos = np.array(sorted(pi), dtype=np.uint64)
        vals = np.zeros((len(pos), 4), dtype=np.uint64)
        for i, k in enumerate(sorted(pi)):
            vals[i] = np.asarray(pi[k], dtype=np.uint64).reshape(4)
        check(lib().zk_transcript_append_public_inputs(self._h, self.curve.curve_id, lb, len(lb), pos.ctypes.data_as(ctypes.c_void_p),
                                                       vals.ctypes.data_as(ctypes.c_void_p), len(pos)), "zk_transcript_append_public_inputs")

    def challenge_scalar(self, label) -> np.ndarray:
        lb = _b(label)
        out = np.zeros(4, dtype=np.uint64)
        check(lib().zk_transcript_challenge_scalar(self._h, self.curve.curve_id, lb, len(lb), out.ctypes.data_as(ctypes.c_void_p)),
              "zk_transcript_challenge_scalar")
        return out

    def circuit_domain_sep(self, n: int):
        check(lib().zk_transcript_circuit_domain_sep(self._h, int(n)), "zk_transcript_circuit_domain_sep")


# `VerifierKey::seed_transcript` (proof_system/widget/mod.rs:252-278): the 15 commitments appended under these labels, in this order
# (q_lookup and the four table commitments are part of the key but are NOT appended), then circuit_domain_sep(n).
VK_SEED_LABELS = ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "q_range", "q_logic", "q_variable_group_add", "q_fixed_group_add",
                  "left_sigma", "right_sigma", "out_sigma", "fourth_sigma")


def seed_transcript(t: "Transcript", vk: dict, n: int) -> "Transcript":
    """vk: name -> G1Affine for every name in VK_SEED_LABELS.  Returns t (the `preprocessed_transcript` of prover.rs:179)."""
    for lb in VK_SEED_LABELS:
        t.append(lb, vk[lb])
    t.circuit_domain_sep(n)
    return t


# The prover's feed order (proof_system/prover.rs line numbers).  Challenges are drawn under `draw` and appended back
# under `put` -- the reference spells two of the put-labels "seperation" (prover.rs:403,407).
ROUND1_COMMITS = ("w_l", "w_r", "w_o", "w_4")                                   # :217-220
ROUND2_COMMITS = ("f", "h1", "h2")                                              # :294,320-321
ROUND2_CHALLENGES = ("beta", "gamma", "delta", "epsilon")                       # :326-337
ROUND4_CHALLENGES = (("alpha", "alpha"),                                        # :398-399
                     ("range separation challenge", "range seperation challenge"),            # :402-403
                     ("logic separation challenge", "logic seperation challenge"),            # :406-407
                     ("fixed base separation challenge", "fixed base separation challenge"),  # :410-415
                     ("variable base separation challenge", "variable base separation challenge"),   # :417-422
                     ("lookup separation challenge", "lookup separation challenge"))          # :424-426
QUOTIENT_COMMITS = ("t_1", "t_2", "t_3", "t_4")                                 # :472-475
EVAL_LABELS = ("a_eval", "b_eval", "c_eval", "d_eval", "left_sig_eval", "right_sig_eval", "out_sig_eval", "perm_eval",
               "f_eval", "q_lookup_eval", "lookup_perm_eval", "h_1_eval", "h_1_next_eval", "h_2_eval")   # :516-544
# labels of `CustomEvaluations.vals`, in push order (linearisation_poly.rs:243-253: `label_eval!` = the identifier's name)
CUSTOM_EVAL_LABELS = ("q_arith_eval", "q_c_eval", "q_l_eval", "q_r_eval", "a_next_eval", "b_next_eval", "d_next_eval")
# order of the 16 fixed evaluations inside the serialised Proof (linearisation_poly.rs:34-104)
PROOF_EVAL_FIELDS = ("a_eval", "b_eval", "c_eval", "d_eval", "left_sigma_eval", "right_sigma_eval", "out_sigma_eval", "permutation_eval",
                     "q_lookup_eval", "z2_next_eval", "h1_eval", "h1_next_eval", "h2_eval", "f_eval", "table_eval", "table_next_eval")
PROOF_COMMITMENTS = ("a_comm", "b_comm", "c_comm", "d_comm", "z_comm", "f_comm", "h_1_comm", "h_2_comm", "z_2_comm",
                     "t_1_comm", "t_2_comm", "t_3_comm", "t_4_comm")             # proof.rs:62-99


class ProverTranscript:
    """Replays `Prover::prove_with_preprocessed`'s transcript traffic round by round: each method takes what the
    prover has just computed, appends it under the reference's labels and returns the challenges the reference draws
    next.  `preprocessed` is the transcript after the verifier key was seeded into it (prover.rs:179 clones it)."""

    def __init__(self, preprocessed: Transcript):
        self.t = preprocessed.clone()

    def public_inputs(self, pi: dict):                                           # :182
        self.t.append_public_inputs("pi", pi)

    def _draw(self, draw, put):
        c = self.t.challenge_scalar(draw)
        self.t.append(put, c)
        return c

    def round1(self, w_commits):                                                  # :217-226
        for lb, cm in zip(ROUND1_COMMITS, w_commits):
            self.t.append(lb, cm)
        return {"zeta": self._draw("zeta", "zeta")}

    def round2(self, f_commit, h1_commit, h2_commit):                             # :294,320-337
        for lb, cm in zip(ROUND2_COMMITS, (f_commit, h1_commit, h2_commit)):
            self.t.append(lb, cm)
        return {k: self._draw(k, k) for k in ROUND2_CHALLENGES}

    def round3(self, z_commit):                                                   # :366 (the z_2 commitment of :387-389 is never appended)
        self.t.append("z", z_commit)
        return {draw: self._draw(draw, put) for draw, put in ROUND4_CHALLENGES}  # :398-426

    def round4(self, t_commits):                                                  # :472-481
        for lb, cm in zip(QUOTIENT_COMMITS, t_commits):
            self.t.append(lb, cm)
        return {"z": self._draw("z", "z")}

    def round5(self, evals: dict, custom_evals):                                  # :516-563,593-594
        """evals: the 14 values of EVAL_LABELS by label; custom_evals: [(label, value)] in vector order.
        Returns (aw_challenge, saw_challenge): both are drawn with nothing appended in between."""
        for lb in EVAL_LABELS:
            self.t.append(lb, evals[lb])
        for lb, v in custom_evals:
            self.t.append(lb, v)
        aw = self.t.challenge_scalar("aggregate_witness")
        saw = self.t.challenge_scalar("aggregate_witness")
        return aw, saw


def proof_serialize(commitments, openings, evals, custom_evals=(), curve="bls12_381") -> bytes:
    """`Proof::serialize` (proof.rs:41-103).  commitments: 13 G1Affine in PROOF_COMMITMENTS order; openings: the two
    kzg10::Proof witnesses (aw, saw); evals: 16 Montgomery scalars in PROOF_EVAL_FIELDS order; custom_evals: [(label, value)]."""
    from . import _lib

    cv = get_curve(curve)
    L = cv.fq_limbs
    if len(commitments) != 13 or len(openings) != 2 or len(evals) != _lib.ZK_PROOF_N_EVALS:
        raise ValueError("a Proof holds 13 commitments, 2 openings and 16 fixed evaluations")
    cm = np.ascontiguousarray(np.stack([p.xy() for p in commitments]), dtype=np.uint64)
    cm_inf = np.array([1 if p.infinity else 0 for p in commitments], dtype=np.uint8)
    op = np.ascontiguousarray(np.stack([p.xy() for p in openings]), dtype=np.uint64)
    op_inf = np.array([1 if p.infinity else 0 for p in openings], dtype=np.uint8)
    ev = np.ascontiguousarray(np.stack([np.asarray(e, dtype=np.uint64).reshape(4) for e in evals]), dtype=np.uint64)
    k = len(custom_evals)
    labels = [_b(lb) for lb, _ in custom_evals]
    lab_arr = (ctypes.c_char_p * max(k, 1))(*labels) if k else (ctypes.c_char_p * 1)()
    cv_arr = (np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.uint64).reshape(4) for _, v in custom_evals]), dtype=np.uint64)
              if k else np.zeros((1, 4), dtype=np.uint64))
    pr = _lib.ZkProof(cm.ctypes.data_as(ctypes.c_void_p), cm_inf.ctypes.data_as(ctypes.c_void_p), op.ctypes.data_as(ctypes.c_void_p),
                      op_inf.ctypes.data_as(ctypes.c_void_p), ev.ctypes.data_as(ctypes.c_void_p), k,
                      ctypes.cast(lab_arr, ctypes.POINTER(ctypes.c_char_p)), cv_arr.ctypes.data_as(ctypes.c_void_p))
    lens = (ctypes.c_uint32 * max(k, 1))(*[len(lb) for lb in labels]) if k else (ctypes.c_uint32 * 1)()
    cap = lib().zk_proof_serialized_size(cv.curve_id, k, lens)
    out = ctypes.create_string_buffer(cap)
    written = ctypes.c_size_t()
    check(lib().zk_proof_serialize(cv.curve_id, ctypes.byref(pr), out, cap, ctypes.byref(written)), "zk_proof_serialize")
    assert L and written.value == cap
    return out.raw[: written.value]
