"""CPU suite for SURVEY.md 8f row N4: canonical wire formats (ark-serialize 0.3), the merlin transcript and the prover's
label schedule, through the C ABI (csrc/wire.hip; host only) against (1) merlin's published conformance vectors,
(2) the pure-Python restatement oracle/wire_oracle.py and (3) the committed fixtures tests/golden/wire.json."""
import json
import os

import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd import _lib
from ark_plonk_amd import transcript as tr
from oracle import bigint_oracle as bo
from oracle import wire_oracle as wo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wire():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "wire.json")))


def fr_limbs(cv, v):
    return zk.curves.fr_to_mont(cv.curve_id, [v])[0]


def point(cv, P):
    L = cv.fq_limbs
    if P is None:
        one = zk.curves.fq_to_mont(cv.curve_id, [1])[0]
        return zk.G1Affine(np.zeros(L, dtype=np.uint64), one, True, cv.name)
    x, y = zk.curves.fq_to_mont(cv.curve_id, [P[0], P[1]])
    return zk.G1Affine(x, y, False, cv.name)


# merlin's conformance vectors (merlin/src/transcript.rs tests `equivalence_simple` / `equivalence_complex` as quoted by the
# Go / C ports' test suites: the expected challenge bytes for protocol label "test protocol")
SIMPLE = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
COMPLEX = "a8c933f54fae76e3f9bea93648c1308e7dfa2152dd51674ff3ca438351cf003c"


def _simple(T):
    t = T(b"test protocol")
    t.append_message(b"some label", b"some data")
    return t.challenge_bytes(b"challenge", 32).hex()


def _complex(T):
    t = T(b"test protocol")
    t.append_message(b"step1", b"some data")
    data = bytes([99]) * 1024
    ch = b""
    for _ in range(32):
        ch = t.challenge_bytes(b"challenge", 32)
        t.append_message(b"bigdata", data)
        t.append_message(b"challengedata", ch)
    return ch.hex()


def test_merlin_conformance_vectors_library_and_oracle():
    assert _simple(tr.Transcript) == SIMPLE and _simple(wo.Transcript) == SIMPLE
    assert _complex(tr.Transcript) == COMPLEX and _complex(wo.Transcript) == COMPLEX


def test_keccak_f1600_zero_state_known_answer():
    # first lane of Keccak-f[1600] applied to the all-zero state (the permutation's published test vector)
    assert wo.keccak_f1600([0] * 25)[0] == 0xF1258F7940E1DDE7


def test_transcript_clone_and_long_messages():
    a = tr.Transcript(b"x")
    a.append_message(b"l", bytes(range(256)) * 3)     # crosses several 166-byte STROBE blocks
    b = a.clone()
    o = wo.Transcript(b"x")
    o.append_message(b"l", bytes(range(256)) * 3)
    exp = o.challenge_bytes(b"c", 200)
    assert a.challenge_bytes(b"c", 200) == exp and b.challenge_bytes(b"c", 200) == exp
    a.append_u64(b"n", 2 ** 63 + 5)
    o.append_u64(b"n", 2 ** 63 + 5)
    assert a.challenge_bytes(b"d", 7) == o.challenge_bytes(b"d", 7)
    assert b.challenge_bytes(b"d", 7) != a.challenge_bytes(b"d", 7)


@pytest.mark.parametrize("cid", [0, 1])
def test_fr_encoding(cid, wire):
    cv = bo.CURVES[cid]
    w = wire[cv.name]["fr"]
    assert _lib.lib().zk_fr_serialized_size(cid) == 32
    for name, e in w.items():
        v = int(e["value"], 16)
        got = tr.fr_serialize(fr_limbs(cv, v), cid)
        assert got.hex() == e["bytes"] == wo.ser_fr(cv, v).hex(), name
        assert np.array_equal(tr.fr_deserialize(got, cid), fr_limbs(cv, v))
    # non-canonical integers are rejected (ark: SerializationError::InvalidData)
    for bad in (cv.r, cv.r + 1, (1 << 256) - 1):
        with pytest.raises(_lib.ZkError):
            tr.fr_deserialize(bad.to_bytes(32, "little"), cid)


@pytest.mark.parametrize("cid", [0, 1])
def test_g1_encoding(cid, wire):
    cv = bo.CURVES[cid]
    n = _lib.lib().zk_g1_compressed_size(cid)
    assert n == (48 if cid == 0 else 32)
    for name, e in wire[cv.name]["g1"].items():
        P = None if e["x"] is None else (int(e["x"], 16), int(e["y"], 16))
        pt = point(cv, P)
        c = tr.g1_serialize(pt, cid)
        u = tr.g1_serialize(pt, cid, compressed=False)
        assert c.hex() == e["compressed"] == wo.ser_g1(cv, P).hex(), name
        assert u.hex() == e["uncompressed"] == wo.ser_g1_uncompressed(cv, P).hex(), name
        back = tr.g1_deserialize(c, cid)
        back_u = tr.g1_deserialize(u, cid, compressed=False)
        assert back == pt and back_u == pt, name
        assert wo.de_g1(cv, c) == P
    # the three encodings of infinity a caller may hold all serialise to the flag form
    L = cv.fq_limbs
    zero = np.zeros(L, dtype=np.uint64)
    one = zk.curves.fq_to_mont(cid, [1])[0]
    exp = wire[cv.name]["g1"]["infinity"]["compressed"]
    assert tr.g1_serialize(zk.G1Affine(zero, zero, False, cv.name), cid).hex() == exp
    assert tr.g1_serialize(zk.G1Affine(zero, one, False, cv.name), cid).hex() == exp
    assert tr.g1_serialize(point(cv, (cv.gx, cv.gy)).__class__(zk.curves.fq_to_mont(cid, [cv.gx])[0], zk.curves.fq_to_mont(cid, [cv.gy])[0],
                                                                  True, cv.name), cid).hex() == exp


@pytest.mark.parametrize("cid", [0, 1])
def test_g1_deserialize_rejects_invalid(cid):
    cv = bo.CURVES[cid]
    n = wo.fq_flag_bytes(cv)
    # an x that is not on the curve
    x = 5
    while pow((x ** 3 + cv.b) % cv.q, (cv.q - 1) // 2, cv.q) == 1:
        x += 1
    with pytest.raises(_lib.ZkError):
        tr.g1_deserialize(x.to_bytes(n, "little"), cid)
    # both flag bits set; x >= q
    g = bytearray(wo.ser_g1(cv, (cv.gx, cv.gy)))
    g[-1] |= 0xC0
    with pytest.raises(_lib.ZkError):
        tr.g1_deserialize(bytes(g), cid)
    # ... in the uncompressed form as well (SWFlags::from_u8 returns None for it; the flags sit on y's last byte)
    gu = bytearray(tr.g1_serialize(point(cv, (cv.gx, cv.gy)), cid, compressed=False))
    assert tr.g1_deserialize(bytes(gu), cid, compressed=False) == point(cv, (cv.gx, cv.gy))
    gu[-1] |= 0xC0
    with pytest.raises(_lib.ZkError):
        tr.g1_deserialize(bytes(gu), cid, compressed=False)
    with pytest.raises(_lib.ZkError):
        tr.g1_deserialize(cv.q.to_bytes(n, "little"), cid)
    if cid == 0:
        # on the curve but outside the prime-order subgroup (BLS12-381 G1 has cofactor > 1; BN254's is 1)
        x = 1
        while True:
            rhs = (x ** 3 + cv.b) % cv.q
            y = pow(rhs, (cv.q + 1) // 4, cv.q)
            if y * y % cv.q == rhs and bo.ec_mul(cv, cv.r, (x, y)) is not None:
                break
            x += 1
        with pytest.raises(_lib.ZkError):
            tr.g1_deserialize(wo.ser_g1(cv, (x, y)), cid)


@pytest.mark.parametrize("cid", [0, 1])
def test_zero_public_inputs_are_not_part_of_the_message(cid):
    """`PublicInputs::insert` (pi.rs:56-65) drops zero values, so the reference's BTreeMap -- the prover's and the one the verifier
    rebuilds -- never holds one: a caller's explicit zero must hash like an absent position."""
    cv = bo.CURVES[cid]
    with_zero = {2: 0, 5: 77, 9: cv.r - 1, 11: 0}
    without = {5: 77, 9: cv.r - 1}
    assert wo.ser_public_inputs(cv, with_zero) == wo.ser_public_inputs(cv, without)
    outs = []
    for pi in (with_zero, without):
        t = tr.Transcript(b"pi-test", cid)
        t.append_public_inputs("pi", {k: fr_limbs(cv, v) for k, v in pi.items()})
        outs.append(t.challenge_scalar("c").tobytes())
    ref = wo.PlonkTranscript(b"pi-test", cv)
    ref.append_message(b"pi", wo.ser_public_inputs(cv, with_zero))
    assert outs[0] == outs[1]
    assert outs[0] == fr_limbs(cv, ref.challenge_scalar(b"c")).tobytes()
    # all-zero map: count 0, nothing else
    assert wo.ser_public_inputs(cv, {4: 0}) == (0).to_bytes(8, "little")


def _replay(cv, w):
    """Drive ark_plonk_amd.transcript.ProverTranscript with the fixture's values; returns (challenges, proof bytes)."""
    cid = cv.curve_id
    T = w["transcript"]
    pre = tr.Transcript(T["label"].encode(), cid)
    pre.circuit_domain_sep(T["n"])
    pt = tr.ProverTranscript(pre)
    cm = {k: point(cv, None if v is None else (int(v[0], 16), int(v[1], 16))) for k, v in T["commitments"].items()}
    ev = {k: fr_limbs(cv, int(v, 16)) for k, v in T["evals"].items()}
    custom = [(lb, fr_limbs(cv, int(v, 16))) for lb, v in T["custom"]]
    pt.public_inputs({int(k): fr_limbs(cv, int(v, 16)) for k, v in T["pi"].items()})
    ch = {}
    ch.update(pt.round1([cm["a"], cm["b"], cm["c"], cm["d"]]))
    ch.update(pt.round2(cm["f"], cm["h1"], cm["h2"]))
    ch.update(pt.round3(cm["z"]))
    ch.update(pt.round4([cm["t1"], cm["t2"], cm["t3"], cm["t4"]]))
    feed = {"a_eval": "a_eval", "b_eval": "b_eval", "c_eval": "c_eval", "d_eval": "d_eval", "left_sig_eval": "left_sigma_eval",
            "right_sig_eval": "right_sigma_eval", "out_sig_eval": "out_sigma_eval", "perm_eval": "permutation_eval", "f_eval": "f_eval",
            "q_lookup_eval": "q_lookup_eval", "lookup_perm_eval": "z2_next_eval", "h_1_eval": "h1_eval", "h_1_next_eval": "h1_next_eval",
            "h_2_eval": "h2_eval"}
    aw, saw = pt.round5({lb: ev[f] for lb, f in feed.items()}, custom)
    ch["aggregate_witness_1"], ch["aggregate_witness_2"] = aw, saw
    order = ["a", "b", "c", "d", "z", "f", "h1", "h2", "z2", "t1", "t2", "t3", "t4"]
    proof = tr.proof_serialize([cm[k] for k in order], [cm["aw"], cm["saw"]], [ev[f] for f in tr.PROOF_EVAL_FIELDS], custom, cid)
    return {k: zk.curves.fr_from_mont(cid, np.asarray(v).reshape(1, 4))[0] for k, v in ch.items()}, proof


@pytest.mark.parametrize("cid", [0, 1])
def test_prover_transcript_schedule_and_proof_bytes(cid, wire):
    """Every challenge of one proof (prover.rs:179-594) and the serialised Proof (proof.rs:41-103) equal the fixture."""
    cv = bo.CURVES[cid]
    w = wire[cv.name]
    ch, proof = _replay(cv, w)
    exp = {k: int(v, 16) for k, v in w["transcript"]["challenges"].items()}
    assert ch == exp
    assert all(v < 1 << 248 for v in ch.values())              # 31 challenge bytes (transcript.rs:41)
    assert proof.hex() == w["proof_bytes"]
    g, f = wo.fq_flag_bytes(cv), 32
    assert len(proof) == 13 * g + 2 * (g + 1) + 16 * f + 8 + sum(8 + len(lb) + f for lb, _ in w["transcript"]["custom"])


def test_fixture_is_what_the_oracle_generates(wire, tmp_path):
    """tests/golden/wire.json is reproducible from the committed generator."""
    import subprocess
    import sys
    before = open(os.path.join(ROOT, "tests", "golden", "wire.json")).read()
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "gen_golden_wire.py")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert open(os.path.join(ROOT, "tests", "golden", "wire.json")).read() == before


def test_bls12_381_generator_compressed_known_layout():
    """x of the BLS12-381 G1 generator, little-endian, flags in the LAST byte: y = 0x08b3... is the smaller root, so no
    flag is set and the encoding is just x reversed (the zcash big-endian form 0x97f1d3a7... carries its flags in the FIRST byte)."""
    cv = bo.BLS12_381
    enc = tr.g1_serialize(point(cv, (cv.gx, cv.gy)), 0)
    assert enc == cv.gx.to_bytes(48, "little") and enc[-1] == 0x17 and enc[0] == 0xBB
    neg = tr.g1_serialize(point(cv, (cv.gx, cv.q - cv.gy)), 0)
    assert neg[:-1] == enc[:-1] and neg[-1] == 0x17 | 0x80
