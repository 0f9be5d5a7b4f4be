"""Generates tests/golden/wire.json -- fixtures for the canonical wire formats and the transcript
from the pure-Python restatement oracle/wire_oracle.py (itself pinned on merlin's conformance vectors, tests/test_wire.py).

    python tests/golden/gen_golden_wire.py

Content per curve: serialised Fr / G1 (compressed + uncompressed) for edge values (0, 1, r-1; infinity, the generator,
[2]G, -G, a point with the larger y), one full prover-transcript run over synthetic commitments / evaluations with every
challenge the reference draws (prover.rs:179-594), and the bytes of the resulting `Proof` (proof.rs:41-103).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import bigint_oracle as bo  # noqa: E402
from oracle import wire_oracle as wo  # noqa: E402


def main():
    out = {}
    for cv in (bo.BLS12_381, bo.BN254):
        G = (cv.gx, cv.gy)
        pts = {"infinity": None, "G": G, "2G": bo.ec_mul(cv, 2, G), "negG": bo.ec_neg(cv, G),
               "kG": bo.ec_mul(cv, 0xDEADBEEFCAFE, G), "big": bo.ec_mul(cv, cv.r - 5, G)}
        frs = {"zero": 0, "one": 1, "r_minus_1": cv.r - 1, "seeded": bo.seeded_scalars(cv, 99, 1)[0]}
        c = {"fr": {k: {"value": hex(v), "bytes": wo.ser_fr(cv, v).hex()} for k, v in frs.items()},
             "g1": {k: {"x": None if P is None else hex(P[0]), "y": None if P is None else hex(P[1]),
                        "compressed": wo.ser_g1(cv, P).hex(), "uncompressed": wo.ser_g1_uncompressed(cv, P).hex()} for k, P in pts.items()}}
        # one proof's transcript traffic over synthetic values
        sc = bo.seeded_scalars(cv, 4242, 64)
        commits = [bo.ec_mul(cv, s, G) for s in sc[:15]]          # 13 commitments + 2 opening witnesses
        commits[7] = None                                          # an infinite commitment (zero polynomial) mid-proof
        evals = sc[16:32]                                          # the 16 fixed evaluations of ProofEvaluations
        custom = [("q_arith_eval", sc[40]), ("a_next_eval", sc[41]), ("q_c_eval", sc[42])]
        pi = {0: sc[50], 7: sc[51], 1000: sc[52]}
        t = wo.PlonkTranscript(b"ark-plonk-amd test", cv)
        t.circuit_domain_sep(1 << 10)
        t.append_message(b"pi", wo.ser_public_inputs(cv, pi))
        names = ["a", "b", "c", "d", "z", "f", "h1", "h2", "z2", "t1", "t2", "t3", "t4", "aw", "saw"]
        cm = dict(zip(names, commits))
        ch = {}

        def draw(d, p):
            v = t.challenge_scalar(d)
            t.append_fr(p, v)
            ch[d.decode()] = hex(v)

        for lb, k in ((b"w_l", "a"), (b"w_r", "b"), (b"w_o", "c"), (b"w_4", "d")):
            t.append_g1(lb, cm[k])
        draw(b"zeta", b"zeta")
        for lb, k in ((b"f", "f"), (b"h1", "h1"), (b"h2", "h2")):
            t.append_g1(lb, cm[k])
        for lb in (b"beta", b"gamma", b"delta", b"epsilon"):
            draw(lb, lb)
        t.append_g1(b"z", cm["z"])
        draw(b"alpha", b"alpha")
        draw(b"range separation challenge", b"range seperation challenge")
        draw(b"logic separation challenge", b"logic seperation challenge")
        draw(b"fixed base separation challenge", b"fixed base separation challenge")
        draw(b"variable base separation challenge", b"variable base separation challenge")
        draw(b"lookup separation challenge", b"lookup separation challenge")
        for lb, k in ((b"t_1", "t1"), (b"t_2", "t2"), (b"t_3", "t3"), (b"t_4", "t4")):
            t.append_g1(lb, cm[k])
        draw(b"z", b"z")
        # evaluation labels (prover.rs:516-544) and which ProofEvaluations field each one carries
        ev = dict(zip(("a_eval", "b_eval", "c_eval", "d_eval", "left_sigma_eval", "right_sigma_eval", "out_sigma_eval", "permutation_eval",
                       "q_lookup_eval", "z2_next_eval", "h1_eval", "h1_next_eval", "h2_eval", "f_eval", "table_eval", "table_next_eval"), evals))
        feed = (("a_eval", "a_eval"), ("b_eval", "b_eval"), ("c_eval", "c_eval"), ("d_eval", "d_eval"),
                ("left_sig_eval", "left_sigma_eval"), ("right_sig_eval", "right_sigma_eval"), ("out_sig_eval", "out_sigma_eval"),
                ("perm_eval", "permutation_eval"), ("f_eval", "f_eval"), ("q_lookup_eval", "q_lookup_eval"),
                ("lookup_perm_eval", "z2_next_eval"), ("h_1_eval", "h1_eval"), ("h_1_next_eval", "h1_next_eval"), ("h_2_eval", "h2_eval"))
        for lb, field in feed:
            t.append_fr(lb.encode(), ev[field])
        for lb, v in custom:
            t.append_fr(lb.encode(), v)
        ch["aggregate_witness_1"] = hex(t.challenge_scalar(b"aggregate_witness"))
        ch["aggregate_witness_2"] = hex(t.challenge_scalar(b"aggregate_witness"))
        c["transcript"] = {
            "label": "ark-plonk-amd test", "n": 1 << 10,
            "pi": {str(k): hex(v) for k, v in pi.items()},
            "commitments": {k: (None if P is None else [hex(P[0]), hex(P[1])]) for k, P in cm.items()},
            "evals": {k: hex(v) for k, v in ev.items()},
            "custom": [[lb, hex(v)] for lb, v in custom],
            "challenges": ch,
        }
        c["proof_bytes"] = wo.proof_bytes(cv, commits[:13], commits[13:], evals, custom).hex()
        out[cv.name] = c
    path = os.path.join(ROOT, "tests", "golden", "wire.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
