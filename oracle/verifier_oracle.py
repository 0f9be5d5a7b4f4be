"""TEST INFRASTRUCTURE -- a restatement of the reference VERIFIER on integers, used only by tests/ to check that the proofs the
device-resident prover (ark_plonk_amd/prover.py) emits satisfy the reference's verification equations.

Follows `Proof::verify` (plonk-core/src/proof_system/proof.rs:110-425): parse the proof bytes (proof.rs:41-103 layout), replay the
transcript (the independent merlin of oracle/wire_oracle.py) to get every challenge, compute r_0 (`compute_r0`, :427-486) and
the 19 scalars of the linearisation commitment (`compute_linearisation_commitment`, :488-611, with widget/arithmetic.rs:128-157,
widget/mod.rs:109-130, permutation.rs:327-388, widget/lookup.rs:223-300), and check the two batched KZG openings (:343-424).

The one thing it does NOT restate is the pairing: `PC::check` (ark-poly-commit kzg10, e(C - v G, H) = e(W, tau H - z H)) is
replaced by the same equation on discrete logarithms, which a test can do because it generated the SRS from a known tau:
w(tau) (tau - z) = sum_k chi^k (c_k(tau) - v_k).  The caller supplies the logarithm of every commitment (and checks separately,
by scalar multiplication, that the commitment IS that multiple of G)."""
from . import bigint_oracle as bo
from . import wire_oracle as wo

COMMITMENTS = ("a_comm", "b_comm", "c_comm", "d_comm", "z_comm", "f_comm", "h_1_comm", "h_2_comm", "z_2_comm", "t_1_comm", "t_2_comm",
               "t_3_comm", "t_4_comm")
EVALS = ("a_eval", "b_eval", "c_eval", "d_eval", "left_sigma_eval", "right_sigma_eval", "out_sigma_eval", "permutation_eval",
         "q_lookup_eval", "z2_next_eval", "h1_eval", "h1_next_eval", "h2_eval", "f_eval", "table_eval", "table_next_eval")


def parse_proof(cv: bo.Curve, data: bytes) -> dict:
    """proof.rs:41-103: 13 compressed commitments, 2 x (compressed point + Option byte), 16 Fr, Vec<(String, Fr)>."""
    g, f = wo.fq_flag_bytes(cv), wo.fr_bytes(cv)
    pos = 0
    out = {"commitments": {}, "evals": {}}
    for name in COMMITMENTS:
        out["commitments"][name] = wo.de_g1(cv, data[pos:pos + g])
        pos += g
    for name in ("aw_opening", "saw_opening"):
        out[name] = wo.de_g1(cv, data[pos:pos + g])
        pos += g
        assert data[pos] == 0                       # random_v: None
        pos += 1
    for name in EVALS:
        out["evals"][name] = int.from_bytes(data[pos:pos + f], "little")
        assert out["evals"][name] < cv.r
        pos += f
    k = int.from_bytes(data[pos:pos + 8], "little")
    pos += 8
    custom = []
    for _ in range(k):
        ln = int.from_bytes(data[pos:pos + 8], "little")
        pos += 8
        label = data[pos:pos + ln].decode()
        pos += ln
        custom.append((label, int.from_bytes(data[pos:pos + f], "little")))
        pos += f
    assert pos == len(data)
    out["custom"] = custom
    out["evals"].update(dict(custom))
    return out


VK_SEED = (("q_m", "q_m"), ("q_l", "q_l"), ("q_r", "q_r"), ("q_o", "q_o"), ("q_c", "q_c"), ("q_4", "q_4"), ("q_arith", "q_arith"),
           ("q_range", "q_range"), ("q_logic", "q_logic"), ("q_variable_group_add", "q_var"), ("q_fixed_group_add", "q_fixed"),
           ("left_sigma", "sigma0"), ("right_sigma", "sigma1"), ("out_sigma", "sigma2"), ("fourth_sigma", "sigma3"))


def seed_transcript(cv: bo.Curve, t: wo.PlonkTranscript, vk_points: dict, n: int):
    """`VerifierKey::seed_transcript` (widget/mod.rs:252-278).  vk_points: this module's key names -> affine point or None."""
    for label, name in VK_SEED:
        t.append_g1(label.encode(), vk_points[name])
    t.circuit_domain_sep(n)
    return t


def replay_transcript(cv: bo.Curve, t: wo.PlonkTranscript, proof: dict, pub_inputs: dict) -> dict:
    """proof.rs:128-300, 343-378: the verifier's transcript traffic.  t: the transcript after the verifier key was seeded."""
    cm, ev = proof["commitments"], proof["evals"]
    ch = {}

    def draw(label, put=None):
        c = t.challenge_scalar(label)
        t.append_fr(put or label, c)
        return c

    t.append_message(b"pi", wo.ser_public_inputs(cv, pub_inputs))
    for lb, name in ((b"w_l", "a_comm"), (b"w_r", "b_comm"), (b"w_o", "c_comm"), (b"w_4", "d_comm")):
        t.append_g1(lb, cm[name])
    ch["zeta"] = draw(b"zeta")
    for lb, name in ((b"f", "f_comm"), (b"h1", "h_1_comm"), (b"h2", "h_2_comm")):
        t.append_g1(lb, cm[name])
    for name in ("beta", "gamma", "delta", "epsilon"):
        ch[name] = draw(name.encode())
    t.append_g1(b"z", cm["z_comm"])
    ch["alpha"] = draw(b"alpha")
    ch["range"] = draw(b"range separation challenge", b"range seperation challenge")
    ch["logic"] = draw(b"logic separation challenge", b"logic seperation challenge")
    ch["fixed"] = draw(b"fixed base separation challenge")
    ch["var"] = draw(b"variable base separation challenge")
    ch["lookup"] = draw(b"lookup separation challenge")
    for k in range(4):
        t.append_g1(f"t_{k + 1}".encode(), cm[f"t_{k + 1}_comm"])
    ch["z"] = draw(b"z")
    for lb, name in ((b"a_eval", "a_eval"), (b"b_eval", "b_eval"), (b"c_eval", "c_eval"), (b"d_eval", "d_eval"),
                     (b"left_sig_eval", "left_sigma_eval"), (b"right_sig_eval", "right_sigma_eval"), (b"out_sig_eval", "out_sigma_eval"),
                     (b"perm_eval", "permutation_eval"), (b"f_eval", "f_eval"), (b"q_lookup_eval", "q_lookup_eval"),
                     (b"lookup_perm_eval", "z2_next_eval"), (b"h_1_eval", "h1_eval"), (b"h_1_next_eval", "h1_next_eval"), (b"h_2_eval", "h2_eval")):
        t.append_fr(lb, ev[name])
    for label, v in proof["custom"]:
        t.append_fr(label.encode(), v)
    ch["aw"] = t.challenge_scalar(b"aggregate_witness")
    ch["saw"] = t.challenge_scalar(b"aggregate_witness")
    return ch


def compute_r0(cv: bo.Curve, log_n: int, ev: dict, ch: dict, pub_inputs: dict) -> int:
    """proof.rs:427-486."""
    p, n = cv.r, 1 << log_n
    z = ch["z"]
    zh = (pow(z, n, p) - 1) % p
    l1 = zh * pow(n * (z - 1) % p, -1, p) % p
    w = cv.root_of_unity(log_n)
    pi_eval = sum(v * pow(w, i, p) % p * zh % p * pow(n * (z - pow(w, i, p)) % p, -1, p) for i, v in pub_inputs.items()) % p   # :635-666
    al, be, ga, de, ep, ls = ch["alpha"], ch["beta"], ch["gamma"], ch["delta"], ch["epsilon"], ch["lookup"]
    b = (ev["a_eval"] + be * ev["left_sigma_eval"] + ga) * (ev["b_eval"] + be * ev["right_sigma_eval"] + ga) % p \
        * (ev["c_eval"] + be * ev["out_sigma_eval"] + ga) % p * ((ev["d_eval"] + ga) * ev["permutation_eval"] % p * al % p) % p
    c = l1 * al * al % p
    e1d = ep * (1 + de) % p
    d = ls * ls % p * ev["z2_next_eval"] % p * (e1d + de * ev["h2_eval"]) % p * (e1d + ev["h2_eval"] + de * ev["h1_next_eval"]) % p
    e = ls * ls * ls % p * l1 % p
    return (pi_eval - b - c - d - e) % p


def linearisation_terms(cv: bo.Curve, log_n: int, ev: dict, ch: dict) -> list:
    """proof.rs:488-611: the 19 (commitment name, scalar) pairs of [r]_1, in the order the verifier pushes them."""
    p, n = cv.r, 1 << log_n
    z = ch["z"]
    a, b, c, d = ev["a_eval"], ev["b_eval"], ev["c_eval"], ev["d_eval"]
    a_n, b_n, d_n = ev["a_next_eval"], ev["b_next_eval"], ev["d_next_eval"]
    q_arith, q_c, q_l, q_r = ev["q_arith_eval"], ev["q_c_eval"], ev["q_l_eval"], ev["q_r_eval"]
    ca, cd = ch["coeff_a"], ch["coeff_d"]
    zh = (pow(z, n, p) - 1) % p
    z_n = (zh + 1) % p
    l1 = zh * pow(n * (z - 1) % p, -1, p) % p
    al, be, ga, de, ep, ze, ls = ch["alpha"], ch["beta"], ch["gamma"], ch["delta"], ch["epsilon"], ch["zeta"], ch["lookup"]
    terms = [("q_m", a * b * q_arith), ("q_l", a * q_arith), ("q_r", b * q_arith), ("q_o", c * q_arith), ("q_4", d * q_arith), ("q_c", q_arith),   # arithmetic.rs:128-157
             ("q_range", bo.range_constraint(p, ch["range"], a, b, c, d, d_n)),                                                                    # widget/mod.rs:109-130
             ("q_logic", bo.logic_constraint(p, ch["logic"], a, b, c, d, a_n, b_n, d_n, q_c)),
             ("q_fixed", bo.fixed_base_constraint(p, ch["fixed"], a, b, c, d, a_n, b_n, d_n, q_l, q_r, q_c, ca, cd)),
             ("q_var", bo.curve_add_constraint(p, ch["var"], a, b, c, d, a_n, b_n, d_n, ca, cd))]
    # widget/lookup.rs:223-300
    opd, e1d = (1 + de) % p, ep * (1 + de) % p
    terms += [("q_lookup", ((a + ze * (b + ze * (c + ze * d))) - ev["f_eval"]) * ls),
              ("z_2_comm", opd * (ep + ev["f_eval"]) % p * (e1d + ev["table_eval"] + de * ev["table_next_eval"]) % p * ls * ls + l1 * ls * ls * ls),
              ("h_1_comm", -ev["z2_next_eval"] * ls * ls % p * (e1d + ev["h2_eval"] + de * ev["h1_next_eval"]))]
    # permutation.rs:327-388
    bz = be * z % p
    x = (a + bz + ga) * (b + bo.PERM_K[1] * bz + ga) % p * (c + bo.PERM_K[2] * bz + ga) % p * ((d + bo.PERM_K[3] * bz + ga) * al % p)
    y = -((a + be * ev["left_sigma_eval"] + ga) * (b + be * ev["right_sigma_eval"] + ga) % p * (c + be * ev["out_sigma_eval"] + ga) % p
          * (be * ev["permutation_eval"] % p * al % p))
    terms += [("z_comm", x + l1 * al * al), ("sigma3", y)]
    # proof.rs:590-606: -Z_H(z) * z^(kn) for the four quotient pieces
    terms += [(f"t_{k + 1}_comm", -zh * pow(z_n, k, p)) for k in range(4)]
    return [(name, s % p) for name, s in terms]


def verify_with_trapdoor(cv: bo.Curve, log_n: int, proof_bytes: bytes, seeded_transcript: wo.PlonkTranscript, pub_inputs: dict,
                         dlog: dict, tau: int, coeff_a: int, coeff_d: int):
    """dlog: name -> polynomial(tau) for the 13 proof commitments, aw_opening / saw_opening, and the verifier key's commitments
    (q_m ... q_lookup as in bigint_oracle.LIN_KEY, sigma0..3, table_1..4).  Returns (ok, challenges, details)."""
    p = cv.r
    proof = parse_proof(cv, proof_bytes)
    ev = proof["evals"]
    ch = replay_transcript(cv, seeded_transcript, proof, pub_inputs)
    ch.update(coeff_a=coeff_a, coeff_d=coeff_d)
    r0 = compute_r0(cv, log_n, ev, ch, pub_inputs)
    lin = sum(s * dlog[name] for name, s in linearisation_terms(cv, log_n, ev, ch)) % p
    ze = ch["zeta"]
    table = (dlog["table_1"] + ze * dlog["table_2"] + ze * ze % p * dlog["table_3"] + ze * ze * ze % p * dlog["table_4"]) % p   # :303-312
    aw = [(lin, -r0 % p), (dlog["sigma0"], ev["left_sigma_eval"]), (dlog["sigma1"], ev["right_sigma_eval"]), (dlog["sigma2"], ev["out_sigma_eval"]),
          (dlog["f_comm"], ev["f_eval"]), (dlog["h_2_comm"], ev["h2_eval"]), (table, ev["table_eval"]), (dlog["a_comm"], ev["a_eval"]),
          (dlog["b_comm"], ev["b_eval"]), (dlog["c_comm"], ev["c_eval"]), (dlog["d_comm"], ev["d_eval"])]                          # :345-372
    saw = [(dlog["z_comm"], ev["permutation_eval"]), (dlog["a_comm"], ev["a_next_eval"]), (dlog["b_comm"], ev["b_next_eval"]),
           (dlog["d_comm"], ev["d_next_eval"]), (dlog["h_1_comm"], ev["h1_next_eval"]), (dlog["z_2_comm"], ev["z2_next_eval"]),
           (table, ev["table_next_eval"])]                                                                                         # :377-395
    z = ch["z"]
    zw = z * cv.root_of_unity(log_n) % p

    def check(pairs, chi, point, w_log):
        acc, pw = 0, 1
        for c_log, v in pairs:
            acc = (acc + pw * (c_log - v)) % p
            pw = pw * chi % p
        return w_log * (tau - point) % p == acc

    ok_aw = check(aw, ch["aw"], z, dlog["aw_opening"])
    ok_saw = check(saw, ch["saw"], zw, dlog["saw_opening"])
    return ok_aw and ok_saw, ch, {"aw": ok_aw, "saw": ok_saw, "r0": r0, "proof": proof}
