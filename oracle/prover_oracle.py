"""TEST INFRASTRUCTURE -- `Prover::prove_with_preprocessed` (plonk-core/src/proof_system/prover.rs:163-638) restated on integers
from the pieces of this directory: bigint_oracle (transforms, grand products, quotient, linearisation, multisets, KZG witness),
wire_oracle (merlin transcript, ark-serialize, proof layout) and, for the commitments only, the C++ restatement's Pippenger
(oracle/ark_cpu.cpp through cpu.py; a pure-Python MSM takes minutes at 128 rows).  Used by tests/test_prover_gpu.py to compare the
device-resident prover's proof BYTES with an independent CPU computation of the same proof."""
import numpy as np

from . import bigint_oracle as bo
from . import wire_oracle as wo

KEYS = ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "q_arith", "q_range", "q_logic", "q_fixed", "q_var", "q_lookup")


def _strip(poly):
    """DensePolynomial::from_coefficients_vec: trailing zeros removed."""
    k = len(poly)
    while k and poly[k - 1] == 0:
        k -= 1
    return poly[:k]


def prove(cv: bo.Curve, log_n: int, sel: dict, sigma_evals, table_cols, wires, pub: dict, t: wo.PlonkTranscript, commit, ca: int, cd: int):
    """sel: selector evaluation vectors keyed by KEYS; sigma_evals / table_cols / wires: 4 lists of n; pub: {row: value};
    t: the transcript after the verifier key was seeded; commit(coefficients) -> affine point or None (KZG10 over the test's SRS).
    Returns (proof bytes, challenges, the polynomial behind every commitment and opening -- for a verifier that checks logarithms)."""
    p, n = cv.r, 1 << log_n
    ifft = lambda ev: bo.ntt(cv, 1, log_n, ev)                  # noqa: E731  domain.ifft
    coset4 = lambda poly: bo.ntt(cv, 2, log_n + 2, poly)        # noqa: E731  domain_4n.coset_fft
    ch = {"coeff_a": ca, "coeff_d": cd}

    def draw(label, put=None):
        c = t.challenge_scalar(label)
        t.append_fr(put or label, c)
        return c

    t.append_message(b"pi", wo.ser_public_inputs(cv, pub))                                                # :182
    w_polys = [ifft(w) for w in wires]                                                                    # :196-203
    w_comm = [commit(q) for q in w_polys]                                                                 # :213
    for lb, cm in zip((b"w_l", b"w_r", b"w_o", b"w_4"), w_comm):
        t.append_g1(lb, cm)                                                                               # :217-220
    ch["zeta"] = draw(b"zeta")                                                                            # :225-226
    t_ev, f_ev, h1_ev, h2_ev = bo.lookup_round2(cv, n, table_cols, sel["q_lookup"], wires, ch["zeta"])    # :229-297
    table_poly, f_poly, h1_poly, h2_poly = ifft(t_ev), ifft(f_ev), ifft(h1_ev), ifft(h2_ev)               # :240-305
    f_comm, h1_comm, h2_comm = commit(f_poly), commit(h1_poly), commit(h2_poly)                           # :289-317
    for lb, cm in ((b"f", f_comm), (b"h1", h1_comm), (b"h2", h2_comm)):
        t.append_g1(lb, cm)                                                                               # :294,320-321
    for name in ("beta", "gamma", "delta", "epsilon"):
        ch[name] = draw(name.encode())                                                                    # :326-337
    z_ev, _ = bo.perm_product(cv, log_n, wires, sigma_evals, ch["beta"], ch["gamma"])                     # :347-358
    z_poly = ifft(z_ev)
    z_comm = commit(z_poly)                                                                               # :361-363
    t.append_g1(b"z", z_comm)                                                                             # :366
    z2_ev, _ = bo.lookup_product(cv, f_ev, t_ev, h1_ev, h2_ev, ch["delta"], ch["epsilon"])                # :370-380
    z2_poly = ifft(z2_ev)
    z2_comm = commit(z2_poly)                                                                             # :387-389 (not appended)
    pi_poly = ifft([pub.get(i, 0) for i in range(n)])                                                     # :392
    ch["alpha"] = draw(b"alpha")                                                                          # :398-426
    ch["range"] = draw(b"range separation challenge", b"range seperation challenge")
    ch["logic"] = draw(b"logic separation challenge", b"logic seperation challenge")
    ch["fixed"] = draw(b"fixed base separation challenge")
    ch["var"] = draw(b"variable base separation challenge")
    ch["lookup"] = draw(b"lookup separation challenge")
    # quotient_poly::compute (quotient_poly.rs:34-178)
    key_polys = {k: ifft(sel[k]) for k in KEYS}
    sigma_polys = [ifft(s) for s in sigma_evals]
    col = {"w_l": w_polys[0], "w_r": w_polys[1], "w_o": w_polys[2], "w_4": w_polys[3], "z": z_poly, "z2": z2_poly, "f": f_poly,
           "table": table_poly, "h1": h1_poly, "h2": h2_poly, "pi": pi_poly, "l1": ifft([1] + [0] * (n - 1))}
    col.update(key_polys)
    col.update({f"sigma{k}": sigma_polys[k] for k in range(4)})
    col = {k: coset4(v) for k, v in col.items()}
    t_poly = bo.ntt(cv, 3, log_n + 2, bo.quotient_evals(cv, log_n, col, ch))                              # :428-453, coset_ifft
    t_parts = [t_poly[k * n:(k + 1) * n] for k in range(4)]                                               # :455-456
    t_comm = [commit(q) for q in t_parts]                                                                 # :459-469
    for k in range(4):
        t.append_g1(f"t_{k + 1}".encode(), t_comm[k])                                                     # :472-475
    ch["z"] = draw(b"z")                                                                                  # :480-481
    lin_key = dict(key_polys)
    lin_key.update({f"sigma{k}": sigma_polys[k] for k in range(4)})
    lin_poly, ev = bo.linearisation(cv, log_n, lin_key, {
        "w_l": w_polys[0], "w_r": w_polys[1], "w_o": w_polys[2], "w_4": w_polys[3], "t_1": t_parts[0], "t_2": t_parts[1], "t_3": t_parts[2],
        "t_4": t_parts[3], "z": z_poly, "z2": z2_poly, "f": f_poly, "h1": h1_poly, "h2": h2_poly, "table": table_poly}, ch)     # :483-512
    for lb, name in ((b"a_eval", "a_eval"), (b"b_eval", "b_eval"), (b"c_eval", "c_eval"), (b"d_eval", "d_eval"),
                     (b"left_sig_eval", "left_sigma_eval"), (b"right_sig_eval", "right_sigma_eval"), (b"out_sig_eval", "out_sigma_eval"),
                     (b"perm_eval", "permutation_eval"), (b"f_eval", "f_eval"), (b"q_lookup_eval", "q_lookup_eval"),
                     (b"lookup_perm_eval", "z2_next_eval"), (b"h_1_eval", "h1_eval"), (b"h_1_next_eval", "h1_next_eval"), (b"h_2_eval", "h2_eval")):
        t.append_fr(lb, ev[name])                                                                         # :516-544
    custom = [(k, ev[k]) for k in ("q_arith_eval", "q_c_eval", "q_l_eval", "q_r_eval", "a_next_eval", "b_next_eval", "d_next_eval")]
    for label, v in custom:
        t.append_fr(label.encode(), v)                                                                    # :546-554
    ch["aw"] = t.challenge_scalar(b"aggregate_witness")                                                   # :563
    aw_polys = [lin_poly, sigma_polys[0], sigma_polys[1], sigma_polys[2], f_poly, h2_poly, table_poly] + w_polys      # :569-591
    ch["saw"] = t.challenge_scalar(b"aggregate_witness")                                                  # :593-594
    saw_polys = [z_poly, w_polys[0], w_polys[1], w_polys[3], h1_poly, z2_poly, table_poly]                # :596-604

    def witness(polys, point, chi):                                                                       # PC::open (kzg.hip header)
        m = max(len(q) for q in polys)
        comb, pw = [0] * m, 1
        for q in polys:
            for i, v in enumerate(q):
                comb[i] = (comb[i] + pw * v) % p
            pw = pw * chi % p
        return bo.kzg_witness_poly(cv, comb, point)

    aw_w = witness(aw_polys, ch["z"], ch["aw"])                                                           # :582-591
    saw_w = witness(saw_polys, ch["z"] * cv.root_of_unity(log_n) % p, ch["saw"])                          # :609-618
    aw_open, saw_open = commit(aw_w), commit(saw_w)
    evals16 = [ev[k] for k in ("a_eval", "b_eval", "c_eval", "d_eval", "left_sigma_eval", "right_sigma_eval", "out_sigma_eval",
                               "permutation_eval", "q_lookup_eval", "z2_next_eval", "h1_eval", "h1_next_eval", "h2_eval", "f_eval",
                               "table_eval", "table_next_eval")]
    data = wo.proof_bytes(cv, w_comm + [z_comm, f_comm, h1_comm, h2_comm, z2_comm] + t_comm, [aw_open, saw_open], evals16, custom)
    polys = {"a_comm": w_polys[0], "b_comm": w_polys[1], "c_comm": w_polys[2], "d_comm": w_polys[3], "z_comm": z_poly, "f_comm": f_poly,
             "h_1_comm": h1_poly, "h_2_comm": h2_poly, "z_2_comm": z2_poly, "t_1_comm": t_parts[0], "t_2_comm": t_parts[1], "t_3_comm": t_parts[2],
             "t_4_comm": t_parts[3], "aw_opening": aw_w, "saw_opening": saw_w}
    polys.update(key_polys)
    polys.update({f"sigma{k}": sigma_polys[k] for k in range(4)})
    polys.update({f"table_{k + 1}": ifft(table_cols[k]) for k in range(4)})
    return data, ch, polys


def cpp_committer(cpu, cid: int, cv: bo.Curve, srs_mont_xy: np.ndarray):
    """commit(coefficients as integers) over an SRS given as Montgomery limb rows, through the C++ restatement's KZG commit."""
    from . import bigint_oracle as _bo
    L = 6 if cid == 0 else 4
    R = 1 << (64 * L)
    rinv = pow(R, -1, cv.q)

    def commit(coeffs):
        if not coeffs:
            return None
        mont = np.array([_bo.int_to_limbs(_bo.to_mont(c % cv.r, cv.r, cv.fr_R), 4) for c in coeffs], dtype=np.uint64).reshape(-1, 4)
        xy, inf = cpu.kzg_commit(cid, srs_mont_xy, mont)
        if inf:
            return None
        x = _bo.limbs_to_int(xy[:L]) * rinv % cv.q
        y = _bo.limbs_to_int(xy[L:]) * rinv % cv.q
        return (x, y)
    return commit
