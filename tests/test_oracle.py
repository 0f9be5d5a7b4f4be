"""CPU suite: pins the oracle (C++ restatement of the ark 0.3 algorithms) against the definitional
big-int oracle, the committed golden vectors and the external constants."""
import os

import numpy as np
import pytest

from oracle import bigint_oracle as bo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from published_points import (EXT_BLS_2G, EXT_BN254_2G, EXT_BN254_3G, EXT_BN254_9G, EXT_BN254_MUL, EXT_BLS_FR_ROOTS,  # noqa: E402
                              EXT_BN254_FR_ROOT_28)


def test_external_constants():
    cv = bo.BLS12_381
    # ark-bls12-381 0.3 Fr::TWO_ADIC_ROOT_OF_UNITY (decimal)
    assert cv.root_of_unity(32) == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    # [2]G1 of BLS12-381: x-coordinate of the published compressed encoding a572cbea...f0f4e
    g2 = bo.ec_add(cv, (cv.gx, cv.gy), (cv.gx, cv.gy))
    assert g2[0] == 0x0572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e
    assert g2[1] == 0x166a9d8cabc673a322fda673779d8e3822ba3ecb8670e461f73bb9021d5fd76a4c56d9d4cd16bd1bba86881979749d28      # EIP-2537's G1 + G1 vector
    assert bo.on_curve(cv, g2)
    # alt_bn128 (BN254): the points every EIP-196 ecAdd / ecMul test uses, [2](1, 2) and [3](1, 2); circom's 2^28-th root of unity
    bn = bo.BN254
    h2 = bo.ec_add(bn, (bn.gx, bn.gy), (bn.gx, bn.gy))
    assert h2 == EXT_BN254_2G and bo.ec_add(bn, h2, (bn.gx, bn.gy)) == EXT_BN254_3G
    assert bn.root_of_unity(28) == EXT_BN254_FR_ROOT_28
    for k, w in EXT_BLS_FR_ROOTS.items():
        assert cv.root_of_unity(k) == w
    assert bo.ec_mul(bn, 9, (bn.gx, bn.gy)) == EXT_BN254_9G
    assert bo.on_curve(bn, EXT_BN254_MUL["point"]) and bo.ec_mul(bn, EXT_BN254_MUL["scalar"], EXT_BN254_MUL["point"]) == EXT_BN254_MUL["result"]
    # Montgomery R of Fr
    assert (1 << 256) % cv.r == 0x1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe
    for c in (bo.BLS12_381, bo.BN254):
        assert bo.on_curve(c, (c.gx, c.gy))
        w = c.root_of_unity(c.two_adicity)
        assert pow(w, 1 << c.two_adicity, c.r) == 1 and pow(w, 1 << (c.two_adicity - 1), c.r) != 1
        assert bo.ec_mul(c, c.r, (c.gx, c.gy)) is None  # generator has order r


def test_ark_window_rule_and_add_counts():
    # SURVEY.md appendix table
    assert bo.ark_window_size(1 << 10) == 8 and bo.ark_window_size(1 << 20) == 15 and bo.ark_window_size(1 << 22) == 17
    assert bo.ark_window_size(31) == 3 and bo.ark_window_size(1 << 18) == 14
    assert bo.ark_msm_adds(1 << 20) == 18939870
    assert bo.ark_msm_adds(1 << 22) == 66846690
    assert bo.ark_msm_adds(1 << 18, 254) == 5603290
    assert bo.ark_msm_adds(1 << 10) == 49088


@pytest.mark.parametrize("cid", [0, 1])
def test_dft_recursive_matches_definition(cid):
    cv = bo.CURVES[cid]
    for log_n in (0, 1, 2, 3, 5):
        vals = bo.seeded_scalars(cv, 11 + log_n, 1 << log_n)
        w = cv.root_of_unity(log_n)
        assert bo.dft_naive(vals, w, cv.r) == bo.ntt(cv, bo.KIND_FFT, log_n, vals)
        ev = bo.ntt(cv, bo.KIND_COSET_FFT, log_n, vals)
        for i in (0, (1 << log_n) - 1):
            assert ev[i] == bo.horner(vals, cv.fr_generator * pow(w, i, cv.r) % cv.r, cv.r)
        assert bo.ntt(cv, bo.KIND_IFFT, log_n, bo.ntt(cv, bo.KIND_FFT, log_n, vals)) == vals
        assert bo.ntt(cv, bo.KIND_COSET_IFFT, log_n, ev) == vals


@pytest.mark.parametrize("cid", [0, 1])
def test_cpu_oracle_ntt_matches_golden(cid, golden, oracle_cpu):
    g = golden[cid]
    keys = [k[:-3] for k in g.files if k.startswith("ntt_") and k.endswith("_in")]
    assert len(keys) == 6 * 5 * 4
    for key in keys:
        _, log_n, _, kind = key.split("_")
        got = oracle_cpu.ntt(cid, int(kind), int(log_n), g[key + "_in"])
        assert np.array_equal(got, g[key + "_out"]), key


@pytest.mark.parametrize("cid", [0, 1])
def test_cpu_oracle_msm_matches_golden(cid, golden, oracle_cpu):
    g = golden[cid]
    srs = g["srs_1024"]
    tau = int(g["srs_tau"][0][0])
    assert np.array_equal(oracle_cpu.srs_powers(cid, tau, 64), srs[:64])
    for n in (1, 2, 31, 32, 33, 1024):
        out, inf = oracle_cpu.msm_g1(cid, srs[:n], g[f"msm_srs_{n}_scalars"])
        assert np.array_equal(out, g[f"msm_srs_{n}_out"]) and inf == int(g[f"msm_srs_{n}_inf"][0]), n
    for name in ("repeat", "cancel", "onebucket", "infbase", "zeros", "ones", "mixed", "maxscalar"):
        out, inf = oracle_cpu.msm_g1(cid, g[f"msm_case_{name}_bases"], g[f"msm_case_{name}_scalars"], inf=g[f"msm_case_{name}_inf"])
        assert inf == int(g[f"msm_case_{name}_outinf"][0]), name
        assert np.array_equal(out, g[f"msm_case_{name}_out"]), name


@pytest.mark.parametrize("cid", [0, 1])
def test_cpu_oracle_kzg_matches_golden(cid, golden, oracle_cpu):
    g = golden[cid]
    srs = g["srs_1024"]
    for k in range(4):
        out, inf = oracle_cpu.kzg_commit(cid, srs, g[f"kzg_poly_{k}"])
        assert inf == 0 and np.array_equal(out, g[f"kzg_commit_{k}"])
    # open: RLC with Fr ops of the oracle, witness by synthetic division, commit
    polys = [g[f"kzg_poly_{k}"] for k in range(4)]
    m = max(p.shape[0] for p in polys)
    comb = np.zeros((m, 4), dtype=np.uint64)
    chi_pow = oracle_cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    for p in polys:
        term = oracle_cpu.fr_op(cid, "mul", p, np.repeat(chi_pow, p.shape[0], axis=0))
        comb[: p.shape[0]] = oracle_cpu.fr_op(cid, "add", comb[: p.shape[0]], term)
        chi_pow = oracle_cpu.fr_op(cid, "mul", chi_pow, g["kzg_chi"].reshape(1, 4))
    w = oracle_cpu.kzg_witness(cid, comb, g["kzg_z"])
    out, inf = oracle_cpu.kzg_commit(cid, srs, w)
    assert inf == int(g["kzg_open_inf"][0]) and np.array_equal(out, g["kzg_open"])


@pytest.mark.parametrize("cid", [0, 1])
def test_cpu_oracle_medium_properties(cid, oracle_cpu):
    """2^12 round trips + Horner spot checks of the C++ restatement (too big for the big-int DFT)."""
    cv = bo.CURVES[cid]
    log_n = 12
    n = 1 << log_n
    vals = bo.seeded_scalars(cv, 0xABC + cid, n // 4)
    vm = oracle_cpu.convert(cid, "fr", True, oracle_cpu.ints_to_limbs(vals, 4))
    ev = oracle_cpu.ntt(cid, bo.KIND_COSET_FFT, log_n, vm)
    back = oracle_cpu.ntt(cid, bo.KIND_COSET_IFFT, log_n, ev)
    assert np.array_equal(back[: n // 4], vm) and not back[n // 4:].any()
    evi = oracle_cpu.limbs_to_ints(oracle_cpu.convert(cid, "fr", False, ev))
    w = cv.root_of_unity(log_n)
    for i in (0, 1, 777, n - 1):
        assert evi[i] == bo.horner(vals, cv.fr_generator * pow(w, i, cv.r) % cv.r, cv.r)
    ev2 = oracle_cpu.ntt(cid, bo.KIND_FFT, log_n, vm)
    assert np.array_equal(oracle_cpu.ntt(cid, bo.KIND_IFFT, log_n, ev2)[: n // 4], vm)


@pytest.mark.parametrize("cid", [0, 1])
def test_grand_product_oracle_matches_fixtures(cid):
    """oracle/bigint_oracle.py perm_product / lookup_product (permutation/mod.rs:652-822) vs tests/golden/grand_product.npz,
    and the closing property of a real wire permutation (the reference's own check, mod.rs:1243-1380)."""
    import ark_plonk_amd.curves as cvs
    gp = np.load(os.path.join(ROOT, "tests", "golden", "grand_product.npz"))
    cv = bo.CURVES[cid]
    ints = lambda a: cvs.fr_from_mont(cid, a)  # noqa: E731
    for log_n in (3, 6):
        pre = f"{cv.name}_perm{log_n}"
        beta, gamma = ints(gp[f"{pre}_beta_gamma"])
        z, last = bo.perm_product(cv, log_n, [ints(gp[f"{pre}_w{k}"]) for k in range(4)], [ints(gp[f"{pre}_s{k}"]) for k in range(4)], beta, gamma)
        assert z == ints(gp[f"{pre}_z"]) and [last] == ints(gp[f"{pre}_last"])
        pre = f"{cv.name}_look{log_n}"
        delta, eps = ints(gp[f"{pre}_delta_eps"])
        p, lastp = bo.lookup_product(cv, *[ints(gp[f"{pre}_{nm}"]) for nm in ("f", "t", "h1", "h2")], delta, eps)
        assert p == ints(gp[f"{pre}_p"]) and [lastp] == ints(gp[f"{pre}_last"])
    pre = f"{cv.name}_permv"
    beta, gamma = ints(gp[f"{pre}_beta_gamma"])
    z, last = bo.perm_product(cv, 4, [ints(gp[f"{pre}_w{k}"]) for k in range(4)], [ints(gp[f"{pre}_s{k}"]) for k in range(4)], beta, gamma)
    assert z[0] == 1 and last == 1 and z == ints(gp[f"{pre}_z"]) and len(set(z)) > 8



def _reference_fixture_only_left_wires(cv):
    """permutation/mod.rs:971-1092 `test_permutation_compute_sigmas_only_left_wires`: four gates, var_zero on L0 R0 L1 L2 L3, var_nine on F0..F3;
    the sigma encodings the test spells out (:1040-1076; K1, K2, K3 = 7, 13, 17: permutation/constants.rs:12-22) and its wire values (:1078-1082)."""
    p = cv.r
    w = cv.root_of_unity(2)
    w2, w3 = w * w % p, w * w % p * w % p
    sig = [[7, w2, w3, 1],                                # left  = {R0, L2, L3, L0}
           [w, w * 7 % p, w2 * 7 % p, w3 * 7 % p],        # right = {L1, R1, R2, R3}
           [13, w * 13 % p, w2 * 13 % p, w3 * 13 % p],    # out   = {O0, O1, O2, O3}
           [w * 17 % p, w2 * 17 % p, w3 * 17 % p, 17]]    # fourth = {F1, F2, F3, F0}
    wires = [[2, 2, 2, 2], [2, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]]
    return wires, sig


def _reference_fixture_two_gates(cv):
    """permutation/mod.rs:1201-1233 `test_basic_slow_permutation_poly`: two gates (v1 v2 v3 v4 | v3 v2 v1 v4), wire values as the test gives
    them; the sigma encodings follow from the map by the rule the four-gate test spells out (a wire maps to the next wire of its variable:
    v1 = {L0, O1}, v2 = {R0, R1}, v3 = {O0, L1}, v4 = {F0, F1}; w = -1 on the two-point domain)."""
    p = cv.r
    w = cv.root_of_unity(1)
    assert w == p - 1
    sig = [[w * 13 % p, 13],     # left   = {O1, O0}
           [w * 7 % p, 7],       # right  = {R1, R0}
           [w, 1],               # out    = {L1, L0}
           [w * 17 % p, 17]]     # fourth = {F1, F0}
    wires = [[1, 3], [2, 2], [3, 1], [1, 1]]
    return wires, sig


@pytest.mark.parametrize("cid", [0, 1])
def test_grand_product_oracle_on_the_reference_fixture(cid):
    """The reference's two deterministic permutation fixtures (above) under the checks of its `test_correct_permutation_poly` (mod.rs:1243-1380),
    which draws beta and gamma at random -- here five seeded pairs: z[0] = 1, the products of the numerators and denominators are equal
    (the value after the last row is 1), z(X) = ifft(z) has z(1) = 1 and degree n - 1, and z(X w) * den(X) = z(X) * num(X) on every root."""
    cv = bo.CURVES[cid]
    p = cv.r
    for log_n, (wires, sig) in ((2, _reference_fixture_only_left_wires(cv)), (1, _reference_fixture_two_gates(cv))):
        n = 1 << log_n
        w = cv.root_of_unity(log_n)
        for seed in range(5):
            beta, gamma = bo.seeded_scalars(cv, 0x9A0 + seed, 2)
            z, last = bo.perm_product(cv, log_n, wires, sig, beta, gamma)
            assert len(z) == n and z[0] == 1 and last == 1
            zp = bo.ntt(cv, bo.KIND_IFFT, log_n, z)
            assert bo.horner(zp, 1, p) == 1 and bo.horner(zp, pow(w, n, p), p) == 1 and zp[n - 1] != 0
            for i in range(n):
                root = pow(w, i, p)
                num = den = 1
                for k in range(4):
                    num = num * (wires[k][i] + beta * bo.PERM_K[k] * root + gamma) % p
                    den = den * (wires[k][i] + beta * sig[k][i] + gamma) % p
                assert bo.horner(zp, root * w % p, p) * den % p == bo.horner(zp, root, p) * num % p


@pytest.mark.parametrize("cid", [0, 1])
def test_quotient_oracle_matches_fixtures(cid):
    """oracle/bigint_oracle.py quotient_evals (quotient_poly.rs:34-178 + widgets) vs tests/golden/quotient.npz."""
    import ark_plonk_amd.curves as cvs
    gq = np.load(os.path.join(ROOT, "tests", "golden", "quotient.npz"))
    cv = bo.CURVES[cid]
    ints = lambda a: cvs.fr_from_mont(cid, a)  # noqa: E731
    col = {name: ints(gq[f"{cv.name}_col_{name}"]) for name in bo.QUOTIENT_COLS}
    ch = dict(zip(bo.QUOTIENT_CHALLENGES, ints(gq[f"{cv.name}_challenges"])))
    assert bo.quotient_evals(cv, 2, col, ch) == ints(gq[f"{cv.name}_quotient"])
    # each widget is switched by its selector alone (widget/mod.rs:83-91): zeroing one selector changes the point
    base = bo.quotient_at(cv, 2, 5, col, ch)
    for sel in ("q_arith", "q_range", "q_logic", "q_fixed", "q_var", "q_lookup"):
        c2 = dict(col)
        c2[sel] = [0] * len(col[sel])
        assert bo.quotient_at(cv, 2, 5, c2, ch) != base, sel


@pytest.mark.parametrize("cid", [0, 1])
def test_linearisation_oracle_matches_fixtures_and_the_quotient_identity(cid):
    """oracle/bigint_oracle.py linearisation (linearisation_poly.rs:164-350) vs tests/golden/linearisation.npz, and pinned on the
    quotient oracle: at a point x of the 4n coset taken as z_challenge, the quotient numerator N(x) = quotient_at(x) * Z_H(x) is
    the linearisation polynomial at x plus Z_H(x) t(x) plus the terms the verifier adds back (proof.rs:556-611
    compute_quotient_evaluation: pi(z), the constant part of the copy product, L_1(z) alpha^2, the lookup constants) -- for ANY
    polynomials, so every scalar and sign of the linearisation is checked against the widgets' quotient terms."""
    import ark_plonk_amd.curves as cvs
    from tests.golden.gen_golden_linearisation import CHALLENGES, EVAL_NAMES, LOG_N, case
    g = np.load(os.path.join(ROOT, "tests", "golden", "linearisation.npz"))
    cv = bo.CURVES[cid]
    p = cv.r
    ints = lambda a: cvs.fr_from_mont(cid, a)  # noqa: E731
    key = {name: ints(g[f"{cv.name}_key_{name}"]) for name in bo.LIN_KEY}
    polys = {name: ints(g[f"{cv.name}_poly_{name}"]) for name in bo.LIN_POLYS}
    ch = dict(zip(CHALLENGES, ints(g[f"{cv.name}_challenges"])))
    k2, p2, c2 = case(cv, LOG_N, 0x9100 + 0x100 * cid)
    assert (k2, p2, c2) == (key, polys, ch)                      # the generator is deterministic
    lin, ev = bo.linearisation(cv, LOG_N, key, polys, ch)
    assert lin == ints(g[f"{cv.name}_lin"]) and [ev[k] for k in EVAL_NAMES] == ints(g[f"{cv.name}_evals"])
    # -- the identity, at coset point i
    n, n4 = 1 << LOG_N, 4 << LOG_N
    col = {name: bo.ntt(cv, 2, LOG_N + 2, src) for name, src in list(key.items()) + [(k, polys[k]) for k in
                                                                                       ("w_l", "w_r", "w_o", "w_4", "z", "z2", "f", "table", "h1", "h2")]}
    pi = bo.seeded_scalars(cv, 0x77, n)
    l1_evals = [1] + [0] * (n - 1)
    col["pi"] = bo.ntt(cv, 2, LOG_N + 2, pi)
    col["l1"] = bo.ntt(cv, 2, LOG_N + 2, bo.ntt(cv, 1, LOG_N, l1_evals))
    for i in (1, 6, 19):
        x = cv.fr_generator * pow(cv.root_of_unity(LOG_N + 2), i, p) % p
        chx = dict(ch, z=x)
        lin_x, e = bo.linearisation(cv, LOG_N, key, polys, chx)
        zh = (pow(x, n, p) - 1) % p
        numerator = bo.quotient_at(cv, LOG_N, i, col, chx) * zh % p
        xn = pow(x, n, p)
        t_x = sum(bo.horner(polys[f"t_{k + 1}"], x, p) * pow(xn, k, p) for k in range(4)) % p
        l1 = col["l1"][i]
        al, be, ga, de, ep, ls = (chx[k] for k in ("alpha", "beta", "gamma", "delta", "epsilon", "lookup"))
        copy_rest = (e["a_eval"] + be * e["left_sigma_eval"] + ga) * (e["b_eval"] + be * e["right_sigma_eval"] + ga) % p \
            * (e["c_eval"] + be * e["out_sigma_eval"] + ga) % p * (e["d_eval"] + ga) % p * e["permutation_eval"] % p * al % p
        e1d = ep * (1 + de) % p
        h1_x = e["h1_eval"]
        look_rest = e["z2_next_eval"] * ls * ls % p * (e1d + de * e["h2_eval"]) % p * (e1d + e["h2_eval"] + de * e["h1_next_eval"]) % p
        back = (bo.horner(lin_x, x, p) + zh * t_x + bo.horner(pi, x, p) - copy_rest - l1 * al * al - look_rest - l1 * ls * ls * ls) % p
        assert back == numerator, i
        assert h1_x == col["h1"][i] and e["a_next_eval"] == col["w_l"][(i + 4) % n4]


def test_combine_split_oracle_on_the_reference_vector():
    """The reference's own known-answer test for `MultiSet::combine_split` (lookup/multiset.rs:335-392 `test_combine_split`,
    run there for BLS12-381 and BLS12-377 Fr): t = 0..6, f = [3,6,0,5,4,3,2,0,0,1,2]; and the Plonkup paper's example quoted in
    the function's doc comment (multiset.rs:125-130)."""
    t = [0, 1, 2, 3, 4, 5, 6]
    f = [3, 6, 0, 5, 4, 3, 2, 0, 0, 1, 2]
    evens, odds = bo.combine_split(t, f)
    assert evens == [0, 0, 1, 2, 2, 3, 4, 5, 6] and odds == [0, 0, 1, 2, 3, 3, 4, 5, 6]
    h1, h2 = bo.combine_split([2, 4, 1, 3], [2, 3, 3, 2])
    assert h1 == [2, 2, 1, 3] and h2 == [2, 4, 3, 3]
    with pytest.raises(KeyError):
        bo.combine_split([1, 2, 3], [4])


def test_to_polynomial_oracle_on_the_reference_fixture():
    """The reference's one deterministic test THROUGH `ifft` (lookup/multiset.rs:290-309 `test_to_polynomial`, run there for BLS12-381 and
    BLS12-377 Fr): the multiset {1, ..., 7} on `EvaluationDomain::new(7 + 1)` -- seven evaluations, zero-extended to the eight-point domain
    by `ifft` (multiset.rs:194-202) -- interpolates to a polynomial of degree 7.  It pins little (the top coefficient is not zero) but it is
    reference-held: both restatements reproduce it, agree on all eight coefficients, and those evaluate back to (1, ..., 7, 0)."""
    from oracle import cpu
    cpu.build()
    for cid in (0, 1):
        cv = bo.CURVES[cid]
        p = cv.r
        coeffs = bo.ntt(cv, bo.KIND_IFFT, 3, [1, 2, 3, 4, 5, 6, 7])
        assert len(coeffs) == 8 and coeffs[7] != 0                                       # s_poly.degree() == 7
        w = cv.root_of_unity(3)
        assert [bo.horner(coeffs, pow(w, i, p), p) for i in range(8)] == [1, 2, 3, 4, 5, 6, 7, 0]
        x = np.array([bo.int_to_limbs(bo.to_mont(v, p, 1 << 256), 4) for v in range(1, 8)], dtype=np.uint64)
        got = cpu.ntt(cid, 1, 3, x)                                                      # in_len = 7 < 8: the C++ restatement extends too
        assert [bo.from_mont(bo.limbs_to_int(r), p, 1 << 256) for r in got] == coeffs


@pytest.mark.parametrize("cid", [0, 1])
def test_oracle_prover_and_oracle_verifier_agree(cid):
    """No GPU: `Prover::prove_with_preprocessed` restated on integers (oracle/prover_oracle.py) produces a proof of a satisfied
    128-row circuit with every gate type (tests/test_prover_gpu.py's builder) that `Proof::verify` restated on integers
    (oracle/verifier_oracle.py, pairing -> known tau) accepts; one wrong witness cell and it does not.  The two restatements share
    the widget formulas of bigint_oracle but nothing else: prover side = quotient by coset evaluations and a coset iFFT, verifier
    side = r_0 and the linearisation scalars from the evaluations."""
    import importlib.util
    from oracle import cpu, prover_oracle as po, verifier_oracle as vo, wire_oracle as wo
    cpu.build()
    src = open(os.path.join(ROOT, "tests", "test_prover_gpu.py")).read()
    ns = {"np": np, "bo": bo, "K": (1, 7, 13, 17),
          "prover": type("P", (), {"SELECTORS": ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
                                                 "q_variable_group_add", "q_lookup")})}
    exec(src[src.index("def add_gadgets"):src.index("def run_case")], ns)          # the circuit builder only (the module itself needs a GPU)
    assert importlib.util.find_spec("oracle.prover_oracle") is not None
    cv = bo.CURVES[cid]
    log_n, tau = 7, 0x7A5C0DE
    n = 1 << log_n
    ca, cd = bo.seeded_scalars(cv, 0x51, 2)
    key_map = {"q_fixed": "q_fixed_group_add", "q_var": "q_variable_group_add"}
    srs = cpu.srs_powers(cid, tau, n + 8)
    for broken in (False, True):
        sel, sigma, table, wires, pub = ns["build_circuit"](cv, log_n, 31 + cid, broken, (ca, cd))
        osel = {k: sel[key_map.get(k, k)] for k in po.KEYS}
        t = wo.PlonkTranscript(b"cpu only", cv)
        t.circuit_domain_sep(n)
        data, ch, polys = po.prove(cv, log_n, osel, sigma, table, wires, pub, t, po.cpp_committer(cpu, cid, cv, srs), ca, cd)
        dlog = {k: bo.horner(v, tau, cv.r) for k, v in polys.items()}
        t2 = wo.PlonkTranscript(b"cpu only", cv)
        t2.circuit_domain_sep(n)
        ok, vch, det = vo.verify_with_trapdoor(cv, log_n, data, t2, pub, dlog, tau, ca, cd)
        assert ok == (not broken), (broken, det["aw"], det["saw"])
        assert vch["z"] == ch["z"] and vch["saw"] == ch["saw"]
        if not broken:      # the commitments in the bytes are the polynomials at tau times G
            pr = det["proof"]
            for k in ("a_comm", "z_2_comm", "t_4_comm"):
                assert pr["commitments"][k] == bo.ec_mul(cv, dlog[k], (cv.gx, cv.gy))
            assert pr["aw_opening"] == bo.ec_mul(cv, dlog["aw_opening"], (cv.gx, cv.gy))


def test_cpu_restatement_against_published_points(oracle_cpu):
    """The C++ restatement (the checker of every -m gpu test) against points this repository did not compute: EIP-2537's G1 + G1,
    EIP-196's [2], [3], [9] (1, 2) and its "chfast1" scalar multiplication."""
    def limbs(cv, pt):
        R = 1 << (64 * cv.fq_limbs)
        return np.array(bo.int_to_limbs(bo.to_mont(pt[0], cv.q, R), cv.fq_limbs) + bo.int_to_limbs(bo.to_mont(pt[1], cv.q, R), cv.fq_limbs),
                        dtype=np.uint64)

    def msm(cid, pts, ks):
        cv = bo.CURVES[cid]
        sc = np.zeros((len(ks), 4), dtype=np.uint64)
        sc[:, 0] = ks
        out, inf = oracle_cpu.msm_g1(cid, np.stack([limbs(cv, p) for p in pts]), sc)
        assert not inf
        return out

    bls, bn = bo.BLS12_381, bo.BN254
    g, h = (bls.gx, bls.gy), (bn.gx, bn.gy)
    assert np.array_equal(msm(0, [g, g], [1, 1]), limbs(bls, EXT_BLS_2G)) and np.array_equal(msm(0, [g], [2]), limbs(bls, EXT_BLS_2G))
    assert np.array_equal(msm(1, [h, h], [1, 1]), limbs(bn, EXT_BN254_2G)) and np.array_equal(msm(1, [h, h], [1, 2]), limbs(bn, EXT_BN254_3G))
    assert np.array_equal(msm(1, [h], [9]), limbs(bn, EXT_BN254_9G))
    assert np.array_equal(msm(1, [EXT_BN254_MUL["point"]], [EXT_BN254_MUL["scalar"]]), limbs(bn, EXT_BN254_MUL["result"]))
