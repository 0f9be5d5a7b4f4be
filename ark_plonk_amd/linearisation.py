"""Host mirror of `plonk-core/src/proof_system/linearisation_poly.rs::compute` (lines 164-350): round 5 of the prover before
its commitments and openings.

The reference does two kinds of O(n) work there, both on the CPU: 23 `DensePolynomial::evaluate` calls (16 polynomials at
z, 7 at z*omega -- the `ProofEvaluations` of the proof) and the linearisation polynomial, a sum of 19 scalar * polynomial
terms.  Here both run on the device-resident coefficient vectors (`zk_poly_evaluate_dev`: one launch pair for the 23;
`zk_poly_lincomb_dev`: one pass over the 19); the scalar formulas in between (a few hundred field operations) are the
reference's, restated on Python integers.  A Rust caller keeps its own scalar code and binds only the two entry points.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import check, lib
from .context import check_dev_tensor, ptr_of
from .curves import fr_from_mont, fr_to_mont, get_curve

# the evaluations of one proof, in the order they are batched (linearisation_poly.rs:203-261)
EVAL_AT_Z = ("w_l", "w_r", "w_o", "w_4", "left_sigma", "right_sigma", "out_sigma", "q_arith", "q_lookup", "q_c", "q_l", "q_r",
             "h1", "h2", "f", "table")
EVAL_AT_ZW = ("z", "w_l", "w_r", "w_4", "z2", "h1", "table")
# `ProofEvaluations` in declaration order (linearisation_poly.rs:34-161): wire, permutation, lookup -- what zk_proof.evals takes
PROOF_EVALS = ("a_eval", "b_eval", "c_eval", "d_eval", "left_sigma_eval", "right_sigma_eval", "out_sigma_eval", "permutation_eval",
               "q_lookup_eval", "z2_next_eval", "h1_eval", "h1_next_eval", "h2_eval", "f_eval", "table_eval", "table_next_eval")
# `CustomEvaluations.vals` in push order (linearisation_poly.rs:243-253)
CUSTOM_EVALS = ("q_arith_eval", "q_c_eval", "q_l_eval", "q_r_eval", "a_next_eval", "b_next_eval", "d_next_eval")
POLYS = ("w_l", "w_r", "w_o", "w_4", "t_1", "t_2", "t_3", "t_4", "z", "z2", "f", "h1", "h2", "table")
KEY_POLYS = ("q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add",
             "q_lookup", "left_sigma", "right_sigma", "out_sigma", "fourth_sigma")
CHALLENGES = ("alpha", "beta", "gamma", "delta", "epsilon", "zeta", "range_challenge", "logic_challenge", "fixed_base_challenge",
              "var_base_challenge", "lookup_challenge", "z_challenge", "coeff_a", "coeff_d")
K1, K2, K3 = 7, 13, 17       # permutation/constants.rs:12-22


def _ptrs(polys, ctx):
    k = len(polys)
    ptrs = (ctypes.c_void_p * k)()
    lens = (ctypes.c_size_t * k)()
    for i, p in enumerate(polys):
        lens[i] = check_dev_tensor(p, 4, ctx.device)
        ptrs[i] = p.data_ptr()
    return ptrs, lens


def evaluate_batch(polys, points_mont, curve="bls12_381", ctx=None) -> np.ndarray:
    """out[k] = polys[k](points[k]): device coefficient tensors, (k, 4) Montgomery points -> (k, 4) uint64 Montgomery values."""
    from .context import default_context
    cv = get_curve(curve)
    ctx = ctx or default_context(polys[0].device.index)
    pts = np.ascontiguousarray(points_mont, dtype=np.uint64).reshape(len(polys), 4)
    out = np.zeros((len(polys), 4), dtype=np.uint64)
    ctx.use_torch_stream()
    for lo in range(0, len(polys), 32):                  # the entry point takes 32 pairs per call
        part = polys[lo:lo + 32]
        ptrs, lens = _ptrs(part, ctx)
        check(lib().zk_poly_evaluate_dev(ctx.handle, cv.curve_id, len(part), ptrs, lens, ptr_of(pts[lo:lo + 32]), ptr_of(out[lo:lo + 32])),
              "zk_poly_evaluate_dev")
    return out


def lincomb(polys, coeffs_mont, out_len=None, out=None, curve="bls12_381", ctx=None):
    """sum_k coeffs[k] * polys[k] over out_len coefficients (default: the longest input), as a device tensor."""
    import torch
    from .context import default_context
    cv = get_curve(curve)
    ctx = ctx or default_context(polys[0].device.index)
    cf = np.ascontiguousarray(coeffs_mont, dtype=np.uint64).reshape(len(polys), 4)
    ptrs, lens = _ptrs(polys, ctx)
    if out_len is None:
        out_len = max(lens) if len(polys) else 0
    if out is None:
        out = torch.empty((out_len, 4), dtype=torch.int64, device=polys[0].device)
    elif check_dev_tensor(out, 4, ctx.device) < out_len:
        raise ValueError("out is shorter than out_len")
    ctx.use_torch_stream()
    check(lib().zk_poly_lincomb_dev(ctx.handle, cv.curve_id, len(polys), ptrs, lens, ptr_of(cf), out.data_ptr(), out_len), "zk_poly_lincomb_dev")
    return out[:out_len]


# ---- the scalar side: widget `constraints` on evaluations (the same functions the quotient uses point by point) ---------
def _delta(f, p):                       # widget/range.rs:66-74, widget/logic.rs:94-102
    return f * (f - 1) * (f - 2) * (f - 3) % p


def _range(sep, a, b, c, d, d_next, p):                                  # widget/range.rs:47-63
    k = sep * sep % p
    return (_delta(c - 4 * d, p) + _delta(b - 4 * c, p) * k + _delta(a - 4 * b, p) * k * k + _delta(d_next - 4 * a, p) * k * k * k) * sep % p


def _logic(sep, a, b, c, d, a_next, b_next, d_next, q_c, p):             # widget/logic.rs:65-133
    k = sep * sep % p
    da, db, dd, w = a_next - 4 * a, b_next - 4 * b, d_next - 4 * d, c
    big_f = w * (w * (4 * w - 18 * (da + db) + 81) + 18 * (da * da + db * db) - 81 * (da + db) + 83)
    xor_and = q_c * (9 * dd - 3 * (da + db)) + 3 * (da + db + dd) - 2 * big_f
    return (_delta(da, p) + _delta(db, p) * k + _delta(dd, p) * k ** 2 + (w - da * db) * k ** 3 + xor_and * k ** 4) * sep % p


def _fixed_base(sep, a, b, c, d, a_next, b_next, d_next, q_l, q_r, q_c, ca, cd, p):   # widget/ecc/fixed_base_scalar_mul.rs:88-156
    k = sep * sep % p
    bit = d_next - d - d
    y_alpha = bit * bit * (q_r - 1) + 1
    x_alpha = q_l * bit
    xy_consistency = (bit * q_c - c) * k
    x_acc = (a_next + a_next * c * a * b * cd - (x_alpha * b + y_alpha * a)) * k ** 2
    y_acc = (b_next - b_next * c * a * b * cd - (y_alpha * b - ca * x_alpha * a)) * k ** 3
    return (bit * (bit - 1) * (bit + 1) + x_acc + y_acc + xy_consistency) * sep % p


def _curve_add(sep, a, b, c, d, a_next, b_next, d_next, ca, cd, p):      # widget/ecc/curve_addition.rs:62-97
    k = sep * sep % p
    x1, x3, y1, y3, x2, y2, x1y2 = a, a_next, b, b_next, c, d, d_next
    y1x2, y1y2, x1x2 = y1 * x2, y1 * y2, x1 * x2
    x3_c = (x1y2 + y1x2 - (x3 + x3 * cd * x1y2 * y1x2)) * k
    y3_c = (y1y2 - ca * x1x2 - (y3 - y3 * cd * x1y2 * y1x2)) * k * k
    return (x1 * y2 - x1y2 + x3_c + y3_c) * sep % p


def compute(domain, key: dict, challenges: dict, polys: dict):
    """linearisation_poly.rs:164-350.  domain: the size-n circuit domain; key / polys: name -> device coefficient tensor for
    every name in KEY_POLYS / POLYS; challenges: name -> 4 Montgomery limbs for every name in CHALLENGES (coeff_a / coeff_d =
    the embedded curve's coefficients, as for the quotient).
    Returns (linearisation polynomial: (n, 4) device tensor, evaluations: name -> 4 Montgomery limbs for PROOF_EVALS + CUSTOM_EVALS)."""
    cv = domain.curve
    p = cv.r
    ctx = domain._ctx_for(polys["w_l"])
    ch = {name: fr_from_mont(cv, np.ascontiguousarray(challenges[name], dtype=np.uint64).reshape(1, 4))[0] for name in CHALLENGES}
    n = domain.size()
    z = ch["z_challenge"]
    zw = z * fr_from_mont(cv, np.asarray(domain.group_gen(), dtype=np.uint64).reshape(1, 4))[0] % p     # shifted_z_challenge (:200-201)
    src = {**key, **polys}
    batch = [src[name] for name in EVAL_AT_Z] + [src[name] for name in EVAL_AT_ZW]
    pts = fr_to_mont(cv, [z] * len(EVAL_AT_Z) + [zw] * len(EVAL_AT_ZW))
    vals = fr_from_mont(cv, evaluate_batch(batch, pts, cv, ctx))
    at_z = dict(zip(EVAL_AT_Z, vals[:len(EVAL_AT_Z)]))
    at_zw = dict(zip(EVAL_AT_ZW, vals[len(EVAL_AT_Z):]))
    a, b, c, d = at_z["w_l"], at_z["w_r"], at_z["w_o"], at_z["w_4"]
    a_next, b_next, d_next = at_zw["w_l"], at_zw["w_r"], at_zw["w_4"]
    ev = {"a_eval": a, "b_eval": b, "c_eval": c, "d_eval": d,
          "left_sigma_eval": at_z["left_sigma"], "right_sigma_eval": at_z["right_sigma"], "out_sigma_eval": at_z["out_sigma"],
          "permutation_eval": at_zw["z"],
          "q_lookup_eval": at_z["q_lookup"], "z2_next_eval": at_zw["z2"], "h1_eval": at_z["h1"], "h1_next_eval": at_zw["h1"],
          "h2_eval": at_z["h2"], "f_eval": at_z["f"], "table_eval": at_z["table"], "table_next_eval": at_zw["table"],
          "q_arith_eval": at_z["q_arith"], "q_c_eval": at_z["q_c"], "q_l_eval": at_z["q_l"], "q_r_eval": at_z["q_r"],
          "a_next_eval": a_next, "b_next_eval": b_next, "d_next_eval": d_next}
    # :263-274: Z_H(z) = z^n - 1, L_1(z) = Z_H(z) / (n (z - 1))   (proof.rs:622-633)
    z_n = pow(z, n, p)
    vanishing = (z_n - 1) % p
    l1 = vanishing * pow(n * (z - 1) % p, -1, p) % p
    al, be, ga = ch["alpha"], ch["beta"], ch["gamma"]
    de, ep, ze, ls = ch["delta"], ch["epsilon"], ch["zeta"], ch["lookup_challenge"]
    ca, cd = ch["coeff_a"], ch["coeff_d"]
    q_arith, q_c, q_l, q_r = ev["q_arith_eval"], ev["q_c_eval"], ev["q_l_eval"], ev["q_r_eval"]
    terms = []     # (polynomial, scalar)
    # widget/arithmetic.rs:66-82: (q_m ab + q_l a + q_r b + q_o c + q_4 d + q_c) * q_arith_eval
    terms += [(key["q_m"], a * b * q_arith), (key["q_l"], a * q_arith), (key["q_r"], b * q_arith), (key["q_o"], c * q_arith),
              (key["q_4"], d * q_arith), (key["q_c"], q_arith)]
    # widget/mod.rs:96-104: selector(X) * constraints(separation challenge, evaluations)   (:353-411)
    terms += [(key["q_range"], _range(ch["range_challenge"], a, b, c, d, d_next, p)),
              (key["q_logic"], _logic(ch["logic_challenge"], a, b, c, d, a_next, b_next, d_next, q_c, p)),
              (key["q_fixed_group_add"], _fixed_base(ch["fixed_base_challenge"], a, b, c, d, a_next, b_next, d_next, q_l, q_r, q_c, ca, cd, p)),
              (key["q_variable_group_add"], _curve_add(ch["var_base_challenge"], a, b, c, d, a_next, b_next, d_next, ca, cd, p))]
    # proof_system/permutation.rs:156-291
    bz = be * z
    ident = (a + bz + ga) * (b + K1 * bz + ga) * (c + K2 * bz + ga) * (d + K3 * bz + ga) * al            # * z(X)
    copy = (a + be * ev["left_sigma_eval"] + ga) * (b + be * ev["right_sigma_eval"] + ga) * (c + be * ev["out_sigma_eval"] + ga) \
        * be * ev["permutation_eval"] * al                                                               # * -fourth_sigma(X)
    terms += [(polys["z"], ident + l1 * al * al), (key["fourth_sigma"], -copy)]
    # widget/lookup.rs:154-203
    opd = 1 + de
    e1d = ep * opd
    compressed = a + ze * (b + ze * (c + ze * d))                                                        # util.rs:152 lc()
    terms += [(key["q_lookup"], (compressed - ev["f_eval"]) * ls),
              (polys["z2"], opd * (ep + ev["f_eval"]) * (e1d + ev["table_eval"] + de * ev["table_next_eval"]) * ls * ls + l1 * ls ** 3),
              (polys["h1"], -ev["z2_next_eval"] * ls * ls * (e1d + ev["h2_eval"] + de * ev["h1_next_eval"]))]
    # :322-331: -Z_H(z) (t_1 + z^n t_2 + z^2n t_3 + z^3n t_4)
    terms += [(polys["t_1"], -vanishing), (polys["t_2"], -vanishing * z_n), (polys["t_3"], -vanishing * z_n ** 2),
              (polys["t_4"], -vanishing * z_n ** 3)]
    lin = lincomb([t[0] for t in terms], fr_to_mont(cv, [t[1] % p for t in terms]), curve=cv, ctx=ctx)
    names = PROOF_EVALS + CUSTOM_EVALS
    mont = fr_to_mont(cv, [ev[name] for name in names])
    return lin, {name: mont[i] for i, name in enumerate(names)}
