"""Pointwise quotient kernel vs the big-int restatement of
plonk-core/src/proof_system/quotient_poly.rs:34-178 and every widget it sums."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import quotient  # noqa: E402
from oracle import bigint_oracle as bo  # noqa: E402

pytestmark = pytest.mark.gpu
GQ = np.load(os.path.join(ROOT, "tests", "golden", "quotient.npz"))
# oracle column / challenge names -> the C struct's field names
COL = {"q_fixed": "q_fixed_group_add", "q_var": "q_variable_group_add"}
CH = {"range": "range_challenge", "logic": "logic_challenge", "fixed": "fixed_base_challenge", "var": "var_base_challenge",
      "lookup": "lookup_challenge"}


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def run(cid, log_n, col_mont, ch_mont, ctx):
    dom = zk.Radix2EvaluationDomain.new(1 << log_n, cid, ctx)
    cols = {COL.get(k, k): dev(v) for k, v in col_mont.items() if not k.startswith("sigma")}
    sig = [dev(col_mont[f"sigma{k}"]) for k in range(4)]
    chal = {CH.get(k, k): v for k, v in ch_mont.items()}
    return quotient.compute_quotient_evals(dom, cols, sig, chal).cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("cid", [0, 1])
def test_golden(cid, ctx):
    cv = bo.CURVES[cid]
    col = {name: GQ[f"{cv.name}_col_{name}"] for name in bo.QUOTIENT_COLS}
    ch = dict(zip(bo.QUOTIENT_CHALLENGES, GQ[f"{cv.name}_challenges"]))
    assert np.array_equal(run(cid, 2, col, ch, ctx), GQ[f"{cv.name}_quotient"])


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("log_n", [0, 3, 6])
def test_vs_bigint_oracle_every_point(cid, log_n, ctx):
    """All selectors random, so every widget contributes at every point; 4n = 4, 32, 256 (one and several lanes per
    workgroup row, the i+4 wrap-around inside and across workgroups)."""
    cv = bo.CURVES[cid]
    n4 = 4 << log_n
    col = {name: bo.seeded_scalars(cv, 0xB00 + 64 * log_n + k, n4) for k, name in enumerate(bo.QUOTIENT_COLS)}
    ch = dict(zip(bo.QUOTIENT_CHALLENGES, bo.seeded_scalars(cv, 0xBF0 + log_n, len(bo.QUOTIENT_CHALLENGES))))
    got = run(cid, log_n, {k: zk.curves.fr_to_mont(cid, v) for k, v in col.items()},
              {k: zk.curves.fr_to_mont(cid, [v])[0] for k, v in ch.items()}, ctx)
    assert zk.curves.fr_from_mont(cid, got) == bo.quotient_evals(cv, log_n, col, ch)


def test_sampled_points_at_2_16(ctx):
    """n = 2^16 (4n = 2^18 points): device inputs, the oracle evaluated at sampled indices including both ends."""
    import torch
    cid, cv, log_n = 0, bo.CURVES[0], 16
    n4 = 4 << log_n
    g = torch.Generator(device="cuda").manual_seed(21)
    cols_t = {name: torch.randint(0, 1 << 62, (n4, 4), dtype=torch.int64, device="cuda", generator=g) for name in bo.QUOTIENT_COLS}
    ch = dict(zip(bo.QUOTIENT_CHALLENGES, bo.seeded_scalars(cv, 0xC00, len(bo.QUOTIENT_CHALLENGES))))
    dom = zk.Radix2EvaluationDomain.new(1 << log_n, cid, ctx)
    cols = {COL.get(k, k): v for k, v in cols_t.items() if not k.startswith("sigma")}
    out = quotient.compute_quotient_evals(dom, cols, [cols_t[f"sigma{k}"] for k in range(4)],
                                          {CH.get(k, k): zk.curves.fr_to_mont(cid, [v])[0] for k, v in ch.items()})
    out_h = out.cpu().numpy().view(np.uint64)
    rnd = random.Random(9)
    idx = [0, 1, 3, 4, 255, 256, 1023, 1024, n4 - 5, n4 - 4, n4 - 1] + [rnd.randrange(n4) for _ in range(21)]

    class Lazy:   # the oracle indexes col[name][i]: convert single elements on demand
        def __init__(self, t):
            self.t = t

        def __getitem__(self, i):
            return zk.curves.fr_from_mont(cid, self.t[i:i + 1].cpu().numpy().view(np.uint64))[0]

    lazy = {k: Lazy(v) for k, v in cols_t.items()}
    for i in idx:
        assert zk.curves.fr_from_mont(cid, out_h[i:i + 1])[0] == bo.quotient_at(cv, log_n, i, lazy, ch), i


def test_compute_is_divisible_for_a_satisfied_arithmetic_circuit(ctx):
    """End-to-end `compute` (coset FFTs + kernel + coset iFFT) on a satisfied toy circuit: only arithmetic gates
    a*b - c = 0, identity permutation (z = 1), empty lookup argument (z2 = 1, t = h1 = h2 = f = 0, delta chosen freely).
    The numerator then vanishes on the domain, so the quotient has degree < 3n: its top n coefficients are zero --
    the property the reference's tests obtain from the pairing check (circuit.rs:392-463)."""
    import torch
    cid, cv, log_n = 0, bo.CURVES[0], 5
    n = 1 << log_n
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    dom4 = zk.Radix2EvaluationDomain.new(4 * n, cid, ctx)
    p = cv.r
    a = bo.seeded_scalars(cv, 0xD00, n)
    b = bo.seeded_scalars(cv, 0xD01, n)
    c = [x * y % p for x, y in zip(a, b)]
    d = [0] * n
    w = cv.root_of_unity(log_n)
    roots = [pow(w, i, p) for i in range(n)]
    m = lambda xs: dev(zk.curves.fr_to_mont(cid, xs))   # noqa: E731
    ifft = lambda xs: dom.ifft(m(xs))                    # noqa: E731
    zero, ones = [0] * n, [1] * n
    polys = {"w_l": ifft(a), "w_r": ifft(b), "w_o": ifft(c), "w_4": ifft(d), "z": ifft(ones), "z2": ifft(ones),
             "f": ifft(zero), "table": ifft(zero), "h1": ifft(zero), "h2": ifft(zero), "pi": ifft(zero)}
    cf = lambda xs: dom4.coset_fft(ifft(xs))             # noqa: E731
    key = {"q_m": cf(ones), "q_l": cf(zero), "q_r": cf(zero), "q_o": cf([p - 1] * n), "q_4": cf(zero), "q_c": cf(zero),
           "q_arith": cf(ones), "q_range": cf(zero), "q_logic": cf(zero), "q_fixed_group_add": cf(zero),
           "q_variable_group_add": cf(zero), "q_lookup": cf(zero)}
    sig = [cf([bo.PERM_K[k] * r % p for r in roots]) for k in range(4)]      # identity permutation: sigma_k(w^i) = K_k w^i
    chv = dict(zip(quotient.CHALLENGES, bo.seeded_scalars(cv, 0xD10, len(quotient.CHALLENGES))))
    chal = {k: zk.curves.fr_to_mont(cid, [v])[0] for k, v in chv.items()}
    t = quotient.compute(dom, dom4, polys, key, sig, chal).cpu().numpy().view(np.uint64)
    assert not t[3 * n:].any(), "quotient of a satisfied circuit must have degree < 3n"
    assert t[:3 * n].any()
    # break one gate: the numerator no longer vanishes on the domain and the division leaves a full-degree polynomial
    c[7] = (c[7] + 1) % p
    polys["w_o"] = ifft(c)
    t_bad = quotient.compute(dom, dom4, polys, key, sig, chal).cpu().numpy().view(np.uint64)
    assert all(t_bad[k].any() for k in range(4 * n - 4, 4 * n))


def test_permutation_argument_end_to_end(ctx):
    """N2 -> NTT -> N1 together on a circuit with real copy constraints: z is built on the device from the wire and
    sigma columns (zk_perm_product_dev), interpolated, and fed with the sigma polynomials to the quotient kernel using
    the same beta / gamma.  With all gate selectors off the numerator is the permutation argument alone, which
    vanishes on the domain exactly when z is the right grand product.  The numerator has degree <= 5n-5 (four wire
    factors and z), so an exact quotient has degree <= 4n-5: its last four coefficients are zero, while the interpolant
    of a non-divisible numerator / Z_H has no reason to have any zero coefficient."""
    from ark_plonk_amd import permutation
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from gen_golden_gp import valid_permutation
    cid, cv, log_n = 0, bo.CURVES[0], 5
    n, p = 1 << log_n, cv.r
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    dom4 = zk.Radix2EvaluationDomain.new(4 * n, cid, ctx)
    wires, sigmas = valid_permutation(cv, log_n, 0xE00)
    m = lambda xs: dev(zk.curves.fr_to_mont(cid, xs))   # noqa: E731
    chv = dict(zip(quotient.CHALLENGES, bo.seeded_scalars(cv, 0xE10, len(quotient.CHALLENGES))))
    chal = {k: zk.curves.fr_to_mont(cid, [v])[0] for k, v in chv.items()}
    w_dev, s_dev = [m(w) for w in wires], [m(sg) for sg in sigmas]
    z_evals, last = permutation.permutation_evals(dom, w_dev, s_dev, chal["beta"], chal["gamma"], return_last=True)
    assert zk.curves.fr_from_mont(cid, last.reshape(1, 4))[0] == 1
    zero, ones = [0] * n, [1] * n
    ifft = lambda t: dom.ifft(t)                         # noqa: E731
    polys = {"w_l": ifft(w_dev[0]), "w_r": ifft(w_dev[1]), "w_o": ifft(w_dev[2]), "w_4": ifft(w_dev[3]), "z": ifft(z_evals),
             "z2": ifft(m(ones)), "f": ifft(m(zero)), "table": ifft(m(zero)), "h1": ifft(m(zero)), "h2": ifft(m(zero)), "pi": ifft(m(zero))}
    cfz = dom4.coset_fft(ifft(m(zero)))
    key = {name: cfz for name in quotient.COLUMNS[12:]}
    sig = [dom4.coset_fft(ifft(sd)) for sd in s_dev]
    t = quotient.compute(dom, dom4, polys, key, sig, chal).cpu().numpy().view(np.uint64)
    assert not t[4 * n - 4:].any() and t[3 * n:4 * n - 4].any()
    # a z built with another gamma is not the grand product of this argument
    chal2 = dict(chal)
    chal2["gamma"] = zk.curves.fr_to_mont(cid, [chv["gamma"] + 1])[0]
    t_bad = quotient.compute(dom, dom4, polys, key, sig, chal2).cpu().numpy().view(np.uint64)
    assert all(t_bad[k].any() for k in range(4 * n - 4, 4 * n))


def test_bad_arguments_are_error_codes(ctx):
    """Null columns, mismatched lengths and oversized domains come back as codes / ValueError, never as a fault."""
    import ctypes
    import torch
    from ark_plonk_amd import _lib, permutation
    L = _lib.lib()
    args = quotient.QuotientArgs()                       # every pointer null
    out = torch.empty((16, 4), dtype=torch.int64, device="cuda")
    assert L.zk_quotient_evals_dev(ctx.handle, 0, 2, ctypes.byref(args), out.data_ptr()) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_quotient_evals_dev(ctx.handle, 0, 31, ctypes.byref(args), out.data_ptr()) == _lib.ZK_ERR_DOMAIN_TOO_LARGE
    assert L.zk_quotient_evals_dev(ctx.handle, 7, 2, ctypes.byref(args), out.data_ptr()) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_quotient_evals_dev(None, 0, 2, ctypes.byref(args), out.data_ptr()) == _lib.ZK_ERR_BAD_ARG
    dom = zk.Radix2EvaluationDomain.new(8, 0, ctx)
    cols = [torch.zeros((8, 4), dtype=torch.int64, device="cuda") for _ in range(8)]
    one = zk.curves.fr_to_mont(0, [1])[0]
    with pytest.raises(ValueError):
        permutation.permutation_evals(dom, cols[:4], cols[4:7] + [cols[7][:4]], one, one)       # a short sigma column
    with pytest.raises(ValueError):
        permutation.lookup_permutation_evals(ctx, 0, cols[0], cols[1][:3], cols[2], cols[3], one, one)
    nullp = (ctypes.c_void_p * 4)()
    assert L.zk_perm_product_dev(ctx.handle, 0, 3, nullp, nullp, one.ctypes.data, one.ctypes.data, out.data_ptr(), None) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_lookup_product_dev(ctx.handle, 0, 0, out.data_ptr(), out.data_ptr(), out.data_ptr(), out.data_ptr(), one.ctypes.data,
                                   one.ctypes.data, out.data_ptr(), None) == _lib.ZK_ERR_BAD_ARG
