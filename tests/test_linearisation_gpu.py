"""Round 5 before its commitments (linearisation_poly.rs:164-350): batched polynomial evaluation, scalar-weighted polynomial
sums and the whole `compute` through the C ABI vs the big-int restatement (oracle/bigint_oracle.py::linearisation, itself
pinned on the quotient oracle by tests/test_oracle.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import _lib, linearisation  # noqa: E402
from ark_plonk_amd.curves import fr_from_mont, fr_to_mont  # noqa: E402
from oracle import bigint_oracle as bo  # noqa: E402
from tests.golden.gen_golden_linearisation import CHALLENGES, EVAL_NAMES, LOG_N, case  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(ROOT, "tests", "golden", "linearisation.npz"))
KEY = {"q_fixed": "q_fixed_group_add", "q_var": "q_variable_group_add", "sigma0": "left_sigma", "sigma1": "right_sigma",
       "sigma2": "out_sigma", "sigma3": "fourth_sigma"}
CH = {"range": "range_challenge", "logic": "logic_challenge", "fixed": "fixed_base_challenge", "var": "var_base_challenge",
      "lookup": "lookup_challenge", "z": "z_challenge"}


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1, 4)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("cid", [0, 1])
def test_evaluate_batch_vs_horner_ragged(cid, ctx):
    """Lengths around the 64-coefficient chunk and the 256-lane fold, an empty polynomial, a constant; points 0, 1, r-1 and
    random -- one batch of 14, every value against Horner's rule on integers (ark-poly `DensePolynomial::evaluate`)."""
    cv = bo.CURVES[cid]
    lens = [0, 1, 2, 63, 64, 65, 127, 1000, 4096, 16383, 16384, 16385, 40001, 300]
    polys = [bo.seeded_scalars(cv, 0xE000 + k, m) for k, m in enumerate(lens)]
    pts = bo.seeded_scalars(cv, 0xE100, len(lens))
    pts[3], pts[4], pts[5] = 0, 1, cv.r - 1
    got = linearisation.evaluate_batch([dev(fr_to_mont(cid, p)) if p else dev(np.zeros((0, 4), dtype=np.uint64)) for p in polys],
                                       fr_to_mont(cid, pts), cid, ctx)
    assert fr_from_mont(cid, got) == [bo.horner(p, x, cv.r) for p, x in zip(polys, pts)]


@pytest.mark.parametrize("cid", [0, 1])
def test_lincomb_vs_oracle_ragged_and_aliased(cid, ctx):
    """32 terms (the entry point's maximum) of different lengths; a shorter and a longer out_len; the output aliasing an input."""
    cv = bo.CURVES[cid]
    p = cv.r
    lens = [257 - 7 * k for k in range(32)]
    polys = [bo.seeded_scalars(cv, 0xE200 + k, m) for k, m in enumerate(lens)]
    cf = bo.seeded_scalars(cv, 0xE2F0, 32)
    cf[5], cf[6] = 0, p - 1
    d_polys = [dev(fr_to_mont(cid, q)) for q in polys]
    want = bo.poly_add(*[bo.poly_scale(q, c, p) for q, c in zip(polys, cf)], p=p)
    assert fr_from_mont(cid, host(linearisation.lincomb(d_polys, fr_to_mont(cid, cf), curve=cid, ctx=ctx))) == want
    assert fr_from_mont(cid, host(linearisation.lincomb(d_polys, fr_to_mont(cid, cf), out_len=100, curve=cid, ctx=ctx))) == want[:100]
    longer = fr_from_mont(cid, host(linearisation.lincomb(d_polys, fr_to_mont(cid, cf), out_len=300, curve=cid, ctx=ctx)))
    assert longer == want + [0] * (300 - len(want))
    out = linearisation.lincomb(d_polys, fr_to_mont(cid, cf), out=d_polys[0], curve=cid, ctx=ctx)
    assert out.data_ptr() == d_polys[0].data_ptr() and fr_from_mont(cid, host(out)) == want


def _compute(cid, log_n, key, polys, ch, ctx):
    dom = zk.Radix2EvaluationDomain.new(1 << log_n, cid, ctx)
    lin, ev = linearisation.compute(dom, {KEY.get(k, k): dev(fr_to_mont(cid, v)) for k, v in key.items()},
                                    {CH.get(k, k): fr_to_mont(cid, [v])[0] for k, v in ch.items()},
                                    {k: dev(fr_to_mont(cid, v)) for k, v in polys.items()})
    return host(lin), ev


@pytest.mark.parametrize("cid", [0, 1])
def test_compute_golden(cid, ctx):
    cv = bo.CURVES[cid]
    key, polys, ch = case(cv, LOG_N, 0x9100 + 0x100 * cid)
    lin, ev = _compute(cid, LOG_N, key, polys, ch, ctx)
    assert np.array_equal(lin, G[f"{cv.name}_lin"])
    assert np.array_equal(np.stack([ev[k] for k in EVAL_NAMES]), G[f"{cv.name}_evals"])
    assert tuple(EVAL_NAMES) == linearisation.PROOF_EVALS + linearisation.CUSTOM_EVALS


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("log_n", [1, 6, 10])
def test_compute_vs_bigint_oracle(cid, log_n, ctx):
    """n = 2, 64, 1024: every coefficient of the linearisation polynomial and the 23 evaluations."""
    cv = bo.CURVES[cid]
    key, polys, ch = case(cv, log_n, 0xA000 + 0x40 * log_n + cid)
    want_lin, want_ev = bo.linearisation(cv, log_n, key, polys, ch)
    lin, ev = _compute(cid, log_n, key, polys, ch, ctx)
    assert fr_from_mont(cid, lin) == want_lin
    assert {k: fr_from_mont(cid, v.reshape(1, 4))[0] for k, v in ev.items()} == want_ev


def test_full_size_evaluations_and_linearity(ctx, oracle_cpu):
    """n = 2^20 (BASELINE config 2): the 23 evaluations of a proof in one call -- one of them against the C++ restatement
    (p(z) = p_0 + z * w_0 with w the witness polynomial of ark_cpu.cpp's synthetic division) -- and, size-independent, the
    evaluation of a 19-term sum equals the sum of the evaluations."""
    import torch
    cid, n = 0, 1 << 20
    g = torch.Generator(device="cuda").manual_seed(11)
    polys = []
    for _ in range(19):
        t = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        t[:, 3] &= (1 << 60) - 1                      # < 2^252 < r: a reduced Montgomery element
        polys.append(t)
    cv = bo.CURVES[cid]
    z, x = bo.seeded_scalars(cv, 0xE300, 2)
    pts = fr_to_mont(cid, [z] * 16 + [x] * 3)
    vals = fr_from_mont(cid, linearisation.evaluate_batch(polys, pts, cid, ctx))
    p0 = host(polys[0])
    w = oracle_cpu.kzg_witness(cid, p0, fr_to_mont(cid, [z])[0])
    assert vals[0] == (fr_from_mont(cid, p0[:1])[0] + z * fr_from_mont(cid, w[:1])[0]) % cv.r
    cf = bo.seeded_scalars(cv, 0xE310, 19)
    comb = linearisation.lincomb(polys, fr_to_mont(cid, cf), curve=cid, ctx=ctx)
    at_x = fr_from_mont(cid, linearisation.evaluate_batch(polys + [comb], fr_to_mont(cid, [x] * 20), cid, ctx))
    assert at_x[19] == sum(c * v for c, v in zip(cf, at_x[:19])) % cv.r and at_x[16:19] == vals[16:19]


def test_argument_errors(ctx):
    import torch
    t = torch.zeros((8, 4), dtype=torch.int64, device="cuda")
    ptrs = (ctypes.c_void_p * 33)(*([t.data_ptr()] * 33))
    lens = (ctypes.c_size_t * 33)(*([8] * 33))
    pts = np.zeros((33, 4), dtype=np.uint64)
    out = np.zeros((33, 4), dtype=np.uint64)
    L = _lib.lib()
    ctx.use_torch_stream()
    assert L.zk_poly_evaluate_dev(ctx.handle, 0, 33, ptrs, lens, pts.ctypes.data, out.ctypes.data) == _lib.ZK_ERR_UNSUPPORTED
    assert L.zk_poly_lincomb_dev(ctx.handle, 0, 33, ptrs, lens, pts.ctypes.data, t.data_ptr(), 8) == _lib.ZK_ERR_UNSUPPORTED
    pts[0] = 0xFFFFFFFFFFFFFFFF                       # not a reduced field element
    assert L.zk_poly_evaluate_dev(ctx.handle, 0, 1, ptrs, lens, pts.ctypes.data, out.ctypes.data) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_poly_lincomb_dev(ctx.handle, 0, 1, ptrs, lens, pts.ctypes.data, t.data_ptr(), 8) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_poly_evaluate_dev(ctx.handle, 5, 1, ptrs, lens, out.ctypes.data, out.ctypes.data) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_poly_evaluate_dev(ctx.handle, 0, 1, None, lens, pts.ctypes.data, out.ctypes.data) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_poly_evaluate_dev(ctx.handle, 0, 0, None, None, None, None) == 0
