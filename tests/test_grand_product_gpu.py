"""Grand-product builders on the device vs the big-int restatement of
plonk-core/src/permutation/mod.rs:652-822, the committed fixtures, and the closing property of a real
wire permutation at the benchmark size (the reference's own test, mod.rs:1243-1380: z(1) = 1 and the
product closes)."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import _lib, permutation  # noqa: E402
from oracle import bigint_oracle as bo  # noqa: E402

pytestmark = pytest.mark.gpu
GP = np.load(os.path.join(ROOT, "tests", "golden", "grand_product.npz"))


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


def mont(cid, xs):
    return zk.curves.fr_to_mont(cid, xs)


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("case", ["perm3", "perm6", "permv"])
def test_permutation_golden(cid, case, ctx):
    pre = f"{bo.CURVES[cid].name}_{case}"
    log_n = {"perm3": 3, "perm6": 6, "permv": 4}[case]
    dom = zk.Radix2EvaluationDomain.new(1 << log_n, cid, ctx)
    w = [dev(GP[f"{pre}_w{k}"]) for k in range(4)]
    s = [dev(GP[f"{pre}_s{k}"]) for k in range(4)]
    bg = GP[f"{pre}_beta_gamma"]
    z, last = permutation.permutation_evals(dom, w, s, bg[0], bg[1], return_last=True)
    assert np.array_equal(host(z), GP[f"{pre}_z"])
    exp_last = GP[f"{pre}_last"][0] if case != "permv" else mont(cid, [1])[0]
    assert np.array_equal(last, exp_last)


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("log_n", [3, 6])
def test_lookup_golden(cid, log_n, ctx):
    pre = f"{bo.CURVES[cid].name}_look{log_n}"
    cols = [dev(GP[f"{pre}_{nm}"]) for nm in ("f", "t", "h1", "h2")]
    de = GP[f"{pre}_delta_eps"]
    p, last = permutation.lookup_permutation_evals(ctx, cid, *cols, de[0], de[1], return_last=True)
    assert np.array_equal(host(p), GP[f"{pre}_p"])
    assert np.array_equal(last, GP[f"{pre}_last"][0])


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("log_n", [0, 1, 7, 11, 13])
def test_vs_bigint_oracle(cid, log_n, ctx):
    """Sizes below, at and above one scan chunk (64) and one term tile (2048); n = 1 and 2 included."""
    cv = bo.CURVES[cid]
    n = 1 << log_n
    cols = [bo.seeded_scalars(cv, 0x700 + 16 * log_n + k, n) for k in range(8)]
    beta, gamma, delta, eps = bo.seeded_scalars(cv, 0x7F0 + log_n, 4)
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    d = [dev(mont(cid, c)) for c in cols]
    z, last = permutation.permutation_evals(dom, d[:4], d[4:], mont(cid, [beta])[0], mont(cid, [gamma])[0], return_last=True)
    ez, elast = bo.perm_product(cv, log_n, cols[:4], cols[4:], beta, gamma)
    assert zk.curves.fr_from_mont(cid, host(z)) == ez
    assert zk.curves.fr_from_mont(cid, last.reshape(1, 4))[0] == elast
    p, lastp = permutation.lookup_permutation_evals(ctx, cid, d[0], d[1], d[2], d[3], mont(cid, [delta])[0], mont(cid, [eps])[0], return_last=True)
    ep, elastp = bo.lookup_product(cv, cols[0], cols[1], cols[2], cols[3], delta, eps)
    assert zk.curves.fr_from_mont(cid, host(p)) == ep
    assert zk.curves.fr_from_mont(cid, lastp.reshape(1, 4))[0] == elastp
    # the polynomial forms = product + ifft on device (what the reference returns)
    coeffs = permutation.compute_lookup_permutation_poly(dom, d[0], d[1], d[2], d[3], mont(cid, [delta])[0], mont(cid, [eps])[0])
    assert zk.curves.fr_from_mont(cid, host(coeffs)) == bo.ntt(cv, bo.KIND_IFFT, log_n, ep)


@pytest.mark.parametrize("cid", [0, 1])
def test_reference_fixture_only_left_wires(cid, ctx):
    """permutation/mod.rs:971-1092: the reference's deterministic four-gate permutation (sigma encodings and wire values as its test spells
    them out) through zk_perm_product_dev and the device iFFT, under the checks of mod.rs:1243-1380 with seeded beta / gamma: equal to the
    restatement, z[0] = 1, the product closes, z(1) = 1, degree 3."""
    from test_oracle import _reference_fixture_only_left_wires, _reference_fixture_two_gates
    cv = bo.CURVES[cid]
    # ... and the two-gate one of `test_basic_slow_permutation_poly` (mod.rs:1201-1233) on the two-point domain
    for log_n, (wires, sig) in ((2, _reference_fixture_only_left_wires(cv)), (1, _reference_fixture_two_gates(cv))):
        n = 1 << log_n
        dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
        dw, ds = [dev(mont(cid, c)) for c in wires], [dev(mont(cid, c)) for c in sig]
        for seed in range(5):
            beta, gamma = bo.seeded_scalars(cv, 0x9A0 + seed, 2)
            z, last = permutation.permutation_evals(dom, dw, ds, mont(cid, [beta])[0], mont(cid, [gamma])[0], return_last=True)
            zi = zk.curves.fr_from_mont(cid, host(z))
            assert zi == bo.perm_product(cv, log_n, wires, sig, beta, gamma)[0] and zi[0] == 1
            assert zk.curves.fr_from_mont(cid, last.reshape(1, 4))[0] == 1
            zp = zk.curves.fr_from_mont(cid, host(dom.ifft(z)))
            assert bo.horner(zp, 1, cv.r) == 1 and zp[n - 1] != 0


def test_lookup_length_not_a_power_of_two(ctx):
    cv = bo.CURVES[0]
    n = 1000
    cols = [bo.seeded_scalars(cv, 0x810 + k, n) for k in range(4)]
    delta, eps = bo.seeded_scalars(cv, 0x820, 2)
    p = permutation.lookup_permutation_evals(ctx, 0, *[dev(mont(0, c)) for c in cols], mont(0, [delta])[0], mont(0, [eps])[0])
    assert zk.curves.fr_from_mont(0, host(p)) == bo.lookup_product(cv, *cols, delta, eps)[0]


def test_zero_denominator_is_an_error(ctx):
    """w_0[5] + beta*sigma_0[5] + gamma = 0: the reference panics (inverse().unwrap(), mod.rs:727)."""
    cv = bo.CURVES[0]
    log_n, n = 6, 64
    cols = [bo.seeded_scalars(cv, 0x900 + k, n) for k in range(8)]
    beta, gamma = bo.seeded_scalars(cv, 0x910, 2)
    cols[0][5] = (-(beta * cols[4][5] + gamma)) % cv.r
    dom = zk.Radix2EvaluationDomain.new(n, 0, ctx)
    d = [dev(mont(0, c)) for c in cols]
    with pytest.raises(_lib.ZkError) as ei:
        permutation.permutation_evals(dom, d[:4], d[4:], mont(0, [beta])[0], mont(0, [gamma])[0])
    assert ei.value.code == _lib.ZK_ERR_NOT_INVERTIBLE
    with pytest.raises(ZeroDivisionError):
        bo.perm_product(cv, log_n, cols[:4], cols[4:], beta, gamma)


@pytest.mark.parametrize("log_n", [20])
def test_real_permutation_closes_at_benchmark_size(log_n, ctx):
    """A random copy-constraint permutation over the 4n wire slots with consistent wire values:
    z[0] = 1, the dropped (n+1)-th value is 1, and z[i+1] * den_i = z[i] * num_i at sampled rows."""
    import torch
    cid, cv = 0, bo.CURVES[0]
    n = 1 << log_n
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    one = mont(cid, [1])[0]
    # omega^i on the device: the evaluations of the polynomial X
    xpoly = torch.zeros((2, 4), dtype=torch.int64, device="cuda")
    xpoly[1] = torch.from_numpy(one.view(np.int64))
    roots = dom.fft(xpoly)
    kroots = []
    for K in bo.PERM_K:
        kk = dev(np.tile(mont(cid, [K]), (n, 1)))
        o = torch.empty_like(roots)
        ctx.use_torch_stream()
        _lib.check(_lib.lib().zk_fr_mul_dev(ctx.handle, cid, roots.data_ptr(), kk.data_ptr(), n, o.data_ptr()))
        kroots.append(o)
    kroots = torch.cat(kroots)                                    # slot (k, i) -> K_k * omega^i
    g = torch.Generator(device="cuda").manual_seed(11)
    var_of = torch.randint(0, n, (4 * n,), device="cuda", generator=g)
    vals = torch.randint(0, 1 << 61, (n, 4), dtype=torch.int64, device="cuda", generator=g)    # < 2^253: valid residues
    order = torch.argsort(var_of, stable=True)
    grp = var_of[order]
    first = torch.ones(4 * n, dtype=torch.bool, device="cuda")
    first[1:] = grp[1:] != grp[:-1]
    idx = torch.arange(4 * n, device="cuda")
    start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values      # index of the group's first slot
    is_last = torch.ones(4 * n, dtype=torch.bool, device="cuda")
    is_last[:-1] = grp[1:] != grp[:-1]
    nxt_sorted = torch.where(is_last, order[start], torch.roll(order, -1))
    sigma_pos = torch.empty(4 * n, dtype=torch.int64, device="cuda")
    sigma_pos[order] = nxt_sorted
    wires = [vals[var_of[k * n:(k + 1) * n]].contiguous() for k in range(4)]
    sigmas = [kroots[sigma_pos[k * n:(k + 1) * n]].contiguous() for k in range(4)]
    beta, gamma = bo.seeded_scalars(cv, 0xA00, 2)
    z, last = permutation.permutation_evals(dom, wires, sigmas, mont(cid, [beta])[0], mont(cid, [gamma])[0], return_last=True)
    zh = host(z)
    assert np.array_equal(zh[0], one) and np.array_equal(last, one)
    assert len(np.unique(zh[:4096], axis=0)) > 4000               # not the trivial all-ones vector
    w = cv.root_of_unity(log_n)
    rnd = random.Random(5)
    wh = [host(x) for x in wires]
    sh = [host(x) for x in sigmas]
    for i in [0, 1, n - 2] + [rnd.randrange(n - 1) for _ in range(29)]:
        zi, zn = zk.curves.fr_from_mont(cid, zh[i:i + 2])
        num = den = 1
        for k in range(4):
            wk = zk.curves.fr_from_mont(cid, wh[k][i:i + 1])[0]
            sk = zk.curves.fr_from_mont(cid, sh[k][i:i + 1])[0]
            num = num * (wk + beta * bo.PERM_K[k] * pow(w, i, cv.r) + gamma) % cv.r
            den = den * (wk + beta * sk + gamma) % cv.r
        assert zn * den % cv.r == zi * num % cv.r, i
