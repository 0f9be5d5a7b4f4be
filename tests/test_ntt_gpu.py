"""GPU parity: the HIP NTT (through the C ABI) against the golden vectors, the CPU oracle on seeded
inputs, and size-independent properties at the benchmark sizes.  Bit-exact (integer arithmetic)."""
import numpy as np
import pytest

import ark_plonk_amd as zk
from oracle import bigint_oracle as bo

pytestmark = pytest.mark.gpu

KINDS = ("fft", "ifft", "coset_fft", "coset_ifft")


def run_kind(dom, kind, x):
    return getattr(dom, KINDS[kind])(x)


def test_published_roots_of_unity(ctx, oracle_cpu):
    """The device transform (and the C++ restatement) against roots of unity published elsewhere: fft of X over the 2^k-point domain
    is (1, w, w^2, ...), so output 1 is the domain's generator -- c-kzg-4844's SCALE2_ROOT_OF_UNITY[k] on BLS12-381 (k = 2, 3, 4),
    circom's 2^28-th root raised to 2^(28-k) on BN254."""
    from published_points import EXT_BLS_FR_ROOTS, EXT_BN254_FR_ROOT_28
    for cid in (0, 1):
        cv = bo.CURVES[cid]
        for k in (2, 3, 4):
            w = EXT_BLS_FR_ROOTS[k] if cid == 0 else pow(EXT_BN254_FR_ROOT_28, 1 << (28 - k), cv.r)
            x = np.zeros((1 << k, 4), dtype=np.uint64)
            x[1] = bo.int_to_limbs(bo.to_mont(1, cv.r, 1 << 256), 4)
            want = [bo.int_to_limbs(bo.to_mont(pow(w, i, cv.r), cv.r, 1 << 256), 4) for i in range(1 << k)]
            got = zk.Radix2EvaluationDomain.new(1 << k, cid, ctx).fft(x)
            assert np.array_equal(got, np.array(want, dtype=np.uint64)), (cid, k)
            assert np.array_equal(oracle_cpu.ntt(cid, 0, k, x), np.array(want, dtype=np.uint64)), (cid, k)


def test_reference_to_polynomial_fixture(ctx):
    """lookup/multiset.rs:290-309 `test_to_polynomial`, the reference's one deterministic test through `ifft`: seven evaluations 1..7 on the
    eight-point domain (zero-extended by the call: in_len = 7) interpolate to degree 7 -- through the device's host-pointer and
    device-pointer entry points, all eight coefficients equal to the big-integer restatement's."""
    import torch
    for cid in (0, 1):
        cv = bo.CURVES[cid]
        want = bo.ntt(cv, bo.KIND_IFFT, 3, [1, 2, 3, 4, 5, 6, 7])
        x = np.array([bo.int_to_limbs(bo.to_mont(v, cv.r, 1 << 256), 4) for v in range(1, 8)], dtype=np.uint64)
        dom = zk.Radix2EvaluationDomain.new(7 + 1, cid, ctx)
        assert dom.size() == 8
        got = dom.ifft(x)
        got_dev = dom.ifft(torch.from_numpy(x.view(np.int64)).cuda()).cpu().numpy().view(np.uint64)
        for g in (got, got_dev):
            vals = [bo.from_mont(bo.limbs_to_int(r), cv.r, 1 << 256) for r in g]
            assert vals == want and vals[7] != 0


@pytest.mark.parametrize("cid", [0, 1])
def test_golden_vectors(cid, golden, ctx):
    g = golden[cid]
    keys = [k[:-3] for k in g.files if k.startswith("ntt_") and k.endswith("_in")]
    assert len(keys) == 120
    for key in keys:
        _, log_n, _, kind = key.split("_")
        dom = zk.Radix2EvaluationDomain.new(1 << int(log_n), cid, ctx)
        got = run_kind(dom, int(kind), g[key + "_in"])
        assert np.array_equal(got, g[key + "_out"]), key


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("log_n", [3, 5, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17])
def test_vs_cpu_oracle_all_kinds(cid, log_n, ctx, oracle_cpu):
    n = 1 << log_n
    rng = np.random.default_rng(1000 * cid + log_n)
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    for kind in range(4):
        for in_len in (n, n // 4 + 3 if n >= 8 else n):
            vals = bo.seeded_scalars(bo.CURVES[cid], 7 * log_n + kind, min(in_len, 64))
            # tile a short seeded block with a random permutation of limbs-independent repeats
            base = oracle_cpu.convert(cid, "fr", True, oracle_cpu.ints_to_limbs(vals, 4))
            reps = -(-in_len // base.shape[0])
            x = np.tile(base, (reps, 1))[:in_len].copy()
            # decorrelate: multiply element i by a per-position factor through the oracle
            fac = np.tile(base[::-1], (reps, 1))[:in_len]
            fac = np.roll(fac, int(rng.integers(1, 50)), axis=0)
            x = oracle_cpu.fr_op(cid, "mul", x, fac)
            x = oracle_cpu.fr_op(cid, "add", x, np.roll(x, 1, axis=0))
            exp = oracle_cpu.ntt(cid, kind, log_n, x)
            got = run_kind(dom, kind, x)
            assert np.array_equal(got, exp), (log_n, kind, in_len)


@pytest.mark.parametrize("cid", [0, 1])
def test_edge_cases(cid, ctx, oracle_cpu):
    # empty input -> all zeros; single element; in-place; domain too large
    dom = zk.Radix2EvaluationDomain.new(64, cid, ctx)
    z = dom.fft(np.zeros((0, 4), dtype=np.uint64))
    assert z.shape == (64, 4) and not z.any()
    one = oracle_cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    ev = dom.fft(one)
    assert np.array_equal(ev, np.repeat(one, 64, axis=0))  # constant polynomial
    buf = oracle_cpu.convert(cid, "fr", True, oracle_cpu.ints_to_limbs(bo.seeded_scalars(bo.CURVES[cid], 5, 64), 4))
    exp = oracle_cpu.ntt(cid, 0, 6, buf)
    inplace = buf.copy()
    dom.fft_in_place(inplace)
    assert np.array_equal(inplace, exp)
    with pytest.raises(ValueError):
        dom.fft(np.zeros((65, 4), dtype=np.uint64))
    assert zk.Radix2EvaluationDomain.new((1 << bo.CURVES[cid].two_adicity) + 1, cid, ctx) is None


@pytest.mark.parametrize("cid", [0, 1])
def test_batch_entry_points(cid, ctx, oracle_cpu):
    """zk_ntt_batch (host) and zk_ntt_batch_dev: ragged inputs, one plan; each output equals the single call."""
    import torch
    cv = bo.CURVES[cid]
    dom = zk.Radix2EvaluationDomain.new(1 << 11, cid, ctx)
    polys = [oracle_cpu.convert(cid, "fr", True, oracle_cpu.ints_to_limbs(bo.seeded_scalars(cv, 40 + k, ln), 4))
             for k, ln in enumerate((2048, 512, 1, 0, 2047))]
    for kind in (0, 1, 2, 3):
        exp = [oracle_cpu.ntt(cid, kind, 11, p) for p in polys]
        got = dom.batch(kind, polys)
        dev = dom.batch(kind, [torch.from_numpy(p.view(np.int64)).cuda() for p in polys])
        for e, g, d in zip(exp, got, dev):
            assert np.array_equal(g, e)
            assert np.array_equal(d.cpu().numpy().view(np.uint64), e)


@pytest.mark.parametrize("log_n", [20, 22])
def test_large_device_properties(log_n, ctx, oracle_cpu):
    """BASELINE sizes, device-resident: the whole coset_fft output (n/4 coefficients on the n domain, the prover's shape) and one
    whole ifft against the CPU restatement at 2^20 AND 2^22 (the 4n domain of the 2^20 benchmark), plus size-independent
    properties (round trips, coset consistency, Horner spot checks)."""
    import torch
    cid = 0
    cv = bo.BLS12_381
    n = 1 << log_n
    quarter = n // 4
    rng = np.random.default_rng(log_n)
    # uniformly random canonical values < 2^254 (< r), interpreted directly as Montgomery residues
    host = rng.integers(0, 1 << 63, size=(quarter, 4), dtype=np.uint64)
    host[:, 3] &= np.uint64((1 << 62) - 1)
    x = torch.from_numpy(host.view(np.int64)).cuda()
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    ev = dom.coset_fft(x)                       # zero-extended n/4 -> n, like quotient_poly.rs:72
    back = dom.coset_ifft(ev)
    torch.cuda.synchronize()
    b = back.cpu().numpy().view(np.uint64)
    assert np.array_equal(b[:quarter], host) and not b[quarter:].any()
    ev2 = dom.fft(x)
    back2 = dom.ifft(ev2)
    b2 = back2.cpu().numpy().view(np.uint64)
    assert np.array_equal(b2[:quarter], host) and not b2[quarter:].any()
    # in-place on the device buffer
    buf = ev2.clone()
    dom.ifft_in_place(buf)
    assert torch.equal(buf, back2)
    # Horner spot checks of e[i] = p(g w^i) at a few i (big-int, definitional)
    coeffs = zk.curves.fr_from_mont(cid, host[:4096])  # low-degree slice check: transform of a short poly
    short = torch.from_numpy(host[:4096].view(np.int64)).cuda()
    evs = dom.coset_fft(short).cpu().numpy().view(np.uint64)
    w = cv.root_of_unity(log_n)
    for i in (0, 1, 12345, n - 1):
        pt = cv.fr_generator * pow(w, i, cv.r) % cv.r
        assert zk.curves.fr_from_mont(cid, evs[i:i + 1])[0] == bo.horner(coeffs, pt, cv.r)
    # full vectors against the CPU restatement (~1 s each at 2^22 on the test box's cores)
    exp = oracle_cpu.ntt(cid, 2, log_n, host)
    assert np.array_equal(ev.cpu().numpy().view(np.uint64), exp)
    exp2 = oracle_cpu.ntt(cid, 1, log_n, exp)
    got2 = dom.ifft(ev).cpu().numpy().view(np.uint64)
    assert np.array_equal(got2, exp2)


def test_linearity_2_24(ctx):
    """NTT(a + b) = NTT(a) + NTT(b) at the largest single-GPU size (2^24, 512 MiB per vector)."""
    import torch
    cid = 0
    log_n = 24
    n = 1 << log_n
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    g = torch.Generator(device="cuda").manual_seed(24)
    a = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    fa = dom.fft(a)
    back = dom.ifft(fa)
    assert torch.equal(back, a)
    del back
    # coset round trip in place
    dom.coset_fft_in_place(fa)
    dom.coset_ifft_in_place(fa)
    fb = dom.fft(a)
    assert torch.equal(fa, fb)


@pytest.mark.parametrize("cid,log_n", [(0, 9), (0, 16), (1, 18), (0, 20)])
def test_batched_launch_equals_single_transforms(cid, log_n, ctx):
    """zk_ntt_batch_dev runs every pass of the whole batch as one launch (blockIdx.y = polynomial): 1-, 2- and 3-pass sizes,
    ragged inputs (the coset ffts of the prover have n coefficients on the 4n domain: the first-stage shortcuts apply to
    the batch only if every input fits a quarter), in place, more than 16 polynomials (two launches)."""
    import torch
    n = 1 << log_n
    dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
    g = torch.Generator(device="cuda").manual_seed(100 + log_n)
    shift = 2 if cid == 1 else 0

    def rnd(rows):
        t = torch.randint(0, 1 << 62, (rows, 4), dtype=torch.int64, device="cuda", generator=g)
        t[:, 3] >>= shift
        return t

    for kind in range(4):
        lens = [n // 4, n // 4, n // 4 - 1, 1, n // 4] if kind == 2 else [n, n - 1, n // 4, 0, n, 7, n]
        if log_n == 9:
            lens = lens * 3                      # 15 / 21 polynomials: past the 16 one launch takes
        polys = [rnd(ln) for ln in lens]
        exp = [dom._run(kind, p) for p in polys]
        got = dom.batch(kind, polys)
        for e, o in zip(exp, got):
            assert torch.equal(e, o), (kind, log_n)
        if kind == 2:                              # one input longer than a quarter switches the shortcut off for the batch
            polys2 = polys[:2] + [rnd(n // 4 + 1)]
            got2 = dom.batch(kind, polys2)
            assert torch.equal(got2[0], exp[0]) and torch.equal(got2[2], dom._run(kind, polys2[2]))
        full = [rnd(n) for _ in range(3)]
        exp_f = [dom._run(kind, p) for p in full]
        dom.batch(kind, full, outs=full)           # in place
        for e, o in zip(exp_f, full):
            assert torch.equal(e, o)
