"""GPU parity: the HIP Pippenger MSM (through the C ABI) against the golden vectors, the CPU oracle
and the KZG identity MSM(s, tau^i G) = (sum s_i tau^i) G at the benchmark sizes.  Bit-exact."""
import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd import _lib
from oracle import bigint_oracle as bo

pytestmark = pytest.mark.gpu

CASES = ("repeat", "cancel", "onebucket", "infbase", "zeros", "ones", "mixed", "maxscalar")


def assert_point(got, exp_xy, exp_inf, cid, tag=""):
    L = bo.CURVES[cid].fq_limbs
    assert got.infinity == bool(exp_inf), tag
    assert np.array_equal(got.x, exp_xy[:L]) and np.array_equal(got.y, exp_xy[L:]), tag


@pytest.mark.parametrize("window", [0, 2, 16])
def test_published_points(window, ctx):
    """The device against points published elsewhere (not computed by this repository's oracle): EIP-2537's G1 + G1 on BLS12-381,
    EIP-196's [2](1, 2) and [3](1, 2) on alt_bn128 -- as 1*G + 1*G (the doubling branch), 2*G, and 1*G + 2*G / 3*G."""
    from published_points import EXT_BLS_2G, EXT_BN254_2G, EXT_BN254_3G, EXT_BN254_9G, EXT_BN254_MUL

    def limbs(cv, pt):
        R = 1 << (64 * cv.fq_limbs)
        return np.array(bo.int_to_limbs(bo.to_mont(pt[0], cv.q, R), cv.fq_limbs) + bo.int_to_limbs(bo.to_mont(pt[1], cv.q, R), cv.fq_limbs),
                        dtype=np.uint64)

    def scal(vals):
        out = np.zeros((len(vals), 4), dtype=np.uint64)
        out[:, 0] = vals
        return out

    ctx.set_msm_window(window)
    try:
        for cid, want2, want3 in ((0, EXT_BLS_2G, None), (1, EXT_BN254_2G, EXT_BN254_3G)):
            cv = bo.CURVES[cid]
            g = limbs(cv, (cv.gx, cv.gy))
            for bases, ks, want in ((np.stack([g, g]), [1, 1], want2), (g.reshape(1, -1), [2], want2),
                                    (np.stack([g, g]), [1, 2], want3), (g.reshape(1, -1), [3], want3)):
                if want is None:
                    continue
                got = zk.VariableBaseMSM.multi_scalar_mul(bases, scal(ks), cid, ctx=ctx)
                assert_point(got, limbs(cv, want), 0, cid, f"curve {cid} scalars {ks} c={window}")
        bn = bo.CURVES[1]
        g = limbs(bn, (bn.gx, bn.gy))
        assert_point(zk.VariableBaseMSM.multi_scalar_mul(g.reshape(1, -1), scal([9]), 1, ctx=ctx), limbs(bn, EXT_BN254_9G), 0, 1, "9 G")
        assert_point(zk.VariableBaseMSM.multi_scalar_mul(np.stack([g] * 9), scal([1] * 9), 1, ctx=ctx), limbs(bn, EXT_BN254_9G), 0, 1, "G x 9")
        pt = limbs(bn, EXT_BN254_MUL["point"]).reshape(1, -1)
        assert_point(zk.VariableBaseMSM.multi_scalar_mul(pt, scal([EXT_BN254_MUL["scalar"]]), 1, ctx=ctx), limbs(bn, EXT_BN254_MUL["result"]), 0, 1, "chfast1")
    finally:
        ctx.set_msm_window(0)


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("window", [0, 3, 5, 8, 13])
def test_golden_srs(cid, window, golden, ctx):
    g = golden[cid]
    ctx.set_msm_window(window)
    try:
        for n in (1, 2, 31, 32, 33, 1024):
            got = zk.VariableBaseMSM.multi_scalar_mul(g["srs_1024"][:n], g[f"msm_srs_{n}_scalars"], cid, ctx=ctx)
            assert_point(got, g[f"msm_srs_{n}_out"], g[f"msm_srs_{n}_inf"][0], cid, f"n={n} c={window}")
    finally:
        ctx.set_msm_window(0)


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("window", [0, 2, 4, 16])
def test_golden_edge_cases(cid, window, golden, ctx):
    g = golden[cid]
    ctx.set_msm_window(window)
    try:
        for name in CASES:
            got = zk.VariableBaseMSM.multi_scalar_mul(g[f"msm_case_{name}_bases"], g[f"msm_case_{name}_scalars"], cid,
                                                      infinity=g[f"msm_case_{name}_inf"], ctx=ctx)
            assert_point(got, g[f"msm_case_{name}_out"], g[f"msm_case_{name}_outinf"][0], cid, f"{name} c={window}")
    finally:
        ctx.set_msm_window(0)


@pytest.mark.parametrize("cid", [0, 1])
def test_truncates_to_shorter_and_empty(cid, golden, ctx):
    g = golden[cid]
    # VariableBaseMSM truncates to min(len(bases), len(scalars))
    got = zk.VariableBaseMSM.multi_scalar_mul(g["srs_1024"][:40], g["msm_srs_33_scalars"], cid, ctx=ctx)
    assert_point(got, g["msm_srs_33_out"], 0, cid)
    got = zk.VariableBaseMSM.multi_scalar_mul(g["srs_1024"][:31], g["msm_srs_1024_scalars"][:31], cid, ctx=ctx)
    exp = zk.VariableBaseMSM.multi_scalar_mul(g["srs_1024"][:31], g["msm_srs_1024_scalars"], cid, ctx=ctx)
    assert got == exp
    empty = zk.VariableBaseMSM.multi_scalar_mul(np.zeros((0, 2 * bo.CURVES[cid].fq_limbs), dtype=np.uint64),
                                                np.zeros((0, 4), dtype=np.uint64), cid, ctx=ctx)
    assert empty.infinity


@pytest.mark.parametrize("cid", [0, 1])
def test_fixed_base_srs_generator_matches_oracle(cid, golden, ctx):
    """The synthetic-SRS utility (scalars[i] * G on device) reproduces the oracle's tau^i G."""
    import torch
    g = golden[cid]
    cv = bo.CURVES[cid]
    tau = int(g["srs_tau"][0][0])
    n = 256
    sc = zk.curves.ints_to_limbs([pow(tau, i, cv.r) for i in range(n)], 4)
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    out = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d_sc.data_ptr(), n, out.data_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), g["srs_1024"][:n])


@pytest.mark.parametrize("cid,log_n", [(0, 12), (0, 16), (1, 14)])
def test_vs_cpu_oracle_medium(cid, log_n, ctx, oracle_cpu):
    import torch
    cv = bo.CURVES[cid]
    n = 1 << log_n
    rng = np.random.default_rng(77 + log_n)
    # bases: k_i * G for random 64-bit k_i, generated on the GPU by the fixed-base utility
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = rng.integers(1, 1 << 62, size=n, dtype=np.uint64)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d_k.data_ptr(), n, bases.data_ptr()))
    scal = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    scal[:, 3] &= np.uint64((1 << 60) - 1)
    scal[5] = 0
    scal[6] = (1, 0, 0, 0)
    scal[7] = scal[8]
    d_s = torch.from_numpy(scal.view(np.int64)).cuda()
    got = zk.VariableBaseMSM.multi_scalar_mul(bases, d_s, cid, ctx=ctx)
    exp_xy, exp_inf = oracle_cpu.msm_g1(cid, bases.cpu().numpy().view(np.uint64), scal)
    assert_point(got, exp_xy, exp_inf, cid)


def _kzg_identity(cid, log_n, ctx, skew=False, precompute=False, offset=0):
    """MSM(s, tau^i G) == (sum_i s_i tau^i mod r) G : O(N) big-int work, valid at any size."""
    import torch
    cv = bo.CURVES[cid]
    n = 1 << log_n
    tau = 0x7A5C0DE
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * tau % cv.r
    sc_tau = zk.curves.ints_to_limbs(pw, 4)
    d_tau = torch.from_numpy(sc_tau.view(np.int64)).cuda()
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d_tau.data_ptr(), n, bases.data_ptr()))
    rng = np.random.default_rng(log_n)
    scal = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    scal[:, 3] &= np.uint64((1 << 61) - 1)
    if skew:
        # the benchmark circuit's wire values: a few small constants repeated (composer.rs:493-548)
        small = zk.curves.ints_to_limbs([6, 7, cv.r - 20, 1], 4)
        scal[: 3 * n // 4] = np.tile(small, (3 * n // 16, 1))
    s_int = zk.curves.limbs_to_ints(scal)
    acc = 0
    for s, p in zip(s_int, pw):
        acc += s * p
    acc %= cv.r
    exp = bo.ec_mul(cv, acc, (cv.gx, cv.gy))
    ck = zk.CommitterKey(bases, cid, ctx)
    if precompute:
        ck.precompute()
    if offset:
        # MSM over powers[offset:], as kzg10::commit does after stripping leading zero coefficients
        acc = 0
        for s, p in zip(s_int[: n - offset], pw[offset:]):
            acc += s * p
        acc %= cv.r
        exp = bo.ec_mul(cv, acc, (cv.gx, cv.gy))
        got = ck.msm(torch.from_numpy(scal[: n - offset].view(np.int64)).cuda(), base_offset=offset)
    else:
        got = ck.msm(torch.from_numpy(scal.view(np.int64)).cuda())
    ck.close()
    assert not got.infinity
    assert zk.curves.fq_from_mont(cid, got.x.reshape(1, -1))[0] == exp[0]
    assert zk.curves.fq_from_mont(cid, got.y.reshape(1, -1))[0] == exp[1]


def test_kzg_identity_2_18_bn254(ctx):
    _kzg_identity(1, 18, ctx)


def test_kzg_identity_2_20(ctx):
    _kzg_identity(0, 20, ctx)


def test_kzg_identity_2_20_skewed_scalars(ctx):
    _kzg_identity(0, 20, ctx, skew=True)


def test_precomputed_table_2_20(ctx):
    _kzg_identity(0, 20, ctx, precompute=True)


def test_precomputed_table_2_20_skewed(ctx):
    _kzg_identity(0, 20, ctx, skew=True, precompute=True)


def test_precomputed_table_2_20_vs_cpu_pippenger_limb_for_limb(ctx, oracle_cpu):
    """The benchmark MSM -- 2^20 points, window table -- against the C++ restatement of ark 0.3's Pippenger (c = 15, 17 windows;
    ~1 s on the test box), limb for limb; the KZG identity above checks the same size against a different kind of witness."""
    import torch
    cid, n = 0, 1 << 20
    cv = bo.CURVES[cid]
    g = torch.Generator(device="cuda").manual_seed(2020)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, ks.data_ptr(), n, bases.data_ptr()))
    rng = np.random.default_rng(77)
    scal = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    scal[:, 3] &= np.uint64((1 << 62) - 1)          # < 2^254 < r: canonical
    scal[:1000] = 0
    scal[1000:2000] = np.array([1, 0, 0, 0], dtype=np.uint64)
    ck = zk.CommitterKey(bases, cid, ctx).precompute()
    got = ck.msm(torch.from_numpy(scal.view(np.int64)).cuda())
    ck.close()
    exp_xy, exp_inf = oracle_cpu.msm_g1(cid, bases.cpu().numpy().view(np.uint64), scal)
    assert_point(got, exp_xy, exp_inf, cid)


def test_precomputed_table_threshold_sizes(ctx):
    # smallest sizes that take the shared-bucket path (ZK_PRE_MIN_N = 2^13) and one just above
    _kzg_identity(0, 13, ctx, precompute=True)
    _kzg_identity(0, 14, ctx, precompute=True, offset=8191)   # n - offset = 8193 scalars


def test_round_batch_with_canonical_jobs(ctx, oracle_cpu):
    """zk_kzg_round_batch_dev: Montgomery-coefficient jobs and canonical-scalar jobs in one batch."""
    import torch
    cid, n = 0, 1 << 13
    cv = bo.CURVES[cid]
    g = torch.Generator(device="cuda").manual_seed(7)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, ks.data_ptr(), n, bases.data_ptr()))
    coeffs = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    canon = torch.from_numpy(oracle_cpu.convert(cid, "fr", False, coeffs.cpu().numpy().view(np.uint64)).view(np.int64)).cuda()
    ck = zk.CommitterKey(bases, cid, ctx).precompute()
    a, b, c = ck.commit_batch([coeffs, canon, coeffs], canonical=[False, True, False])
    single = ck.commit(coeffs)
    ck.close()
    assert a == single and b == single and c == single


def test_precomputed_table_offset_and_bn254(ctx):
    _kzg_identity(0, 16, ctx, precompute=True, offset=777)
    _kzg_identity(1, 16, ctx, precompute=True)


def test_precomputed_table_matches_plain_path(ctx, oracle_cpu):
    """Same SRS, same scalars: shared-bucket path == per-window path == CPU oracle (2^14 points)."""
    import torch
    cid, n = 0, 1 << 14
    cv = bo.CURVES[cid]
    rng = np.random.default_rng(5)
    ks = np.zeros((n, 4), dtype=np.uint64)
    ks[:, 0] = rng.integers(1, 1 << 62, size=n, dtype=np.uint64)
    d_k = torch.from_numpy(ks.view(np.int64)).cuda()
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d_k.data_ptr(), n, bases.data_ptr()))
    scal = np.stack([np.frombuffer(int(v).to_bytes(32, "little"), dtype="<u8") for v in
                     bo.seeded_scalars(cv, 4242, n)]).astype(np.uint64)
    scal[3] = 0
    scal[4] = zk.curves.ints_to_limbs([cv.r - 1], 4)[0]
    d_s = torch.from_numpy(scal.view(np.int64)).cuda()
    ck = zk.CommitterKey(bases, cid, ctx)
    plain = ck.msm(d_s)
    ck.precompute()
    pre = ck.msm(d_s)
    ck.close()
    exp_xy, exp_inf = oracle_cpu.msm_g1(cid, bases.cpu().numpy().view(np.uint64), scal)
    assert plain == pre
    assert_point(pre, exp_xy, exp_inf, cid)


@pytest.mark.parametrize("cid", [0, 1])
def test_kzg_commit_golden(cid, golden, ctx):
    g = golden[cid]
    ck = zk.CommitterKey(g["srs_1024"], cid, ctx)
    for k in range(4):
        got = ck.commit(g[f"kzg_poly_{k}"])
        assert_point(got, g[f"kzg_commit_{k}"], 0, cid, f"poly {k}")
    ck.close()


def test_sharded_msm_partials_match_single(golden, ctx):
    """Multi-GPU shape on one card: two point-shards -> Jacobian partials -> zk_g1_sum_partials."""
    import torch
    cid = 0
    g = golden[cid]
    n = 1024
    ck = zk.CommitterKey(g["srs_1024"], cid, ctx)
    sc = torch.from_numpy(g["msm_srs_1024_scalars"].view(np.int64)).cuda()
    parts = [ck.msm_partial(sc[:600], 0), ck.msm_partial(sc[600:], 600)]
    got = zk.sum_partials(np.stack(parts), cid)
    assert_point(got, g["msm_srs_1024_out"], 0, cid)
    ck.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_kzg_open_golden(cid, golden, ctx):
    """PC::open (prover.rs:582-591): RLC of the polynomials, witness by division by (X - z), commit."""
    import torch
    g = golden[cid]
    ck = zk.CommitterKey(g["srs_1024"], cid, ctx)
    polys = [torch.from_numpy(g[f"kzg_poly_{k}"].view(np.int64)).cuda() for k in range(4)]
    got = ck.open(polys, g["kzg_z"], g["kzg_chi"])
    assert_point(got, g["kzg_open"], g["kzg_open_inf"][0], cid)
    # a single constant polynomial has an empty witness -> commitment to zero = infinity
    one = torch.from_numpy(g["kzg_poly_0"][:1].view(np.int64)).cuda()
    assert ck.open([one], g["kzg_z"], g["kzg_chi"]).infinity
    ck.close()


def test_kzg_open_vs_cpu_oracle_2_16(ctx, oracle_cpu):
    """11 polynomials of 2^16 coefficients (the aw opening of prover.rs:582-591), ragged lengths."""
    import torch
    cid, n = 0, 1 << 16
    cv = bo.CURVES[cid]
    rng = np.random.default_rng(16)
    tau = 0x7A5C0DE
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * tau % cv.r
    d_tau = torch.from_numpy(zk.curves.ints_to_limbs(pw, 4).view(np.int64)).cuda()
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d_tau.data_ptr(), n, bases.data_ptr()))
    lens = [n, n, n - 1, n, n - 5, n, n, n // 2, n, n, 3]
    polys = []
    for ln in lens:
        a = rng.integers(0, 1 << 62, size=(ln, 4), dtype=np.uint64)
        polys.append(a)
    z = oracle_cpu.convert(cid, "fr", True, zk.curves.ints_to_limbs(bo.seeded_scalars(cv, 1, 1), 4))[0]
    chi = oracle_cpu.convert(cid, "fr", True, zk.curves.ints_to_limbs(bo.seeded_scalars(cv, 2, 1), 4))[0]
    # oracle: RLC with its Fr ops, witness by synthetic division, commit
    comb = np.zeros((n, 4), dtype=np.uint64)
    chi_pow = oracle_cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    for p in polys:
        term = oracle_cpu.fr_op(cid, "mul", p, np.repeat(chi_pow, p.shape[0], axis=0))
        comb[: p.shape[0]] = oracle_cpu.fr_op(cid, "add", comb[: p.shape[0]], term)
        chi_pow = oracle_cpu.fr_op(cid, "mul", chi_pow, chi.reshape(1, 4))
    w = oracle_cpu.kzg_witness(cid, comb, z)
    b_host = bases.cpu().numpy().view(np.uint64)
    exp_xy, exp_inf = oracle_cpu.kzg_commit(cid, b_host, w)
    ck = zk.CommitterKey(bases, cid, ctx)
    got = ck.open([torch.from_numpy(p.view(np.int64)).cuda() for p in polys], z, chi)
    assert_point(got, exp_xy, exp_inf, cid)
    ck.precompute()
    got2 = ck.open([torch.from_numpy(p.view(np.int64)).cuda() for p in polys], z, chi)
    assert got2 == got
    ck.close()


def test_commit_batch_equals_single_commits(ctx):
    """zk_kzg_commit_batch_dev (pipelined over two buffer sets / two streams) == one commit at a time."""
    import torch
    cid, n = 0, 1 << 14
    cv = bo.CURVES[cid]
    g = torch.Generator(device="cuda").manual_seed(99)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, ks.data_ptr(), n, bases.data_ptr()))
    polys = [torch.randint(0, 1 << 62, (ln, 4), dtype=torch.int64, device="cuda", generator=g)
             for ln in (n, n, n - 1, n, n - 100, n, n)]
    ck = zk.CommitterKey(bases, cid, ctx)
    plain = [ck.commit(p) for p in polys]           # per-window path
    assert ck.commit_batch(polys) == plain           # batch without table falls back to single commits
    ck.precompute()
    single = [ck.commit(p) for p in polys]
    batch = ck.commit_batch(polys)
    again = ck.commit_batch(polys[:3])
    ck.close()
    assert single == plain and batch == plain and again == plain[:3]


def test_round_batch_with_very_different_lengths(ctx):
    """One PC::commit call may hold polynomials of different degree (prover.rs:459-469: t_4 is shorter; a linearisation
    polynomial next to sigma polynomials): lengths from the table-path threshold to the full SRS, odd lengths
    (single-scalar digit kernel) and one below the threshold (whole batch falls back to one-at-a-time)."""
    import torch
    cid, n = 0, 1 << 16
    cv = bo.CURVES[cid]
    g = torch.Generator(device="cuda").manual_seed(123)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, ks.data_ptr(), n, bases.data_ptr()))
    ck = zk.CommitterKey(bases, cid, ctx)
    lens = (n, 8192, n - 1, n // 2 + 3, 8193, 3 * n // 4)
    polys = [torch.randint(0, 1 << 62, (ln, 4), dtype=torch.int64, device="cuda", generator=g) for ln in lens]
    plain = [ck.commit(p) for p in polys]                       # per-window path, one at a time
    ck.precompute()
    assert ck.commit_batch(polys) == plain                      # fused table path, mixed geometry inputs
    mixed = polys + [polys[0][:100]]                            # one job below the threshold
    assert ck.commit_batch(mixed) == plain + [ck.commit(polys[0][:100])]
    ck.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_quad_cooperative_group_law(cid, ctx):
    """csrc/ecq.cuh (a point spread over four lanes, four product rounds per addition) against the single-lane law of
    ecu.cuh on the device: generic pairs, doubling, cancellation, infinities and a fed-back chain."""
    import ctypes
    bad = ctypes.c_uint32(123)
    mask = ctypes.c_uint32(0)
    _lib.check(_lib.lib().zk_selftest_quad_dev(ctx.handle, cid, 6000, ctypes.byref(bad), ctypes.byref(mask)))
    assert bad.value == 0, f"{bad.value} mismatching quads, case mask {mask.value:#x}"


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("table", [False, True])
def test_repeated_and_opposite_points_take_the_general_law(cid, table, ctx, oracle_cpu):
    """Doubling and cancellation inside the bucket accumulation at scale (the golden edge cases have a handful of points): an SRS
    made of 32 distinct points repeated 256 times each, with scalars drawn from a handful of values and their negatives, puts
    equal and opposite points next to each other in most buckets of both MSM paths; then an MSM that cancels to infinity, then
    the same buffers again with generic scalars."""
    import torch
    cv = bo.CURVES[cid]
    n = 1 << 13
    rng = np.random.default_rng(99 + cid)
    ks = np.zeros((32, 4), dtype=np.uint64)
    ks[:, 0] = rng.integers(1, 1 << 40, size=32, dtype=np.uint64)
    d_k = torch.from_numpy(np.tile(ks, (n // 32, 1)).view(np.int64)).cuda()
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, d_k.data_ptr(), n, bases.data_ptr()))
    vals = bo.seeded_scalars(cv, 4321, 6)
    pool = vals + [cv.r - v for v in vals] + [0, 1, cv.r - 1]
    scal = zk.curves.ints_to_limbs([pool[int(i)] for i in rng.integers(0, len(pool), size=n)], 4)
    bases_h = bases.cpu().numpy().view(np.uint64)
    exp_xy, exp_inf = oracle_cpu.msm_g1(cid, bases_h, scal)
    ck = zk.CommitterKey(bases, cid, ctx)
    if table:
        ck.precompute()
    got = ck.msm(torch.from_numpy(scal.view(np.int64)).cuda())
    assert_point(got, exp_xy, exp_inf, cid)
    # everything cancels: sum_i s_i P_i + sum_i (r - s_i) P_i = infinity
    half = n // 2
    s2 = scal.copy()
    s2[half:] = zk.curves.ints_to_limbs([(cv.r - v) % cv.r for v in zk.curves.limbs_to_ints(scal[:half])], 4)
    bases2 = torch.cat([bases[:half], bases[:half]])
    ck2 = zk.CommitterKey(bases2, cid, ctx)
    if table:
        ck2.precompute()
    assert ck2.msm(torch.from_numpy(s2.view(np.int64)).cuda()).infinity
    ck2.close()
    # the same buffers again with generic scalars
    scal3 = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    exp_xy, exp_inf = oracle_cpu.msm_g1(cid, bases_h, scal3)
    assert_point(ck.msm(torch.from_numpy(scal3.view(np.int64)).cuda()), exp_xy, exp_inf, cid)
    ck.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_table_path_irregular_sizes_across_the_lane_rounding_rules(cid, ctx, oracle_cpu):
    """Lengths that are not powers of two, on both sides of every threshold of the accumulation's lane plan (msm_plan.hip: pre_plan_geom) --
    one round of resident lanes (131072) exceeded or not, whole-round rounding taken or refused (chunks below 16), two rounds
    becoming three, the 16-bit -> 17-bit table window at 2^19 points -- as single MSMs and as round batches (in which every job but
    the last takes the long-chunk plan), odd and even, against the C++ restatement's KZG commitment limb for limb."""
    import torch
    cv = bo.CURVES[cid]
    n_max = 786433
    g = torch.Generator(device="cuda").manual_seed(4242 + cid)
    ks = torch.randint(1, 1 << 62, (n_max, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n_max, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, ks.data_ptr(), n_max, bases.data_ptr()))
    bases_h = bases.cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(99 + cid)
    for n_srs in (262145, n_max):                          # a 16-bit table (below 2^19 points) and a 17-bit one
        ck = zk.CommitterKey(bases[:n_srs].contiguous(), cid, ctx).precompute()
        sizes = [s for s in (131071, 131073, 140001, 163839, 163840, 196609, 262143, 262145, 327681, 524287, 524289, 700001, n_max) if s <= n_srs]
        polys, want = [], []
        for n in sizes:
            p = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
            p[:, 3] >>= 3                                  # Montgomery residues below both moduli
            polys.append(torch.from_numpy(p.view(np.int64)).cuda())
            want.append(oracle_cpu.kzg_commit(cid, bases_h[:n_srs], p))
        for p, (exp_xy, exp_inf) in zip(polys, want):      # single MSMs: the default plan
            assert_point(ck.commit(p), exp_xy, exp_inf, cid)
        for lo in range(0, len(polys), 5):                 # round batches: long-chunk plans for all but the last job
            got = ck.commit_batch(polys[lo:lo + 5])
            for pt, (exp_xy, exp_inf) in zip(got, want[lo:lo + 5]):
                assert_point(pt, exp_xy, exp_inf, cid)
        got = ck.commit_batch(polys[::-1][:6])             # ... and in another order (another job is last)
        for pt, (exp_xy, exp_inf) in zip(got, want[::-1][:6]):
            assert_point(pt, exp_xy, exp_inf, cid)
        ck.close()
