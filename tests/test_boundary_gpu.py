"""The drop-in boundary as a Rust shim would drive it: host-pointer entry points on pageable memory, the
content-addressed SRS cache behind zk_srs_register (PC::trim runs on every gen_proof, circuit.rs:276), one SRS shared by
several zk_ctx, the commitment cache of SURVEY.md 8f N3 inside the ABI, and the accepted encodings of the point at infinity."""
import threading

import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd.prover_schedule import DropInSchedule, ProofSchedule
from conftest import srs_from_powers, tau_powers
from oracle import bigint_oracle as bo

pytestmark = pytest.mark.gpu


def _polys(n, k, seed, lens=None):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(k):
        ln = n if lens is None else lens[i]
        p = rng.integers(0, 1 << 62, size=(ln, 4), dtype=np.uint64)
        p[:, 3] >>= np.uint64(2)
        out.append(p)
    return out


@pytest.mark.parametrize("cid", [0, 1])
def test_srs_register_is_content_addressed(cid, ctx, oracle_cpu):
    """Two consecutive registrations of the same bases return the cached handle (no upload, the window table included) and
    the outputs do not change; different bases, a different length or different flags miss."""
    n = 1 << 13
    pw_c, _ = tau_powers(oracle_cpu, cid, n)
    srs = srs_from_powers(ctx, cid, pw_c).cpu().numpy().view(np.uint64)
    zk.srs_cache_config(0)
    zk.srs_cache_config(32 << 30)
    s0 = zk.srs_cache_stats()
    ck1 = zk.CommitterKey(srs, cid, ctx).precompute()
    p = _polys(n, 1, 5)[0]
    exp = oracle_cpu.kzg_commit(cid, srs, p)
    a = ck1.commit(p)
    ck2 = zk.CommitterKey(srs.copy(), cid, ctx)            # another host buffer, the same content: PC::trim on the next gen_proof
    s1 = zk.srs_cache_stats()
    assert ck2._h.value == ck1._h.value and s1["hits"] == s0["hits"] + 1 and s1["misses"] == s0["misses"] + 1
    assert ck2.table_windows() == ck1.table_windows() > 0  # the table came with the handle
    b = ck2.commit(p)
    assert a == b and np.array_equal(a.xy(), exp[0]) and not a.infinity
    # not the same SRS: one limb changed / one point fewer / an infinity flag set
    other = srs.copy()
    other[17, 0] ^= np.uint64(1)
    other[17] = srs[18]
    ck3 = zk.CommitterKey(other, cid, ctx)
    ck4 = zk.CommitterKey(srs[:-1], cid, ctx)
    flags = np.zeros(n, dtype=np.uint8)
    flags[3] = 1
    ck5 = zk.CommitterKey(srs, cid, ctx, infinity=flags)
    ck6 = zk.CommitterKey(srs, cid, ctx, infinity=np.zeros(n, dtype=np.uint8))   # all-zero flags == no flags
    assert len({ck1._h.value, ck3._h.value, ck4._h.value, ck5._h.value}) == 4 and ck6._h.value == ck1._h.value
    assert ck5.commit(p) != a
    for ck in (ck1, ck2, ck3, ck4, ck5, ck6):
        ck.close()
    # every handle released: the entries stay resident for the next trim ...
    ck7 = zk.CommitterKey(srs, cid, ctx)
    assert ck7.table_windows() > 0 and ck7.commit(p) == a
    ck7.close()
    # ... until the idle budget says otherwise
    zk.srs_cache_config(0)
    assert zk.srs_cache_stats()["entries"] == 0
    zk.srs_cache_config(32 << 30)


def test_one_srs_shared_by_several_contexts(ctx, oracle_cpu):
    """A zk_srs (and its window table) belongs to the device: four contexts / threads / streams commit over ONE copy."""
    import torch
    cid, n = 0, 1 << 14
    pw_c, _ = tau_powers(oracle_cpu, cid, n)
    bases = srs_from_powers(ctx, cid, pw_c)
    ck = zk.CommitterKey(bases, cid, ctx).precompute()
    polys = _polys(n, 4, 9)
    exp = [ck.commit(torch.from_numpy(p.view(np.int64)).cuda()) for p in polys]
    errs = []

    def worker(k):
        try:
            cx = zk.Context(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                mine = ck.with_ctx(cx)
                d = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
                for _ in range(3):
                    assert mine.commit_batch(d) == exp
                    assert mine.commit(d[k]) == exp[k]
                st.synchronize()
            cx.close()
        except Exception as e:   # surfaced below
            errs.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    # the registering ctx may go away first: the handle does not point into it
    cx2 = zk.Context(0)
    ck2 = zk.CommitterKey(bases, cid, cx2)
    keep = ck2.with_ctx(ctx)
    cx2.close()
    assert keep.commit(torch.from_numpy(polys[0].view(np.int64)).cuda()) == exp[0]
    ck2.close()
    ck.close()


@pytest.mark.parametrize("staging", [True, False])
def test_host_pointer_commit_batch_and_open(staging, ctx, oracle_cpu):
    """zk_kzg_commit_batch / zk_kzg_open on pageable host buffers == the device-resident entry points == the CPU restatement,
    with the pinned staging ring and with plain hipMemcpyAsync."""
    import torch
    cid, n = 0, 1 << 15
    pw_c, _ = tau_powers(oracle_cpu, cid, n)
    srs = srs_from_powers(ctx, cid, pw_c).cpu().numpy().view(np.uint64)
    ctx.set_staging(staging)
    try:
        ck = zk.CommitterKey(srs, cid, ctx)
        lens = [n, n - 1, 8192, n, 3, n - 7, n]
        polys = _polys(n, 7, 15, lens)
        d = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
        exp = [oracle_cpu.kzg_commit(cid, srs, p) for p in polys]
        plain = ck.commit_batch(polys)                     # no table: one at a time, uploads still staged
        ck.precompute()
        ctx.io_stats(reset=True)
        tab = ck.commit_batch(polys)                       # fused table batch, upload k+1 under MSM k
        io = ctx.io_stats()
        assert io["h2d_bytes"] == 32 * sum(lens) and io["d2h_bytes"] == 0
        dev = ck.commit_batch(d)
        for a, b, c, (xy, inf) in zip(plain, tab, dev, exp):
            assert a == b == c and a.infinity == bool(inf) and np.array_equal(a.xy(), xy)
        assert ck.commit(polys[1]) == tab[1]
        z = np.array([0x1234567, 0x89abcdef, 0x13579bdf, 0x0fedcba9], dtype=np.uint64)
        chi = np.array([0x2468ace, 0x7654321, 0x2222222, 0x0111111], dtype=np.uint64)
        assert ck.open(polys, z, chi) == ck.open(d, z, chi)
        # the host NTT entry point through the same ring
        dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
        assert np.array_equal(dom.coset_fft(polys[2]), oracle_cpu.ntt(cid, 2, 15, polys[2]))
        buf = polys[0].copy()
        dom.ifft_in_place(buf)
        assert np.array_equal(buf, oracle_cpu.ntt(cid, 1, 15, polys[0]))
        ck.close()
    finally:
        ctx.set_staging(False)      # the default


@pytest.mark.parametrize("log_n", [10, 13])
def test_drop_in_schedule_equals_resident_schedule(log_n, ctx, oracle_cpu):
    """The per-proof schedule through the host-pointer calls (bench.py's drop_in leg) gives the 29 points of the
    device-resident schedule, with and without the window table."""
    n = 1 << log_n
    cv = zk.get_curve(0)
    pw_c, _ = tau_powers(oracle_cpu, 0, n)
    bases = srs_from_powers(ctx, 0, pw_c)
    ck_dev = zk.CommitterKey(bases, cv, ctx).precompute()
    ref = ProofSchedule(log_n, ctx, ck_dev, cv).run_once()
    ck_host = zk.CommitterKey(bases.cpu().numpy().view(np.uint64), cv, ctx)
    got_plain = DropInSchedule(log_n, ctx, ck_host, cv).run_once()
    ck_host.precompute()
    ctx.io_stats(reset=True)
    got = DropInSchedule(log_n, ctx, ck_host, cv).run_once()
    io = ctx.io_stats()
    assert got == ref and got_plain == ref
    # PCIe volume of one proof: 17 (n + n) + 13 (n + 4n) + (4n + 4n) elements for the transforms; 27 commit inputs + 18 opening inputs
    # (t_4 etc. have n coefficients here), 32 B each
    assert io["d2h_bytes"] == 32 * n * (17 + 13 * 4 + 4) and io["h2d_bytes"] == 32 * n * (17 + 13 + 4 + 27 + 18)
    ck_host.close()
    ck_dev.close()


@pytest.mark.parametrize("log_n", [10, 13])
def test_commitment_cache_in_the_abi(log_n, ctx, oracle_cpu):
    """with zk_ctx_set_commit_cache the unchanged 29-commit schedule computes 20 MSMs on the first proof and 17
    afterwards (prover.rs:569-607 re-commits 12 polynomials; the sigma commitments persist), outputs identical; a mutated
    coefficient misses; switching the cache off empties it."""
    import torch
    n = 1 << log_n
    cv = zk.get_curve(0)
    pw_c, _ = tau_powers(oracle_cpu, 0, n)
    bases = srs_from_powers(ctx, 0, pw_c)
    ck = zk.CommitterKey(bases, cv, ctx).precompute()
    plain = ProofSchedule(log_n, ctx, ck, cv)
    ref0, ref1 = plain.run_once(), plain.run_once()        # two proofs = two different witnesses
    assert ref0 != ref1 and ref0[14:17] == ref1[14:17]     # ... over the same prover key (sigma commitments)
    s = ProofSchedule(log_n, ctx, ck, cv, dedup="abi")
    h0 = ctx.commit_cache_stats()
    first = s.run_once()
    assert s.msms_run == 20 and first == ref0
    second = s.run_once()
    assert s.msms_run == 17 and second == ref1
    h1 = ctx.commit_cache_stats()
    assert h1["hits"] - h0["hits"] == 9 + 12 and h1["misses"] - h0["misses"] == 20 + 17
    assert s.run_once(proof_id=0) == ref0                   # an old proof again: identical whatever is still cached
    # content-addressed, not pointer-addressed: the same buffer with one limb changed misses, a copy of it hits
    p = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda")
    a = ck.commit(p)
    m0 = ctx.commit_cache_stats()
    assert ck.commit(p.clone()) == a and ctx.commit_cache_stats()["hits"] == m0["hits"] + 1
    p[n // 2, 1] ^= 1
    b = ck.commit(p)
    assert b != a and ctx.commit_cache_stats()["misses"] == m0["misses"] + 1
    # a Montgomery coefficient vector and the same bytes taken as canonical scalars are different jobs
    c1, c2 = ck.commit_batch([p, p], canonical=[False, True])
    assert c1 == b and c2 != b
    # host-pointer commits go through the same cache
    hp = p.cpu().numpy().view(np.uint64)
    m1 = ctx.commit_cache_stats()
    assert ck.commit_batch([hp])[0] == b and ctx.commit_cache_stats()["hits"] == m1["hits"] + 1
    ctx.set_commit_cache(False)
    assert ctx.commit_cache_stats()["entries"] == 0
    assert ck.commit(p) == b
    ck.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_infinity_encodings_of_bases(cid, golden, ctx):
    """arkworks' GroupAffine::zero() is (0, 1) + flag, and that is what this library emits for an infinite result.
    A base given as (0, 1) WITHOUT a flag (device path, C++ CommitterKey, an output fed back) must count as infinity, like
    (0, 0) and like a flagged point -- on the host path, the device path and with device-side flags."""
    import torch
    g = golden[cid]
    cv = bo.CURVES[cid]
    L = cv.fq_limbs
    bases = g["msm_case_infbase_bases"].copy()
    flags = g["msm_case_infbase_inf"]
    sc = g["msm_case_infbase_scalars"]
    exp_xy, exp_inf = g["msm_case_infbase_out"], g["msm_case_infbase_outinf"][0]
    assert flags.any()
    one = zk.curves.fq_to_mont(cid, [1])[0]
    as_zero_one = bases.copy()
    as_zero_one[flags != 0, :L] = 0
    as_zero_one[flags != 0, L:] = one
    as_zero_zero = bases.copy()
    as_zero_zero[flags != 0] = 0
    d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    for enc in (as_zero_one, as_zero_zero):
        got = zk.VariableBaseMSM.multi_scalar_mul(enc, sc, cid, ctx=ctx)                                  # host path, no flags
        assert got.infinity == bool(exp_inf) and np.array_equal(got.xy(), exp_xy)
        got = zk.VariableBaseMSM.multi_scalar_mul(torch.from_numpy(enc.view(np.int64)).cuda(), d_sc, cid, ctx=ctx)   # device path
        assert got.infinity == bool(exp_inf) and np.array_equal(got.xy(), exp_xy)
    # device-side flags over finite-looking coordinates (zk_srs_register_dev's d_inf_flags)
    ck = zk.CommitterKey(torch.from_numpy(bases.view(np.int64)).cuda(), cid, ctx, infinity=torch.from_numpy(flags.astype(np.uint8)).cuda())
    got = ck.msm(d_sc)
    ck.close()
    assert got.infinity == bool(exp_inf) and np.array_equal(got.xy(), exp_xy)
    # an infinite OUTPUT fed back as a base: P - P = O, then MSM([O, G], [5, 1]) = G
    cancel = zk.VariableBaseMSM.multi_scalar_mul(g["msm_case_cancel_bases"], g["msm_case_cancel_scalars"], cid, infinity=g["msm_case_cancel_inf"], ctx=ctx)
    if cancel.infinity:
        gen = np.concatenate(zk.curves.fq_to_mont(cid, [cv.gx, cv.gy]))
        two = np.stack([cancel.xy(), gen])
        got = zk.VariableBaseMSM.multi_scalar_mul(two, np.array([[5, 0, 0, 0], [1, 0, 0, 0]], dtype=np.uint64), cid, ctx=ctx)
        assert not got.infinity and np.array_equal(got.xy(), gen)


def test_dev_helpers_alloc_upload_copy_download(ctx):
    """The device-memory helpers a non-torch host binds (Rust `DevicePoly`, host/ark_plonk_amd.hpp): alloc / upload / copy (device to
    device, queued on the ctx stream: `split_tx_poly`'s four quarters, rust-shim device.rs `slice_dev`) / download / free; a commit
    of a copied quarter equals the commit of the same range of the source."""
    import ctypes
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd import _lib
    L = _lib.lib()
    cv = zk.get_curve(0)
    n = 1 << 13
    rng = np.random.default_rng(5)
    host = rng.integers(0, 1 << 62, size=(4 * n, 4), dtype=np.uint64)
    src, dst = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(L.zk_dev_alloc(ctx.handle, 4 * n * 32, ctypes.byref(src)))
    _lib.check(L.zk_dev_alloc(ctx.handle, n * 32, ctypes.byref(dst)))
    _lib.check(L.zk_dev_upload(ctx.handle, src, host.ctypes.data_as(ctypes.c_void_p), 4 * n * 32))
    back = np.zeros((n, 4), dtype=np.uint64)
    for k in range(4):
        _lib.check(L.zk_dev_copy(ctx.handle, dst, ctypes.c_void_p(src.value + k * n * 32), n * 32))
        _lib.check(L.zk_dev_download(ctx.handle, back.ctypes.data_as(ctypes.c_void_p), dst, n * 32))       # waits for the stream
        assert np.array_equal(back, host[k * n:(k + 1) * n]), k
    assert L.zk_dev_copy(ctx.handle, dst, src, 0) == 0                       # nothing to do
    assert L.zk_dev_copy(ctx.handle, None, src, 32) == _lib.ZK_ERR_BAD_ARG
    assert L.zk_dev_copy(None, dst, src, 32) == _lib.ZK_ERR_BAD_ARG
    _lib.check(L.zk_dev_free(ctx.handle, src))
    _lib.check(L.zk_dev_free(ctx.handle, dst))


@pytest.mark.parametrize("log_n", [10, 13])
def test_residency_cache_of_the_host_pointer_calls(log_n, ctx, oracle_cpu):
    """zk_ctx_set_residency_cache (round 5): the unchanged caller's schedule with the cache on gives the same 29 points while the
    vectors the library itself produced (ifft outputs -> commit / coset_fft / open inputs) and the ones it has seen (the prover key's
    sigma polynomials) are not uploaded again; a host buffer REWRITTEN after it was produced is a miss and gives the right result
    (a hit is taken on the caller's current bytes, never on the pointer); a cache too small to hold a proof still gives the same
    points; switching the cache off drops every entry."""
    n = 1 << log_n
    cv = zk.get_curve(0)
    pw_c, _ = tau_powers(oracle_cpu, 0, n)
    bases = srs_from_powers(ctx, 0, pw_c)
    ck = zk.CommitterKey(bases.cpu().numpy().view(np.uint64), cv, ctx).precompute()
    sched = DropInSchedule(log_n, ctx, ck, cv)
    ref = sched.run_once(proof_id=0)
    ctx.io_stats(reset=True)
    sched.run_once(proof_id=1)
    plain_h2d = ctx.io_stats()["h2d_bytes"]
    try:
        ctx.set_residency_cache(True, 0, 32 * n)              # n-sized vectors are cached, the 4n-sized coset evaluations are not
        s0 = ctx.residency_cache_stats()
        assert sched.run_once(proof_id=0) == ref
        ctx.io_stats(reset=True)
        assert sched.run_once(proof_id=0) == ref              # second proof of the same witness: sigma polys etc. resident from the first
        io = ctx.io_stats()
        s1 = ctx.residency_cache_stats()
        assert s1["hits"] - s0["hits"] >= 2 * 58 - 8 and s1["entries"] > 0 and s1["bytes"] <= 2 << 30
        # what still goes up: the 13 evaluation vectors (new every proof: here the same, so they hit too), the 4n quotient evaluations,
        # the four quarters of t -- far less than the uncached schedule's 17 + 13 + 4 + 27 + 18 vectors
        assert io["h2d_bytes"] <= 32 * n * (4 + 4 + 13) and io["h2d_bytes"] < plain_h2d // 3
        assert io["d2h_bytes"] == 32 * n * (17 + 13 * 4 + 4)          # every result still comes down
        # soundness: a transform's output rewritten by the caller before it is committed -- the pointer is the same, the bytes are not
        d = sched.dom_n
        ev = np.ascontiguousarray(sched.evals[0])
        coeffs = np.zeros((n, 4), dtype=np.uint64)
        d._run(1, ev, out=coeffs)
        want = ck.commit_batch([coeffs])[0]
        h_before = ctx.residency_cache_stats()["hits"]
        assert ck.commit_batch([coeffs])[0] == want and ctx.residency_cache_stats()["hits"] == h_before + 1      # resident: a hit
        coeffs[5, 0] ^= np.uint64(1)
        changed = ck.commit_batch([coeffs])[0]
        assert changed != want                                                  # the rewritten vector was committed, not the resident copy
        exp_xy, exp_inf = oracle_cpu.kzg_commit(0, bases.cpu().numpy().view(np.uint64), coeffs)
        assert np.array_equal(changed.xy(), exp_xy) and changed.infinity == bool(exp_inf)
        # in-place transform (ark's fft_in_place: in == out): the input is digested before it is overwritten
        buf = np.ascontiguousarray(sched.evals[1]).copy()
        exp = np.zeros((n, 4), dtype=np.uint64)
        d._run(1, np.ascontiguousarray(sched.evals[1]), out=exp)
        d._run(1, buf, out=buf)
        assert np.array_equal(buf, exp)
        # a cache that cannot hold a proof (room for three vectors): same points, entries bounded
        ctx.set_residency_cache(True, 3 * 32 * n, 32 * n)
        ctx.set_residency_cache(False)
        ctx.set_residency_cache(True, 3 * 32 * n, 32 * n)
        assert sched.run_once(proof_id=0) == ref
        st = ctx.residency_cache_stats()
        assert st["entries"] <= 3 and st["bytes"] <= 3 * 32 * n
    finally:
        ctx.set_residency_cache(False)
        ctx.set_residency_cache(True, 2 << 30, 64 << 20)      # the defaults back ...
        ctx.set_residency_cache(False)                        # ... and off
    assert ctx.residency_cache_stats()["entries"] == 0
    assert sched.run_once(proof_id=0) == ref
    ck.close()
