"""Deferred commitment rounds (zk_kzg_round_begin_dev / zk_kzg_open_begin_dev ... zk_kzg_round_end): the reference issues
f | h_1 | h_2 (prover.rs:289-317), z | z_2 (prover.rs:361-389) and the four calls of its last round (prover.rs:579-618) as
separate blocking PC calls although no result of one feeds the next; the deferred form queues them and collects once.
Results must equal the blocking calls bit for bit."""
import ctypes

import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd import _lib
from ark_plonk_amd.prover_schedule import ProofSchedule

pytestmark = pytest.mark.gpu


def _ck(ctx, cv, n, seed=5):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cv.curve_id, ks.data_ptr(), n, bases.data_ptr()))
    return zk.CommitterKey(bases, cv, ctx)


def _polys(n, k, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g) for _ in range(k)]


@pytest.mark.parametrize("cid", [0, 1])
def test_deferred_round_equals_blocking_calls(ctx, cid):
    """Commits of different lengths (table path, per-window path, empty), an opening and a canonical-scalar job, begun by five
    calls and collected by one: the points of the blocking entry points, in submission order."""
    cv = zk.get_curve(cid)
    n = 1 << 14
    ck = _ck(ctx, cv, n).precompute()
    p = _polys(n, 6, 11)
    z = np.array([0x1234567, 0x89abcdef, 0x13579bdf, 0x0fedcba9], dtype=np.uint64)
    chi = np.array([0x2468ace, 0x7654321, 0x2222222, 0x0111111], dtype=np.uint64)
    short = p[3][:1000]                    # below the table threshold: computed at begin, in its own buffer set
    odd = p[4][: n - 3]                    # odd length: the separate into_repr pass of the sort
    empty = p[5][:0]
    w = zk.msm.kzg_witness([p[0], p[1]], z, chi, cid, ctx)
    exp = ck.commit_batch([p[0], p[1]]) + [ck.commit(short)] + [ck.commit(odd)] + [ck.commit(empty)]
    exp += [ck.open([p[2], p[0], odd], z, chi)] + ck.commit_batch([w], canonical=[True])
    assert ck.round_pending() == 0
    assert ck.commit_begin([p[0], p[1]]) == 2
    assert ck.commit_begin([short]) == 3
    assert ck.commit_begin([odd, empty]) == 5
    assert ck.open_begin([p[2], p[0], odd], z, chi) == 6
    assert ck.commit_begin([w], canonical=[True]) == 7
    got = ck.round_end()
    assert ck.round_pending() == 0
    assert len(got) == len(exp) == 7
    for k, (a, b) in enumerate(zip(got, exp)):
        assert a == b, k
    # a second round on the same buffers, other inputs: nothing of the first round survives
    q = _polys(n, 3, 12)
    exp2 = ck.commit_batch(q)
    ck.commit_begin(q[:1])
    ck.commit_begin(q[1:])
    assert ck.round_end(3) == exp2
    ck.close()


def test_deferred_round_interleaved_with_transforms_and_sixteen_jobs(ctx):
    """The shape of the prover's last round: 7 commits, an opening, 7 commits, an opening -- sixteen jobs, the most a round holds --
    with NTTs queued between the calls (they share the stream, not the MSM buffers)."""
    cv = zk.get_curve(0)
    log_n = 13
    n = 1 << log_n
    ck = _ck(ctx, cv, n).precompute()
    dom = zk.Radix2EvaluationDomain.new(n, 0, ctx)
    ev = _polys(n, 14, 21)
    z = np.array([5, 6, 7, 8], dtype=np.uint64)
    chi = np.array([9, 10, 11, 12], dtype=np.uint64)
    co = [dom.ifft(e) for e in ev]
    exp = ck.commit_batch(co[:7]) + [ck.open(co[:7] + co[10:14], z, chi)] + ck.commit_batch(co[7:14]) + [ck.open(co[7:14], z, chi)]
    ck.commit_begin([dom.ifft(e) for e in ev[:7]])
    ck.open_begin(co[:7] + co[10:14], z, chi)
    late = [dom.ifft(e) for e in ev[7:14]]          # transforms queued while the round is open
    ck.commit_begin(late)
    assert ck.open_begin(late, z, chi) == 16
    with pytest.raises(_lib.ZkError) as e:           # a seventeenth job does not fit
        ck.commit_begin(co[:1])
    assert e.value.code == _lib.ZK_ERR_UNSUPPORTED
    assert ck.round_end() == exp
    ck.close()


def test_open_round_blocks_the_blocking_entry_points_and_survives_a_wrong_count(ctx):
    cv = zk.get_curve(0)
    n = 1 << 13
    ck = _ck(ctx, cv, n).precompute()
    p = _polys(n, 2, 31)
    exp = ck.commit_batch(p)
    ck.commit_begin(p[:1])
    for call in (lambda: ck.commit(p[1]), lambda: ck.commit_batch(p), lambda: ck.msm(p[1]),
                 lambda: ck.open(p, np.ones(4, dtype=np.uint64), np.ones(4, dtype=np.uint64)),
                 lambda: ctx.set_commit_cache(True)):
        with pytest.raises(_lib.ZkError) as e:
            call()
        assert e.value.code == _lib.ZK_ERR_PENDING
    ck.commit_begin(p[1:])
    with pytest.raises(_lib.ZkError) as e:           # wrong job count: refused, the round stays open
        ck.round_end(3)
    assert e.value.code == _lib.ZK_ERR_BAD_ARG and ck.round_pending() == 2
    assert ck.round_end(2) == exp
    # abort: the round is dropped, the ctx usable again
    ck.commit_begin(p)
    ck.round_abort()
    assert ck.round_pending() == 0 and ck.commit_batch(p) == exp
    # an empty round closes with no outputs
    assert ck.round_end(0) == []
    ck.close()


def test_deferred_round_with_the_commitment_cache_and_partials(ctx):
    """With the ABI's commitment cache on every job is answered at begin (hits cost no MSM); the sharded form returns Jacobian
    partials that sum to the same points."""
    cv = zk.get_curve(0)
    n = 1 << 13
    ck = _ck(ctx, cv, n).precompute()
    p = _polys(n, 3, 41)
    exp = ck.commit_batch(p)
    ctx.set_commit_cache(True)
    try:
        ck.commit_begin(p[:2])
        ck.commit_begin([p[2], p[0]])
        got = ck.round_end()
        assert got == exp + exp[:1]
        st = ctx.commit_cache_stats()
        assert st["hits"] >= 1
    finally:
        ctx.set_commit_cache(False)
    ck.commit_begin(p[:1])
    ck.commit_begin(p[1:])
    parts = ck.round_end_partial(3)
    assert parts.shape == (3, 3 * cv.fq_limbs)
    for k in range(3):
        assert zk.msm.sum_partials(parts[k:k + 1], 0) == exp[k]
    ck.close()


@pytest.mark.parametrize("log_n", [10, 13])
def test_schedule_deferred_equals_blocking(ctx, log_n):
    """The per-proof schedule with its eleven calls collected in five groups (default) against every call blocking."""
    cv = zk.get_curve("bls12_381")
    ck = _ck(ctx, cv, 1 << log_n, seed=3).precompute()
    a = ProofSchedule(log_n, ctx, ck, cv, defer_calls=False)
    b = ProofSchedule(log_n, ctx, ck, cv)
    assert b.defer_calls
    pa, pb = a.run_once(proof_id=0), b.run_once(proof_id=0)
    assert a.msms_run == b.msms_run == 29
    assert pa == pb
    assert b.run_once(proof_id=1) == a.run_once(proof_id=1)
    ck.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_round_reduce_lets_other_work_run_under_the_host_tail(ctx, cid):
    """zk_kzg_round_reduce queues the reductions and closes the round to new jobs; transforms launched before zk_kzg_round_end run
    behind them; the points are those of the plain round, the transforms' results those of the same calls made alone."""
    import torch
    cv = zk.get_curve(cid)
    n = 1 << 14
    ck = _ck(ctx, cv, n).precompute()
    p = _polys(n, 4, 23)
    if cid == 1:
        for t in p:
            t[:, 3] >>= 2                   # BN254: keep the Montgomery residues below r
    dom = zk.GeneralEvaluationDomain.new(n, cv, ctx)
    exp = ck.commit_batch(p[:3])
    exp_fft = dom.fft(p[3]).clone()
    ck.round_reduce()                        # no open round: a no-op
    assert ck.commit_begin(p[:2]) == 2
    assert ck.commit_begin([p[2]]) == 3
    ck.round_reduce()
    ck.round_reduce()                        # twice is once
    with pytest.raises(_lib.ZkError) as e:   # the round takes no further jobs ...
        ck.commit_begin([p[3]])
    assert e.value.code == _lib.ZK_ERR_PENDING
    with pytest.raises(_lib.ZkError) as e:   # ... and the blocking calls stay refused until it is collected
        ck.commit(p[3])
    assert e.value.code == _lib.ZK_ERR_PENDING
    got_fft = dom.fft(p[3])                  # queued behind the reductions, runs while round_end's host part works
    assert ck.round_pending() == 3
    got = ck.round_end()
    assert got == exp
    torch.cuda.synchronize()
    assert torch.equal(got_fft, exp_fft)
    # the next round starts clean
    assert ck.commit_begin([p[3]]) == 1
    assert ck.round_end() == [ck.commit(p[3])]
    # a reduced round refuses a wrong count and stays collectable; an aborted one leaves the ctx usable; a round of jobs that were
    # computed at begin (below the table threshold) has nothing to reduce
    ck.commit_begin(p[:2])
    ck.round_reduce()
    with pytest.raises(_lib.ZkError) as e:
        ck.round_end(3)
    assert e.value.code == _lib.ZK_ERR_BAD_ARG and ck.round_pending() == 2
    assert ck.round_end(2) == exp[:2]
    ck.commit_begin(p[:3])
    ck.round_reduce()
    ck.round_abort()
    assert ck.round_pending() == 0 and ck.commit_batch(p[:3]) == exp
    short = [t[:500] for t in p[:2]]
    ck.commit_begin(short)
    ck.round_reduce()
    assert ck.round_end() == [ck.commit(t) for t in short]


def test_schedule_with_hoisted_transforms_gives_the_same_commitments(ctx):
    cv = zk.get_curve(0)
    log_n = 13
    n = 1 << log_n
    ck = _ck(ctx, cv, n, seed=9).precompute()
    a = ProofSchedule(log_n, ctx, ck, cv, hoist=True).run_once(0)
    b = ProofSchedule(log_n, ctx, ck, cv, hoist=False).run_once(0)
    c = ProofSchedule(log_n, ctx, ck, cv, defer_calls=False).run_once(0)
    assert a == b == c and len(a) == 29


@pytest.mark.parametrize("merge,long_rounds", [(0, None), (1, 1), (1, 3)])
def test_round_launch_shapes_give_the_same_points(ctx, merge, long_rounds):
    """The tuning options the in-process A/B uses (tools/ab_proof.py, zk_ctx_set_option) select how a round's work is launched: one sort +
    one accumulation launch per job at submission (msm_merge = 0, the shape before round 4), or one launch per kernel for the whole round
    with long chunks for all jobs but the last (default) or round 3's chunk length inside the merged launch (long_rounds = 3).  Same points
    from every shape, for jobs of different lengths (2^17 + 2^15 points: several rounds of lanes), through begin / reduce / end."""
    cv = zk.get_curve(0)
    n = (1 << 17) + (1 << 15)
    ck = _ck(ctx, cv, n, seed=77).precompute()
    polys = _polys(n, 5, 78)
    polys[1] = polys[1][: n - 1]
    polys[3] = polys[3][: 1 << 13]
    assert ctx.get_option("msm_merge") == 1 and ctx.get_option("long_rounds") == 1          # the defaults
    want = ck.commit_batch(polys)
    try:
        ctx.set_option("msm_merge", merge)
        if long_rounds is not None:
            ctx.set_option("long_rounds", long_rounds)
        assert ck.commit_batch(polys) == want
        ck.commit_begin(polys[:2])
        ck.commit_begin(polys[2:3])
        with pytest.raises(RuntimeError):
            ctx.set_option("msm_merge", 1 - merge)        # a round is open: the plan of its jobs must not change (ZK_ERR_PENDING)
        ck.commit_begin(polys[3:])
        ck.round_reduce()
        assert ck.round_end(5) == want
    finally:
        ctx.set_option("msm_merge", 1)
        ctx.set_option("long_rounds", 1)
    ck.close()


def test_ctx_options_api(ctx):
    """zk_ctx_set_option / zk_ctx_get_option: every key of the header round-trips, unknown keys and out-of-range values are refused,
    and the reduction-geometry options change launch shapes only (same points)."""
    from ark_plonk_amd import _lib
    L = _lib.lib()
    import ctypes
    import os
    for key in ctx.OPTIONS:
        old = ctx.get_option(key)
        assert old == {"msm_merge": 1, "pre_logg": -1, "long_rounds": 1, "mem_reserve_mb": 1024,
                       "host_workers": min(15, max(0, (os.cpu_count() or 1) - 1))}.get(key, 0), key
    assert L.zk_ctx_set_option(ctx.handle, b"no_such_key", 1) == _lib.ZK_ERR_UNSUPPORTED
    v = ctypes.c_int64()
    assert L.zk_ctx_get_option(ctx.handle, b"no_such_key", ctypes.byref(v)) == _lib.ZK_ERR_UNSUPPORTED
    for key, bad in (("pre_vw", 48), ("pre_vw", 4), ("chunk_l", 4), ("combine_sg", 3), ("pre_max_log_n", 12), ("pre_max_log_n", 26), ("long_rounds", 0),
                     ("msm_merge", 2), ("pre_logg", 6)):
        assert L.zk_ctx_set_option(ctx.handle, key.encode(), bad) == _lib.ZK_ERR_BAD_ARG, (key, bad)
    assert L.zk_ctx_set_option(None, b"pre_vw", 64) == _lib.ZK_ERR_BAD_ARG and L.zk_ctx_set_option(ctx.handle, None, 64) == _lib.ZK_ERR_BAD_ARG
    cv = zk.get_curve(0)
    n = 1 << 15
    ck = _ck(ctx, cv, n, seed=91).precompute()
    polys = _polys(n, 3, 92)
    want = ck.commit_batch(polys)
    try:
        for opts in ({"pre_vw": 32, "pre_logg": 3}, {"pre_vw": 128, "pre_logg": 2}, {"chunk_l": 48}, {"combine_sg": 4}, {"combine_sg": 2}):
            for k, val in opts.items():
                ctx.set_option(k, val)
                assert ctx.get_option(k) == val
            assert ck.commit_batch(polys) == want, opts
            for k in opts:
                ctx.set_option(k, -1 if k == "pre_logg" else 0)
    finally:
        for k in ("pre_vw", "chunk_l", "combine_sg"):
            ctx.set_option(k, 0)
        ctx.set_option("pre_logg", -1)
    ck.close()


# ---- the memory budget of a deferred round (round 6): no room for another job's buffer set -> the queued jobs are
# closed early, their points parked, their sets reused; the round returns exactly the points it would have returned
def _budget_round(ck, ctx, p, z, chi):
    """sixteen jobs in the shape of the prover's last round, with a short vector (computed at begin) and an odd length in it"""
    ck.commit_begin(p[:4])
    ck.commit_begin([p[4][:1000], p[5][: p[5].shape[0] - 3], p[6]])
    ck.open_begin(p[:7] + p[10:12], z, chi)
    ck.commit_begin(p[7:14])
    return ck.open_begin(p[7:14], z, chi)


@pytest.mark.parametrize("cid,window", [(0, 0), (1, 0), (0, 20)])
def test_round_under_a_memory_budget_closes_early_and_returns_the_same_points(ctx, cid, window):
    cv = zk.get_curve(cid)
    n = 1 << 14
    ck = _ck(ctx, cv, n, seed=61).precompute(window)
    p = _polys(n, 14, 62)
    if cid == 1:
        for t in p:
            t[:, 3] >>= 2
    z = np.array([5, 6, 7, 8], dtype=np.uint64)
    chi = np.array([9, 10, 11, 12], dtype=np.uint64)
    # the reference round on a fresh ctx (the session's shared ctx still holds the sets of earlier, larger tests): sixteen sets
    ctx1 = zk.Context(ctx.device)
    ctx1.use_torch_stream()
    ck1 = ck.with_ctx(ctx1)
    assert _budget_round(ck1, ctx1, p, z, chi) == 16
    want = ck1.round_end()
    st0 = ctx1.round_mem_stats()
    ctx1.close()
    assert want[:4] == ck.commit_batch(p[:4]) and want[7] == ck.open(p[:7] + p[10:12], z, chi)
    per_set = st0["set_bytes"] // 16
    assert st0["device_total"] > st0["device_free"] > 0 and per_set > 0 and st0["early_closes"] == 0
    # fresh ctx (no buffer sets yet): the queued sets may hold three jobs' worth -> the round closes early several times
    ctx2 = zk.Context(ctx.device)
    ctx2.use_torch_stream()
    ck2 = ck.with_ctx(ctx2)
    try:
        ctx2.set_option("round_mem_limit_mb", max(1, (3 * per_set) >> 20) + 1)
        assert _budget_round(ck2, ctx2, p, z, chi) == 16 and ck2.round_pending() == 16
        got = ck2.round_end()
        st = ctx2.round_mem_stats()
        assert got == want
        assert st["early_closes"] >= 2, st
        assert st["set_bytes"] < st0["set_bytes"] // 2, (st, st0)        # the sets of closed jobs were reused, not allocated again
        # the same round again: the sets exist now, nothing new is needed, nothing closes early
        assert _budget_round(ck2, ctx2, p, z, chi) == 16
        assert ck2.round_end() == want
        # reduce-then-end and the Jacobian form see parked and queued jobs alike
        _budget_round(ck2, ctx2, p, z, chi)
        ck2.round_reduce()
        assert ck2.round_end() == want
        _budget_round(ck2, ctx2, p, z, chi)
        parts = ck2.round_end_partial(16)
        assert [zk.msm.sum_partials(parts[k:k + 1], cid) for k in range(16)] == want
        # the blocking batch comes in pieces under the same budget (a third ctx: no sets yet)
        ctx3 = zk.Context(ctx.device)
        ctx3.use_torch_stream()
        ctx3.set_option("round_mem_limit_mb", max(1, (2 * per_set) >> 20) + 1)
        ck3 = ck.with_ctx(ctx3)
        assert ck3.commit_batch(p[:7]) == ck.commit_batch(p[:7])
        assert ck3.commit_batch(p[7:14]) == want[8:15]
        assert ctx3.round_mem_stats()["early_closes"] >= 1
        # ... and so does the host-pointer batch (uploads per job in front of its digits)
        host = [t.cpu().numpy().view(np.uint64) for t in p[7:14]]
        assert ck3.commit_batch(host) == want[8:15]
        ctx3.close()
    finally:
        ctx2.set_option("round_mem_limit_mb", 0)
    ctx2.close()
    ck.close()


def test_round_under_a_memory_budget_device_exchange_forms(ctx):
    """Jobs parked by an early close enter the device forms of the multi-GPU exchange as jobs computed at submission do: their point
    as virtual-window sum S_0, every other sum the point at infinity."""
    import torch
    cv = zk.get_curve(0)
    n = 1 << 14
    ck = _ck(ctx, cv, n, seed=63).precompute()
    p = _polys(n, 6, 64)
    want = ck.commit_batch(p)
    ctx2 = zk.Context(ctx.device)
    ctx2.use_torch_stream()
    ck2 = ck.with_ctx(ctx2)
    ck2.commit_begin(p)
    assert ck2.round_end() == want
    per_set = ctx2.round_mem_stats()["set_bytes"] // 6
    ctx2.close()
    ctx2 = zk.Context(ctx.device)
    ctx2.use_torch_stream()
    ck2 = ck.with_ctx(ctx2)
    ctx2.set_option("round_mem_limit_mb", max(1, (2 * per_set) >> 20) + 1)
    words = ck2.winsums_dev_words()
    buf = torch.zeros((6, words), dtype=torch.int64, device="cuda")
    for t in p:
        ck2.commit_begin([t])
    ck2.round_end_winsums_dev(buf, 6)
    assert ctx2.round_mem_stats()["early_closes"] >= 1
    assert ck2.sum_winsums_dev(buf.reshape(1, -1), 1, 6) == want
    ctx2.close()
    ck.close()


def test_round_budget_by_the_device_free_memory_rule(ctx):
    """The same early close through the REAL rule -- hipMemGetInfo minus option "mem_reserve_mb" -- instead of the test hook: the reserve
    is set so that about three buffer sets of a 2^18-point job still fit beside it, on a fresh ctx that owns none yet."""
    cv = zk.get_curve(0)
    n = 1 << 18
    ck = _ck(ctx, cv, n, seed=65).precompute()
    p = _polys(n, 8, 66)
    want = ck.commit_batch(p)
    ctx1 = zk.Context(ctx.device)
    ctx1.use_torch_stream()
    ck1 = ck.with_ctx(ctx1)
    ck1.commit_begin(p)
    assert ck1.round_end() == want
    per_set = ctx1.round_mem_stats()["set_bytes"] // 8
    ctx1.close()
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    ctx2 = zk.Context(ctx.device)
    ctx2.use_torch_stream()
    ck2 = ck.with_ctx(ctx2)
    free = ctx2.round_mem_stats()["device_free"]
    assert free > 4 * per_set
    ctx2.set_option("mem_reserve_mb", int((free - 3.5 * per_set) // (1 << 20)))
    for t in p:
        ck2.commit_begin([t])
    assert ck2.round_pending() == 8
    got = ck2.round_end()
    st = ctx2.round_mem_stats()
    assert got == want
    assert st["early_closes"] >= 1 and st["set_bytes"] <= 4.5 * per_set, (st, per_set)
    ctx2.set_option("mem_reserve_mb", 1024)
    ctx2.close()
    ck.close()
