"""The per-proof call schedule (ark_plonk_amd/prover_schedule.py <- prover.rs:163-638) on a small domain."""
import numpy as np
import pytest

import ark_plonk_amd as zk
from ark_plonk_amd import _lib
from ark_plonk_amd.prover_schedule import ProofSchedule

pytestmark = pytest.mark.gpu


def _ck(ctx, cv, n):
    import torch
    g = torch.Generator(device="cuda").manual_seed(3)
    ks = torch.randint(1, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    ks[:, 1:] = 0
    bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cv.curve_id, ks.data_ptr(), n, bases.data_ptr()))
    return zk.CommitterKey(bases, cv, ctx)


@pytest.mark.parametrize("log_n", [10, 13])
def test_schedule_table_path_matches_plain_path_and_dedup(ctx, log_n):
    """29 outputs per proof; the window-table path, the per-window path and the de-duplicated schedule
    (17 MSMs) must agree point for point."""
    cv = zk.get_curve("bls12_381")
    n = 1 << log_n
    ck = _ck(ctx, cv, n)
    plain = ProofSchedule(log_n, ctx, ck, cv).run_once()
    assert len(plain) == 29
    ck.precompute()
    s_tab = ProofSchedule(log_n, ctx, ck, cv)
    tab = s_tab.run_once()
    assert s_tab.msms_run == 29
    s_dd = ProofSchedule(log_n, ctx, ck, cv, dedup=True)
    dd = s_dd.run_once()
    assert s_dd.msms_run == 20          # first proof: the prover key's sigma commitments are computed once ...
    dd2 = s_dd.run_once(proof_id=0)
    assert s_dd.msms_run == 17          # ... and stay cached: lin, table, W_z, W_zw + the 13 of rounds 1-4
    fused = ProofSchedule(log_n, ctx, ck, cv, fuse_round5=True).run_once()
    for a, b, c, e, f in zip(plain, tab, dd, dd2, fused):
        assert a == b and a == c and a == e and a == f
    # round 5 re-commits polynomials of rounds 1-3 (prover.rs:569-607): f, h2 in aw; z, w_l, ... in saw
    assert tab[17] == tab[4] and tab[18] == tab[6] and tab[21] == tab[7] and tab[22] == tab[0] and tab[27] == tab[19]
    ck.close()


def test_schedule_with_device_grand_products(ctx):
    """N2 inside the schedule: z / z2 built on the device change exactly the outputs that depend on them."""
    cv = zk.get_curve("bls12_381")
    log_n = 11
    ck = _ck(ctx, cv, 1 << log_n).precompute()
    base = ProofSchedule(log_n, ctx, ck, cv).run_once()
    gp = ProofSchedule(log_n, ctx, ck, cv, grand_products=True).run_once()
    again = ProofSchedule(log_n, ctx, ck, cv, grand_products=True).run_once()
    assert all(a == b for a, b in zip(gp, again))
    same = [a == b for a, b in zip(base, gp)]
    # rounds 1-2 (outputs 0..6) and the quotient commitments (9..12) do not involve z / z2 here
    assert all(same[:7]) and all(same[9:13]) and not same[7] and not same[8]
    ck.close()


def test_schedule_with_device_quotient(ctx):
    """N1 inside the schedule: only the four quotient commitments t_1..t_4 (outputs 9..12) depend on it."""
    cv = zk.get_curve("bls12_381")
    log_n = 11
    ck = _ck(ctx, cv, 1 << log_n).precompute()
    base = ProofSchedule(log_n, ctx, ck, cv).run_once()
    q = ProofSchedule(log_n, ctx, ck, cv, quotient=True).run_once()
    same = [a == b for a, b in zip(base, q)]
    assert all(same[:9]) and not any(same[9:13]) and all(same[13:])
    ck.close()


def test_config1_plumbing_size_end_to_end_vs_cpu_oracle(ctx, oracle_cpu):
    """BASELINE config 1 (examples/simple_circuit.rs: padded size 2^9, SRS of 2^10): the whole per-proof schedule on the GPU --
    31 transforms, 27 commitments, 2 openings, per-window MSM path (the SRS is below the table threshold) -- against the same
    schedule recomputed with the CPU restatement of the ark 0.3 algorithms: all 29 points equal."""
    import torch
    cid, log_n = 0, 9
    n = 1 << log_n
    cv = zk.get_curve(cid)
    ks = np.zeros((2 * n, 4), dtype=np.uint64)
    ks[:, 0] = 7 + 5 * np.arange(2 * n, dtype=np.uint64)
    bases = torch.empty((2 * n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, torch.from_numpy(ks.view(np.int64)).cuda().data_ptr(), 2 * n, bases.data_ptr()))
    ck = zk.CommitterKey(bases, cv, ctx)
    sched = ProofSchedule(log_n, ctx, ck, cv)
    got = sched.run_once(proof_id=0)
    srs_h = bases.cpu().numpy().view(np.uint64)
    H = lambda t: t.cpu().numpy().view(np.uint64)          # noqa: E731
    ifft = lambda e: oracle_cpu.ntt(cid, 1, log_n, H(e))   # noqa: E731
    c = [ifft(sched.evals[i]) for i in range(4)]
    c += [ifft(sched.aux_evals[k]) for k in range(6)]       # table f h1 h2 z z2
    c.append(ifft(sched.aux_evals[6]))                       # pi
    c.append(ifft(sched.aux_evals[7]))                       # l1 (stands in for lin)
    t = oracle_cpu.ntt(cid, 3, log_n + 2, H(sched.quot))
    sig = [H(s) for s in sched.sigma]
    commit = lambda p: oracle_cpu.kzg_commit(cid, srs_h, p)  # noqa: E731

    def opening(polys):
        comb = np.zeros((n, 4), dtype=np.uint64)
        chi_pow = oracle_cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))
        for p in polys:
            term = oracle_cpu.fr_op(cid, "mul", p, np.repeat(chi_pow, p.shape[0], axis=0))
            comb[: p.shape[0]] = oracle_cpu.fr_op(cid, "add", comb[: p.shape[0]], term)
            chi_pow = oracle_cpu.fr_op(cid, "mul", chi_pow, sched.chi_mont.reshape(1, 4))
        return commit(oracle_cpu.kzg_witness(cid, comb, sched.z_mont))

    aw = [c[11], sig[0], sig[1], sig[2], c[5], c[7], c[4]]
    saw = [c[8], c[0], c[1], c[3], c[6], c[9], c[4]]
    exp = [commit(p) for p in c[:4]] + [commit(c[5]), commit(c[6]), commit(c[7]), commit(c[8]), commit(c[9])]
    exp += [commit(t[i * n:(i + 1) * n]) for i in range(4)]
    exp += [commit(p) for p in aw] + [opening(aw + c[:4])] + [commit(p) for p in saw] + [opening(saw)]
    assert len(exp) == 29
    for k, (g, (xy, inf)) in enumerate(zip(got, exp)):
        assert g.infinity == bool(inf) and np.array_equal(g.xy(), xy), k
    ck.close()


def test_schedule_with_linearisation_on_device(ctx):
    """linearisation=True: `lin` is the 19-term sum built from the round's polynomials and the 23 evaluations come back
    (linearisation_poly.rs:164-350); everything that does not depend on `lin` is unchanged."""
    from ark_plonk_amd import linearisation
    cv = zk.get_curve("bls12_381")
    log_n = 10
    ck = _ck(ctx, cv, 1 << log_n)
    plain = ProofSchedule(log_n, ctx, ck, cv).run_once(proof_id=0)
    s = ProofSchedule(log_n, ctx, ck, cv, linearisation=True)
    got = s.run_once(proof_id=0)
    changed = [k for k in range(29) if got[k] != plain[k]]
    assert changed == [13, 20]                       # commit(lin) and the opening at z (aw_open holds lin)
    assert set(s.last_evals) == set(linearisation.PROOF_EVALS + linearisation.CUSTOM_EVALS)
    again = s.run_once(proof_id=0)
    assert again == got


def test_schedule_with_round2_on_device(ctx):
    """lookup_round2=True: table / f / h_1 / h_2 come from four table columns, q_lookup and the wires (prover.rs:228-317);
    h_1 and h_2 have n rows each, the run is reproducible, and with the grand products on, z_2 closes (its last step returns to 1
    only if (f, t, h_1, h_2) really are a Plonkup instance: zk_lookup_product_dev reports the closing value)."""
    import torch
    from ark_plonk_amd import lookup, permutation
    cv = zk.get_curve("bls12_381")
    log_n = 10
    n = 1 << log_n
    ck = _ck(ctx, cv, n)
    s = ProofSchedule(log_n, ctx, ck, cv, lookup_round2=True, grand_products=True)
    a = s.run_once(proof_id=0)
    b = s.run_once(proof_id=0)
    assert a == b and len(a) == 29
    t = lookup.compress_table(s.table_cols, s.zeta_mont, cv, ctx)
    f = lookup.compress_query(s.q_lookup, s.evals, s.zeta_mont, t, curve=cv, ctx=ctx)
    h1, h2 = lookup.combine_split(t, f, cv, ctx)
    assert h1.shape[0] == n and h2.shape[0] == n
    z2, last = permutation.lookup_permutation_evals(ctx, cv, f, t, h1, h2, s.chi_mont, s.z_mont, return_last=True)
    one = zk.curves.fr_to_mont(cv, [1])[0]
    assert np.array_equal(np.asarray(last, dtype=np.uint64).reshape(4), one)
    assert np.array_equal(z2[0].cpu().numpy().view(np.uint64), one)
