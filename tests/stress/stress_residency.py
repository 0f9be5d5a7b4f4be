"""Randomised cross-check of the residency cache of the host-pointer entry points (zk_ctx_set_residency_cache, round 5): a pool of host
vectors goes through random zk_ntt calls (all four kinds, in place or not), zk_kzg_commit_batch and zk_kzg_open calls on a ctx WITH the
cache, and is now and then rewritten by the "caller" between calls -- at one element, wholesale, or swapped with another vector, i.e.
every way a stale device copy could be taken for the current bytes.  Every result is compared with a second ctx WITHOUT the cache (and
the transforms and commitments also with the CPU restatement): a hit is only ever allowed to be a copy of the caller's current bytes.
Small capacities force evictions in the middle of batches.
usage: [SEED=..] python tests/stress/stress_residency.py [seconds]   (a short budget runs under tests/test_stress_gpu.py)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import _lib  # noqa: E402
from oracle import cpu  # noqa: E402


def run(budget: float = 60.0, seed: int = 5, ctx=None, max_log_n: int = 15, verify: bool = False):
    """verify=True: the same session with option "cache_verify" on (and the commitment cache too): every hit is compared with the
    caller's bytes / recomputed before it is believed; not one may be found wrong."""
    cpu.build()
    own = ctx is None
    if own:
        ctx = zk.Context(0)
    plain = zk.Context(ctx.device)                       # the same calls without the cache
    v0 = ctx.cache_verify_stats()
    if verify:
        ctx.set_option("cache_verify", 1)
        ctx.set_commit_cache(True)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    calls = hits0 = 0
    rounds = 0
    try:
        while time.time() < t_end:
            cid = int(rng.integers(0, 2))
            cv = zk.get_curve(cid)
            log_n = int(rng.integers(8, max_log_n + 1))
            n = 1 << log_n
            ks = np.zeros((n, 4), dtype=np.uint64)
            ks[:, 0] = rng.integers(1, 1 << 40, size=n, dtype=np.uint64)
            bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
            ctx.use_torch_stream()
            _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, torch.from_numpy(ks.view(np.int64)).cuda().data_ptr(), n, bases.data_ptr()))
            torch.cuda.synchronize()
            bases_h = bases.cpu().numpy().view(np.uint64)
            ck = zk.CommitterKey(bases_h, cv, ctx)
            ck_p = zk.CommitterKey(bases_h, cv, plain)
            if rng.random() < 0.7 and n >= 8192:
                ck.precompute()
                ck_p.precompute()
            dom = zk.Radix2EvaluationDomain.new(n, cv, ctx)
            dom_p = zk.Radix2EvaluationDomain.new(n, cv, plain)
            dom4 = zk.Radix2EvaluationDomain.new(4 * n, cv, ctx)
            dom4_p = zk.Radix2EvaluationDomain.new(4 * n, cv, plain)

            def rnd(rows):
                x = rng.integers(0, 1 << 62, size=(rows, 4), dtype=np.uint64)
                if cid == 1:
                    x[:, 3] >>= np.uint64(2)
                return x
            pool = [rnd(n) for _ in range(6)]
            big = np.zeros((4 * n, 4), dtype=np.uint64)
            # capacities from "holds everything" down to "two vectors": evictions inside batches and openings
            cap = int(rng.choice([0, 64 * n * 32, 4 * n * 32, 2 * n * 32]))
            ctx.set_residency_cache(True, cap if cap else (2 << 30), int(rng.choice([n * 32, 4 * n * 32])))
            st0 = ctx.residency_cache_stats()["hits"]
            for _ in range(int(rng.integers(10, 30))):
                op = rng.random()
                i = int(rng.integers(0, len(pool)))
                if op < 0.30:            # a transform of one pool vector, into another one or in place
                    kind = int(rng.integers(0, 4))
                    j = i if rng.random() < 0.4 else int(rng.integers(0, len(pool)))
                    src = pool[i].copy()
                    if kind >= 2:        # coset transforms on the 4n domain: n coefficients in (kind 2), 4n evaluations in (kind 3)
                        if kind == 2:
                            got = dom4._run(2, pool[i], out=big).copy()
                            exp = dom4_p._run(2, src)
                        else:
                            ev = big.copy()
                            got = dom4._run(3, big, out=big).copy()
                            exp = dom4_p._run(3, ev)
                        assert np.array_equal(got, exp), (cid, log_n, kind)
                        if kind == 3 and rng.random() < 0.5:
                            q = int(rng.integers(0, 4))
                            pool[j][:] = big[q * n:(q + 1) * n]      # a quarter of the quotient becomes a polynomial (split_tx_poly)
                    else:
                        got = dom._run(kind, pool[i], out=pool[j])
                        exp = dom_p._run(kind, src)
                        assert np.array_equal(got, exp) and np.array_equal(got, cpu.ntt(cid, kind, log_n, src)), (cid, log_n, kind, i, j)
                elif op < 0.55:          # PC::commit of a slice of the pool (repeats allowed)
                    k = int(rng.integers(1, 6))
                    idx = [int(rng.integers(0, len(pool))) for _ in range(k)]
                    lens = [int(rng.choice([n, n, n - 1, n // 2 + 1])) for _ in range(k)]
                    polys = [pool[a][:ln] for a, ln in zip(idx, lens)]
                    got = ck.commit_batch(polys)
                    exp = ck_p.commit_batch([p.copy() for p in polys])
                    assert got == exp, (cid, log_n, idx, lens)
                    a0 = polys[0]
                    exp_xy, exp_inf = cpu.kzg_commit(cid, bases_h, np.ascontiguousarray(a0))
                    assert got[0].infinity == bool(exp_inf) and np.array_equal(got[0].xy(), exp_xy)
                elif op < 0.70:          # PC::open
                    k = int(rng.integers(1, 8))
                    idx = [int(rng.integers(0, len(pool))) for _ in range(k)]
                    z, chi = rnd(1)[0], rnd(1)[0]
                    got = ck.open([pool[a] for a in idx], z, chi)
                    exp = ck_p.open([pool[a].copy() for a in idx], z, chi)
                    assert got == exp, (cid, log_n, idx)
                elif op < 0.80:          # the caller rewrites one element of a vector the library may hold a copy of
                    pool[i][int(rng.integers(0, n)), int(rng.integers(0, 3))] ^= np.uint64(1 << int(rng.integers(0, 60)))
                elif op < 0.88:          # ... or the whole vector
                    pool[i][:] = rnd(n)
                elif op < 0.94:          # ... or swaps the contents of two buffers (pointers stay, bytes move)
                    j = int(rng.integers(0, len(pool)))
                    tmp = pool[i].copy()
                    pool[i][:] = pool[j]
                    pool[j][:] = tmp
                else:                    # ... or the library is told to forget, and to remember again
                    ctx.set_residency_cache(False)
                    ctx.set_residency_cache(True)
                calls += 1
            hits0 += ctx.residency_cache_stats()["hits"] - st0
            ctx.set_residency_cache(False)
            assert ctx.residency_cache_stats()["entries"] == 0
            ck.close()
            ck_p.close()
            rounds += 1
    finally:
        ctx.set_residency_cache(False)
        v1 = ctx.cache_verify_stats()
        if verify:
            ctx.set_commit_cache(False)
            ctx.set_option("cache_verify", 0)
        plain.close()
        if own:
            ctx.close()
    if verify:
        assert v1["checked"] - v0["checked"] > 0 and v1["mismatches"] == v0["mismatches"], (v0, v1)
        print(f"cache_verify: {v1['checked'] - v0['checked']} hits checked against the caller's bytes / a recomputation, 0 mismatches", flush=True)
    print(f"stress ok: {rounds} rounds, {calls} host-pointer calls / caller rewrites with the residency cache on ({hits0} hits) equal to the same "
          f"calls without it and to the CPU restatement (seed {seed}, {budget:.0f} s)", flush=True)
    return calls


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(os.environ.get("SEED", "5")), verify=os.environ.get("VERIFY", "0") == "1")
