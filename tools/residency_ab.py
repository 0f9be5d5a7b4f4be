"""Micro A/B of the residency cache (zk_ctx_set_residency_cache) on the three host-pointer calls it serves, n = 2^20:
coset_fft of a resident coefficient vector, a 7-polynomial commit batch, an 11-polynomial opening -- cache off / on."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import ark_plonk_amd as zk
    from bench import build_srs
    log_n = 20
    n = 1 << log_n
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    srs = build_srs(ctx, cv, n, 0, n, torch).cpu().numpy().view(np.uint64)
    ck = zk.CommitterKey(srs, cv, ctx).precompute()
    d = zk.Radix2EvaluationDomain.new(n, cv, ctx)
    d4 = zk.Radix2EvaluationDomain.new(4 * n, cv, ctx)
    rng = np.random.default_rng(1)
    evs = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(11)]
    coef = [np.zeros((n, 4), dtype=np.uint64) for _ in range(11)]
    ev4 = np.zeros((4 * n, 4), dtype=np.uint64)
    z = np.array([0x1234567, 0x89abcdef, 0x13579bdf, 0x0fedcba9], dtype=np.uint64)
    chi = np.array([0x2468ace, 0x7654321, 0x2222222, 0x0111111], dtype=np.uint64)

    def med(f, reps=7):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    for on in (False, True):
        ctx.set_residency_cache(on)
        for k in range(11):
            d._run(1, evs[k], out=coef[k])            # produces (and, with the cache on, keeps) the coefficient vectors
        t_ifft = med(lambda: d._run(1, evs[0], out=coef[0]))
        t_coset = med(lambda: d4._run(2, coef[1], out=ev4))
        t_commit7 = med(lambda: ck.commit_batch(coef[:7]))
        t_commit1 = med(lambda: ck.commit_batch(coef[:1]))
        t_open = med(lambda: ck.open(coef, z, chi))
        st = ctx.residency_cache_stats()
        print(f"cache {'on ' if on else 'off'}: ifft(n) {t_ifft:6.2f}  coset_fft(4n) {t_coset:6.2f}  commit x7 {t_commit7:6.2f}  commit x1 {t_commit1:6.2f}  "
              f"open x11 {t_open:6.2f} ms   stats {st}", flush=True)
    ctx.set_residency_cache(False)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def schedule_breakdown():
    """the whole drop-in schedule, time per kind of call, cache off / on"""
    import collections
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import DropInSchedule
    from bench import build_srs
    log_n = 20
    n = 1 << log_n
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    srs = build_srs(ctx, cv, n, 0, n, torch).cpu().numpy().view(np.uint64)
    ck = zk.CommitterKey(srs, cv, ctx).precompute()
    sched = DropInSchedule(log_n, ctx, ck, cv)
    acc = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    hm = collections.defaultdict(lambda: [0, 0])

    def wrap(obj, name, key_fn):
        orig = getattr(obj, name)

        def f(*a, **k):
            s0 = ctx.residency_cache_stats()
            t0 = time.perf_counter()
            r = orig(*a, **k)
            key = key_fn(*a, **k)
            acc[key] += time.perf_counter() - t0
            cnt[key] += 1
            s1 = ctx.residency_cache_stats()
            hm[key][0] += s1["hits"] - s0["hits"]
            hm[key][1] += s1["misses"] - s0["misses"]
            return r
        setattr(obj, name, f)
    wrap(sched.dom_n, "_run", lambda kind, x, out=None: f"ntt_n kind {kind}")
    wrap(sched.dom_4n, "_run", lambda kind, x, out=None: f"ntt_4n kind {kind}")
    wrap(ck, "commit_batch", lambda polys, **k: f"commit x{len(polys)}")
    wrap(ck, "open", lambda polys, *a: f"open x{len(polys)}")
    for on in (False, True):
        ctx.set_residency_cache(on)
        sched.run_once()
        acc.clear()
        cnt.clear()
        hm.clear()
        k = 3
        t0 = time.perf_counter()
        for _ in range(k):
            sched.run_once()
        dt = (time.perf_counter() - t0) / k * 1e3
        print(f"cache {'on ' if on else 'off'}: {dt:7.2f} ms per proof | " + "  ".join(f"{key}: {cnt[key] // k} x {acc[key] / cnt[key] * 1e3:.2f}" for key in sorted(acc)), flush=True)
        if on:
            print("   hits / misses per proof: " + "  ".join(f"{key}: {hm[key][0] / k:.0f} / {hm[key][1] / k:.0f}" for key in sorted(hm)), flush=True)
    ctx.set_residency_cache(False)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "schedule":
    schedule_breakdown()
