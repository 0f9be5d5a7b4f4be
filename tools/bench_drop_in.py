"""bench.py's `drop_in` leg: K proofs through the HOST-POINTER entry points a Rust shim binds -- zk_ntt, zk_kzg_commit_batch,
zk_kzg_open on pageable numpy buffers, SRS registered once -- then the same caller with the commitment cache, with the residency cache, and
as T concurrent callers.  Moved out of bench.py in round 6; `E` is the namespace bench.py builds (args, zk, torch, ctx, cv, build_srs, digest)."""
from __future__ import annotations

import time

import numpy as np


def drop_in_region(E, k: int, log_n: int):
    """K proofs through the host-pointer entry points, pageable numpy buffers, SRS registered once."""
    from ark_plonk_amd.prover_schedule import DropInSchedule
    args, zk, torch, ctx, cv, build_srs, digest = E.args, E.zk, E.torch, E.ctx, E.cv, E.build_srs, E.digest
    n = 1 << log_n
    srs = build_srs(ctx, cv, n, 0, n, torch).cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    ck = zk.CommitterKey(srs, cv, ctx)                  # zk_srs_register: upload + digest
    t_reg = time.perf_counter() - t0
    if not args.no_precompute:
        ck.precompute(args.table_window)
    t0 = time.perf_counter()
    ck2 = zk.CommitterKey(srs, cv, ctx)                 # PC::trim on the next gen_proof: a cache hit
    t_hit = time.perf_counter() - t0
    sched = DropInSchedule(log_n, ctx, ck2, cv)
    sched.run_once()
    ctx.io_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(k):
        sched.run_once()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    io = ctx.io_stats()
    pts = sched.run_once(proof_id=0) if args.check else None
    # the same caller with the library's commitment cache switched on (one call in the shim)
    ctx.set_commit_cache(True)
    sched.run_once()
    t0 = time.perf_counter()
    for _ in range(k):
        sched.run_once()
    torch.cuda.synchronize()
    dt_cache = time.perf_counter() - t0
    ctx.set_commit_cache(False)
    # the same caller with the library's residency cache switched on (one more call in the shim): vectors the library produced or has
    # seen are not uploaded again -- every proof has its own witness, so the evaluation vectors still miss, as they would in production
    res = None
    try:
        ctx.set_residency_cache(True)
        for _ in range(5):                  # until the cache has reached its capacity: from then on evicted buffers are reused and
            sched.run_once()                # no call allocates device memory any more (the steady state of a proving service)
        ctx.io_stats(reset=True)
        st0 = ctx.residency_cache_stats()
        t0 = time.perf_counter()
        for _ in range(k):
            sched.run_once()
        torch.cuda.synchronize()
        dt_res = time.perf_counter() - t0
        io_r = ctx.io_stats()
        st1 = ctx.residency_cache_stats()
        pts_r = sched.run_once(proof_id=0) if args.check else None
        res = {"proofs_per_s": k / dt_res, "ms_per_proof": dt_res / k * 1e3, "h2d_bytes_per_proof": io_r["h2d_bytes"] // k,
               "d2h_bytes_per_proof": io_r["d2h_bytes"] // k, "hits_per_proof": (st1["hits"] - st0["hits"]) / k,
               "misses_per_proof": (st1["misses"] - st0["misses"]) / k, "resident_bytes": st1["bytes"],
               "same_points_as_uncached": (digest(pts_r) == digest(pts)) if args.check else None,
               "how": "zk_ctx_set_residency_cache(ctx, 1, 0, 0): zk_ntt keeps the device copy of every output of at most 64 MiB under a keyed 256-bit "
                      "digest of the bytes the caller receives; zk_ntt / zk_kzg_commit_batch / zk_kzg_open digest their inputs on the host pool and "
                      "use the resident copy on a match (prover.rs:196-213,569-618: an ifft output goes back up as a commit, coset_fft and "
                      "opening input)"}
    except Exception as e:
        res = {"error": repr(e)}
    finally:
        ctx.set_residency_cache(False)
    # T unchanged callers at once (a proving service running `Prover::prove` in T worker threads): one zk_ctx, one proof and one set
    # of pageable vectors per thread, ONE resident SRS; a caller's transfers run under the other callers' kernels
    callers = None
    try:
        import threading
        T = args.drop_in_callers
        ctxs = [zk.Context(ctx.device) for _ in range(T)]
        cks = [zk.CommitterKey(srs, cv, c) for c in ctxs]
        scheds = [DropInSchedule(log_n, c, ckc, cv) for c, ckc in zip(ctxs, cks)]
        for s_ in scheds:
            s_.run_once()
        bar = threading.Barrier(T + 1)
        errs = []

        def caller(s_):
            try:
                bar.wait()
                for _ in range(k):
                    s_.run_once()
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                errs.append(repr(e))
            finally:
                bar.wait()
        ths = [threading.Thread(target=caller, args=(s_,)) for s_ in scheds]
        for t_ in ths:
            t_.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt_c = time.perf_counter() - t0
        for t_ in ths:
            t_.join()
        same = all(digest(s_.run_once(proof_id=0)) == digest(pts) for s_ in scheds) if args.check else None
        callers = {"callers": T, "proofs_per_s": T * k / dt_c, "ms_per_proof_per_caller": dt_c / k * 1e3, "proofs_each": k,
                   "same_points_as_one_caller": same, "errors": errs or None,
                   "how": "T host threads, each with its own zk_ctx, proof and pageable vectors, all calling zk_ntt / zk_kzg_commit_batch / zk_kzg_open "
                          "against one GPU and one resident SRS (residency cache off); tools/drop_in_callers.py sweeps T = 1..8"}
        for ckc in cks:
            ckc.close()
        for c in ctxs:
            c.close()
        del scheds
    except Exception as e:
        callers = {"error": repr(e)}
    out = {"proofs_per_s": k / dt, "proofs_per_s_with_commit_cache": k / dt_cache, "with_residency_cache": res, "concurrent_callers": callers,
           "ms_per_proof": dt / k * 1e3, "steps": k,
           "h2d_bytes_per_proof": io["h2d_bytes"] // k, "d2h_bytes_per_proof": io["d2h_bytes"] // k,
           "pcie_GBps_over_whole_proof": (io["h2d_bytes"] + io["d2h_bytes"]) / dt / 1e9,
           "srs_register_ms_first": t_reg * 1e3, "srs_register_ms_cached": t_hit * 1e3, "srs_cache": zk.srs_cache_stats(),
           "calls": "31 zk_ntt (in place on caller vectors) + 9 zk_kzg_commit_batch (4|1|1|1|1|1|4|7|7 polynomials) + 2 zk_kzg_open per proof; "
                    "pageable host buffers, reused across proofs",
           "digest": digest(pts) if args.check else None}
    ck2.close()
    ck.close()
    zk.srs_cache_config(0)          # drop the resident copy before the next leg
    zk.srs_cache_config(32 << 30)
    return out

