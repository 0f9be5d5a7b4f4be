"""A/B of library tuning options INSIDE one process: the pool's boxes drift by several per cent with their power state (a box is
slower right after a long run: tools/power_probe.py), so two bench.py runs one after the other cannot resolve a 1 % effect.  Here one
SRS, one table and one schedule serve every configuration; the configurations alternate proof by proof and the report is the
per-configuration median plus the paired differences against the first one.

  python tools/ab_proof.py --pairs 12 msm_merge=0 msm_merge=1 "msm_merge=1 long_rounds=3"

Each configuration is a space-separated list of option=value -- the keys of zk_ctx_set_option (include/ark_plonk_amd.h), set on the
ctx between proofs (never while a round is open); the ZK_* environment spellings of rounds 2-4 are accepted and mapped
(ZK_MSM_MERGE=0 -> msm_merge=0): the library itself no longer reads the environment.  SCHED.name=value: a ProofSchedule keyword."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--pairs", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--proofs", type=int, default=2, help="timed proofs per visit of a configuration (after one untimed)")
    ap.add_argument("--block-every-call", action="store_true")
    ap.add_argument("--curve", default="bls12_381", choices=["bls12_381", "bn254"])
    args = ap.parse_args()
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import ProofSchedule
    from bench import build_srs

    # option=value: a tuning option of the ctx; SCHED.name=value: a keyword of ProofSchedule (one schedule object per distinct set)
    def conv(v):
        return {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
    raw = [dict(kv.split("=", 1) for kv in c.split()) for c in args.configs]
    def opt_key(k):
        return k[3:].lower() if k.startswith("ZK_") else k
    cfgs = [{opt_key(k): int(v) for k, v in c.items() if not k.startswith("SCHED.")} for c in raw]
    sched_kw = [tuple(sorted((k[6:], conv(v)) for k, v in c.items() if k.startswith("SCHED."))) for c in raw]
    keys = sorted({k for c in cfgs for k in c})
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve(args.curve)
    n = 1 << args.log_n
    srs = build_srs(ctx, cv, n, 0, n, torch)
    ck = zk.CommitterKey(srs, cv, ctx)
    del srs
    ck.precompute(0)
    scheds = {kw: ProofSchedule(args.log_n, ctx, ck, cv, defer_calls=not args.block_every_call, **dict(kw)) for kw in set(sched_kw)}

    defaults = {k: ctx.get_option(k) for k in keys}

    def use(c):
        for k in keys:
            ctx.set_option(k, c.get(k, defaults[k]))

    import hashlib
    digs = []
    for c, kw in zip(cfgs, sched_kw):
        use(c)
        sched = scheds[kw]
        pts = sched.run_once(proof_id=0)
        digs.append(hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in pts)).hexdigest()[:12])
        sched.run_once()
    torch.cuda.synchronize()
    times = [[] for _ in cfgs]
    for _ in range(args.pairs):
        for i, c in enumerate(cfgs):
            use(c)
            sched = scheds[sched_kw[i]]
            sched.run_once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.proofs):
                sched.run_once()
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / args.proofs * 1e3)
    print(f"{args.curve} n = 2^{args.log_n}, {args.pairs} visits x {args.proofs} proofs per configuration, alternating; ms per proof")
    for i, c in enumerate(args.configs):
        t = times[i]
        d = [a - b for a, b in zip(t, times[0])]
        print(f"  [{i}] {c:45s} median {statistics.median(t):7.3f}  mean {statistics.mean(t):7.3f}  min {min(t):7.3f}  "
              f"vs [0]: median {statistics.median(d):+6.3f} ms ({statistics.median(d) / statistics.median(times[0]) * 100:+5.2f} %)  digest {digs[i]}")
    ck.close()
    ctx.close()


if __name__ == "__main__":
    main()
