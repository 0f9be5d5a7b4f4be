#!/usr/bin/env python3
"""bench.py -- proofs/s of the NTT + MSM prover hot path on MI355X (BASELINE.json metric).

A "step" = one proof's hot path: the reference's exact per-proof schedule of 31 transforms and 29
KZG commitments (ark_plonk_amd/prover_schedule.py <- prover.rs:163-638) at n = 2^20 constraints,
BLS12-381 + KZG10, on synthetic polynomials with every input (SRS, evaluation vectors) already
resident in HBM.  N > 1: one process per GPU (torchrun).  `value` is the throughput of N replicas --
whole proofs are independent, so every rank runs the schedule K times with no data-path collective
(SURVEY.md 8e "whole proofs: replicas only", scaling "weak").  The same run then times a second leg,
reported under `msm_sharded` and never part of `value`: one proof stream with every MSM sharded by
points over the ranks and combined by an RCCL all-gather of Jacobian partials per prover round (NTTs
replicated) -- the single-proof latency mode.  `--mode shard` makes that leg the headline instead
(scaling "strong").

N = 1 adds a `concurrent_streams` leg (also never part of `value`): the same GPU with 4 proofs in flight
(one thread + zk_ctx + HIP stream each), i.e. the throughput a proving service would see.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (msm_accumulate) with the
algorithmic bytes of SURVEY.md 8d (128 B per point) over its HIP-event-timed launches;
`cpu_baseline` times the oracle's CPU restatement of the ark 0.3 algorithms on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# The kernel is integer-VALU bound (SURVEY.md 8d asks for the u32-MAC rate next to the HBM fraction).
# Instruction counts of the mixed-addition path of msm_accumulate (gfx950 ISA of this build, BLS12-381:
# 6 products + 2 squares + 1 double product of 14 x 29-bit limbs) and the measured issue costs of
# profiles/r01_ubench_valu.txt.
MADD_MADS = 3542            # v_mad_u64_u32 per mixed addition
MADD_OTHER_VALU = 1221      # and/shift/add/mul_lo around them
CLOCK_HZ = 2.4e9            # MI355X max engine clock (MI355X_MICROARCH.md chip table)
LANES = 256 * 4 * 64        # CUs x SIMDs x lanes
MAD_CYCLES_FULL = 3.99      # cycles per wave-instruction per SIMD at >= 4 waves/SIMD
# issue cycles of one mixed addition at the kernel's 2 waves/SIMD (220 VGPRs): mad 4.77, mul_lo 4.70, 64-bit shift/add 4.45, rest 2.64
MADD_CYCLES_2WAVES = 3542 * 4.77 + 126 * 4.70 + 468 * 4.45 + 627 * 2.64


def ark_adds(n: int, bits: int = 255) -> int:
    """reference-equivalent G1 additions of one MSM: W*N + 2*W*(2^c - 1)."""
    if n < 32:
        c = 3
    else:
        c = (n - 1).bit_length() * 69 // 100 + 2
    w = -(-bits // c)
    return w * n + 2 * w * ((1 << c) - 1)


def build_srs(ctx, cv, n, lo, hi, torch):
    """Synthetic KZG SRS slice P_i = tau^i G for i in [lo, hi), generated on the GPU."""
    from ark_plonk_amd import _lib, curves
    tau = 0x7A5C0DE
    t = pow(tau, lo, cv.r)
    pw = []
    for _ in range(hi - lo):
        pw.append(t)
        t = t * tau % cv.r
    sc = torch.from_numpy(curves.ints_to_limbs(pw, 4).view(np.int64)).cuda()
    out = torch.empty((hi - lo, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    ctx.use_torch_stream()
    _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cv.curve_id, sc.data_ptr(), hi - lo, out.data_ptr()))
    torch.cuda.synchronize()
    return out


def cpu_baseline(log_n: int, cid: int = 0, bits: int = 255):
    """Bounded sample of the same workload on the host cores with the oracle's CPU restatement."""
    from oracle import cpu
    cpu.build()
    cores = cpu.num_threads()
    rng = np.random.default_rng(1)
    s_ntt = min(log_n, 18)
    s_msm = min(log_n, 16)
    x = rng.integers(0, 1 << 62, size=(1 << s_ntt, 4), dtype=np.uint64)
    cpu.ntt(cid, 1, 10, x[:1024])  # warm OpenMP
    t0 = time.perf_counter()
    cpu.ntt(cid, 1, s_ntt, x)
    t_ntt_n = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu.ntt(cid, 2, s_ntt + 2, x)
    t_ntt_4n = time.perf_counter() - t0
    srs = cpu.srs_powers(cid, 0x7A5C0DE, 1 << 10)
    bases = np.tile(srs, ((1 << s_msm) >> 10, 1))
    sc = rng.integers(0, 1 << 62, size=(1 << s_msm, 4), dtype=np.uint64)
    t0 = time.perf_counter()
    cpu.msm_g1(cid, bases, sc, threads=cores)
    t_msm = time.perf_counter() - t0
    # scale to the benchmark size by the reference-equivalent operation counts
    bf = lambda k: (1 << k) // 2 * k  # noqa: E731
    t_proof = (17 * t_ntt_n * bf(log_n) / bf(s_ntt) + 14 * t_ntt_4n * bf(log_n + 2) / bf(s_ntt + 2)
               + 29 * t_msm * ark_adds(1 << log_n, bits) / ark_adds(1 << s_msm, bits))
    return {
        "value": 1.0 / t_proof, "unit": "proofs/s", "cores": cores, "kind": "port",
        "sample": f"oracle/ark_cpu.cpp (OpenMP): ifft 2^{s_ntt} {t_ntt_n:.3f}s, coset_fft 2^{s_ntt + 2} {t_ntt_4n:.3f}s, "
                  f"MSM 2^{s_msm} {t_msm:.3f}s (threads over windows as ark/rayon); scaled to n=2^{log_n} by butterfly / "
                  f"G1-add counts x the 17+14 NTT, 29 MSM per-proof schedule",
        "msm_adds_per_s": ark_adds(1 << s_msm, bits) / t_msm,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381", choices=["bls12_381", "bn254"],
                    help="bn254 = BASELINE.json's second-curve config (same kernels, 4-limb base field); the headline is bls12_381")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precompute", action="store_true", help="per-window MSM path (no window-multiples table)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-card rehearsals)")
    ap.add_argument("--mode", default="auto", choices=["auto", "replica", "shard"],
                    help="N > 1: 'replica' (default) = one proof stream per GPU, value = total proofs/s (weak scaling) followed by an "
                         "untimed-for-value sharded leg; 'shard' = every MSM point-sharded over the ranks (strong scaling)")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent proof streams per GPU (threads with their own zk_ctx + HIP stream); the K steps are shared out")
    ap.add_argument("--streams-leg", type=int, default=4,
                    help="N = 1: after the timed region, also report the throughput with this many concurrent proof streams (0/1 = skip)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="replica mode: skip the extra sharded-MSM leg")
    ap.add_argument("--dedup", action="store_true",
                    help="NOT the headline workload: commitments cached by polynomial label (SURVEY.md 8f N3), 17 MSMs per proof instead of 29")
    ap.add_argument("--grand-products", action="store_true",
                    help="also build the z / z2 evaluation vectors on the device (SURVEY.md 8f N2) inside each step")
    ap.add_argument("--fuse-round5", action="store_true",
                    help="the 16 MSMs of prover.rs:579-618 as ONE batch instead of the reference's four calls (needs those calls merged in the caller)")
    ap.add_argument("--quotient", action="store_true",
                    help="also compute the 4n quotient evaluations on the device (SURVEY.md 8f N1) inside each step")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no in-library HIP-event scopes in the timed region (roofline fields empty)")
    ap.add_argument("--check", action="store_true", help="print a digest of the 29 commitments (cross-rank / cross-N comparison)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if world > 1:
        torch.cuda.set_device(local_rank % n_dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % n_dev))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)

    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import ProofSchedule

    dev = torch.cuda.current_device()
    ctx = zk.Context(dev)
    ctx.use_torch_stream()
    cv = zk.get_curve(args.curve)
    sbits = cv.r.bit_length()
    log_n = args.log_n
    n = 1 << log_n
    steps = args.steps
    mode = args.mode
    if mode == "auto":
        mode = "replica" if world > 1 else "single"
    if world == 1:
        mode = "single"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def digest(points):
        import hashlib
        return hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in points)).hexdigest()

    def timed_region(sharded: bool, n_streams: int = 1, steps: int = steps):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.
        --streams S > 1 (replicas / single GPU only): the K steps are dealt round-robin to S concurrent proof
        streams (one thread + zk_ctx + HIP stream each) on this rank's GPU."""
        import threading
        lo, hi = (rank * n // world, (rank + 1) * n // world) if sharded else (0, n)
        S = 1 if sharded else max(1, min(n_streams, steps))
        lanes = []
        for i in range(S):
            cx = ctx if i == 0 else zk.Context(dev)
            st = torch.cuda.current_stream() if i == 0 else torch.cuda.Stream()
            with torch.cuda.stream(st):
                srs = build_srs(cx, cv, n, lo, hi, torch)
                ck = zk.CommitterKey(srs, cv, cx)
                del srs
                if not args.no_precompute:
                    ck.precompute()   # window-multiples table resident in HBM (one-time, like PC::trim)
                if sharded:
                    sched = ProofSchedule(log_n, cx, ck, cv, rank=rank, world=world, dist=dist, dedup=args.dedup, grand_products=args.grand_products, quotient=args.quotient, fuse_round5=args.fuse_round5)
                else:
                    sched = ProofSchedule(log_n, cx, ck, cv, dedup=args.dedup, grand_products=args.grand_products, quotient=args.quotient, fuse_round5=args.fuse_round5)
                pts = None
                for _ in range(args.warmup):
                    pts = sched.run_once()
            lanes.append({"ctx": cx, "stream": st, "ck": ck, "sched": sched, "pts": pts, "k": steps // S + (1 if i < steps % S else 0)})
        barrier()
        ctx.profile(not args.no_profile)
        ctx.profile_reset()
        if S == 1:
            t0 = time.perf_counter()
            for _ in range(steps):
                lanes[0]["sched"].run_once()
            barrier()
            dt = time.perf_counter() - t0
        else:
            gate = threading.Barrier(S + 1)
            errs = []

            def worker(ln):
                try:
                    with torch.cuda.stream(ln["stream"]):
                        gate.wait()
                        for _ in range(ln["k"]):
                            ln["sched"].run_once()
                except Exception as e:  # surfaced below: a failed lane must fail the run
                    errs.append(e)
                    gate.abort()

            th = [threading.Thread(target=worker, args=(ln,)) for ln in lanes]
            for t in th:
                t.start()
            gate.wait()
            t0 = time.perf_counter()
            for t in th:
                t.join()
            barrier()
            dt = time.perf_counter() - t0
            if errs:
                raise errs[0]
        ctx.profile(False)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        prof = {k: ctx.profile_get(k) for k in ("msm_accumulate", "ntt_pass", "msm_sort", "msm_reduce", "grand_product", "quotient")}
        digs = []
        if args.check:
            for ln in lanes:
                with torch.cuda.stream(ln["stream"]):
                    digs.append(digest(ln["pts"] if ln["pts"] is not None else ln["sched"].run_once()))
            if len(set(digs)) != 1:
                raise RuntimeError(f"proof streams disagree: {digs}")
        res = {"dt": dt, "prof": prof, "points_per_launch": hi - lo, "digest": digs[0] if digs else None,
               "ntt_bytes": lanes[0]["sched"].ntt_bytes(), "streams": S, "steps_profiled": lanes[0]["k"]}
        for ln in lanes:
            ln["ck"].close()
        lanes.clear()
        return res

    main_sharded = mode == "shard"
    r = timed_region(main_sharded, args.streams)
    dt = r["dt"]
    proofs = steps * (1 if (main_sharded or world == 1) else world)   # replicas: every rank proves K times
    value = proofs / dt
    acc_ms, acc_n = r["prof"]["msm_accumulate"]
    ntt_ms, ntt_n = r["prof"]["ntt_pass"]
    sort_ms, _ = r["prof"]["msm_sort"]
    red_ms, _ = r["prof"]["msm_reduce"]
    alg_bytes = (32.0 + 16.0 * cv.fq_limbs) * r["points_per_launch"]   # 32 B scalar + packed affine base (96 B BLS12-381), once
    avg_s = (acc_ms / max(acc_n, 1)) * 1e-3
    achieved = alg_bytes / avg_s / 1e9 if acc_n else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_msm_accumulate.json")
    if os.path.exists(pmc_path) and not main_sharded and log_n == 20 and cv.curve_id == 0:
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    msm_total_s = (acc_ms + sort_ms + red_ms) * 1e-3
    valu = None
    if acc_n and not args.no_precompute and cv.curve_id == 0:
        madds = 16.0 * r["points_per_launch"]               # one mixed addition per (16-bit window, point)
        mac_s = madds * MADD_MADS / avg_s
        peak_mac_s = LANES * CLOCK_HZ / MAD_CYCLES_FULL
        bound2 = LANES * CLOCK_HZ / MADD_CYCLES_2WAVES
        valu = {"mixed_adds_per_s": madds / avg_s, "u32_mac_per_s": mac_s, "peak_u32_mac_per_s": peak_mac_s, "frac_of_mad_peak": mac_s / peak_mac_s,
                "issue_bound_mixed_adds_per_s_at_2_waves_per_simd": bound2, "frac_of_issue_bound": madds / avg_s / bound2}
    kp = r["steps_profiled"]            # proofs seen by the profiled zk_ctx (= K unless --streams > 1)
    S = r["streams"]
    if world == 1:
        par = "1 GPU"
    elif main_sharded:
        par = f"MSM point-sharded over {world} GPUs + RCCL all-gather of partials; NTT replicated"
    else:
        par = f"{world} replicas (one whole proof stream per GPU, no data-path collective)"
    line = {
        "metric": "proofs/sec at 2^20 constraints (BLS12-381, KZG10); MSM G1-adds/s",
        "value": value, "unit": "proofs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if main_sharded else "weak",
        "vs_baseline": None, "dtype": "u32 limbs (256/384-bit Montgomery integers)", "data": "synthetic",
        "config": {"workload": f"per-proof hot path of Prover::prove at n=2^{log_n}: 13 ifft(n)+4 fft(n)+13 coset_fft(4n)+"
                               f"1 coset_ifft(4n)+29 KZG commits (MSM ~n), {cv.name}, SRS+inputs HBM-resident",
                   "log_n": log_n, "curve": cv.name, "parallelism": par},
        "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_launch_ms": avg_s * 1e3, "launches": int(acc_n), "alg_bytes_per_launch": alg_bytes,
                     "valu": valu},
        # rank 0's kernels: G1 additions the reference's Pippenger would have issued / time in the MSM kernels
        "msm_g1_adds_per_s": ((29 * kp * ark_adds(n, sbits)) / msm_total_s * (1 if (main_sharded or world == 1) else world)
                              if (msm_total_s and S == 1) else None),
        "msm_ms_per_proof": msm_total_s / kp * 1e3,
        "ntt_GBps": (r["ntt_bytes"] * kp) / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None,
        "ntt_ms_per_proof": ntt_ms / kp,
    }
    if args.grand_products:
        gp_ms, gp_n = r["prof"]["grand_product"]
        line["config"]["workload"] += " + z and z2 grand products on device"
        line["grand_product_ms_per_proof"] = gp_ms / kp
    if args.fuse_round5:
        line["config"]["workload"] += "; round 5's four PC calls merged into one batch"
    if args.quotient:
        line["config"]["workload"] += " + pointwise quotient on device"
        line["quotient_ms_per_proof"] = r["prof"]["quotient"][0] / kp
    if args.dedup:
        line["config"]["workload"] += " -- WITH commitment de-duplication: 17 MSMs computed, 12 served from the per-proof cache"
        line["msm_g1_adds_per_s"] = None
    if S > 1:
        line["config"]["parallelism"] += f", {S} concurrent proof streams per GPU (kernel times below overlap other streams' work)"
    if args.check:
        line["commitments_sha256"] = r["digest"]
    if world == 1 and S == 1 and args.streams_leg > 1:
        # second leg, not part of `value`: the GPU's throughput with several proofs in flight (a proving service):
        # small kernels of one proof fill the registers/issue slots the 2-waves/SIMD accumulate of another leaves idle
        try:
            k2 = max(steps, 2 * args.streams_leg)
            r2 = timed_region(False, args.streams_leg, k2)
            line["concurrent_streams"] = {"streams": r2["streams"], "steps": k2, "proofs_per_s": k2 / r2["dt"],
                                          "ms_per_proof_aggregate": r2["dt"] / k2 * 1e3,
                                          "commitments_match": (r2["digest"] == r["digest"]) if args.check else None}
        except Exception as e:
            line["concurrent_streams"] = {"error": repr(e)}
    if world > 1 and mode == "replica" and not args.no_sharded_leg:
        # second leg, not part of `value`: the same proofs with every MSM point-sharded over the ranks
        # (one RCCL all-gather of Jacobian partials per prover round) -- single-proof latency
        try:
            rs = timed_region(True)
            sa_ms, sa_n = rs["prof"]["msm_accumulate"]
            line["msm_sharded"] = {
                "ms_per_proof": rs["dt"] / steps * 1e3, "proofs_per_s": steps / rs["dt"],
                "speedup_vs_one_gpu_replica": (dt / steps) / (rs["dt"] / steps),
                "collective": "RCCL all_gather of 3L-limb Jacobian partials, one per PC::commit / PC::open call (11 per proof)",
                "points_per_rank": rs["points_per_launch"], "accumulate_avg_launch_ms": sa_ms / max(sa_n, 1),
                "commitments_match_replicas": (rs["digest"] == r["digest"]) if args.check else None,
            }
        except Exception as e:  # the headline above stands on its own
            line["msm_sharded"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(log_n, cv.curve_id, sbits)
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                line["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
