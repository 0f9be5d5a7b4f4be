#!/usr/bin/env python3
"""bench.py -- proofs/s of the NTT + MSM prover hot path on MI355X (BASELINE.json metric).

A "step" = one proof's hot path: the reference's exact per-proof schedule of 31 transforms and 29
KZG commitments (ark_plonk_amd/prover_schedule.py <- prover.rs:163-638) at n = 2^20 constraints,
BLS12-381 + KZG10, on synthetic polynomials with every input (SRS, evaluation vectors) already
resident in HBM.  N > 1: one process per GPU (torchrun); each MSM is sharded by points over the
ranks and combined by an RCCL all-gather of Jacobian partials; NTTs are replicated per rank.

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


def cpu_baseline(log_n: int):
    """Bounded sample of the same workload on the host cores with the oracle's CPU restatement."""
    from oracle import cpu
    cpu.build()
    cores = cpu.num_threads()
    rng = np.random.default_rng(1)
    s_ntt = min(log_n, 18)
    s_msm = min(log_n, 16)
    x = rng.integers(0, 1 << 62, size=(1 << s_ntt, 4), dtype=np.uint64)
    cpu.ntt(0, 1, 10, x[:1024])  # warm OpenMP
    t0 = time.perf_counter()
    cpu.ntt(0, 1, s_ntt, x)
    t_ntt_n = time.perf_counter() - t0
    t0 = time.perf_counter()
    cpu.ntt(0, 2, s_ntt + 2, x)
    t_ntt_4n = time.perf_counter() - t0
    srs = cpu.srs_powers(0, 0x7A5C0DE, 1 << 10)
    bases = np.tile(srs, ((1 << s_msm) >> 10, 1))
    sc = rng.integers(0, 1 << 62, size=(1 << s_msm, 4), dtype=np.uint64)
    t0 = time.perf_counter()
    cpu.msm_g1(0, bases, sc, threads=cores)
    t_msm = time.perf_counter() - t0
    # scale to the benchmark size by the reference-equivalent operation counts
    bf = lambda k: (1 << k) // 2 * k  # noqa: E731
    t_proof = (17 * t_ntt_n * bf(log_n) / bf(s_ntt) + 14 * t_ntt_4n * bf(log_n + 2) / bf(s_ntt + 2)
               + 29 * t_msm * ark_adds(1 << log_n) / ark_adds(1 << s_msm))
    return {
        "value": 1.0 / t_proof, "unit": "proofs/s", "cores": cores, "kind": "port",
        "sample": f"oracle/ark_cpu.cpp (OpenMP): ifft 2^{s_ntt} {t_ntt_n:.3f}s, coset_fft 2^{s_ntt + 2} {t_ntt_4n:.3f}s, "
                  f"MSM 2^{s_msm} {t_msm:.3f}s (threads over windows as ark/rayon); scaled to n=2^{log_n} by butterfly / "
                  f"G1-add counts x the 17+14 NTT, 29 MSM per-proof schedule",
        "msm_adds_per_s": ark_adds(1 << s_msm) / t_msm,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precompute", action="store_true", help="per-window MSM path (no window-multiples table)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-card rehearsals)")
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
    cv = zk.get_curve("bls12_381")
    log_n = args.log_n
    n = 1 << log_n
    lo, hi = rank * n // world, (rank + 1) * n // world
    srs = build_srs(ctx, cv, n, lo, hi, torch)
    ck = zk.CommitterKey(srs, cv, ctx)
    del srs
    if not args.no_precompute:
        ck.precompute()   # window-multiples table resident in HBM (one-time, like PC::trim)
    sched = ProofSchedule(log_n, ctx, ck, cv, rank=rank, world=world, dist=dist if world > 1 else None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pts = None
    for _ in range(args.warmup):
        pts = sched.run_once()
    barrier()
    ctx.profile(not args.no_profile)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sched.run_once()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    acc_ms, acc_n = ctx.profile_get("msm_accumulate")
    ntt_ms, ntt_n = ctx.profile_get("ntt_pass")
    sort_ms, _ = ctx.profile_get("msm_sort")
    red_ms, _ = ctx.profile_get("msm_reduce")
    steps = args.steps
    value = steps / dt
    pts_per_launch = (hi - lo)
    alg_bytes = 128.0 * pts_per_launch                      # 32 B scalar + 96 B affine base, once
    avg_s = (acc_ms / max(acc_n, 1)) * 1e-3
    achieved = alg_bytes / avg_s / 1e9 if acc_n else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_msm_accumulate.json")
    if os.path.exists(pmc_path) and world == 1 and log_n == 20:
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    msm_total_s = (acc_ms + sort_ms + red_ms) * 1e-3
    line = {
        "metric": "proofs/sec at 2^20 constraints (BLS12-381, KZG10); MSM G1-adds/s",
        "value": value, "unit": "proofs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None, "dtype": "u32 limbs (256/384-bit Montgomery integers)", "data": "synthetic",
        "config": {"workload": f"per-proof hot path of Prover::prove at n=2^{log_n}: 13 ifft(n)+4 fft(n)+13 coset_fft(4n)+"
                               f"1 coset_ifft(4n)+29 KZG commits (MSM ~n), BLS12-381, SRS+inputs HBM-resident",
                   "log_n": log_n, "curve": "bls12_381",
                   "parallelism": "1 GPU" if world == 1 else f"MSM point-sharded over {world} GPUs + RCCL all-gather of partials; NTT replicated"},
        "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_launch_ms": avg_s * 1e3, "launches": int(acc_n), "alg_bytes_per_launch": alg_bytes},
        "msm_g1_adds_per_s": (29 * steps * ark_adds(n)) / msm_total_s if (world == 1 and msm_total_s) else None,
        "msm_ms_per_proof": msm_total_s / steps * 1e3,
        "ntt_GBps": (sched.ntt_bytes() * steps) / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None,
        "ntt_ms_per_proof": ntt_ms / steps,
    }
    if args.check:
        import hashlib
        if pts is None:
            pts = sched.run_once()
        h = hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in pts)).hexdigest()
        line["commitments_sha256"] = h
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(log_n)
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                line["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
