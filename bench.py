#!/usr/bin/env python3
"""bench.py -- proofs/s of the NTT + MSM prover hot path on MI355X (BASELINE.json metric).

A "step" = one proof's hot path: the reference's exact per-proof schedule of 31 transforms and 29
KZG commitments (ark_plonk_amd/prover_schedule.py <- prover.rs:163-638) at n = 2^20 constraints,
BLS12-381 + KZG10, on synthetic polynomials with every input (SRS, evaluation vectors) already
resident in HBM.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run`, one process per GPU, before anything touches the GPU) and relays rank 0's
line; under an external torchrun it is a rank.  `value` for N > 1 is the throughput of N replicas -- whole
proofs are independent, so every rank runs the schedule K times with no data-path collective (SURVEY.md 8e
"whole proofs: replicas only", scaling "weak").  The same run then times extra legs, never part of `value`:
`msm_sharded` (this size) and `msm_sharded_n22` (BASELINE config 3, n = 2^22): one proof stream with every MSM
sharded by points over the ranks and combined by ONE RCCL all-gather per group of PC calls (`--exchange`: the jobs' virtual-window
sums, 32 KiB each, straight out of the last reduction kernel, or host Jacobians -- the default; NTTs replicated).  `--mode shard` makes the sharded form the headline instead (scaling "strong").  Every N > 1 line carries `ranks`: the
world, the backend, the (host, PCI address) of the card each rank drove, how many of those are distinct and an all-reduced sum of
ones; with backend nccl the run refuses to start when ranks share a card.  `--rehearse` walks the N > 1 control flow without a GPU.

N = 1 adds, also never part of `value` (wall seconds of every leg: `leg_s`):
  concurrent_streams  the same GPU with 4 proofs in flight (one thread + zk_ctx + HIP stream each, ONE shared SRS);
  blocking_calls      every one of the reference's eleven PC calls blocking, as an unchanged Prover::prove issues them;
  power               socket power, clock and the firmware's throttler residencies over the schedule (which limiter holds the clock);
  drop_in             the same schedule through the host-pointer calls a Rust shim binds (zk_ntt, zk_kzg_commit_batch,
                      zk_kzg_open on pageable buffers, SRS registered once): the unchanged-caller number, with the
                      PCIe bytes it moves; `with_residency_cache` = the library keeps what it produced, `concurrent_callers` = four
                      such callers (threads) at once, one's transfers under the others' kernels;
  dedup               the schedule with the library's content-addressed commitment cache;
  no_precompute       the per-window MSM path (no window table), for the table's cost/benefit;
  configs             BASELINE.json's other configurations through the same code path with default options: config 3's size on one
                      card (2^22), config 4 (BN254, 2^18), config 5's size point (2^25, deferred form) -- proofs/s, roofline fraction,
                      commitments digest, job-set memory, and at 2^22 / 2^18 a KZG identity check of one commitment by Python integers.
`--extra-legs all` adds three legs outside SURVEY.md section 8's hot path (tools/bench_extra_legs.py).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (msm_accumulate) with the
algorithmic bytes of SURVEY.md 8d (128 B per point) over its HIP-event-timed launches;
`cpu_baseline` times the oracle's CPU restatement of the ark 0.3 algorithms at the benchmark sizes.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# The kernel is integer-VALU bound (SURVEY.md 8d asks for the u32-MAC rate next to the HBM fraction).
# Instruction counts of the mixed-addition path of msm_accumulate (gfx950 ISA of this build, BLS12-381:
# 6 products + 2 squares + 1 double product of 13 signed 30-bit limbs, csrc/fields.cuh) and the measured issue
# costs of profiles/r01/r01_ubench_valu.txt.
MADD_MADS = 3055            # v_mad_i64_i32 per mixed addition
MADD_OTHER_VALU = 1299      # 64-bit shift / add, mul_lo, digit and carry-step instructions around them
CLOCK_HZ = 2.4e9            # MI355X max engine clock (MI355X_MICROARCH.md chip table)
LANES = 256 * 4 * 64        # CUs x SIMDs x lanes
MAD_CYCLES_FULL = 3.99      # cycles per wave-instruction per SIMD at >= 4 waves/SIMD
# issue cycles of one mixed addition at the kernel's 2 waves/SIMD (215 VGPRs): mad 4.77, mul_lo 4.70, 64-bit shift/add 4.45, rest 2.64
MADD_CYCLES_2WAVES = 3055 * 4.77 + 117 * 4.70 + 492 * 4.45 + 690 * 2.64
# measured on MI355X (tools/experiments/clock_probe.hip, profiles/r02/r02_clock_probe.txt): what a stream of nothing but independent
# v_mad_u64_u32 reaches, by HIP events, at 2 and at 8 waves per SIMD -- the nominal 16 lanes x 1024 SIMDs x 2.4 GHz is 3.93e13
PURE_MAD_RATE_2_WAVES = 2.66e13
PURE_MAD_RATE_8_WAVES = 3.41e13
CURVE_TITLE = {"bls12_381": "BLS12-381", "bn254": "BN254"}
TAU = 0x7A5C0DE             # the synthetic SRS is P_i = TAU^i G: a known tau lets a commitment be checked as (sum p_i tau^i) G
# BASELINE.json configs 3, 4 and 5 as the default run's `configs` leg times them: curve:log_n:steps
DEFAULT_CONFIGS = "bls12_381:22:2,bn254:18:5,bls12_381:25:1"


def ark_adds(n: int, bits: int = 255) -> int:
    """reference-equivalent G1 additions of one MSM: W*N + 2*W*(2^c - 1)."""
    if n < 32:
        c = 3
    else:
        c = (n - 1).bit_length() * 69 // 100 + 2
    w = -(-bits // c)
    return w * n + 2 * w * ((1 << c) - 1)


def build_srs(ctx, cv, n, lo, hi, torch):
    """Synthetic KZG SRS slice P_i = tau^i G for i in [lo, hi), generated on the GPU: the Montgomery powers by doubling
    (pw[k:2k] = pw[0:k] * tau^k, zk_fr_mul_dev), into_repr, then the library's fixed-base utility."""
    from ark_plonk_amd import _lib, curves
    L = _lib.lib()
    m = hi - lo
    ctx.use_torch_stream()
    pw = torch.empty((m, 4), dtype=torch.int64, device="cuda")
    pw[0] = torch.from_numpy(curves.fr_to_mont(cv, [pow(TAU, lo, cv.r)]).view(np.int64))[0].cuda()
    k = 1
    while k < m:
        j = min(k, m - k)
        step = torch.from_numpy(curves.fr_to_mont(cv, [pow(TAU, k, cv.r)]).view(np.int64)).cuda().expand(j, 4).contiguous()
        _lib.check(L.zk_fr_mul_dev(ctx.handle, cv.curve_id, pw.data_ptr(), step.data_ptr(), j, pw[k:].data_ptr()), "zk_fr_mul_dev")
        k *= 2
    _lib.check(L.zk_fr_from_mont_dev(ctx.handle, cv.curve_id, pw.data_ptr(), m, pw.data_ptr()), "zk_fr_from_mont_dev")
    out = torch.empty((m, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
    _lib.check(L.zk_g1_fixed_base_batch_dev(ctx.handle, cv.curve_id, pw.data_ptr(), m, out.data_ptr()))
    torch.cuda.synchronize()
    return out


def kzg_identity_holds(cv, coeffs_mont, point) -> bool:
    """commit(p) over the SRS P_i = tau^i G is (sum_i p_i tau^i) G: the scalar side by Horner over Python integers (the raw limbs read as
    integers are R p_i: one multiplication by R^-1 at the end), the group side by affine double-and-add (curves.g1_mul) -- no library
    call and no oracle in either.  coeffs_mont: (n, 4) uint64 host array of Montgomery coefficients; point: G1Affine."""
    from ark_plonk_amd import curves
    acc = 0
    for v in reversed(curves.limbs_to_ints(coeffs_mont)):
        acc = (acc * TAU + v) % cv.r
    k = acc * pow(1 << 256, -1, cv.r) % cv.r
    exp = curves.g1_mul(cv, k)
    if exp is None:
        return bool(point.infinity)
    if point.infinity:
        return False
    return curves.fq_from_mont(cv, point.x.reshape(1, -1))[0] == exp[0] and curves.fq_from_mont(cv, point.y.reshape(1, -1))[0] == exp[1]


def cpu_baseline(log_n: int, cid: int = 0, bits: int = 255, budget_s: float = 45.0):
    """The oracle's CPU restatement (oracle/ark_cpu.cpp, OpenMP) timed AT the benchmark sizes on this box's host cores: one
    ifft(n), one coset_fft(4n) and one MSM(n), multiplied by the schedule's 17 / 14 / 29.  The MSM is timed in the
    ark/rayon shape (threads over windows: at most W = 17 busy) -- that is the `value` -- and in an all-cores shape
    (the points cut into per-thread MSMs, summed), SURVEY.md 8d.  Sizes shrink (and are scaled back by operation counts) only
    if a first small probe says the full-size sample would exceed `budget_s`."""
    from oracle import cpu
    cpu.build()
    cores = cpu.num_threads()
    rng = np.random.default_rng(1)
    # warm OpenMP: the first PARALLEL region creates the thread team (0.25 s for 128 threads on these hosts -- rounds 1-3 timed it as
    # part of the 2^20 ifft, whose own loops are parallel only from 2^12 elements on), so the warm-up must be large enough to have one
    cpu.ntt(cid, 1, 16, rng.integers(0, 1 << 62, size=(1 << 16, 4), dtype=np.uint64))
    cpu.ntt(cid, 2, 16, rng.integers(0, 1 << 62, size=(1 << 14, 4), dtype=np.uint64))
    # probe at 2^14 to bound the sample
    probe = 14
    srs_s = cpu.srs_powers(cid, 0x7A5C0DE, 1 << 10)
    sc_p = rng.integers(0, 1 << 62, size=(1 << probe, 4), dtype=np.uint64)
    t0 = time.perf_counter()
    cpu.msm_g1(cid, np.tile(srs_s, ((1 << probe) >> 10, 1)), sc_p, threads=cores)
    t_probe = time.perf_counter() - t0
    s_msm = log_n
    while s_msm > 14 and t_probe * ark_adds(1 << s_msm, bits) / ark_adds(1 << probe, bits) > budget_s / 3:
        s_msm -= 1
    s_ntt = log_n
    x = rng.integers(0, 1 << 62, size=(1 << s_ntt, 4), dtype=np.uint64)

    def best_of(reps, fn):
        """the CPU side gets its best run: the box's host cores are shared with other jobs of the pool (0.09 - 0.27 s for the same 2^20
        ifft from session to session), and a first run also pays the page faults of its output vector"""
        best, res = None, None
        for _ in range(reps):
            t0 = time.perf_counter()
            res = fn()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, res

    t_ntt_n, _ = best_of(3, lambda: cpu.ntt(cid, 1, s_ntt, x))
    t_ntt_4n, _ = best_of(3, lambda: cpu.ntt(cid, 2, s_ntt + 2, x))
    bases = np.tile(srs_s, ((1 << s_msm) >> 10, 1))
    sc = rng.integers(0, 1 << 62, size=(1 << s_msm, 4), dtype=np.uint64)
    t_msm, (ref_xy, ref_inf) = best_of(2, lambda: cpu.msm_g1(cid, bases, sc, threads=cores))
    # all-cores shape: (window, point range) tasks, each with its own buckets, so that every core of the host is busy -- not what
    # ark 0.3 does (threads over the windows only), reported next to it
    t_all, parts = None, 0
    try:
        t0 = time.perf_counter()
        all_xy, all_inf = cpu.msm_g1_all_cores(cid, bases, sc, threads=cores)
        t_all = time.perf_counter() - t0
        if not (np.array_equal(all_xy, ref_xy) and all_inf == ref_inf):
            t_all = None
        c_w = cpu.window_size(1 << s_msm)
        parts = max(1, -(-cores // max(1, -(-bits // c_w))))
    except Exception:
        t_all = None
    scale = ark_adds(1 << log_n, bits) / ark_adds(1 << s_msm, bits)
    t_proof = 17 * t_ntt_n + 14 * t_ntt_4n + 29 * t_msm * scale
    out = {
        "value": 1.0 / t_proof, "unit": "proofs/s", "cores": cores, "kind": "port",
        "sample": f"oracle/ark_cpu.cpp (OpenMP, {cores} threads) at the benchmark sizes: ifft 2^{s_ntt} {t_ntt_n:.3f}s, coset_fft 2^{s_ntt + 2} "
                  f"{t_ntt_4n:.3f}s, MSM 2^{s_msm} {t_msm:.3f}s (threads over windows as ark/rayon"
                  + ("" if s_msm == log_n else f"; scaled x{scale:.2f} to 2^{log_n} by G1-add count") + "); best of 3 / 3 / 2 runs; x 17 / 14 / 29 per proof",
        "note": "a restatement of the ark 0.3 algorithms in portable C++ (4 / 6 x 64-bit CIOS Montgomery with unsigned __int128, no assembly): "
                "the transforms run the reference's butterflies cache-blocked (the same values, ~3 instead of log2 n passes over memory); "
                "the MSM keeps ark's own parallel shape (threads over the windows only -- at most 17 busy at 2^20 points), which is what "
                "bounds the reference's rayon prover too; arkworks built with its x86-64 assembly (`asm` feature) would run the field "
                "products roughly 1.5-2x faster than this port -- an estimate, the reference cannot be built here",
        "msm_adds_per_s": ark_adds(1 << s_msm, bits) / t_msm,
        "ntt_ms": {"ifft_n": t_ntt_n * 1e3, "coset_fft_4n": t_ntt_4n * 1e3}, "msm_ms": t_msm * 1e3 * scale,
    }
    if t_all:
        t_proof_all = 17 * t_ntt_n + 14 * t_ntt_4n + 29 * t_all * scale
        out["all_cores"] = {"value": 1.0 / t_proof_all, "msm_ms": t_all * 1e3 * scale, "msm_adds_per_s": ark_adds(1 << s_msm, bits) / t_all,
                            "shape": f"MSM cut into (window, point range) tasks, {parts} ranges per window, own buckets per task, every core busy "
                                     "(not ark's shape); result equal to the ark-shaped run"}
    return out


from tools.power_sampler import FirmwareThrottlers, PowerSampler  # noqa: E402  (the `power` leg: hwmon + firmware throttler residencies)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_ranks: int) -> int:
    """Start `n_ranks` ranks of this script (one per GPU) and relay their output.  Runs before any import that could
    touch the GPU; the parent never initialises HIP."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---- N > 1: what the collective library really saw ------------------------------------------------------------------------------
def device_identity(torch, dev_index: int) -> str:
    """PCI address (+ UUID where torch exposes it) of the HIP device this rank drives: two ranks on one card have equal strings."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
    except Exception as e:       # a rehearsal on a host without a GPU
        return f"none:{e.__class__.__name__}"
    pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    uuid = getattr(pr, "uuid", None)
    return f"{pci}/{uuid}" if uuid is not None else pci


def ranks_handshake(dist, torch, backend: str, world: int, rank: int, local_rank: int, dev_index, tensor_device=None):
    """Run once after init_process_group: every rank's (host, device identity, local rank, pid) all-gathered, and an all_reduce(SUM) of
    a 1 per rank through the data path's own backend (on the rank's GPU for RCCL).  The `ranks` object of the line: evidence that N
    processes drove N different cards, or the reason the run refused to start (backend nccl and fewer distinct devices than ranks)."""
    ident = "cpu" if dev_index is None else device_identity(torch, dev_index)
    me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "device": ident, "pid": os.getpid()}
    seen = [None] * world
    dist.all_gather_object(seen, me)
    if tensor_device is None:
        tensor_device = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")
    one = torch.ones(1, dtype=torch.int64, device=tensor_device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    seen.sort(key=lambda d: d["rank"])
    distinct = len({(d["host"], d["device"]) for d in seen})
    return {"world": world, "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "distinct_devices": distinct,
            "shared_card": distinct < world, "sum_check": int(one.item()), "hosts": len({d["host"] for d in seen}),
            "devices": [f"{d['rank']}:{d['host']}:{d['device']}:local{d['local_rank']}" for d in seen]}


# rank 0 of an N > 1 run: the line as far as it is known.  `value` exists as soon as the headline region has been timed; should the job be
# torn down during one of the EXTRA legs (a rank lost inside a collective: torchrun SIGTERMs the others), a thread that owns SIGTERM prints
# the line with what it has -- the main thread may be stuck inside a collective and never run a Python-level handler.
_LINE = {"line": None, "printed": False, "leg": None}


def emit_line_so_far():
    """rank 0: print the headline line if it exists and has not been printed (a job ending early inside an extra leg)."""
    if _LINE["line"] is not None and not _LINE["printed"]:
        line = dict(_LINE["line"])
        line["aborted_in_leg"] = _LINE["leg"]
        line["aborted_why"] = _LINE.get("why") or "the job was torn down (SIGTERM / an exception) during the leg"
        _LINE["printed"] = True
        print(json.dumps(line), flush=True)


def install_sigterm_line_printer():
    import signal
    import threading
    signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})          # inherited by every thread created from here on

    def waiter():
        signal.sigwait({signal.SIGTERM})
        emit_line_so_far()
        os._exit(143)

    threading.Thread(target=waiter, daemon=True).start()


class LegDeadline:
    """rank 0 of an N > 1 run, around every EXTRA leg: ranks that are alive but out of step inside a collective kill nobody, so torchrun
    tears nothing down, and the backend's own timeout ends in abort() -- which no handler survives.  `seconds` (below --dist-timeout)
    after the leg began, the headline line that already exists is printed with `aborted_in_leg` / `aborted_why` and rank 0 exits
    non-zero, which makes torchrun end the other ranks.  Other ranks and N = 1: nothing."""

    def __init__(self, seconds: float, active: bool):
        self.seconds, self.active, self.timer = seconds, active and seconds > 0, None

    def _fire(self):
        _LINE["why"] = f"the leg did not return within {self.seconds:.0f} s (--leg-timeout)"
        emit_line_so_far()
        os._exit(42)

    def __enter__(self):
        if self.active:
            import threading
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *a):
        if self.timer is not None:
            self.timer.cancel()
        return False


REHEARSAL_GROUPS = (4, 3, 2, 4, 16)        # jobs per group of PC calls of one proof (prover.rs:213 | 289-317 | 361-389 | 459-469 | 579-618)
REHEARSAL_WORDS = 2 * 64 * 32              # int64 words of one job's window sums at the default geometry (BLS12-381: 128 points x 256 B)


def rehearse(args, world: int, rank: int, local_rank: int) -> int:
    """bench.py's N > 1 control flow with no GPU and no compute (see --rehearse)."""
    import torch
    import torch.distributed as dist
    from datetime import timedelta
    if args.backend == "nccl":
        print("bench.py --rehearse is a CPU rehearsal: use --backend gloo", file=sys.stderr)
        return 2
    if world > 1:
        if rank == 0:
            install_sigterm_line_printer()        # before the backend starts its threads: they must inherit the blocked mask
        dist.init_process_group(args.backend, timeout=timedelta(seconds=args.dist_timeout))
        info = ranks_handshake(dist, torch, args.backend, world, rank, local_rank, None)
    else:
        info = None

    def barrier():
        if world > 1:
            dist.barrier()

    def one_proof(step: int):
        for gi, jobs in enumerate(REHEARSAL_GROUPS):
            if rank == args.fault_rank and gi == 2 and step == (1 if args.fault_at == "step" else 3):
                if args.fault_at == "hang":
                    time.sleep(10 ** 6)                        # alive, but never reaches the next collective
                os._exit(41)                                   # a rank lost between two collectives of a proof
            mine = torch.full((jobs * REHEARSAL_WORDS,), (rank + 1) * 1000 + gi, dtype=torch.int64)
            out = torch.empty((world, jobs * REHEARSAL_WORDS), dtype=torch.int64)
            if world > 1:
                dist.all_gather_into_tensor(out.view(-1), mine)
            else:
                out[0] = mine
            for r in range(world):                             # rank-major, as zk_g1_sum_winsums_dev reads it
                if not bool((out[r] == (r + 1) * 1000 + gi).all()):
                    raise RuntimeError(f"rank {rank}: row {r} of group {gi} is not rank {r}'s tensor")

    for _ in range(args.warmup):
        one_proof(0)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_proof(1 if k == 0 else 2)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        line = {"metric": "REHEARSAL of the N > 1 control flow: no GPU, no compute, not a measurement", "rehearsal": True, "value": None,
                "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64 tensors of the exchange's shapes", "data": "synthetic",
                "config": {"workload": "five all_gather_into_tensor per step of 4|3|2|4|16 x 32 KiB, contents checked on every rank"},
                "collectives_per_step": len(REHEARSAL_GROUPS), "bytes_per_rank_per_step": sum(REHEARSAL_GROUPS) * REHEARSAL_WORDS * 8, "ranks": info}
        _LINE["line"], _LINE["leg"] = line, "rehearsal_leg"
    # an "extra leg" after the timed region, as the real run has them: a job torn down in here still gets its headline line out
    with LegDeadline(args.leg_timeout, world > 1 and rank == 0):
        one_proof(3)
        barrier()
    if rank == 0:
        _LINE["leg"] = None
        _LINE["printed"] = True
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--curve", default="bls12_381", choices=["bls12_381", "bn254"],
                    help="bn254 = BASELINE.json's second-curve config (same kernels, 4-limb base field); the headline is bls12_381")
    ap.add_argument("--data", default="uniform", choices=["uniform", "benchcircuit"],
                    help="wire columns: uniform random, or BenchCircuit's periodic {6,7,-20,1} rows + 3 blinding rows (SURVEY.md 8d config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precompute", action="store_true", help="per-window MSM path (no window-multiples table)")
    ap.add_argument("--table-window", type=int, default=0, help="window c of the SRS table (0 = library default; 16..21), see zk_srs_precompute_ex")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-card rehearsals)")
    ap.add_argument("--mode", default="auto", choices=["auto", "replica", "shard"],
                    help="N > 1: 'replica' (default) = one proof stream per GPU, value = total proofs/s (weak scaling) followed by "
                         "untimed-for-value sharded legs; 'shard' = every MSM point-sharded over the ranks (strong scaling)")
    ap.add_argument("--shard-axis", default="points", choices=["points", "windows"],
                    help="sharded MSMs (N > 1): 'points' = rank g owns SRS[g n/G, (g+1) n/G) and that slice of every polynomial (SURVEY.md 8e's "
                         "preferred axis); 'windows' = rank g holds the whole SRS and the table rows of the windows g, g + G, ... "
                         "(BASELINE.json north_star's wording; zk_srs_precompute_rows)")
    ap.add_argument("--drop-in-callers", type=int, default=4, help="drop_in leg: host threads calling the host-pointer entry points at once")
    ap.add_argument("--exchange", default="host", choices=["winsums", "host"],
                    help="sharded MSMs, what the ranks all-gather per group of PC calls (default 'host': the faster of the two on ONE card with a "
                         "stand-in collective, profiles/r05/r05_sim_rank.txt -- by 1-2 %% over 'winsums', which removes a host round trip before and "
                         "after every collective that a one-card run cannot see; an N > 1 run times BOTH, legs msm_sharded and "
                         "msm_sharded_winsums): 'winsums' = every job's 2 VW virtual-window sums as "
                         "the last reduction kernel of the single-GPU path leaves them on the device (32 KiB per job), added element-wise by one "
                         "kernel, one host combine per job (zk_kzg_round_end_winsums_dev -> all_gather_into_tensor -> zk_g1_sum_winsums_dev); "
                         "'host' = round 3's host Jacobian partials (D2H, host combine, H2D, all_gather, D2H)")
    ap.add_argument("--host-partials", action="store_true", help="= --exchange host")
    ap.add_argument("--dist-timeout", type=float, default=300.0,
                    help="N > 1: timeout (s) of the process group: a rank lost inside a collective takes the job down after this long instead of "
                         "hanging it (every collective of this program completes in milliseconds)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 control flow WITHOUT the GPU and without any compute: rendezvous, the `ranks` handshake, barriers, the max-over-ranks "
                         "timing and the five all-gathers of a sharded proof with tensors of the real shapes (4|3|2|4|16 jobs x 32 KiB) whose "
                         "contents every rank checks.  Prints a line with \"rehearsal\": true and value null -- never a measurement.  Runs on "
                         "CPU-only hosts (tests/test_distributed.py: 8 gloo ranks, and one of them killed)")
    ap.add_argument("--fault-rank", type=int, default=-1,
                    help="test hook: this rank dies (os._exit(41), no clean-up) in the middle of the run -- --rehearse: inside the first timed step, "
                         "between two collectives (--fault-at step) or inside the extra leg that follows the timed region (--fault-at leg); "
                         "otherwise inside the msm_sharded leg.  The job must end non-zero within --dist-timeout, not hang; killed inside a leg, "
                         "rank 0 still prints the headline line it already has, with `aborted_in_leg`")
    ap.add_argument("--fault-at", default="step", choices=["step", "leg", "hang"],
                    help="see --fault-rank; 'hang' (--rehearse): the rank stays alive inside the extra leg and never reaches its next collective")
    ap.add_argument("--leg-timeout", type=float, default=150.0,
                    help="N > 1, rank 0: an extra leg (msm_sharded ...) that has not returned after this many seconds no longer holds the headline "
                         "line back -- it is printed with `aborted_in_leg` and the job ends non-zero.  Keep it below --dist-timeout (the backend's "
                         "own timeout aborts the process); a leg takes seconds")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="a tuning option of the library (zk_ctx_set_option: msm_merge, pre_vw, pre_logg, chunk_l, long_rounds, combine_sg, pre_max_log_n), "
                         "set on every zk_ctx of the run and recorded in config.options; the library reads no environment variable")
    ap.add_argument("--streams", type=int, default=1,
                    help="concurrent proof streams per GPU (threads with their own zk_ctx + HIP stream); the K steps are shared out")
    ap.add_argument("--streams-leg", type=int, default=4,
                    help="N = 1: after the timed region, also report the throughput with this many concurrent proof streams (0/1 = skip)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="replica mode: skip the extra sharded-MSM legs")
    ap.add_argument("--sharded-n22-leg", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: also time BASELINE config 3 (n = 2^22, MSMs point-sharded); auto = on for the default workload")
    ap.add_argument("--configs", default="auto", metavar="SPEC",
                    help="the `configs` leg (N = 1): BASELINE.json's other configurations timed by this very run, never part of `value` -- "
                         f"'auto' = {DEFAULT_CONFIGS} (curve:log_n:steps; config 3's size on one card, config 4, config 5's size point in the "
                         "deferred form) when the headline is the default one; 'off'; or a list of the same form")
    ap.add_argument("--extra-legs", default="auto", choices=["auto", "on", "off", "all"],
                    help="N = 1: blocking_calls, power, drop_in, dedup and no_precompute legs; auto = on when log-n >= 16; 'all' adds the legs "
                         "outside SURVEY.md section 8's hot path (data_benchcircuit, with_device_glue, full_proof)")
    ap.add_argument("--dedup", action="store_true",
                    help="NOT the headline workload: the library's commitment cache on (SURVEY.md 8f N3), 17 MSMs per proof instead of 29")
    ap.add_argument("--grand-products", action="store_true",
                    help="also build the z / z2 evaluation vectors on the device (SURVEY.md 8f N2) inside each step")
    ap.add_argument("--fuse-round5", action="store_true",
                    help="the 16 MSMs of prover.rs:579-618 as ONE batch instead of the reference's four calls (needs those calls merged in the caller)")
    ap.add_argument("--quotient", action="store_true",
                    help="also compute the 4n quotient evaluations on the device (SURVEY.md 8f N1) inside each step")
    ap.add_argument("--no-ntt-batch", action="store_true",
                    help="issue every transform as its own zk_ntt_dev call (a patched ark-poly sees one fft at a time) instead of batching the "
                         "independent adjacent ones (4 wire iffts | h1,h2 | 4 sigma ffts | 12 coset ffts) into one launch per pass")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no in-library HIP-event scopes in the timed region (roofline fields empty)")
    ap.add_argument("--no-check", dest="check", action="store_false",
                    help="skip the digests: by default every leg re-proves proof 0 once OUTSIDE its timed region and the line reports "
                         "whether the 29 commitments of each leg equal the headline's (commitments_match* fields)")
    ap.add_argument("--check", dest="check", action="store_true", help="(default) see --no-check")
    ap.set_defaults(check=True)
    ap.add_argument("--no-hoist", action="store_true",
                    help="launch every transform where the reference calls it, instead of queueing those that do not depend on a round's "
                         "challenge behind that round's reductions (zk_kzg_round_reduce) before the host waits")
    ap.add_argument("--block-every-call", action="store_true",
                    help="the headline schedule blocks in each of its eleven PC calls (what an unchanged Prover::prove does) instead of "
                         "collecting f|h1|h2, z|z2 and the last round's four calls once per group (zk_kzg_round_begin_dev ... zk_kzg_round_end)")
    return ap.parse_args()


def main():
    t_start = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.host_partials:
        args.exchange = "host"
    if args.rehearse:
        sys.exit(rehearse(args, world, rank, local_rank))
    n_dev = torch.cuda.device_count()
    ranks_info = None
    if world > 1:
        # One process per GPU.  With RCCL a rank that sees no GPU at all is an error before the rendezvous (exit 2).  Fewer visible
        # devices than local ranks is NOT decided here -- a launcher may show every rank one card of its own -- but by the handshake
        # below: every rank sees the same gathered list of (host, PCI address), and with backend nccl fewer distinct cards than ranks
        # is exit 3 on all of them.  Folding ranks onto one card (local_rank % n_dev) is for the gloo rehearsals on a one-card box and
        # is reported in the line (`ranks.shared_card`).
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if args.backend == "nccl" and n_dev == 0:
            print(f"bench.py: {local_world} ranks on this host but no GPU visible: backend nccl needs one GPU per rank "
                  f"(rank {rank}); use --backend gloo --rehearse for a CPU rehearsal", file=sys.stderr)
            sys.exit(2)
        if rank == 0:
            install_sigterm_line_printer()        # before HIP / the backend start their threads: they must inherit the blocked mask
        dev_index = local_rank % max(n_dev, 1)
        torch.cuda.set_device(dev_index)
        from datetime import timedelta
        tmo = timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)
        ranks_info = ranks_handshake(dist, torch, args.backend, world, rank, local_rank, dev_index)
        if args.backend == "nccl" and (ranks_info["distinct_devices"] < world or ranks_info["sum_check"] != world):
            if rank == 0:
                print(f"bench.py: RCCL saw {ranks_info['distinct_devices']} distinct device(s) for {world} ranks (sum check "
                      f"{ranks_info['sum_check']}): {ranks_info['devices']}", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)

    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import ProofSchedule

    options = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.option}

    def new_ctx(d):
        c = zk.Context(d)
        for k, v in options.items():
            c.set_option(k, v)
        return c

    dev = torch.cuda.current_device()
    ctx = new_ctx(dev)
    ctx.use_torch_stream()
    if ranks_info is not None:
        # the ranks of a host share its cores: each ctx's host pool is sized min(15, cores / LOCAL_WORLD_SIZE - 1) (ark_plonk_amd/context.py)
        ranks_info["host_pool"] = {"workers_per_ctx": ctx.get_option("host_workers"), "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1),
                                   "host_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}
    cv = zk.get_curve(args.curve)
    sbits = cv.r.bit_length()
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

    def all_ok(ok: bool) -> bool:
        """every rank agrees on whether a leg may start / succeeded, so no rank is left alone in a collective"""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def digest(points):
        import hashlib
        return hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in points)).hexdigest()

    def timed_region(sharded: bool, n_streams: int = 1, steps: int = steps, log_n: int = args.log_n, precompute: bool = not args.no_precompute,
                     dedup=("abi" if args.dedup else False), warmup: int = args.warmup, glue: bool = False, defer_calls: bool = not args.block_every_call,
                     data: str = args.data, cv=cv, ctx=ctx, breakdown: bool = True, table_window: int = args.table_window, keep=None):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.
        n_streams S > 1 (replicas / single GPU only): the K steps are dealt round-robin to S concurrent proof
        streams (one thread + zk_ctx + HIP stream each) on this rank's GPU, all using ONE device-resident SRS.
        cv / ctx: another curve or zk_ctx than the headline's (the `configs` leg); breakdown=False: no extra proofs with every scope on;
        keep(sched, points): called with the schedule and the points of the digest proof before everything is freed."""
        import threading
        n = 1 << log_n
        by_windows = sharded and args.shard_axis == "windows" and world > 1
        lo, hi = (rank * n // world, (rank + 1) * n // world) if (sharded and not by_windows) else (0, n)
        S = 1 if sharded else max(1, min(n_streams, steps))
        srs = build_srs(ctx, cv, n, lo, hi, torch)
        ck0 = zk.CommitterKey(srs, cv, ctx)
        del srs
        if by_windows:
            ck0.precompute(table_window, rows=(rank, world))   # this rank's windows only: about 1/G of the table over the whole SRS
        elif precompute:
            ck0.precompute(table_window)   # window-multiples table resident in HBM (one-time, like PC::trim)
        lanes = []
        for i in range(S):
            cx = ctx if i == 0 else new_ctx(dev)
            st = torch.cuda.current_stream() if i == 0 else torch.cuda.Stream()
            ck = ck0 if i == 0 else ck0.with_ctx(cx)     # the SRS and its table belong to the device, not to a ctx
            with torch.cuda.stream(st):
                kw = dict(dedup=dedup, grand_products=args.grand_products or glue, quotient=args.quotient or glue, linearisation=glue, lookup_round2=glue, fuse_round5=args.fuse_round5,
                          data=data, ntt_batch=not args.no_ntt_batch, defer_calls=defer_calls, hoist=not args.no_hoist)
                if sharded:
                    sched = ProofSchedule(log_n, cx, ck, cv, rank=rank, world=world, dist=dist, shard_axis=args.shard_axis,
                                          exchange=args.exchange, **kw)
                else:
                    sched = ProofSchedule(log_n, cx, ck, cv, **kw)
                pts = None
                for _ in range(warmup):
                    pts = sched.run_once()
            lanes.append({"ctx": cx, "stream": st, "ck": ck, "sched": sched, "pts": pts, "k": steps // S + (1 if i < steps % S else 0)})
        barrier()
        # inside the timed region only the dominant kernel is bracketed by events (58 scopes per proof); the full per-kernel
        # breakdown (~200 scopes, ~2 ms per proof of instrumentation) is taken on extra proofs right after it
        ctx.profile(0 if args.no_profile else 2)
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
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        prof = {"msm_accumulate": ctx.profile_get("msm_accumulate"),
                # units of those launches: since round 4 ONE msm_accumulate_batch launch sums every MSM of a round (2 .. 16 jobs)
                "msm_accumulate_jobs": ctx.profile_get("msm_accumulate_jobs"), "msm_accumulate_points": ctx.profile_get("msm_accumulate_points")}
        msms_run = lanes[0]["sched"].msms_run        # of the last timed proof (the digest run below repeats proof 0)
        kb = 0
        if not args.no_profile and breakdown:
            kb = 2
            ctx.profile(1)
            ctx.profile_reset()
            for _ in range(kb):
                lanes[0]["sched"].run_once()
            barrier()
            ctx.profile(False)
            for k in ("ntt_pass", "msm_sort", "msm_reduce", "grand_product", "quotient", "poly_evaluate", "poly_lincomb", "lookup_query", "lookup_combine_split"):
                prof[k] = ctx.profile_get(k)
            prof["breakdown_accumulate"] = ctx.profile_get("msm_accumulate")
        else:
            for k in ("ntt_pass", "msm_sort", "msm_reduce", "grand_product", "quotient", "poly_evaluate", "poly_lincomb", "lookup_query", "lookup_combine_split", "breakdown_accumulate"):
                prof[k] = (0.0, 0)
        digs = []
        if args.check:
            for ln in lanes:
                with torch.cuda.stream(ln["stream"]):
                    pts0 = ln["sched"].run_once(proof_id=0)
                    digs.append(digest(pts0))
                    if keep is not None and ln is lanes[0]:
                        keep(ln["sched"], pts0)
            if len(set(digs)) != 1:
                raise RuntimeError(f"proof streams disagree: {digs}")
        res = {"dt": dt, "prof": prof, "points_per_launch": hi - lo, "digest": digs[0] if digs else None,
               "exchange": getattr(lanes[0]["sched"], "exchange", None) if sharded else None, "collectives": lanes[0]["sched"].collectives,
               "ntt_bytes": lanes[0]["sched"].ntt_bytes(), "streams": S, "steps_profiled": lanes[0]["k"], "msms_run": msms_run, "kb": kb,
               "windows": ck0.table_windows(), "window_bits": ck0.table_window_bits() if precompute else 0, "mem": ctx.round_mem_stats()}
        if dedup == "abi":
            for ln in lanes:
                ln["ctx"].set_commit_cache(False)
        for ln in lanes[1:]:
            ln["ctx"].close()
        ck0.close()
        lanes.clear()
        return res

    def drop_in_region(k: int, log_n: int = args.log_n):
        from types import SimpleNamespace
        from tools import bench_drop_in
        return bench_drop_in.drop_in_region(SimpleNamespace(args=args, zk=zk, torch=torch, ctx=ctx, cv=cv, build_srs=build_srs, digest=digest), k, log_n)

    main_sharded = mode == "shard"
    log_n = args.log_n
    n = 1 << log_n
    r = timed_region(main_sharded, args.streams)
    dt = r["dt"]
    proofs = steps * (1 if (main_sharded or world == 1) else world)   # replicas: every rank proves K times
    value = proofs / dt
    acc_ms, acc_n = r["prof"]["msm_accumulate"]
    ntt_ms, ntt_n = r["prof"]["ntt_pass"]
    sort_ms, _ = r["prof"]["msm_sort"]
    red_ms, _ = r["prof"]["msm_reduce"]
    # algorithmic bytes of the launches timed: 32 B scalar + packed affine base (96 B BLS12-381) per point, once.
    # A launch of the dominant kernel sums the jobs of one prover round (4 | 3 | 2 | 4 | 16 MSMs at the headline schedule), so the
    # units come from the library's own counters of the scope: points and jobs the timed launches processed.
    acc_jobs = r["prof"]["msm_accumulate_jobs"][1] or acc_n
    acc_points = r["prof"]["msm_accumulate_points"][1] or acc_n * r["points_per_launch"]
    launch_points = acc_points / max(acc_n, 1)                                     # average points per launch
    alg_bytes = (32.0 + 16.0 * cv.fq_limbs) * launch_points
    avg_s = (acc_ms / max(acc_n, 1)) * 1e-3
    achieved = alg_bytes / avg_s / 1e9 if acc_n else 0.0
    traffic, traffic_source = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_msm_accumulate.json")
    if os.path.exists(pmc_path) and not main_sharded and log_n == 20 and cv.curve_id == 0 and not args.no_precompute:
        try:
            pm = json.load(open(pmc_path))
            traffic = pm.get("hbm_bytes_per_launch")
            traffic_source = {"file": "profiles/pmc_msm_accumulate.json", "collected": pm.get("collected"), "commit": pm.get("commit"),
                              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command on an earlier box; a "
                                      "constant of the kernel build named in `commit`, not a counter of the run that printed this line"}
        except Exception:
            traffic = None
    kb = max(r["kb"], 1)                 # proofs of the breakdown pass (all scopes on, right after the timed region)
    bacc_ms, _ = r["prof"]["breakdown_accumulate"]
    msm_total_s = (bacc_ms + sort_ms + red_ms) * 1e-3
    valu = None
    W = r["windows"]
    if acc_n and W and cv.curve_id == 0:
        madds = float(W) * launch_points                      # one mixed addition per (window, point) on the table path
        mac_s = madds * MADD_MADS / avg_s
        peak_mac_s = LANES * CLOCK_HZ / MAD_CYCLES_FULL
        bound2 = LANES * CLOCK_HZ / MADD_CYCLES_2WAVES
        valu = {"mixed_adds_per_scalar": W, "mixed_adds_per_s": madds / avg_s, "u32_mac_per_s": mac_s, "peak_u32_mac_per_s": peak_mac_s,
                "frac_of_mad_peak": mac_s / peak_mac_s,
                "issue_bound_mixed_adds_per_s_at_2_waves_per_simd": bound2, "frac_of_issue_bound": madds / avg_s / bound2,
                "mads_vs_measured_pure_mad_stream_at_2_waves_per_simd": mac_s / PURE_MAD_RATE_2_WAVES,
                "mads_vs_measured_pure_mad_stream_at_8_waves_per_simd": mac_s / PURE_MAD_RATE_8_WAVES,
                "note": "the kernel issues 1299 other vector instructions per 3055 multiply-adds on top of this rate, and at two waves per SIMD a 64-bit "
                        "shift or add costs what a multiply-add costs (profiles/r01/r01_ubench_valu.txt): the count that matters is all of them "
                        "(profiles/r03/r03_notes.md)"}
    kp = r["steps_profiled"]            # proofs seen by the profiled zk_ctx (= K unless --streams > 1)
    S = r["streams"]
    if world == 1:
        par = "1 GPU"
    elif main_sharded:
        par = (f"MSM sharded by {args.shard_axis} over {world} GPUs + one all-gather per group of PC calls "
               f"(exchange form '{r['exchange']}'); NTT replicated")
    else:
        par = f"{world} replicas (one whole proof stream per GPU, no data-path collective)"
    line = {
        "metric": f"proofs/sec at 2^{log_n} constraints ({CURVE_TITLE[cv.name]}, KZG10); MSM G1-adds/s",
        "value": value, "unit": "proofs/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if main_sharded else "weak",
        "vs_baseline": None, "dtype": "u32 limbs (256/384-bit Montgomery integers)", "data": "synthetic",
        "config": {"workload": f"per-proof hot path of Prover::prove at n=2^{log_n}: 13 ifft(n)+4 fft(n)+13 coset_fft(4n)+"
                               f"1 coset_ifft(4n)+29 KZG commits (MSM ~n), {cv.name}, SRS+inputs HBM-resident"
                               + (", wire columns as BenchCircuit builds them (periodic {6,7,-20,1} rows + 3 blinding rows)" if args.data != "uniform" else ""),
                   "log_n": log_n, "curve": cv.name, "parallelism": par, "options": options or None, "msm_path": "per-window" if args.no_precompute else f"window table, {W} shared-bucket windows",
                   "pc_calls": ("the reference's eleven PC::commit / PC::open calls (4|1|1|1|1|1|4|7|1|7|1 polynomials), every call blocking"
                                if args.block_every_call or args.dedup else
                                "the reference's eleven PC::commit / PC::open calls (4|1|1|1|1|1|4|7|1|7|1 polynomials, 29 MSMs); calls whose inputs do not depend on each "
                                "other's results (f|h1|h2, z|z2, the last round's four: prover.rs:289-317,361-389,579-618) go through zk_kzg_round_begin_dev / "
                                "zk_kzg_open_begin_dev and are collected by one zk_kzg_round_end per group: 5 host waits per proof, identical points "
                                "(`blocking_calls` leg: every call blocking)"),
                   "ntt_calls": "one zk_ntt_dev per transform" if args.no_ntt_batch else
                                "adjacent independent transforms as zk_ntt_batch_dev (4 wire iffts | h1,h2 | 4 sigma ffts | 12 coset ffts), the rest single"
                                + ("" if (args.no_hoist or args.block_every_call or args.dedup) else
                                   "; transforms whose inputs do not depend on a round's challenge are queued behind that round's reductions before the "
                                   "host waits (zk_kzg_round_reduce): sigma ffts behind round 1, pi / L_1 iffts behind round 2, the twelve coset ffts "
                                   "behind round 3 -- same transforms, sizes and batches (--no-hoist: the reference's call order)")},
        "roofline": {"bound": "hbm", "kernel": "msm_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_note": "each of the W digits of a scalar gathers its own 128-B row of the window table instead of re-deriving "
                                     "2^(c w) P: HBM bytes traded for doublings, by design",
                     "avg_launch_ms": avg_s * 1e3, "launches": int(acc_n), "alg_bytes_per_launch": alg_bytes,
                     "msms_per_launch": acc_jobs / max(acc_n, 1), "points_per_launch": launch_points,
                     "ms_per_msm": acc_ms / max(acc_jobs, 1),
                     "launch_note": "one msm_accumulate_batch launch sums every MSM of a prover round (job = a block range); achieved = 128 B x the "
                                    "points the timed launches processed / their HIP-event time",
                     "valu": valu},
        # rank 0's kernels: G1 additions the reference's Pippenger would have issued / time in the MSM kernels
        "msm_g1_adds_per_s": ((29 * kb * ark_adds(n, sbits)) / msm_total_s * (1 if (main_sharded or world == 1) else world)
                              if (msm_total_s and not args.dedup) else None),
        "msm_ms_per_proof": msm_total_s / kb * 1e3,
        "msm_breakdown_ms_per_proof": {"accumulate": bacc_ms / kb, "sort": sort_ms / kb, "reduce": red_ms / kb},
        "ntt_GBps": (r["ntt_bytes"] * kb) / (ntt_ms * 1e-3) / 1e9 if ntt_ms else None,
        "ntt_ms_per_proof": ntt_ms / kb,
        "breakdown_note": f"msm_* / ntt_* are per-kernel-scope HIP-event times of {kb} extra proofs run right after the timed region with every scope on; "
                          "inside the timed region only msm_accumulate carries events (roofline)",
    }
    # the second kernel family north_star names: the NTT passes, against the same HBM roofline (2 * 32 * N bytes per transform)
    if ntt_ms:
        ntt_ach = (r["ntt_bytes"] * kb) / (ntt_ms * 1e-3) / 1e9
        ntt_traffic, ntt_src, ntt_issue = None, None, None
        ntt_pmc = os.path.join(ROOT, "profiles", "pmc_ntt.json")
        if os.path.exists(ntt_pmc) and not main_sharded and log_n == 20 and cv.curve_id == 0:
            try:
                pm = json.load(open(ntt_pmc))
                ntt_traffic = pm.get("hbm_bytes_per_proof")
                ntt_src = {"file": "profiles/pmc_ntt.json", "collected": pm.get("collected"), "commit": pm.get("commit")}
                ntt_issue = pm.get("issue")      # SQ counters of the same session: what really bounds the passes (vector issue)
            except Exception:
                ntt_traffic = None
        line["roofline_ntt"] = {"bound": "hbm", "kernel": "ntt_pass_mid<S> / ntt_pass_final<S>: every pass of the 31 transforms of a proof (17 of 2^%d, 14 of 2^%d)" % (log_n, log_n + 2),
                                "achieved": ntt_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ntt_ach / HBM_PEAK_GBS,
                                "alg_bytes_per_proof": r["ntt_bytes"], "ms_per_proof": ntt_ms / kb, "calls_timed": int(ntt_n),
                                "traffic": ntt_traffic, "traffic_unit": "HBM bytes per proof (all NTT passes)", "traffic_source": ntt_src,
                                "issue": ntt_issue,
                                "timing": f"HIP events around every zk_ntt(_batch)_dev call of {kb} proofs run right after the timed region (the scopes "
                                          "of the timed region itself carry msm_accumulate only)",
                                "note": "three passes per transform: the vector crosses HBM three times and every mid pass also reads a pass-boundary "
                                        "twiddle table of the vector's size; the passes are vector-issue bound (DESIGN.md 4.1), not HBM bound"}
    if args.grand_products:
        gp_ms, gp_n = r["prof"]["grand_product"]
        line["config"]["workload"] += " + z and z2 grand products on device"
        line["grand_product_ms_per_proof"] = gp_ms / kb
    if args.fuse_round5:
        line["config"]["workload"] += "; round 5's four PC calls merged into one batch"
    if args.quotient:
        line["config"]["workload"] += " + pointwise quotient on device"
        line["quotient_ms_per_proof"] = r["prof"]["quotient"][0] / kb
    if args.dedup:
        line["config"]["workload"] += f" -- WITH the commitment cache (zk_ctx_set_commit_cache): {r['msms_run']} MSMs computed per proof, the rest served from the cache"
    if S > 1:
        line["config"]["parallelism"] += f", {S} concurrent proof streams per GPU (kernel times below overlap other streams' work)"
    if args.check:
        line["commitments_sha256"] = r["digest"]
    if ranks_info is not None:
        # what the collective library saw: N processes, the card each one drove, and a SUM of ones through the data path's backend
        line["ranks"] = ranks_info
    _LINE["line"] = line          # N > 1: from here on a torn-down job still prints its headline (install_sigterm_line_printer)

    def acc_per_msm(rr):
        a_ms, a_n = rr["prof"]["msm_accumulate"]
        return a_ms / max(rr["prof"]["msm_accumulate_jobs"][1] or a_n, 1)

    line["leg_s"] = {"headline": round(time.perf_counter() - t_start, 2)}      # wall seconds of every leg (the headline's includes start-up)

    def leg(name, fn):
        """an extra leg never takes the headline down; every rank agrees on its outcome first"""
        _LINE["leg"] = name
        t_leg = time.perf_counter()
        try:
            _leg(name, fn)
        finally:
            line["leg_s"][name] = round(time.perf_counter() - t_leg, 2)

    def _leg(name, fn):
        with LegDeadline(args.leg_timeout, world > 1 and rank == 0):
            try:
                res = fn()
                ok = True
            except Exception as e:
                res, ok = {"error": repr(e)}, False
            if not all_ok(ok) and ok:
                res = {"error": "another rank failed this leg"}
        line[name] = res

    extra = args.extra_legs in ("on", "all") or (args.extra_legs == "auto" and log_n >= 16)
    extra_all = args.extra_legs == "all"       # legs outside SURVEY.md section 8's hot path
    if world == 1 and S == 1 and args.streams_leg > 1:
        # the GPU's throughput with several proofs in flight (a proving service): small kernels of one proof fill the
        # registers/issue slots the 2-waves/SIMD accumulate of another leaves idle
        def streams_leg():
            k2 = max(steps, 2 * args.streams_leg)
            r2 = timed_region(False, args.streams_leg, k2)
            return {"streams": r2["streams"], "steps": k2, "proofs_per_s": k2 / r2["dt"], "ms_per_proof_aggregate": r2["dt"] / k2 * 1e3,
                    "shared_srs": True, "commitments_match": (r2["digest"] == r["digest"]) if args.check else None}
        leg("concurrent_streams", streams_leg)
    if world == 1 and S == 1 and extra and not args.block_every_call and not args.dedup:
        def blocking_leg():
            k2 = max(2, min(steps, 5))
            rb = timed_region(False, 1, k2, warmup=1, defer_calls=False)
            return {"proofs_per_s": k2 / rb["dt"], "ms_per_proof": rb["dt"] / k2 * 1e3, "steps": k2, "accumulate_ms_per_msm": acc_per_msm(rb),
                    "what": "the same 29 MSMs with every one of the eleven PC calls blocking (zk_kzg_round_batch_dev per call), as an unchanged Prover::prove issues them",
                    "commitments_match": (rb["digest"] == r["digest"]) if args.check else None}
        leg("blocking_calls", blocking_leg)
    if world == 1 and S == 1 and extra:
        def power_leg():
            # what bounds the step: socket power and shader clock while the headline schedule runs (hwmon, sampled by a host thread;
            # a separate leg so that the sampler never runs inside the timed region of `value`)
            n_ = 1 << log_n
            ckq = zk.CommitterKey(build_srs(ctx, cv, n_, 0, n_, torch), cv, ctx)
            if not args.no_precompute:
                ckq.precompute(args.table_window)
            sch = ProofSchedule(log_n, ctx, ckq, cv, defer_calls=not args.block_every_call, hoist=not args.no_hoist, ntt_batch=not args.no_ntt_batch)
            k2 = max(12, min(4 * steps, 40))
            bdf = device_identity(torch, dev).split("/")[0]
            fw = FirmwareThrottlers(bdf)
            with PowerSampler(bdf=bdf) as ps:
                time.sleep(0.25)
                for _ in range(2):
                    sch.run_once()
                torch.cuda.synchronize()
                fw_a = fw.snapshot()
                t0 = time.perf_counter()
                for _ in range(k2):
                    sch.run_once()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fw_b = fw.snapshot()
                time.sleep(0.1)
            ckq.close()
            out = ps.summary(t0, t1)
            if out is None:
                return {"error": "no amdgpu hwmon files readable on this host"}
            thr = fw.between(fw_a, fw_b)
            out.update({"proofs": k2, "proofs_per_s_while_sampled": k2 / (t1 - t0),
                        "joules_per_proof": out["socket_power_w"]["mean"] * (t1 - t0) / k2,
                        "limiter": thr["limiter"] if thr else None, "throttlers": thr if thr else {"error": fw.error},
                        "what": "amdgpu hwmon of THIS card (matched by PCI address; power1_average|power1_input, freq1_input = sclk) every ~4 ms over a "
                                "run of the headline schedule, and the firmware's own throttler residencies over the same run: the socket power "
                                "limit (PPT) is what holds the clock below 2.4 GHz under msm_accumulate -- no thermal limiter is ever active "
                                "(profiles/r05/r05_notes.md)"})
            return out
        leg("power", power_leg)
    if world == 1 and S == 1 and extra and not (args.grand_products or args.quotient or args.fuse_round5 or args.data != "uniform"):
        def drop_in_leg():
            d = drop_in_region(max(2, min(steps, 3)))
            d["commitments_match_resident"] = (d.pop("digest") == r["digest"]) if args.check else None
            d["vs_resident"] = d["proofs_per_s"] / value
            return d
        leg("drop_in", drop_in_leg)
        if not args.dedup:
            def dedup_leg():
                k2 = max(2, min(steps, 5))
                r3 = timed_region(False, 1, k2, dedup="abi")
                return {"proofs_per_s": k2 / r3["dt"], "ms_per_proof": r3["dt"] / k2 * 1e3, "msms_computed_per_proof": r3["msms_run"],
                        "how": "zk_ctx_set_commit_cache(1): 256-bit device digest of every coefficient vector; prover.rs:569-607 re-commits 12 polynomials",
                        "commitments_match": (r3["digest"] == r["digest"]) if args.check else None}
            leg("dedup", dedup_leg)
    if world == 1 and S == 1 and extra_all and not (args.dedup or args.grand_products or args.quotient or args.fuse_round5 or args.data != "uniform"):
        # legs outside SURVEY.md section 8's hot path: tools/bench_extra_legs.py
        from types import SimpleNamespace
        from tools import bench_extra_legs as xl
        E = SimpleNamespace(args=args, zk=zk, torch=torch, ctx=ctx, cv=cv, log_n=log_n, steps=steps, value=value, r=r, dev=dev,
                            timed_region=timed_region, acc_per_msm=acc_per_msm, build_srs=build_srs, new_ctx=new_ctx)
        leg("data_benchcircuit", lambda: xl.data_benchcircuit(E))
        leg("with_device_glue", lambda: xl.with_device_glue(E))
        leg("full_proof", lambda: xl.full_proof(E))
    if world == 1 and S == 1 and extra and not (args.grand_products or args.quotient or args.fuse_round5 or args.data != "uniform"):
        if not args.no_precompute:
            def nopre_leg():
                k2 = max(2, min(steps, 3))
                r4 = timed_region(False, 1, k2, precompute=False, warmup=1)
                return {"proofs_per_s": k2 / r4["dt"], "ms_per_proof": r4["dt"] / k2 * 1e3, "accumulate_ms_per_msm": acc_per_msm(r4),
                        "path": "per-window buckets, host Horner; bases read once per window, no 128-B row gathers of a table",
                        "commitments_match": (r4["digest"] == r["digest"]) if args.check else None}
            leg("no_precompute", nopre_leg)
    # ---- BASELINE.json's other configurations, timed by this very run (never part of `value`): config 3's size on one card (2^22),
    # config 4 (BN254, 2^18) and config 5's size point (2^25) -- the reference's own harness sweeps its sizes in one invocation
    # (benches/plonk.rs:95-162).  Default options throughout: the deferred form of the schedule, the library's default table window.
    spec = args.configs
    if spec == "auto":
        spec = DEFAULT_CONFIGS if (world == 1 and S == 1 and log_n == 20 and cv.curve_id == 0 and args.extra_legs != "off"
                                   and not (args.no_precompute or args.dedup or args.block_every_call)) else "off"
    if world == 1 and S == 1 and spec != "off":
        def accumulate_roofline(rr, cvx):
            a_ms, a_n = rr["prof"]["msm_accumulate"]
            pts = rr["prof"]["msm_accumulate_points"][1] or a_n * rr["points_per_launch"]
            if not a_n or not a_ms:
                return None
            ach = (32.0 + 16.0 * cvx.fq_limbs) * pts / (a_ms * 1e-3) / 1e9
            return {"bound": "hbm", "kernel": "msm_accumulate", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "avg_launch_ms": a_ms / a_n, "launches": int(a_n), "traffic": None}

        def configs_leg():
            out, t_leg0 = [], time.perf_counter()
            for item in spec.split(","):
                cname, lg, k2 = item.split(":")
                lg, k2 = int(lg), int(k2)
                cvx = zk.get_curve(cname)
                t_c = time.perf_counter()
                row = {"config": f"{CURVE_TITLE[cvx.name]} n=2^{lg}", "log_n": lg, "curve": cvx.name, "steps": k2}
                cx = new_ctx(dev)                     # a ctx of its own: closing it returns the job buffer sets, plans and staging of this size
                cx.use_torch_stream()
                kept = {}

                def keep(sched, pts, kept=kept, lg=lg):
                    # one MSM of the digest proof for the identity check below: the first wire polynomial and its commitment
                    if lg <= 22:
                        kept["coef"] = sched.coef[0].cpu().numpy().view(np.uint64)
                        kept["point"] = pts[0]
                try:
                    torch.cuda.empty_cache()
                    rr = timed_region(False, 1, k2, log_n=lg, warmup=1, cv=cvx, ctx=cx, breakdown=False, table_window=0, keep=keep,
                                      defer_calls=True, dedup=False, data="uniform", precompute=True)
                    row.update({"proofs_per_s": k2 / rr["dt"], "ms_per_proof": rr["dt"] / k2 * 1e3,
                                "msm_path": f"window table, c = {rr['window_bits']}, {rr['windows']} shared-bucket windows; deferred rounds (5 host waits per proof)",
                                "roofline": accumulate_roofline(rr, cvx), "commitments_sha256": rr["digest"],
                                "job_sets_gib": round(rr["mem"]["set_bytes"] / 2**30, 2), "early_closes": rr["mem"]["early_closes"]})
                    if "coef" in kept:
                        row["kzg_identity_holds"] = kzg_identity_holds(cvx, kept["coef"], kept["point"])
                        row["kzg_identity_what"] = ("commit(w_l) of the digest proof == (sum_i p_i tau^i) G: Horner over Python integers and an "
                                                    "affine double-and-add, no library call in the check")
                except Exception as e:     # noqa: BLE001 -- a size that fails is reported, the others still run
                    row["error"] = repr(e)
                finally:
                    kept.clear()
                    cx.close()
                    torch.cuda.empty_cache()
                row["seconds"] = round(time.perf_counter() - t_c, 2)
                out.append(row)
            return {"what": "BASELINE.json configs 3 (its size, on one card), 4 and 5 (its size point: DESIGN.md 5) through the headline's code "
                            "path with default options; per config: warm-up 1, then `steps` timed proofs; commitments digest of proof 0",
                    "rows": out, "seconds": round(time.perf_counter() - t_leg0, 2)}
        leg("configs", configs_leg)
    if world > 1 and mode == "replica" and not args.no_sharded_leg:
        # the same proofs with every MSM point-sharded over the ranks (one RCCL all-gather of Jacobian partials per
        # prover round) -- single-proof latency
        def shard_leg(lg, exchange=None):
            def run():
                if rank == args.fault_rank:
                    os._exit(41)              # test hook: a rank lost inside an extra leg (tests/test_distributed.py)
                saved = args.exchange
                if exchange:
                    args.exchange = exchange
                try:
                    rs = timed_region(True, log_n=lg)
                finally:
                    args.exchange = saved
                d = {"log_n": lg, "ms_per_proof": rs["dt"] / steps * 1e3, "proofs_per_s": steps / rs["dt"],
                     "collective": "all_gather of the jobs' partials, one per group of PC calls (5 per proof; 11 with --block-every-call): "
                                   + {"host": "3L-limb host Jacobian partials (--exchange host)",
                                      "winsums": "the 2 VW virtual-window sums of every job (32 KiB) as the single-GPU path's last reduction kernel writes "
                                                 "them, added element-wise by zk_g1_sum_winsums_dev, combined per job on the host pool"}[rs["exchange"]],
                     "exchange": rs["exchange"], "collectives_per_proof": rs["collectives"] // max(steps + args.warmup + (1 if args.check else 0) + rs["kb"], 1),
                     "shard_axis": args.shard_axis,
                     "points_per_rank": rs["points_per_launch"], "accumulate_ms_per_msm": acc_per_msm(rs),
                     "commitments_sha256": rs["digest"]}
                if lg == log_n:
                    d["speedup_vs_one_gpu_replica"] = (dt / steps) / (rs["dt"] / steps)
                    d["commitments_match_replicas"] = (rs["digest"] == r["digest"]) if args.check else None
                elif args.check:
                    # the unsharded form of the same size, one proof per rank on its own GPU (2^22: 8 GiB of window table per rank)
                    ru = timed_region(False, steps=1, log_n=lg, warmup=0)
                    d["commitments_match_unsharded"] = ru["digest"] == rs["digest"]
                    d["unsharded_ms_per_proof"] = ru["dt"] * 1e3
                return d
            return run
        leg("msm_sharded", shard_leg(log_n))
        # the same with the other form of the exchange: which of the two wins on a real node is this run's to measure
        other = "winsums" if args.exchange != "winsums" else "host"
        leg("msm_sharded_" + other, shard_leg(log_n, other))
        n22 = args.sharded_n22_leg == "on" or (args.sharded_n22_leg == "auto" and log_n == 20 and cv.curve_id == 0)
        if n22 and log_n != 22:
            leg("msm_sharded_n22", shard_leg(22))
    _LINE["leg"] = None
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            t_cb = time.perf_counter()
            try:
                line["cpu_baseline"] = cpu_baseline(log_n, cv.curve_id, sbits)
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                line["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
            line["leg_s"]["cpu_baseline"] = round(time.perf_counter() - t_cb, 2)
        line["leg_s"]["total"] = round(time.perf_counter() - t_start, 2)
        if not _LINE["printed"]:
            _LINE["printed"] = True
            print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        # a peer lost inside an extra leg surfaces on rank 0 either as SIGTERM from the launcher (install_sigterm_line_printer) or as an
        # exception out of a collective: the headline that was already measured is printed either way, the exit code stays non-zero
        if int(os.environ.get("RANK", "0")) == 0:
            emit_line_so_far()
        raise
