"""Per-rank compute time of the point-sharded schedule, measured on ONE GPU: runs rank 0 of a world
of W with a stand-in for torch.distributed whose all_gather returns W copies of the local tensor.
No communication is timed -- this shows where a rank's time goes (NTT replica, MSM shard, reductions).

usage: python tools/sim_rank.py [W ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd.prover_schedule import ProofSchedule  # noqa: E402
from bench import build_srs  # noqa: E402


class FakeDist:
    def __init__(self, world):
        self.world = world

    def get_backend(self):
        return "nccl"

    def all_gather(self, outs, t):
        for o in outs:
            o.copy_(t)


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    log_n = int(os.environ.get("LOG_N", "20"))
    n = 1 << log_n
    torch.cuda.set_device(0)
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    for w in worlds:
        srs = build_srs(ctx, cv, n, 0, n // w, torch)
        ck = zk.CommitterKey(srs, cv, ctx).precompute()
        del srs
        sched = ProofSchedule(log_n, ctx, ck, cv, rank=0, world=w, dist=FakeDist(w) if w > 1 else None)
        sched.run_once()
        torch.cuda.synchronize()
        ctx.profile(True)
        ctx.profile_reset()
        steps = 3
        t0 = time.perf_counter()
        for _ in range(steps):
            sched.run_once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        ctx.profile(False)
        parts = {k: ctx.profile_get(k)[0] / steps for k in ("msm_accumulate", "msm_sort", "msm_reduce", "ntt_pass", "fr_convert", "kzg_open_prep")}
        print(f"world={w}: {dt:.2f} ms/step  " + "  ".join(f"{k}={v:.2f}" for k, v in parts.items()), flush=True)
        ck.close()
        del sched


if __name__ == "__main__":
    main()
