"""Per-rank compute time of the sharded schedule, measured on ONE GPU: runs rank 0 of a world
of W with a stand-in for torch.distributed whose all_gather returns W copies of the local tensor.
No communication is timed -- this shows where a rank's time goes (NTT replica, MSM shard, reductions) for both shard axes
(points: SRS[0, n/W) and that slice of every polynomial; windows: the whole SRS, the table rows 0, W, 2W, ...) and the two forms of the
exchange (winsums: the jobs' virtual-window sums, zk_kzg_round_end_winsums_dev + zk_g1_sum_winsums_dev; host: Jacobian partials through
the host).  Round 5 also timed a third form, one device-resident point per job: last of the three (profiles/r05/r05_sim_rank.txt), retired.
AN ESTIMATE, NOT A MEASUREMENT of N GPUs.

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

    def all_gather_into_tensor(self, out, t):
        out.view(self.world, -1).copy_(t.unsqueeze(0).expand(self.world, -1))


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    log_n = int(os.environ.get("LOG_N", "20"))
    n = 1 << log_n
    torch.cuda.set_device(0)
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    import statistics
    visits = int(os.environ.get("VISITS", "6"))
    per_visit = int(os.environ.get("STEPS", "2"))
    for w in worlds:
        for axis in ("points", "windows"):
            if w == 1 and axis == "windows":
                continue
            if axis == "points":
                srs = build_srs(ctx, cv, n, 0, n // w, torch)
                ck = zk.CommitterKey(srs, cv, ctx).precompute()
            else:
                srs = build_srs(ctx, cv, n, 0, n, torch)
                ck = zk.CommitterKey(srs, cv, ctx).precompute(rows=(0, w))
            del srs
            forms = ("-",) if w == 1 else ("winsums", "host")
            # the forms ALTERNATE visit by visit over one key and one set of inputs (a box drifts by several per cent with its power
            # state: configurations measured one after the other cannot resolve the ~2 % between the forms)
            scheds = {f: ProofSchedule(log_n, ctx, ck, cv, rank=0, world=w, dist=FakeDist(w) if w > 1 else None, shard_axis=axis,
                                       exchange=None if w == 1 else f) for f in forms}
            for f in forms:
                scheds[f].run_once()
            torch.cuda.synchronize()
            times = {f: [] for f in forms}
            for _ in range(visits):
                for f in forms:
                    scheds[f].run_once()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(per_visit):
                        scheds[f].run_once()
                    torch.cuda.synchronize()
                    times[f].append((time.perf_counter() - t0) / per_visit * 1e3)
            for f in forms:
                ctx.profile(True)
                ctx.profile_reset()
                for _ in range(2):
                    scheds[f].run_once()
                torch.cuda.synchronize()
                ctx.profile(False)
                parts = {k: ctx.profile_get(k)[0] / 2 for k in ("msm_accumulate", "msm_sort", "msm_reduce", "msm_sum_winsums", "ntt_pass", "kzg_open_prep")}
                t = times[f]
                print(f"world={w} axis={axis:7s} exchange={f:8s}: median {statistics.median(t):6.2f}  min {min(t):6.2f} ms/step over {visits} alternating visits x {per_visit}  | "
                      + "  ".join(f"{k}={v:.2f}" for k, v in parts.items()), flush=True)
            ck.close()
            del scheds


if __name__ == "__main__":
    main()
