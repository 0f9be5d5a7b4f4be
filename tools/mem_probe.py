"""Device memory of one proof of the headline schedule (deferred form, the default) at the sizes given:
    python tools/mem_probe.py [log_n ...]          (default 20 22 24 25)
One JSON line per size: GiB used after the SRS table, after the schedule's own vectors, after one proof; what the library's job
buffer sets hold (zk_round_mem_stats) and how often the memory budget closed a round early.  DESIGN.md section 5's table."""
import gc
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import ark_plonk_amd as zk
from ark_plonk_amd.prover_schedule import ProofSchedule
from bench import build_srs

GIB = float(1 << 30)


def used():
    f, t = torch.cuda.mem_get_info()
    return (t - f) / GIB


def probe(log_n):
    n = 1 << log_n
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve(0)
    base = used()
    srs = build_srs(ctx, cv, n, 0, n, torch)
    ck = zk.CommitterKey(srs, cv, ctx).precompute()
    del srs
    torch.cuda.empty_cache()
    row = {"log_n": log_n, "window_bits": ck.table_window_bits(), "table_gib": round(used() - base, 2)}
    s = ProofSchedule(log_n, ctx, ck, cv)
    row["schedule_vectors_gib"] = round(used() - base - row["table_gib"], 2)
    try:
        s.run_once()
        torch.cuda.synchronize()
        st = ctx.round_mem_stats()
        row["after_one_proof_gib"] = round(used() - base, 2)
        row["job_sets_gib"] = round(st["set_bytes"] / GIB, 2)
        row["early_closes"] = st["early_closes"]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        s.run_once()
        ev1.record()
        torch.cuda.synchronize()
        row["ms_per_proof"] = round(ev0.elapsed_time(ev1), 1)
    except Exception as e:      # noqa: BLE001 -- the probe reports what happened
        row["failed"] = str(e)
    print(json.dumps(row), flush=True)
    del s
    ck.close()
    ctx.close()
    gc.collect()
    torch.cuda.empty_cache()


if __name__ == "__main__":
    for a in ([int(x) for x in sys.argv[1:]] or [20, 22, 24, 25]):
        probe(a)
