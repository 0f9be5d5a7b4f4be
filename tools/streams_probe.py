"""Throughput of S concurrent proof streams on ONE GPU (one thread + zk_ctx + HIP stream each)."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd.prover_schedule import ProofSchedule  # noqa: E402
from bench import build_srs  # noqa: E402

log_n = int(os.environ.get("LOG_N", "20"))
n = 1 << log_n
torch.cuda.set_device(0)
cv = zk.get_curve("bls12_381")
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    ctxs, cks, scheds, streams = [], [], [], []
    for s in range(S):
        ctx = zk.Context(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            srs = build_srs(ctx, cv, n, 0, n, torch)
            ck = zk.CommitterKey(srs, cv, ctx).precompute()
            del srs
            sc = ProofSchedule(log_n, ctx, ck, cv)
            sc.run_once()
        torch.cuda.synchronize()
        ctxs.append(ctx); cks.append(ck); scheds.append(sc); streams.append(st)
    K = 4
    bar = threading.Barrier(S + 1)

    def worker(i):
        with torch.cuda.stream(streams[i]):
            bar.wait()
            for _ in range(K):
                scheds[i].run_once()
            streams[i].synchronize()
        bar.wait()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    print(f"streams={S}: {S * K / dt:.3f} proofs/s ({dt / (S * K) * 1e3:.2f} ms per proof aggregate, {dt / K * 1e3:.1f} ms latency)", flush=True)
    for ck in cks:
        ck.close()
    del scheds, cks, ctxs
