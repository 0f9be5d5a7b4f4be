"""Is a single n = 2^20 transform latency-bound?  Time K transforms on one stream vs the same K on each of two
streams (two contexts) running concurrently."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
K = 200
torch.cuda.set_device(0)
ctxs = [zk.Context(0) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
doms = [zk.Radix2EvaluationDomain.new(n, 0, c) for c in ctxs]
g = torch.Generator(device="cuda").manual_seed(1)
xs = [torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g) for _ in range(2)]
outs = [torch.empty_like(x) for x in xs]


def run(i, k):
    with torch.cuda.stream(streams[i]):
        for _ in range(k):
            doms[i]._run(1, xs[i], out=outs[i])
        streams[i].synchronize()


for i in range(2):
    run(i, 5)
t0 = time.perf_counter()
run(0, K)
t1 = time.perf_counter() - t0
th = [threading.Thread(target=run, args=(i, K)) for i in range(2)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
t2 = time.perf_counter() - t0
print(f"log_n={log_n}: one stream {t1 / K * 1e6:.1f} us per ifft; two streams {t2 / K * 1e6:.1f} us per pair "
      f"({t2 / t1:.2f}x the single time for twice the work)", flush=True)
