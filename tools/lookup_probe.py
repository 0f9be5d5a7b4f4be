#!/usr/bin/env python3
"""Times round 2's device kernels (zk_lookup_query_dev, zk_lookup_combine_split_dev) at n = 2^LOG on a padded table and a
query column with DUMMY of its rows on the dummy value.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times."""
import sys
import time

import torch

import ark_plonk_amd as zk
from ark_plonk_amd import lookup

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dummy = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
n = 1 << log_n
ctx = zk.Context(0)
g = torch.Generator(device="cuda").manual_seed(1)
rows = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
rows[:, 3] &= (1 << 60) - 1
rep = torch.arange(n, device="cuda")
rep[n // 4:] = 0
t = rows[rep].contiguous()
pick = torch.randint(0, n // 4, (n,), device="cuda", generator=g)
pick[torch.rand(n, device="cuda", generator=g) < dummy] = 0
f = t[pick].contiguous()
for _ in range(3):
    h1, h2 = lookup.combine_split(t, f, 0, ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    h1, h2 = lookup.combine_split(t, f, 0, ctx)
torch.cuda.synchronize()
print(f"combine_split 2^{log_n} dummy={dummy}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per call, halves {h1.shape[0]} {h2.shape[0]}")
