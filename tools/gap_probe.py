"""Host-side tail of a round batch: wall time of commit_batch minus the GPU time its kernels took."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from bench import build_srs  # noqa: E402

log_n = 20
n = 1 << log_n
torch.cuda.set_device(0)
ctx = zk.Context(0)
ctx.use_torch_stream()
cv = zk.get_curve("bls12_381")
srs = build_srs(ctx, cv, n, 0, n, torch)
ck = zk.CommitterKey(srs, cv, ctx).precompute()
g = torch.Generator(device="cuda").manual_seed(1)
polys = [torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g) for _ in range(16)]
for k in (1, 2, 4, 16):
    ck.commit_batch(polys[:k])
    torch.cuda.synchronize()
    ctx.profile(True)
    ctx.profile_reset()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ck.commit_batch(polys[:k])
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile(False)
    gpu = sum(ctx.profile_get(nm)[0] for nm in ("msm_accumulate", "msm_sort", "msm_reduce", "fr_convert")) / reps
    print(f"jobs={k}: wall {wall:.3f} ms, gpu scopes {gpu:.3f} ms, tail {wall - gpu:.3f} ms", flush=True)
