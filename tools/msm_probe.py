"""Dev probe: per-phase MSM timings (HIP events via the ctx profiler) for chosen (n, window) pairs."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ark_plonk_amd as zk
from ark_plonk_amd import _lib

ctx = zk.Context(0)
ctx.use_torch_stream()
cv = zk.get_curve(0)
nmax = 1 << int(os.environ.get('LOG_NMAX', '20'))
g = torch.Generator(device="cuda").manual_seed(1)
ks = torch.randint(1, 1 << 62, (nmax, 4), dtype=torch.int64, device="cuda", generator=g)
ks[:, 1:] = 0
bases = torch.empty((nmax, 12), dtype=torch.int64, device="cuda")
_lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, 0, ks.data_ptr(), nmax, bases.data_ptr()))
ck = zk.CommitterKey(bases, 0, ctx)
if os.environ.get("PRE"):
    t0 = time.perf_counter(); ck.precompute(); torch.cuda.synchronize(); print(f"precompute {time.perf_counter()-t0:.3f}s", flush=True)
scal = torch.randint(0, 1 << 62, (nmax, 4), dtype=torch.int64, device="cuda", generator=g)
cases = [(nmax, 0), (nmax - 1, 0), (nmax - 1, 16), (nmax, 15), (nmax, 14), (nmax, 13), (nmax // 2, 0), (nmax, 17), (nmax, 18)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for n, c in cases:
    ctx.set_msm_window(c)
    ck.msm(scal[:n])
    ctx.profile(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        ck.msm(scal[:n])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ctx.profile(False)
    s = {k: ctx.profile_get(k)[0] / 3 for k in ("msm_sort", "msm_accumulate", "msm_reduce")}
    print(f"n={n} c={c}: wall {dt*1e3:.2f} ms  sort {s['msm_sort']:.2f}  acc {s['msm_accumulate']:.2f}  reduce {s['msm_reduce']:.2f}", flush=True)
