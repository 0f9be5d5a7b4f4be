"""Host-pointer entry points: what the PCIe side achieves.  Times zk_ntt on pageable host buffers (pinned staging ring vs plain
hipMemcpyAsync), the same transform device-resident, and a 7-polynomial zk_kzg_commit_batch against its device-resident form.
usage: python tools/pcie_probe.py [log_n]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import _lib  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << log_n
ctx = zk.Context(0)
ctx.use_torch_stream()
dom = zk.Radix2EvaluationDomain.new(n, 0, ctx)
rng = np.random.default_rng(0)
x = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
buf = x.copy()
out = np.empty_like(x)
d = torch.from_numpy(x.view(np.int64)).cuda()
dout = torch.empty_like(d)


def timeit(fn, k=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k


t_dev = timeit(lambda: dom._run(1, d, out=dout))
print(f"ifft 2^{log_n} device-resident: {t_dev * 1e3:.3f} ms")
for staged in (True, False):
    ctx.set_staging(staged)
    t = timeit(lambda: dom._run(1, x, out=out))
    io = 2 * n * 32
    print(f"ifft 2^{log_n} host buffers, {'pinned staging ring' if staged else 'plain hipMemcpyAsync (pageable)'}: {t * 1e3:.3f} ms "
          f"-> {(io / (t - t_dev)) / 1e9:.1f} GB/s for the {io >> 20} MiB of H2D + D2H")
    t = timeit(lambda: dom._run(2, x[: n // 4], out=out))
    io = (n // 4 + n) * 32
    print(f"  coset_fft n/4 -> n: {t * 1e3:.3f} ms ({io >> 20} MiB)")
ctx.set_staging(True)
# raw copies
t = timeit(lambda: _lib.check(_lib.lib().zk_dev_upload(ctx.handle, d.data_ptr(), x.ctypes.data, n * 32)))
print(f"zk_dev_upload (plain pageable H2D) {n * 32 / t / 1e9:.1f} GB/s")
t = timeit(lambda: _lib.check(_lib.lib().zk_dev_download(ctx.handle, out.ctypes.data, d.data_ptr(), n * 32)))
print(f"zk_dev_download (plain pageable D2H) {n * 32 / t / 1e9:.1f} GB/s")
pin = torch.empty((n, 4), dtype=torch.int64).pin_memory()
t = timeit(lambda: d.copy_(pin, non_blocking=True))
print(f"pinned H2D {n * 32 / t / 1e9:.1f} GB/s")
t = timeit(lambda: pin.copy_(d, non_blocking=True))
print(f"pinned D2H {n * 32 / t / 1e9:.1f} GB/s")
t = timeit(lambda: np.copyto(out, x))
print(f"host memcpy one thread {n * 32 / t / 1e9:.1f} GB/s")
# host pinned buffers handed to the host entry points are sent directly
xp = pin.numpy().view(np.uint64)
xp[:] = x
outp = torch.empty((n, 4), dtype=torch.int64).pin_memory().numpy().view(np.uint64)
t = timeit(lambda: dom._run(1, xp, out=outp))
print(f"ifft 2^{log_n} host PINNED buffers: {t * 1e3:.3f} ms -> {(2 * n * 32 / (t - t_dev)) / 1e9:.1f} GB/s")
