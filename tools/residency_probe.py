"""would a residency cache in front of the host-pointer entry points pay?  A hit replaces an upload by a 256-bit
digest of the caller's bytes (the only sound key: the caller may have rewritten the buffer), so the gain per hit is
(upload time - digest time).  This probe measures both for one 2^20-element vector (32 MiB) on this box:
  digest: zk_srs_register of bytes the registry already holds = host_digest256 of the buffer + a lookup (hostio.hip, the ctx-less pool);
  upload: zk_dev_upload of the same number of bytes from pageable memory, and hipMemcpy through torch from pinned memory.
usage: python tools/residency_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd import _lib
    L = _lib.lib()
    ctx = zk.Context(0)
    cv = zk.get_curve("bn254")           # 64-byte affine points: 2^19 of them are the 32 MiB of a 2^20-element Fr vector
    n_pts = 1 << 19
    rng = np.random.default_rng(3)
    pts = rng.integers(0, 1 << 60, size=(n_pts, 8), dtype=np.uint64)     # bytes only: the registry digests them (never used as points)
    pts[:, 3] &= (1 << 59) - 1
    pts[:, 7] &= (1 << 59) - 1
    nbytes = pts.nbytes
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = L.zk_srs_register(ctx.handle, cv.curve_id, pts.ctypes.data_as(ctypes.c_void_p), None, n_pts, ctypes.byref(h))
    t_first = time.perf_counter() - t0
    assert rc == 0, rc
    hits = []
    for _ in range(20):
        h2 = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = L.zk_srs_register(ctx.handle, cv.curve_id, pts.ctypes.data_as(ctypes.c_void_p), None, n_pts, ctypes.byref(h2))
        hits.append(time.perf_counter() - t0)
        assert rc == 0
        L.zk_srs_free(h2)
    d = ctypes.c_void_p()
    _lib.check(L.zk_dev_alloc(ctx.handle, nbytes, ctypes.byref(d)))
    ups = []
    for _ in range(20):
        t0 = time.perf_counter()
        _lib.check(L.zk_dev_upload(ctx.handle, d, pts.ctypes.data_as(ctypes.c_void_p), nbytes))
        ups.append(time.perf_counter() - t0)
    pin = torch.from_numpy(pts.view(np.int64)).pin_memory()
    dev = torch.empty_like(pin, device="cuda")
    pups = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        pups.append(time.perf_counter() - t0)
    _lib.check(L.zk_dev_free(ctx.handle, d))
    L.zk_srs_free(h)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"vector: {nbytes / 2**20:.0f} MiB; host threads: {os.cpu_count()}")
    print(f"digest + lookup (zk_srs_register hit): median {med(hits) * 1e3:.3f} ms = {nbytes / med(hits) / 1e9:.1f} GB/s   (first registration {t_first * 1e3:.1f} ms)")
    print(f"upload, pageable (zk_dev_upload)     : median {med(ups) * 1e3:.3f} ms = {nbytes / med(ups) / 1e9:.1f} GB/s")
    print(f"upload, pinned (torch copy_)         : median {med(pups) * 1e3:.3f} ms = {nbytes / med(pups) / 1e9:.1f} GB/s")
    print(f"gain of a hit over an upload: {(med(ups) - med(hits)) * 1e3:.3f} ms per 32 MiB vector; cost of a miss: +{med(hits) * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
