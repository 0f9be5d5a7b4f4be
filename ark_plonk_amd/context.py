"""Context = one zk_ctx = one GPU (one process per GPU)."""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

from . import _lib
from ._lib import check, lib

_default = {}
_default_lock = threading.Lock()


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class Context:
    """Owns a zk_ctx on HIP device `device`.  All product compute goes through it."""

    def __init__(self, device: int = 0, stream=None):
        self._h = ctypes.c_void_p()
        check(lib().zk_ctx_create(int(device), ctypes.byref(self._h)), "zk_ctx_create")
        self.device = int(device)
        self._stream_ptr = None
        self._stream_thread = None
        if stream is not None:
            self.set_stream(stream)
        # several ranks on one host (torchrun sets LOCAL_WORLD_SIZE): the ranks' host pools share the cores instead of each starting
        # min(15, cores - 1) helpers.  The library itself reads no environment variable; its host-side mirror does.
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
        if lws > 1:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            self.set_option("host_workers", max(0, min(15, cores // lws - 1)))

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().zk_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("context closed")
        return self._h

    # -- stream / sync
    def set_stream(self, stream):
        """stream: hipStream_t as int (0 = HIP's default stream) or a torch.cuda.Stream;
        None = back to the ctx's own stream."""
        if stream is None:
            check(lib().zk_ctx_use_own_stream(self.handle), "zk_ctx_use_own_stream")
            self._stream_ptr = None
            self._stream_thread = None
            return
        ptr = int(getattr(stream, "cuda_stream", stream))
        if self._stream_ptr == ptr:
            return
        # the stream is ctx state set by a separate call: two threads alternating streams on one Context would race
        # between set_stream and the launch (include/ark_plonk_amd.h: one ctx per thread and stream)
        me = threading.get_ident()
        if self._stream_thread not in (None, me) and self._stream_ptr is not None:
            raise RuntimeError("Context is bound to another thread's stream: use one Context per thread and stream "
                               "(contexts are cheap and share CommitterKey handles via CommitterKey.with_ctx)")
        check(lib().zk_ctx_set_stream(self.handle, ctypes.c_void_p(ptr)), "zk_ctx_set_stream")
        self._stream_ptr = ptr
        self._stream_thread = me

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device))

    def sync(self):
        check(lib().zk_ctx_sync(self.handle), "zk_ctx_sync")

    def set_msm_window(self, c: int):
        check(lib().zk_ctx_set_msm_window(self.handle, int(c)), "zk_ctx_set_msm_window")

    # -- tuning options of the MSM planner (zk_ctx_set_option; the ZK_* environment hooks of rounds 2-4 are gone)
    OPTIONS = ("msm_merge", "pre_vw", "pre_logg", "chunk_l", "long_rounds", "combine_sg", "pre_max_log_n", "mem_reserve_mb",
               "round_mem_limit_mb", "host_workers", "cache_verify")

    def set_option(self, key: str, value: int):
        check(lib().zk_ctx_set_option(self.handle, key.encode(), int(value)), f"zk_ctx_set_option({key})")

    def get_option(self, key: str) -> int:
        v = ctypes.c_int64()
        check(lib().zk_ctx_get_option(self.handle, key.encode(), ctypes.byref(v)), f"zk_ctx_get_option({key})")
        return v.value

    def round_mem_stats(self) -> dict:
        """Memory budget of the deferred rounds (zk_round_mem_stats): early closes so far, bytes held by the job buffer sets, and the
        device's free / total memory as the budget sees it."""
        v = [ctypes.c_uint64() for _ in range(4)]
        check(lib().zk_round_mem_stats(self.handle, *[ctypes.byref(x) for x in v]), "zk_round_mem_stats")
        return {"early_closes": v[0].value, "set_bytes": v[1].value, "device_free": v[2].value, "device_total": v[3].value}

    # -- N3: content-addressed commitment cache (prover.rs:569-607 re-commits 12 polynomials)
    def set_commit_cache(self, on: bool = True, capacity: int = 0):
        check(lib().zk_ctx_set_commit_cache(self.handle, 1 if on else 0, int(capacity)), "zk_ctx_set_commit_cache")

    def commit_cache_stats(self) -> dict:
        v = [ctypes.c_uint64() for _ in range(3)]
        check(lib().zk_commit_cache_stats(self.handle, *[ctypes.byref(x) for x in v]), "zk_commit_cache_stats")
        return {"hits": v[0].value, "misses": v[1].value, "entries": v[2].value}

    # -- residency cache of the host-pointer entry points (vectors this ctx produced or uploaded are not sent again)
    def set_residency_cache(self, on: bool = True, capacity_bytes: int = 0, max_vector_bytes: int = 0):
        check(lib().zk_ctx_set_residency_cache(self.handle, 1 if on else 0, int(capacity_bytes), int(max_vector_bytes)), "zk_ctx_set_residency_cache")

    def residency_cache_stats(self) -> dict:
        v = [ctypes.c_uint64() for _ in range(4)]
        check(lib().zk_residency_cache_stats(self.handle, *[ctypes.byref(x) for x in v]), "zk_residency_cache_stats")
        return {"hits": v[0].value, "misses": v[1].value, "entries": v[2].value, "bytes": v[3].value}

    def cache_verify_stats(self) -> dict:
        """option "cache_verify": hits of the commitment / residency cache that were checked, and how many did not hold."""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib().zk_cache_verify_stats(self.handle, ctypes.byref(a), ctypes.byref(b)), "zk_cache_verify_stats")
        return {"checked": a.value, "mismatches": b.value}

    # -- host-pointer entry points: PCIe volume and staging mode
    def io_stats(self, reset: bool = False) -> dict:
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib().zk_io_stats(self.handle, ctypes.byref(a), ctypes.byref(b), 1 if reset else 0), "zk_io_stats")
        return {"h2d_bytes": a.value, "d2h_bytes": b.value}

    def set_staging(self, pinned_ring: bool = False):
        check(lib().zk_ctx_set_staging(self.handle, 1 if pinned_ring else 0), "zk_ctx_set_staging")

    # -- profiling
    def profile(self, on=True):
        """True / 1: every kernel scope; 2: msm_accumulate only; False / 0: off."""
        check(lib().zk_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        check(lib().zk_profile_reset(self.handle))

    def profile_get(self, name: str):
        t = ctypes.c_double()
        n = ctypes.c_uint64()
        check(lib().zk_profile_get(self.handle, name.encode(), ctypes.byref(t), ctypes.byref(n)))
        return t.value, n.value


def default_context(device: int = 0) -> Context:
    with _default_lock:
        ctx = _default.get(device)
        if ctx is None:
            ctx = Context(device)
            _default[device] = ctx
        return ctx


def ptr_of(x):
    """Device/host address of a numpy array or torch tensor."""
    if _is_torch(x):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(x.ctypes.data)


def as_host_u64(x, cols: int) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.uint64)
    if a.size % cols:
        raise ValueError(f"array size {a.size} not a multiple of {cols} limbs")
    return a.reshape(-1, cols)


def check_dev_tensor(t, cols: int, device: int):
    import torch
    if not t.is_cuda:
        raise ValueError("expected a CUDA/HIP tensor")
    if t.device.index != device:
        raise ValueError(f"tensor on device {t.device.index}, context on {device}")
    if t.dtype not in (torch.int64, torch.uint64):
        raise ValueError("expected an int64/uint64 limb tensor")
    if not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    if t.numel() % cols:
        raise ValueError(f"tensor size not a multiple of {cols} limbs")
    return t.numel() // cols
