"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/libark_cpu_oracle.so (the CPU restatement
of the arkworks 0.3 NTT / Pippenger algorithms, oracle/ark_cpu.cpp).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libark_cpu_oracle.so")
_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ark_cpu.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libark_cpu_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.ora_ntt.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64p, ctypes.c_size_t, _u64p]
        L.ora_msm_g1.argtypes = [ctypes.c_int, _u64p, _u8p, _u64p, ctypes.c_size_t, _u64p, _u8p, ctypes.c_int]
        L.ora_msm_g1_chunked.argtypes = [ctypes.c_int, _u64p, _u8p, _u64p, ctypes.c_size_t, _u64p, _u8p, ctypes.c_int, ctypes.c_int]
        L.ora_srs_powers.argtypes = [ctypes.c_int, _u64p, ctypes.c_size_t, _u64p]
        L.ora_kzg_commit.argtypes = [ctypes.c_int, _u64p, ctypes.c_size_t, _u64p, ctypes.c_size_t, _u64p, _u8p, ctypes.c_int]
        L.ora_kzg_witness.argtypes = [ctypes.c_int, _u64p, ctypes.c_size_t, _u64p, _u64p]
        L.ora_convert.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64p, ctypes.c_size_t, _u64p]
        L.ora_fr_op.argtypes = [ctypes.c_int, ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, _u64p]
        L.ora_window_size.argtypes = [ctypes.c_size_t]
        L.ora_set_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _p8(a):
    return None if a is None else a.ctypes.data_as(_u8p)


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


FQ_LIMBS = {0: 6, 1: 4}


def num_threads() -> int:
    return lib().ora_num_threads()


def set_threads(t: int):
    lib().ora_set_threads(t)


def ntt(curve_id: int, kind: int, log_n: int, values_mont: np.ndarray) -> np.ndarray:
    """values_mont: (in_len, 4) uint64 Montgomery limbs -> (2^log_n, 4)."""
    v = _c64(values_mont).reshape(-1, 4)
    out = np.empty((1 << log_n, 4), dtype=np.uint64)
    rc = lib().ora_ntt(curve_id, kind, log_n, _p64(v), v.shape[0], _p64(out))
    if rc:
        raise ValueError(f"ora_ntt rc={rc}")
    return out


def msm_g1(curve_id: int, bases_xy: np.ndarray, scalars: np.ndarray, inf=None, threads: int = 0):
    """bases_xy: (n, 2*L) Montgomery; scalars: (n, 4) canonical. Returns (xy (2L,), inf flag)."""
    L = FQ_LIMBS[curve_id]
    b = _c64(bases_xy).reshape(-1, 2 * L)
    s = _c64(scalars).reshape(-1, 4)
    n = min(b.shape[0], s.shape[0])
    inf_a = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(2 * L, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    rc = lib().ora_msm_g1(curve_id, _p64(b), _p8(inf_a), _p64(s), n, _p64(out), _p8(oinf),
                          threads if threads > 0 else num_threads())
    if rc:
        raise ValueError(f"ora_msm_g1 rc={rc}")
    return out, int(oinf[0])


def window_size(n: int) -> int:
    """ark 0.3's Pippenger window for n points: 3 below 32, else ceil(log2 n) * 69 / 100 + 2."""
    return int(lib().ora_window_size(n))


def msm_g1_all_cores(curve_id: int, bases_xy: np.ndarray, scalars: np.ndarray, inf=None, threads: int = 0, parts: int = 0):
    """The same MSM cut into (window, point range) tasks so that every core is busy -- not ark's shape (threads over the windows
    only); `parts` ranges per window (0: threads / windows, at least 1)."""
    L = FQ_LIMBS[curve_id]
    b = _c64(bases_xy).reshape(-1, 2 * L)
    s = _c64(scalars).reshape(-1, 4)
    n = min(b.shape[0], s.shape[0])
    inf_a = None if inf is None else np.ascontiguousarray(inf, dtype=np.uint8)
    out = np.zeros(2 * L, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    t = threads if threads > 0 else num_threads()
    if parts <= 0:
        c = lib().ora_window_size(n)
        w = -(-(255 if curve_id == 0 else 254) // c)
        parts = max(1, -(-t // w))
    rc = lib().ora_msm_g1_chunked(curve_id, _p64(b), _p8(inf_a), _p64(s), n, _p64(out), _p8(oinf), t, parts)
    if rc:
        raise ValueError(f"ora_msm_g1_chunked rc={rc}")
    return out, int(oinf[0])


def srs_powers(curve_id: int, tau: int, n: int) -> np.ndarray:
    L = FQ_LIMBS[curve_id]
    t = np.array([(tau >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.empty((n, 2 * L), dtype=np.uint64)
    rc = lib().ora_srs_powers(curve_id, _p64(t), n, _p64(out))
    if rc:
        raise ValueError(f"ora_srs_powers rc={rc}")
    return out


def kzg_commit(curve_id: int, powers_xy: np.ndarray, coeffs_mont: np.ndarray, threads: int = 0):
    L = FQ_LIMBS[curve_id]
    p = _c64(powers_xy).reshape(-1, 2 * L)
    c = _c64(coeffs_mont).reshape(-1, 4)
    out = np.zeros(2 * L, dtype=np.uint64)
    oinf = np.zeros(1, dtype=np.uint8)
    rc = lib().ora_kzg_commit(curve_id, _p64(p), p.shape[0], _p64(c), c.shape[0], _p64(out), _p8(oinf),
                              threads if threads > 0 else num_threads())
    if rc:
        raise ValueError(f"ora_kzg_commit rc={rc}")
    return out, int(oinf[0])


def kzg_witness(curve_id: int, coeffs_mont: np.ndarray, z_mont: np.ndarray) -> np.ndarray:
    c = _c64(coeffs_mont).reshape(-1, 4)
    z = _c64(z_mont).reshape(4)
    out = np.zeros((max(c.shape[0] - 1, 0), 4), dtype=np.uint64)
    lib().ora_kzg_witness(curve_id, _p64(c), c.shape[0], _p64(z), _p64(out))
    return out


def convert(curve_id: int, which: str, to_mont: bool, arr: np.ndarray) -> np.ndarray:
    """which: 'fr' | 'fq'."""
    L = 4 if which == "fr" else FQ_LIMBS[curve_id]
    a = _c64(arr).reshape(-1, L)
    out = np.empty_like(a)
    rc = lib().ora_convert(curve_id, 0 if which == "fr" else 1, 0 if to_mont else 1, _p64(a), a.shape[0], _p64(out))
    if rc:
        raise ValueError("ora_convert")
    return out


def fr_op(curve_id: int, op: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    x = _c64(a).reshape(-1, 4)
    y = _c64(b).reshape(-1, 4)
    out = np.empty_like(x)
    lib().ora_fr_op(curve_id, {"mul": 0, "add": 1, "sub": 2}[op], _p64(x), _p64(y), x.shape[0], _p64(out))
    return out


def ints_to_limbs(vals, L: int) -> np.ndarray:
    out = np.empty((len(vals), L), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(L):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(arr) -> list:
    a = np.asarray(arr, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    return [sum(int(a[i, k]) << (64 * k) for k in range(a.shape[1])) for i in range(a.shape[0])]
