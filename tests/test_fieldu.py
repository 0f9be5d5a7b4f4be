"""CPU check of the unsaturated fields (29-bit limbs for the scalar fields: fieldu.cuh; signed 30-bit limbs for the base
fields: fields.cuh) and of the lazy XYZZ group law (ecu.cuh) that the HIP kernels use, compiled for the host and compared
with big-int arithmetic."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import bigint_oracle as bo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "fu_check.cpp")
SO = os.path.join(ROOT, "tests", "native", "libfu_check.so")


@pytest.fixture(scope="module")
def fu():
    deps = [SRC] + [os.path.join(ROOT, "ark_plonk_amd", "csrc", f) for f in ("fieldu.cuh", "fields.cuh", "ecu.cuh", "curve_params.h", "zk_common.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", SRC, "-o", SO])
    L = ctypes.CDLL(SO)
    L.fu_op.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.fu_xyzz_chain.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return L


FIELDS = {0: (bo.BLS12_381.q, 12), 1: (bo.BLS12_381.r, 8), 2: (bo.BN254.q, 8), 3: (bo.BN254.r, 8)}   # 0, 2: signed limbs


def words(v, n):
    return np.frombuffer(int(v).to_bytes(4 * n, "little"), dtype="<u4").copy()


def unwords(a):
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u4").tobytes(), "little")


@pytest.mark.parametrize("field", [0, 1, 2, 3])
def test_field_ops(field, fu):
    p, n = FIELDS[field]
    R = 1 << (32 * n)
    Rinv = pow(R, -1, p)
    rng = np.random.default_rng(field)
    specials = [0, 1, p - 1, p - 2, 2, (1 << 29) - 1, 1 << 29, (p - 1) // 2]
    # limb patterns at the edges of the digit ranges: every 29- / 30-bit digit all ones, 2^29 or 2^29 - 1 in every 30-bit digit
    specials += [v % p for v in ((1 << (32 * n)) - 1, sum(1 << (30 * i + 29) for i in range(13)), sum(((1 << 29) - 1) << (30 * i) for i in range(13)),
                                 sum(((1 << 29) + 1) << (30 * i) for i in range(13)), (p + 1) // 2, (1 << (p.bit_length() - 1)) - 1)]
    vals = specials + [int.from_bytes(rng.bytes(4 * n + 8), "little") % p for _ in range(40)]
    out = np.zeros(n, dtype="<u4")
    for i, x in enumerate(vals):
        y = vals[(i * 7 + 3) % len(vals)]
        xm, ym = x * R % p, y * R % p   # arkworks Montgomery form in, same form out
        exp = {
            0: x * y % p, 1: (x + y) % p, 2: (x - y) % p, 3: (x - y) % p, 4: x * x % p,
            5: pow(x, -1, p) if x else 0, 6: (-x) % p, 7: 2 * x % p,
            8: pow(((x - y) * (2 * x + y) - x * y) % p, 2, p),
        }
        if field in (0, 2):   # dot2 and the one-step X3 are the base fields'
            exp[9] = (x * y + (x - y) * (2 * x + y)) % p
            exp[10] = (x * y - y * y) % p
            exp[11] = (x * y - 2 * x * x - y * y) % p
        wa, wb = words(xm, n), words(ym, n)   # keep the buffers alive across the call
        for op, e in exp.items():
            fu.fu_op(field, op, wa.ctypes.data, wb.ctypes.data, out.ctypes.data)
            got = unwords(out) * Rinv % p
            assert unwords(out) < p, (field, op)
            assert got == e, (field, op, hex(x), hex(y))


@pytest.mark.parametrize("cid", [0, 1])
def test_xyzz_chain_matches_affine_group_law(cid, fu):
    cv = bo.CURVES[cid]
    W = 2 * cv.fq_limbs
    R = cv.fq_R
    G = (cv.gx, cv.gy)
    pts = [bo.ec_mul(cv, k, G) for k in (1, 2, 3, 5, 7, 1, 11, 7, 13, 5)]
    rng = np.random.default_rng(cid + 10)

    def run(seq):
        """seq: list of (point or None, flags)"""
        arr = np.zeros((len(seq), 2 * W), dtype="<u4")
        fl = np.zeros(len(seq), dtype=np.uint8)
        for i, (pt, f) in enumerate(seq):
            if pt is not None:
                arr[i, :W] = words(pt[0] * R % cv.q, W)
                arr[i, W:] = words(pt[1] * R % cv.q, W)
            fl[i] = f
        out = np.zeros(2 * W, dtype="<u4")
        inf = fu.fu_xyzz_chain(cid, arr.ctypes.data, fl.ctypes.data, len(seq), out.ctypes.data)
        if inf:
            return None
        rinv = pow(R, -1, cv.q)
        return (unwords(out[:W]) * rinv % cv.q, unwords(out[W:]) * rinv % cv.q)

    def expect(seq):
        acc = None
        for pt, f in seq:
            if pt is None:
                continue
            q = bo.ec_neg(cv, pt) if f & 1 else pt
            if f & 4:
                acc = bo.ec_add(cv, acc, acc)
            acc = bo.ec_add(cv, acc, q)
        return acc

    cases = [
        [(pts[0], 0)],
        [(pts[0], 0), (pts[0], 0)],                       # P + P  (madd doubling branch)
        [(pts[0], 0), (pts[0], 1)],                       # P - P  (infinity)
        [(pts[0], 0), (pts[0], 1), (pts[2], 0)],          # back from infinity
        [(pts[1], 0), (pts[2], 0), (pts[3], 2), (pts[3], 2)],   # full add, then full add of same point
        [(pts[1], 2), (pts[1], 2)],                       # add-2008-s doubling branch
        [(pts[1], 2), (pts[1], 3)],                       # add-2008-s cancellation
        [(pts[4], 0), (None, 0), (pts[5], 4), (pts[6], 5), (pts[7], 6)],
        [(p, int(rng.integers(0, 8))) for p in pts] * 3,  # long mixed chain: lazy bounds hold over many ops
    ]
    for seq in cases:
        assert run(seq) == expect(seq), seq
