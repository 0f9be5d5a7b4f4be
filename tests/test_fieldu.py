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
    L.fs_raw_op.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5
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


@pytest.mark.parametrize("field", [0, 1])
def test_signed_limbs_at_the_edge_of_the_column_bound(field, fu):
    """fields.cuh promises: operands with |limb| <= 2^29 + 2 and |value| < 12p go through mul / sqr / dot2 without leaving the signed
    64-bit column, and come out with strict limbs in (-p, p).  Raw limb patterns (no conversion on the way in): every limb at +-(2^29 + 2)
    in all sign combinations that matter (all equal, alternating, one against the other), the top limb at +-12p's, plus random ones."""
    p = bo.BLS12_381.q if field == 0 else bo.BN254.q
    nl = fu.fs_limbs(field)
    Rp = 1 << (30 * nl)
    E = (1 << 29) + 2
    top = (12 * p >> (30 * (nl - 1))) - 1
    rng = np.random.default_rng(field + 77)

    def val(l):
        return sum(int(v) << (30 * i) for i, v in enumerate(l))

    def pat(kind):
        if kind == 0:
            l = [E] * (nl - 1) + [top]
        elif kind == 1:
            l = [-E] * (nl - 1) + [-top]
        elif kind == 2:
            l = [E if i % 2 == 0 else -E for i in range(nl - 1)] + [top]
        elif kind == 3:
            l = [-E if i % 2 == 0 else E for i in range(nl - 1)] + [-top]
        else:
            l = [int(v) for v in rng.integers(-E, E + 1, size=nl - 1)] + [int(rng.integers(-top, top + 1))]
        return np.array(l, dtype=np.int32)

    def strict(l, hi=1):
        return all(-(1 << 29) <= int(v) < (1 << 29) for v in l[:-1]) and abs(val(l)) < hi * p

    out = np.zeros(nl, dtype=np.int32)
    kinds = [0, 1, 2, 3, 4, 4, 4]
    for ka in kinds:
        for kb in kinds:
            a, b, c, d = pat(ka), pat(kb), pat(kb if ka < 4 else 4), pat(ka if kb < 4 else 4)
            A, B, C, D = val(a), val(b), val(c), val(d)
            fu.fs_raw_op(field, 0, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, out.ctypes.data)
            assert strict(out) and (val(out) * Rp - A * B) % p == 0, ("mul", ka, kb)
            fu.fs_raw_op(field, 1, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, out.ctypes.data)
            assert strict(out) and (val(out) * Rp - A * A) % p == 0, ("sqr", ka)
            fu.fs_raw_op(field, 2, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, out.ctypes.data)
            assert strict(out) and (val(out) * Rp - A * B - C * D) % p == 0, ("dot2", ka, kb)
            fu.fs_raw_op(field, 4, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, out.ctypes.data)
            assert val(out) == A + B + C - D and all(abs(int(v)) <= E for v in out[:-1]), ("add3 / sub", ka, kb)
    # the one-step X3: strict operands at the ends of their range
    S = (1 << 29)
    for sa, sb in ((S - 1, -S), (-S, S - 1), (S - 1, S - 1), (-S, -S)):
        a = np.array([sa] * (nl - 1) + [5], dtype=np.int32)
        b = np.array([sb] * (nl - 1) + [-3], dtype=np.int32)
        fu.fs_raw_op(field, 3, a.ctypes.data, b.ctypes.data, b.ctypes.data, b.ctypes.data, out.ctypes.data)
        assert val(out) == val(a) - 3 * val(b) and all(abs(int(v)) <= E for v in out[:-1]), (sa, sb)
