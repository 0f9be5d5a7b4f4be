"""Curve ids and limb helpers of the host layer (product code; independent of oracle/)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class CurveInfo:
    name: str
    curve_id: int
    r: int
    q: int
    fq_limbs: int        # 64-bit limbs per Fq element
    two_adicity: int
    scalar_bits: int
    b: int = 0           # G1: y^2 = x^3 + b
    gx: int = 0          # the standard G1 generator (on the curve: tests/test_abi.py)
    gy: int = 0


BLS12_381 = CurveInfo(
    "bls12_381", 0,
    0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    6, 32, 255, 4,
    0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
    0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,
)
BN254 = CurveInfo(
    "bn254", 1,
    21888242871839275222246405745257275088548364400416034343698204186575808495617,
    21888242871839275222246405745257275088696311157297823662689037894645226208583,
    4, 28, 254, 3, 1, 2,
)

_BY_KEY = {0: BLS12_381, 1: BN254, "bls12_381": BLS12_381, "bn254": BN254, "bls12-381": BLS12_381}


def get_curve(c) -> CurveInfo:
    if isinstance(c, CurveInfo):
        return c
    try:
        return _BY_KEY[c]
    except KeyError:
        raise ValueError(f"unknown curve {c!r}")


def ints_to_limbs(vals, limbs: int) -> np.ndarray:
    """Python ints -> (len, limbs) little-endian uint64 array."""
    nbytes = limbs * 8
    buf = b"".join(int(v).to_bytes(nbytes, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, limbs).copy()


def limbs_to_ints(arr) -> list:
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    nbytes = a.shape[1] * 8
    raw = a.astype("<u8").tobytes()
    return [int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "little") for i in range(a.shape[0])]


def fr_to_mont(curve, vals) -> np.ndarray:
    cv = get_curve(curve)
    R = 1 << 256
    return ints_to_limbs([(int(v) % cv.r) * R % cv.r for v in vals], 4)


def fr_from_mont(curve, arr) -> list:
    cv = get_curve(curve)
    rinv = pow(1 << 256, -1, cv.r)
    return [v * rinv % cv.r for v in limbs_to_ints(arr)]


def fq_to_mont(curve, vals) -> np.ndarray:
    cv = get_curve(curve)
    R = 1 << (64 * cv.fq_limbs)
    return ints_to_limbs([(int(v) % cv.q) * R % cv.q for v in vals], cv.fq_limbs)


def fq_from_mont(curve, arr) -> list:
    cv = get_curve(curve)
    rinv = pow(1 << (64 * cv.fq_limbs), -1, cv.q)
    return [v * rinv % cv.q for v in limbs_to_ints(arr)]


def g1_mul(curve, k: int):
    """k * G on y^2 = x^3 + b by affine double-and-add over Python integers: (x, y), or None for the point at infinity.  A few hundred
    modular inversions -- milliseconds; for checks of single results (bench.py's KZG identity), never on a compute path."""
    cv = get_curve(curve)
    q = cv.q
    k %= cv.r
    acc, add = None, (cv.gx, cv.gy)

    def plus(p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        (x1, y1), (x2, y2) = p1, p2
        if x1 == x2:
            if (y1 + y2) % q == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, q) % q
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
        x3 = (lam * lam - x1 - x2) % q
        return x3, (lam * (x1 - x3) - y1) % q

    while k:
        if k & 1:
            acc = plus(acc, add)
        add = plus(add, add)
        k >>= 1
    return acc
