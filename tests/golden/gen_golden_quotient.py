#!/usr/bin/env python3
"""Golden vectors for the pointwise quotient kernel from the big-int oracle:
tests/golden/quotient.npz -- per curve, n = 4 (16 coset points): 28 seeded columns, 13 challenges /
coefficients and the 16 expected quotient evaluations, all as Montgomery limb arrays (the ABI form)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_oracle as bo  # noqa: E402


def mont(cv, xs):
    return np.array([bo.int_to_limbs(bo.to_mont(x, cv.r, cv.fr_R), 4) for x in xs], dtype=np.uint64).reshape(-1, 4)


def case(cv, log_n, seed):
    n4 = 4 << log_n
    col = {name: bo.seeded_scalars(cv, seed + k, n4) for k, name in enumerate(bo.QUOTIENT_COLS)}
    chv = bo.seeded_scalars(cv, seed + 0x80, len(bo.QUOTIENT_CHALLENGES))
    ch = dict(zip(bo.QUOTIENT_CHALLENGES, chv))
    return col, ch


def main():
    out = {}
    for cid in (0, 1):
        cv = bo.CURVES[cid]
        col, ch = case(cv, 2, 0x7100 + 0x100 * cid)
        q = bo.quotient_evals(cv, 2, col, ch)
        for name in bo.QUOTIENT_COLS:
            out[f"{cv.name}_col_{name}"] = mont(cv, col[name])
        out[f"{cv.name}_challenges"] = mont(cv, [ch[k] for k in bo.QUOTIENT_CHALLENGES])
        out[f"{cv.name}_quotient"] = mont(cv, q)
    path = os.path.join(ROOT, "tests", "golden", "quotient.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
