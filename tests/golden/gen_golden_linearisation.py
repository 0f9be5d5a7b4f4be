#!/usr/bin/env python3
"""Golden vectors for round 5's linearisation (linearisation_poly.rs:164-350) from the big-int oracle:
tests/golden/linearisation.npz -- per curve, n = 8: the 16 prover-key polynomials, the 14 round polynomials (ragged: t_4 one
coefficient short, f two), 15 challenges / coefficients, the expected linearisation polynomial and the 23 evaluations, all as
Montgomery limb arrays (the ABI form)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_oracle as bo  # noqa: E402

LOG_N = 3
EVAL_NAMES = ("a_eval", "b_eval", "c_eval", "d_eval", "left_sigma_eval", "right_sigma_eval", "out_sigma_eval", "permutation_eval",
              "q_lookup_eval", "z2_next_eval", "h1_eval", "h1_next_eval", "h2_eval", "f_eval", "table_eval", "table_next_eval",
              "q_arith_eval", "q_c_eval", "q_l_eval", "q_r_eval", "a_next_eval", "b_next_eval", "d_next_eval")
CHALLENGES = bo.QUOTIENT_CHALLENGES + ("z",)


def mont(cv, xs):
    return np.array([bo.int_to_limbs(bo.to_mont(x, cv.r, cv.fr_R), 4) for x in xs], dtype=np.uint64).reshape(-1, 4)


def case(cv, log_n, seed):
    n = 1 << log_n
    key = {name: bo.seeded_scalars(cv, seed + k, n) for k, name in enumerate(bo.LIN_KEY)}
    polys = {name: bo.seeded_scalars(cv, seed + 0x40 + k, n) for k, name in enumerate(bo.LIN_POLYS)}
    polys["t_4"] = polys["t_4"][:n - 1]
    polys["f"] = polys["f"][:n - 2]
    ch = dict(zip(CHALLENGES, bo.seeded_scalars(cv, seed + 0x80, len(CHALLENGES))))
    return key, polys, ch


def main():
    out = {}
    for cid in (0, 1):
        cv = bo.CURVES[cid]
        key, polys, ch = case(cv, LOG_N, 0x9100 + 0x100 * cid)
        lin, ev = bo.linearisation(cv, LOG_N, key, polys, ch)
        for name in bo.LIN_KEY:
            out[f"{cv.name}_key_{name}"] = mont(cv, key[name])
        for name in bo.LIN_POLYS:
            out[f"{cv.name}_poly_{name}"] = mont(cv, polys[name])
        out[f"{cv.name}_challenges"] = mont(cv, [ch[k] for k in CHALLENGES])
        out[f"{cv.name}_lin"] = mont(cv, lin)
        out[f"{cv.name}_evals"] = mont(cv, [ev[k] for k in EVAL_NAMES])
    path = os.path.join(ROOT, "tests", "golden", "linearisation.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
