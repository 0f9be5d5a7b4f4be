#!/usr/bin/env python3
"""Golden vectors for the grand-product builders from the big-int oracle:
tests/golden/grand_product.npz.  Inputs and expected outputs are Montgomery limb arrays (the ABI form).

Cases per curve: n = 8 and n = 64 with seeded uniform columns (generic sigma: the product does not close),
and an n = 16 case whose sigma is a real wire permutation with consistent wire values (the dropped
(n+1)-th value is 1, permutation/mod.rs:1243-1380 checks exactly that property)."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_oracle as bo  # noqa: E402


def mont(cv, xs):
    return np.array([bo.int_to_limbs(bo.to_mont(x, cv.r, cv.fr_R), 4) for x in xs], dtype=np.uint64).reshape(-1, 4)


def valid_permutation(cv, log_n, seed):
    """Wire values constant on the cycles of a random permutation of the 4n wire slots; sigma_k[i] = K_k' * w^i'."""
    rnd = random.Random(seed)
    n = 1 << log_n
    w = cv.root_of_unity(log_n)
    roots = [pow(w, i, cv.r) for i in range(n)]
    n_vars = n  # ~4 slots per variable
    var_of = [rnd.randrange(n_vars) for _ in range(4 * n)]
    val = [rnd.randrange(cv.r) for _ in range(n_vars)]
    slots = {}
    for pos, v in enumerate(var_of):
        slots.setdefault(v, []).append(pos)
    sigma_pos = list(range(4 * n))
    for ps in slots.values():
        for a, b in zip(ps, ps[1:] + ps[:1]):
            sigma_pos[a] = b                       # slot a maps to the next slot of the same variable
    wires = [[val[var_of[k * n + i]] for i in range(n)] for k in range(4)]
    sigmas = [[bo.PERM_K[sigma_pos[k * n + i] // n] * roots[sigma_pos[k * n + i] % n] % cv.r for i in range(n)] for k in range(4)]
    return wires, sigmas


def main():
    out = {}
    for cid in (0, 1):
        cv = bo.CURVES[cid]
        pre = cv.name
        for log_n in (3, 6):
            n = 1 << log_n
            cols = [bo.seeded_scalars(cv, 0x6000 + 16 * log_n + k, n) for k in range(8)]
            beta, gamma = bo.seeded_scalars(cv, 0x6100 + log_n, 2)
            z, last = bo.perm_product(cv, log_n, cols[:4], cols[4:], beta, gamma)
            for k in range(4):
                out[f"{pre}_perm{log_n}_w{k}"] = mont(cv, cols[k])
                out[f"{pre}_perm{log_n}_s{k}"] = mont(cv, cols[4 + k])
            out[f"{pre}_perm{log_n}_beta_gamma"] = mont(cv, [beta, gamma])
            out[f"{pre}_perm{log_n}_z"] = mont(cv, z)
            out[f"{pre}_perm{log_n}_last"] = mont(cv, [last])
            f, t, h1, h2 = (bo.seeded_scalars(cv, 0x6200 + 16 * log_n + k, n) for k in range(4))
            delta, eps = bo.seeded_scalars(cv, 0x6300 + log_n, 2)
            p, lastp = bo.lookup_product(cv, f, t, h1, h2, delta, eps)
            for nm, col in (("f", f), ("t", t), ("h1", h1), ("h2", h2)):
                out[f"{pre}_look{log_n}_{nm}"] = mont(cv, col)
            out[f"{pre}_look{log_n}_delta_eps"] = mont(cv, [delta, eps])
            out[f"{pre}_look{log_n}_p"] = mont(cv, p)
            out[f"{pre}_look{log_n}_last"] = mont(cv, [lastp])
        wires, sigmas = valid_permutation(cv, 4, 0x6400 + cid)
        beta, gamma = bo.seeded_scalars(cv, 0x6500, 2)
        z, last = bo.perm_product(cv, 4, wires, sigmas, beta, gamma)
        assert last == 1 and z[0] == 1
        for k in range(4):
            out[f"{pre}_permv_w{k}"] = mont(cv, wires[k])
            out[f"{pre}_permv_s{k}"] = mont(cv, sigmas[k])
        out[f"{pre}_permv_beta_gamma"] = mont(cv, [beta, gamma])
        out[f"{pre}_permv_z"] = mont(cv, z)
    path = os.path.join(ROOT, "tests", "golden", "grand_product.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
