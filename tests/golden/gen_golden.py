#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the definitional big-int oracle (oracle/bigint_oracle.py).

The reference (heliaxdev/ark-plonk) holds no golden vectors for the NTT/MSM boundary and cannot be
built here (Rust; arithmetic in un-vendored crates) -- SURVEY.md 8c.  These fixtures are therefore
produced from the mathematical definitions (naive/recursive DFT over Fr, affine double-and-add over
G1) and pin BOTH the C++ CPU restatement (oracle/ark_cpu.cpp) and the HIP path.  Values are stored
exactly as they cross the C ABI: Montgomery limbs for Fr/Fq elements, canonical limbs for scalars.

Run from the repo root:  python3 tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bigint_oracle as bo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def limbs(vals, n):
    buf = b"".join(int(v).to_bytes(8 * n, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, n).copy()


def fr_mont(cv, vals):
    return limbs([v * cv.fr_R % cv.r for v in vals], 4)


def fq_mont(cv, vals):
    return limbs([v * cv.fq_R % cv.q for v in vals], cv.fq_limbs)


def points_mont(cv, pts):
    """list of affine points (or None) -> (n, 2L) limbs + inf flags; infinity stored as (0, 1)."""
    xs, flags = [], []
    for p in pts:
        if p is None:
            xs += [0, 1]
            flags.append(1)
        else:
            xs += [p[0], p[1]]
            flags.append(0)
    return fq_mont(cv, xs).reshape(-1, 2 * cv.fq_limbs), np.array(flags, dtype=np.uint8)


def gen_ntt(cv):
    out = {}
    for log_n in (0, 1, 2, 4, 6, 10):
        n = 1 << log_n
        inputs = {
            "uniform": bo.seeded_scalars(cv, 0x5EED0000 + log_n, n),
            "zeros": [0] * n,
            "onehot": [0] * (n // 2) + [1] + [0] * (n - n // 2 - 1),
            "short": bo.seeded_scalars(cv, 0x5EED1000 + log_n, max(1, n // 4)),
            "rminus1": [cv.r - 1] * n,
        }
        for iname, vals in inputs.items():
            for kind in range(4):
                key = f"ntt_{log_n}_{iname}_{kind}"
                out[key + "_in"] = fr_mont(cv, vals)
                out[key + "_out"] = fr_mont(cv, bo.ntt(cv, kind, log_n, vals))
    return out


def gen_msm(cv):
    out = {}
    tau = 0x7A5C0DE
    G = (cv.gx, cv.gy)
    nmax = 1024
    srs = bo.srs_powers(cv, tau, nmax)
    b_arr, _ = points_mont(cv, srs)
    out["srs_tau"] = limbs([tau], 4)
    out["srs_1024"] = b_arr
    for n in (1, 2, 31, 32, 33, 1024):
        sc = bo.seeded_scalars(cv, 0x5EED2000 + n, n)
        if n >= 31:
            sc[0] = 0
            sc[1] = 1
            sc[2] = cv.r - 1
            sc[3] = 2
            sc[5] = (1 << 200) + 1
        res = bo.msm(cv, srs[:n], sc)
        # KZG identity cross-check inside the generator: MSM(s, tau^i G) = (sum s_i tau^i) G
        s = sum(sc[i] * pow(tau, i, cv.r) for i in range(n)) % cv.r
        assert bo.ec_mul(cv, s, G) == res
        out[f"msm_srs_{n}_scalars"] = limbs(sc, 4)
        p, f = points_mont(cv, [res])
        out[f"msm_srs_{n}_out"] = p[0]
        out[f"msm_srs_{n}_inf"] = f
    # edge cases on explicit base lists
    P = [bo.ec_mul(cv, k, G) for k in (1, 2, 3, 5, 7, 11)]
    cases = {
        # repeated bases with equal scalars -> forces the P + P doubling branch inside a bucket
        "repeat": ([P[2]] * 8, [0x1234567] * 8),
        # P and -P with equal scalars -> cancels to infinity
        "cancel": ([P[3], bo.ec_neg(cv, P[3])], [987654321, 987654321]),
        # everything in one bucket of every window (identical scalars, distinct points)
        "onebucket": (P, [0xABCDEF0123456789ABCDEF] * len(P)),
        # infinity bases mixed in
        "infbase": ([P[0], None, P[1], None, P[4]], [5, 6, 7, 8, 9]),
        # all-zero scalars
        "zeros": (P, [0] * len(P)),
        # scalars = 1 (ark adds these directly in window 0) and r - 1
        "ones": (P, [1, 1, cv.r - 1, 1, cv.r - 1, 1]),
        # a point and its negation in the same bucket together with a repeat
        "mixed": ([P[1], bo.ec_neg(cv, P[1]), P[1], P[5], P[5]], [77, 77, 77, 1 << 128, 1 << 128]),
        # maximum scalar everywhere
        "maxscalar": (P, [cv.r - 1] * len(P)),
    }
    for name, (pts, sc) in cases.items():
        arr, flags = points_mont(cv, pts)
        res = bo.msm(cv, pts, sc)
        out[f"msm_case_{name}_bases"] = arr
        out[f"msm_case_{name}_inf"] = flags
        out[f"msm_case_{name}_scalars"] = limbs(sc, 4)
        p, f = points_mont(cv, [res])
        out[f"msm_case_{name}_out"] = p[0]
        out[f"msm_case_{name}_outinf"] = f
    # KZG commit / open at n = 64
    n = 64
    polys = [bo.seeded_scalars(cv, 0x5EED3000 + k, n - (k % 3)) for k in range(4)]
    polys[1][0] = 0
    polys[1][1] = 0  # leading zero coefficients (stripped by kzg10::commit)
    z = bo.seeded_scalars(cv, 0x5EED3100, 1)[0]
    chi = bo.seeded_scalars(cv, 0x5EED3200, 1)[0]
    for k, pcoef in enumerate(polys):
        out[f"kzg_poly_{k}"] = fr_mont(cv, pcoef)
        p, f = points_mont(cv, [bo.kzg_commit(cv, srs, pcoef)])
        out[f"kzg_commit_{k}"] = p[0]
    out["kzg_z"] = fr_mont(cv, [z])[0]
    out["kzg_chi"] = fr_mont(cv, [chi])[0]
    p, f = points_mont(cv, [bo.kzg_open(cv, srs, polys, z, chi)])
    out["kzg_open"] = p[0]
    out["kzg_open_inf"] = f
    return out


def gen_constants(cv):
    out = {}
    for log_n in (0, 1, 5, 10, 20, cv.two_adicity):
        w = cv.root_of_unity(log_n)
        out[f"group_gen_{log_n}"] = fr_mont(cv, [w])[0]
        out[f"group_gen_inv_{log_n}"] = fr_mont(cv, [pow(w, -1, cv.r)])[0]
        out[f"size_inv_{log_n}"] = fr_mont(cv, [pow(1 << log_n, -1, cv.r)])[0]
    out["generator"] = fr_mont(cv, [cv.fr_generator])[0]
    out["generator_inv"] = fr_mont(cv, [pow(cv.fr_generator, -1, cv.r)])[0]
    G2 = bo.ec_add(cv, (cv.gx, cv.gy), (cv.gx, cv.gy))
    out["g1_double"] = points_mont(cv, [G2])[0][0]
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for cv in (bo.BLS12_381, bo.BN254):
        d = {}
        d.update(gen_constants(cv))
        d.update(gen_ntt(cv))
        d.update(gen_msm(cv))
        path = os.path.join(OUT, f"{cv.name}.npz")
        np.savez_compressed(path, **d)
        print(path, len(d), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
