"""Randomised cross-check of the opening witness (RLC + synthetic division on the device, csrc/kzg.hip) against the
CPU restatement: 1..16 polynomials of random ragged lengths (1 .. 2^17, around the 64-coefficient chunk and the
1024-lane scan boundaries), both curves.  Compares the witness scalars (canonical) limb for limb.
usage: python tests/stress/stress_kzg.py [seconds]   (also collected, with a short budget, by tests/test_stress_gpu.py)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from oracle import cpu  # noqa: E402



def run(budget: float = 90.0, seed: int = 3, ctx=None, max_len: int = 1 << 17):
    cpu.build()
    own = ctx is None
    if own:
        ctx = zk.Context(0)
    ctx.use_torch_stream()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    checks = 0

    def rnd_fr(cid, k):
        x = rng.integers(0, 1 << 62, size=(k, 4), dtype=np.uint64)
        if cid == 1:
            x[:, 3] >>= np.uint64(2)
        return x

    while time.time() < t_end:
        cid = int(rng.integers(0, 2))
        m = int(rng.choice([1, 2, 63, 64, 65, 127, 4096, 65535, 65536, 65537, max_len, int(rng.integers(1, max_len))]))
        m = min(m, max_len)
        k = int(rng.integers(1, 17))
        lens = [m] + [int(rng.integers(1, m + 1)) for _ in range(k - 1)]
        rng.shuffle(lens)
        polys = [rnd_fr(cid, ln) for ln in lens]
        z, chi = rnd_fr(cid, 1)[0], rnd_fr(cid, 1)[0]
        comb = np.zeros((m, 4), dtype=np.uint64)
        chi_pow = cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))
        for p in polys:
            term = cpu.fr_op(cid, "mul", p, np.repeat(chi_pow, p.shape[0], axis=0))
            comb[: p.shape[0]] = cpu.fr_op(cid, "add", comb[: p.shape[0]], term)
            chi_pow = cpu.fr_op(cid, "mul", chi_pow, chi.reshape(1, 4))
        exp = cpu.kzg_witness(cid, comb, z)                                       # Montgomery coefficients of the witness
        exp = cpu.convert(cid, "fr", False, exp) if exp.shape[0] else exp         # the device hands out into_repr values
        got = zk.kzg_witness([torch.from_numpy(p.view(np.int64)).cuda() for p in polys], z, chi, cid, ctx)
        got = got.cpu().numpy().view(np.uint64)
        assert got.shape[0] == max(m - 1, 0) and np.array_equal(got, exp[: got.shape[0]]), (cid, m, lens)
        checks += 1
    if own:
        ctx.close()
    print(f"stress ok: {checks} opening witnesses checked against the CPU restatement (seed {seed}, {budget:.0f} s)", flush=True)
    return checks


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 90.0, int(os.environ.get("SEED", "3")))
