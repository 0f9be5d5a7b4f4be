"""Randomised cross-check of the kernels behind prover rounds 2 and 5 (csrc/lookup.hip, csrc/kzg.hip: zk_poly_evaluate_dev,
zk_poly_lincomb_dev, zk_lookup_combine_split_dev) against CPU restatements: batches of 1..32 polynomials of random ragged
lengths evaluated at random points (p(z) = p_0 + z w_0 with w from ark_cpu.cpp's synthetic division) and summed with random
coefficients (vectorised Fr ops of the C++ restatement); tables with random amounts of repetition and padding against query
columns with a random share of dummy rows (`MultiSet::combine_split` as a Python dict over the 32-byte values), both curves.
usage: python tests/stress/stress_rounds.py [seconds]   (also collected, with a short budget, by tests/test_stress_gpu.py)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import linearisation, lookup  # noqa: E402
from oracle import bigint_oracle as bo  # noqa: E402
from oracle import cpu  # noqa: E402


def run(budget: float = 90.0, seed: int = 5, ctx=None, max_len: int = 1 << 17):
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

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1, 4)).cuda()

    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 16383, 16384, 16385, 65537]
    while time.time() < t_end:
        cid = int(rng.integers(0, 2))
        # -- evaluations and sums
        k = int(rng.integers(1, 33))
        lens = [min(max_len, int(rng.choice(sizes + [int(rng.integers(1, max_len + 1))]))) for _ in range(k)]
        polys = [rnd_fr(cid, m) for m in lens]
        pts = rnd_fr(cid, k)
        if k > 2:
            pts[0] = 0
            pts[1] = cpu.convert(cid, "fr", True, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
        d_polys = [dev(p) for p in polys]
        got = linearisation.evaluate_batch(d_polys, pts, cid, ctx)
        for j, (p, z) in enumerate(zip(polys, pts)):
            exp = p[0].copy()
            if p.shape[0] > 1:
                w = cpu.kzg_witness(cid, p, z)
                exp = cpu.fr_op(cid, "add", p[:1], cpu.fr_op(cid, "mul", w[:1], z.reshape(1, 4)))[0]
            assert np.array_equal(got[j], exp), ("evaluate", cid, lens, j)
        cf = rnd_fr(cid, k)
        out_len = int(rng.choice([max(lens), max(1, max(lens) // 2), max(lens) + 7]))
        want = np.zeros((out_len, 4), dtype=np.uint64)
        for p, c in zip(polys, cf):
            m = min(p.shape[0], out_len)
            want[:m] = cpu.fr_op(cid, "add", want[:m], cpu.fr_op(cid, "mul", p[:m], np.repeat(c.reshape(1, 4), m, axis=0)))
        comb = linearisation.lincomb(d_polys, cf, out_len=out_len, curve=cid, ctx=ctx).cpu().numpy().view(np.uint64)
        assert np.array_equal(comb, want), ("lincomb", cid, lens, out_len)
        checks += k + 1
        # -- combine_split
        n_t = min(max_len, int(rng.choice(sizes + [int(rng.integers(1, max_len + 1))])))
        n_f = int(rng.integers(0, 2 * n_t + 1)) if rng.random() < 0.8 else n_t
        distinct = max(1, int(n_t * rng.choice([0.001, 0.05, 0.25, 1.0])))
        pool = rnd_fr(cid, distinct)
        order = rng.integers(0, distinct, n_t) if rng.random() < 0.5 else np.minimum(np.arange(n_t), distinct - 1) % distinct
        t = pool[order]
        pick = rng.integers(0, n_t, n_f)
        pick[rng.random(n_f) < rng.choice([0.0, 0.5, 0.95])] = 0
        f = t[pick]
        key = lambda a: [bytes(r) for r in np.ascontiguousarray(a).view(np.uint8).reshape(-1, 32)]  # noqa: E731
        w1, w2 = bo.combine_split(key(t), key(f))
        h1, h2 = lookup.combine_split(dev(t), dev(f) if n_f else torch.zeros((0, 4), dtype=torch.int64, device="cuda"), cid, ctx)
        assert key(h1.cpu().numpy()) == w1 and key(h2.cpu().numpy()) == w2, ("combine_split", cid, n_t, n_f, distinct)
        checks += 1
    if own:
        ctx.close()
    print(f"stress ok: {checks} evaluations / sums / multiset splits checked against the CPU restatements (seed {seed}, {budget:.0f} s)", flush=True)
    return checks


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 90.0, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
