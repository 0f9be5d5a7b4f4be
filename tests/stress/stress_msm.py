"""Randomised cross-check of the commit paths against the CPU restatement: random lengths (odd / even / threshold
neighbours), random batch shapes (blocking batches, host-pointer batches, deferred rounds cut at random places), uniform and
heavily skewed scalars, table and per-window paths.
usage: [SEED=..] [MAX_LOG_N=17] python tests/stress/stress_msm.py [seconds]   (also collected, with a short budget, by tests/test_stress_gpu.py)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import _lib  # noqa: E402
from oracle import cpu  # noqa: E402



def run(budget: float = 120.0, seed: int = 1, ctx=None, max_log_n: int = 17):
    cpu.build()
    own = ctx is None
    if own:
        ctx = zk.Context(0)
    ctx.use_torch_stream()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = checks = 0
    while time.time() < t_end:
        cid = int(rng.integers(0, 2))
        cv = zk.get_curve(cid)
        log_n = int(rng.integers(13, max_log_n))
        n = 1 << log_n
        ks = np.zeros((n, 4), dtype=np.uint64)
        ks[:, 0] = rng.integers(1, 1 << 40, size=n, dtype=np.uint64)
        bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, torch.from_numpy(ks.view(np.int64)).cuda().data_ptr(), n, bases.data_ptr()))
        bases_h = bases.cpu().numpy().view(np.uint64)
        ck = zk.CommitterKey(bases, cid, ctx)
        if rng.random() < 0.8:
            # default window, or the forms the default only takes at 2^19 points and above: c = 17 (folded scalars on BLS12-381,
            # int32 sort), and the wide reduction (c = 18 .. 21; 20 is the default from 2^22 points on)
            ck.precompute(int(rng.choice([0, 0, 17, 17, 18, 20, 20, 19, 21])))
        k = int(rng.integers(1, 8))
        polys = []
        for _ in range(k):
            ln = min(n, int(rng.choice([n, n - 1, n - 2, 8192, 8193, 8191, int(rng.integers(1, n + 1))])))
            p = rng.integers(0, 1 << 62, size=(ln, 4), dtype=np.uint64)
            if cid == 1:
                p[:, 3] >>= np.uint64(2)      # BN254: r ~ 2^253.6, keep the Montgomery residues canonical (< 2^252)
            mode = rng.random()
            if mode < 0.2:      # heavy skew: most coefficients equal
                p[rng.random(ln) < 0.9] = p[0]
            elif mode < 0.3:    # sparse
                p[rng.random(ln) < 0.95] = 0
            polys.append(p)
        how = rng.random()
        if how < 0.25:
            got = ck.commit_batch(polys)          # host-pointer batch (zk_kzg_commit_batch: staged uploads under the MSMs)
        elif how < 0.6:
            # deferred round (zk_kzg_round_begin_dev ... zk_kzg_round_end): the batch cut into 1..k calls at random places
            d_polys = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
            cuts = sorted(set(int(c) for c in rng.integers(1, k + 1, size=int(rng.integers(0, 3))) if c < k))
            lo = 0
            for hi in cuts + [k]:
                ck.commit_begin(d_polys[lo:hi])
                lo = hi
            got = ck.round_end(k)
        else:
            got = ck.commit_batch([torch.from_numpy(p.view(np.int64)).cuda() for p in polys])
        for p, g in zip(polys, got):
            exp_xy, exp_inf = cpu.kzg_commit(cid, bases_h, p)
            assert g.infinity == bool(exp_inf) and np.array_equal(g.xy(), exp_xy), (cid, n, len(p))
            checks += 1
        ck.close()
        rounds += 1
    if own:
        ctx.close()
    msg = f"stress ok: {rounds} rounds, {checks} commitments checked against the CPU restatement (seed {seed}, {budget:.0f} s)"
    print(msg, flush=True)
    return checks


if __name__ == "__main__":
    # MAX_LOG_N=21 reaches the sizes where the table defaults to 17-bit windows and msm_accumulate to three rounds of lanes
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(os.environ.get("SEED", "1")), max_log_n=int(os.environ.get("MAX_LOG_N", "17")))
