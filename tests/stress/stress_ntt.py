"""Randomised cross-check of the transforms against the CPU restatement: random log N (1..20), kind, input length
(empty, short, a quarter, full), both curves, host and device entry points, in place and out of place.
usage: python tests/stress/stress_ntt.py [seconds]   (also collected, with a short budget, by tests/test_stress_gpu.py)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from oracle import cpu  # noqa: E402



def run(budget: float = 90.0, seed: int = 2, ctx=None, max_log_n: int = 20):
    cpu.build()
    own = ctx is None
    if own:
        ctx = zk.Context(0)
    ctx.use_torch_stream()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    checks = 0
    while time.time() < t_end:
        cid = int(rng.integers(0, 2))
        log_n = int(rng.integers(1, max_log_n + 1))
        n = 1 << log_n
        kind = int(rng.integers(0, 4))
        in_len = int(rng.choice([0, 1, n // 4, n // 4 + 1, n - 1, n, int(rng.integers(0, n + 1))]))
        x = rng.integers(0, 1 << 62, size=(in_len, 4), dtype=np.uint64)
        if cid == 1 and in_len:
            x[:, 3] >>= np.uint64(2)          # BN254: r ~ 2^253.6, keep the residues canonical (< 2^252)
        exp = cpu.ntt(cid, kind, log_n, x)
        dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
        if rng.random() < 0.3:
            got = dom._run(kind, x)                                    # host buffers (pinned staging ring)
        else:
            d = torch.from_numpy(x.view(np.int64)).cuda()
            if in_len == n and rng.random() < 0.5:
                got = dom._run(kind, d, out=d).cpu().numpy().view(np.uint64)   # in place
            else:
                got = dom._run(kind, d).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, exp), (cid, log_n, kind, in_len)
        checks += 1
    if own:
        ctx.close()
    print(f"stress ok: {checks} transforms checked against the CPU restatement (seed {seed}, {budget:.0f} s)", flush=True)
    return checks


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 90.0, int(os.environ.get("SEED", "2")))
