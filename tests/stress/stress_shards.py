"""Randomised cross-check of the sharded forms of a round of commitments (`zk_srs_precompute_rows`, `zk_kzg_round_end_partial`,
`zk_g1_sum_partials_batch`, `zk_kzg_round_end_winsums_dev`, `zk_g1_sum_winsums_dev`) against the CPU restatement: G "ranks" played one after the other on one card, G in 2..8,
  * by points  -- rank g owns SRS[g n/G, (g+1) n/G) and that slice of every coefficient vector (ragged vectors leave late ranks short or
                  empty), its own whole window table;
  * by windows -- rank g owns rows g, g+G, ... of the window table over the whole SRS (c = 16 or 17) and sees whole vectors;
and the exchange in its two forms: Jacobian partials through the host (`round_end_partial` + `sum_partials_batch`), or every job's
virtual-window sums left on the device -- a (G, jobs x words) tensor as the all-gather would leave it -- and added element-wise
(`round_end_winsums_dev`, `sum_winsums_dev`).
The sum over the ranks must be the commitment of the CPU restatement, limb for limb.
usage: [SEED=..] [MAX_LOG_N=17] python tests/stress/stress_shards.py [seconds]   (a short budget runs under tests/test_stress_gpu.py)"""
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


def run(budget: float = 120.0, seed: int = 3, ctx=None, max_log_n: int = 17):
    cpu.build()
    own = ctx is None
    if own:
        ctx = zk.Context(0)
    ctx.use_torch_stream()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    rounds = checks = 0
    seen = set()
    while time.time() < t_end:
        cid = int(rng.integers(0, 2))
        cv = zk.get_curve(cid)
        log_n = int(rng.integers(14, max_log_n + 1))
        n = 1 << log_n
        G = int(rng.choice([2, 3, 4, 5, 8]))
        axis = "windows" if rng.random() < 0.5 else "points"
        form = str(rng.choice(["host", "winsums"]))
        on_device = form != "host"
        ks = np.zeros((n, 4), dtype=np.uint64)
        ks[:, 0] = rng.integers(1, 1 << 40, size=n, dtype=np.uint64)
        bases = torch.empty((n, 2 * cv.fq_limbs), dtype=torch.int64, device="cuda")
        _lib.check(_lib.lib().zk_g1_fixed_base_batch_dev(ctx.handle, cid, torch.from_numpy(ks.view(np.int64)).cuda().data_ptr(), n, bases.data_ptr()))
        bases_h = bases.cpu().numpy().view(np.uint64)
        k = int(rng.integers(1, 6))
        polys = []
        for _ in range(k):
            ln = min(n, int(rng.choice([n, n - 1, n // 2 + 1, n // G, 8193, int(rng.integers(1, n + 1))])))
            p = rng.integers(0, 1 << 62, size=(ln, 4), dtype=np.uint64)
            if cid == 1:
                p[:, 3] >>= np.uint64(2)
            mode = rng.random()
            if mode < 0.15:
                p[rng.random(ln) < 0.9] = p[0]
            elif mode < 0.25:
                p[rng.random(ln) < 0.95] = 0
            elif mode < 0.3:
                p[:] = 0                                  # the point at infinity on every rank
            polys.append(p)
        d_polys = [torch.from_numpy(p.view(np.int64)).cuda() for p in polys]
        c_bits = int(rng.choice([16, 17])) if axis == "windows" else int(rng.choice([0, 17]))
        host_parts, dev_parts, words = [], [], None
        for g in range(G):
            if axis == "points":
                lo, hi = g * n // G, (g + 1) * n // G
                ck = zk.CommitterKey(bases[lo:hi].contiguous(), cid, ctx)
                if hi - lo >= 8192 or form == "winsums" or rng.random() < 0.5:      # the window-sum form needs every rank's table (one geometry)
                    ck.precompute(c_bits)
                mine = [p[lo:max(lo, min(hi, p.shape[0]))] for p in d_polys]
            else:
                ck = zk.CommitterKey(bases, cid, ctx).precompute(c_bits, rows=(g, G))
                a, b, r = ck.table_rows()
                assert (a, b) == (g, G), (a, b, r, g, G)
                mine = d_polys
            live = [q for q in mine if q.shape[0] > 0]
            if on_device:
                words = ck.winsums_dev_words()
                assert words > 0, (form, axis, c_bits)
                buf = torch.zeros((k, words), dtype=torch.int64, device="cuda")       # all-zero row = the point at infinity (empty shard)
                if live:
                    sub = torch.empty((len(live), words), dtype=torch.int64, device="cuda")
                    for q in live:
                        ck.commit_begin([q])
                    if rng.random() < 0.5:
                        ck.round_reduce_winsums_dev(sub)
                    ck.round_end_winsums_dev(sub, len(live))
                    idx = torch.tensor([j for j, q in enumerate(mine) if q.shape[0] > 0], dtype=torch.int64, device="cuda")
                    buf.index_copy_(0, idx, sub)
                dev_parts.append(buf.reshape(-1))
            else:
                part = np.zeros((k, 3 * cv.fq_limbs), dtype=np.uint64)                  # Z = 0: infinity
                if live:
                    got = ck.commit_batch_partial(live)
                    part[[j for j, q in enumerate(mine) if q.shape[0] > 0]] = got
                host_parts.append(part)
            keep = ck          # the summing call needs any key of the curve on this ctx
            if g < G - 1:
                ck.close()
        if on_device:
            got = keep.sum_winsums_dev(torch.stack(dev_parts), G, k)
        else:
            got = zk.sum_partials_batch(np.stack(host_parts), cid)
        keep.close()
        for p, pt in zip(polys, got):
            exp_xy, exp_inf = cpu.kzg_commit(cid, bases_h, p)
            assert pt.infinity == bool(exp_inf) and np.array_equal(pt.xy(), exp_xy), (cid, n, G, axis, form, c_bits, len(p))
            checks += 1
        seen.add((axis, form))
        rounds += 1
    if own:
        ctx.close()
    print(f"stress ok: {rounds} sharded rounds ({sorted(seen)}), {checks} commitments summed over 2..8 ranks equal the CPU restatement "
          f"(seed {seed}, {budget:.0f} s)", flush=True)
    return checks


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(os.environ.get("SEED", "3")), max_log_n=int(os.environ.get("MAX_LOG_N", "17")))
