"""How a round's tail compares between its two forms ("finish the window combine on the device ... measure it once"):
  host   zk_kzg_round_end: the last kernel writes 64 (S_v, T_v) pairs per job to pinned memory; the host pool adds them (~230 point
         operations per job) and normalises (one Fermat inversion per job)
  device zk_kzg_round_end_partial_dev + zk_g1_sum_partials_dev(1 rank): one more quad-cooperative launch forms every job's sum on the
         device, one more writes it out in the arkworks layout; the host only normalises
Small vectors (2^13), so that the tail is most of the call.  usage: python tools/round_tail_probe.py [jobs ...]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import ark_plonk_amd as zk
    from bench import build_srs
    jobs_list = [int(a) for a in sys.argv[1:]] or [1, 4, 16]
    ctx = zk.Context(0)
    ctx.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    n = 1 << 13
    ck = zk.CommitterKey(build_srs(ctx, cv, n, 0, n, torch), cv, ctx).precompute()
    g = torch.Generator(device="cuda").manual_seed(5)
    polys = [torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g) for _ in range(16)]
    buf = torch.zeros((16, ck.partial_dev_words()), dtype=torch.int64, device="cuda")
    print("n = 2^13 per job; median of 30 rounds, microseconds from the first begin to the results")
    for jobs in jobs_list:
        def host():
            for p in polys[:jobs]:
                ck.commit_begin([p])
            return ck.round_end(jobs)

        def device():
            for p in polys[:jobs]:
                ck.commit_begin([p])
            ck.round_end_partial_dev(buf, jobs)
            return ck.sum_partials_dev(buf[:jobs].reshape(1, -1), 1, jobs)

        assert host() == device()
        res = {}
        for name, fn in (("host", host), ("device", device)):
            ts = []
            for _ in range(30):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e6)
            res[name] = statistics.median(ts)
        print(f"  {jobs:2d} jobs: host tail {res['host']:8.1f} us   device tail {res['device']:8.1f} us   difference {res['device'] - res['host']:+8.1f} us")
    ck.close()
    ctx.close()


if __name__ == "__main__":
    main()
