"""The unchanged caller, T of them: T host threads, each with its own zk_ctx and its own proof, run the drop-in schedule (31 zk_ntt +
9 zk_kzg_commit_batch + 2 zk_kzg_open per proof on pageable host vectors) against ONE GPU and one shared SRS -- what a proving service
that runs the reference's `Prover::prove` in T worker threads would do.  One caller's PCIe transfers run under another caller's kernels.
usage: python tools/drop_in_callers.py [log_n] [proofs per caller]          prints proofs/s for T = 1, 2, 3, 4, residency cache off / on"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import ark_plonk_amd as zk
    from ark_plonk_amd.prover_schedule import DropInSchedule
    from bench import build_srs

    def digest(points):
        import hashlib
        return hashlib.sha256(b"".join(p.xy().tobytes() + bytes([p.infinity]) for p in points)).hexdigest()
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n = 1 << log_n
    ctx0 = zk.Context(0)
    ctx0.use_torch_stream()
    cv = zk.get_curve("bls12_381")
    srs = build_srs(ctx0, cv, n, 0, n, torch).cpu().numpy().view(np.uint64)
    ck0 = zk.CommitterKey(srs, cv, ctx0).precompute()
    ref = digest(DropInSchedule(log_n, ctx0, ck0, cv).run_once(proof_id=0))
    for T in [int(x) for x in os.environ.get("CALLERS", "1,2,3,4").split(",")]:
        ctxs = [zk.Context(0) for _ in range(T)]
        cks = [zk.CommitterKey(srs, cv, c) for c in ctxs]              # PC::trim in every worker: a lookup of the resident SRS + table
        scheds = [DropInSchedule(log_n, c, ck, cv) for c, ck in zip(ctxs, cks)]
        for on in (False, True):
            for c in ctxs:
                c.set_residency_cache(on, (2 << 30) // T if on else 0, 0)
            for s in scheds:
                for _ in range(5 if on else 1):
                    s.run_once()
            bar = threading.Barrier(T + 1)
            errs = []

            def worker(s):
                try:
                    bar.wait()
                    for _ in range(k):
                        s.run_once()
                    torch.cuda.synchronize()
                except Exception as e:      # noqa: BLE001
                    errs.append(repr(e))
                finally:
                    bar.wait()
            ths = [threading.Thread(target=worker, args=(s,)) for s in scheds]
            for t in ths:
                t.start()
            bar.wait()
            t0 = time.perf_counter()
            bar.wait()
            dt = time.perf_counter() - t0
            for t in ths:
                t.join()
            same = all(digest(s.run_once(proof_id=0)) == ref for s in scheds)
            print(f"callers {T}  residency cache {'on ' if on else 'off'}: {T * k / dt:6.2f} proofs/s  ({dt / k * 1e3:7.1f} ms per proof per caller)  "
                  f"same 29 points as one caller: {same}  {'errors ' + str(errs) if errs else ''}", flush=True)
        for c in ctxs:
            c.set_residency_cache(False)
        for ck in cks:
            ck.close()
        for c in ctxs:
            c.close()
    ck0.close()
    ctx0.close()


if __name__ == "__main__":
    main()
