#!/usr/bin/env python3
"""Times prover.prove (reference call structure and lean) on prover.example_circuit at n = 2^LOG; under rocprofv3 --kernel-trace
--stats the kernel totals say how much of the wall time is not kernels."""
import sys
import time

import torch

import ark_plonk_amd as zk
from ark_plonk_amd import prover, transcript

sys.path.insert(0, ".")
import bench  # noqa: E402  (build_srs)

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << log_n
ctx = zk.Context(0)
cv = zk.get_curve(0)
pk, wires, pub = prover.example_circuit(log_n, cv, ctx)
ck = zk.CommitterKey(bench.build_srs(ctx, cv, n, 0, n, torch), cv, ctx)
ck.precompute()
pre = transcript.seed_transcript(transcript.Transcript(b"probe", cv), pk.verifier_key(ck), n)
one = zk.curves.fr_to_mont(cv, [1])[0]
for lean in (False, True):
    prover.prove(pk, ck, wires, pub, pre, one, one, lean=lean)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        prover.prove(pk, ck, wires, pub, pre, one, one, lean=lean)
    torch.cuda.synchronize()
    print(f"lean={lean}: {(time.perf_counter() - t0) / reps * 1e3:.1f} ms per proof", flush=True)
