"""The legs of `bench.py --extra-legs all` that lie OUTSIDE SURVEY.md section 8's hot path (they earn no coverage and no
longer ride in the default run): the BenchCircuit-shaped data, the O(n) glue of the prover's rounds inside the synthetic step, and a real
proof of a satisfied circuit end to end.  Each function takes the namespace `E` bench.py builds for them (its locals: args, zk, torch, ctx,
cv, log_n, steps, value, r, dev, timed_region, acc_per_msm, build_srs, new_ctx) and returns the leg's object."""
from __future__ import annotations

import time


def data_benchcircuit(E):
    timed_region, acc_per_msm, value = E.timed_region, E.acc_per_msm, E.value
    # SURVEY.md 8d config 2's "realistic" vector on this very binary: wire columns as benches/plonk.rs' BenchCircuit builds them
    # (composer.rs:493-548: periodic {6, 7, -20, 1} / {-20, 6, 7, 0} rows + 3 blinding rows, zero-padded) -- data-independence
    k2 = 3
    rc_ = timed_region(False, 1, k2, warmup=1, data="benchcircuit")
    return {"proofs_per_s": k2 / rc_["dt"], "ms_per_proof": rc_["dt"] / k2 * 1e3, "steps": k2, "accumulate_ms_per_msm": acc_per_msm(rc_),
            "vs_uniform": (k2 / rc_["dt"]) / value, "commitments_sha256": rc_["digest"],
            "what": "--data benchcircuit: the same schedule over BenchCircuit-shaped wire columns (benches/plonk.rs:53-62); other inputs as in the headline"}


def with_device_glue(E):
    timed_region, steps = E.timed_region, E.steps
    k2 = max(2, min(steps, 3))
    r5 = timed_region(False, 1, k2, warmup=1, glue=True)
    return {"proofs_per_s": k2 / r5["dt"], "ms_per_proof": r5["dt"] / k2 * 1e3,
            "quotient_ms_per_proof": r5["prof"]["quotient"][0] / max(r5["kb"], 1), "grand_product_ms_per_proof": r5["prof"]["grand_product"][0] / max(r5["kb"], 1),
            "evaluations_ms_per_proof": r5["prof"]["poly_evaluate"][0] / max(r5["kb"], 1),
            "linearisation_ms_per_proof": r5["prof"]["poly_lincomb"][0] / max(r5["kb"], 1),
            "lookup_round2_ms_per_proof": (r5["prof"]["lookup_query"][0] + r5["prof"]["lookup_combine_split"][0]) / max(r5["kb"], 1),
            "what": "SURVEY.md 8f N1 + N2 and the O(n) work of rounds 2 and 5 inside the step: the compressed table / query columns and "
                    "h_1, h_2 (zk_lookup_query_dev, zk_lookup_combine_split_dev; prover.rs:228-317), z and z2 built on the device (zk_perm_product_dev / "
                    "zk_lookup_product_dev), the 4n quotient evaluations computed on the device (zk_quotient_evals_dev) from the 12 coset-fft "
                    "outputs, the 23 evaluations of the proof (zk_poly_evaluate_dev) and the 19-term linearisation polynomial "
                    "(zk_poly_lincomb_dev; linearisation_poly.rs:164-350) -- instead of synthetic inputs / a stand-in polynomial"}


def full_proof(E):
    args, zk, torch, ctx, cv, log_n, steps, dev, build_srs, new_ctx = E.args, E.zk, E.torch, E.ctx, E.cv, E.log_n, E.steps, E.dev, E.build_srs, E.new_ctx
    # a REAL proof: a satisfied circuit (arithmetic, range, logic, ECC and lookup gates, public inputs, copy constraints) built on the device, proved by
    # ark_plonk_amd/prover.py -- Prover::prove_with_preprocessed's five rounds with every O(n) step through the C ABI, challenges from
    # the library's merlin transcript -- and serialised; self-check: the verifier's identity lin(z) = -r_0 on the result
    from ark_plonk_amd import prover, transcript
    n = 1 << log_n
    pk, wires, pub = prover.example_circuit(log_n, cv, ctx)
    ckp = zk.CommitterKey(build_srs(ctx, cv, n, 0, n, torch), cv, ctx)
    ckp.precompute(args.table_window)
    pre = transcript.seed_transcript(transcript.Transcript(b"bench", cv), pk.verifier_key(ckp), n)   # Circuit::compile's part of the transcript
    one = zk.curves.fr_to_mont(cv, [1])[0]
    a = (pk, ckp, wires, pub, pre, one, one)
    prover.prove(*a)
    torch.cuda.synchronize()
    k2 = max(2, min(steps, 5))
    t0 = time.perf_counter()
    for _ in range(k2):
        proof = prover.prove(*a)
    torch.cuda.synchronize()
    dtp = time.perf_counter() - t0
    ok = prover.check_identity(pk, proof, pub)
    data = proof.to_bytes()
    nbytes = len(data)
    prover.prove(*a, lean=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k2):
        lean = prover.prove(*a, lean=True)
    torch.cuda.synchronize()
    dtl = time.perf_counter() - t0
    same = lean.to_bytes() == data
    # the same lean proofs, three in flight (a thread + zk_ctx + HIP stream each over ONE prover key, SRS and window table)
    import threading
    S3 = 3
    lanes3 = []
    for i in range(S3):
        cx = ctx if i == 0 else new_ctx(dev)
        st3 = torch.cuda.current_stream() if i == 0 else torch.cuda.Stream()
        lanes3.append((cx, st3, (pk if i == 0 else pk.with_ctx(cx), ckp if i == 0 else ckp.with_ctx(cx)) + a[2:]))
    for cx, st3, a3 in lanes3[1:]:
        with torch.cuda.stream(st3):
            prover.prove(*a3, lean=True)
    torch.cuda.synchronize()
    gate, errs3, outs3 = threading.Barrier(S3 + 1), [], [None] * S3

    def lane3(i):
        cx, st3, a3 = lanes3[i]
        try:
            with torch.cuda.stream(st3):
                gate.wait()
                for _ in range(k2):
                    outs3[i] = prover.prove(*a3, lean=True)
                st3.synchronize()
        except Exception as e:
            errs3.append(e)
            gate.abort()
    th3 = [threading.Thread(target=lane3, args=(i,)) for i in range(S3)]
    for t in th3:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in th3:
        t.join()
    torch.cuda.synchronize()
    dt3 = time.perf_counter() - t0
    if errs3:
        raise errs3[0]
    same3 = all(o.to_bytes() == data for o in outs3)
    for cx, _, _ in lanes3[1:]:
        cx.close()
    ckp.close()
    return {"proofs_per_s": k2 / dtp, "ms_per_proof": dtp / k2 * 1e3, "proof_bytes": nbytes, "verifier_identity_holds": bool(ok),
            "lean": {"proofs_per_s": k2 / dtl, "ms_per_proof": dtl / k2 * 1e3, "msms": 15, "same_proof_bytes": bool(same),
                     "three_in_flight": {"proofs_per_s": S3 * k2 / dt3, "ms_per_proof_aggregate": dt3 / (S3 * k2) * 1e3, "same_proof_bytes": bool(same3)},
                     "how": "the 14 commitments of prover.rs:579,606 are used by nobody (SonicKZG10's open does not read them, the Proof holds none but z's, "
                            "the verifier rebuilds them): 15 MSMs in 5 calls instead of 29 in 11, identical bytes"},
            "what": "a satisfied circuit of 2^%d rows proved end to end on the device (31 NTTs, 29 MSMs, round-2 lookup multisets, both grand "
                    "products, the pointwise quotient, 23 evaluations, the linearisation polynomial, merlin transcript, proof bytes); "
                    "tests/test_prover_gpu.py checks such proofs against the reference verifier's equations" % log_n}
