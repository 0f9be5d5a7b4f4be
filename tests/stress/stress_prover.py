"""Randomised end-to-end check of the device-resident prover (ark_plonk_amd/prover.py) against the restated reference verifier
(oracle/verifier_oracle.py): random satisfied circuits of 2^5 .. 2^MAX rows on both curves (tests/test_prover_gpu.py's builder,
random seeds: different gate / lookup / copy-constraint layouts and public inputs) must verify, in the reference's call structure
and in the lean one (same bytes); the same circuit with one wrong cell must be rejected.
usage: python tests/stress/stress_prover.py [seconds] [seed]   (also collected, with a short budget, by tests/test_stress_gpu.py)"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ark_plonk_amd as zk  # noqa: E402
from ark_plonk_amd import prover, transcript  # noqa: E402
from ark_plonk_amd.curves import fr_to_mont  # noqa: E402
from oracle import bigint_oracle as bo  # noqa: E402
from oracle import cpu  # noqa: E402
from oracle import verifier_oracle as vo  # noqa: E402
from oracle import wire_oracle as wo  # noqa: E402


def _helpers():
    spec = importlib.util.spec_from_file_location("test_prover_gpu", os.path.join(ROOT, "tests", "test_prover_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(budget: float = 120.0, seed: int = 7, ctx=None, max_log_n: int = 11):
    from tests.conftest import TAU, srs_from_powers, tau_powers
    cpu.build()
    H = _helpers()
    own = ctx is None
    if own:
        ctx = zk.Context(0)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    accepted = rejected = 0
    cks = {}
    while time.time() < t_end:
        cid = int(rng.integers(0, 2))
        log_n = int(rng.integers(5, max_log_n + 1))
        n = 1 << log_n
        cv = bo.CURVES[cid]
        broken = bool(rng.random() < 0.3)
        ca, cd = bo.seeded_scalars(cv, int(rng.integers(1, 1 << 20)), 2)
        sel, sigma, table, wires, pub = H.build_circuit(cv, log_n, int(rng.integers(1, 1 << 30)), break_cell=broken, coeffs=(ca, cd))
        dom = zk.Radix2EvaluationDomain.new(n, cid, ctx)
        dom4 = zk.Radix2EvaluationDomain.new(4 * n, cid, ctx)
        pk = prover.ProverKey(dom, dom4, {k: H.dev(cid, v) for k, v in sel.items()}, [H.dev(cid, s) for s in sigma], [H.dev(cid, t) for t in table])
        if (cid, log_n) not in cks:
            pw_canon, _ = tau_powers(cpu, cid, n + 8)
            cks[(cid, log_n)] = zk.CommitterKey(srs_from_powers(ctx, cid, pw_canon), cid, ctx)
        ck = cks[(cid, log_n)]
        label = b"stress %d" % int(rng.integers(0, 1000))
        vk = pk.verifier_key(ck)
        pre = transcript.seed_transcript(transcript.Transcript(label, cid), vk, n)
        args = (pk, ck, [H.dev(cid, w) for w in wires], {i: fr_to_mont(cid, [v])[0] for i, v in pub.items()}, pre, fr_to_mont(cid, [ca])[0],
                fr_to_mont(cid, [cd])[0])
        lean = bool(rng.integers(0, 2))
        proof = prover.prove(*args, lean=lean)
        data = proof.to_bytes()
        t = vo.seed_transcript(cv, wo.PlonkTranscript(label, cv), H.oracle_points(cid, vk), n)
        ok, _, det = vo.verify_with_trapdoor(cv, log_n, data, t, pub, H.dlogs(cid, ctx, pk, proof), TAU, ca, cd)
        assert ok == (not broken), (cid, log_n, broken, lean, det["aw"], det["saw"])
        assert prover.check_identity(pk, proof, args[3]) == (not broken)
        if not broken and rng.random() < 0.3:
            assert prover.prove(*args, lean=not lean).to_bytes() == data
        accepted += not broken
        rejected += broken
    for ck in cks.values():
        ck.close()
    if own:
        ctx.close()
    print(f"stress ok: {accepted} proofs of random satisfied circuits accepted by the restated verifier, {rejected} broken witnesses rejected "
          f"(seed {seed}, {budget:.0f} s)", flush=True)
    return accepted + rejected


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
