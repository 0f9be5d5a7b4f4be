"""`Prover::prove_with_preprocessed` (proof_system/prover.rs:163-638) end to end on device-resident vectors: every O(n) step is
one of the library's entry points, the transcript is the library's merlin, and the result is a `Proof` whose bytes are the
reference's serialisation (proof.rs:41-103).  The circuit comes in as what the composer and `preprocess.rs` hand the prover:
the four padded wire columns, the public inputs, and a prover key built from selector / sigma / table evaluation vectors.

    round 1  4 iffts, 4 commitments                                              prover.rs:188-226
    round 2  compressed table, query column, combine_split, 4 iffts, 3 commits   prover.rs:228-321   (lookup.py)
    round 3  z and z_2 grand products, 2 iffts, 2 commitments, pi ifft           prover.rs:323-392   (permutation.py)
    round 4  12 + 1 coset ffts, pointwise quotient, coset ifft, 4 commitments    prover.rs:394-481   (quotient.py)
    round 5  23 evaluations, linearisation polynomial, 14 commits, 2 openings    prover.rs:483-618   (linearisation.py, msm.py)

Nothing here is on the timed path of bench.py (its schedule replays the same calls on synthetic inputs); this module is the
proof that the pieces compose into a proof the reference's verifier equations accept (tests/test_prover_gpu.py)."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import linearisation, lookup, permutation, quotient
from .curves import fr_from_mont, fr_to_mont, get_curve
from .domain import Radix2EvaluationDomain
from .transcript import (CUSTOM_EVAL_LABELS, EVAL_LABELS, PROOF_EVAL_FIELDS, ProverTranscript, Transcript, proof_serialize)

SELECTORS = quotient.COLUMNS[12:]       # q_m q_l q_r q_o q_4 q_c q_arith q_range q_logic q_fixed_group_add q_variable_group_add q_lookup
# transcript label of an evaluation -> field of ProofEvaluations (prover.rs:516-544)
_EVAL_FIELD = {"a_eval": "a_eval", "b_eval": "b_eval", "c_eval": "c_eval", "d_eval": "d_eval", "left_sig_eval": "left_sigma_eval",
               "right_sig_eval": "right_sigma_eval", "out_sig_eval": "out_sigma_eval", "perm_eval": "permutation_eval", "f_eval": "f_eval",
               "q_lookup_eval": "q_lookup_eval", "lookup_perm_eval": "z2_next_eval", "h_1_eval": "h1_eval", "h_1_next_eval": "h1_next_eval",
               "h_2_eval": "h2_eval"}
_SEP = {"range separation challenge": "range_challenge", "logic separation challenge": "logic_challenge",
        "fixed base separation challenge": "fixed_base_challenge", "variable base separation challenge": "var_base_challenge",
        "lookup separation challenge": "lookup_challenge", "alpha": "alpha"}


class ProverKey:
    """What `preprocess.rs:138-212` leaves with the prover, device-resident: every selector and sigma polynomial in coefficient
    form and as 4n coset evaluations, the sigma evaluations over the domain, the four table columns."""

    def __init__(self, domain: Radix2EvaluationDomain, domain_4n: Radix2EvaluationDomain, selector_evals: dict, sigma_evals, table_cols):
        if set(selector_evals) != set(SELECTORS) or len(sigma_evals) != 4 or len(table_cols) != 4:
            raise ValueError("12 selector columns, 4 sigma columns and 4 table columns expected")
        self.domain, self.domain_4n = domain, domain_4n
        self.q_lookup_evals = selector_evals["q_lookup"]
        self.polys = {name: domain.ifft(selector_evals[name]) for name in SELECTORS}
        self.evals4n = {name: domain_4n.coset_fft(self.polys[name]) for name in SELECTORS}
        self.sigma_evals = list(sigma_evals)
        self.sigma_polys = [domain.ifft(s) for s in sigma_evals]
        self.sigma4n = [domain_4n.coset_fft(p) for p in self.sigma_polys]
        self.table_cols = list(table_cols)
        self.l1_4n = quotient.l1_coset_evals(domain, domain_4n, selector_evals["q_m"].device)     # depends on n only (the reference rebuilds it per proof, quotient_poly.rs:292-294)

    def with_ctx(self, ctx) -> "ProverKey":
        """The same device-resident key driven from another Context (its own stream / thread) of the same GPU."""
        other = object.__new__(ProverKey)
        other.__dict__.update(self.__dict__)
        other.domain = Radix2EvaluationDomain.new(self.domain.size(), self.domain.curve, ctx)
        other.domain_4n = Radix2EvaluationDomain.new(self.domain_4n.size(), self.domain.curve, ctx)
        return other

    def verifier_key(self, ck) -> dict:
        """The commitments of `preprocess.rs:351-374` + `lookup/preprocess.rs:63-64`: 12 selectors, 4 sigmas, 4 table columns
        (name -> G1Affine; `transcript.seed_transcript` appends 15 of them).  Two batches of commitments over the prover's SRS."""
        names = list(SELECTORS) + ["left_sigma", "right_sigma", "out_sigma", "fourth_sigma"]
        polys = [self.polys[k] for k in SELECTORS] + list(self.sigma_polys)
        out = dict(zip(names, ck.commit_batch(polys)))
        tables = ck.commit_batch([self.domain.ifft(t) for t in self.table_cols])
        out.update({f"table_{k + 1}": tables[k] for k in range(4)})
        return out

    def linearisation_key(self) -> dict:
        k = dict(self.polys)
        k.update(left_sigma=self.sigma_polys[0], right_sigma=self.sigma_polys[1], out_sigma=self.sigma_polys[2], fourth_sigma=self.sigma_polys[3])
        return k


@dataclass
class Proof:
    """`Proof<F, PC>` (proof.rs:41-103)."""
    commitments: dict                       # PROOF_COMMITMENTS name -> G1Affine
    aw_opening: object
    saw_opening: object
    evaluations: dict                       # ProofEvaluations field -> 4 Montgomery limbs (16 fixed + the custom ones)
    challenges: dict = field(default_factory=dict)      # not part of the proof: what the transcript produced
    polys: dict = field(default_factory=dict)           # not part of the proof: the device polynomial behind every commitment / opening
    curve: str = "bls12_381"

    def to_bytes(self) -> bytes:
        from .transcript import PROOF_COMMITMENTS
        return proof_serialize([self.commitments[k] for k in PROOF_COMMITMENTS], [self.aw_opening, self.saw_opening],
                               [self.evaluations[k] for k in PROOF_EVAL_FIELDS], [(k, self.evaluations[k]) for k in CUSTOM_EVAL_LABELS], self.curve)


def prove(pk: ProverKey, ck, wires, public_inputs: dict, preprocessed: Transcript, coeff_a_mont, coeff_d_mont, lean: bool = False) -> Proof:
    """`Prover::prove_with_preprocessed` (prover.rs:163-638) on the device: see `_prove`.  A failure between a round's first
    `commit_begin` and its `round_end` (the "challenges must be different" assertion, an out-of-memory inside a transform) would leave
    the round open on the ctx -- every later blocking commit / open / MSM would return ZK_ERR_PENDING -- so the round is dropped
    (`zk_kzg_round_abort`) before the exception travels on."""
    try:
        return _prove(pk, ck, wires, public_inputs, preprocessed, coeff_a_mont, coeff_d_mont, lean)
    except BaseException:
        try:
            if ck.round_pending():
                ck.round_abort()
        except Exception:
            pass
        raise


def _prove(pk: ProverKey, ck, wires, public_inputs: dict, preprocessed: Transcript, coeff_a_mont, coeff_d_mont, lean: bool = False) -> Proof:
    """wires: the four wire columns padded to n (prover.rs:188-192), device tensors; public_inputs: position -> 4 Montgomery limbs
    (pi.rs:28-36); preprocessed: the transcript after the verifier key was seeded into it; coeff_a / coeff_d: the embedded
    curve's coefficients (`P::COEFF_A`, `P::COEFF_D`).
    lean=False issues the reference's 29 MSMs in the reference's 11 calls.  lean=True produces the SAME proof with 15 MSMs in 5
    calls: the 14 commitments of prover.rs:579,606 are never used (ark-poly-commit 0.3's SonicKZG10 `open` does not read its `commitments` argument (published source; the crate is not in this container), the
    `Proof` holds none of them except z's, which round 3 already has, and the verifier rebuilds them -- proof.rs:343-395), f / h_1 /
    h_2 and z / z_2 have no transcript challenge between them, and the two opening witnesses are known together."""
    import torch
    d, d4 = pk.domain, pk.domain_4n
    cv = get_curve(d.curve)
    n = d.size()
    ctx = d._ctx_for(wires[0])
    tr = ProverTranscript(preprocessed)
    tr.public_inputs(public_inputs)                                                                       # :182
    ch = {"coeff_a": np.asarray(coeff_a_mont, dtype=np.uint64), "coeff_d": np.asarray(coeff_d_mont, dtype=np.uint64)}
    # -- round 1
    w_polys = d.batch(1, list(wires))                                                                     # :196-203
    w_commits = ck.commit_batch(w_polys)                                                                  # :213
    ch.update(tr.round1(w_commits))                                                                       # :217-226
    # -- round 2
    t_ev = lookup.compress_table(pk.table_cols, ch["zeta"], cv, ctx)                                      # :229-237
    table_poly = d.ifft(t_ev)                                                                             # :240-242
    f_ev = lookup.compress_query(pk.q_lookup_evals, list(wires), ch["zeta"], t_ev, n=n, curve=cv, ctx=ctx)   # :244-276
    f_poly = d.ifft(f_ev)                                                                                 # :279-283
    # lean=False: the reference's three PC::commit calls, issued where the reference issues them and collected once -- no challenge
    # is drawn between them (zk_kzg_round_begin_dev ... zk_kzg_round_end; the points and the transcript bytes are the same)
    if not lean:
        ck.commit_begin([f_poly])                                                                         # :289-291
    h1_ev, h2_ev = lookup.combine_split(t_ev, f_ev, cv, ctx)                                              # :295-297
    h1_poly, h2_poly = d.batch(1, [h1_ev, h2_ev])                                                         # :300-305
    if lean:
        f_commit, h1_commit, h2_commit = ck.commit_batch([f_poly, h1_poly, h2_poly])
    else:
        ck.commit_begin([h1_poly])                                                                        # :312-317
        ck.commit_begin([h2_poly])
        f_commit, h1_commit, h2_commit = ck.round_end(3)
    ch.update(tr.round2(f_commit, h1_commit, h2_commit))                                                  # :294,320-337
    for a, b in (("beta", "gamma"), ("beta", "delta"), ("beta", "epsilon"), ("gamma", "delta"), ("gamma", "epsilon"), ("delta", "epsilon")):
        assert not np.array_equal(ch[a], ch[b]), "challenges must be different"                           # :340-345
    # -- round 3
    z_poly = d.ifft(permutation.permutation_evals(d, list(wires), pk.sigma_evals, ch["beta"], ch["gamma"]))     # :347-358
    if not lean:
        ck.commit_begin([z_poly])                                                                         # :361-363
    z2_poly = d.ifft(permutation.lookup_permutation_evals(ctx, cv, f_ev, t_ev, h1_ev, h2_ev, ch["delta"], ch["epsilon"]))   # :370-380
    if lean:
        z_commit, z2_commit = ck.commit_batch([z_poly, z2_poly])
    else:
        ck.commit_begin([z2_poly])                                                                        # :387-389
        z_commit, z2_commit = ck.round_end(2)
    pi_ev = torch.zeros((n, 4), dtype=torch.int64, device=wires[0].device)                                # :392 into_dense_poly
    for pos, v in public_inputs.items():
        pi_ev[pos] = torch.from_numpy(np.asarray(v, dtype=np.uint64).reshape(4).view(np.int64)).to(pi_ev.device)
    pi_poly = d.ifft(pi_ev)
    r3 = tr.round3(z_commit)                                                                              # :366,398-426
    ch.update({_SEP[k]: v for k, v in r3.items()})
    # -- round 4
    q_ch = {name: ch[name] for name in quotient.CHALLENGES}
    t_poly = quotient.compute(d, d4, {"w_l": w_polys[0], "w_r": w_polys[1], "w_o": w_polys[2], "w_4": w_polys[3], "z": z_poly, "z2": z2_poly,
                                      "f": f_poly, "table": table_poly, "h1": h1_poly, "h2": h2_poly, "pi": pi_poly},
                              pk.evals4n, pk.sigma4n, q_ch, l1_4n=pk.l1_4n)                               # :428-453
    t_parts = [t_poly[k * n:(k + 1) * n] for k in range(4)]                                               # :455-456 split_tx_poly
    t_commits = ck.commit_batch(t_parts)                                                                  # :459-469
    ch["z_challenge"] = tr.round4(t_commits)["z"]                                                         # :472-481
    # -- round 5
    lin_poly, ev = linearisation.compute(d, pk.linearisation_key(), {name: ch[name] for name in linearisation.CHALLENGES}, {
        "w_l": w_polys[0], "w_r": w_polys[1], "w_o": w_polys[2], "w_4": w_polys[3], "t_1": t_parts[0], "t_2": t_parts[1], "t_3": t_parts[2],
        "t_4": t_parts[3], "z": z_poly, "z2": z2_poly, "f": f_poly, "h1": h1_poly, "h2": h2_poly, "table": table_poly})   # :483-512
    aw_ch, saw_ch = tr.round5({lb: ev[_EVAL_FIELD[lb]] for lb in EVAL_LABELS}, [(lb, ev[lb]) for lb in CUSTOM_EVAL_LABELS])  # :516-563,593-594
    ch["aw_challenge"], ch["saw_challenge"] = aw_ch, saw_ch
    aw_polys = [lin_poly, pk.sigma_polys[0], pk.sigma_polys[1], pk.sigma_polys[2], f_poly, h2_poly, table_poly]     # :569-577
    # lean=False: the four PC calls of the round (both opening challenges are already drawn) as one deferred round of 16 jobs
    if not lean:
        ck.commit_begin(aw_polys)                                                                         # :579 (the verifier rebuilds these)
    from .msm import kzg_witness
    aw_witness = kzg_witness(aw_polys + list(w_polys), ch["z_challenge"], aw_ch, cv, ctx)                 # :582-591 PC::open =
    if not lean:
        ck.commit_begin([aw_witness], canonical=[True])                                                   #   witness polynomial + its commitment
    saw_polys = [z_poly, w_polys[0], w_polys[1], w_polys[3], h1_poly, z2_poly, table_poly]                # :596-604
    if not lean:
        ck.commit_begin(saw_polys)                                                                        # :606
    zw = fr_to_mont(cv, [fr_from_mont(cv, ch["z_challenge"].reshape(1, 4))[0] * fr_from_mont(cv, np.asarray(d.group_gen()).reshape(1, 4))[0] % cv.r])[0]
    saw_witness = kzg_witness(saw_polys, zw, saw_ch, cv, ctx)                                             # :609-618
    if lean:
        saw_commits = [z_commit]
        aw_opening, saw_opening = ck.commit_batch([aw_witness, saw_witness], canonical=[True, True])
    else:
        ck.commit_begin([saw_witness], canonical=[True])
        last = ck.round_end(16)
        aw_opening, saw_commits, saw_opening = last[7], last[8:15], last[15]
    commitments = {"a_comm": w_commits[0], "b_comm": w_commits[1], "c_comm": w_commits[2], "d_comm": w_commits[3], "z_comm": saw_commits[0],
                   "f_comm": f_commit, "h_1_comm": h1_commit, "h_2_comm": h2_commit, "z_2_comm": z2_commit, "t_1_comm": t_commits[0],
                   "t_2_comm": t_commits[1], "t_3_comm": t_commits[2], "t_4_comm": t_commits[3]}           # :620-637
    assert saw_commits[0] == z_commit
    polys = {"a_comm": w_polys[0], "b_comm": w_polys[1], "c_comm": w_polys[2], "d_comm": w_polys[3], "z_comm": z_poly, "f_comm": f_poly,
             "h_1_comm": h1_poly, "h_2_comm": h2_poly, "z_2_comm": z2_poly, "t_1_comm": t_parts[0], "t_2_comm": t_parts[1],
             "t_3_comm": t_parts[2], "t_4_comm": t_parts[3], "lin": lin_poly, "aw_witness": aw_witness, "saw_witness": saw_witness}
    return Proof(commitments, aw_opening, saw_opening, ev, ch, polys, cv.name)


def check_identity(pk: ProverKey, proof: Proof, public_inputs: dict) -> bool:
    """The equation the verifier's first opening rests on, checked on the polynomials themselves: lin(z) = -r_0 with r_0 as
    `Proof::compute_r0` builds it from the evaluations (proof.rs:427-486).  It holds iff the quotient really is
    (gates + permutation + lookup) / Z_H at z, i.e. (with overwhelming probability over z) iff the witness satisfies the circuit."""
    d = pk.domain
    cv = get_curve(d.curve)
    p, n = cv.r, d.size()
    ctx = d._ctx_for(proof.polys["lin"])
    ch = {k: fr_from_mont(cv, np.asarray(v, dtype=np.uint64).reshape(1, 4))[0] for k, v in proof.challenges.items()}
    ev = {k: fr_from_mont(cv, np.asarray(v, dtype=np.uint64).reshape(1, 4))[0] for k, v in proof.evaluations.items()}
    z = ch["z_challenge"]
    lin_z = fr_from_mont(cv, linearisation.evaluate_batch([proof.polys["lin"]], proof.challenges["z_challenge"].reshape(1, 4), cv, ctx))[0]
    zh = (pow(z, n, p) - 1) % p
    l1 = zh * pow(n * (z - 1) % p, -1, p) % p
    w = fr_from_mont(cv, np.asarray(d.group_gen(), dtype=np.uint64).reshape(1, 4))[0]
    pi_z = 0
    for i, v in public_inputs.items():                                                                    # proof.rs:635-680
        wi = pow(w, i, p)
        pi_z += fr_from_mont(cv, np.asarray(v, dtype=np.uint64).reshape(1, 4))[0] * wi % p * zh % p * pow(n * (z - wi) % p, -1, p)
    al, be, ga, de, ep, ls = ch["alpha"], ch["beta"], ch["gamma"], ch["delta"], ch["epsilon"], ch["lookup_challenge"]
    b = (ev["a_eval"] + be * ev["left_sigma_eval"] + ga) * (ev["b_eval"] + be * ev["right_sigma_eval"] + ga) \
        * (ev["c_eval"] + be * ev["out_sigma_eval"] + ga) * (ev["d_eval"] + ga) * ev["permutation_eval"] * al
    e1d = ep * (1 + de)
    dd = ls * ls * ev["z2_next_eval"] * (e1d + de * ev["h2_eval"]) * (e1d + ev["h2_eval"] + de * ev["h1_next_eval"])
    r0 = (pi_z - b - l1 * al * al - dd - ls ** 3 * l1) % p
    return (lin_z + r0) % p == 0


def _gadget_runs(p: int, base: int, rng, ca: int, cd: int):
    """Short runs of the reference's other gates on host integers, starting at row `base`: six range rows (widget/range.rs:47-63:
    base-4 accumulator chains), six logic rows (widget/logic.rs:65-133: q_c = 1 AND, q_c = -1 XOR on base-4 digits), four
    curve-addition rows (widget/ecc/curve_addition.rs:62-97) and five fixed-base steps (widget/ecc/fixed_base_scalar_mul.rs:88-156)
    over a twisted Edwards curve with coefficients (ca, cd).  Every run ends in an ordinary arithmetic row that takes the run's
    "next row" values in its a, b, d wires.  Returns (cells {(wire, row): value}, selectors {(name, row): value}, gadget rows, end)."""
    inv = lambda x: pow(x % p, -1, p)  # noqa: E731
    fe = lambda: int.from_bytes(rng.bytes(40), "little") % p  # noqa: E731
    cells, sels, gadget = {}, {}, []
    r = base

    def put(row, a=None, b=None, c=None, d=None):
        for w, v in enumerate((a, b, c, d)):
            if v is not None:
                cells[(w, row)] = v % p
    acc = fe()
    for _ in range(6):                                   # range: c - 4d, b - 4c, a - 4b, d_next - 4a are base-4 digits
        q = [int(v) for v in rng.integers(0, 4, 4)]
        d_ = acc
        c_ = 4 * d_ + q[0]
        b_ = 4 * c_ + q[1]
        a_ = 4 * b_ + q[2]
        acc = (4 * a_ + q[3]) % p
        put(r, a_, b_, c_, d_)
        sels[("q_range", r)] = 1
        gadget.append(r)
        r += 1
    put(r, d=acc)
    r += 1
    xa, xb, xd = fe(), fe(), fe()
    for k in range(6):                                   # logic: c holds the product of the two digits
        qa, qb = (int(v) for v in rng.integers(0, 4, 2))
        is_and = k % 2 == 0
        put(r, xa, xb, qa * qb, xd)
        xa, xb = (4 * xa + qa) % p, (4 * xb + qb) % p
        xd = (4 * xd + ((qa & qb) if is_and else (qa ^ qb))) % p
        sels[("q_logic", r)] = 1
        sels[("q_c", r)] = 1 if is_and else p - 1
        gadget.append(r)
        r += 1
    put(r, xa, xb, None, xd)
    r += 1
    for _ in range(4):                                   # curve addition; the next row holds the sum and x1 * y2
        x1, y1, x2, y2 = fe(), fe(), fe(), fe()
        put(r, x1, y1, x2, y2)
        x1y2, y1x2 = x1 * y2 % p, y1 * x2 % p
        x3 = (x1y2 + y1x2) * inv(1 + cd * x1y2 * y1x2) % p
        y3 = (y1 * y2 - ca * x1 * x2) * inv(1 - cd * x1y2 * y1x2) % p
        sels[("q_variable_group_add", r)] = 1
        gadget.append(r)
        r += 1
        put(r, x3, y3, None, x1y2)
        r += 1
    ax, ay, ad = fe(), fe(), fe()
    for bit in (1, 0, p - 1, 1, p - 1):                  # fixed-base step: bit = d_next - 2d in {-1, 0, 1}, c = bit * xy_beta
        xb_, yb_, xyb = fe(), fe(), fe()
        sels[("q_l", r)], sels[("q_r", r)], sels[("q_c", r)], sels[("q_fixed_group_add", r)] = xb_, yb_, xyb, 1
        xy_alpha = bit * xyb % p
        put(r, ax, ay, xy_alpha, ad)
        x_alpha, y_alpha = xb_ * bit % p, (bit * bit * (yb_ - 1) + 1) % p
        t = cd * xy_alpha % p * ax % p * ay % p
        ax, ay = (x_alpha * ay + y_alpha * ax) * inv(1 + t) % p, (y_alpha * ay - ca * x_alpha * ax) * inv(1 - t) % p
        ad = (2 * ad + bit) % p
        gadget.append(r)
        r += 1
    put(r, ax, ay, None, ad)
    r += 1
    return cells, sels, gadget, r


def example_circuit(log_n: int, curve="bls12_381", ctx=None, seed: int = 99, coeffs=(1, 1)):
    """A satisfied circuit of n = 2^log_n rows built ON the device: arithmetic gates with random selectors on about two thirds of
    the rows (q_o = -1, the output wire computed from the others), public inputs on rows 1 and 3, six copy constraints that tie
    cells of different wires and rows (2-cycles in sigma), lookup gates into a four-column table of n/4 distinct rows padded
    with its first row on the remaining third, five all-zero padding rows and -- from 128 rows up -- short runs of range, logic,
    curve-addition and fixed-base gates (`_gadget_runs`; coeffs = the embedded curve's (COEFF_A, COEFF_D) as integers, to be
    passed to `prove` in Montgomery form).  Below 128 rows those four selectors are zero.
    Returns (ProverKey, [w_l, w_r, w_o, w_4], public inputs {row: 4 Montgomery limbs})."""
    import torch
    from ._lib import check, lib
    from .context import default_context
    cv = get_curve(curve)
    cid, p, n = cv.curve_id, cv.r, 1 << log_n
    if n < 32:
        raise ValueError("example_circuit needs at least 32 rows")
    ctx = ctx or default_context(0)
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device=dev).manual_seed(seed)
    dom = Radix2EvaluationDomain.new(n, cv, ctx)
    dom4 = Radix2EvaluationDomain.new(4 * n, cv, ctx)

    def rnd(rows):
        t = torch.randint(0, 1 << 62, (rows, 4), dtype=torch.int64, device=dev, generator=g)
        t[:, 3] &= (1 << 60) - 1                     # < 2^252: a reduced element of either curve's scalar field
        return t

    def mul(x, y):
        o = torch.empty_like(x)
        ctx.use_torch_stream()
        check(lib().zk_fr_mul_dev(ctx.handle, cid, x.data_ptr(), y.data_ptr(), n, o.data_ptr()), "zk_fr_mul_dev")
        return o

    def const(v):
        return torch.from_numpy(fr_to_mont(cv, [v])[0].view(np.int64)).to(dev)

    zero = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    used = n - 5
    row = torch.arange(n, device=dev)
    rows = n // 4
    tsrc = [rnd(n) for _ in range(4)]
    rep = torch.where(row < rows, row, torch.zeros_like(row))
    table = [c[rep].contiguous() for c in tsrc]
    g_base = used - 48
    reserved = (row >= g_base) & (row < used) if n >= 128 else torch.zeros_like(row, dtype=torch.bool)
    is_lookup = (torch.randint(0, 3, (n,), device=dev, generator=g) == 0) & (row > 4) & (row < used) & ~reserved
    live = (~is_lookup) & (row < used)
    pick = torch.randint(0, rows, (n,), device=dev, generator=g)
    lk, lv = is_lookup.unsqueeze(1), live.unsqueeze(1)
    a, b, d = (torch.where(lk, table[k][pick], torch.where(lv, rnd(n), zero)).contiguous() for k in (0, 1, 3))
    x_poly = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    x_poly[1] = const(1)
    omega_i = dom.fft(x_poly)                        # the evaluations of X over the domain: omega^i
    ks = (1, linearisation.K1, linearisation.K2, linearisation.K3)
    sigma = [linearisation.lincomb([omega_i], fr_to_mont(cv, [ks[k]]), curve=cv, ctx=ctx).clone() for k in range(4)]
    free = torch.nonzero(live & (row >= 5)).flatten()[:12].tolist()
    cols = {0: a, 1: b, 3: d}
    for t in range(0, len(free) - 1, 2):
        i, j = free[t], free[t + 1]
        (w1, r1), (w2, r2) = ((0, i), (1, j)) if t % 4 == 0 else ((3, i), (0, j))      # a_i = b_j, then d_i = a_j
        cols[w2][r2] = cols[w1][r1]
        s1, s2 = sigma[w1][r1].clone(), sigma[w2][r2].clone()
        sigma[w1][r1], sigma[w2][r2] = s2, s1
    sel = {name: zero.clone() for name in SELECTORS}
    c_fixed = {}
    if n >= 128:                                     # the other gates: cells and selectors computed on host integers, written row by row
        cells, gsel, gadget, end = _gadget_runs(p, g_base, np.random.default_rng(seed), int(coeffs[0]) % p, int(coeffs[1]) % p)
        assert end <= used
        wires_abd = {0: a, 1: b, 3: d}
        for (w, r_), v in cells.items():
            if w == 2:
                c_fixed[r_] = v
            else:
                wires_abd[w][r_] = const(v)
        grows = torch.tensor(gadget, device=dev)
        live[grows] = False                          # gadget rows carry no arithmetic gate
        lv = live.unsqueeze(1)
    for name in ("q_m", "q_l", "q_r", "q_4", "q_c"):
        sel[name] = torch.where(lv, rnd(n), zero).contiguous()
    sel["q_o"] = torch.where(lv, const(p - 1).expand(n, 4), zero).contiguous()
    sel["q_arith"] = torch.where(lv, const(1).expand(n, 4), zero).contiguous()
    if n >= 128:
        for (name, r_), v in gsel.items():
            sel[name][r_] = const(v)
    sel["q_lookup"] = torch.where(lk, const(1).expand(n, 4), zero).contiguous()
    pub = {1: rnd(1)[0].cpu().numpy().view(np.uint64), 3: rnd(1)[0].cpu().numpy().view(np.uint64)}
    pi = zero.clone()
    for i, v in pub.items():
        pi[i] = torch.from_numpy(v.view(np.int64)).to(dev)
    c_arith = linearisation.lincomb([mul(sel["q_m"], mul(a, b)), mul(sel["q_l"], a), mul(sel["q_r"], b), mul(sel["q_4"], d), sel["q_c"], pi],
                                    fr_to_mont(cv, [1] * 6), curve=cv, ctx=ctx)
    c = torch.where(lk, table[2][pick], c_arith).contiguous()
    for r_, v in c_fixed.items():
        c[r_] = const(v)
    return ProverKey(dom, dom4, sel, sigma, table), [a, b, c, d], pub
