"""The hot-path call schedule of one `Prover::prove_with_preprocessed` (plonk-core/src/proof_system/
prover.rs:163-638), replayed through the C ABI on device-resident data.

Only the transforms and commitments are reproduced -- the same kinds, sizes and order the reference
issues them: 13 ifft(n) + 4 fft(n) + 13 coset_fft(4n) + 1 coset_ifft(4n) and 29 MSMs
of ~n points, in the reference's eleven PC::commit / PC::open calls (4 | 1 | 1 | 1 | 1 | 1 | 4 | 7 | 1 | 7 | 1
polynomials, prover.rs:213,289,312,315,361,387,459,579,582,606,609).  Calls whose inputs do not depend on each
other's results -- f | h_1 | h_2 (no challenge is drawn between them, prover.rs:289-317), z | z_2 (prover.rs:361-389)
and the four calls of the last round (prover.rs:579-618) -- are issued through the deferred form of the ABI
(zk_kzg_round_begin_dev / zk_kzg_open_begin_dev ... zk_kzg_round_end): still eleven calls and 29 MSMs with identical
results, but the host blocks five times per proof instead of eleven and the bucket reductions of a group run as one
launch per kernel (`defer_calls=False` blocks in every call, as an unchanged `Prover::prove` does).  The serial CPU glue between
the calls (transcript, linearisation) is out of scope, so polynomial *values* are synthetic unless the
optional device builders of SURVEY.md 8f are switched on (grand_products: z / z2; quotient: the 4n
quotient evaluations); what is preserved is which buffers feed which call and their lengths, e.g. the
coset_fft inputs have n coefficients zero-extended to 4n (quotient_poly.rs:72-120) and the openings
are MSMs of n-1.

Multi-GPU (one process per GPU): every MSM is sharded by points over the ranks -- rank g owns
SRS[g*n/G, (g+1)*n/G) -- and combined by an all-gather of the 3L-limb Jacobian partials (RCCL over
xGMI); the NTTs are replicated per rank (they stay single-GPU, BASELINE.json north_star).
"""
from __future__ import annotations

import numpy as np

from .curves import get_curve
from .domain import Radix2EvaluationDomain
from .msm import CommitterKey, sum_partials_batch


def all_gather_partials(dist, parts: np.ndarray, world: int, device) -> np.ndarray:
    """The one exchange step of the sharded MSM: every rank's (jobs, 3L) Jacobian partials -> (world, jobs, 3L) on every rank.
    `device`: where the collective's tensors live -- the rank's GPU for RCCL (backend "nccl": 144 bytes per job over xGMI), the
    CPU for gloo.  One small H2D, one all_gather, one D2H per group of PC calls (5 per proof); the G-way sums are host work on
    a few hundred bytes (zk_g1_sum_partials_batch)."""
    import torch
    jobs, l3 = parts.shape
    mine = torch.from_numpy(np.ascontiguousarray(parts).view(np.int64).reshape(-1)).to(device, non_blocking=True)
    out = torch.empty((world, jobs * l3), dtype=torch.int64, device=device)
    if hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(out.view(-1), mine)
    else:
        dist.all_gather(list(out.unbind(0)), mine)
    return out.cpu().numpy().view(np.uint64).reshape(world, jobs, l3)


def all_gather_partials_dev(dist, mine, world: int):
    """The same exchange with the partials never leaving the device: `mine` is the int64 device tensor the last
    reduction kernel of the round wrote (jobs x window-sum words, zk_kzg_round_end_winsums_dev), the result the (world, jobs x words)
    device tensor zk_g1_sum_winsums_dev reads.  RCCL (backend "nccl") gathers device tensors directly: no host copy between the
    last kernel and the collective.  gloo has no device all_gather: the rehearsal backend stages through the CPU here."""
    import torch
    flat = mine.reshape(-1)
    if dist.get_backend() == "nccl":
        out = torch.empty((world, flat.numel()), dtype=torch.int64, device=mine.device)
        if hasattr(dist, "all_gather_into_tensor"):
            dist.all_gather_into_tensor(out.view(-1), flat)
        else:
            dist.all_gather(list(out.unbind(0)), flat)
        return out
    host = flat.cpu()
    out = torch.empty((world, host.numel()), dtype=torch.int64)
    if hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(out.view(-1), host)
    else:
        dist.all_gather(list(out.unbind(0)), host)
    return out.to(mine.device)


class ProofSchedule:
    def __init__(self, log_n: int, ctx, ck: CommitterKey, curve="bls12_381", rank: int = 0, world: int = 1,
                 dist=None, seed: int = 0x5EED0000, dedup=False,
                 grand_products: bool = False, quotient: bool = False, fuse_round5: bool = False, data: str = "uniform",
                 ntt_batch: bool = True, linearisation: bool = False, lookup_round2: bool = False, defer_calls: bool = True,
                 hoist: bool = True, shard_axis: str = "points", partials_on_device: bool | None = None, exchange: str | None = None):
        import torch
        self.torch = torch
        self.cv = get_curve(curve)
        self.log_n = log_n
        self.n = 1 << log_n
        self.ctx = ctx
        self.ck = ck
        self.rank, self.world, self.dist = rank, world, dist
        # commitments cached by polynomial label, so the 12 polynomials the reference
        # commits a second time in round 5 (prover.rs:569-607) cost no MSM: 29 -> 17 per proof (20 on the first,
        # which also commits the prover key's sigma polynomials), same outputs
        # dedup: False | True ("label": the Python-side label cache of round 1, kept for comparison) | "abi" (the library's
        # content-addressed commitment cache, zk_ctx_set_commit_cache: what an unchanged Prover::prove gets)
        self.dedup = dedup is True or dedup == "label"
        self.dedup_abi = dedup == "abi"
        if self.dedup_abi:
            ctx.set_commit_cache(True)
        # prover.rs:579-618 issues PC::commit(aw) / PC::open / PC::commit(saw) / PC::open as four calls.  The default
        # replays them as four batches -- what a drop-in PC implementation sees.  All 16 MSMs depend only on values
        # known before the first call (both opening challenges are drawn with no transcript append in between), so
        # a prover that merges the four calls can run them as ONE batch: fuse_round5=True (needs that caller change).
        self.fuse_round5 = fuse_round5
        # eleven PC calls, five host waits: see the module docstring.  False = every call blocks (the drop-in shape).
        self.defer_calls = defer_calls
        # hoist: transforms whose inputs do not depend on a round's challenge are queued BEHIND that round's reductions, before the
        # host waits for them (zk_kzg_round_reduce): the sigma ffts (prover-key data, permutation/mod.rs:671-674) behind round 1, the
        # public-input and L_1 iffts (pi.rs:115, quotient_poly.rs:325) behind round 2, the twelve coset ffts of the quotient round
        # (quotient_poly.rs:72-120: wires, z, z_2, f, table, h_1, h_2 are all committed before alpha is drawn) behind round 3.  Same
        # transforms, same sizes, same batches; they run while the host combines the window sums, normalises and hashes.  (Putting them
        # on a second stream, and dealing a round's jobs to two streams, were measured even or slower in round 4 and retired in round 6:
        # profiles/design_history_msm.md.)
        self.hoist = bool(hoist)
        self._pending = []          # per open call: ("q", n_jobs) queued in the ABI's round | ("r", [points]) already computed
        # z and z2 evaluation vectors built on the device from the wire / sigma /
        # lookup columns (permutation/mod.rs:652-822) instead of taken as synthetic inputs
        self.grand_products = grand_products
        # the 4n quotient evaluations computed on the device from the 13 coset-FFT outputs
        # and (synthetic) prover-key evaluations instead of taken as a synthetic input
        self.quotient = quotient
        # round 5 before its commitments (linearisation_poly.rs:164-350): the 23 evaluations of the proof and the 19-term
        # linearisation polynomial computed on the device from the round's polynomials and (synthetic) prover-key
        # polynomials, instead of a stand-in polynomial being committed as `lin`
        self.linearisation = linearisation
        self.last_evals = None
        # round 2 (prover.rs:228-317): the compressed table, the compressed query column and h_1 / h_2 (MultiSet::combine_split)
        # built on the device from four table columns, q_lookup and the wire columns, instead of taken as synthetic inputs
        self.lookup_round2 = lookup_round2
        # Independent transforms the reference issues back to back go out as ONE zk_ntt_batch_dev (one launch per pass,
        # blockIdx.y = polynomial): the four wire iffts (prover.rs:196-203), h_1 / h_2 (prover.rs:302-305), the four sigma ffts
        # (permutation/mod.rs:671-674) and the twelve coset ffts of quotient_poly.rs:72-120.  ntt_batch=False issues them one by one,
        # as a patched ark-poly (which sees one fft_in_place at a time) would.
        self.ntt_batch = ntt_batch
        self._cache = {}
        self.msms_run = 0
        self._cur_id = 0
        self._next_id = 0
        self.dom_n = Radix2EvaluationDomain.new(self.n, curve, ctx)
        self.dom_4n = Radix2EvaluationDomain.new(4 * self.n, curve, ctx)
        dev = torch.device("cuda", ctx.device)
        g = torch.Generator(device=dev).manual_seed(seed)
        n = self.n

        def rnd(rows):
            # uniform 254-bit residues (< r for both curves' 4-limb Fr): valid Montgomery elements
            t = torch.randint(0, 1 << 62, (rows, 4), dtype=torch.int64, device=dev, generator=g)
            return t

        # 17 distinct polynomials of the proof: w_l w_r w_o w_4 | f h1 h2 | z z2 | t1..t4 | lin table | 2 witnesses
        self.evals = [rnd(n) for _ in range(4)]          # wire evaluations (prover.rs:188-192)
        if data == "benchcircuit":
            # the wire columns of benches/plonk.rs' BenchCircuit -- 3 random blinding rows
            # (composer.rs:580-596), then add_dummy_constraints pairs (composer.rs:493-548: rows (6, 7, -20, 1) and
            # (-20, 6, 7, 0)) up to 2^(log_n-1) + 2 gates, zero-padded to n (preprocess.rs:61-88)
            from .curves import fr_to_mont
            pat = fr_to_mont(self.cv, [6, 7, self.cv.r - 20, 1, self.cv.r - 20, 6, 7, 0])
            rows = torch.from_numpy(pat.view(np.int64)).to(dev).view(2, 4, 4)     # [row of the pair][wire][limb]
            gates = min(n - 4, n // 2 + 2)
            for k in range(4):
                col = torch.zeros((n, 4), dtype=torch.int64, device=dev)
                col[:3] = self.evals[k][:3]
                body = rows[:, k, :].repeat((gates + 1) // 2, 1)[:gates]
                col[3:3 + gates] = body
                self.evals[k] = col
        elif data != "uniform":
            raise ValueError("data: 'uniform' or 'benchcircuit'")
        self.aux_evals = [rnd(n) for _ in range(9)]      # table, f, h1, h2, z, z2, pi, l1, l1*alpha^2 evaluation vectors
        self.sigma = [rnd(n) for _ in range(4)]          # sigma polynomials (coefficients, from the prover key)
        self.coef = [torch.empty((n, 4), dtype=torch.int64, device=dev) for _ in range(13)]
        self.ev4n = torch.empty((4 * n, 4), dtype=torch.int64, device=dev)
        self.quot = rnd(4 * n)                            # quotient evaluations over the coset
        self.scratch_n = torch.empty((n, 4), dtype=torch.int64, device=dev)
        from .quotient import COLUMNS
        if quotient or ntt_batch:     # the twelve coset-fft outputs live side by side (a batch writes them in one launch)
            self.cos = {name: torch.empty((4 * n, 4), dtype=torch.int64, device=dev) for name in COLUMNS[:12]}
        if quotient:
            self.key4n = {name: rnd(4 * n) for name in COLUMNS[12:]}     # selector evaluations of the prover key over the coset
            self.sigma4n = [rnd(4 * n) for _ in range(4)]
            self.q_chal = {name: np.array([0x1111 * (k + 1), 0x2222, 0x3333, 0x0444], dtype=np.uint64)
                           for k, name in enumerate(__import__("ark_plonk_amd.quotient", fromlist=["CHALLENGES"]).CHALLENGES)}
        if lookup_round2:
            # a table of n / 4 distinct rows padded to n with its first row (repeats adjacent, as a padded table's are), lookup
            # gates on about half of the rows (never row 0, which _set_proof varies per proof); a lookup row's wires hold a
            # table row, as a satisfied circuit's do
            self.table_cols = [rnd(n) for _ in range(4)]
            rep = torch.arange(n, device=dev)
            rep[max(n // 4, 1):] = 0
            self.table_cols = [col[rep].contiguous() for col in self.table_cols]
            q = torch.randint(0, 2, (n,), device=dev, generator=g)
            q[0] = 0
            self.q_lookup = torch.zeros((n, 4), dtype=torch.int64, device=dev)
            self.q_lookup[:, 0] = q                       # any non-zero value selects (prover.rs:265 `is_zero`)
            row = torch.randint(0, n, (n,), device=dev, generator=g)
            for k in range(4):
                self.evals[k] = torch.where(q.bool().unsqueeze(1), self.table_cols[k][row], self.evals[k]).contiguous()
            self.zeta_mont = np.array([0x0badcafe, 0x12345678, 0x9abcdef0, 0x01234567], dtype=np.uint64)
        if linearisation:
            from . import linearisation as lin_mod
            self.key_polys = {name: rnd(n) for name in lin_mod.KEY_POLYS[:12]}       # selector polynomials of the prover key
            for k, name in enumerate(lin_mod.KEY_POLYS[12:]):
                self.key_polys[name] = self.sigma[k]
            self.lin_chal = {name: np.array([0x1357 * (k + 1), 0x2468, 0x3579, 0x0123], dtype=np.uint64)
                             for k, name in enumerate(lin_mod.CHALLENGES)}
        # shard of the MSMs this rank owns.  "points" (SURVEY.md 8e's preferred axis): SRS[lo, hi) and the same slice of every
        # polynomial.  "windows" (BASELINE.json north_star's wording): the whole SRS and every whole polynomial, but only the table rows
        # of the windows rank, rank + world, ... (the key was built with CommitterKey.precompute(rows=(rank, world))): the rank sorts
        # all n scalars' digits for its windows into its own bucket set.  The partial and the collective are the same either way.
        if shard_axis not in ("points", "windows"):
            raise ValueError("shard_axis: 'points' or 'windows'")
        self.shard_axis = shard_axis
        if shard_axis == "windows" and world > 1:
            if ck.table_rows()[:2] != (rank, world):
                raise ValueError("window sharding needs a key precomputed with rows=(rank, world)")
            self.lo, self.hi = 0, n
        else:
            self.lo = rank * n // world
            self.hi = (rank + 1) * n // world
        # The exchange of a sharded round (world > 1), two forms, same points:
        #   "winsums"            every job's 2 VW virtual-window sums, written by the reduction kernel the single-GPU path ends with, straight
        #                        into the collective's send buffer; all-gathered (32 KiB per job); added element-wise by one kernel; ONE
        #                        host wait, the combine per job on the host pool (zk_kzg_round_end_winsums_dev / zk_g1_sum_winsums_dev)
        #   "host" (default)     window sums to the host, host combine, H2D, all-gather of 3L-limb Jacobians, D2H, host sum
        # Default = the faster on ONE card (profiles/r05/r05_sim_rank.txt: "host" by 1-2 % over "winsums"); on a node the device form saves
        # a host round trip before and after every collective -- bench.py times both there.  `partials_on_device=True` without
        # `exchange` selects "winsums".  (A third form -- one point per job formed by a further dependent launch -- measured last on one
        # card in round 5 and was retired in round 6: profiles/design_history_msm.md.)
        # The device forms need the deferred rounds, a table with c <= 17 and no commitment cache; otherwise "host" is used.
        if exchange is None:
            exchange = "winsums" if partials_on_device is True else "host"
        if exchange not in ("winsums", "host"):
            raise ValueError("exchange: 'winsums' or 'host'")
        geom = ck.winsums_geometry() if world > 1 else None
        device_form_ok = world > 1 and defer_calls and not (self.dedup or self.dedup_abi) and geom is not None
        if world > 1 and dist is not None and hasattr(dist, "all_gather_object"):
            # once per schedule, on EVERY rank (a rank that skipped this collective would leave the others waiting in it): element-wise sums
            # of window sums mean something only if every rank reduces in the same geometry, and a rank whose table has no device form
            # (20-bit windows: shards of 2^22 points and more) takes everybody to the host form
            seen = [None] * world
            dist.all_gather_object(seen, (exchange, bool(device_form_ok)) + tuple(geom or ()))
            if not all(s_[1] for s_ in seen):
                device_form_ok = False
            elif len(set(seen)) != 1:
                raise RuntimeError(f"ranks disagree on the exchange form / table geometry (window bits, windows, virtual windows, buckets): {seen}")
        if not device_form_ok:
            exchange = "host"
        self.exchange = exchange
        self.partials_on_device = exchange != "host"
        if self.partials_on_device:
            self._pw = ck.winsums_dev_words()
            self._pbuf = torch.zeros((16, self._pw), dtype=torch.int64, device=dev)       # partials of the library's pending jobs
            self._pfull = torch.zeros((16, self._pw), dtype=torch.int64, device=dev)      # ... with all-zero rows (infinity) for empty shards
        self.collectives = 0
        self.points = []
        # evaluation point and opening challenge (transcript outputs in the reference): fixed field elements
        self.z_mont = np.array([0x1234567, 0x89abcdef, 0x13579bdf, 0x0fedcba9], dtype=np.uint64)
        self.chi_mont = np.array([0x2468ace, 0x7654321, 0x2222222, 0x0111111], dtype=np.uint64)

    # -- the commitments of one prover round: submitted back to back, collected together (the
    #    transcript needs them only at the end of the round)
    def _commit_round(self, polys, canonical=None, labels=None):
        self._round_begin(polys, canonical, labels)
        return self._round_end()

    def _immediate(self):
        # the label cache and the ABI's commitment cache answer at once; so does a schedule asked to block in every call
        return self.dedup or self.dedup_abi or not self.defer_calls

    def _round_begin(self, polys, canonical=None, labels=None):
        """One PC::commit call of the reference: queued in the open round (or computed at once, see _immediate)."""
        polys = list(polys)
        if self._immediate():
            self._pending.append(("r", self._commit_now(polys, canonical, labels)))
            return
        self.msms_run += len(polys)
        if self.world == 1:
            self.ck.commit_begin(polys, canonical=canonical)
        else:
            # sharded: this rank's slice of every polynomial
            for sl, kd in zip(self._slices(polys), canonical or [False] * len(polys)):
                if sl.shape[0]:
                    self.ck.commit_begin([sl], canonical=[kd])
                    self._pending.append(("q", 1))
                else:
                    self._pending.append(("z", 1))      # nothing of this polynomial falls into the rank's shard
            return
        self._pending.append(("q", len(polys)))

    def _open_begin(self, polys, label):
        """One PC::open call of the reference (prover.rs:582-591, 609-618)."""
        if self._immediate() or self.world > 1:
            from .msm import kzg_witness
            # the witness polynomial is computed by every rank (replicated, like the NTTs); its MSM shards
            w = kzg_witness(polys, self.z_mont, self.chi_mont, self.cv.curve_id, self.ctx)
            self._round_begin([w], canonical=[True], labels=[label])
            return
        self.msms_run += 1
        self.ck.open_begin(polys, self.z_mont, self.chi_mont)
        self._pending.append(("q", 1))

    def _hoisting(self):
        return self.hoist and not self._immediate()

    def _round_reduce(self):
        """Queue the open round's reductions now; what is launched until `_round_end` runs behind them, under the host's part."""
        if not self._immediate() and any(kind == "q" for kind, _ in self._pending):
            if self.world > 1 and self.exchange == "winsums":
                self.ck.round_reduce_winsums_dev(self._pbuf)
            else:
                self.ck.round_reduce()

    def _round_end(self):
        """Close the open round: the points of every call since the last close, in call order."""
        pend, self._pending = self._pending, []
        nq = sum(n for kind, n in pend if kind == "q")
        got = []
        if self.world == 1:
            if nq:
                got = self.ck.round_end(nq)
        elif any(kind != "r" for kind, _ in pend) and self.partials_on_device:
            got = self._gather_dev(nq, pend)
        elif any(kind != "r" for kind, _ in pend):
            L3 = 3 * self.cv.fq_limbs
            got = self._gather(self.ck.round_end_partial(nq) if nq else np.zeros((0, L3), dtype=np.uint64), pend)
        out, q = [], 0
        for kind, v in pend:
            if kind == "r":
                out += v
            else:                     # world > 1: one entry per job ("q" queued, "z" empty shard), `got` holds both
                out += got[q:q + v]
                q += v
        return out

    def _slices(self, polys):
        return [p[self.lo:max(self.lo, min(self.hi, p.shape[0]))] for p in polys]

    def _gather(self, parts, pend):
        """Sharded round: ONE all-gather of every job's 3L-limb Jacobian partial, then the G-way sums."""
        L3 = 3 * self.cv.fq_limbs
        jobs = [kind for kind, _ in pend if kind != "r"]
        full = np.zeros((len(jobs), L3), dtype=np.uint64)         # Z = 0: the empty shard's partial is the point at infinity
        q = 0
        for k, kind in enumerate(jobs):
            if kind == "q":
                full[k] = parts[q]
                q += 1
        return self._all_gather_sum(full)

    def _gather_dev(self, nq, pend):
        """Sharded round, device form: the library leaves the queued jobs' 2 VW window sums in `_pbuf` (no host wait), ONE all-gather of
        the device tensor, one summing kernel.  Jobs whose shard is empty ("z") are all-zero rows = the point at infinity (every window
        sum infinite)."""
        jobs = [kind for kind, _ in pend if kind != "r"]
        if nq:
            self.ck.round_end_winsums_dev(self._pbuf, nq)
        if nq == len(jobs):
            mine = self._pbuf[:nq]
        else:
            self._pfull.zero_()
            idx = self.torch.tensor([k for k, kind in enumerate(jobs) if kind == "q"], dtype=self.torch.int64, device=self._pbuf.device)
            if nq:
                self._pfull.index_copy_(0, idx, self._pbuf[:nq])
            mine = self._pfull[:len(jobs)]
        allp = all_gather_partials_dev(self.dist, mine, self.world)
        self.collectives += 1
        return self.ck.sum_winsums_dev(allp, self.world, len(jobs))

    def _all_gather_sum(self, parts):
        self.collectives += 1
        dev = self.torch.device("cuda", self.ctx.device) if self.dist.get_backend() == "nccl" else self.torch.device("cpu")
        allp = all_gather_partials(self.dist, parts, self.world, dev)
        return sum_partials_batch(allp, self.cv.curve_id)

    def _commit_now(self, polys, canonical=None, labels=None):
        if self.dedup and labels is not None:
            todo = [k for k, lb in enumerate(labels) if lb not in self._cache and lb not in labels[:k]]
            if todo:
                got = self._commit_now([polys[k] for k in todo], None if canonical is None else [canonical[k] for k in todo])
                for k, pt in zip(todo, got):
                    self._cache[labels[k]] = pt
            return [self._cache[lb] for lb in labels]
        self.msms_run += len(polys)
        if self.world == 1:
            if self.dedup_abi:
                before = self.ctx.commit_cache_stats()["hits"]
                res = self.ck.commit_batch(polys, canonical=canonical)
                self.msms_run -= self.ctx.commit_cache_stats()["hits"] - before
                return res
            return self.ck.commit_batch(polys, canonical=canonical)
        # sharded: this rank's slice of every polynomial, one fused batch, ONE all-gather for the call
        L3 = 3 * self.cv.fq_limbs
        slices = self._slices(polys)
        if all(sl.shape[0] > 0 for sl in slices):
            parts = self.ck.commit_batch_partial(slices, canonical=canonical)
        else:
            kinds = canonical or [False] * len(polys)
            parts = np.stack([self.ck.commit_batch_partial([sl], canonical=[kd])[0] if sl.shape[0] else np.zeros(L3, dtype=np.uint64)
                              for sl, kd in zip(slices, kinds)])
        return self._all_gather_sum(parts)

    def _set_proof(self, proof_id):
        """Every proof has its own witness: proof k's evaluation vectors are the seeded ones with k added to their first
        limb, so no two proofs commit the same polynomials (a content-addressed commitment cache must only find what the
        reference really repeats: the 12 re-commits of round 5 and the prover key's sigma polynomials)."""
        if proof_id is None:
            proof_id = self._next_id
            self._next_id += 1
        d = proof_id - self._cur_id
        if d:
            for t in self.evals + self.aux_evals + [self.quot]:
                t[0, 0] += d
        self._cur_id = proof_id

    def run_once(self, proof_id=None):
        """One proof's hot path.  Returns the 29 commitments/openings (G1Affine) in call order.
        proof_id: which synthetic witness (None = the next one; pass the same id to reproduce a proof).
        A failure inside an open round drops the round (zk_kzg_round_abort) and the schedule's own count of its jobs, so that the
        ctx takes blocking calls again and the next proof starts clean."""
        try:
            return self._run_once(proof_id)
        except BaseException:
            self._pending = []
            try:
                if self.ck.round_pending():
                    self.ck.round_abort()
            except Exception:
                pass
            raise

    def _run_once(self, proof_id=None):
        self._set_proof(proof_id)
        d, d4, n = self.dom_n, self.dom_4n, self.n
        out = []
        # commitments of prover-key polynomials (sigma_1..3) outlive a proof; everything else is per proof
        self._cache = {k: v for k, v in self._cache.items() if k.startswith("sigma")}
        self.msms_run = 0
        c = self.coef
        # Round 1: 4 ifft + 4 commits (prover.rs:196-203, 213)
        if self.ntt_batch:
            c[0:4] = d.batch(1, self.evals)
        else:
            for i in range(4):
                c[i] = d.ifft(self.evals[i])
        hoist = self._hoisting()
        sig = None
        self._round_begin(c[:4], labels=["w_l", "w_r", "w_o", "w_4"])
        if hoist:
            self._round_reduce()
            sig = d.batch(0, self.sigma) if self.ntt_batch else [d.fft(self.sigma[i]) for i in range(4)]     # of round 3
        out += self._round_end()
        # Round 2: table ifft, f ifft + commit, h1/h2 ifft + commits (prover.rs:240-242,281-291,302-317)
        t_ev, f_ev, h1_ev, h2_ev = self.aux_evals[0:4]
        if self.lookup_round2:
            from . import lookup
            t_ev = lookup.compress_table(self.table_cols, self.zeta_mont, self.cv, self.ctx)                          # prover.rs:229-237
            f_ev = lookup.compress_query(self.q_lookup, self.evals, self.zeta_mont, t_ev, curve=self.cv, ctx=self.ctx)   # :244-276
        c[4] = d.ifft(t_ev)                       # table_poly
        c[5] = d.ifft(f_ev)                       # f_poly
        self._round_begin([c[5]], labels=["f"])                # PC::commit(f): prover.rs:289; collected with h_1 / h_2 below
        if self.lookup_round2:
            h1_ev, h2_ev = lookup.combine_split(t_ev, f_ev, self.cv, self.ctx)                                        # :295-297
        if self.ntt_batch:
            c[6], c[7] = d.batch(1, [h1_ev, h2_ev])
        else:
            c[6] = d.ifft(h1_ev)                      # h1
            c[7] = d.ifft(h2_ev)                      # h2
        self._round_begin([c[6]], labels=["h1"])              # two PC::commit calls of one polynomial each (prover.rs:312-317)
        self._round_begin([c[7]], labels=["h2"])
        if hoist:
            self._round_reduce()
            c[10], c[11] = d.ifft(self.aux_evals[6]), d.ifft(self.aux_evals[7])     # pi (of round 3), l1 (of round 4)
        out += self._round_end()                              # f, h_1, h_2 enter the transcript before beta is drawn (prover.rs:320)
        # Round 3: sigma ffts, z ifft + commit, z2 ifft + commit, pi ifft (permutation/mod.rs:671-674,751,800; pi.rs:115)
        if sig is None:
            sig = d.batch(0, self.sigma) if self.ntt_batch else [d.fft(self.sigma[i]) for i in range(4)]
        z_evals, z2_evals = self.aux_evals[4], self.aux_evals[5]
        if self.grand_products:
            from . import permutation
            z_evals = permutation.permutation_evals(d, self.evals, sig, self.chi_mont, self.z_mont)       # beta, gamma
        c[8] = d.ifft(z_evals)                    # z
        self._round_begin([c[8]], labels=["z"])               # prover.rs:361
        if self.grand_products:
            z2_evals = permutation.lookup_permutation_evals(self.ctx, self.cv, f_ev, t_ev, h1_ev, h2_ev, self.chi_mont, self.z_mont)  # delta, epsilon
        c[9] = d.ifft(z2_evals)                   # z2
        self._round_begin([c[9]], labels=["z2"])              # prover.rs:387
        names = ("l1", "z", "w_l", "w_r", "w_o", "w_4", "z2", "f", "table", "h1", "h2", "pi")

        def quotient_coset_ffts(dn, d4n):
            qpolys = (c[11], c[8], c[0], c[1], c[2], c[3], c[9], c[5], c[4], c[6], c[7], c[10])
            if self.ntt_batch:
                d4n.batch(2, qpolys, outs=[self.cos[name] for name in names])          # coset_fft, n coefficients zero-extended to 4n
            else:
                for name, poly in zip(names, qpolys):
                    d4n._run(2, poly, out=self.cos[name] if self.quotient else self.ev4n)

        if hoist:
            self._round_reduce()
            quotient_coset_ffts(d, d4)                        # of round 4: none of the twelve inputs depends on alpha
        out += self._round_end()                              # z, z_2 enter the transcript before alpha is drawn (prover.rs:398)
        # Round 4: quotient (quotient_poly.rs:71-120,205,292-294,175-177)
        if not hoist:
            c[10] = d.ifft(self.aux_evals[6])         # pi
            c[11] = d.ifft(self.aux_evals[7])         # l1
            quotient_coset_ffts(d, d4)
        c[12] = d.ifft(self.aux_evals[8])         # l1 * alpha^2
        d4._run(2, c[12], out=self.ev4n)
        quot = self.quot
        if self.quotient:
            from .quotient import compute_quotient_evals
            quot = compute_quotient_evals(d, {**self.cos, **self.key4n}, self.sigma4n, self.q_chal)
        t = d4.coset_ifft(quot)                   # quotient polynomial, 4n coefficients
        out += self._commit_round([t[i * n:(i + 1) * n] for i in range(4)], labels=["t1", "t2", "t3", "t4"])   # t_1..t_4 (prover.rs:455-469)
        # Round 5 (prover.rs:569-618): aw commits (7), opening of the 7 + 4 wire polynomials at z, saw commits (7),
        # opening at z*omega.  aw = [lin, sigma_1..3 of the prover key, f, h_2, table]; saw = [z, w_l, w_r, w_4, h_1, z_2, table]
        lin_poly = c[11]                                                                  # c[11] stands in for lin_poly ...
        if self.linearisation:                                                            # ... unless it is computed (prover.rs:485-555)
            from . import linearisation as lin_mod
            lin_poly, self.last_evals = lin_mod.compute(d, self.key_polys, dict(self.lin_chal, z_challenge=self.z_mont), {
                "w_l": c[0], "w_r": c[1], "w_o": c[2], "w_4": c[3], "t_1": t[0:n], "t_2": t[n:2 * n], "t_3": t[2 * n:3 * n], "t_4": t[3 * n:4 * n],
                "z": c[8], "z2": c[9], "f": c[5], "h1": c[6], "h2": c[7], "table": c[4]})
        aw = [lin_poly, self.sigma[0], self.sigma[1], self.sigma[2], c[5], c[7], c[4]]
        aw_labels = ["lin", "sigma1", "sigma2", "sigma3", "f", "h2", "table"]
        aw_open = aw + [c[0], c[1], c[2], c[3]]   # prover.rs:582-591
        saw = [c[8], c[0], c[1], c[3], c[6], c[9], c[4]]
        saw_labels = ["z", "w_l", "w_r", "w_4", "h1", "z2", "table"]
        from .msm import kzg_witness
        # the witness polynomials are computed by every rank (replicated, like the NTTs); their MSMs shard
        if self.fuse_round5:
            w1 = kzg_witness(aw_open, self.z_mont, self.chi_mont, self.cv.curve_id, self.ctx)
            w2 = kzg_witness(saw, self.z_mont, self.chi_mont, self.cv.curve_id, self.ctx)   # at z*omega (prover.rs:609-618)
            out += self._commit_round(aw + [w1] + saw + [w2], canonical=[False] * 7 + [True] + [False] * 7 + [True],
                                      labels=aw_labels + ["W_z"] + saw_labels + ["W_zw"])
        else:
            self._round_begin(aw, labels=aw_labels)                                          # PC::commit(aw_polys)
            self._open_begin(aw_open, "W_z")                                                 # PC::open
            self._round_begin(saw, labels=saw_labels)                                        # PC::commit(saw_polys)
            self._open_begin(saw, "W_zw")                                                    # PC::open at z*omega
            out += self._round_end()
        assert len(out) == 29
        return out

    # work counters for the report
    def ntt_bytes(self) -> int:
        return 17 * 2 * 32 * self.n + 14 * 2 * 32 * 4 * self.n

    def msm_count(self) -> int:
        return 29


class DropInSchedule:
    """The same per-proof schedule through the HOST-POINTER entry points a Rust shim binds: every
    transform is one `zk_ntt` on ordinary (pageable) numpy buffers -- the patched ark-poly sees one `fft_in_place` at a
    time -- every `PC::commit(ck, polys)` is one `zk_kzg_commit_batch` over the caller's slices and every `PC::open` one
    `zk_kzg_open`; the SRS is registered once (zk_srs_register is content-addressed, so the reference's `PC::trim` on every
    gen_proof, circuit.rs:276, is a lookup).  Same seeded inputs and same outputs as `ProofSchedule`; what differs is that
    every buffer crosses PCIe on every call, as it does for an unchanged `Prover::prove`."""

    def __init__(self, log_n: int, ctx, ck: CommitterKey, curve="bls12_381", seed: int = 0x5EED0000):
        import torch
        self.cv = get_curve(curve)
        self.log_n, self.n, self.ctx, self.ck = log_n, 1 << log_n, ctx, ck
        self.dom_n = Radix2EvaluationDomain.new(self.n, curve, ctx)
        self.dom_4n = Radix2EvaluationDomain.new(4 * self.n, curve, ctx)
        dev = torch.device("cuda", ctx.device)
        g = torch.Generator(device=dev).manual_seed(seed)
        n = self.n

        def rnd(rows):   # the same draws, in the same order, as ProofSchedule
            return torch.randint(0, 1 << 62, (rows, 4), dtype=torch.int64, device=dev, generator=g).cpu().numpy().view(np.uint64)

        self.evals = [rnd(n) for _ in range(4)]
        self.aux_evals = [rnd(n) for _ in range(9)]
        self.sigma = [rnd(n) for _ in range(4)]
        self.quot = rnd(4 * n)
        self.ev4n = np.zeros((4 * n, 4), dtype=np.uint64)
        self.t4n = np.zeros((4 * n, 4), dtype=np.uint64)
        self.coef = [np.zeros((n, 4), dtype=np.uint64) for _ in range(13)]
        self.scratch_n = np.zeros((n, 4), dtype=np.uint64)
        self.z_mont = np.array([0x1234567, 0x89abcdef, 0x13579bdf, 0x0fedcba9], dtype=np.uint64)
        self.chi_mont = np.array([0x2468ace, 0x7654321, 0x2222222, 0x0111111], dtype=np.uint64)
        self.msms_run = 0
        self._cur_id = 0
        self._next_id = 0

    def run_once(self, proof_id=None):
        if proof_id is None:
            proof_id = self._next_id
            self._next_id += 1
        delta = proof_id - self._cur_id
        if delta:                                                   # the same per-proof witnesses as ProofSchedule._set_proof
            for t in self.evals + self.aux_evals + [self.quot]:
                t[0, 0] = np.uint64((int(t[0, 0]) + delta) % (1 << 64))
        self._cur_id = proof_id
        d, d4, n = self.dom_n, self.dom_4n, self.n
        ck = self.ck
        out = []
        c = self.coef
        # every transform is the shim's `*_in_place` on a caller-owned vector: ark's `domain.ifft(&evals)` is
        # `let mut v = evals.to_vec(); self.ifft_in_place(&mut v); v` -- the copy is the caller's, the call is in place.
        # The result vectors are allocated once and reused, so the page faults of fresh allocations (the caller's
        # allocator, not this library) stay out of the number.

        def ifft(k, src):
            d._run(1, src, out=c[k])

        for i in range(4):
            ifft(i, self.evals[i])
        out += ck.commit_batch(c[:4])                               # prover.rs:213
        ifft(4, self.aux_evals[0])
        ifft(5, self.aux_evals[1])
        out += ck.commit_batch([c[5]])                              # :289
        ifft(6, self.aux_evals[2])
        ifft(7, self.aux_evals[3])
        out += ck.commit_batch([c[6]])                              # :312
        out += ck.commit_batch([c[7]])                              # :315
        for i in range(4):
            d._run(0, self.sigma[i], out=self.scratch_n)            # permutation/mod.rs:671-674
        ifft(8, self.aux_evals[4])
        out += ck.commit_batch([c[8]])                              # :361
        ifft(9, self.aux_evals[5])
        out += ck.commit_batch([c[9]])                              # :387
        ifft(10, self.aux_evals[6])
        ifft(11, self.aux_evals[7])
        for poly in (c[11], c[8], c[0], c[1], c[2], c[3], c[9], c[5], c[4], c[6], c[7], c[10]):
            d4._run(2, poly, out=self.ev4n)                         # quotient_poly.rs:72-120
        ifft(12, self.aux_evals[8])
        d4._run(2, c[12], out=self.ev4n)
        t = d4._run(3, self.quot, out=self.t4n)                     # quotient_poly.rs:175-177
        out += ck.commit_batch([t[i * n:(i + 1) * n] for i in range(4)])    # :459
        aw = [c[11], self.sigma[0], self.sigma[1], self.sigma[2], c[5], c[7], c[4]]
        saw = [c[8], c[0], c[1], c[3], c[6], c[9], c[4]]
        out += ck.commit_batch(aw)                                  # :579
        out.append(ck.open(aw + [c[0], c[1], c[2], c[3]], self.z_mont, self.chi_mont))   # :582
        out += ck.commit_batch(saw)                                 # :606
        out.append(ck.open(saw, self.z_mont, self.chi_mont))        # :609
        self.msms_run = 29
        assert len(out) == 29
        return out
