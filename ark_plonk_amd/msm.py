"""Host mirror of `ark_ec::msm::VariableBaseMSM` and of the KZG10 commit/open glue.

Mirrors:
  VariableBaseMSM::multi_scalar_mul(bases: &[G1Affine], scalars: &[BigInteger256]) -> G1Projective
      (plonk-core/src/commitment.rs:45,83)  -- here returned already `.into()` affine
  PolynomialCommitment::commit / open of SonicKZG10 without hiding or degree bounds
      (proof_system/prover.rs:213,289-291,...,582-591) via `CommitterKey` (device-resident powers_of_g)
Bases: (n, 2L) uint64 limbs x||y Montgomery (L = 6 BLS12-381, 4 BN254) + optional infinity flags.
Scalars: (n, 4) uint64 canonical limbs (`into_repr`).  Truncates to the shorter slice like the reference.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import ZK_ERR_UNSUPPORTED, check, lib
from .context import Context, _is_torch, as_host_u64, check_dev_tensor, default_context, ptr_of
from .curves import get_curve


def _has_torch_cuda() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except ImportError:
        return False


class G1Affine:
    """(x, y, infinity) with x, y Montgomery limbs -- the fields of ark's GroupAffine."""

    __slots__ = ("x", "y", "infinity", "curve")

    def __init__(self, x, y, infinity, curve):
        self.x, self.y, self.infinity, self.curve = x, y, bool(infinity), curve

    def __eq__(self, o):
        return (isinstance(o, G1Affine) and self.infinity == o.infinity and np.array_equal(self.x, o.x)
                and np.array_equal(self.y, o.y))

    def __repr__(self):
        return f"G1Affine(inf={self.infinity}, x={[hex(int(v)) for v in self.x]}, y={[hex(int(v)) for v in self.y]})"

    def xy(self) -> np.ndarray:
        return np.concatenate([self.x, self.y])


def _point(out_xy, out_inf, cv) -> G1Affine:
    L = cv.fq_limbs
    return G1Affine(out_xy[:L].copy(), out_xy[L:].copy(), int(out_inf[0]) != 0, cv.name)


class CommitterKey:
    """Device-resident `powers_of_g` (what `PC::trim` hands the prover; circuit.rs:236,276)."""

    def __init__(self, powers_of_g, curve="bls12_381", ctx: Context | None = None, infinity=None):
        """Host arrays go through zk_srs_register, which is content-addressed: registering the same bases again (the
        reference trims on every gen_proof, circuit.rs:276) returns the resident handle, window table included.
        Device tensors (zk_srs_register_dev) are never cached.  The handle belongs to the device: pass `ctx=` of any
        Context on the same GPU to the call methods to use it from another proof stream."""
        self.curve = get_curve(curve)
        L = self.curve.fq_limbs
        self._h = ctypes.c_void_p()
        if _is_torch(powers_of_g):
            self.ctx = ctx or default_context(powers_of_g.device.index)
            n = check_dev_tensor(powers_of_g, 2 * L, self.ctx.device)
            d_inf = None
            if infinity is not None:
                import torch
                if not _is_torch(infinity) or infinity.dtype != torch.uint8 or infinity.numel() != n or not infinity.is_cuda:
                    raise ValueError("device bases take device uint8 infinity flags of the same length")
                d_inf = ptr_of(infinity.contiguous())
            self.ctx.use_torch_stream()
            check(lib().zk_srs_register_dev(self.ctx.handle, self.curve.curve_id, ptr_of(powers_of_g), d_inf, n, ctypes.byref(self._h)),
                  "zk_srs_register_dev")
        else:
            self.ctx = ctx or default_context(0)
            a = as_host_u64(powers_of_g, 2 * L)
            n = a.shape[0]
            inf = None
            if infinity is not None:
                inf = np.ascontiguousarray(infinity, dtype=np.uint8)
                if inf.shape[0] != n:
                    raise ValueError("infinity flags length mismatch")
            check(lib().zk_srs_register(self.ctx.handle, self.curve.curve_id, ptr_of(a), None if inf is None else ptr_of(inf), n,
                                        ctypes.byref(self._h)), "zk_srs_register")
        self.n = n

    def __len__(self):
        return self.n

    def precompute(self, window_bits: int = 0, rows=None):
        """Build the window-multiples table; later MSMs share one bucket set.  window_bits: 0 = default (c = 16, 16 rows, below 2^19
        points; c = 17 from there on: 15 rows for 255-bit scalars, which are folded to k <= (r - 1) / 2; c = 20, 13 rows, from 2^22 points on), else 16 .. 21 (fewer rows =
        fewer additions per scalar, more buckets to reduce).
        rows = (g, G): the multi-GPU form sharded by WINDOWS -- this rank builds only the rows of the windows g, g + G, ... of the
        whole SRS, and every MSM / commit over the key returns the rank's partial (zk_srs_precompute_rows)."""
        self.ctx.use_torch_stream() if _has_torch_cuda() else None
        if rows is None:
            check(lib().zk_srs_precompute_ex(self.ctx.handle, self._h, int(window_bits)), "zk_srs_precompute_ex")
        else:
            g, G = rows
            check(lib().zk_srs_precompute_rows(self.ctx.handle, self._h, int(window_bits), int(g), int(G)), "zk_srs_precompute_rows")
        return self

    def table_rows(self):
        """(first_window, window_stride, rows) of the table this key holds: (0, 1, windows) for a whole table."""
        a, b, r = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        check(lib().zk_srs_table_rows(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(r)), "zk_srs_table_rows")
        return a.value, b.value, r.value

    def table_windows(self) -> int:
        """rows of the window table = mixed additions per scalar on the shared-bucket path (0: no table)."""
        c, w = ctypes.c_uint32(), ctypes.c_uint32()
        check(lib().zk_srs_table_info(self._h, ctypes.byref(c), ctypes.byref(w)), "zk_srs_table_info")
        return w.value

    def table_window_bits(self) -> int:
        c, w = ctypes.c_uint32(), ctypes.c_uint32()
        check(lib().zk_srs_table_info(self._h, ctypes.byref(c), ctypes.byref(w)), "zk_srs_table_info")
        return c.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False):
                lib().zk_srs_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- MSM over powers[base_offset : base_offset + len(scalars)]
    def msm(self, scalars, base_offset: int = 0) -> G1Affine:
        L = self.curve.fq_limbs
        out = np.zeros(2 * L, dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        if _is_torch(scalars):
            n = check_dev_tensor(scalars, 4, self.ctx.device)
            self.ctx.use_torch_stream()
            check(lib().zk_msm_g1_srs_dev(self.ctx.handle, self._h, base_offset, ptr_of(scalars), n, ptr_of(out), ptr_of(inf)),
                  "zk_msm_g1_srs_dev")
        else:
            s = as_host_u64(scalars, 4)
            check(lib().zk_msm_g1_srs(self.ctx.handle, self._h, base_offset, ptr_of(s), s.shape[0], ptr_of(out), ptr_of(inf)),
                  "zk_msm_g1_srs")
        return _point(out, inf, self.curve)

    def msm_partial(self, scalars, base_offset: int = 0) -> np.ndarray:
        """This rank's Jacobian partial (3L limbs) for the multi-GPU all-gather."""
        L = self.curve.fq_limbs
        out = np.zeros(3 * L, dtype=np.uint64)
        n = check_dev_tensor(scalars, 4, self.ctx.device)
        self.ctx.use_torch_stream()
        check(lib().zk_msm_g1_srs_partial_dev(self.ctx.handle, self._h, base_offset, ptr_of(scalars), n, ptr_of(out)),
              "zk_msm_g1_srs_partial_dev")
        return out

    # -- KZG10::commit(powers, polynomial, hiding_bound = None)
    def commit(self, coeffs_mont) -> G1Affine:
        L = self.curve.fq_limbs
        out = np.zeros(2 * L, dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        if _is_torch(coeffs_mont):
            n = check_dev_tensor(coeffs_mont, 4, self.ctx.device)
            self.ctx.use_torch_stream()
            check(lib().zk_kzg_commit_dev(self.ctx.handle, self._h, ptr_of(coeffs_mont), n, ptr_of(out), ptr_of(inf)), "zk_kzg_commit_dev")
        else:
            a = as_host_u64(coeffs_mont, 4)
            check(lib().zk_kzg_commit(self.ctx.handle, self._h, ptr_of(a), a.shape[0], ptr_of(out), ptr_of(inf)), "zk_kzg_commit")
        return _point(out, inf, self.curve)

    def with_ctx(self, ctx: Context) -> "CommitterKey":
        """The same device-resident SRS (and window table) driven from another Context of the same GPU."""
        if ctx.device != self.ctx.device:
            raise ValueError("an SRS handle belongs to one device")
        check(lib().zk_srs_retain(self._h), "zk_srs_retain")     # the copy owns a reference: it may outlive `self`
        other = object.__new__(CommitterKey)
        other.curve, other.ctx, other.n, other._h, other._borrowed = self.curve, ctx, self.n, self._h, False
        return other

    def commit_batch(self, polys, canonical=None) -> list:
        """The MSMs of one prover round (<= 16 vectors) as one batch.  Device tensors: zk_kzg_round_batch_dev;
        canonical[k] marks inputs that already are canonical scalars (opening witnesses) rather than Montgomery
        coefficients.  Host arrays: zk_kzg_commit_batch (what GpuKZG10::commit forwards `polys` to)."""
        L = self.curve.fq_limbs
        k = len(polys)
        if k and not _is_torch(polys[0]):
            if canonical is not None and any(canonical):
                raise ValueError("host batches hold Montgomery coefficients only")
            arrs = [as_host_u64(p, 4) for p in polys]
            ptrs = (ctypes.c_void_p * k)()
            lens = (ctypes.c_size_t * k)()
            for i, a in enumerate(arrs):
                ptrs[i], lens[i] = a.ctypes.data, a.shape[0]
            out = np.zeros((k, 2 * L), dtype=np.uint64)
            inf = np.zeros(k, dtype=np.uint8)
            check(lib().zk_kzg_commit_batch(self.ctx.handle, self._h, k, ptrs, lens, ptr_of(out), ptr_of(inf)), "zk_kzg_commit_batch")
            return [_point(out[i], inf[i:i + 1], self.curve) for i in range(k)]
        kinds = None
        if canonical is not None:
            kinds = np.ascontiguousarray([1 if f else 0 for f in canonical], dtype=np.uint8)
        ptrs = (ctypes.c_void_p * k)()
        lens = (ctypes.c_size_t * k)()
        for i, p in enumerate(polys):
            lens[i] = check_dev_tensor(p, 4, self.ctx.device)
            ptrs[i] = p.data_ptr()
        out = np.zeros((k, 2 * L), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.uint8)
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_round_batch_dev(self.ctx.handle, self._h, k, ptrs, lens, None if kinds is None else ptr_of(kinds), ptr_of(out),
                                           ptr_of(inf)), "zk_kzg_round_batch_dev")
        return [_point(out[i], inf[i:i + 1], self.curve) for i in range(k)]

    def commit_batch_partial(self, polys, canonical=None) -> np.ndarray:
        """Sharded form: Jacobian partials (k, 3L) of the given coefficient slices over this rank's SRS shard."""
        L = self.curve.fq_limbs
        k = len(polys)
        kinds = None
        if canonical is not None:
            kinds = np.ascontiguousarray([1 if f else 0 for f in canonical], dtype=np.uint8)
        ptrs = (ctypes.c_void_p * k)()
        lens = (ctypes.c_size_t * k)()
        for i, p in enumerate(polys):
            lens[i] = check_dev_tensor(p, 4, self.ctx.device)
            ptrs[i] = p.data_ptr()
        out = np.zeros((k, 3 * L), dtype=np.uint64)
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_round_batch_partial_dev(self.ctx.handle, self._h, k, ptrs, lens, None if kinds is None else ptr_of(kinds),
                                                   ptr_of(out)), "zk_kzg_round_batch_partial_dev")
        return out

    # -- deferred rounds (zk_kzg_round_begin_dev ... zk_kzg_round_end): commitments whose inputs do not depend on each
    #    other's results -- f | h_1 | h_2, z | z_2, the four calls of the last round -- queued by several calls, collected by one
    def commit_begin(self, polys, canonical=None) -> int:
        """Queue the MSMs of `polys` (device tensors) in the ctx's open round; returns the number of jobs now pending."""
        k = len(polys)
        kinds = None
        if canonical is not None:
            kinds = np.ascontiguousarray([1 if f else 0 for f in canonical], dtype=np.uint8)
        ptrs = (ctypes.c_void_p * k)()
        lens = (ctypes.c_size_t * k)()
        for i, p in enumerate(polys):
            lens[i] = check_dev_tensor(p, 4, self.ctx.device)
            ptrs[i] = p.data_ptr()
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_round_begin_dev(self.ctx.handle, self._h, k, ptrs, lens, None if kinds is None else ptr_of(kinds)),
              "zk_kzg_round_begin_dev")
        return self.round_pending()

    def open_begin(self, polys, point_mont, challenge_mont) -> int:
        """PC::open as a job of the open round: the witness polynomial is built now, its MSM is deferred."""
        k = len(polys)
        ptrs = (ctypes.c_void_p * k)()
        lens = (ctypes.c_size_t * k)()
        for i, p in enumerate(polys):
            lens[i] = check_dev_tensor(p, 4, self.ctx.device)
            ptrs[i] = p.data_ptr()
        z = np.ascontiguousarray(point_mont, dtype=np.uint64).reshape(4)
        ch = np.ascontiguousarray(challenge_mont, dtype=np.uint64).reshape(4)
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_open_begin_dev(self.ctx.handle, self._h, k, ptrs, lens, ptr_of(z), ptr_of(ch)), "zk_kzg_open_begin_dev")
        return self.round_pending()

    def round_pending(self) -> int:
        n = ctypes.c_uint32()
        check(lib().zk_kzg_round_pending(self.ctx.handle, ctypes.byref(n)), "zk_kzg_round_pending")
        return n.value

    def round_reduce(self):
        """Queue the round's reduction kernels now (optional): the round takes no further jobs, and `round_end` then waits for these
        kernels only, so work queued on the stream in between -- transforms that do not depend on this round's results -- runs while
        the host finishes the round."""
        check(lib().zk_kzg_round_reduce(self.ctx.handle), "zk_kzg_round_reduce")

    def round_end(self, n_jobs: int | None = None) -> list:
        """Close the round: one G1Affine per job, in submission order."""
        L = self.curve.fq_limbs
        k = self.round_pending() if n_jobs is None else n_jobs
        out = np.zeros((max(k, 1), 2 * L), dtype=np.uint64)
        inf = np.zeros(max(k, 1), dtype=np.uint8)
        check(lib().zk_kzg_round_end(self.ctx.handle, k, ptr_of(out), ptr_of(inf)), "zk_kzg_round_end")
        return [_point(out[i], inf[i:i + 1], self.curve) for i in range(k)]

    def round_end_partial(self, n_jobs: int | None = None) -> np.ndarray:
        """Close the round on a sharded SRS: this rank's Jacobian partials (jobs, 3L)."""
        L = self.curve.fq_limbs
        k = self.round_pending() if n_jobs is None else n_jobs
        out = np.zeros((max(k, 1), 3 * L), dtype=np.uint64)
        check(lib().zk_kzg_round_end_partial(self.ctx.handle, k, ptr_of(out)), "zk_kzg_round_end_partial")
        return out[:k]

    def winsums_dev_words(self) -> int:
        """int64 words of one job's window sums (zk_winsums_dev_bytes / 8); 0 where the form does not exist (no table, c >= 18)."""
        return lib().zk_winsums_dev_bytes(self.ctx.handle, self._h) // 8

    def winsums_geometry(self):
        """(window_bits, windows of a full-width scalar, virtual windows, buckets per virtual window), or None: what every rank of a
        sharded MSM must agree on for its window sums to be addable element-wise."""
        g = (ctypes.c_uint32 * 4)()
        rc = lib().zk_winsums_geometry(self.ctx.handle, self._h, g)
        if rc == ZK_ERR_UNSUPPORTED:
            return None
        check(rc, "zk_winsums_geometry")
        return tuple(g)

    def round_reduce_winsums_dev(self, d_out):
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_round_reduce_winsums_dev(self.ctx.handle, d_out.data_ptr()), "zk_kzg_round_reduce_winsums_dev")

    def round_end_winsums_dev(self, d_out, n_jobs: int | None = None):
        """Close the round leaving every job's window sums ON THE DEVICE in `d_out` (jobs x winsums_dev_words int64, no host wait)."""
        k = self.round_pending() if n_jobs is None else n_jobs
        if d_out.numel() < k * self.winsums_dev_words():
            raise ValueError("window-sum buffer too small")
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_round_end_winsums_dev(self.ctx.handle, k, d_out.data_ptr()), "zk_kzg_round_end_winsums_dev")

    def sum_winsums_dev(self, d_all, ranks: int, n_jobs: int) -> list:
        """All-gathered window sums (ranks x jobs x winsums_dev_words, rank-major) -> one G1Affine per job: one element-wise kernel,
        one wait, the combine + inversion per job on the host pool (as zk_kzg_round_end)."""
        L = self.curve.fq_limbs
        out = np.zeros((max(n_jobs, 1), 2 * L), dtype=np.uint64)
        inf = np.zeros(max(n_jobs, 1), dtype=np.uint8)
        self.ctx.use_torch_stream()
        check(lib().zk_g1_sum_winsums_dev(self.ctx.handle, self._h, d_all.data_ptr(), ranks, n_jobs, ptr_of(out), ptr_of(inf)),
              "zk_g1_sum_winsums_dev")
        return [_point(out[i], inf[i:i + 1], self.curve) for i in range(n_jobs)]

    def round_abort(self):
        check(lib().zk_kzg_round_abort(self.ctx.handle), "zk_kzg_round_abort")

    # -- PC::open(ck, polys, comms, point, opening_challenge, rands, None)
    def open(self, polys, point_mont, challenge_mont) -> G1Affine:
        L = self.curve.fq_limbs
        out = np.zeros(2 * L, dtype=np.uint64)
        inf = np.zeros(1, dtype=np.uint8)
        k = len(polys)
        ptrs = (ctypes.c_void_p * k)()
        lens = (ctypes.c_size_t * k)()
        z = np.ascontiguousarray(point_mont, dtype=np.uint64).reshape(4)
        ch = np.ascontiguousarray(challenge_mont, dtype=np.uint64).reshape(4)
        if k and not _is_torch(polys[0]):
            arrs = [as_host_u64(p, 4) for p in polys]
            for i, a in enumerate(arrs):
                ptrs[i], lens[i] = a.ctypes.data, a.shape[0]
            check(lib().zk_kzg_open(self.ctx.handle, self._h, k, ptrs, lens, ptr_of(z), ptr_of(ch), ptr_of(out), ptr_of(inf)), "zk_kzg_open")
            return _point(out, inf, self.curve)
        for i, p in enumerate(polys):
            lens[i] = check_dev_tensor(p, 4, self.ctx.device)
            ptrs[i] = p.data_ptr()
        self.ctx.use_torch_stream()
        check(lib().zk_kzg_open_dev(self.ctx.handle, self._h, k, ptrs, lens, ptr_of(z), ptr_of(ch), ptr_of(out), ptr_of(inf)),
              "zk_kzg_open_dev")
        return _point(out, inf, self.curve)


def kzg_witness(polys, point_mont, challenge_mont, curve="bls12_381", ctx: Context | None = None):
    """(sum_k chi^k p_k - value) / (X - z) as canonical scalars on the device (the CPU part of PC::open)."""
    import torch
    cv = get_curve(curve)
    ctx = ctx or default_context(polys[0].device.index)
    k = len(polys)
    ptrs = (ctypes.c_void_p * k)()
    lens = (ctypes.c_size_t * k)()
    m = 0
    for i, p in enumerate(polys):
        lens[i] = check_dev_tensor(p, 4, ctx.device)
        ptrs[i] = p.data_ptr()
        m = max(m, lens[i])
    out = torch.empty((max(m, 1), 4), dtype=torch.int64, device=polys[0].device)
    z = np.ascontiguousarray(point_mont, dtype=np.uint64).reshape(4)
    ch = np.ascontiguousarray(challenge_mont, dtype=np.uint64).reshape(4)
    n_out = ctypes.c_size_t()
    ctx.use_torch_stream()
    check(lib().zk_kzg_witness_dev(ctx.handle, cv.curve_id, k, ptrs, lens, ptr_of(z), ptr_of(ch), ptr_of(out), ctypes.byref(n_out)),
          "zk_kzg_witness_dev")
    return out[: n_out.value]


def srs_cache_stats() -> dict:
    """Counters of the content-addressed SRS cache behind zk_srs_register."""
    v = [ctypes.c_uint64() for _ in range(4)]
    check(lib().zk_srs_cache_stats(*[ctypes.byref(x) for x in v]), "zk_srs_cache_stats")
    return {"hits": v[0].value, "misses": v[1].value, "entries": v[2].value, "resident_bytes": v[3].value}


def srs_cache_config(max_idle_bytes: int):
    check(lib().zk_srs_cache_config(int(max_idle_bytes)), "zk_srs_cache_config")


class VariableBaseMSM:
    """`ark_ec::msm::VariableBaseMSM`."""

    @staticmethod
    def multi_scalar_mul(bases, scalars, curve="bls12_381", infinity=None, ctx: Context | None = None) -> G1Affine:
        cv = get_curve(curve)
        L = cv.fq_limbs
        if _is_torch(bases) or _is_torch(scalars):
            if not (_is_torch(bases) and _is_torch(scalars)):
                raise ValueError("bases and scalars must both be device tensors or both host arrays")
            ctx = ctx or default_context(bases.device.index)
            nb = check_dev_tensor(bases, 2 * L, ctx.device)
            ns = check_dev_tensor(scalars, 4, ctx.device)
            n = min(nb, ns)
            ck = CommitterKey(bases.view(-1)[: n * 2 * L].view(n, 2 * L), cv, ctx)
            try:
                return ck.msm(scalars.view(-1)[: n * 4].view(n, 4))
            finally:
                ck.close()
        ctx = ctx or default_context(0)
        b = as_host_u64(bases, 2 * L)
        s = as_host_u64(scalars, 4)
        n = min(b.shape[0], s.shape[0])
        b, s = b[:n], s[:n]
        inf = None
        if infinity is not None:
            inf = np.ascontiguousarray(infinity, dtype=np.uint8)[:n]
        out = np.zeros(2 * L, dtype=np.uint64)
        oinf = np.zeros(1, dtype=np.uint8)
        check(lib().zk_msm_g1(ctx.handle, cv.curve_id, ptr_of(b), None if inf is None else ptr_of(inf), ptr_of(s), n,
                              ptr_of(out), ptr_of(oinf)), "zk_msm_g1")
        return _point(out, oinf, cv)


def sum_partials(partials, curve="bls12_381") -> G1Affine:
    """Combine the all-gathered Jacobian partials of a sharded MSM (host, tiny)."""
    cv = get_curve(curve)
    L = cv.fq_limbs
    p = as_host_u64(partials, 3 * L)
    out = np.zeros(2 * L, dtype=np.uint64)
    inf = np.zeros(1, dtype=np.uint8)
    check(lib().zk_g1_sum_partials(cv.curve_id, ptr_of(p), p.shape[0], ptr_of(out), ptr_of(inf)), "zk_g1_sum_partials")
    return _point(out, inf, cv)


def sum_partials_batch(partials, curve="bls12_381") -> list:
    """All-gathered partials of one prover round, shape (ranks, jobs, 3L) -> one G1Affine per job."""
    cv = get_curve(curve)
    L = cv.fq_limbs
    p = np.ascontiguousarray(partials, dtype=np.uint64)
    if p.ndim != 3 or p.shape[2] != 3 * L:
        raise ValueError(f"expected (ranks, jobs, {3 * L}) limbs, got {p.shape}")
    ranks, jobs = p.shape[0], p.shape[1]
    out = np.zeros((jobs, 2 * L), dtype=np.uint64)
    inf = np.zeros(max(jobs, 1), dtype=np.uint8)
    check(lib().zk_g1_sum_partials_batch(cv.curve_id, ptr_of(p), ranks, jobs, ptr_of(out), ptr_of(inf)), "zk_g1_sum_partials_batch")
    return [_point(out[k], inf[k:k + 1], cv) for k in range(jobs)]
