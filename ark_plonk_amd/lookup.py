"""Host mirror of round 2 of `Prover::prove` (proof_system/prover.rs:228-317) on device-resident vectors: the compressed
table, the compressed query column and `MultiSet::combine_split` (lookup/multiset.rs:131-176) -- what the reference computes
on the CPU between the wire iffts and the iffts / commitments of f, h_1 and h_2."""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import check, lib
from .context import check_dev_tensor, default_context, ptr_of
from .curves import fr_from_mont, fr_to_mont, get_curve
from .linearisation import lincomb


class ElementNotIndexed(ValueError):
    """`Error::ElementNotIndexed` (error.rs): a query value is not in the table."""


def compress_table(columns, zeta_mont, curve="bls12_381", ctx=None):
    """`MultiSet::compress(&[table_1..table_4], zeta)` (prover.rs:229-237): t_1 + zeta t_2 + zeta^2 t_3 + zeta^3 t_4 per row."""
    cv = get_curve(curve)
    z = fr_from_mont(cv, np.ascontiguousarray(zeta_mont, dtype=np.uint64).reshape(1, 4))[0]
    coeffs = fr_to_mont(cv, [pow(z, k, cv.r) for k in range(len(columns))])
    return lincomb(list(columns), coeffs, curve=cv, ctx=ctx)


def compress_query(q_lookup, wires, zeta_mont, table_compressed, n=None, curve="bls12_381", ctx=None):
    """prover.rs:244-279: the compressed query column f over n rows (default: the wires' length)."""
    import torch
    cv = get_curve(curve)
    ctx = ctx or default_context(wires[0].device.index)
    rows = [check_dev_tensor(w, 4, ctx.device) for w in wires]
    n = rows[0] if n is None else n
    if len(wires) != 4 or any(r < n for r in rows):
        raise ValueError("four wire columns of at least n rows expected")
    q_len = check_dev_tensor(q_lookup, 4, ctx.device)
    if check_dev_tensor(table_compressed, 4, ctx.device) < 1:
        raise ValueError("empty table")
    out = torch.empty((n, 4), dtype=torch.int64, device=wires[0].device)
    w = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in wires])
    z = np.ascontiguousarray(zeta_mont, dtype=np.uint64).reshape(4)
    ctx.use_torch_stream()
    check(lib().zk_lookup_query_dev(ctx.handle, cv.curve_id, n, q_lookup.data_ptr(), q_len, w, ptr_of(z), table_compressed.data_ptr(), out.data_ptr()),
          "zk_lookup_query_dev")
    return out


def combine_split(t, f, curve="bls12_381", ctx=None):
    """`t.combine_split(&f)` -> (h_1, h_2) device tensors; raises ElementNotIndexed like the reference returns the error."""
    import torch
    from ._lib import ZK_ERR_NOT_INDEXED
    cv = get_curve(curve)
    ctx = ctx or default_context(t.device.index)
    n_t, n_f = check_dev_tensor(t, 4, ctx.device), check_dev_tensor(f, 4, ctx.device)
    cap = (n_t + n_f + 1) // 2
    h1 = torch.empty((cap, 4), dtype=torch.int64, device=t.device)
    h2 = torch.empty((cap, 4), dtype=torch.int64, device=t.device)
    l1, l2 = ctypes.c_size_t(), ctypes.c_size_t()
    ctx.use_torch_stream()
    rc = lib().zk_lookup_combine_split_dev(ctx.handle, cv.curve_id, t.data_ptr(), n_t, f.data_ptr(), n_f, h1.data_ptr(), h2.data_ptr(),
                                           ctypes.byref(l1), ctypes.byref(l2))
    if rc == ZK_ERR_NOT_INDEXED:
        raise ElementNotIndexed("a query value is not in the table")
    check(rc, "zk_lookup_combine_split_dev")
    return h1[:l1.value], h2[:l2.value]
