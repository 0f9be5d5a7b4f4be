"""Host mirror of the grand-product builders of `plonk-core/src/permutation/mod.rs`.

  Permutation::compute_permutation_poly(domain, wires, beta, gamma, sigma_polys)      mod.rs:652-752
  Permutation::compute_lookup_permutation_poly(domain, f, t, h_1, h_2, delta, epsilon) mod.rs:754-822

Same argument meaning; inputs are device-resident (n, 4) int64 tensors of Montgomery Fr limbs and the
challenges are 4-limb Montgomery values.  Both return the COEFFICIENTS of the polynomial like the
reference (product scan + `domain.ifft`, all on the device); `*_evals` stop before the iFFT.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import check, lib
from .context import check_dev_tensor, ptr_of
from .domain import Radix2EvaluationDomain


def _limbs4(x):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(4)


def permutation_evals(domain: Radix2EvaluationDomain, wires, sigma_evals, beta_mont, gamma_mont, return_last: bool = False):
    """z over the domain: z[0] = 1, z[i+1] = z[i] * num_i / den_i (mod.rs:686-747).  sigma_evals = domain.fft(sigma_k)."""
    import torch
    if len(wires) != 4 or len(sigma_evals) != 4:
        raise ValueError("four wire columns and four sigma columns expected")
    ctx = domain._ctx_for(wires[0])
    n = domain.size()
    w = (ctypes.c_void_p * 4)()
    s = (ctypes.c_void_p * 4)()
    for k in range(4):
        if check_dev_tensor(wires[k], 4, ctx.device) != n or check_dev_tensor(sigma_evals[k], 4, ctx.device) != n:
            raise ValueError("every column must hold domain.size() elements")   # mod.rs:749 assert_eq!(n, z.len())
        w[k], s[k] = wires[k].data_ptr(), sigma_evals[k].data_ptr()
    out = torch.empty((n, 4), dtype=torch.int64, device=wires[0].device)
    beta, gamma = _limbs4(beta_mont), _limbs4(gamma_mont)
    last = np.zeros(4, dtype=np.uint64)
    ctx.use_torch_stream()
    check(lib().zk_perm_product_dev(ctx.handle, domain.curve.curve_id, domain.log_size_of_group(), w, s, ptr_of(beta), ptr_of(gamma),
                                    ptr_of(out), ptr_of(last)), "zk_perm_product_dev")
    return (out, last) if return_last else out


def compute_permutation_poly(domain: Radix2EvaluationDomain, wires, beta_mont, gamma_mont, sigma_polys):
    """Coefficients of z(X).  sigma_polys: coefficient vectors (<= n each), evaluated here as the reference does (mod.rs:671-676)."""
    sig = [domain.fft(p) for p in sigma_polys]
    return domain.ifft(permutation_evals(domain, wires, sig, beta_mont, gamma_mont))


def lookup_permutation_evals(ctx, curve, f, t, h_1, h_2, delta_mont, epsilon_mont, return_last: bool = False):
    import torch
    from .curves import get_curve
    cv = get_curve(curve)
    n = check_dev_tensor(f, 4, ctx.device)
    for x in (t, h_1, h_2):
        if check_dev_tensor(x, 4, ctx.device) != n:
            raise ValueError("f, t, h_1, h_2 must have the same length")        # mod.rs:766-769 assert_eq!
    out = torch.empty((n, 4), dtype=torch.int64, device=f.device)
    delta, eps = _limbs4(delta_mont), _limbs4(epsilon_mont)
    last = np.zeros(4, dtype=np.uint64)
    ctx.use_torch_stream()
    check(lib().zk_lookup_product_dev(ctx.handle, cv.curve_id, n, ptr_of(f), ptr_of(t), ptr_of(h_1), ptr_of(h_2), ptr_of(delta), ptr_of(eps),
                                      ptr_of(out), ptr_of(last)), "zk_lookup_product_dev")
    return (out, last) if return_last else out


def compute_lookup_permutation_poly(domain: Radix2EvaluationDomain, f, t, h_1, h_2, delta_mont, epsilon_mont):
    """Coefficients of z2(X) (mod.rs:754-800)."""
    ctx = domain._ctx_for(f)
    if f.shape[0] != domain.size():
        raise ValueError("f must hold domain.size() elements")                  # mod.rs:766
    return domain.ifft(lookup_permutation_evals(ctx, domain.curve, f, t, h_1, h_2, delta_mont, epsilon_mont))
