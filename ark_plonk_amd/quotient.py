"""Host mirror of `plonk-core/src/proof_system/quotient_poly.rs::compute`.

`compute_quotient_evals` takes what the reference has after its 13 coset FFTs -- the 4n coset evaluations
of the witness-side polynomials, the prover key's selector / sigma evaluations and the round challenges --
and returns the 4n quotient evaluations (the input of the final `coset_ifft`), computed by one kernel.
`compute` goes all the way like the reference: coset FFTs of the coefficient vectors, the kernel, coset iFFT.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import check, lib
from .context import check_dev_tensor
from .domain import Radix2EvaluationDomain

COLUMNS = ("w_l", "w_r", "w_o", "w_4", "z", "z2", "f", "table", "h1", "h2", "pi", "l1",
           "q_m", "q_l", "q_r", "q_o", "q_4", "q_c", "q_arith", "q_range", "q_logic", "q_fixed_group_add",
           "q_variable_group_add", "q_lookup")
CHALLENGES = ("alpha", "beta", "gamma", "delta", "epsilon", "zeta", "range_challenge", "logic_challenge",
              "fixed_base_challenge", "var_base_challenge", "lookup_challenge", "coeff_a", "coeff_d")


class QuotientArgs(ctypes.Structure):
    """`zk_quotient_args` of include/ark_plonk_amd.h."""
    _fields_ = ([(name, ctypes.c_void_p) for name in COLUMNS] + [("sigma", ctypes.c_void_p * 4)]
                + [(name, ctypes.c_uint64 * 4) for name in CHALLENGES])


def compute_quotient_evals(domain: Radix2EvaluationDomain, columns: dict, sigmas, challenges: dict):
    """columns: name -> (4n, 4) int64 device tensor for every name in COLUMNS; sigmas: 4 such tensors
    (left, right, out, fourth); challenges: name -> 4 Montgomery limbs for every name in CHALLENGES.
    `domain` is the size-n circuit domain (quotient_poly.rs:34-36)."""
    import torch
    n4 = 4 * domain.size()
    ctx = domain._ctx_for(columns["w_l"])
    args = QuotientArgs()
    for name in COLUMNS:
        t = columns[name]
        if check_dev_tensor(t, 4, ctx.device) != n4:
            raise ValueError(f"{name}: expected {n4} evaluations over the 4n coset")
        setattr(args, name, t.data_ptr())
    if len(sigmas) != 4:
        raise ValueError("four sigma evaluation vectors expected")
    for k in range(4):
        if check_dev_tensor(sigmas[k], 4, ctx.device) != n4:
            raise ValueError(f"sigma[{k}]: expected {n4} evaluations over the 4n coset")
        args.sigma[k] = sigmas[k].data_ptr()
    for name in CHALLENGES:
        v = np.ascontiguousarray(challenges[name], dtype=np.uint64).reshape(4)
        getattr(args, name)[:] = [int(x) for x in v]
    out = torch.empty((n4, 4), dtype=torch.int64, device=columns["w_l"].device)
    ctx.use_torch_stream()
    check(lib().zk_quotient_evals_dev(ctx.handle, domain.curve.curve_id, domain.log_size_of_group(), ctypes.byref(args), out.data_ptr()),
          "zk_quotient_evals_dev")
    return out


def l1_coset_evals(domain: Radix2EvaluationDomain, domain_4n: Radix2EvaluationDomain, device):
    """coset_fft of the first Lagrange polynomial over the 4n domain (quotient_poly.rs:68-69,313-326): depends on n only, so a
    prover key may hold it."""
    import torch
    from .curves import fr_to_mont
    l1_evals = torch.zeros((domain.size(), 4), dtype=torch.int64, device=device)
    l1_evals[0] = torch.from_numpy(fr_to_mont(domain.curve, [1])[0].view(np.int64)).to(device)
    return domain_4n.coset_fft(domain.ifft(l1_evals))


def compute(domain: Radix2EvaluationDomain, domain_4n: Radix2EvaluationDomain, polys: dict, key_evals: dict, sigmas, challenges: dict,
            l1_4n=None):
    """quotient_poly.rs:34-178: polys = coefficient vectors of w_l, w_r, w_o, w_4, z, z2, f, table, h1, h2, pi
    (l1 is built here as the reference does, quotient_poly.rs:68-69,313-326, unless the caller holds it: l1_4n); key_evals = the
    prover key's 4n coset evaluations.  The eleven coset FFTs go out as one batch (one launch per pass).
    Returns the 4n coefficients of the quotient polynomial."""
    names = ("w_l", "w_r", "w_o", "w_4", "z", "z2", "f", "table", "h1", "h2", "pi")
    cols = dict(zip(names, domain_4n.batch(2, [polys[name] for name in names])))
    cols["l1"] = l1_4n if l1_4n is not None else l1_coset_evals(domain, domain_4n, polys["w_l"].device)
    cols.update(key_evals)
    return domain_4n.coset_ifft(compute_quotient_evals(domain, cols, sigmas, challenges))
