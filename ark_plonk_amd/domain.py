"""Host mirror of ark_poly's radix-2 `EvaluationDomain` as plonk-core uses it.

Mirrors (same names, argument meaning and error behaviour):
  GeneralEvaluationDomain::<F>::new(num_coeffs) -> Option<Self>   (prover.rs:169-173, quotient_poly.rs:64-69)
  .size(), EvaluationDomainExt::{log_size_of_group,size_inv,group_gen,group_gen_inv,generator_inv} (util.rs:24-88)
  .fft / .ifft / .coset_fft / .coset_ifft and the *_in_place forms
      (prover.rs:196-203, permutation/mod.rs:671-674,751, quotient_poly.rs:72-120,175-177)
All field elements are 4 x uint64 little-endian limbs in Montgomery form -- the in-memory value of an
arkworks `Fr`.  numpy arrays use the host-buffer C entry point; torch CUDA tensors stay on the device
and run on the current torch stream.  Everything executes in the HIP library: there is no CPU path.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import DomainInfo, check, lib
from .context import Context, _is_torch, as_host_u64, check_dev_tensor, default_context, ptr_of
from .curves import get_curve

KIND_FFT, KIND_IFFT, KIND_COSET_FFT, KIND_COSET_IFFT = 0, 1, 2, 3


class Radix2EvaluationDomain:
    """`GeneralEvaluationDomain::Radix2` for the scalar field of `curve`."""

    def __init__(self, info: DomainInfo, curve, ctx: Context | None):
        self.curve = get_curve(curve)
        self._ctx = ctx
        self._size = int(info.size)
        self._log = int(info.log_size_of_group)
        self._size_inv = np.array(list(info.size_inv), dtype=np.uint64)
        self._group_gen = np.array(list(info.group_gen), dtype=np.uint64)
        self._group_gen_inv = np.array(list(info.group_gen_inv), dtype=np.uint64)
        self._generator = np.array(list(info.generator), dtype=np.uint64)
        self._generator_inv = np.array(list(info.generator_inv), dtype=np.uint64)

    # ---- construction: EvaluationDomain::new returns None when the size exceeds the 2-adicity
    @classmethod
    def new(cls, num_coeffs: int, curve="bls12_381", ctx: Context | None = None):
        info = DomainInfo()
        rc = lib().zk_domain_new(get_curve(curve).curve_id, int(num_coeffs), ctypes.byref(info))
        if rc == _lib.ZK_ERR_DOMAIN_TOO_LARGE:
            return None
        check(rc, "zk_domain_new")
        return cls(info, curve, ctx)

    # ---- accessors (EvaluationDomain / EvaluationDomainExt)
    def size(self) -> int:
        return self._size

    def log_size_of_group(self) -> int:
        return self._log

    def size_inv(self):
        return self._size_inv.copy()

    def group_gen(self):
        return self._group_gen.copy()

    def group_gen_inv(self):
        return self._group_gen_inv.copy()

    def generator(self):
        return self._generator.copy()

    def generator_inv(self):
        return self._generator_inv.copy()

    # ---- transforms
    def _ctx_for(self, x) -> Context:
        if self._ctx is not None:
            return self._ctx
        dev = x.device.index if _is_torch(x) else 0
        return default_context(dev)

    def _run(self, kind: int, x, out=None):
        cid = self.curve.curve_id
        n = self._size
        ctx = self._ctx_for(x)
        if _is_torch(x):
            import torch
            in_len = check_dev_tensor(x, 4, ctx.device)
            if in_len > n:
                raise ValueError("input longer than the domain")  # ark panics (slice length mismatch on resize is truncation-free)
            if out is None:
                out = torch.empty((n, 4), dtype=x.dtype, device=x.device)
            else:
                if check_dev_tensor(out, 4, ctx.device) != n:
                    raise ValueError("output tensor must hold exactly domain.size() elements")
            ctx.use_torch_stream()
            check(lib().zk_ntt_dev(ctx.handle, cid, kind, self._log, ptr_of(x), in_len, ptr_of(out)), "zk_ntt_dev")
            return out
        a = as_host_u64(x, 4)
        if a.shape[0] > n:
            raise ValueError("input longer than the domain")
        if out is None:
            out = np.empty((n, 4), dtype=np.uint64)
        check(lib().zk_ntt(ctx.handle, cid, kind, self._log, ptr_of(a), a.shape[0], ptr_of(out)), "zk_ntt")
        return out

    def fft(self, coeffs):
        """Evaluations of the polynomial over the domain; input zero-extended to size()."""
        return self._run(KIND_FFT, coeffs)

    def ifft(self, evals):
        return self._run(KIND_IFFT, evals)

    def coset_fft(self, coeffs):
        return self._run(KIND_COSET_FFT, coeffs)

    def coset_ifft(self, evals):
        return self._run(KIND_COSET_IFFT, evals)

    def batch(self, kind: int, polys, outs=None):
        """n_polys transforms of one kind sharing the plan (the 13 coset_fft of quotient_poly.rs:72-120):
        host arrays -> zk_ntt_batch, device tensors -> zk_ntt_batch_dev (every pass of the whole batch is ONE launch).
        outs (device tensors only): preallocated result tensors of size() elements each.  Returns the list of outputs."""
        import ctypes
        polys = list(polys)
        k = len(polys)
        if k == 0:
            return []
        cid, n = self.curve.curve_id, self._size
        ctx = self._ctx_for(polys[0])
        ins = (ctypes.c_void_p * k)()
        outs_p = (ctypes.c_void_p * k)()
        lens = (ctypes.c_size_t * k)()
        if _is_torch(polys[0]):
            import torch
            res = []
            for i, x in enumerate(polys):
                lens[i] = check_dev_tensor(x, 4, ctx.device)
                if lens[i] > n:
                    raise ValueError("input longer than the domain")
                if outs is not None:
                    if check_dev_tensor(outs[i], 4, ctx.device) != n:
                        raise ValueError("output tensor must hold exactly domain.size() elements")
                    res.append(outs[i])
                else:
                    res.append(torch.empty((n, 4), dtype=x.dtype, device=x.device))
                ins[i], outs_p[i] = x.data_ptr(), res[i].data_ptr()
            ctx.use_torch_stream()
            check(lib().zk_ntt_batch_dev(ctx.handle, cid, kind, self._log, k, ins, lens, outs_p), "zk_ntt_batch_dev")
            return res
        arrs = [as_host_u64(x, 4) for x in polys]
        res = [np.empty((n, 4), dtype=np.uint64) for _ in arrs]
        for i, a in enumerate(arrs):
            if a.shape[0] > n:
                raise ValueError("input longer than the domain")
            lens[i] = a.shape[0]
            ins[i], outs_p[i] = a.ctypes.data, res[i].ctypes.data
        check(lib().zk_ntt_batch(ctx.handle, cid, kind, self._log, k, ins, lens, outs_p), "zk_ntt_batch")
        return res

    def _in_place(self, kind, buf):
        if _is_torch(buf):
            if buf.numel() != 4 * self._size:
                raise ValueError("in-place transforms need a buffer of exactly size() elements")
            return self._run(kind, buf, out=buf)
        a = as_host_u64(buf, 4)
        if a.shape[0] != self._size:
            raise ValueError("in-place transforms need a buffer of exactly size() elements")
        return self._run(kind, a, out=a)

    def fft_in_place(self, buf):
        return self._in_place(KIND_FFT, buf)

    def ifft_in_place(self, buf):
        return self._in_place(KIND_IFFT, buf)

    def coset_fft_in_place(self, buf):
        return self._in_place(KIND_COSET_FFT, buf)

    def coset_ifft_in_place(self, buf):
        return self._in_place(KIND_COSET_IFFT, buf)


# plonk-core names the enum `GeneralEvaluationDomain`; for the 2-adic fields it is always Radix2.
GeneralEvaluationDomain = Radix2EvaluationDomain
