//! What the patched `ark-poly` calls (patches/ark-poly-0.3.0.md): the reference names `GeneralEvaluationDomain` concretely
//! (prover.rs:23,169; quotient_poly.rs:19-22), so there is no plugin point above `Radix2EvaluationDomain`'s own methods.
//!
//! `DomainCoeff<F>` is a blanket impl for any `Copy + Send + Sync + Add + Sub + AddAssign + SubAssign + Zero + MulAssign<F>`
//! type; the reference only ever transforms `T = F = ark_bls12_381::Fr`.  The hook recognises exactly that case by `TypeId`
//! and reports "not handled" for everything else, so the patched methods fall through to ark's own code.
use crate::{check, ctx, fr_mut_ptr, CURVE};
use ark_bls12_381::Fr;
use ark_ff::Zero;
use core::any::TypeId;
use plonk_gpu_sys as sys;

/// `kind`: `ZK_NTT_FFT` / `ZK_NTT_IFFT` / `ZK_NTT_COSET_FFT` / `ZK_NTT_COSET_IFFT` -- the four `*_in_place` methods.
///
/// Contract of the reference methods (ark-poly 0.3.0 `domain/radix2/mod.rs`, `domain/mod.rs`): the vector is resized to
/// `domain.size()` with zeros, transformed in natural order, and the methods are infallible.  Returns `true` when the transform
/// was done on the GPU (the vector then has `size` elements); `false` leaves `coeffs` untouched for the CPU path.
pub fn try_transform_in_place<F: 'static, T: 'static>(kind: u32, log_size_of_group: u32, size: usize, coeffs: &mut Vec<T>) -> bool {
    if TypeId::of::<T>() != TypeId::of::<Fr>() || TypeId::of::<F>() != TypeId::of::<Fr>() {
        return false;
    }
    if coeffs.len() > size || log_size_of_group < 10 {
        return false; // a longer input is ark's error to report; tiny transforms are faster where they are
    }
    let c = ctx();
    if c.is_null() {
        return false;
    }
    // SAFETY: T is Fr (checked above): same layout, same allocator
    let v: &mut Vec<Fr> = unsafe { &mut *(coeffs as *mut Vec<T> as *mut Vec<Fr>) };
    let in_len = v.len(); // the zero extension is fused into the first pass on the device: only in_len elements are uploaded
    v.resize(size, Fr::zero());
    let p = fr_mut_ptr(v.as_mut_slice());
    let rc = unsafe { sys::zk_ntt(c, CURVE, kind as i32, log_size_of_group, p as *const u64, in_len, p) };
    if check(rc).is_ok() {
        return true;
    }
    v.truncate(in_len); // back to what the caller handed in: the CPU path resizes again
    false
}
