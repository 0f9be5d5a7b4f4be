//! What the patched `ark-ec` calls (patches/ark-ec-0.3.0.md) from `VariableBaseMSM::multi_scalar_mul` -- for callers that reach
//! the MSM without going through a `PolynomialCommitment` (commitment.rs:45,83 and anything else in the ark ecosystem).
//! `GpuKZG10::commit` does not need this hook: it hands whole polynomials to `zk_kzg_commit_batch`.
use crate::{ctx, pack_affine, unpack_affine, CURVE, FQ_LIMBS};
use ark_bls12_381::{Fr, G1Affine, G1Projective};
use ark_ff::{BigInteger256, PrimeField};
use core::any::TypeId;
use plonk_gpu_sys as sys;

/// `multi_scalar_mul(bases, scalars)` for `G = ark_bls12_381::G1Affine`: `Some(sum)` when the GPU did it.  The reference
/// truncates to the shorter slice and is infallible; scalars are `BigInteger256` = canonical (non-Montgomery) limbs, which
/// is what the ABI takes.
pub fn try_msm<G: 'static, S: 'static>(bases: &[G], scalars: &[S]) -> Option<G1Projective> {
    if TypeId::of::<G>() != TypeId::of::<G1Affine>() || TypeId::of::<S>() != TypeId::of::<<Fr as PrimeField>::BigInt>() {
        return None;
    }
    let n = core::cmp::min(bases.len(), scalars.len());
    if n < 1 << 12 {
        return None; // below a few thousand points the upload costs more than ark's Pippenger
    }
    let c = ctx();
    if c.is_null() {
        return None;
    }
    // SAFETY: the TypeId checks above
    let b: &[G1Affine] = unsafe { core::slice::from_raw_parts(bases.as_ptr() as *const G1Affine, n) };
    let s: &[BigInteger256] = unsafe { core::slice::from_raw_parts(scalars.as_ptr() as *const BigInteger256, n) };
    let (xy, inf) = pack_affine(b);
    let mut out = [0u64; 2 * FQ_LIMBS];
    let mut out_inf = 0u8;
    // BigInteger256 is `[u64; 4]` little-endian: n x 4 limbs, contiguous
    let rc = unsafe { sys::zk_msm_g1(c, CURVE, xy.as_ptr(), inf.as_ptr(), s.as_ptr() as *const u64, n, out.as_mut_ptr(), &mut out_inf) };
    if rc != sys::ZK_OK {
        return None;
    }
    Some(unpack_affine(&out, out_inf).into())
}
