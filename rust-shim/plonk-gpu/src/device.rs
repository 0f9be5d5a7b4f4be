//! The device-resident side of the boundary: what a patched `Prover::prove_with_preprocessed`
//! (patches/plonk-core-device-prover.patch) drives instead of the host-pointer calls of `kzg.rs` / `ntt_hook.rs`.
//!
//! The host-pointer calls move every vector across PCIe on every call -- 5.1 GB per proof at n = 2^20, which is why an unchanged
//! `Prover::prove` reaches 5.7 proofs/s where the device-resident schedule reaches 12.3 (bench.py `drop_in` leg against `value`).
//! Here a polynomial goes up once ([`DevicePoly::upload`]), is transformed and committed where it lies
//! ([`GpuDomain`], [`GpuKZG10::commit_dev`] ...), and PC calls whose results are needed only together are OPENED one by one and
//! CLOSED by one wait ([`GpuKZG10::round_begin`] / [`GpuKZG10::open_begin`] ... [`GpuKZG10::round_end`]):
//! `zk_kzg_round_begin_dev` / `zk_kzg_open_begin_dev` / `zk_kzg_round_reduce` / `zk_kzg_round_end` of include/ark_plonk_amd.h.
//!
//! Reference call sites served (plonk-core/src):
//! * `domain.ifft(&w_l_scalar)` x4, `PC::commit(ck, w_polys)`                    -- proof_system/prover.rs:196-213
//! * `domain.ifft(..)` of table / f / h_1 / h_2 and their three `PC::commit`s     -- prover.rs:240-242,281-291,302-317
//! * `PC::commit(z)`, `PC::commit(z_2)`                                           -- prover.rs:361-363,387-389
//! * `PC::commit(t_1..t_4)`                                                       -- prover.rs:459-469
//! * `PC::commit(aw)`, `PC::open(aw ++ w, z)`, `PC::commit(saw)`, `PC::open(saw, z omega)` -- prover.rs:579-618
//!
//! Round 5: the O(n) steps BETWEEN those calls too -- the round-2 multisets ([`lookup_query_dev`], [`combine_split_dev`]), both
//! grand products ([`perm_product_dev`], [`lookup_product_dev`]: permutation/mod.rs:652-822 up to their `ifft`), the pointwise
//! quotient ([`quotient_evals_dev`]: quotient_poly.rs:34-178 between its coset ffts and its `coset_ifft`), the 23 evaluations and the
//! linearisation polynomial ([`evaluate_dev`], [`lincomb_dev`]: linearisation_poly.rs:164-350) -- and the prover key's vectors resident
//! across proofs ([`resident`]), so that `prove_on_device` moves NO polynomial over PCIe between the witness upload and the proof.
//!
//! [`GpuBackend`] packages all of it behind `plonk_core::commitment::DeviceBackend`, the trait the patch adds to plonk-core, so the
//! prover stays generic in `PC` and free of GPU types.  Shipped as source (no Rust toolchain in this repository's pipeline);
//! `tests/test_rust_shim.py` checks every `sys::zk_*` call below against the header.

use crate::kzg::SrsHandle;
use crate::{check, ctx, fr_ptr, unpack_affine, GpuError, GpuKZG10, CURVE, FQ_LIMBS, FR_LIMBS};
use ark_bls12_381::{Fr, G1Affine};
use ark_ff::Zero;
use ark_poly_commit::kzg10;
use core::any::Any;
use core::cell::RefCell;
use core::ffi::c_void;
use plonk_core::commitment::{DeviceBackend, DeviceVec, QuotientChallenges, QuotientColumns, RoundItem, Transform};
use plonk_core::error::Error as PlonkError;
use plonk_gpu_sys as sys;
use std::collections::HashMap;
use std::sync::{Arc, Mutex, Once};

const FR_BYTES: usize = 8 * FR_LIMBS;

/// A vector of `Fr` (Montgomery limbs, the layout of `&[Fr]`) in HBM; freed on drop.
#[derive(Debug)]
pub struct DevicePoly {
    ptr: *mut c_void,
    len: usize,
}

// the allocation belongs to the device; the library serialises the calls of one ctx
unsafe impl Send for DevicePoly {}
unsafe impl Sync for DevicePoly {}

impl DevicePoly {
    /// `len` elements of uninitialised device memory.
    pub fn alloc(len: usize) -> Result<Self, GpuError> {
        let mut p: *mut c_void = core::ptr::null_mut();
        check(unsafe { sys::zk_dev_alloc(ctx(), core::cmp::max(len, 1) * FR_BYTES, &mut p) })?;
        Ok(DevicePoly { ptr: p, len })
    }

    /// Host -> device, once; the caller's slice is free again when this returns.
    pub fn upload(v: &[Fr]) -> Result<Self, GpuError> {
        let d = Self::alloc(v.len())?;
        if !v.is_empty() {
            check(unsafe { sys::zk_dev_upload(ctx(), d.ptr, fr_ptr(v) as *const c_void, v.len() * FR_BYTES) })?;
        }
        Ok(d)
    }

    /// Device -> host (waits for the work queued on the ctx stream before it: results of transforms included).
    pub fn download(&self) -> Result<Vec<Fr>, GpuError> {
        let mut v = vec![Fr::zero(); self.len];
        if self.len != 0 {
            check(unsafe { sys::zk_dev_download(ctx(), v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * FR_BYTES) })?;
        }
        Ok(v)
    }

    pub fn len(&self) -> usize {
        self.len
    }

    pub fn is_empty(&self) -> bool {
        self.len == 0
    }

    /// The first `len` elements as a vector of their own length (`DensePolynomial::from_coefficients_vec` strips trailing zeros,
    /// util.rs:175-184; the device has no reason to, but an opening's `n - 1` coefficients are a prefix).
    pub fn truncated(&self, len: usize) -> DevSlice<'_> {
        DevSlice { ptr: self.ptr as *const c_void, len: core::cmp::min(len, self.len), _owner: core::marker::PhantomData }
    }

    pub fn as_slice(&self) -> DevSlice<'_> {
        self.truncated(self.len)
    }
}

impl Drop for DevicePoly {
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            // zk_dev_free waits for the stream: kernels queued on the vector have finished when the memory goes
            let _ = unsafe { sys::zk_dev_free(ctx(), self.ptr) };
        }
    }
}

/// A borrowed range of a [`DevicePoly`] (what the `_dev` entry points take: pointer + length).
#[derive(Clone, Copy, Debug)]
pub struct DevSlice<'a> {
    ptr: *const c_void,
    len: usize,
    _owner: core::marker::PhantomData<&'a DevicePoly>,
}

impl<'a> DevSlice<'a> {
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    /// `count` elements from element `first` on (the four quarters of the quotient polynomial, prover.rs:107-123).
    pub fn range(&self, first: usize, count: usize) -> DevSlice<'a> {
        let first = core::cmp::min(first, self.len);
        let count = core::cmp::min(count, self.len - first);
        DevSlice { ptr: unsafe { (self.ptr as *const u8).add(first * FR_BYTES) } as *const c_void, len: count, _owner: core::marker::PhantomData }
    }
}

/// `Radix2EvaluationDomain` of size 2^log_n on the device: the four transforms of ark-poly's `EvaluationDomain` on device-resident
/// vectors, natural order in and out, inputs shorter than the domain zero-extended (fused into the first pass).
#[derive(Clone, Copy, Debug)]
pub struct GpuDomain {
    pub log_n: u32,
}

impl GpuDomain {
    /// `GeneralEvaluationDomain::new(num_coeffs)`: the next power of two (prover.rs:169-173).
    pub fn new(num_coeffs: usize) -> Self {
        let size = num_coeffs.next_power_of_two();
        GpuDomain { log_n: size.trailing_zeros() }
    }

    pub fn size(&self) -> usize {
        1usize << self.log_n
    }

    fn run(&self, kind: u32, input: DevSlice<'_>) -> Result<DevicePoly, GpuError> {
        if input.len() > self.size() {
            return Err(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("input longer than the domain") });
        }
        let out = DevicePoly::alloc(self.size())?;
        check(unsafe { sys::zk_ntt_dev(ctx(), CURVE, kind as i32, self.log_n, input.ptr, input.len(), out.ptr) })?;
        Ok(out)
    }

    pub fn fft_dev(&self, coeffs: DevSlice<'_>) -> Result<DevicePoly, GpuError> {
        self.run(sys::ZK_NTT_FFT, coeffs)
    }
    pub fn ifft_dev(&self, evals: DevSlice<'_>) -> Result<DevicePoly, GpuError> {
        self.run(sys::ZK_NTT_IFFT, evals)
    }
    /// `coset_fft` of `coeffs.len()` coefficients on this domain: the reference hands over n coefficients for the 4n domain
    /// (quotient_poly.rs:72-120); only those are read.
    pub fn coset_fft_dev(&self, coeffs: DevSlice<'_>) -> Result<DevicePoly, GpuError> {
        self.run(sys::ZK_NTT_COSET_FFT, coeffs)
    }
    pub fn coset_ifft_dev(&self, evals: DevSlice<'_>) -> Result<DevicePoly, GpuError> {
        self.run(sys::ZK_NTT_COSET_IFFT, evals)
    }

    /// Up to 16 independent transforms of one kind as ONE launch per pass (`zk_ntt_batch_dev`): the four wire iffts
    /// (prover.rs:196-203), h_1 / h_2 (prover.rs:302-305), the sigma ffts (permutation/mod.rs:671-674), the coset ffts of
    /// quotient_poly.rs:72-120.
    pub fn batch_dev(&self, kind: u32, inputs: &[DevSlice<'_>]) -> Result<Vec<DevicePoly>, GpuError> {
        let mut outs = Vec::with_capacity(inputs.len());
        for chunk in inputs.chunks(16) {
            let mut polys = Vec::with_capacity(chunk.len());
            for _ in chunk {
                polys.push(DevicePoly::alloc(self.size())?);
            }
            let ins: Vec<*const c_void> = chunk.iter().map(|s| s.ptr).collect();
            let lens: Vec<usize> = chunk.iter().map(|s| s.len()).collect();
            let ptrs: Vec<*mut c_void> = polys.iter().map(|p| p.ptr).collect();
            check(unsafe {
                sys::zk_ntt_batch_dev(ctx(), CURVE, kind as i32, self.log_n, chunk.len() as u32, ins.as_ptr(), lens.as_ptr(), ptrs.as_ptr())
            })?;
            outs.extend(polys);
        }
        Ok(outs)
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// the O(n) steps between the transforms and the commitments (SURVEY.md 8f N1 / N2 and the rounds around them)

fn fr_limbs(x: &Fr) -> *const u64 {
    fr_ptr(core::slice::from_ref(x))
}

/// `count` elements of `v` from `first` on as a vector of its own: one device-to-device copy queued on the ctx stream
/// (`Prover::split_tx_poly`, prover.rs:107-123).
pub fn slice_dev(v: DevSlice<'_>, first: usize, count: usize) -> Result<DevicePoly, GpuError> {
    let src = v.range(first, count);
    let out = DevicePoly::alloc(src.len())?;
    if !src.is_empty() {
        check(unsafe { sys::zk_dev_copy(ctx(), out.ptr, src.ptr, src.len() * FR_BYTES) })?;
    }
    Ok(out)
}

/// `sum_k scalars[k] * polys[k]` over `out_len` coefficients (`zk_poly_lincomb_dev`, at most 32 terms): the scalar-times-polynomial
/// sums of linearisation_poly.rs:288-336 and `MultiSet::compress` (prover.rs:229-237).
pub fn lincomb_dev(polys: &[DevSlice<'_>], scalars: &[Fr], out_len: usize) -> Result<DevicePoly, GpuError> {
    if polys.len() != scalars.len() || polys.len() > 32 {
        return Err(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("lincomb: one scalar per polynomial, at most 32 terms") });
    }
    let (ptrs, lens) = ptrs_and_lens(polys);
    let out = DevicePoly::alloc(out_len)?;
    check(unsafe {
        sys::zk_poly_lincomb_dev(ctx(), CURVE, polys.len() as u32, ptrs.as_ptr(), lens.as_ptr(), fr_ptr(scalars), out.ptr, out_len)
    })?;
    Ok(out)
}

/// `polys[k].evaluate(&points[k])` (`zk_poly_evaluate_dev`; linearisation_poly.rs:203-261 evaluates 16 polynomials at z and 7 at
/// z omega): blocks until the values are on the host.
pub fn evaluate_dev(polys: &[DevSlice<'_>], points: &[Fr]) -> Result<Vec<Fr>, GpuError> {
    if polys.len() != points.len() {
        return Err(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("evaluate: one point per polynomial") });
    }
    let mut out = vec![Fr::zero(); polys.len()];
    let mut lo = 0;
    while lo < polys.len() {
        let hi = core::cmp::min(lo + 32, polys.len());
        let (ptrs, lens) = ptrs_and_lens(&polys[lo..hi]);
        check(unsafe {
            sys::zk_poly_evaluate_dev(
                ctx(),
                CURVE,
                (hi - lo) as u32,
                ptrs.as_ptr(),
                lens.as_ptr(),
                fr_ptr(&points[lo..hi]),
                crate::fr_mut_ptr(&mut out[lo..hi]),
            )
        })?;
        lo = hi;
    }
    Ok(out)
}

/// The compressed query column of round 2 (prover.rs:244-279, `zk_lookup_query_dev`): n rows.
pub fn lookup_query_dev(n: usize, q_lookup: DevSlice<'_>, wires: [DevSlice<'_>; 4], zeta: &Fr, compressed_table: DevSlice<'_>) -> Result<DevicePoly, GpuError> {
    if compressed_table.is_empty() || wires.iter().any(|w| w.len() < n) {
        return Err(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("lookup_query: columns shorter than n") });
    }
    let w = [wires[0].ptr, wires[1].ptr, wires[2].ptr, wires[3].ptr];
    let out = DevicePoly::alloc(n)?;
    check(unsafe {
        sys::zk_lookup_query_dev(ctx(), CURVE, n, q_lookup.ptr, q_lookup.len(), w.as_ptr(), fr_limbs(zeta), compressed_table.ptr, out.ptr)
    })?;
    Ok(out)
}

/// `t.combine_split(&f)` (lookup/multiset.rs:131-176, `zk_lookup_combine_split_dev`) as (h_1, h_2); an element of f that t does
/// not hold is the library's ZK_ERR_NOT_INDEXED (the reference's `Error::ElementNotIndexed`).
pub fn combine_split_dev(t: DevSlice<'_>, f: DevSlice<'_>) -> Result<(DevicePoly, DevicePoly), GpuError> {
    let half = (t.len() + f.len() + 1) / 2;
    let mut h1 = DevicePoly::alloc(half)?;
    let mut h2 = DevicePoly::alloc(half)?;
    let (mut l1, mut l2) = (0usize, 0usize);
    check(unsafe { sys::zk_lookup_combine_split_dev(ctx(), CURVE, t.ptr, t.len(), f.ptr, f.len(), h1.ptr, h2.ptr, &mut l1, &mut l2) })?;
    h1.len = core::cmp::min(l1, half);
    h2.len = core::cmp::min(l2, half);
    Ok((h1, h2))
}

/// The n evaluations of the permutation polynomial z (`zk_perm_product_dev`): everything `compute_permutation_poly`
/// (permutation/mod.rs:652-752) does before its `domain.ifft`.  A zero denominator is ZK_ERR_NOT_INVERTIBLE (the reference panics).
pub fn perm_product_dev(log_n: u32, wires: [DevSlice<'_>; 4], sigma_evals: [DevSlice<'_>; 4], beta: &Fr, gamma: &Fr) -> Result<DevicePoly, GpuError> {
    let n = 1usize << log_n;
    if wires.iter().chain(sigma_evals.iter()).any(|v| v.len() < n) {
        return Err(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("perm_product: columns shorter than n") });
    }
    let w = [wires[0].ptr, wires[1].ptr, wires[2].ptr, wires[3].ptr];
    let s = [sigma_evals[0].ptr, sigma_evals[1].ptr, sigma_evals[2].ptr, sigma_evals[3].ptr];
    let out = DevicePoly::alloc(n)?;
    check(unsafe { sys::zk_perm_product_dev(ctx(), CURVE, log_n, w.as_ptr(), s.as_ptr(), fr_limbs(beta), fr_limbs(gamma), out.ptr, core::ptr::null_mut()) })?;
    Ok(out)
}

/// The n evaluations of the lookup product z_2 (`zk_lookup_product_dev`; permutation/mod.rs:754-822): columns (f, t, h_1, h_2).
pub fn lookup_product_dev(n: usize, columns: [DevSlice<'_>; 4], delta: &Fr, epsilon: &Fr) -> Result<DevicePoly, GpuError> {
    if columns.iter().any(|v| v.len() < n) {
        return Err(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("lookup_product: columns shorter than n") });
    }
    let out = DevicePoly::alloc(n)?;
    check(unsafe {
        sys::zk_lookup_product_dev(
            ctx(),
            CURVE,
            n,
            columns[0].ptr,
            columns[1].ptr,
            columns[2].ptr,
            columns[3].ptr,
            fr_limbs(delta),
            fr_limbs(epsilon),
            out.ptr,
            core::ptr::null_mut(),
        )
    })?;
    Ok(out)
}

fn limbs_of(x: &Fr) -> [u64; FR_LIMBS] {
    let mut l = [0u64; FR_LIMBS];
    // Fr = Fp256(BigInteger256([u64; 4])): the Montgomery limbs, as everywhere on this boundary (lib.rs `layout_checks`)
    unsafe { core::ptr::copy_nonoverlapping(fr_limbs(x), l.as_mut_ptr(), FR_LIMBS) };
    l
}

/// The 4n quotient evaluations over the coset (`zk_quotient_evals_dev`): every argument of `args` a vector of 4n evaluations.
pub fn quotient_evals_dev(log_n: u32, args: &sys::ZkQuotientArgs) -> Result<DevicePoly, GpuError> {
    let out = DevicePoly::alloc(4usize << log_n)?;
    check(unsafe { sys::zk_quotient_evals_dev(ctx(), CURVE, log_n, args as *const sys::ZkQuotientArgs, out.ptr) })?;
    Ok(out)
}

// ---- vectors that never change (the prover key): uploaded once per process, found again by the identity of the host slice.
// Key: (address, length); the fingerprint -- 64 elements sampled across the vector -- catches an allocator handing the same
// address to other data of the same length (a ProverKey is immutable once `Circuit::compile` has built it: its fields are
// `pub(crate)` and plonk-core never writes to one).
type ResidentMap = HashMap<(usize, usize), (u64, Arc<DevicePoly>)>;
static RESIDENT_INIT: Once = Once::new();
static mut RESIDENT: Option<Mutex<ResidentMap>> = None; // written once, under RESIDENT_INIT (the reference's pinned toolchain has no const Mutex::new)

fn resident_map() -> &'static Mutex<ResidentMap> {
    RESIDENT_INIT.call_once(|| unsafe { RESIDENT = Some(Mutex::new(HashMap::new())) });
    unsafe { RESIDENT.as_ref().expect("initialised by call_once") }
}

/// Fingerprint of the WHOLE vector (a sample of ~65 elements let a second prover key of the same size at the same
/// address, differing only in unsampled rows, reuse the first key's device copy -- every later proof silently invalid).  Four
/// interleaved FNV-1a lanes over the limbs, folded: a pass over 32 MiB takes a few milliseconds once per proof and column, against
/// the 12 ms upload it saves; the key is still (address, length), so a hit also needs the same allocation.
fn fingerprint(v: &[Fr]) -> u64 {
    let words = unsafe { core::slice::from_raw_parts(fr_ptr(v), v.len() * FR_LIMBS) };
    const P: u64 = 0x0000_0100_0000_01b3;
    let mut h = [0xcbf2_9ce4_8422_2325u64 ^ (v.len() as u64), 0x8422_2325_cbf2_9ce4, 0x9ce4_8422_2325_cbf2, 0x2325_cbf2_9ce4_8422];
    for chunk in words.chunks_exact(4) {
        for k in 0..4 {
            h[k] = (h[k] ^ chunk[k]).wrapping_mul(P);
        }
    }
    for (k, w) in words.chunks_exact(4).remainder().iter().enumerate() {
        h[k] = (h[k] ^ *w).wrapping_mul(P);
    }
    let mut out = h[0];
    for k in 1..4 {
        out = (out.rotate_left(23) ^ h[k]).wrapping_mul(P);
    }
    out
}

/// A shared handle to a resident vector.
#[derive(Clone, Debug)]
pub struct SharedPoly(pub Arc<DevicePoly>);

/// The device copy of a vector that does not change for the life of the process (a column of the prover key).
pub fn resident(v: &[Fr]) -> Result<SharedPoly, GpuError> {
    let key = (v.as_ptr() as usize, v.len());
    let fp = fingerprint(v);
    let mut map = resident_map().lock().unwrap_or_else(|e| e.into_inner());
    if let Some((have, poly)) = map.get(&key) {
        if *have == fp {
            return Ok(SharedPoly(poly.clone()));
        }
    }
    let poly = Arc::new(DevicePoly::upload(v)?);
    map.insert(key, (fp, poly.clone()));
    Ok(SharedPoly(poly))
}

/// Drop every resident vector (a service that switches circuits calls this when it drops a prover key).
pub fn forget_resident() {
    resident_map().lock().unwrap_or_else(|e| e.into_inner()).clear();
}

fn ptrs_and_lens(polys: &[DevSlice<'_>]) -> (Vec<*const c_void>, Vec<usize>) {
    (polys.iter().map(|s| s.ptr).collect(), polys.iter().map(|s| s.len()).collect())
}

fn points_from(xy: &[u64], inf: &[u8]) -> Vec<G1Affine> {
    (0..inf.len()).map(|k| unpack_affine(&xy[2 * FQ_LIMBS * k..2 * FQ_LIMBS * (k + 1)], inf[k])).collect()
}

impl GpuKZG10 {
    /// `PC::commit(ck, polys, None)` over device-resident coefficient vectors (Montgomery form; `into_repr` runs on the device):
    /// one blocking call, the jobs sorted / accumulated / reduced as one launch per kernel.  At most 16 polynomials.
    pub fn commit_dev(srs: &SrsHandle, polys: &[DevSlice<'_>]) -> Result<Vec<G1Affine>, GpuError> {
        let (ptrs, lens) = ptrs_and_lens(polys);
        let mut xy = vec![0u64; 2 * FQ_LIMBS * polys.len()];
        let mut inf = vec![0u8; polys.len()];
        check(unsafe {
            sys::zk_kzg_commit_batch_dev(ctx(), srs.raw(), polys.len() as u32, ptrs.as_ptr(), lens.as_ptr(), xy.as_mut_ptr(), inf.as_mut_ptr())
        })?;
        Ok(points_from(&xy, &inf))
    }

    /// `PC::open(ck, polys, _, point, challenge, _, None)`: sum_k challenge^k p_k, the witness (p - p(point)) / (X - point) and its
    /// commitment, all on the device.
    pub fn open_dev(srs: &SrsHandle, polys: &[DevSlice<'_>], point: &Fr, challenge: &Fr) -> Result<G1Affine, GpuError> {
        let (ptrs, lens) = ptrs_and_lens(polys);
        let mut xy = [0u64; 2 * FQ_LIMBS];
        let mut inf = 0u8;
        check(unsafe {
            sys::zk_kzg_open_dev(
                ctx(),
                srs.raw(),
                polys.len() as u32,
                ptrs.as_ptr(),
                lens.as_ptr(),
                fr_ptr(core::slice::from_ref(point)),
                fr_ptr(core::slice::from_ref(challenge)),
                xy.as_mut_ptr(),
                &mut inf,
            )
        })?;
        Ok(unpack_affine(&xy, inf))
    }

    /// One `PC::commit` call of a round: queued, not waited for.  The vectors are read when this returns in stream order (the digit
    /// kernel is queued here); keep them allocated until [`GpuKZG10::round_end`].
    pub fn round_begin(srs: &SrsHandle, polys: &[DevSlice<'_>]) -> Result<(), GpuError> {
        let (ptrs, lens) = ptrs_and_lens(polys);
        check(unsafe { sys::zk_kzg_round_begin_dev(ctx(), srs.raw(), polys.len() as u32, ptrs.as_ptr(), lens.as_ptr(), core::ptr::null()) })
    }

    /// One `PC::open` call of a round: the witness polynomial is built now, its MSM joins the round.
    pub fn open_begin(srs: &SrsHandle, polys: &[DevSlice<'_>], point: &Fr, challenge: &Fr) -> Result<(), GpuError> {
        let (ptrs, lens) = ptrs_and_lens(polys);
        check(unsafe {
            sys::zk_kzg_open_begin_dev(
                ctx(),
                srs.raw(),
                polys.len() as u32,
                ptrs.as_ptr(),
                lens.as_ptr(),
                fr_ptr(core::slice::from_ref(point)),
                fr_ptr(core::slice::from_ref(challenge)),
            )
        })
    }

    /// Queue the round's sort placement, accumulation and reduction kernels now; what the caller queues until `round_end`
    /// (transforms that do not depend on the round's results, downloads) runs behind them, under the host's part of the round.
    pub fn round_reduce() -> Result<(), GpuError> {
        check(unsafe { sys::zk_kzg_round_reduce(ctx()) })
    }

    /// Close the round: one point per job, in submission order (a commitment, or the witness commitment of an opening).
    pub fn round_end(n_jobs: usize) -> Result<Vec<G1Affine>, GpuError> {
        let mut xy = vec![0u64; 2 * FQ_LIMBS * core::cmp::max(n_jobs, 1)];
        let mut inf = vec![0u8; core::cmp::max(n_jobs, 1)];
        check(unsafe { sys::zk_kzg_round_end(ctx(), n_jobs as u32, xy.as_mut_ptr(), inf.as_mut_ptr()) })?;
        inf.truncate(n_jobs);
        Ok(points_from(&xy, &inf))
    }

    /// Drop an open round after a failure (waits for the queued kernels): the ctx takes blocking calls again.
    pub fn round_abort() {
        let _ = unsafe { sys::zk_kzg_round_abort(ctx()) };
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// plonk_core::commitment::DeviceBackend: what the patched prover sees

impl DeviceVec<Fr> for DevicePoly {
    fn len(&self) -> usize {
        self.len
    }
    fn to_host(&self) -> Result<Vec<Fr>, PlonkError> {
        self.download().map_err(device_error)
    }
    fn as_any(&self) -> &dyn Any {
        self
    }
}

impl DeviceVec<Fr> for SharedPoly {
    fn len(&self) -> usize {
        self.0.len
    }
    fn to_host(&self) -> Result<Vec<Fr>, PlonkError> {
        self.0.download().map_err(device_error)
    }
    fn as_any(&self) -> &dyn Any {
        // the vector itself: `as_poly` finds a `DevicePoly` behind an owned and behind a shared handle alike
        &*self.0
    }
}

fn device_error(e: GpuError) -> PlonkError {
    PlonkError::DeviceError { error: e.to_string() }
}

fn as_poly<'a>(v: &'a dyn DeviceVec<Fr>) -> Result<&'a DevicePoly, PlonkError> {
    v.as_any().downcast_ref::<DevicePoly>().ok_or_else(|| PlonkError::DeviceError { error: String::from("not a vector of this backend") })
}

/// The GPU behind `PC::device_backend(ck)`: one SRS (its window table resident), the process-wide ctx, and the kinds of the jobs
/// of the open round (a commitment or an opening) in submission order.
pub struct GpuBackend {
    srs: Arc<SrsHandle>,
    open_jobs: RefCell<Vec<bool>>, // true: the job is an opening (its result becomes kzg10::Proof { w, random_v: None })
}

impl GpuBackend {
    pub fn new(srs: Arc<SrsHandle>) -> Self {
        GpuBackend { srs, open_jobs: RefCell::new(Vec::new()) }
    }

    fn fail<T>(&self, e: GpuError) -> Result<T, PlonkError> {
        // a failed call may leave a round open on the ctx -- also one whose first `commit_begin` failed part-way, with jobs queued in
        // the library and none recorded here (the ABI keeps the jobs queued so far open): settle it unconditionally (a no-op without
        // an open round), or every later blocking call returns ZK_ERR_PENDING and the next proof appends to a stale round
        GpuKZG10::round_abort();
        self.open_jobs.borrow_mut().clear();
        Err(device_error(e))
    }
}

impl DeviceBackend<Fr, GpuKZG10> for GpuBackend {
    fn upload(&self, v: &[Fr]) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        match DevicePoly::upload(v) {
            Ok(d) => Ok(Box::new(d)),
            Err(e) => self.fail(e),
        }
    }

    fn transform_batch(&self, kind: Transform, domain_size: usize, inputs: &[&dyn DeviceVec<Fr>]) -> Result<Vec<Box<dyn DeviceVec<Fr>>>, PlonkError> {
        let dom = GpuDomain::new(domain_size);
        let k = match kind {
            Transform::Fft => sys::ZK_NTT_FFT,
            Transform::Ifft => sys::ZK_NTT_IFFT,
            Transform::CosetFft => sys::ZK_NTT_COSET_FFT,
            Transform::CosetIfft => sys::ZK_NTT_COSET_IFFT,
        };
        let mut slices = Vec::with_capacity(inputs.len());
        for v in inputs {
            slices.push(as_poly(*v)?.as_slice());
        }
        match dom.batch_dev(k, &slices) {
            Ok(outs) => Ok(outs.into_iter().map(|d| Box::new(d) as Box<dyn DeviceVec<Fr>>).collect()),
            Err(e) => self.fail(e),
        }
    }

    fn commit_begin(&self, polys: &[(&dyn DeviceVec<Fr>, usize)]) -> Result<(), PlonkError> {
        let mut slices = Vec::with_capacity(polys.len());
        for (v, len) in polys {
            slices.push(as_poly(*v)?.truncated(*len));
        }
        match GpuKZG10::round_begin(&self.srs, &slices) {
            Ok(()) => {
                self.open_jobs.borrow_mut().extend(core::iter::repeat(false).take(polys.len()));
                Ok(())
            }
            Err(e) => self.fail(e),
        }
    }

    fn open_begin(&self, polys: &[(&dyn DeviceVec<Fr>, usize)], point: &Fr, challenge: &Fr) -> Result<(), PlonkError> {
        let mut slices = Vec::with_capacity(polys.len());
        for (v, len) in polys {
            slices.push(as_poly(*v)?.truncated(*len));
        }
        match GpuKZG10::open_begin(&self.srs, &slices, point, challenge) {
            Ok(()) => {
                self.open_jobs.borrow_mut().push(true);
                Ok(())
            }
            Err(e) => self.fail(e),
        }
    }

    fn round_reduce(&self) -> Result<(), PlonkError> {
        match GpuKZG10::round_reduce() {
            Ok(()) => Ok(()),
            Err(e) => self.fail(e),
        }
    }

    fn round_end(&self) -> Result<Vec<RoundItem<Fr, GpuKZG10>>, PlonkError> {
        let kinds: Vec<bool> = self.open_jobs.borrow().clone();
        match GpuKZG10::round_end(kinds.len()) {
            Ok(points) => {
                self.open_jobs.borrow_mut().clear();
                Ok(points
                    .into_iter()
                    .zip(kinds)
                    .map(|(g, is_open)| {
                        if is_open {
                            RoundItem::Opening(kzg10::Proof { w: g, random_v: None })
                        } else {
                            RoundItem::Commitment(kzg10::Commitment(g))
                        }
                    })
                    .collect())
            }
            Err(e) => self.fail(e),
        }
    }

    fn resident(&self, v: &[Fr]) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        match resident(v) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }

    fn slice(&self, v: &dyn DeviceVec<Fr>, first: usize, count: usize) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        match slice_dev(as_poly(v)?.as_slice(), first, count) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }

    fn lincomb(&self, terms: &[(&dyn DeviceVec<Fr>, usize, Fr)], out_len: usize) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        let mut slices = Vec::with_capacity(terms.len());
        let mut scalars = Vec::with_capacity(terms.len());
        for (v, len, s) in terms {
            slices.push(as_poly(*v)?.truncated(*len));
            scalars.push(*s);
        }
        match lincomb_dev(&slices, &scalars, out_len) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }

    fn evaluate(&self, polys: &[(&dyn DeviceVec<Fr>, usize, Fr)]) -> Result<Vec<Fr>, PlonkError> {
        let mut slices = Vec::with_capacity(polys.len());
        let mut points = Vec::with_capacity(polys.len());
        for (v, len, p) in polys {
            slices.push(as_poly(*v)?.truncated(*len));
            points.push(*p);
        }
        match evaluate_dev(&slices, &points) {
            Ok(v) => Ok(v),
            Err(e) => self.fail(e),
        }
    }

    fn lookup_query(
        &self,
        n: usize,
        q_lookup: (&dyn DeviceVec<Fr>, usize),
        wires: [&dyn DeviceVec<Fr>; 4],
        zeta: &Fr,
        compressed_table: &dyn DeviceVec<Fr>,
    ) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        let w = [as_poly(wires[0])?.as_slice(), as_poly(wires[1])?.as_slice(), as_poly(wires[2])?.as_slice(), as_poly(wires[3])?.as_slice()];
        match lookup_query_dev(n, as_poly(q_lookup.0)?.truncated(q_lookup.1), w, zeta, as_poly(compressed_table)?.as_slice()) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }

    fn combine_split(&self, t: &dyn DeviceVec<Fr>, f: &dyn DeviceVec<Fr>) -> Result<(Box<dyn DeviceVec<Fr>>, Box<dyn DeviceVec<Fr>>), PlonkError> {
        match combine_split_dev(as_poly(t)?.as_slice(), as_poly(f)?.as_slice()) {
            Ok((h1, h2)) => Ok((Box::new(h1), Box::new(h2))),
            Err(e) => self.fail(e),
        }
    }

    fn permutation_product(
        &self,
        n: usize,
        wires: [&dyn DeviceVec<Fr>; 4],
        sigma_evals: [&dyn DeviceVec<Fr>; 4],
        beta: &Fr,
        gamma: &Fr,
    ) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        if !n.is_power_of_two() {
            return self.fail(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("permutation_product: n must be a power of two") });
        }
        let w = [as_poly(wires[0])?.as_slice(), as_poly(wires[1])?.as_slice(), as_poly(wires[2])?.as_slice(), as_poly(wires[3])?.as_slice()];
        let s = [
            as_poly(sigma_evals[0])?.as_slice(),
            as_poly(sigma_evals[1])?.as_slice(),
            as_poly(sigma_evals[2])?.as_slice(),
            as_poly(sigma_evals[3])?.as_slice(),
        ];
        match perm_product_dev(n.trailing_zeros(), w, s, beta, gamma) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }

    fn lookup_product(&self, n: usize, columns: [&dyn DeviceVec<Fr>; 4], delta: &Fr, epsilon: &Fr) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        let c = [
            as_poly(columns[0])?.as_slice(),
            as_poly(columns[1])?.as_slice(),
            as_poly(columns[2])?.as_slice(),
            as_poly(columns[3])?.as_slice(),
        ];
        match lookup_product_dev(n, c, delta, epsilon) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }

    fn quotient(&self, n: usize, columns: &QuotientColumns<'_, Fr>, challenges: &QuotientChallenges<Fr>) -> Result<Box<dyn DeviceVec<Fr>>, PlonkError> {
        if !n.is_power_of_two() {
            return self.fail(GpuError { code: sys::ZK_ERR_BAD_ARG, message: String::from("quotient: n must be a power of two") });
        }
        // every column holds the 4n evaluations over the coset
        let col = |v: &dyn DeviceVec<Fr>| -> Result<*const c_void, PlonkError> {
            let p = as_poly(v)?;
            if p.len() < 4 * n {
                return Err(PlonkError::DeviceError { error: String::from("quotient: a column holds fewer than 4n evaluations") });
            }
            Ok(p.ptr as *const c_void)
        };
        let args = sys::ZkQuotientArgs {
            w_l: col(columns.wires[0])?,
            w_r: col(columns.wires[1])?,
            w_o: col(columns.wires[2])?,
            w_4: col(columns.wires[3])?,
            z: col(columns.z)?,
            z2: col(columns.z2)?,
            f: col(columns.f)?,
            table: col(columns.table)?,
            h1: col(columns.h1)?,
            h2: col(columns.h2)?,
            pi: col(columns.pi)?,
            l1: col(columns.l1)?,
            q_m: col(columns.arithmetic[0])?,
            q_l: col(columns.arithmetic[1])?,
            q_r: col(columns.arithmetic[2])?,
            q_o: col(columns.arithmetic[3])?,
            q_4: col(columns.arithmetic[4])?,
            q_c: col(columns.arithmetic[5])?,
            q_arith: col(columns.arithmetic[6])?,
            q_range: col(columns.selectors[0])?,
            q_logic: col(columns.selectors[1])?,
            q_fixed_group_add: col(columns.selectors[2])?,
            q_variable_group_add: col(columns.selectors[3])?,
            q_lookup: col(columns.selectors[4])?,
            sigma: [col(columns.sigma[0])?, col(columns.sigma[1])?, col(columns.sigma[2])?, col(columns.sigma[3])?],
            alpha: limbs_of(&challenges.round[0]),
            beta: limbs_of(&challenges.round[1]),
            gamma: limbs_of(&challenges.round[2]),
            delta: limbs_of(&challenges.round[3]),
            epsilon: limbs_of(&challenges.round[4]),
            zeta: limbs_of(&challenges.round[5]),
            range_challenge: limbs_of(&challenges.separation[0]),
            logic_challenge: limbs_of(&challenges.separation[1]),
            fixed_base_challenge: limbs_of(&challenges.separation[2]),
            var_base_challenge: limbs_of(&challenges.separation[3]),
            lookup_challenge: limbs_of(&challenges.separation[4]),
            coeff_a: limbs_of(&challenges.curve[0]),
            coeff_d: limbs_of(&challenges.curve[1]),
        };
        match quotient_evals_dev(n.trailing_zeros(), &args) {
            Ok(p) => Ok(Box::new(p)),
            Err(e) => self.fail(e),
        }
    }
}
