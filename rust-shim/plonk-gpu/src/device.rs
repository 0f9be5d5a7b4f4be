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
use plonk_core::commitment::{DeviceBackend, DeviceVec, RoundItem, Transform};
use plonk_core::error::Error as PlonkError;
use plonk_gpu_sys as sys;
use std::sync::Arc;

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
        // a failed call leaves the round open on the ctx: settle it, or every later blocking call returns ZK_ERR_PENDING
        if !self.open_jobs.borrow().is_empty() {
            GpuKZG10::round_abort();
            self.open_jobs.borrow_mut().clear();
        }
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
}
