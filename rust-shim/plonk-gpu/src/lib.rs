//! Rust side of the drop-in boundary of `libark_plonk_amd.so` for heliaxdev/ark-plonk.
//!
//! * [`GpuKZG10`] -- a `plonk_core::commitment::HomomorphicCommitment<Fr>` (plonk-core/src/commitment.rs:8-19) with the key,
//!   commitment and proof types of `SonicKZG10<Bls12_381, DensePolynomial<Fr>>`; `trim` additionally parks the SRS on the GPU,
//!   `commit` and `open` run there (`zk_kzg_commit_batch`, `zk_kzg_open`), everything else is SonicKZG10's.  It satisfies
//!   `Prover::<Fr, P, PC>` (proof_system/prover.rs:32-37) and `Circuit::gen_proof::<PC>` (circuit.rs:264-287) unchanged.
//! * [`device`] -- the device-resident form: [`DevicePoly`], [`GpuDomain`], the deferred rounds of [`GpuKZG10`] and [`GpuBackend`], the
//!   `plonk_core::commitment::DeviceBackend` that patches/plonk-core-device-prover.patch lets `Prover::prove_with_preprocessed` drive
//!   (the path bench.py's headline measures; the host-pointer calls above are its `drop_in` leg).
//! * [`ntt_hook`] / [`msm_hook`] -- what the patched `ark-poly` / `ark-ec` (patches/) call from
//!   `Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place` and `VariableBaseMSM::multi_scalar_mul`.
//!
//! This crate is shipped as source: the repository's pipeline has no Rust toolchain.  All numerical semantics are pinned on the
//! C ABI (tests/), and `tests/test_rust_shim.py` checks every `zk_*` call below against `include/ark_plonk_amd.h`.
//!
//! Layout assumptions (ark 0.3.0), asserted at start-up by [`layout_checks`]:
//! * `ark_bls12_381::Fr` / `Fq` are `Fp256` / `Fp384` = `(BigInteger([u64; N]), PhantomData)`: `size_of == 8 N`, the limbs are the
//!   little-endian Montgomery residue -- exactly what the ABI calls "Montgomery limbs".
//! * `GroupAffine<P>` is `{ x, y, infinity: bool, PhantomData }` and NOT `repr(C)`: points are copied field by field.

pub mod device;
pub mod kzg;
pub mod msm_hook;
pub mod ntt_hook;

pub use device::{DevSlice, DevicePoly, GpuBackend, GpuDomain};
pub use kzg::{gpu_fallbacks, GpuCommitterKey, GpuKZG10};

use ark_bls12_381::{Fq, Fr, G1Affine};
use ark_ff::Zero;
use plonk_gpu_sys as sys;
use std::sync::Once;

/// BLS12-381 in the ABI's numbering (`ZK_CURVE_BLS12_381`).
pub const CURVE: i32 = sys::ZK_CURVE_BLS12_381;
/// u64 limbs of a base-field element.
pub const FQ_LIMBS: usize = 6;
/// u64 limbs of a scalar.
pub const FR_LIMBS: usize = 4;

/// Error of a failed ABI call: the code and `zk_strerror`'s text.
#[derive(Debug, Clone)]
pub struct GpuError {
    pub code: i32,
    pub message: String,
}

impl core::fmt::Display for GpuError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "ark_plonk_amd: {} (code {})", self.message, self.code)
    }
}

impl std::error::Error for GpuError {}

/// `0` is success, everything else an error (include/ark_plonk_amd.h, "error codes").
pub fn check(rc: i32) -> Result<(), GpuError> {
    if rc == sys::ZK_OK {
        return Ok(());
    }
    let message = unsafe {
        let p = sys::zk_strerror(rc);
        if p.is_null() {
            String::from("unknown error")
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    };
    Err(GpuError { code: rc, message })
}

struct CtxCell(*mut sys::ZkCtx);
// the library serialises calls on one ctx with an internal mutex (include/ark_plonk_amd.h, "Threading")
unsafe impl Sync for CtxCell {}
unsafe impl Send for CtxCell {}

static INIT: Once = Once::new();
static mut CTX: CtxCell = CtxCell(core::ptr::null_mut());
static mut PER_THREAD: bool = false;

fn env_is_one(name: &str) -> bool {
    std::env::var(name).map(|v| v == "1").unwrap_or(false)
}

/// A `zk_ctx` on GPU `ARK_PLONK_AMD_DEVICE` (default 0) with the shim's opt-in caches applied; null when no usable device exists.
fn new_ctx() -> *mut sys::ZkCtx {
    let device = std::env::var("ARK_PLONK_AMD_DEVICE").ok().and_then(|v| v.parse::<i32>().ok()).unwrap_or(0);
    let mut c: *mut sys::ZkCtx = core::ptr::null_mut();
    let rc = unsafe { sys::zk_ctx_create(device, &mut c) };
    if rc != sys::ZK_OK {
        return core::ptr::null_mut();
    }
    // opt-in: the twelve polynomials prover.rs:569-607 commits a second time are served from the library's
    // content-addressed commitment cache (17 MSMs per proof instead of 29; see the header for the trust model)
    if env_is_one("ARK_PLONK_AMD_COMMIT_CACHE") {
        let _ = unsafe { sys::zk_ctx_set_commit_cache(c, 1, 0) };
    }
    // opt-in: the host-pointer hooks stop re-uploading what the library itself produced -- an `ifft` output that comes back as a
    // `PC::commit`, `coset_fft` or `PC::open` input is found by a digest of its bytes and used where it lies (58 of the 83
    // vectors an unchanged Prover::prove uploads per proof; see the header for the trust model and the sizes)
    if env_is_one("ARK_PLONK_AMD_RESIDENCY_CACHE") {
        let _ = unsafe { sys::zk_ctx_set_residency_cache(c, 1, 0, 0) };
    }
    c
}

/// The calling thread's own `zk_ctx` (`ARK_PLONK_AMD_CTX_PER_THREAD=1`), destroyed when the thread ends.
struct ThreadCtx(core::cell::Cell<*mut sys::ZkCtx>, core::cell::Cell<bool>);

impl Drop for ThreadCtx {
    fn drop(&mut self) {
        let c = self.0.get();
        if !c.is_null() {
            unsafe { sys::zk_ctx_destroy(c) };
        }
    }
}

thread_local! {
    static THREAD_CTX: ThreadCtx = ThreadCtx(core::cell::Cell::new(core::ptr::null_mut()), core::cell::Cell::new(false));
}

/// The `zk_ctx` the hooks of the calling thread use; null when no usable device exists, in which case every hook reports "not handled"
/// and the caller's CPU path runs.  By default ONE process-wide context: the library serialises the calls of all threads on it.
/// With `ARK_PLONK_AMD_CTX_PER_THREAD=1` every thread that calls a hook gets its own context (its own HIP stream, staging buffers and
/// caches; the registered SRS and its window table belong to the GPU and are shared): a service that runs `Prover::prove` in T worker
/// threads then has one thread's PCIe transfers under the other threads' kernels -- 5.7 proofs/s with one caller, 9.4-9.8 with four,
/// 11.0 with eight on one MI355X at n = 2^20 (profiles/r05/r05_drop_in_callers.txt).  Meant for callers that drive the prover from a few
/// long-lived threads (prover.rs calls every hook from the thread that called `prove`), not from a large work-stealing pool.
pub fn ctx() -> *mut sys::ZkCtx {
    INIT.call_once(|| {
        layout_checks();
        unsafe { PER_THREAD = env_is_one("ARK_PLONK_AMD_CTX_PER_THREAD") };
        if !unsafe { PER_THREAD } {
            let c = new_ctx();
            unsafe { CTX = CtxCell(c) };
        }
    });
    if unsafe { PER_THREAD } {
        return THREAD_CTX.with(|t| {
            if !t.1.get() {
                t.1.set(true); // one attempt per thread: a thread without a usable device stays on the CPU path
                t.0.set(new_ctx());
            }
            t.0.get()
        });
    }
    unsafe { CTX.0 }
}

/// The field and point layouts this crate relies on.
pub fn layout_checks() {
    assert_eq!(core::mem::size_of::<Fr>(), 8 * FR_LIMBS, "ark_bls12_381::Fr is not 4 x u64");
    assert_eq!(core::mem::size_of::<Fq>(), 8 * FQ_LIMBS, "ark_bls12_381::Fq is not 6 x u64");
}

/// `&[Fr]` as the ABI's `const uint64_t*` (4 Montgomery limbs per element).
pub fn fr_ptr(v: &[Fr]) -> *const u64 {
    v.as_ptr() as *const u64
}

/// `&mut [Fr]` as the ABI's `uint64_t*`.
pub fn fr_mut_ptr(v: &mut [Fr]) -> *mut u64 {
    v.as_mut_ptr() as *mut u64
}

fn fq_limbs(x: &Fq) -> [u64; FQ_LIMBS] {
    // Fp384(pub BigInteger384(pub [u64; 6]), PhantomData): the Montgomery residue
    (x.0).0
}

fn fq_from_limbs(l: &[u64]) -> Fq {
    let mut b = ark_ff::BigInteger384::default();
    b.0.copy_from_slice(&l[..FQ_LIMBS]);
    ark_ff::Fp384::new(b) // `new` takes the internal (Montgomery) representation in ark-ff 0.3
}

/// `[G1Affine]` (not `repr(C)`) -> packed `x || y` Montgomery limbs (12 u64 per point) + one infinity flag per point.
pub fn pack_affine(points: &[G1Affine]) -> (Vec<u64>, Vec<u8>) {
    let mut xy = Vec::with_capacity(points.len() * 2 * FQ_LIMBS);
    let mut inf = Vec::with_capacity(points.len());
    for p in points {
        xy.extend_from_slice(&fq_limbs(&p.x));
        xy.extend_from_slice(&fq_limbs(&p.y));
        inf.push(p.infinity as u8);
    }
    (xy, inf)
}

/// One affine point back from the ABI (`out_xy`: 12 limbs, `out_inf`: flag).  Infinity is arkworks' `G1Affine::zero()`.
pub fn unpack_affine(xy: &[u64], inf: u8) -> G1Affine {
    if inf != 0 {
        return G1Affine::zero();
    }
    G1Affine::new(fq_from_limbs(&xy[..FQ_LIMBS]), fq_from_limbs(&xy[FQ_LIMBS..2 * FQ_LIMBS]), false)
}
