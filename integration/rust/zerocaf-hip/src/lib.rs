//! zerocaf-hip: batched MI355X backend for the `zerocaf` crate.
//!
//! Binds `libzerocaf_hip.so` (C ABI: `include/zerocaf_hip.h`, declarations generated into
//! [`ffi`]) and exposes batch counterparts of zerocaf's operator surface: one method per
//! reference operation, slices of the reference's own `Copy` types in, `Vec`s of them out.
//! Per-element panics / `None` of the reference become `Option`s; a failed call becomes
//! `Err(HipError)`.  Single elements keep using the CPU crate.
//!
//! Inside the zerocaf crate the same file is `src/backend/hip/mod.rs` with `zerocaf::`
//! replaced by `crate::`.
//!
//! NOTE: the build image of this repository has no Rust toolchain, so this crate has not been
//! compiled there.  The ABI underneath is exercised end to end through ctypes and C++.
#![allow(non_snake_case)]

pub mod ffi;

use std::ffi::CStr;
use std::fmt;
use std::os::raw::{c_int, c_uint, c_void};

use zerocaf::edwards::{AffinePoint, CompressedEdwardsY, EdwardsPoint, ProjectivePoint};
use zerocaf::field::FieldElement;
use zerocaf::ristretto::{CompressedRistretto, RistrettoPoint};
use zerocaf::scalar::Scalar;

use ffi::ZcCtx;

// ------------------------------------------------------------------ errors
/// A call on the whole batch failed (bad argument, HIP error, no device).
#[derive(Debug, Clone)]
pub struct HipError {
    /// Negative `zc_status` value.
    pub code: i32,
    /// `zc_last_error()` of the calling thread.
    pub message: String,
}

impl fmt::Display for HipError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        write!(f, "zerocaf_hip error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for HipError {}

pub type Result<T> = std::result::Result<T, HipError>;

fn check(rc: c_int) -> Result<()> {
    if rc == ffi::ZC_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(ffi::zc_last_error()) }
        .to_string_lossy()
        .into_owned();
    Err(HipError { code: rc as i32, message })
}

/// Which reference algorithm's formula sequence `ed_mul` reproduces (`src/edwards.rs:102-153`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum MulAlgorithm {
    /// `double_and_add` = `Mul<Scalar>`: identical (X:Y:Z:T) limbs.
    DoubleAndAdd,
    /// `ltr_bin_mul`: identical limbs.
    LtrBinMul,
    /// `binary_naf_mul`: identical limbs.
    BinaryNafMul,
    /// Windowed core: the same group element (`==`, same encodings), different limbs.
    Fast,
}

impl MulAlgorithm {
    fn flags(self) -> c_uint {
        match self {
            MulAlgorithm::DoubleAndAdd => ffi::ZC_SCALAR_MUL_STRICT,
            MulAlgorithm::LtrBinMul => ffi::ZC_SCALAR_MUL_LTR_BIN,
            MulAlgorithm::BinaryNafMul => ffi::ZC_SCALAR_MUL_BINARY_NAF,
            MulAlgorithm::Fast => ffi::ZC_SCALAR_MUL_FAST,
        }
    }
}

// ------------------------------------------------------------------ flat buffers
// The reference structs carry no layout guarantee (not #[repr(C)]), so batches are copied
// into the flat layout of the ABI: FE / Scalar = 5 x u64, point = X|Y|Z|T = 20 x u64.
fn limbs(c: &[u64]) -> [u64; 5] {
    let mut l = [0u64; 5];
    l.copy_from_slice(&c[..5]);
    l
}

fn bytes32(c: &[u8]) -> [u8; 32] {
    let mut a = [0u8; 32];
    a.copy_from_slice(&c[..32]);
    a
}

fn flat_fe(v: &[FieldElement]) -> Vec<u64> {
    let mut o = Vec::with_capacity(v.len() * 5);
    for x in v {
        o.extend_from_slice(&x.0);
    }
    o
}

fn flat_sc(v: &[Scalar]) -> Vec<u64> {
    let mut o = Vec::with_capacity(v.len() * 5);
    for x in v {
        o.extend_from_slice(&x.0);
    }
    o
}

fn flat_ed(v: &[EdwardsPoint]) -> Vec<u64> {
    let mut o = Vec::with_capacity(v.len() * 20);
    for p in v {
        o.extend_from_slice(&p.X.0);
        o.extend_from_slice(&p.Y.0);
        o.extend_from_slice(&p.Z.0);
        o.extend_from_slice(&p.T.0);
    }
    o
}

fn flat_ris(v: &[RistrettoPoint]) -> Vec<u64> {
    let mut o = Vec::with_capacity(v.len() * 20);
    for p in v {
        o.extend_from_slice(&(p.0).X.0);
        o.extend_from_slice(&(p.0).Y.0);
        o.extend_from_slice(&(p.0).Z.0);
        o.extend_from_slice(&(p.0).T.0);
    }
    o
}

fn flat_proj(v: &[ProjectivePoint]) -> Vec<u64> {
    let mut o = Vec::with_capacity(v.len() * 15);
    for p in v {
        o.extend_from_slice(&p.X.0);
        o.extend_from_slice(&p.Y.0);
        o.extend_from_slice(&p.Z.0);
    }
    o
}

fn unflat_fe(v: &[u64]) -> Vec<FieldElement> {
    v.chunks_exact(5).map(|c| FieldElement(limbs(c))).collect()
}

fn unflat_sc(v: &[u64]) -> Vec<Scalar> {
    v.chunks_exact(5).map(|c| Scalar(limbs(c))).collect()
}

fn unflat_ed(v: &[u64]) -> Vec<EdwardsPoint> {
    v.chunks_exact(20)
        .map(|c| EdwardsPoint {
            X: FieldElement(limbs(&c[0..5])),
            Y: FieldElement(limbs(&c[5..10])),
            Z: FieldElement(limbs(&c[10..15])),
            T: FieldElement(limbs(&c[15..20])),
        })
        .collect()
}

fn unflat_ris(v: &[u64]) -> Vec<RistrettoPoint> {
    unflat_ed(v).into_iter().map(RistrettoPoint).collect()
}

fn unflat_proj(v: &[u64]) -> Vec<ProjectivePoint> {
    v.chunks_exact(15)
        .map(|c| ProjectivePoint {
            X: FieldElement(limbs(&c[0..5])),
            Y: FieldElement(limbs(&c[5..10])),
            Z: FieldElement(limbs(&c[10..15])),
        })
        .collect()
}

fn unflat_b32(v: &[u8]) -> Vec<[u8; 32]> {
    v.chunks_exact(32).map(bytes32).collect()
}

fn masked<T>(vals: Vec<T>, ok: &[u8]) -> Vec<Option<T>> {
    vals.into_iter()
        .zip(ok.iter())
        .map(|(v, &o)| if o == 1 { Some(v) } else { None })
        .collect()
}

fn flags(v: Vec<u8>) -> Vec<bool> {
    v.into_iter().map(|b| b == 1).collect()
}

type Bin = unsafe extern "C" fn(*mut ZcCtx, *const u64, *const u64, *mut u64, usize) -> c_int;
type Un = unsafe extern "C" fn(*mut ZcCtx, *const u64, *mut u64, usize) -> c_int;
type Flag = unsafe extern "C" fn(*mut ZcCtx, *const u64, *mut u8, usize) -> c_int;

// ------------------------------------------------------------------ the backend
/// One context of `libzerocaf_hip.so`: streams and device scratch of one or more GPUs.
/// Calls are serialised inside the library, so a backend may be shared between threads.
pub struct HipBackend {
    ctx: *mut ZcCtx,
}

unsafe impl Send for HipBackend {}
unsafe impl Sync for HipBackend {}

impl Drop for HipBackend {
    fn drop(&mut self) {
        unsafe {
            ffi::zc_ctx_destroy(self.ctx);
        }
    }
}

impl HipBackend {
    /// `devices = &[]` uses the current HIP device; more than one device shards host batches
    /// into contiguous ranges (independent elements, no exchange step).
    pub fn new(devices: &[i32]) -> Result<Self> {
        let mut ctx: *mut ZcCtx = std::ptr::null_mut();
        let ids: Vec<c_int> = devices.iter().map(|&d| d as c_int).collect();
        let ptr = if ids.is_empty() { std::ptr::null() } else { ids.as_ptr() };
        check(unsafe { ffi::zc_ctx_create(ptr, ids.len() as c_int, &mut ctx) })?;
        Ok(HipBackend { ctx })
    }

    /// Number of visible HIP devices (0 without a GPU: there is no CPU fallback).
    pub fn device_count() -> i32 {
        unsafe { ffi::zc_device_count() as i32 }
    }

    pub fn version() -> String {
        unsafe { CStr::from_ptr(ffi::zc_version()) }.to_string_lossy().into_owned()
    }

    /// Launch on the caller's `hipStream_t` (device 0 of the context).
    pub unsafe fn set_stream(&self, hip_stream: *mut c_void) -> Result<()> {
        check(ffi::zc_ctx_set_stream(self.ctx, hip_stream, 1))
    }

    /// Back to the context's own stream.
    pub fn use_own_stream(&self) -> Result<()> {
        check(unsafe { ffi::zc_ctx_set_stream(self.ctx, std::ptr::null_mut(), 0) })
    }

    pub fn synchronize(&self) -> Result<()> {
        check(unsafe { ffi::zc_ctx_synchronize(self.ctx) })
    }

    /// The HIP devices behind the context's slots (what `zc_ctx_create(NULL, 0, ..)` picked, or the list it was given).
    pub fn devices(&self) -> Result<Vec<i32>> {
        let n = unsafe { ffi::zc_ctx_device_count(self.ctx) };
        if n < 0 {
            check(n)?;
        }
        (0..n)
            .map(|slot| {
                let d = unsafe { ffi::zc_ctx_device(self.ctx, slot) };
                if d < 0 {
                    check(d)?;
                }
                Ok(d)
            })
            .collect()
    }

    // -------------------------------------------------------------- call shapes
    fn bin(&self, f: Bin, a: &[u64], b: &[u64], n: usize, wout: usize) -> Result<Vec<u64>> {
        let mut out = vec![0u64; n * wout];
        check(unsafe { f(self.ctx, a.as_ptr(), b.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(out)
    }

    fn un(&self, f: Un, a: &[u64], n: usize, wout: usize) -> Result<Vec<u64>> {
        let mut out = vec![0u64; n * wout];
        check(unsafe { f(self.ctx, a.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(out)
    }

    fn flag(&self, f: Flag, a: &[u64], n: usize) -> Result<Vec<u8>> {
        let mut out = vec![0u8; n];
        check(unsafe { f(self.ctx, a.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(out)
    }

    // -------------------------------------------------------------- FieldElement (backend/u64/field.rs)
    /// `a + b` (`:191-207`).
    pub fn fe_add(&self, a: &[FieldElement], b: &[FieldElement]) -> Result<Vec<FieldElement>> {
        assert_eq!(a.len(), b.len());
        Ok(unflat_fe(&self.bin(ffi::zc_fe_add, &flat_fe(a), &flat_fe(b), a.len(), 5)?))
    }

    /// `a - b` (`:217-240`).
    pub fn fe_sub(&self, a: &[FieldElement], b: &[FieldElement]) -> Result<Vec<FieldElement>> {
        assert_eq!(a.len(), b.len());
        Ok(unflat_fe(&self.bin(ffi::zc_fe_sub, &flat_fe(a), &flat_fe(b), a.len(), 5)?))
    }

    /// `a * b` (`:250-275`).
    pub fn fe_mul(&self, a: &[FieldElement], b: &[FieldElement]) -> Result<Vec<FieldElement>> {
        assert_eq!(a.len(), b.len());
        Ok(unflat_fe(&self.bin(ffi::zc_fe_mul, &flat_fe(a), &flat_fe(b), a.len(), 5)?))
    }

    /// `a.pow(&e)` (`:325-355`).
    pub fn fe_pow(&self, a: &[FieldElement], e: &[FieldElement]) -> Result<Vec<FieldElement>> {
        assert_eq!(a.len(), e.len());
        Ok(unflat_fe(&self.bin(ffi::zc_fe_pow, &flat_fe(a), &flat_fe(e), a.len(), 5)?))
    }

    /// `-a` (`:170-189`).
    pub fn fe_neg(&self, a: &[FieldElement]) -> Result<Vec<FieldElement>> {
        Ok(unflat_fe(&self.un(ffi::zc_fe_neg, &flat_fe(a), a.len(), 5)?))
    }

    /// `a.square()` (`:302-315`).
    pub fn fe_square(&self, a: &[FieldElement]) -> Result<Vec<FieldElement>> {
        Ok(unflat_fe(&self.un(ffi::zc_fe_square, &flat_fe(a), a.len(), 5)?))
    }

    /// `a.half()` (`:317-323`).
    pub fn fe_half(&self, a: &[FieldElement]) -> Result<Vec<FieldElement>> {
        Ok(unflat_fe(&self.un(ffi::zc_fe_half, &flat_fe(a), a.len(), 5)?))
    }

    /// `a.inverse()` (`:854-925`); `None` where the reference panics (a = 0).
    pub fn fe_invert(&self, a: &[FieldElement]) -> Result<Vec<Option<FieldElement>>> {
        let (fa, n) = (flat_fe(a), a.len());
        let (mut out, mut ok) = (vec![0u64; n * 5], vec![0u8; n]);
        check(unsafe { ffi::zc_fe_invert(self.ctx, fa.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        Ok(masked(unflat_fe(&out), &ok))
    }

    /// `a / b` (`:277-300`); `None` where b = 0.
    pub fn fe_div(&self, a: &[FieldElement], b: &[FieldElement]) -> Result<Vec<Option<FieldElement>>> {
        assert_eq!(a.len(), b.len());
        let (fa, fb, n) = (flat_fe(a), flat_fe(b), a.len());
        let (mut out, mut ok) = (vec![0u64; n * 5], vec![0u8; n]);
        check(unsafe { ffi::zc_fe_div(self.ctx, fa.as_ptr(), fb.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        Ok(masked(unflat_fe(&out), &ok))
    }

    /// `a.legendre_symbol()` as 0 / 1 (`:703-706`; 1 also for a = 0, like the reference).
    pub fn fe_legendre_symbol(&self, a: &[FieldElement]) -> Result<Vec<u8>> {
        self.flag(ffi::zc_fe_legendre_symbol, &flat_fe(a), a.len())
    }

    /// `a.is_positive()` (`:552-557`).
    pub fn fe_is_positive(&self, a: &[FieldElement]) -> Result<Vec<bool>> {
        Ok(flags(self.flag(ffi::zc_fe_is_positive, &flat_fe(a), a.len())?))
    }

    /// `a.mod_sqrt(Choice::from(sign))` (`:357-441`).
    pub fn fe_mod_sqrt(&self, a: &[FieldElement], sign: bool) -> Result<Vec<Option<FieldElement>>> {
        let (fa, n) = (flat_fe(a), a.len());
        let (mut out, mut ok) = (vec![0u64; n * 5], vec![0u8; n]);
        check(unsafe {
            ffi::zc_fe_mod_sqrt(self.ctx, fa.as_ptr(), sign as c_int, out.as_mut_ptr(), ok.as_mut_ptr(), n)
        })?;
        Ok(masked(unflat_fe(&out), &ok))
    }

    /// `u.sqrt_ratio_i(&v)` (`:462-503`): `(was_square, root)`.
    pub fn fe_sqrt_ratio_i(&self, u: &[FieldElement], v: &[FieldElement]) -> Result<Vec<(bool, FieldElement)>> {
        assert_eq!(u.len(), v.len());
        let (fu, fv, n) = (flat_fe(u), flat_fe(v), u.len());
        let (mut out, mut sq) = (vec![0u64; n * 5], vec![0u8; n]);
        check(unsafe {
            ffi::zc_fe_sqrt_ratio_i(self.ctx, fu.as_ptr(), fv.as_ptr(), out.as_mut_ptr(), sq.as_mut_ptr(), n)
        })?;
        Ok(sq.into_iter().map(|s| s == 1).zip(unflat_fe(&out)).collect())
    }

    /// `a.inv_sqrt()` (`:443-460`): `(was_square, root)`.
    pub fn fe_inv_sqrt(&self, a: &[FieldElement]) -> Result<Vec<(bool, FieldElement)>> {
        let (fa, n) = (flat_fe(a), a.len());
        let (mut out, mut sq) = (vec![0u64; n * 5], vec![0u8; n]);
        check(unsafe { ffi::zc_fe_inv_sqrt(self.ctx, fa.as_ptr(), out.as_mut_ptr(), sq.as_mut_ptr(), n) })?;
        Ok(sq.into_iter().map(|s| s == 1).zip(unflat_fe(&out)).collect())
    }

    /// `FieldElement::from_bytes` (`:563-587`).
    pub fn fe_from_bytes(&self, bytes: &[[u8; 32]]) -> Result<Vec<FieldElement>> {
        let flat: Vec<u8> = bytes.iter().flat_map(|b| b.iter().copied()).collect();
        let n = bytes.len();
        let mut out = vec![0u64; n * 5];
        check(unsafe { ffi::zc_fe_from_bytes(self.ctx, flat.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(unflat_fe(&out))
    }

    /// `a.to_bytes()` (`:591-631`).
    pub fn fe_to_bytes(&self, a: &[FieldElement]) -> Result<Vec<[u8; 32]>> {
        let (fa, n) = (flat_fe(a), a.len());
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::zc_fe_to_bytes(self.ctx, fa.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(unflat_b32(&out))
    }

    // -------------------------------------------------------------- Scalar (backend/u64/scalar.rs)
    /// `a + b` (`:184-200`).
    pub fn sc_add(&self, a: &[Scalar], b: &[Scalar]) -> Result<Vec<Scalar>> {
        assert_eq!(a.len(), b.len());
        Ok(unflat_sc(&self.bin(ffi::zc_sc_add, &flat_sc(a), &flat_sc(b), a.len(), 5)?))
    }

    /// `a - b` (`:210-237`).
    pub fn sc_sub(&self, a: &[Scalar], b: &[Scalar]) -> Result<Vec<Scalar>> {
        assert_eq!(a.len(), b.len());
        Ok(unflat_sc(&self.bin(ffi::zc_sc_sub, &flat_sc(a), &flat_sc(b), a.len(), 5)?))
    }

    /// `a * b` (`:247-270`).
    pub fn sc_mul(&self, a: &[Scalar], b: &[Scalar]) -> Result<Vec<Scalar>> {
        assert_eq!(a.len(), b.len());
        Ok(unflat_sc(&self.bin(ffi::zc_sc_mul, &flat_sc(a), &flat_sc(b), a.len(), 5)?))
    }

    /// `-a` (`:139-155`).
    pub fn sc_neg(&self, a: &[Scalar]) -> Result<Vec<Scalar>> {
        Ok(unflat_sc(&self.un(ffi::zc_sc_neg, &flat_sc(a), a.len(), 5)?))
    }

    /// `a.square()` (`:272-283`).
    pub fn sc_square(&self, a: &[Scalar]) -> Result<Vec<Scalar>> {
        Ok(unflat_sc(&self.un(ffi::zc_sc_square, &flat_sc(a), a.len(), 5)?))
    }

    /// `a.half()` (`:285-291`).
    pub fn sc_half(&self, a: &[Scalar]) -> Result<Vec<Scalar>> {
        Ok(unflat_sc(&self.un(ffi::zc_sc_half, &flat_sc(a), a.len(), 5)?))
    }

    /// `a.pow(&e)` (`:300-322`).
    pub fn sc_pow(&self, a: &[Scalar], e: &[Scalar]) -> Result<Vec<Scalar>> {
        assert_eq!(a.len(), e.len());
        Ok(unflat_sc(&self.bin(ffi::zc_sc_pow, &flat_sc(a), &flat_sc(e), a.len(), 5)?))
    }

    /// `a >> shift` (`Shr<u8>`, `:165-182`).
    pub fn sc_shr(&self, a: &[Scalar], shift: u8) -> Result<Vec<Scalar>> {
        let (fa, n) = (flat_sc(a), a.len());
        let mut out = vec![0u64; n * 5];
        check(unsafe { ffi::zc_sc_shr(self.ctx, fa.as_ptr(), shift as c_uint, out.as_mut_ptr(), n) })?;
        Ok(unflat_sc(&out))
    }

    /// `a.into_bits()` (`:352-366`).
    pub fn sc_into_bits(&self, a: &[Scalar]) -> Result<Vec<[u8; 256]>> {
        let (fa, n) = (flat_sc(a), a.len());
        let mut out = vec![0u8; n * 256];
        check(unsafe { ffi::zc_sc_into_bits(self.ctx, fa.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(out.chunks_exact(256).map(|c| { let mut b = [0u8; 256]; b.copy_from_slice(c); b }).collect())
    }

    /// `a.compute_NAF()` (`width == 0`, `:370-389`) / `a.compute_window_NAF(width)` (`:396-415`).
    pub fn sc_compute_naf(&self, a: &[Scalar], width: u8) -> Result<Vec<[i8; 256]>> {
        let (fa, n) = (flat_sc(a), a.len());
        let mut out = vec![0i8; n * 256];
        check(unsafe { ffi::zc_sc_compute_naf(self.ctx, fa.as_ptr(), width as c_uint, out.as_mut_ptr(), n) })?;
        Ok(out.chunks_exact(256).map(|c| { let mut d = [0i8; 256]; d.copy_from_slice(c); d }).collect())
    }

    /// `Scalar::from_bytes` (`:445-467`); `None` where the reference asserts (value > L - 1).
    pub fn sc_from_bytes(&self, bytes: &[[u8; 32]]) -> Result<Vec<Option<Scalar>>> {
        let flat: Vec<u8> = bytes.iter().flat_map(|b| b.iter().copied()).collect();
        let n = bytes.len();
        let (mut out, mut ok) = (vec![0u64; n * 5], vec![0u8; n]);
        check(unsafe { ffi::zc_sc_from_bytes(self.ctx, flat.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        Ok(masked(unflat_sc(&out), &ok))
    }

    /// `a.to_bytes()` (`:477-516`).
    pub fn sc_to_bytes(&self, a: &[Scalar]) -> Result<Vec<[u8; 32]>> {
        let (fa, n) = (flat_sc(a), a.len());
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::zc_sc_to_bytes(self.ctx, fa.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(unflat_b32(&out))
    }

    // -------------------------------------------------------------- EdwardsPoint (src/edwards.rs)
    /// `p + q` (`:465-501`), identical limbs.
    pub fn ed_add(&self, p: &[EdwardsPoint], q: &[EdwardsPoint]) -> Result<Vec<EdwardsPoint>> {
        assert_eq!(p.len(), q.len());
        Ok(unflat_ed(&self.bin(ffi::zc_ed_add, &flat_ed(p), &flat_ed(q), p.len(), 20)?))
    }

    /// `p - q` (`:503-545`), identical limbs.
    pub fn ed_sub(&self, p: &[EdwardsPoint], q: &[EdwardsPoint]) -> Result<Vec<EdwardsPoint>> {
        assert_eq!(p.len(), q.len());
        Ok(unflat_ed(&self.bin(ffi::zc_ed_sub, &flat_ed(p), &flat_ed(q), p.len(), 20)?))
    }

    /// `p.double()` (`:579-592`).
    pub fn ed_double(&self, p: &[EdwardsPoint]) -> Result<Vec<EdwardsPoint>> {
        Ok(unflat_ed(&self.un(ffi::zc_ed_double, &flat_ed(p), p.len(), 20)?))
    }

    /// `-p` (`:440-463`).
    pub fn ed_neg(&self, p: &[EdwardsPoint]) -> Result<Vec<EdwardsPoint>> {
        Ok(unflat_ed(&self.un(ffi::zc_ed_neg, &flat_ed(p), p.len(), 20)?))
    }

    /// Variable-base scalar multiplication by the named reference algorithm.
    pub fn ed_mul(&self, p: &[EdwardsPoint], k: &[Scalar], alg: MulAlgorithm) -> Result<Vec<EdwardsPoint>> {
        assert_eq!(p.len(), k.len());
        let (fp, fk, n) = (flat_ed(p), flat_sc(k), p.len());
        let mut out = vec![0u64; n * 20];
        check(unsafe {
            ffi::zc_ed_scalar_mul(self.ctx, fp.as_ptr(), fk.as_ptr(), out.as_mut_ptr(), n, alg.flags())
        })?;
        Ok(unflat_ed(&out))
    }

    /// Batched `&P * &k` (`:547-561`): bit-identical (X:Y:Z:T) limbs.
    pub fn mul_batch(&self, p: &[EdwardsPoint], k: &[Scalar]) -> Result<Vec<EdwardsPoint>> {
        self.ed_mul(p, k, MulAlgorithm::DoubleAndAdd)
    }

    /// `mul_by_pow_2(&p, kexp)` (`:186-191`); `Err` where `Scalar::two_pow_k` asserts (kexp >= 250).
    pub fn ed_mul_by_pow_2(&self, p: &[EdwardsPoint], kexp: u64) -> Result<Vec<EdwardsPoint>> {
        let (fp, n) = (flat_ed(p), p.len());
        let mut out = vec![0u64; n * 20];
        check(unsafe { ffi::zc_ed_mul_by_pow_2(self.ctx, fp.as_ptr(), kexp, out.as_mut_ptr(), n) })?;
        Ok(unflat_ed(&out))
    }

    /// `mul_by_cofactor(&p)` (`:174-179`).
    pub fn ed_mul_by_cofactor(&self, p: &[EdwardsPoint]) -> Result<Vec<EdwardsPoint>> {
        Ok(unflat_ed(&self.un(ffi::zc_ed_mul_by_cofactor, &flat_ed(p), p.len(), 20)?))
    }

    /// `AffinePoint::from(p)` (`:1071-1092`); `None` where Z = 0.
    pub fn ed_to_affine(&self, p: &[EdwardsPoint]) -> Result<Vec<Option<AffinePoint>>> {
        let (fp, n) = (flat_ed(p), p.len());
        let (mut xy, mut ok) = (vec![0u64; n * 10], vec![0u8; n]);
        check(unsafe { ffi::zc_ed_to_affine(self.ctx, fp.as_ptr(), xy.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        let pts: Vec<AffinePoint> = xy
            .chunks_exact(10)
            .map(|c| AffinePoint { X: FieldElement(limbs(&c[0..5])), Y: FieldElement(limbs(&c[5..10])) })
            .collect();
        Ok(masked(pts, &ok))
    }

    /// `p == q` (`:360-370`).
    pub fn ed_eq(&self, p: &[EdwardsPoint], q: &[EdwardsPoint]) -> Result<Vec<bool>> {
        assert_eq!(p.len(), q.len());
        let (fp, fq, n) = (flat_ed(p), flat_ed(q), p.len());
        let mut eq = vec![0u8; n];
        check(unsafe { ffi::zc_ed_eq(self.ctx, fp.as_ptr(), fq.as_ptr(), eq.as_mut_ptr(), n) })?;
        Ok(flags(eq))
    }

    /// `p.compress()` (`:613-629`); `None` where the reference's `unwrap`s panic.
    pub fn ed_compress(&self, p: &[EdwardsPoint]) -> Result<Vec<Option<CompressedEdwardsY>>> {
        let (fp, n) = (flat_ed(p), p.len());
        let (mut out, mut ok) = (vec![0u8; n * 32], vec![0u8; n]);
        check(unsafe { ffi::zc_ed_compress(self.ctx, fp.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        let enc: Vec<CompressedEdwardsY> = out.chunks_exact(32).map(|c| CompressedEdwardsY(bytes32(c))).collect();
        Ok(masked(enc, &ok))
    }

    /// `c.decompress()` (`:313-326`).
    pub fn ed_decompress(&self, c: &[CompressedEdwardsY]) -> Result<Vec<Option<EdwardsPoint>>> {
        let flat: Vec<u8> = c.iter().flat_map(|e| e.0.iter().copied()).collect();
        let n = c.len();
        let (mut out, mut ok) = (vec![0u64; n * 20], vec![0u8; n]);
        check(unsafe { ffi::zc_ed_decompress(self.ctx, flat.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        Ok(masked(unflat_ed(&out), &ok))
    }

    /// `p.is_valid()` (`:393-400`).
    pub fn ed_is_valid(&self, p: &[EdwardsPoint]) -> Result<Vec<bool>> {
        Ok(flags(self.flag(ffi::zc_ed_is_valid, &flat_ed(p), p.len())?))
    }

    /// `k * BASEPOINT` from a fixed-base table: equal to `&BASEPOINT * &k` under `==`
    /// (not limb-identical).
    pub fn ed_mul_base(&self, k: &[Scalar]) -> Result<Vec<EdwardsPoint>> {
        Ok(unflat_ed(&self.un(ffi::zc_ed_mul_base, &flat_sc(k), k.len(), 20)?))
    }

    /// `window_naf_mul` (src/edwards.rs:155-171) with its table indexed correctly, in one launch;
    /// `width` 2..=7.  Equal to `&BASEPOINT * &k` under `==` for canonical scalars.
    pub fn ed_mul_base_wnaf(&self, k: &[Scalar], width: u8) -> Result<Vec<EdwardsPoint>> {
        let fk = flat_sc(k);
        let mut out = vec![0u64; 20 * k.len()];
        check(unsafe { ffi::zc_ed_mul_base_wnaf(self.ctx, fk.as_ptr(), width as u32, out.as_mut_ptr(), k.len()) })?;
        Ok(unflat_ed(&out))
    }

    /// The bucket method's plan for a shard of `n` pairs on this context (a query, no device work):
    /// `[c, W, affine, record payload bytes, run length, buckets per segment, sort passes, window groups, record stride,
    /// windows per group x 4, run length per group x 4]`.
    pub fn msm_plan(&self, n: usize, points_aligned16: bool) -> Result<[i32; 17]> {
        let mut v = [0i32; 17];
        check(unsafe { ffi::zc_msm_plan(self.ctx, n, points_aligned16 as i32, v.as_mut_ptr(), 17) })?;
        Ok(v)
    }

    /// `sum_i k_i * P_i` (bucket method); equal to the fold of `Mul` and `Add` under `==`.
    pub fn msm(&self, p: &[EdwardsPoint], k: &[Scalar]) -> Result<EdwardsPoint> {
        assert_eq!(p.len(), k.len());
        let (fp, fk) = (flat_ed(p), flat_sc(k));
        let mut out = vec![0u64; 20];
        check(unsafe { ffi::zc_msm(self.ctx, fp.as_ptr(), fk.as_ptr(), p.len(), out.as_mut_ptr()) })?;
        Ok(unflat_ed(&out)[0])
    }

    // -------------------------------------------------------------- sharded MSM: the exchange step
    /// This device's partial sum, left in device memory (`out_dev`: 160 bytes of HIP memory).
    pub unsafe fn msm_partial(&self, points: *const u64, scalars: *const u64, n: usize, out_dev: *mut u64) -> Result<()> {
        check(ffi::zc_msm_partial(self.ctx, points, scalars, n, out_dev))
    }

    /// `((p_0 + p_1) + p_2) + ...` in index order, one kernel launch.
    pub fn fold_ordered(&self, parts: &[EdwardsPoint]) -> Result<EdwardsPoint> {
        let fp = flat_ed(parts);
        let mut out = vec![0u64; 20];
        check(unsafe { ffi::zc_ed_fold_ordered(self.ctx, fp.as_ptr(), parts.len(), out.as_mut_ptr()) })?;
        Ok(unflat_ed(&out)[0])
    }

    /// Rank 0: the 128-byte RCCL id every rank passes to `comm_init` (any host transport).
    pub fn comm_unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        check(unsafe { ffi::zc_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    pub fn comm_init(&self, id: &[u8; 128], rank: i32, world: i32) -> Result<()> {
        check(unsafe { ffi::zc_comm_init(self.ctx, id.as_ptr(), rank, world) })
    }

    pub fn comm_destroy(&self) -> Result<()> {
        check(unsafe { ffi::zc_comm_destroy(self.ctx) })
    }

    /// Ranks of the context's RCCL communicator as RCCL reports them (`ncclCommCount`); 0 without one.
    pub fn comm_size(&self) -> Result<i32> {
        let mut ranks = 0i32;
        check(unsafe { ffi::zc_comm_size(self.ctx, &mut ranks) })?;
        Ok(ranks)
    }

    /// This rank's shard of a global MSM: local bucket method, `ncclAllGather` of the 160-byte
    /// partial sums, ordered fold on the device; every rank gets the same point.
    pub fn msm_sharded(&self, p: &[EdwardsPoint], k: &[Scalar]) -> Result<EdwardsPoint> {
        assert_eq!(p.len(), k.len());
        let (fp, fk) = (flat_ed(p), flat_sc(k));
        let mut out = vec![0u64; 20];
        check(unsafe { ffi::zc_msm_sharded(self.ctx, fp.as_ptr(), fk.as_ptr(), p.len(), out.as_mut_ptr()) })?;
        Ok(unflat_ed(&out)[0])
    }

    /// Launch stream of device slot `slot` of a multi-device context.
    pub unsafe fn set_stream_dev(&self, slot: i32, hip_stream: *mut c_void) -> Result<()> {
        check(ffi::zc_ctx_set_stream_dev(self.ctx, slot, hip_stream, 1))
    }

    /// Pin a long-lived host buffer so host batches copy asynchronously.
    pub unsafe fn host_register(ptr: *mut c_void, bytes: usize) -> Result<()> {
        check(ffi::zc_host_register(ptr, bytes))
    }

    pub unsafe fn host_unregister(ptr: *mut c_void) -> Result<()> {
        check(ffi::zc_host_unregister(ptr))
    }

    // -------------------------------------------------------------- ProjectivePoint (src/edwards.rs:666-998)
    /// `p + q` (`:809-865`).
    pub fn proj_add(&self, p: &[ProjectivePoint], q: &[ProjectivePoint]) -> Result<Vec<ProjectivePoint>> {
        assert_eq!(p.len(), q.len());
        Ok(unflat_proj(&self.bin(ffi::zc_proj_add, &flat_proj(p), &flat_proj(q), p.len(), 15)?))
    }

    /// `p.double()` (`:905-942`).
    pub fn proj_double(&self, p: &[ProjectivePoint]) -> Result<Vec<ProjectivePoint>> {
        Ok(unflat_proj(&self.un(ffi::zc_proj_double, &flat_proj(p), p.len(), 15)?))
    }

    /// `EdwardsPoint::from(p)` (`:402-417`).
    pub fn proj_to_extended(&self, p: &[ProjectivePoint]) -> Result<Vec<EdwardsPoint>> {
        Ok(unflat_ed(&self.un(ffi::zc_proj_to_extended, &flat_proj(p), p.len(), 20)?))
    }

    /// `-p` (`:787-807`).
    pub fn proj_neg(&self, p: &[ProjectivePoint]) -> Result<Vec<ProjectivePoint>> {
        Ok(unflat_proj(&self.un(ffi::zc_proj_neg, &flat_proj(p), p.len(), 15)?))
    }

    /// `p - q` (`:851-879`).
    pub fn proj_sub(&self, p: &[ProjectivePoint], q: &[ProjectivePoint]) -> Result<Vec<ProjectivePoint>> {
        assert_eq!(p.len(), q.len());
        Ok(unflat_proj(&self.bin(ffi::zc_proj_sub, &flat_proj(p), &flat_proj(q), p.len(), 15)?))
    }

    /// `p == q` (`:701-711`).
    pub fn proj_eq(&self, p: &[ProjectivePoint], q: &[ProjectivePoint]) -> Result<Vec<bool>> {
        assert_eq!(p.len(), q.len());
        let (fp, fq, n) = (flat_proj(p), flat_proj(q), p.len());
        let mut eq = vec![0u8; n];
        check(unsafe { ffi::zc_proj_eq(self.ctx, fp.as_ptr(), fq.as_ptr(), eq.as_mut_ptr(), n) })?;
        Ok(flags(eq))
    }

    /// `p.is_valid()` (`:733-748`).
    pub fn proj_is_valid(&self, p: &[ProjectivePoint]) -> Result<Vec<bool>> {
        Ok(flags(self.flag(ffi::zc_proj_is_valid, &flat_proj(p), p.len())?))
    }

    /// `&p * &k` (`:881-912`).
    pub fn proj_scalar_mul(&self, p: &[ProjectivePoint], k: &[Scalar]) -> Result<Vec<ProjectivePoint>> {
        assert_eq!(p.len(), k.len());
        Ok(unflat_proj(&self.bin(ffi::zc_proj_scalar_mul, &flat_proj(p), &flat_sc(k), p.len(), 15)?))
    }

    /// `p.coset4()` (`:603-610`).
    pub fn ed_coset4(&self, p: &[EdwardsPoint]) -> Result<Vec<[EdwardsPoint; 4]>> {
        let out = unflat_ed(&self.un(ffi::zc_ed_coset4, &flat_ed(p), p.len(), 80)?);
        Ok(out.chunks_exact(4).map(|c| [c[0], c[1], c[2], c[3]]).collect())
    }

    // -------------------------------------------------------------- Ristretto (src/ristretto.rs)
    /// `p.compress()` (`:398-425`).
    pub fn ris_compress(&self, p: &[RistrettoPoint]) -> Result<Vec<CompressedRistretto>> {
        let (fp, n) = (flat_ris(p), p.len());
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::zc_ris_compress(self.ctx, fp.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(out.chunks_exact(32).map(|c| CompressedRistretto(bytes32(c))).collect())
    }

    /// `c.decompress()` (`:96-154`).
    pub fn ris_decompress(&self, c: &[CompressedRistretto]) -> Result<Vec<Option<RistrettoPoint>>> {
        let flat: Vec<u8> = c.iter().flat_map(|e| e.0.iter().copied()).collect();
        let n = c.len();
        let (mut out, mut ok) = (vec![0u64; n * 20], vec![0u8; n]);
        check(unsafe { ffi::zc_ris_decompress(self.ctx, flat.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n) })?;
        Ok(masked(unflat_ris(&out), &ok))
    }

    /// `p == q` (`:166-176`).
    pub fn ris_eq(&self, p: &[RistrettoPoint], q: &[RistrettoPoint]) -> Result<Vec<bool>> {
        assert_eq!(p.len(), q.len());
        let (fp, fq, n) = (flat_ris(p), flat_ris(q), p.len());
        let mut eq = vec![0u8; n];
        check(unsafe { ffi::zc_ris_eq(self.ctx, fp.as_ptr(), fq.as_ptr(), eq.as_mut_ptr(), n) })?;
        Ok(flags(eq))
    }

    /// Fused `c.decompress()? * k` then `.compress()` (`:96-154`, `:330-392`, `:398-425`).
    pub fn ris_roundtrip_mul(&self, c: &[CompressedRistretto], k: &[Scalar]) -> Result<Vec<Option<CompressedRistretto>>> {
        assert_eq!(c.len(), k.len());
        let flat: Vec<u8> = c.iter().flat_map(|e| e.0.iter().copied()).collect();
        let (fk, n) = (flat_sc(k), c.len());
        let (mut out, mut ok) = (vec![0u8; n * 32], vec![0u8; n]);
        check(unsafe {
            ffi::zc_ris_roundtrip_mul(self.ctx, flat.as_ptr(), fk.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), n)
        })?;
        let enc: Vec<CompressedRistretto> = out.chunks_exact(32).map(|b| CompressedRistretto(bytes32(b))).collect();
        Ok(masked(enc, &ok))
    }

    /// `p.is_valid()` (`:205-222`).
    pub fn ris_is_valid(&self, p: &[RistrettoPoint]) -> Result<Vec<bool>> {
        Ok(flags(self.flag(ffi::zc_ris_is_valid, &flat_ris(p), p.len())?))
    }

    /// `RistrettoPoint::elligator_ristretto_flavor(&r0)` (`:430-471`).
    pub fn ris_elligator(&self, r0: &[FieldElement]) -> Result<Vec<RistrettoPoint>> {
        Ok(unflat_ris(&self.un(ffi::zc_ris_elligator, &flat_fe(r0), r0.len(), 20)?))
    }

    /// `RistrettoPoint::from_uniform_bytes(&bytes)` (`:493-507`).
    pub fn ris_from_uniform_bytes(&self, bytes: &[[u8; 64]]) -> Result<Vec<RistrettoPoint>> {
        let flat: Vec<u8> = bytes.iter().flat_map(|b| b.iter().copied()).collect();
        let n = bytes.len();
        let mut out = vec![0u64; n * 20];
        check(unsafe { ffi::zc_ris_from_uniform_bytes(self.ctx, flat.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(unflat_ris(&out))
    }

    /// `(RISTRETTO_BASEPOINT * k).compress()`: key generation, identical bytes.
    pub fn ris_mul_base_compress(&self, k: &[Scalar]) -> Result<Vec<CompressedRistretto>> {
        let (fk, n) = (flat_sc(k), k.len());
        let mut out = vec![0u8; n * 32];
        check(unsafe { ffi::zc_ris_mul_base_compress(self.ctx, fk.as_ptr(), out.as_mut_ptr(), n) })?;
        Ok(out.chunks_exact(32).map(|c| CompressedRistretto(bytes32(c))).collect())
    }
}

// ------------------------------------------------------------------ batch traits
/// Batch counterpart of `Mul<&Scalar>` (`src/edwards.rs:547-577`, `src/ristretto.rs:330-392`).
pub trait BatchMul: Sized {
    fn mul_batch_hip(points: &[Self], scalars: &[Scalar], be: &HipBackend) -> Result<Vec<Self>>;
}

impl BatchMul for EdwardsPoint {
    fn mul_batch_hip(points: &[Self], scalars: &[Scalar], be: &HipBackend) -> Result<Vec<Self>> {
        be.mul_batch(points, scalars)
    }
}

impl BatchMul for RistrettoPoint {
    fn mul_batch_hip(points: &[Self], scalars: &[Scalar], be: &HipBackend) -> Result<Vec<Self>> {
        let eds: Vec<EdwardsPoint> = points.iter().map(|p| p.0).collect();
        Ok(be.mul_batch(&eds, scalars)?.into_iter().map(RistrettoPoint).collect())
    }
}

/// Batch counterpart of `Add` / `Double` on points (`src/edwards.rs:465-501`, `:579-592`).
pub trait BatchGroup: Sized {
    fn add_batch_hip(a: &[Self], b: &[Self], be: &HipBackend) -> Result<Vec<Self>>;
    fn double_batch_hip(a: &[Self], be: &HipBackend) -> Result<Vec<Self>>;
}

impl BatchGroup for EdwardsPoint {
    fn add_batch_hip(a: &[Self], b: &[Self], be: &HipBackend) -> Result<Vec<Self>> {
        be.ed_add(a, b)
    }
    fn double_batch_hip(a: &[Self], be: &HipBackend) -> Result<Vec<Self>> {
        be.ed_double(a)
    }
}

impl BatchGroup for RistrettoPoint {
    fn add_batch_hip(a: &[Self], b: &[Self], be: &HipBackend) -> Result<Vec<Self>> {
        let (ea, eb): (Vec<EdwardsPoint>, Vec<EdwardsPoint>) =
            (a.iter().map(|p| p.0).collect(), b.iter().map(|p| p.0).collect());
        Ok(be.ed_add(&ea, &eb)?.into_iter().map(RistrettoPoint).collect())
    }
    fn double_batch_hip(a: &[Self], be: &HipBackend) -> Result<Vec<Self>> {
        let ea: Vec<EdwardsPoint> = a.iter().map(|p| p.0).collect();
        Ok(be.ed_double(&ea)?.into_iter().map(RistrettoPoint).collect())
    }
}

/// Batch counterpart of `Mul` on field elements and scalars (`field.rs:250-275`, `scalar.rs:247-270`).
pub trait BatchRingMul: Sized {
    fn mul_batch_hip(a: &[Self], b: &[Self], be: &HipBackend) -> Result<Vec<Self>>;
}

impl BatchRingMul for FieldElement {
    fn mul_batch_hip(a: &[Self], b: &[Self], be: &HipBackend) -> Result<Vec<Self>> {
        be.fe_mul(a, b)
    }
}

impl BatchRingMul for Scalar {
    fn mul_batch_hip(a: &[Self], b: &[Self], be: &HipBackend) -> Result<Vec<Self>> {
        be.sc_mul(a, b)
    }
}

#[cfg(test)]
mod tests {
    //! Run on a box with an MI355X: `ZEROCAF_HIP_LIB_DIR=... cargo test`.  Each test repeats one
    //! of the reference's own checks with the batch path beside the CPU path.
    use super::*;
    use zerocaf::constants::{BASEPOINT, RISTRETTO_BASEPOINT};
    use zerocaf::traits::ops::Double;

    fn scalars() -> Vec<Scalar> {
        (1u8..=16).map(Scalar::from).collect()
    }

    #[test]
    fn scalar_mul_limbs_match_cpu() {
        let be = HipBackend::new(&[]).expect("no MI355X visible");
        let ks = scalars();
        let ps = vec![BASEPOINT; ks.len()];
        let got = be.mul_batch(&ps, &ks).unwrap();
        for (i, k) in ks.iter().enumerate() {
            let want = &BASEPOINT * k;
            assert_eq!(got[i].X.0, want.X.0);
            assert_eq!(got[i].Y.0, want.Y.0);
            assert_eq!(got[i].Z.0, want.Z.0);
            assert_eq!(got[i].T.0, want.T.0);
        }
    }

    #[test]
    fn add_double_and_compress_match_cpu() {
        let be = HipBackend::new(&[]).expect("no MI355X visible");
        let p = &BASEPOINT * &Scalar::from(7u8);
        let q = &BASEPOINT * &Scalar::from(11u8);
        let sum = be.ed_add(&[p], &[q]).unwrap()[0];
        let cpu = &p + &q;
        assert_eq!(sum.X.0, cpu.X.0);
        assert_eq!(sum.T.0, cpu.T.0);
        let dbl = be.ed_double(&[p]).unwrap()[0];
        assert_eq!(dbl.Z.0, p.double().Z.0);
        let enc = be.ed_compress(&[p]).unwrap()[0].expect("valid point");
        assert_eq!(enc.0, p.compress().0);
    }

    #[test]
    fn ristretto_roundtrip_matches_cpu() {
        let be = HipBackend::new(&[]).expect("no MI355X visible");
        let ks = scalars();
        let encs: Vec<CompressedRistretto> = ks.iter().map(|k| (RISTRETTO_BASEPOINT * *k).compress()).collect();
        let got = be.ris_roundtrip_mul(&encs, &ks).unwrap();
        for i in 0..ks.len() {
            let want = (encs[i].decompress().unwrap() * ks[i]).compress();
            assert_eq!(got[i].expect("valid encoding").0, want.0);
        }
    }
}
