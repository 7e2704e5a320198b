// Standalone form of the shim in INTEGRATION.md section 2 (inside the zerocaf crate the
// `use zerocaf::` paths become `use crate::`).  Not compiled in the build image (no Rust toolchain).
#![allow(non_snake_case)]
//! Batched MI355X backend: binds libzerocaf_hip.so (include/zerocaf_hip.h).
use zerocaf::edwards::EdwardsPoint;
use zerocaf::field::FieldElement;
use zerocaf::ristretto::{CompressedRistretto, RistrettoPoint};
use zerocaf::scalar::Scalar;
use std::os::raw::{c_int, c_uint, c_void};

#[repr(C)] pub struct ZcCtx { _private: [u8; 0] }

#[link(name = "zerocaf_hip")]
extern "C" {
    fn zc_ctx_create(devices: *const c_int, ndev: c_int, out: *mut *mut ZcCtx) -> c_int;
    fn zc_ctx_destroy(ctx: *mut ZcCtx) -> c_int;
    fn zc_ctx_set_stream(ctx: *mut ZcCtx, hip_stream: *mut c_void, external: c_int) -> c_int;
    fn zc_ctx_synchronize(ctx: *mut ZcCtx) -> c_int;
    fn zc_fe_mul(ctx: *mut ZcCtx, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    fn zc_fe_square(ctx: *mut ZcCtx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    fn zc_fe_invert(ctx: *mut ZcCtx, a: *const u64, out: *mut u64, ok: *mut u8, n: usize) -> c_int;
    fn zc_sc_mul(ctx: *mut ZcCtx, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    fn zc_ed_add(ctx: *mut ZcCtx, p: *const u64, q: *const u64, out: *mut u64, n: usize) -> c_int;
    fn zc_ed_double(ctx: *mut ZcCtx, p: *const u64, out: *mut u64, n: usize) -> c_int;
    fn zc_ed_scalar_mul(ctx: *mut ZcCtx, p: *const u64, k: *const u64, out: *mut u64, n: usize, flags: c_uint) -> c_int;
    fn zc_ed_mul_by_pow_2(ctx: *mut ZcCtx, p: *const u64, kexp: u64, out: *mut u64, n: usize) -> c_int;
    fn zc_ed_compress(ctx: *mut ZcCtx, p: *const u64, out32: *mut u8, ok: *mut u8, n: usize) -> c_int;
    fn zc_ed_decompress(ctx: *mut ZcCtx, in32: *const u8, out: *mut u64, ok: *mut u8, n: usize) -> c_int;
    fn zc_ris_compress(ctx: *mut ZcCtx, p: *const u64, out32: *mut u8, n: usize) -> c_int;
    fn zc_ris_decompress(ctx: *mut ZcCtx, in32: *const u8, out: *mut u64, ok: *mut u8, n: usize) -> c_int;
    fn zc_ris_roundtrip_mul(ctx: *mut ZcCtx, in32: *const u8, k: *const u64, out32: *mut u8, ok: *mut u8, n: usize) -> c_int;
    fn zc_msm(ctx: *mut ZcCtx, points: *const u64, scalars: *const u64, n: usize, out_point: *mut u64) -> c_int;
    // ... the remaining entry points of zerocaf_hip.h follow the same pattern
}

pub struct HipBackend { ctx: *mut ZcCtx }
unsafe impl Send for HipBackend {}

fn flat_points(ps: &[EdwardsPoint]) -> Vec<u64> {
    let mut v = Vec::with_capacity(ps.len() * 20);
    for p in ps { v.extend_from_slice(&p.X.0); v.extend_from_slice(&p.Y.0);
                  v.extend_from_slice(&p.Z.0); v.extend_from_slice(&p.T.0); }
    v
}
fn flat_scalars(ks: &[Scalar]) -> Vec<u64> { ks.iter().flat_map(|k| k.0.iter().copied()).collect() }
fn unflat_points(v: &[u64]) -> Vec<EdwardsPoint> {
    v.chunks_exact(20).map(|c| { let f = |i: usize| { let mut l = [0u64; 5]; l.copy_from_slice(&c[5*i..5*i+5]); FieldElement(l) };
        EdwardsPoint { X: f(0), Y: f(1), Z: f(2), T: f(3) } }).collect()
}

impl HipBackend {
    /// `devices = &[]` uses the current HIP device; more than one device shards host batches.
    pub fn new(devices: &[i32]) -> Result<Self, i32> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { zc_ctx_create(if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() },
                                        devices.len() as c_int, &mut ctx) };
        if rc == 0 { Ok(HipBackend { ctx }) } else { Err(rc) }
    }
    /// Batched `&P * &k` (src/edwards.rs:547-561): bit-identical (X:Y:Z:T) limbs.
    pub fn mul_batch(&self, points: &[EdwardsPoint], scalars: &[Scalar]) -> Vec<EdwardsPoint> {
        assert_eq!(points.len(), scalars.len());
        let (p, k) = (flat_points(points), flat_scalars(scalars));
        let mut out = vec![0u64; p.len()];
        let rc = unsafe { zc_ed_scalar_mul(self.ctx, p.as_ptr(), k.as_ptr(), out.as_mut_ptr(), points.len(), 0) };
        assert_eq!(rc, 0, "zc_ed_scalar_mul failed");
        unflat_points(&out)
    }
    /// Batched decompress -> `* k` -> compress (src/ristretto.rs:96-154, :330-392, :398-425).
    pub fn ristretto_roundtrip_mul(&self, enc: &[CompressedRistretto], ks: &[Scalar]) -> Vec<Option<CompressedRistretto>> {
        let inb: Vec<u8> = enc.iter().flat_map(|e| e.0.iter().copied()).collect();
        let k = flat_scalars(ks);
        let (mut out, mut ok) = (vec![0u8; inb.len()], vec![0u8; enc.len()]);
        let rc = unsafe { zc_ris_roundtrip_mul(self.ctx, inb.as_ptr(), k.as_ptr(), out.as_mut_ptr(), ok.as_mut_ptr(), enc.len()) };
        assert_eq!(rc, 0);
        out.chunks_exact(32).zip(ok).map(|(b, o)| if o == 1 { let mut a = [0u8; 32]; a.copy_from_slice(b); Some(CompressedRistretto(a)) } else { None }).collect()
    }
}
impl Drop for HipBackend { fn drop(&mut self) { unsafe { zc_ctx_destroy(self.ctx); } } }

/// Batch counterpart of the operator traits in src/traits.rs.
pub trait BatchMul { fn mul_batch_hip(points: &[Self], scalars: &[Scalar], be: &HipBackend) -> Vec<Self> where Self: Sized; }
impl BatchMul for RistrettoPoint {
    fn mul_batch_hip(points: &[Self], scalars: &[Scalar], be: &HipBackend) -> Vec<Self> {
        let eds: Vec<EdwardsPoint> = points.iter().map(|p| p.0).collect();
        be.mul_batch(&eds, scalars).into_iter().map(RistrettoPoint).collect()
    }
}
