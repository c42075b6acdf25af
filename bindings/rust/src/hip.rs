//! `src/hip.rs` for the reference crate: raw declarations of include/plonk_hip.h and a small safe wrapper.
//!
//! Written against arkworks 0.5 (`ark-ff`, `ark-ec`, `ark-bn254`, `ark-poly-commit`) as pinned by the reference's
//! Cargo.toml:9-18.  NOT COMPILED here (no Rust toolchain in this repository's build environment); the C++ host layer
//! `plonk_amd/host/` is the tested equivalent and follows the same call sequence.
//!
//! Data conventions of the C ABI:
//!  * a field element is 4 little-endian u64 limbs in Montgomery form, which is exactly `Fp::0.0` of ark-ff's
//!    `Fp256<MontBackend<_, 4>>`; elements are copied limb by limb (no reliance on struct layout);
//!  * a G1 point is x || y (8 limbs) plus an `is_inf` flag; `ark_ec::short_weierstrass::Affine` is never passed by pointer.
#![allow(non_camel_case_types, dead_code)]

use ark_bn254::{Fq, Fr, G1Affine};
use ark_ff::BigInt;
use std::ffi::CStr;
use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct plk_ctx {
    _private: [u8; 0],
}

pub const PLK_OK: i32 = 0;
pub const PLK_ERR_SRS_TOO_SHORT: i32 = 4;
pub const PLK_ERR_NOT_DIVISIBLE: i32 = 5;
pub const PLK_ERR_INVERSE_OF_ZERO: i32 = 6;
pub const PLK_ERR_INDEX_OUT_OF_RANGE: i32 = 7;

extern "C" {
    pub fn plk_create(device: i32, out: *mut *mut plk_ctx) -> i32;
    pub fn plk_destroy(ctx: *mut plk_ctx) -> i32;
    pub fn plk_last_error(ctx: *mut plk_ctx) -> *const c_char;
    pub fn plk_malloc(ctx: *mut plk_ctx, bytes: u64, dev: *mut *mut c_void) -> i32;
    pub fn plk_free(ctx: *mut plk_ctx, dev: *mut c_void) -> i32;
    pub fn plk_upload(ctx: *mut plk_ctx, dev_dst: *mut c_void, host_src: *const c_void, bytes: u64) -> i32;
    pub fn plk_download(ctx: *mut plk_ctx, host_dst: *mut c_void, dev_src: *const c_void, bytes: u64) -> i32;
    pub fn plk_copy_dev(ctx: *mut plk_ctx, dev_dst: *mut c_void, dev_src: *const c_void, bytes: u64, total_bytes: u64) -> i32;
    pub fn plk_sync(ctx: *mut plk_ctx) -> i32;
    // PCS::trim output -> HBM (src/main.rs:183)
    pub fn plk_srs_load(ctx: *mut plk_ctx, g_xy: *const u64, len: u64, gamma_g_xy: *const u64) -> i32;
    pub fn plk_srs_setup(ctx: *mut plk_ctx, beta: *const u64, g_xy: *const u64, gamma_g_xy: *const u64, len: u64) -> i32;
    pub fn plk_srs_load_shard(ctx: *mut plk_ctx, g_xy: *const u64, shard_first: u64, shard_len: u64, total_len: u64, gamma_g_xy: *const u64) -> i32;
    // domain.fft / domain.ifft (src/prover.rs:232,233,237; src/circuit.rs:404,438,454,543; src/poly_utils.rs:9,27,28)
    pub fn plk_ntt(ctx: *mut plk_ctx, data: *mut u64, log_n: u32, inverse: i32, coset_gen: *const u64) -> i32;
    pub fn plk_ntt_dev(ctx: *mut plk_ctx, data: *mut c_void, log_n: u32, inverse: i32, coset_gen: *const u64) -> i32;
    pub fn plk_ntt_dev_enqueue(ctx: *mut plk_ctx, data: *mut c_void, log_n: u32, inverse: i32, coset_gen: *const u64) -> i32;
    // interpolate_selector / interpolate_permutation_polynomials (src/poly_utils.rs:7-31)
    pub fn plk_interpolate_selector(ctx: *mut plk_ctx, values: *const u64, len: u64, coeffs_out: *mut u64, log_n_out: *mut u32) -> i32;
    pub fn plk_interpolate_permutation(ctx: *mut plk_ctx, sigma: *const u64, sigma_len: u64, log_n4: u32, s_id_out: *mut u64, s_sigma_out: *mut u64) -> i32;
    // Circuit::build_grand_product (src/circuit.rs:491-544)
    pub fn plk_grand_product(ctx: *mut plk_ctx, witness_flat: *const u64, sigma: *const u64, log_n: u32, beta: *const u64, gamma: *const u64, z_coeffs_out: *mut u64) -> i32;
    pub fn plk_grand_product_dev(ctx: *mut plk_ctx, witness_flat: *const c_void, sigma: *const c_void, log_n: u32, beta: *const u64, gamma: *const u64, z_coeffs_out: *mut c_void) -> i32;
    // Circuit::build_quotient_polynomial (src/circuit.rs:442-489)
    pub fn plk_quotient(ctx: *mut plk_ctx, a: *const u64, b: *const u64, c: *const u64, q_add_vals: *const u64, q_mul_vals: *const u64, z_vals: *const u64,
                        sigma: *const u64, log_n: u32, beta: *const u64, gamma: *const u64, t_out: *mut u64, t_len: *mut u64) -> i32;
    pub fn plk_quotient_dev(ctx: *mut plk_ctx, a: *const c_void, b: *const c_void, c: *const c_void, q_add_vals: *const c_void, q_mul_vals: *const c_void,
                            z_vals: *const c_void, sigma: *const c_void, log_n: u32, beta: *const u64, gamma: *const u64, t_out: *mut c_void, t_len: *mut u64) -> i32;
    // DensePolynomial::evaluate (src/prover.rs:266-274)
    pub fn plk_poly_eval(ctx: *mut plk_ctx, coeffs: *const u64, len: u64, x: *const u64, out: *mut u64) -> i32;
    // PCS::commit (src/prover.rs:150-156,210,253) and PCS::open (src/prover.rs:295-303); tickets let several run at once
    pub fn plk_kzg_commit(ctx: *mut plk_ctx, coeffs: *const u64, len: u64, blind: *const u64, out_xy: *mut u64, out_is_inf: *mut i32) -> i32;
    pub fn plk_kzg_open(ctx: *mut plk_ctx, coeffs: *const u64, len: u64, blind: *const u64, zeta: *const u64, challenge: *const u64,
                        w_xy: *mut u64, w_is_inf: *mut i32, random_v: *mut u64, eval: *mut u64) -> i32;
    pub fn plk_kzg_commit_begin(ctx: *mut plk_ctx, coeffs_dev: *const c_void, len: u64, blind: *const u64, ticket: *mut i32) -> i32;
    pub fn plk_kzg_commit_end(ctx: *mut plk_ctx, ticket: i32, out_xy: *mut u64, out_is_inf: *mut i32) -> i32;
    pub fn plk_kzg_open_begin(ctx: *mut plk_ctx, coeffs_dev: *const c_void, len: u64, blind: *const u64, zeta: *const u64, challenge: *const u64, ticket: *mut i32) -> i32;
    pub fn plk_kzg_submit(ctx: *mut plk_ctx) -> i32;
    pub fn plk_kzg_cancel(ctx: *mut plk_ctx) -> i32;
    pub fn plk_kzg_ready(ctx: *mut plk_ctx, ticket: i32, ready: *mut i32) -> i32;
    pub fn plk_kzg_open_end(ctx: *mut plk_ctx, ticket: i32, w_xy: *mut u64, w_is_inf: *mut i32, random_v: *mut u64, eval: *mut u64) -> i32;
}

/// Limbs of a slice of field elements, Montgomery form, as the C ABI wants them.
pub fn fr_limbs(v: &[Fr]) -> Vec<u64> {
    let mut out = Vec::with_capacity(4 * v.len());
    for x in v {
        out.extend_from_slice(&(x.0).0);
    }
    out
}
pub fn fr_from_limbs(l: &[u64]) -> Fr {
    Fr::new_unchecked(BigInt([l[0], l[1], l[2], l[3]])) // new_unchecked takes the Montgomery representation as is
}
pub fn g1_limbs(points: &[G1Affine]) -> Vec<u64> {
    let mut out = Vec::with_capacity(8 * points.len());
    for p in points {
        assert!(!p.infinity, "SRS powers are never the identity");
        out.extend_from_slice(&(p.x.0).0);
        out.extend_from_slice(&(p.y.0).0);
    }
    out
}
pub fn g1_from_limbs(xy: &[u64; 8], is_inf: i32) -> G1Affine {
    if is_inf != 0 {
        return G1Affine::identity();
    }
    let x = Fq::new_unchecked(BigInt([xy[0], xy[1], xy[2], xy[3]]));
    let y = Fq::new_unchecked(BigInt([xy[4], xy[5], xy[6], xy[7]]));
    G1Affine::new_unchecked(x, y)
}

/// One `plk_ctx`.  The reference is single threaded, so the context is neither `Send` nor `Sync`.
pub struct Backend {
    ctx: *mut plk_ctx,
}

impl Backend {
    pub fn new(device: i32) -> Self {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { plk_create(device, &mut ctx) };
        assert!(rc == PLK_OK, "plk_create failed with status {rc} (is an MI355X visible?)");
        Backend { ctx }
    }
    pub fn raw(&self) -> *mut plk_ctx {
        self.ctx
    }
    /// The reference has no `Result`s on this path: every failure is a panic (`unwrap`, `assert!`, `expect`), and so it is here,
    /// with the reference's own message where it has one (src/circuit.rs:487).
    pub fn check(&self, rc: i32) {
        if rc == PLK_OK {
            return;
        }
        let detail = unsafe { CStr::from_ptr(plk_last_error(self.ctx)) }.to_string_lossy().into_owned();
        match rc {
            PLK_ERR_NOT_DIVISIBLE => panic!("t(X) not divisible by Z_H(X)"),
            PLK_ERR_INVERSE_OF_ZERO => panic!("called `Option::unwrap()` on a `None` value ({detail})"),
            PLK_ERR_INDEX_OUT_OF_RANGE => panic!("index out of bounds ({detail})"),
            PLK_ERR_SRS_TOO_SHORT => panic!("called `Result::unwrap()` on an `Err` value: TooManyCoefficients ({detail})"),
            _ => panic!("libplonk_hip status {rc}: {detail}"),
        }
    }
    /// `KZGProverKey` (marlin_pc::CommitterKey: `powers`, `powers_of_gamma_g`) -> HBM, once per key.
    pub fn load_srs(&self, powers: &[G1Affine], powers_of_gamma_g: &[G1Affine]) {
        assert!(powers_of_gamma_g.len() >= 3, "hiding bound 1 needs three gamma powers");
        let g = g1_limbs(powers);
        let gg = g1_limbs(&powers_of_gamma_g[..3]);
        self.check(unsafe { plk_srs_load(self.ctx, g.as_ptr(), powers.len() as u64, gg.as_ptr()) });
    }
    pub fn ntt(&self, data: &mut [Fr], inverse: bool) {
        assert!(data.len().is_power_of_two());
        let mut l = fr_limbs(data);
        self.check(unsafe { plk_ntt(self.ctx, l.as_mut_ptr(), data.len().trailing_zeros(), inverse as i32, std::ptr::null()) });
        for (x, c) in data.iter_mut().zip(l.chunks_exact(4)) {
            *x = fr_from_limbs(c);
        }
    }
    pub fn commit(&self, coeffs: &[Fr], blind: &[Fr; 3]) -> G1Affine {
        let (c, b) = (fr_limbs(coeffs), fr_limbs(blind));
        let (mut xy, mut inf) = ([0u64; 8], 0i32);
        self.check(unsafe { plk_kzg_commit(self.ctx, c.as_ptr(), coeffs.len() as u64, b.as_ptr(), xy.as_mut_ptr(), &mut inf) });
        g1_from_limbs(&xy, inf)
    }
    /// Returns (w, random_v, f(zeta)).
    pub fn open(&self, coeffs: &[Fr], blind: &[Fr; 3], zeta: &Fr, challenge: &Fr) -> (G1Affine, Fr, Fr) {
        let (c, b) = (fr_limbs(coeffs), fr_limbs(blind));
        let (z, ch) = (fr_limbs(&[*zeta]), fr_limbs(&[*challenge]));
        let (mut xy, mut inf, mut rv, mut ev) = ([0u64; 8], 0i32, [0u64; 4], [0u64; 4]);
        self.check(unsafe {
            plk_kzg_open(self.ctx, c.as_ptr(), coeffs.len() as u64, b.as_ptr(), z.as_ptr(), ch.as_ptr(), xy.as_mut_ptr(), &mut inf, rv.as_mut_ptr(), ev.as_mut_ptr())
        });
        (g1_from_limbs(&xy, inf), fr_from_limbs(&rv), fr_from_limbs(&ev))
    }
    pub fn grand_product(&self, witness_flat: &[Fr], sigma: &[usize], n: usize, beta: &Fr, gamma: &Fr) -> Vec<Fr> {
        assert!(n.is_power_of_two() && witness_flat.len() >= 3 * n && sigma.len() >= 3 * n);
        let w = fr_limbs(&witness_flat[..3 * n]);
        let s: Vec<u64> = sigma[..3 * n].iter().map(|&i| i as u64).collect();
        let mut out = vec![0u64; 4 * n];
        self.check(unsafe {
            plk_grand_product(self.ctx, w.as_ptr(), s.as_ptr(), n.trailing_zeros(), fr_limbs(&[*beta]).as_ptr(), fr_limbs(&[*gamma]).as_ptr(), out.as_mut_ptr())
        });
        out.chunks_exact(4).map(fr_from_limbs).collect()
    }
    /// Coefficients of t (empty for the zero polynomial).  Panics like src/circuit.rs:487 when the remainder is not zero.
    #[allow(clippy::too_many_arguments)]
    pub fn quotient(&self, a: &[Fr], b: &[Fr], c: &[Fr], q_add_vals: &[Fr], q_mul_vals: &[Fr], z_vals: &[Fr], sigma: &[usize], n: usize, beta: &Fr, gamma: &Fr) -> Vec<Fr> {
        let pad = |v: &[Fr]| {
            // the reference indexes past the end of a trimmed column and panics (src/circuit.rs:356-358); keep that behaviour
            assert!(v.len() >= n, "index out of bounds: the len is {} but the index is {}", v.len(), n - 1);
            fr_limbs(&v[..n])
        };
        let s: Vec<u64> = sigma[..3 * n].iter().map(|&i| i as u64).collect();
        let mut out = vec![0u64; 4 * n];
        let mut t_len = 0u64;
        self.check(unsafe {
            plk_quotient(self.ctx, pad(a).as_ptr(), pad(b).as_ptr(), pad(c).as_ptr(), pad(q_add_vals).as_ptr(), pad(q_mul_vals).as_ptr(), pad(z_vals).as_ptr(),
                         s.as_ptr(), n.trailing_zeros(), fr_limbs(&[*beta]).as_ptr(), fr_limbs(&[*gamma]).as_ptr(), out.as_mut_ptr(), &mut t_len)
        });
        out[..4 * t_len as usize].chunks_exact(4).map(fr_from_limbs).collect()
    }
}

impl Drop for Backend {
    fn drop(&mut self) {
        unsafe { plk_destroy(self.ctx) };
    }
}
