// Reference-side binding for libaeonflux_gpu.so.  SOURCE ONLY: this image has no Rust
// toolchain, so this file has not been compiled here.  It is the `mod gpu` a maintainer would add to the aeonflux
// crate (/root/reference/src/) together with `build.rs` emitting `cargo:rustc-link-lib=dylib=aeonflux_gpu`.
//
// It keeps the crate's own types at the surface: `GpuIssuer::verify_batch(&[ProofOfValidCredential])` has, per
// item, exactly the result of `Issuer::verify(&presentation)` (src/issuer.rs:141-147).
#![allow(non_snake_case)]

use core::ffi::c_void;

use crate::amacs::EncryptedAttribute;
use crate::errors::CredentialError;
use crate::issuer::Issuer;
use crate::nizk::presentation::ProofOfValidCredential;

pub const AFX_MAX_ATTRIBUTES: usize = 32;

#[repr(C)]
pub struct AfxShape {
    pub n_attributes: u32,
    pub kinds: [u8; AFX_MAX_ATTRIBUTES],
    pub n_responses: u32,
    pub n_hidden_scalars: u32,
    pub hidden_scalar_indices: [u16; AFX_MAX_ATTRIBUTES],
    pub n_enc_proofs: u32,
    pub enc_indices: [u16; AFX_MAX_ATTRIBUTES],
}

#[repr(C)]
pub struct AfxEncProofSoa {
    pub challenge: *const u8, pub responses: *const u8, pub pk: *const u8, pub E1: *const u8, pub E2: *const u8,
    pub C_y_1: *const u8, pub C_y_2: *const u8, pub C_y_3: *const u8, pub C_y_2p: *const u8,
}

#[repr(C)]
pub struct AfxPresentationSoa {
    pub challenge: *const u8, pub responses: *const u8, pub C_x_0: *const u8, pub C_x_1: *const u8, pub C_V: *const u8,
    pub C_y: *const u8, pub attr_values: *const u8, pub enc: *const AfxEncProofSoa,
}

extern "C" {
    fn afx_ctx_create(out: *mut *mut c_void, device: i32, sysparams: *const u8, sysparams_len: usize,
                      amacs_key: *const u8, amacs_key_len: usize, issuer_params: *const u8) -> i32;
    fn afx_ctx_destroy(ctx: *mut c_void);
    fn afx_verify_presentations(ctx: *mut c_void, shape: *const AfxShape, batch: *const AfxPresentationSoa,
                                count: usize, status: *mut u8) -> i32;
}

/// `Issuer` with its parameters, tables and key resident on one MI355X.
pub struct GpuIssuer { ctx: *mut c_void }

/// Column-major staging of a batch: every field one `[count][32]` array, repeated fields `[k][count][32]`.
struct Columns {
    challenge: Vec<u8>, responses: Vec<u8>, c_x_0: Vec<u8>, c_x_1: Vec<u8>, c_v: Vec<u8>, c_y: Vec<u8>, attr_values: Vec<u8>,
    enc: Vec<[Vec<u8>; 9]>,
}

impl GpuIssuer {
    pub fn new(issuer: &Issuer, device: i32) -> Result<GpuIssuer, CredentialError> {
        let sp = issuer.system_parameters.to_bytes();                 // src/parameters.rs:155-184
        let key = issuer.amacs_key.to_bytes();                        // src/amacs.rs:110-125
        let mut ip = [0u8; 64];                                       // C_W || I (src/issuer.rs:155,163)
        ip[..32].copy_from_slice(issuer.issuer_parameters.C_W.compress().as_bytes());
        ip[32..].copy_from_slice(issuer.issuer_parameters.I.compress().as_bytes());
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { afx_ctx_create(&mut ctx, device, sp.as_ptr(), sp.len(), key.as_ptr(), key.len(), ip.as_ptr()) };
        if rc != 0 { return Err(CredentialError::NoIssuerKey); }
        Ok(GpuIssuer { ctx })
    }

    /// Batch `Issuer::verify`.  All presentations must share one shape (same attribute kinds, hidden indices and
    /// number of proofs of encryption); group mixed traffic by shape first.
    pub fn verify_batch(&self, batch: &[ProofOfValidCredential]) -> Vec<Result<(), CredentialError>> {
        if batch.is_empty() { return Vec::new(); }
        let count = batch.len();
        let (shape, cols) = marshal(batch);
        let enc_soa: Vec<AfxEncProofSoa> = cols.enc.iter().map(|e| AfxEncProofSoa {
            challenge: e[0].as_ptr(), responses: e[1].as_ptr(), pk: e[2].as_ptr(), E1: e[3].as_ptr(), E2: e[4].as_ptr(),
            C_y_1: e[5].as_ptr(), C_y_2: e[6].as_ptr(), C_y_3: e[7].as_ptr(), C_y_2p: e[8].as_ptr() }).collect();
        let soa = AfxPresentationSoa {
            challenge: cols.challenge.as_ptr(), responses: cols.responses.as_ptr(), C_x_0: cols.c_x_0.as_ptr(),
            C_x_1: cols.c_x_1.as_ptr(), C_V: cols.c_v.as_ptr(), C_y: cols.c_y.as_ptr(), attr_values: cols.attr_values.as_ptr(),
            enc: enc_soa.as_ptr() };
        let mut status = vec![0u8; count];
        let rc = unsafe { afx_verify_presentations(self.ctx, &shape, &soa, count, status.as_mut_ptr()) };
        assert!(rc == 0, "aeonflux_gpu: engine error {}", rc);
        status.iter().map(|s| if *s == 0 { Ok(()) } else { Err(CredentialError::VerificationFailure) }).collect()
    }
}

impl Drop for GpuIssuer {
    fn drop(&mut self) { unsafe { afx_ctx_destroy(self.ctx) } }   // wipes every key copy (src/amacs.rs:64-82)
}

/// ProofOfValidCredential (src/nizk/presentation.rs:118-127) -> shape + columns.  Lives inside the crate because the
/// struct's fields are private.
fn marshal(batch: &[ProofOfValidCredential]) -> (AfxShape, Columns) {
    let count = batch.len();
    let p0 = &batch[0];
    let n = p0.encrypted_attributes.len();
    let nr = p0.proof.responses.len();
    let ne = p0.proofs_of_encryption.len();
    let mut shape = AfxShape { n_attributes: n as u32, kinds: [0; 32], n_responses: nr as u32,
        n_hidden_scalars: p0.hidden_scalar_indices.len() as u32, hidden_scalar_indices: [0; 32],
        n_enc_proofs: ne as u32, enc_indices: [0; 32] };
    for (i, a) in p0.encrypted_attributes.iter().enumerate() {
        shape.kinds[i] = match a { EncryptedAttribute::PublicScalar(_) => 0, EncryptedAttribute::SecretScalar => 1,
                                   EncryptedAttribute::PublicPoint(_) => 2, EncryptedAttribute::SecretPoint => 3 };
    }
    for (i, h) in p0.hidden_scalar_indices.iter().enumerate() { shape.hidden_scalar_indices[i] = *h; }
    for (i, (_, e)) in p0.proofs_of_encryption.iter().enumerate() { shape.enc_indices[i] = e.index; }
    let col = |k: usize| vec![0u8; 32 * k * count];
    let mut c = Columns { challenge: col(1), responses: col(nr), c_x_0: col(1), c_x_1: col(1), c_v: col(1), c_y: col(n),
                          attr_values: col(n), enc: (0..ne).map(|_| [col(1), col(6), col(1), col(1), col(1), col(1), col(1), col(1), col(1)]).collect() };
    let put = |dst: &mut Vec<u8>, row: usize, item: usize, src: &[u8; 32]| dst[32 * (row * count + item)..32 * (row * count + item) + 32].copy_from_slice(src);
    for (i, p) in batch.iter().enumerate() {
        put(&mut c.challenge, 0, i, p.proof.challenge.as_bytes());
        for (k, r) in p.proof.responses.iter().enumerate() { put(&mut c.responses, k, i, r.as_bytes()); }
        put(&mut c.c_x_0, 0, i, p.C_x_0.compress().as_bytes());
        put(&mut c.c_x_1, 0, i, p.C_x_1.compress().as_bytes());
        put(&mut c.c_v, 0, i, p.C_V.compress().as_bytes());
        for (k, y) in p.C_y.iter().enumerate() { put(&mut c.c_y, k, i, y.compress().as_bytes()); }
        for (k, a) in p.encrypted_attributes.iter().enumerate() {
            match a {
                EncryptedAttribute::PublicScalar(m) => put(&mut c.attr_values, k, i, m.as_bytes()),
                EncryptedAttribute::PublicPoint(M) => put(&mut c.attr_values, k, i, M.compress().as_bytes()),
                _ => {}
            }
        }
        for (e, (_, q)) in p.proofs_of_encryption.iter().enumerate() {
            put(&mut c.enc[e][0], 0, i, q.proof.challenge.as_bytes());
            for (k, r) in q.proof.responses.iter().enumerate() { put(&mut c.enc[e][1], k, i, r.as_bytes()); }
            put(&mut c.enc[e][2], 0, i, q.public_key.pk.compress().as_bytes());
            put(&mut c.enc[e][3], 0, i, q.ciphertext.E1.compress().as_bytes());
            put(&mut c.enc[e][4], 0, i, q.ciphertext.E2.compress().as_bytes());
            put(&mut c.enc[e][5], 0, i, q.C_y_1.compress().as_bytes());
            put(&mut c.enc[e][6], 0, i, q.C_y_2.compress().as_bytes());
            put(&mut c.enc[e][7], 0, i, q.C_y_3.compress().as_bytes());
            put(&mut c.enc[e][8], 0, i, q.C_y_2_prime.compress().as_bytes());
        }
    }
    (shape, c)
}
