// Reference-side binding for libaeonflux_gpu.so.  SOURCE ONLY: this image has no Rust
// toolchain, so this file has not been compiled here.  It is the `mod gpu` a maintainer would add to the aeonflux
// crate (/root/reference/src/) together with `build.rs` emitting `cargo:rustc-link-lib=dylib=aeonflux_gpu`.
//
// It keeps the crate's own types at the surface, one batch method per call site of the drop-in boundary:
//   GpuIssuer::verify_batch(&[ProofOfValidCredential])      per item = Issuer::verify            (src/issuer.rs:141-147)
//   GpuIssuer::issue_batch(Vec<CredentialRequest>, csprng)   per item = Issuer::issue             (src/issuer.rs:111-124)
//   GpuUser::show_batch(&[AnonymousCredential], ...)         per item = AnonymousCredential::show (src/credential.rs:37-46)
//   GpuUser::verify_issuance_batch(Vec<CredentialIssuance>)  per item = CredentialIssuance::verify (src/issuer.rs:48-57)
// and GpuIssuer::new_multi(issuer, &[devices]) puts the same issuer on several GPUs (afx_group_*).
//
// Randomness.  The engine takes every random draw as an input array.  The shim draws, from the CALLER's csprng only:
//   issue : per request, in the reference's order: 64 bytes for `Scalar::random` (t, src/amacs.rs:289), then 64 bytes for
//           `RistrettoPoint::random` (U, src/amacs.rs:290); after all of those, 32 bytes per request that stand in for the draw
//           zkp's `prove_compact` makes from thread_rng() through merlin's `TranscriptRngBuilder::finalize` [3P].
//   show  : per credential 64 bytes for `Scalar::random` (z, src/nizk/presentation.rs:162); after all of those, 32 bytes per
//           credential for the presentation proof's own `prove_compact` (presentation.rs:284), then 32 bytes per
//           ProofOfEncryption, one per SecretPoint attribute in attribute order (presentation.rs:293-309 ->
//           src/nizk/encryption.rs:141).
// So a caller's deterministic csprng yields the reference's t, U and z, and the proofs' synthetic nonces come from the same
// generator instead of the reference's hidden thread_rng() - `rand` is only a dev-dependency of the crate (Cargo.toml:42-45) and
// the shim must not need it.  Every buffer that held such bytes (and the user's symmetric keys) is zeroized on drop, as the
// crate does for what they become (src/amacs.rs:64-82, src/symmetric.rs:51-64).
// The #[repr(C)] structs below mirror include/aeonflux_gpu.h field for field; tests/test_integration_layouts.py checks
// names, order and widths against the header without a Rust compiler.
#![allow(non_snake_case)]

use core::ffi::c_void;

use curve25519_dalek::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek::scalar::Scalar;
use rand_core::{CryptoRng, RngCore};
use zeroize::Zeroize;
use zkp::CompactProof;

use crate::amacs::{Amac, Attribute, EncryptedAttribute};
use crate::credential::AnonymousCredential;
use crate::errors::CredentialError;
use crate::issuer::{CredentialIssuance, Issuer};
use crate::nizk::encryption::ProofOfEncryption;
use crate::nizk::issuance::ProofOfIssuance;
use crate::nizk::presentation::ProofOfValidCredential;
use crate::parameters::{IssuerParameters, SystemParameters};
use crate::symmetric::{Ciphertext, Keypair as SymmetricKeypair, PublicKey as SymmetricPublicKey};
use crate::user::CredentialRequest;

pub const AFX_MAX_ATTRIBUTES: usize = 32;

#[derive(Clone, Copy)]
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

#[repr(C)]
pub struct AfxAttributesSoa {
    pub n_attributes: u32,
    pub kinds: [u8; AFX_MAX_ATTRIBUTES],
    pub values: *const u8,
}

#[repr(C)]
pub struct AfxIssueRandomness { pub t_wide: *const u8, pub U_wide: *const u8, pub rng_seed: *const u8 }

#[repr(C)]
pub struct AfxIssuanceSoa { pub t: *mut u8, pub U: *mut u8, pub V: *mut u8, pub challenge: *mut u8, pub responses: *mut u8 }

#[repr(C)]
pub struct AfxCredentialsSoa {
    pub n_attributes: u32,
    pub kinds: [u8; AFX_MAX_ATTRIBUTES],
    pub values: *const u8, pub M2: *const u8, pub m3: *const u8, pub t: *const u8, pub U: *const u8, pub V: *const u8,
}

#[repr(C)]
pub struct AfxKeypairsSoa { pub a: *const u8, pub a0: *const u8, pub a1: *const u8, pub pk: *const u8 }

#[repr(C)]
pub struct AfxShowRandomness { pub z_wide: *const u8, pub rng_seed: *const u8, pub enc_seeds: *const u8 }

#[repr(C)]
pub struct AfxEncProofOut {
    pub challenge: *mut u8, pub responses: *mut u8, pub pk: *mut u8, pub E1: *mut u8, pub E2: *mut u8,
    pub C_y_1: *mut u8, pub C_y_2: *mut u8, pub C_y_3: *mut u8, pub C_y_2p: *mut u8,
}

#[repr(C)]
pub struct AfxPresentationOut {
    pub challenge: *mut u8, pub responses: *mut u8, pub C_x_0: *mut u8, pub C_x_1: *mut u8, pub C_V: *mut u8,
    pub C_y: *mut u8, pub attr_values: *mut u8, pub enc: *const AfxEncProofOut,
}

#[repr(C)]
pub struct AfxPresentationGroup {
    pub shape: AfxShape,
    pub batch: AfxPresentationSoa,
    pub count: usize,
    pub positions: *const u64,
}

/// Bytes that must not outlive the call: randomness the proofs' nonces come from, staged symmetric keys.
struct Wiped(Vec<u8>);
impl Wiped { fn new(len: usize) -> Wiped { Wiped(vec![0u8; len]) } }
// (the slice impl: the crate takes zeroize without its `alloc` feature, Cargo.toml:39, so `Vec<u8>: Zeroize` is not there)
impl Drop for Wiped { fn drop(&mut self) { self.0.as_mut_slice().zeroize(); } }

// per-item status bytes (AFX_ST_*) and amacs::Attribute kinds (AFX_ATTR_*) of include/aeonflux_gpu.h
const ST_OK: u8 = 0;
const ST_VERIFICATION_FAILURE: u8 = 1;
const ST_MAC_CREATION: u8 = 2;
const ST_NO_SYMMETRIC_KEY: u8 = 3;
const ATTR_PUBLIC_SCALAR: u8 = 0;
const ATTR_SECRET_SCALAR: u8 = 1;
const ATTR_PUBLIC_POINT: u8 = 2;
const ATTR_EITHER_POINT: u8 = 3;
const ATTR_SECRET_POINT: u8 = 4;

extern "C" {
    fn afx_ctx_create(out: *mut *mut c_void, device: i32, sysparams: *const u8, sysparams_len: usize,
                      amacs_key: *const u8, amacs_key_len: usize, issuer_params: *const u8) -> i32;
    fn afx_ctx_destroy(ctx: *mut c_void);
    fn afx_verify_presentations(ctx: *mut c_void, shape: *const AfxShape, batch: *const AfxPresentationSoa,
                                count: usize, status: *mut u8) -> i32;
    fn afx_verify_presentations_mixed(ctx: *mut c_void, groups: *const AfxPresentationGroup, n_groups: usize, status: *mut u8,
                                      status_len: usize) -> i32;
    fn afx_group_verify_presentations_mixed(group: *mut c_void, groups: *const AfxPresentationGroup, n_groups: usize, status: *mut u8,
                                            status_len: usize) -> i32;
    fn afx_issue(ctx: *mut c_void, requests: *const AfxAttributesSoa, rnd: *const AfxIssueRandomness, count: usize,
                 out: *const AfxIssuanceSoa, status: *mut u8) -> i32;
    fn afx_verify_issuances(ctx: *mut c_void, attrs: *const AfxAttributesSoa, issuances: *const AfxIssuanceSoa,
                            n_responses: u32, count: usize, status: *mut u8) -> i32;
    fn afx_show(ctx: *mut c_void, creds: *const AfxCredentialsSoa, keypairs: *const AfxKeypairsSoa,
                rnd: *const AfxShowRandomness, count: usize, out: *const AfxPresentationOut, shape_out: *mut AfxShape,
                status: *mut u8) -> i32;
    fn afx_group_create(out: *mut *mut c_void, devices: *const i32, n_devices: u32, sysparams: *const u8, sysparams_len: usize,
                        amacs_key: *const u8, amacs_key_len: usize, issuer_params: *const u8) -> i32;
    fn afx_group_destroy(group: *mut c_void);
    fn afx_group_verify_presentations(group: *mut c_void, shape: *const AfxShape, batch: *const AfxPresentationSoa,
                                      count: usize, status: *mut u8) -> i32;
    fn afx_group_issue(group: *mut c_void, requests: *const AfxAttributesSoa, rnd: *const AfxIssueRandomness, count: usize,
                       out: *const AfxIssuanceSoa, status: *mut u8) -> i32;
    fn afx_group_verify_issuances(group: *mut c_void, attrs: *const AfxAttributesSoa, issuances: *const AfxIssuanceSoa,
                                  n_responses: u32, count: usize, status: *mut u8) -> i32;
    fn afx_group_show(group: *mut c_void, creds: *const AfxCredentialsSoa, keypairs: *const AfxKeypairsSoa,
                      rnd: *const AfxShowRandomness, count: usize, out: *const AfxPresentationOut, shape_out: *mut AfxShape,
                      status: *mut u8) -> i32;
    fn afx_group_size(group: *const c_void) -> u32;
    fn afx_group_member(group: *mut c_void, index: u32) -> *mut c_void;
    fn afx_ctx_set_secret_independent_addressing(ctx: *mut c_void, enable: i32) -> i32;
}

/// The crate multiplies by secrets in constant time (dalek's `*` and `multiscalar_mul`: src/amacs.rs:267-270,
/// src/nizk/presentation.rs:162-184, zkp's Prover).  The engine's kernels have no secret-dependent branches in any mode; with this
/// switched on no memory ADDRESS depends on a secret scalar either (every table entry is read and the wanted one selected), at
/// the cost INTEGRATION.md section 3 quotes.  Applied to the one context or to every member of the group.
fn set_secret_independent(ctx: *mut c_void, group: *mut c_void, enable: bool) -> Result<(), CredentialError> {
    let mut rc = 0;
    unsafe {
        if group.is_null() { rc |= afx_ctx_set_secret_independent_addressing(ctx, enable as i32); }
        else { for i in 0..afx_group_size(group) { rc |= afx_ctx_set_secret_independent_addressing(afx_group_member(group, i), enable as i32); } }
    }
    if rc != 0 { Err(CredentialError::NoIssuerKey) } else { Ok(()) }   // the only failure: the device could not hold the 4-bit tables
}

/// `Issuer` with its parameters, tables and key resident on one MI355X (`ctx`) or on several (`group`: the batch is split
/// contiguously over the devices inside the library, one host thread per device, no collective).
pub struct GpuIssuer { ctx: *mut c_void, group: *mut c_void, n: usize }

/// The user's side (no issuer key): `AnonymousCredential::show` and `CredentialIssuance::verify`.
pub struct GpuUser { ctx: *mut c_void, group: *mut c_void, n: usize }

fn issuer_params_bytes(ip: &IssuerParameters) -> [u8; 64] {
    let mut b = [0u8; 64];                                            // C_W || I (src/issuer.rs:155,163)
    b[..32].copy_from_slice(ip.C_W.compress().as_bytes());
    b[32..].copy_from_slice(ip.I.compress().as_bytes());
    b
}
fn cell(col: &[u8], row: usize, count: usize, item: usize) -> [u8; 32] {
    let mut b = [0u8; 32];
    b.copy_from_slice(&col[32 * (row * count + item)..32 * (row * count + item) + 32]);
    b
}
// outputs of the engine are canonical scalars / valid encodings by construction; a failure here is an engine bug
fn sc(col: &[u8], row: usize, count: usize, item: usize) -> Scalar { Scalar::from_canonical_bytes(cell(col, row, count, item)).expect("engine returned a non-canonical scalar") }
fn pt(col: &[u8], row: usize, count: usize, item: usize) -> RistrettoPoint { CompressedRistretto(cell(col, row, count, item)).decompress().expect("engine returned an invalid point") }

/// amacs::Attribute (src/amacs.rs:168-179) -> kind byte + the 32-byte value the tag and the proofs use (Messages::from_attributes,
/// src/amacs.rs:225-243: the scalar itself, the point, or a plaintext's M1) + (M2, m3) for plaintext kinds.
fn attribute_cells(a: &Attribute) -> (u8, [u8; 32], Option<([u8; 32], [u8; 32])>) {
    match a {
        Attribute::PublicScalar(m) => (ATTR_PUBLIC_SCALAR, *m.as_bytes(), None),
        Attribute::SecretScalar(m) => (ATTR_SECRET_SCALAR, *m.as_bytes(), None),
        Attribute::PublicPoint(M)  => (ATTR_PUBLIC_POINT, *M.compress().as_bytes(), None),
        Attribute::EitherPoint(p)  => (ATTR_EITHER_POINT, *p.M1.compress().as_bytes(), Some((*p.M2.compress().as_bytes(), *p.m3.as_bytes()))),
        Attribute::SecretPoint(p)  => (ATTR_SECRET_POINT, *p.M1.compress().as_bytes(), Some((*p.M2.compress().as_bytes(), *p.m3.as_bytes()))),
    }
}

/// Column-major staging of a batch: every field one `[count][32]` array, repeated fields `[k][count][32]`.
struct Columns {
    challenge: Vec<u8>, responses: Vec<u8>, c_x_0: Vec<u8>, c_x_1: Vec<u8>, c_v: Vec<u8>, c_y: Vec<u8>, attr_values: Vec<u8>,
    enc: Vec<[Vec<u8>; 9]>,
}

impl GpuIssuer {
    pub fn new(issuer: &Issuer, device: i32) -> Result<GpuIssuer, CredentialError> {
        let sp = issuer.system_parameters.to_bytes();                 // src/parameters.rs:155-184
        let key = issuer.amacs_key.to_bytes();                        // src/amacs.rs:110-125
        let ip = issuer_params_bytes(&issuer.issuer_parameters);
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { afx_ctx_create(&mut ctx, device, sp.as_ptr(), sp.len(), key.as_ptr(), key.len(), ip.as_ptr()) };
        if rc != 0 { return Err(CredentialError::NoIssuerKey); }
        Ok(GpuIssuer { ctx, group: core::ptr::null_mut(), n: issuer.system_parameters.NUMBER_OF_ATTRIBUTES as usize })
    }

    /// The same issuer on several GPUs of one node: every batch call below is split contiguously over `devices`.
    pub fn new_multi(issuer: &Issuer, devices: &[i32]) -> Result<GpuIssuer, CredentialError> {
        let sp = issuer.system_parameters.to_bytes();
        let key = issuer.amacs_key.to_bytes();
        let ip = issuer_params_bytes(&issuer.issuer_parameters);
        let mut group = core::ptr::null_mut();
        let rc = unsafe { afx_group_create(&mut group, devices.as_ptr(), devices.len() as u32, sp.as_ptr(), sp.len(), key.as_ptr(), key.len(), ip.as_ptr()) };
        if rc != 0 { return Err(CredentialError::NoIssuerKey); }
        Ok(GpuIssuer { ctx: core::ptr::null_mut(), group, n: issuer.system_parameters.NUMBER_OF_ATTRIBUTES as usize })
    }

    /// Constant-address table reads for every scalar of `issue_batch` and for the issuer key's terms of `verify_batch` (see
    /// `set_secret_independent`): what a deployment that relies on the crate's constant-time arithmetic switches on.
    pub fn set_secret_independent_addressing(&self, enable: bool) -> Result<(), CredentialError> { set_secret_independent(self.ctx, self.group, enable) }

    /// Batch `Issuer::issue` (src/issuer.rs:111-124): consumes the requests like the reference does and returns one
    /// `Result` per request, in order.  All requests must share one attribute layout (same kinds per position).
    pub fn issue_batch<C: CryptoRng + RngCore>(&self, requests: Vec<CredentialRequest>, csprng: &mut C)
        -> Vec<Result<CredentialIssuance, CredentialError>>
    {
        let count = requests.len();
        if count == 0 { return Vec::new(); }
        let na = requests[0].attributes.len();
        assert!(na <= AFX_MAX_ATTRIBUTES, "issue_batch: more than AFX_MAX_ATTRIBUTES attributes");
        let mut soa = AfxAttributesSoa { n_attributes: na as u32, kinds: [0; AFX_MAX_ATTRIBUTES], values: core::ptr::null() };
        let mut values = vec![0u8; 32 * na * count];
        for (i, r) in requests.iter().enumerate() {
            assert!(r.attributes.len() == na, "issue_batch: mixed attribute counts; group requests by layout first");
            for (k, a) in r.attributes.iter().enumerate() {
                let (kind, v, _) = attribute_cells(a);
                if i == 0 { soa.kinds[k] = kind; } else { assert!(soa.kinds[k] == kind, "issue_batch: mixed attribute kinds"); }
                values[32 * (k * count + i)..32 * (k * count + i) + 32].copy_from_slice(&v);
            }
        }
        soa.values = values.as_ptr();
        // the reference's draws, in its order (see the header of this file)
        let (mut t_wide, mut u_wide, mut seed) = (Wiped::new(64 * count), Wiped::new(64 * count), Wiped::new(32 * count));
        for i in 0..count {
            csprng.fill_bytes(&mut t_wide.0[64 * i..64 * i + 64]);    // Scalar::random          (src/amacs.rs:289)
            csprng.fill_bytes(&mut u_wide.0[64 * i..64 * i + 64]);    // RistrettoPoint::random  (src/amacs.rs:290)
        }
        csprng.fill_bytes(&mut seed.0);                               // in place of zkp prove_compact's thread_rng() draw, 32 B per proof
        let rnd = AfxIssueRandomness { t_wide: t_wide.0.as_ptr(), U_wide: u_wide.0.as_ptr(), rng_seed: seed.0.as_ptr() };
        let nr = self.n + 5;                                          // w, w', x_0, x_1, y_0..y_{n-1}, "1" (src/nizk/issuance.rs:52-68)
        let (mut t, mut u, mut v, mut ch, mut rs) = (vec![0u8; 32 * count], vec![0u8; 32 * count], vec![0u8; 32 * count], vec![0u8; 32 * count], vec![0u8; 32 * nr * count]);
        let out = AfxIssuanceSoa { t: t.as_mut_ptr(), U: u.as_mut_ptr(), V: v.as_mut_ptr(), challenge: ch.as_mut_ptr(), responses: rs.as_mut_ptr() };
        let mut status = vec![0u8; count];
        let rc = unsafe {
            if self.group.is_null() { afx_issue(self.ctx, &soa, &rnd, count, &out, status.as_mut_ptr()) }
            else { afx_group_issue(self.group, &soa, &rnd, count, &out, status.as_mut_ptr()) }
        };
        assert!(rc == 0, "aeonflux_gpu: engine error {}", rc);
        requests.into_iter().enumerate().map(|(i, request)| {
            if status[i] == ST_MAC_CREATION { return Err(CredentialError::MacCreation); }   // amacs.rs:285-287 -> errors.rs:141-142
            assert!(status[i] == ST_OK, "aeonflux_gpu: unexpected issue status {}", status[i]);
            let amac = Amac { t: sc(&t, 0, count, i), U: pt(&u, 0, count, i), V: pt(&v, 0, count, i) };
            let proof = CompactProof { challenge: sc(&ch, 0, count, i), responses: (0..nr).map(|k| sc(&rs, k, count, i)).collect() };
            Ok(CredentialIssuance { proof: ProofOfIssuance(proof), credential: AnonymousCredential { amac, attributes: request.attributes } })
        }).collect()
    }

    /// Batch `Issuer::verify` (src/issuer.rs:141-147) over ANY presentations: like the reference, which reads the shape from
    /// each presentation's own fields (src/nizk/presentation.rs:293-309, :324-443), the batch is grouped by the full shape -
    /// attribute kinds, hidden scalar indices, response count, the indices of the attached proofs of encryption - and every
    /// group is verified under its own statement.  Results come back in the order given.  A presentation whose vectors do not
    /// fit together (lengths the reference would index out of range on, `presentation.rs:346,351,407`; a proof of encryption
    /// without its six responses, which zkp rejects) is answered with `VerificationFailure` without reaching the engine.
    pub fn verify_batch(&self, batch: &[ProofOfValidCredential]) -> Vec<Result<(), CredentialError>> {
        let total = batch.len();
        if total == 0 { return Vec::new(); }
        let mut status = vec![ST_VERIFICATION_FAILURE; total];
        let mut by_shape: std::collections::BTreeMap<Vec<u8>, Vec<usize>> = std::collections::BTreeMap::new();
        for (i, p) in batch.iter().enumerate() {
            if let Some(key) = shape_key(p) { by_shape.entry(key).or_insert_with(Vec::new).push(i); }   // else: stays a failure
        }
        // the staged columns, position lists and struct arrays of every group live until the call returns
        let staged: Vec<(AfxShape, Columns, Vec<u64>)> = by_shape.values().map(|members| {
            let items: Vec<&ProofOfValidCredential> = members.iter().map(|i| &batch[*i]).collect();
            let (shape, cols) = marshal(&items);
            (shape, cols, members.iter().map(|i| *i as u64).collect())
        }).collect();
        let enc_soas: Vec<Vec<AfxEncProofSoa>> = staged.iter().map(|(_, cols, _)| cols.enc.iter().map(|e| AfxEncProofSoa {
            challenge: e[0].as_ptr(), responses: e[1].as_ptr(), pk: e[2].as_ptr(), E1: e[3].as_ptr(), E2: e[4].as_ptr(),
            C_y_1: e[5].as_ptr(), C_y_2: e[6].as_ptr(), C_y_3: e[7].as_ptr(), C_y_2p: e[8].as_ptr() }).collect()).collect();
        let groups: Vec<AfxPresentationGroup> = staged.iter().zip(enc_soas.iter()).map(|((shape, cols, positions), enc)| AfxPresentationGroup {
            shape: *shape,
            batch: AfxPresentationSoa {
                challenge: cols.challenge.as_ptr(), responses: cols.responses.as_ptr(), C_x_0: cols.c_x_0.as_ptr(),
                C_x_1: cols.c_x_1.as_ptr(), C_V: cols.c_v.as_ptr(), C_y: cols.c_y.as_ptr(), attr_values: cols.attr_values.as_ptr(),
                enc: enc.as_ptr() },
            count: positions.len(),
            positions: positions.as_ptr(),
        }).collect();
        let rc = unsafe {
            if self.group.is_null() { afx_verify_presentations_mixed(self.ctx, groups.as_ptr(), groups.len(), status.as_mut_ptr(), total) }
            else { afx_group_verify_presentations_mixed(self.group, groups.as_ptr(), groups.len(), status.as_mut_ptr(), total) }
        };
        assert!(rc == 0, "aeonflux_gpu: engine error {}", rc);
        status.iter().map(|s| if *s == ST_OK { Ok(()) } else { Err(CredentialError::VerificationFailure) }).collect()
    }
}

impl Drop for GpuIssuer {
    // wipes every key copy, host and device (Zeroize + Drop of amacs::SecretKey, src/amacs.rs:64-82)
    fn drop(&mut self) { unsafe { if self.group.is_null() { afx_ctx_destroy(self.ctx) } else { afx_group_destroy(self.group) } } }
}

impl GpuUser {
    pub fn new(system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters, device: i32) -> Result<GpuUser, CredentialError> {
        let sp = system_parameters.to_bytes();
        let ip = issuer_params_bytes(issuer_parameters);
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { afx_ctx_create(&mut ctx, device, sp.as_ptr(), sp.len(), core::ptr::null(), 0, ip.as_ptr()) };
        if rc != 0 { return Err(CredentialError::NoSystemParameters); }
        Ok(GpuUser { ctx, group: core::ptr::null_mut(), n: system_parameters.NUMBER_OF_ATTRIBUTES as usize })
    }

    /// The same on several GPUs of the node: batches are split contiguously over `devices` inside the library.
    pub fn new_multi_user(system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters, devices: &[i32]) -> Result<GpuUser, CredentialError> {
        let sp = system_parameters.to_bytes();
        let ip = issuer_params_bytes(issuer_parameters);
        let mut group = core::ptr::null_mut();
        let rc = unsafe { afx_group_create(&mut group, devices.as_ptr(), devices.len() as u32, sp.as_ptr(), sp.len(), core::ptr::null(), 0, ip.as_ptr()) };
        if rc != 0 { return Err(CredentialError::NoSystemParameters); }
        Ok(GpuUser { ctx: core::ptr::null_mut(), group, n: system_parameters.NUMBER_OF_ATTRIBUTES as usize })
    }

    /// Batch `AnonymousCredential::show` (src/credential.rs:37-46 -> src/nizk/presentation.rs:139-321).  One keypair per
    /// credential (or `None`: a credential with a SecretPoint attribute then yields `NoSymmetricKey`, presentation.rs:150-157).
    /// All credentials must share one layout after their hide_attribute / reveal_attribute calls.
    /// Constant-address table reads for every scalar of `show_batch` (blindings, the credential's `t`, the symmetric key).
    pub fn set_secret_independent_addressing(&self, enable: bool) -> Result<(), CredentialError> { set_secret_independent(self.ctx, self.group, enable) }

    pub fn show_batch<C: CryptoRng + RngCore>(&self, creds: &[AnonymousCredential], keypairs: Option<&[SymmetricKeypair]>, csprng: &mut C)
        -> Vec<Result<ProofOfValidCredential, CredentialError>>
    {
        let count = creds.len();
        if count == 0 { return Vec::new(); }
        let na = creds[0].attributes.len();
        assert!(na <= AFX_MAX_ATTRIBUTES, "show_batch: more than AFX_MAX_ATTRIBUTES attributes");
        let mut cs = AfxCredentialsSoa { n_attributes: na as u32, kinds: [0; AFX_MAX_ATTRIBUTES], values: core::ptr::null(), M2: core::ptr::null(),
                                         m3: core::ptr::null(), t: core::ptr::null(), U: core::ptr::null(), V: core::ptr::null() };
        // hidden attribute values are secrets of the user (amacs::Attribute zeroizes them, src/amacs.rs:184-200)
        let (mut values_w, mut m2_w, mut m3_w) = (Wiped::new(32 * na * count), Wiped::new(32 * na * count), Wiped::new(32 * na * count));
        let (values, m2, m3) = (&mut values_w.0, &mut m2_w.0, &mut m3_w.0);
        let (mut t, mut u, mut v) = (vec![0u8; 32 * count], vec![0u8; 32 * count], vec![0u8; 32 * count]);
        for (i, c) in creds.iter().enumerate() {
            assert!(c.attributes.len() == na, "show_batch: mixed attribute counts; group credentials by layout first");
            for (k, a) in c.attributes.iter().enumerate() {
                let (kind, val, plain) = attribute_cells(a);
                if i == 0 { cs.kinds[k] = kind; } else { assert!(cs.kinds[k] == kind, "show_batch: mixed attribute kinds"); }
                let at = 32 * (k * count + i);
                values[at..at + 32].copy_from_slice(&val);
                if let Some((p2, s3)) = plain { m2[at..at + 32].copy_from_slice(&p2); m3[at..at + 32].copy_from_slice(&s3); }
            }
            t[32 * i..32 * i + 32].copy_from_slice(c.amac.t.as_bytes());
            u[32 * i..32 * i + 32].copy_from_slice(c.amac.U.compress().as_bytes());
            v[32 * i..32 * i + 32].copy_from_slice(c.amac.V.compress().as_bytes());
        }
        cs.values = values.as_ptr(); cs.M2 = m2.as_ptr(); cs.m3 = m3.as_ptr(); cs.t = t.as_ptr(); cs.U = u.as_ptr(); cs.V = v.as_ptr();
        let hs = (0..na).filter(|k| cs.kinds[*k] == ATTR_SECRET_SCALAR).count();
        let secret_points: Vec<usize> = (0..na).filter(|k| cs.kinds[*k] == ATTR_SECRET_POINT).collect();
        let nsp = secret_points.len();
        // keypairs (symmetric::Keypair, src/symmetric.rs:52-81)
        let (mut ka, mut ka0, mut ka1, mut kpk) = (Wiped::new(32 * count), Wiped::new(32 * count), Wiped::new(32 * count), vec![0u8; 32 * count]);
        if let Some(kps) = keypairs {
            assert!(kps.len() == count, "show_batch: one keypair per credential");
            for (i, kp) in kps.iter().enumerate() {
                ka.0[32 * i..32 * i + 32].copy_from_slice(kp.secret.a.as_bytes());
                ka0.0[32 * i..32 * i + 32].copy_from_slice(kp.secret.a0.as_bytes());
                ka1.0[32 * i..32 * i + 32].copy_from_slice(kp.secret.a1.as_bytes());
                kpk[32 * i..32 * i + 32].copy_from_slice(kp.public.pk.compress().as_bytes());
            }
        }
        let kp_soa = AfxKeypairsSoa { a: ka.0.as_ptr(), a0: ka0.0.as_ptr(), a1: ka1.0.as_ptr(), pk: kpk.as_ptr() };
        // the reference's csprng draws in its order, then the proofs' seeds (see the header of this file)
        let (mut z_wide, mut seed, mut enc_seeds) = (Wiped::new(64 * count), Wiped::new(32 * count), Wiped::new(32 * count * nsp.max(1)));
        for i in 0..count { csprng.fill_bytes(&mut z_wide.0[64 * i..64 * i + 64]); }   // Scalar::random (presentation.rs:162)
        csprng.fill_bytes(&mut seed.0);                                               // in place of thread_rng() in the presentation proof's prove_compact (:284)
        csprng.fill_bytes(&mut enc_seeds.0);                                          // ... and in each ProofOfEncryption's (:301), [secret point][credential]
        let rnd = AfxShowRandomness { z_wide: z_wide.0.as_ptr(), rng_seed: seed.0.as_ptr(), enc_seeds: enc_seeds.0.as_ptr() };
        // outputs
        let col = |k: usize| vec![0u8; 32 * k * count];
        let (mut o_ch, mut o_rs, mut o_x0, mut o_x1, mut o_cv, mut o_cy, mut o_av) = (col(1), col(3 + hs), col(1), col(1), col(1), col(na), col(na));
        let mut enc_cols: Vec<[Vec<u8>; 9]> = (0..nsp).map(|_| [col(1), col(6), col(1), col(1), col(1), col(1), col(1), col(1), col(1)]).collect();
        let enc_out: Vec<AfxEncProofOut> = enc_cols.iter_mut().map(|e| AfxEncProofOut {
            challenge: e[0].as_mut_ptr(), responses: e[1].as_mut_ptr(), pk: e[2].as_mut_ptr(), E1: e[3].as_mut_ptr(), E2: e[4].as_mut_ptr(),
            C_y_1: e[5].as_mut_ptr(), C_y_2: e[6].as_mut_ptr(), C_y_3: e[7].as_mut_ptr(), C_y_2p: e[8].as_mut_ptr() }).collect();
        let out = AfxPresentationOut { challenge: o_ch.as_mut_ptr(), responses: o_rs.as_mut_ptr(), C_x_0: o_x0.as_mut_ptr(), C_x_1: o_x1.as_mut_ptr(),
                                       C_V: o_cv.as_mut_ptr(), C_y: o_cy.as_mut_ptr(), attr_values: o_av.as_mut_ptr(), enc: enc_out.as_ptr() };
        let mut shape = AfxShape { n_attributes: 0, kinds: [0; 32], n_responses: 0, n_hidden_scalars: 0, hidden_scalar_indices: [0; 32],
                                   n_enc_proofs: 0, enc_indices: [0; 32] };
        let mut status = vec![0u8; count];
        let kp_ptr: *const AfxKeypairsSoa = if keypairs.is_some() { &kp_soa } else { core::ptr::null() };
        let rc = unsafe {
            if self.group.is_null() { afx_show(self.ctx, &cs, kp_ptr, &rnd, count, &out, &mut shape, status.as_mut_ptr()) }
            else { afx_group_show(self.group, &cs, kp_ptr, &rnd, count, &out, &mut shape, status.as_mut_ptr()) }
        };
        assert!(rc == 0, "aeonflux_gpu: engine error {}", rc);
        // rebuild ProofOfValidCredential (src/nizk/presentation.rs:118-127) per item
        (0..count).map(|i| {
            match status[i] {
                ST_OK => {}
                ST_NO_SYMMETRIC_KEY => return Err(CredentialError::NoSymmetricKey),
                ST_VERIFICATION_FAILURE | _ => return Err(CredentialError::VerificationFailure),
            }
            let proof = CompactProof { challenge: sc(&o_ch, 0, count, i), responses: (0..3 + hs).map(|k| sc(&o_rs, k, count, i)).collect() };
            let encrypted_attributes = (0..na).map(|k| match shape.kinds[k] {
                0 => EncryptedAttribute::PublicScalar(sc(&o_av, k, count, i)),
                1 => EncryptedAttribute::SecretScalar,
                2 => EncryptedAttribute::PublicPoint(pt(&o_av, k, count, i)),
                _ => EncryptedAttribute::SecretPoint,
            }).collect();
            let proofs_of_encryption = (0..nsp).map(|e| {
                let c = &enc_cols[e];
                let index = shape.enc_indices[e];
                (index, ProofOfEncryption {
                    proof: CompactProof { challenge: sc(&c[0], 0, count, i), responses: (0..6).map(|k| sc(&c[1], k, count, i)).collect() },
                    public_key: SymmetricPublicKey { pk: pt(&c[2], 0, count, i) },
                    ciphertext: Ciphertext { E1: pt(&c[3], 0, count, i), E2: pt(&c[4], 0, count, i) },
                    index,
                    C_y_1: pt(&c[5], 0, count, i), C_y_2: pt(&c[6], 0, count, i), C_y_3: pt(&c[7], 0, count, i), C_y_2_prime: pt(&c[8], 0, count, i),
                })
            }).collect();
            Ok(ProofOfValidCredential {
                proof, proofs_of_encryption, encrypted_attributes,
                hidden_scalar_indices: shape.hidden_scalar_indices[..shape.n_hidden_scalars as usize].to_vec(),
                C_x_0: pt(&o_x0, 0, count, i), C_x_1: pt(&o_x1, 0, count, i), C_V: pt(&o_cv, 0, count, i),
                C_y: (0..na).map(|k| pt(&o_cy, k, count, i)).collect(),
            })
        }).collect()
    }

    /// Batch `CredentialIssuance::verify` (src/issuer.rs:48-57): consumes the issuances and moves each credential out on success.
    pub fn verify_issuance_batch(&self, issuances: Vec<CredentialIssuance>) -> Vec<Result<AnonymousCredential, CredentialError>> {
        let count = issuances.len();
        if count == 0 { return Vec::new(); }
        let na = issuances[0].credential.attributes.len();
        let nr = issuances[0].proof.0.responses.len();
        assert!(na <= AFX_MAX_ATTRIBUTES && nr <= AFX_MAX_ATTRIBUTES + 5, "verify_issuance_batch: layout beyond AFX_MAX_ATTRIBUTES");
        let mut soa = AfxAttributesSoa { n_attributes: na as u32, kinds: [0; AFX_MAX_ATTRIBUTES], values: core::ptr::null() };
        let mut values = vec![0u8; 32 * na * count];
        let col = |k: usize| vec![0u8; 32 * k * count];
        let (mut t, mut u, mut v, mut ch, mut rs) = (col(1), col(1), col(1), col(1), col(nr));
        for (i, iss) in issuances.iter().enumerate() {
            assert!(iss.credential.attributes.len() == na && iss.proof.0.responses.len() == nr, "verify_issuance_batch: mixed layouts");
            for (k, a) in iss.credential.attributes.iter().enumerate() {
                let (kind, val, _) = attribute_cells(a);
                if i == 0 { soa.kinds[k] = kind; } else { assert!(soa.kinds[k] == kind, "verify_issuance_batch: mixed attribute kinds"); }
                values[32 * (k * count + i)..32 * (k * count + i) + 32].copy_from_slice(&val);
            }
            t[32 * i..32 * i + 32].copy_from_slice(iss.credential.amac.t.as_bytes());
            u[32 * i..32 * i + 32].copy_from_slice(iss.credential.amac.U.compress().as_bytes());
            v[32 * i..32 * i + 32].copy_from_slice(iss.credential.amac.V.compress().as_bytes());
            ch[32 * i..32 * i + 32].copy_from_slice(iss.proof.0.challenge.as_bytes());
            for (k, r) in iss.proof.0.responses.iter().enumerate() { rs[32 * (k * count + i)..32 * (k * count + i) + 32].copy_from_slice(r.as_bytes()); }
        }
        soa.values = values.as_ptr();
        let d = AfxIssuanceSoa { t: t.as_mut_ptr(), U: u.as_mut_ptr(), V: v.as_mut_ptr(), challenge: ch.as_mut_ptr(), responses: rs.as_mut_ptr() };
        let mut status = vec![0u8; count];
        let rc = unsafe {
            if self.group.is_null() { afx_verify_issuances(self.ctx, &soa, &d, nr as u32, count, status.as_mut_ptr()) }
            else { afx_group_verify_issuances(self.group, &soa, &d, nr as u32, count, status.as_mut_ptr()) }
        };
        assert!(rc == 0, "aeonflux_gpu: engine error {}", rc);
        issuances.into_iter().enumerate().map(|(i, iss)| if status[i] == ST_OK { Ok(iss.credential) } else { Err(CredentialError::VerificationFailure) }).collect()
    }
}

impl Drop for GpuUser {
    fn drop(&mut self) { unsafe { if self.group.is_null() { afx_ctx_destroy(self.ctx) } else { afx_group_destroy(self.group) } } }
}

fn enc_kind(a: &EncryptedAttribute) -> u8 {
    match a { EncryptedAttribute::PublicScalar(_) => 0, EncryptedAttribute::SecretScalar => 1,
              EncryptedAttribute::PublicPoint(_) => 2, EncryptedAttribute::SecretPoint => 3 }
}

/// Everything of a presentation that is not a scalar or a point, as bytes: two presentations may share a GPU batch iff their
/// keys are equal.  `None`: vectors that do not fit together or do not fit the ABI - the reference would index out of range
/// (`self.C_y[i]`, `self.encrypted_attributes[i]`, presentation.rs:346,351,374) or zkp would reject the response count.
fn shape_key(p: &ProofOfValidCredential) -> Option<Vec<u8>> {
    let (n, nr, hs, ne) = (p.encrypted_attributes.len(), p.proof.responses.len(), p.hidden_scalar_indices.len(), p.proofs_of_encryption.len());
    if n > AFX_MAX_ATTRIBUTES || hs > AFX_MAX_ATTRIBUTES || ne > AFX_MAX_ATTRIBUTES || nr > AFX_MAX_ATTRIBUTES + 3 || p.C_y.len() != n { return None; }
    if p.proofs_of_encryption.iter().any(|(_, q)| q.proof.responses.len() != 6) { return None; }
    let mut k = Vec::with_capacity(16 + n + 2 * (hs + ne));
    for v in [n as u32, nr as u32, hs as u32, ne as u32].iter() { k.extend_from_slice(&v.to_le_bytes()); }
    k.extend(p.encrypted_attributes.iter().map(enc_kind));
    for h in p.hidden_scalar_indices.iter() { k.extend_from_slice(&h.to_le_bytes()); }
    for (_, q) in p.proofs_of_encryption.iter() { k.extend_from_slice(&q.index.to_le_bytes()); }
    Some(k)
}

/// Presentations of ONE shape (equal `shape_key`s) -> shape + columns.  Lives inside the crate because the struct's fields
/// (src/nizk/presentation.rs:118-127) are private.
fn marshal(batch: &[&ProofOfValidCredential]) -> (AfxShape, Columns) {
    let count = batch.len();
    let p0 = batch[0];
    let n = p0.encrypted_attributes.len();
    let nr = p0.proof.responses.len();
    let ne = p0.proofs_of_encryption.len();
    let mut shape = AfxShape { n_attributes: n as u32, kinds: [0; 32], n_responses: nr as u32,
        n_hidden_scalars: p0.hidden_scalar_indices.len() as u32, hidden_scalar_indices: [0; 32],
        n_enc_proofs: ne as u32, enc_indices: [0; 32] };
    for (i, a) in p0.encrypted_attributes.iter().enumerate() { shape.kinds[i] = enc_kind(a); }
    for (i, h) in p0.hidden_scalar_indices.iter().enumerate() { shape.hidden_scalar_indices[i] = *h; }
    for (i, (_, e)) in p0.proofs_of_encryption.iter().enumerate() { shape.enc_indices[i] = e.index; }
    let col = |k: usize| vec![0u8; 32 * k * count];
    let mut c = Columns { challenge: col(1), responses: col(nr), c_x_0: col(1), c_x_1: col(1), c_v: col(1), c_y: col(n),
                          attr_values: col(n), enc: (0..ne).map(|_| [col(1), col(6), col(1), col(1), col(1), col(1), col(1), col(1), col(1)]).collect() };
    let put = |dst: &mut Vec<u8>, row: usize, item: usize, src: &[u8; 32]| dst[32 * (row * count + item)..32 * (row * count + item) + 32].copy_from_slice(src);
    for (i, p) in batch.iter().enumerate() {
        debug_assert!(shape_key(p) == shape_key(p0), "marshal: presentations of different shapes in one group");
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
