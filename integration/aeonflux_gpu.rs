// Reference-side binding for libaeonflux_gpu.so.  SOURCE ONLY: this image has no Rust
// toolchain, so this file has not been compiled here.  It is the `mod gpu` a maintainer would add to the aeonflux
// crate (/root/reference/src/) together with `build.rs` emitting `cargo:rustc-link-lib=dylib=aeonflux_gpu`.
//
// 1. The crate's own entry points, signature for signature, each a batch of one on the engine's latency plan:
//      GpuIssuer::issue(&self, request, csprng)                      = Issuer::issue               (src/issuer.rs:111-118)
//      GpuIssuer::verify(&self, presentation)                        = Issuer::verify              (src/issuer.rs:141-145)
//      GpuCredential::show(&self, system_parameters, issuer_parameters, keypair, csprng)
//                                                                    = AnonymousCredential::show   (src/credential.rs:37-43)
//      GpuIssuance::verify(self, system_parameters, issuer_parameters) = CredentialIssuance::verify (src/issuer.rs:48-52)
//    (`GpuUser::credential(&cred)` / `GpuUser::issuance(iss)` pair the crate's value with the engine that serves it, so `self` is
//    what it is in the crate.)  `install_issuer` / `install_user` register an engine for the process; the three-line patches of
//    INTEGRATION.md section 1 then make the crate's own methods delegate - no call site changes.
// 2. Batch forms of the same four, over ANY mix of attribute layouts / presentation shapes, results in the order given:
//      GpuIssuer::issue_batch, GpuIssuer::verify_batch, GpuUser::show_batch, GpuUser::verify_issuance_batch
//    and GpuIssuer::new_multi / GpuUser::new_multi_user put the same engine on several GPUs (afx_group_*).
//
// Errors are values (src/errors.rs:73-89): a per-item engine status becomes the `CredentialError` the crate returns for that
// outcome; a batch-level engine failure (no device, a HIP error, out of memory: AFX_E_*) becomes an `Err` for every item of the
// call (`engine_error` below maps it; `last_engine_code()` keeps the raw code so a caller can tell an infrastructure failure from
// a cryptographic one and fall back to the crate's CPU path).  Nothing here panics on caller data or on an engine fault.
//
// Randomness.  The engine takes every random draw as an input array.  The shim draws, from the CALLER's csprng only:
//   issue : per request, in the reference's order: 64 bytes for `Scalar::random` (t, src/amacs.rs:289), then 64 bytes for
//           `RistrettoPoint::random` (U, src/amacs.rs:290); after all of those, 32 bytes per request that stand in for the draw
//           zkp's `prove_compact` makes from thread_rng() through merlin's `TranscriptRngBuilder::finalize` [3P].
//   show  : per credential 64 bytes for `Scalar::random` (z, src/nizk/presentation.rs:162); after all of those, 32 bytes per
//           credential for the presentation proof's own `prove_compact` (presentation.rs:284); after all of those, credential by
//           credential, 32 bytes per ProofOfEncryption, one per SecretPoint attribute in attribute order
//           (presentation.rs:293-309 -> src/nizk/encryption.rs:141).
// So a caller's deterministic csprng yields the reference's t, U and z, and the proofs' synthetic nonces come from the same
// generator instead of the reference's hidden thread_rng() - `rand` is only a dev-dependency of the crate (Cargo.toml:42-45) and
// the shim must not need it.  Every buffer that held such bytes, the staged issuer key, the user's symmetric keys and hidden
// attribute values are zeroized on drop, as the crate does for what they become (src/amacs.rs:64-82, src/symmetric.rs:51-64).
//
// `#![no_std]` like the crate (src/lib.rs:11): only `core` and `alloc` are used.  The #[repr(C)] structs below mirror
// include/aeonflux_gpu.h field for field; tests/test_integration_layouts.py checks names, order and widths, the four
// signatures above and the absence of panicking paths against the header and the crate without a Rust compiler.
#![allow(non_snake_case)]

extern crate alloc;

use alloc::boxed::Box;
use alloc::collections::BTreeMap;
use alloc::vec;
use alloc::vec::Vec;
use core::ffi::c_void;
use core::sync::atomic::{AtomicI32, AtomicPtr, Ordering};

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

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxEncProofSoa {
    pub challenge: *const u8, pub responses: *const u8, pub pk: *const u8, pub E1: *const u8, pub E2: *const u8,
    pub C_y_1: *const u8, pub C_y_2: *const u8, pub C_y_3: *const u8, pub C_y_2p: *const u8,
}

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxPresentationSoa {
    pub challenge: *const u8, pub responses: *const u8, pub C_x_0: *const u8, pub C_x_1: *const u8, pub C_V: *const u8,
    pub C_y: *const u8, pub attr_values: *const u8, pub enc: *const AfxEncProofSoa,
}

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxAttributesSoa {
    pub n_attributes: u32,
    pub kinds: [u8; AFX_MAX_ATTRIBUTES],
    pub values: *const u8,
}

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxIssueRandomness { pub t_wide: *const u8, pub U_wide: *const u8, pub rng_seed: *const u8 }

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxIssuanceSoa { pub t: *mut u8, pub U: *mut u8, pub V: *mut u8, pub challenge: *mut u8, pub responses: *mut u8 }

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxCredentialsSoa {
    pub n_attributes: u32,
    pub kinds: [u8; AFX_MAX_ATTRIBUTES],
    pub values: *const u8, pub M2: *const u8, pub m3: *const u8, pub t: *const u8, pub U: *const u8, pub V: *const u8,
}

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxKeypairsSoa { pub a: *const u8, pub a0: *const u8, pub a1: *const u8, pub pk: *const u8 }

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxShowRandomness { pub z_wide: *const u8, pub rng_seed: *const u8, pub enc_seeds: *const u8 }

#[derive(Clone, Copy)]
#[repr(C)]
pub struct AfxEncProofOut {
    pub challenge: *mut u8, pub responses: *mut u8, pub pk: *mut u8, pub E1: *mut u8, pub E2: *mut u8,
    pub C_y_1: *mut u8, pub C_y_2: *mut u8, pub C_y_3: *mut u8, pub C_y_2p: *mut u8,
}

#[derive(Clone, Copy)]
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

#[repr(C)]
pub struct AfxIssueGroup {
    pub requests: AfxAttributesSoa,
    pub rnd: AfxIssueRandomness,
    pub out: AfxIssuanceSoa,
    pub count: usize,
    pub positions: *const u64,
}

#[repr(C)]
pub struct AfxIssuanceGroup {
    pub attrs: AfxAttributesSoa,
    pub issuances: AfxIssuanceSoa,
    pub n_responses: u32,
    pub count: usize,
    pub positions: *const u64,
}

#[repr(C)]
pub struct AfxShowGroup {
    pub creds: AfxCredentialsSoa,
    pub keypairs: *const AfxKeypairsSoa,
    pub rnd: AfxShowRandomness,
    pub out: AfxPresentationOut,
    pub shape_out: AfxShape,
    pub count: usize,
    pub positions: *const u64,
}

/// Bytes that must not outlive the call: randomness the proofs' nonces come from, the staged issuer key, staged symmetric
/// keys and hidden attribute values.
struct Wiped(Vec<u8>);
impl Wiped { fn new(len: usize) -> Wiped { Wiped(vec![0u8; len]) } }
// (the slice impl: the crate takes zeroize without its `alloc` feature, Cargo.toml:39, so `Vec<u8>: Zeroize` is not there)
impl Drop for Wiped { fn drop(&mut self) { self.0.as_mut_slice().zeroize(); } }

// per-item status bytes (AFX_ST_*), batch-level return codes (AFX_E_*) and amacs::Attribute kinds (AFX_ATTR_*) of include/aeonflux_gpu.h
const ST_OK: u8 = 0;
const ST_VERIFICATION_FAILURE: u8 = 1;
const ST_MAC_CREATION: u8 = 2;
const ST_NO_SYMMETRIC_KEY: u8 = 3;
const E_BAD_ARGS: i32 = -1;
const E_BAD_PARAMS: i32 = -2;
const E_NO_KEY: i32 = -5;
const ATTR_PUBLIC_SCALAR: u8 = 0;
const ATTR_SECRET_SCALAR: u8 = 1;
const ATTR_PUBLIC_POINT: u8 = 2;
const ATTR_EITHER_POINT: u8 = 3;
const ATTR_SECRET_POINT: u8 = 4;

extern "C" {
    fn afx_ctx_create(out: *mut *mut c_void, device: i32, sysparams: *const u8, sysparams_len: usize,
                      amacs_key: *const u8, amacs_key_len: usize, issuer_params: *const u8) -> i32;
    fn afx_ctx_destroy(ctx: *mut c_void);
    fn afx_verify_presentations_mixed(ctx: *mut c_void, groups: *const AfxPresentationGroup, n_groups: usize, status: *mut u8,
                                      status_len: usize) -> i32;
    fn afx_group_verify_presentations_mixed(group: *mut c_void, groups: *const AfxPresentationGroup, n_groups: usize, status: *mut u8,
                                            status_len: usize) -> i32;
    fn afx_issue_mixed(ctx: *mut c_void, groups: *const AfxIssueGroup, n_groups: usize, status: *mut u8, status_len: usize) -> i32;
    fn afx_group_issue_mixed(group: *mut c_void, groups: *const AfxIssueGroup, n_groups: usize, status: *mut u8, status_len: usize) -> i32;
    fn afx_verify_issuances_mixed(ctx: *mut c_void, groups: *const AfxIssuanceGroup, n_groups: usize, status: *mut u8, status_len: usize) -> i32;
    fn afx_group_verify_issuances_mixed(group: *mut c_void, groups: *const AfxIssuanceGroup, n_groups: usize, status: *mut u8, status_len: usize) -> i32;
    fn afx_show_mixed(ctx: *mut c_void, groups: *mut AfxShowGroup, n_groups: usize, status: *mut u8, status_len: usize) -> i32;
    fn afx_group_show_mixed(group: *mut c_void, groups: *mut AfxShowGroup, n_groups: usize, status: *mut u8, status_len: usize) -> i32;
    fn afx_group_create(out: *mut *mut c_void, devices: *const i32, n_devices: u32, sysparams: *const u8, sysparams_len: usize,
                        amacs_key: *const u8, amacs_key_len: usize, issuer_params: *const u8) -> i32;
    fn afx_group_destroy(group: *mut c_void);
    fn afx_group_size(group: *const c_void) -> u32;
    fn afx_group_member(group: *mut c_void, index: u32) -> *mut c_void;
    fn afx_ctx_set_secret_independent_addressing(ctx: *mut c_void, mode: i32) -> i32;
}

/// Which of the crate's operations an engine failure interrupted: decides the nearest `CredentialError`.
#[derive(Clone, Copy)]
enum Op { Create, Issue, Verify, Show, VerifyIssuance }

/// A batch-level engine return code (AFX_E_*, include/aeonflux_gpu.h) as the crate's error type (src/errors.rs:73-89), which has
/// no variant for "the accelerator failed".  Codes about the caller's data map to what the crate would say; an infrastructure
/// failure (no device, HIP error, out of memory) maps to the operation's own failure variant, so that nothing is ever accepted or
/// issued on a fault: verification fails closed.  `last_engine_code()` tells the two apart.
fn engine_error(rc: i32, op: Op) -> CredentialError {
    match (rc, op) {
        (E_BAD_PARAMS, _) => CredentialError::NoSystemParameters,
        (E_NO_KEY, _) => CredentialError::NoIssuerKey,
        (E_BAD_ARGS, _) => CredentialError::MissingData,
        (_, Op::Create) => CredentialError::NoSystemParameters,
        (_, Op::Issue) => CredentialError::CredentialIssuance,
        (_, Op::Verify) | (_, Op::VerifyIssuance) => CredentialError::VerificationFailure,
        (_, Op::Show) => CredentialError::MissingData,
    }
}

/// The crate multiplies by secrets in constant time (dalek's `*` and `multiscalar_mul`: src/amacs.rs:267-270,
/// src/nizk/presentation.rs:162-184, zkp's Prover).  The engine's kernels have no secret-dependent branches in any mode; the mode
/// says where no memory ADDRESS may depend on a secret scalar either (every table entry is read and the wanted one selected), at
/// the cost INTEGRATION.md section 2b quotes.  A new engine runs `ProverSide`: issue and show are covered, verification runs the
/// fast tables; `Everywhere` adds the issuer key's terms of `verify`; `Nowhere` is for a device of the engine's own.
#[derive(Clone, Copy)]
pub enum SecretAddressing { Nowhere = 0, Everywhere = 1, ProverSide = 2 }

/// Applied to the one context or to every member of the group.
fn set_secret_addressing(ctx: *mut c_void, group: *mut c_void, mode: SecretAddressing) -> Result<(), CredentialError> {
    let mut rc = 0;
    unsafe {
        if group.is_null() { rc |= afx_ctx_set_secret_independent_addressing(ctx, mode as i32); }
        else { for i in 0..afx_group_size(group) { rc |= afx_ctx_set_secret_independent_addressing(afx_group_member(group, i), mode as i32); } }
    }
    if rc != 0 { Err(engine_error(rc, Op::Create)) } else { Ok(()) }
}

/// `Issuer` with its parameters, tables and key resident on one MI355X (`ctx`) or on several (`group`: the batch is split
/// contiguously over the devices inside the library, one host thread per device, no collective).
pub struct GpuIssuer { ctx: *mut c_void, group: *mut c_void, n: usize, params: Vec<u8>, issuer_params: [u8; 64], last_rc: AtomicI32 }

/// The user's side (no issuer key): `AnonymousCredential::show` and `CredentialIssuance::verify`.
pub struct GpuUser { ctx: *mut c_void, group: *mut c_void, n: usize, params: Vec<u8>, issuer_params: [u8; 64], last_rc: AtomicI32 }

// The handles are plain pointers into the library, which serialises the calls on a context with a mutex of its own
// (include/aeonflux_gpu.h: "concurrent calls on one ctx are serialised"); a group's members likewise.
unsafe impl Send for GpuIssuer {}
unsafe impl Sync for GpuIssuer {}
unsafe impl Send for GpuUser {}
unsafe impl Sync for GpuUser {}

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
fn put(dst: &mut [u8], row: usize, count: usize, item: usize, src: &[u8; 32]) {
    dst[32 * (row * count + item)..32 * (row * count + item) + 32].copy_from_slice(src);
}
// outputs of the engine are canonical scalars / valid encodings by construction; anything else is reported, not trusted
fn sc(col: &[u8], row: usize, count: usize, item: usize) -> Result<Scalar, CredentialError> {
    Scalar::from_canonical_bytes(cell(col, row, count, item)).ok_or(CredentialError::ScalarFormatError)
}
fn pt(col: &[u8], row: usize, count: usize, item: usize) -> Result<RistrettoPoint, CredentialError> {
    CompressedRistretto(cell(col, row, count, item)).decompress().ok_or(CredentialError::PointDecompressionError)
}

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

/// The attribute layout of a request / credential as bytes: two of them may share a GPU batch iff their keys are equal.
/// `None`: more attributes than the ABI carries (AFX_MAX_ATTRIBUTES).
fn layout_key(attributes: &Vec<Attribute>) -> Option<Vec<u8>> {
    if attributes.len() > AFX_MAX_ATTRIBUTES { return None; }
    Some(attributes.iter().map(|a| attribute_cells(a).0).collect())
}
fn attributes_soa(key: &[u8], values: *const u8) -> AfxAttributesSoa {
    let mut soa = AfxAttributesSoa { n_attributes: key.len() as u32, kinds: [0; AFX_MAX_ATTRIBUTES], values };
    soa.kinds[..key.len()].copy_from_slice(key);
    soa
}
/// Members of a BTreeMap of index lists -> (key, members) pairs in a fixed order.
fn grouped(map: BTreeMap<Vec<u8>, Vec<usize>>) -> Vec<(Vec<u8>, Vec<usize>)> { map.into_iter().collect() }
fn positions_of(members: &[usize]) -> Vec<u64> { members.iter().map(|i| *i as u64).collect() }

/// Column-major staging of a batch of presentations: every field one `[count][32]` array, repeated fields `[k][count][32]`.
struct Columns {
    challenge: Vec<u8>, responses: Vec<u8>, c_x_0: Vec<u8>, c_x_1: Vec<u8>, c_v: Vec<u8>, c_y: Vec<u8>, attr_values: Vec<u8>,
    enc: Vec<[Vec<u8>; 9]>,
}

impl GpuIssuer {
    pub fn new(issuer: &Issuer, device: i32) -> Result<GpuIssuer, CredentialError> {
        let sp = issuer.system_parameters.to_bytes();                 // src/parameters.rs:155-184
        let key = Wiped(issuer.amacs_key.to_bytes());                 // src/amacs.rs:110-125; the crate zeroizes the key (src/amacs.rs:64-82), so does this copy
        let ip = issuer_params_bytes(&issuer.issuer_parameters);
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { afx_ctx_create(&mut ctx, device, sp.as_ptr(), sp.len(), key.0.as_ptr(), key.0.len(), ip.as_ptr()) };
        if rc != 0 { return Err(engine_error(rc, Op::Create)); }
        Ok(GpuIssuer { ctx, group: core::ptr::null_mut(), n: issuer.system_parameters.NUMBER_OF_ATTRIBUTES as usize, params: sp, issuer_params: ip,
                       last_rc: AtomicI32::new(0) })
    }

    /// The same issuer on several GPUs of one node: every batch call below is split contiguously over `devices`.
    pub fn new_multi(issuer: &Issuer, devices: &[i32]) -> Result<GpuIssuer, CredentialError> {
        let sp = issuer.system_parameters.to_bytes();
        let key = Wiped(issuer.amacs_key.to_bytes());
        let ip = issuer_params_bytes(&issuer.issuer_parameters);
        let mut group = core::ptr::null_mut();
        let rc = unsafe { afx_group_create(&mut group, devices.as_ptr(), devices.len() as u32, sp.as_ptr(), sp.len(), key.0.as_ptr(), key.0.len(), ip.as_ptr()) };
        if rc != 0 { return Err(engine_error(rc, Op::Create)); }
        Ok(GpuIssuer { ctx: core::ptr::null_mut(), group, n: issuer.system_parameters.NUMBER_OF_ATTRIBUTES as usize, params: sp, issuer_params: ip,
                       last_rc: AtomicI32::new(0) })
    }

    /// Where secrets are kept out of table addresses (`SecretAddressing`; the default covers `issue`, `Everywhere` adds the issuer
    /// key's terms of `verify`).
    pub fn set_secret_addressing(&self, mode: SecretAddressing) -> Result<(), CredentialError> { set_secret_addressing(self.ctx, self.group, mode) }

    /// The engine's return code (AFX_E_*, 0 = none) of the most recent call that failed as a whole.
    pub fn last_engine_code(&self) -> i32 { self.last_rc.load(Ordering::Relaxed) }

    /// Was this engine built from `issuer`'s parameters?  (`IssuerParameters` = (C_W, I) commit to the key, src/parameters.rs:349-362.)
    pub fn serves(&self, issuer: &Issuer) -> bool {
        self.issuer_params == issuer_params_bytes(&issuer.issuer_parameters) && self.params == issuer.system_parameters.to_bytes()
    }

    /// `Issuer::issue` (src/issuer.rs:111-118), on the GPU: one request through the engine's latency plan.
    pub fn issue<C>(
        &self,
        request: CredentialRequest,
        csprng: &mut C,
    ) -> Result<CredentialIssuance, CredentialError>
    where
        C: CryptoRng + RngCore,
    {
        self.issue_batch(vec![request], csprng).pop().unwrap_or(Err(CredentialError::CredentialIssuance))
    }

    /// `Issuer::verify` (src/issuer.rs:141-145), on the GPU: one presentation through the engine's latency plan.
    pub fn verify(
        &self,
        presentation: &ProofOfValidCredential,
    ) -> Result<(), CredentialError>
    {
        self.verify_batch(core::slice::from_ref(presentation)).pop().unwrap_or(Err(CredentialError::VerificationFailure))
    }

    /// Batch `Issuer::issue` (src/issuer.rs:111-124) over requests of ANY attribute layouts: consumes the requests like the
    /// reference does and returns one `Result` per request, in order.  The requests are grouped by layout (`layout_key`: the kind
    /// of every attribute, src/amacs.rs:168-179) and all groups go to the engine in one call.
    pub fn issue_batch<C: CryptoRng + RngCore>(&self, requests: Vec<CredentialRequest>, csprng: &mut C)
        -> Vec<Result<CredentialIssuance, CredentialError>>
    {
        let count = requests.len();
        if count == 0 { return Vec::new(); }
        // the reference's draws, in its order and the caller's (see the header of this file)
        let (mut t_wide, mut u_wide, mut seed) = (Wiped::new(64 * count), Wiped::new(64 * count), Wiped::new(32 * count));
        for i in 0..count {
            csprng.fill_bytes(&mut t_wide.0[64 * i..64 * i + 64]);    // Scalar::random          (src/amacs.rs:289)
            csprng.fill_bytes(&mut u_wide.0[64 * i..64 * i + 64]);    // RistrettoPoint::random  (src/amacs.rs:290)
        }
        csprng.fill_bytes(&mut seed.0);                               // in place of zkp prove_compact's thread_rng() draw, 32 B per proof
        let mut out: Vec<Option<Result<CredentialIssuance, CredentialError>>> = (0..count).map(|_| None).collect();
        let mut by_layout: BTreeMap<Vec<u8>, Vec<usize>> = BTreeMap::new();
        for (i, r) in requests.iter().enumerate() {
            match layout_key(&r.attributes) {
                Some(k) => by_layout.entry(k).or_insert_with(Vec::new).push(i),
                // more attributes than any engine context has (n <= AFX_MAX_ATTRIBUTES): the wrong number for this issuer (src/amacs.rs:285-287)
                None => out[i] = Some(Err(CredentialError::MacCreation)),
            }
        }
        let nr = self.n + 5;                                          // w, w', x_0, x_1, y_0..y_{n-1}, "1" (src/nizk/issuance.rs:52-68)
        struct Stage { key: Vec<u8>, members: Vec<usize>, positions: Vec<u64>, values: Wiped, t_wide: Wiped, u_wide: Wiped, seed: Wiped,
                       t: Vec<u8>, u: Vec<u8>, v: Vec<u8>, ch: Vec<u8>, rs: Vec<u8> }
        let mut stages: Vec<Stage> = grouped(by_layout).into_iter().map(|(key, members)| {
            let (m, na) = (members.len(), key.len());
            let mut st = Stage { positions: positions_of(&members), values: Wiped::new(32 * na * m), t_wide: Wiped::new(64 * m), u_wide: Wiped::new(64 * m),
                                 seed: Wiped::new(32 * m), t: vec![0u8; 32 * m], u: vec![0u8; 32 * m], v: vec![0u8; 32 * m], ch: vec![0u8; 32 * m],
                                 rs: vec![0u8; 32 * nr * m], key, members };
            for (j, i) in st.members.iter().enumerate() {
                for (k, a) in requests[*i].attributes.iter().enumerate() { put(&mut st.values.0, k, m, j, &attribute_cells(a).1); }
                st.t_wide.0[64 * j..64 * j + 64].copy_from_slice(&t_wide.0[64 * i..64 * i + 64]);
                st.u_wide.0[64 * j..64 * j + 64].copy_from_slice(&u_wide.0[64 * i..64 * i + 64]);
                st.seed.0[32 * j..32 * j + 32].copy_from_slice(&seed.0[32 * i..32 * i + 32]);
            }
            st
        }).collect();
        let groups: Vec<AfxIssueGroup> = stages.iter_mut().map(|st| AfxIssueGroup {
            requests: attributes_soa(&st.key, st.values.0.as_ptr()),
            rnd: AfxIssueRandomness { t_wide: st.t_wide.0.as_ptr(), U_wide: st.u_wide.0.as_ptr(), rng_seed: st.seed.0.as_ptr() },
            out: AfxIssuanceSoa { t: st.t.as_mut_ptr(), U: st.u.as_mut_ptr(), V: st.v.as_mut_ptr(), challenge: st.ch.as_mut_ptr(), responses: st.rs.as_mut_ptr() },
            count: st.members.len(),
            positions: st.positions.as_ptr(),
        }).collect();
        let mut status = vec![ST_MAC_CREATION; count];
        let rc = unsafe {
            if self.group.is_null() { afx_issue_mixed(self.ctx, groups.as_ptr(), groups.len(), status.as_mut_ptr(), count) }
            else { afx_group_issue_mixed(self.group, groups.as_ptr(), groups.len(), status.as_mut_ptr(), count) }
        };
        if rc != 0 {
            self.last_rc.store(rc, Ordering::Relaxed);
            return (0..count).map(|_| Err(engine_error(rc, Op::Issue))).collect();
        }
        let mut requests: Vec<Option<CredentialRequest>> = requests.into_iter().map(Some).collect();
        for st in stages.iter() {
            let m = st.members.len();
            for (j, i) in st.members.iter().enumerate() {
                out[*i] = Some(match status[*i] {
                    ST_OK => (|| -> Result<CredentialIssuance, CredentialError> {
                        let amac = Amac { t: sc(&st.t, 0, m, j)?, U: pt(&st.u, 0, m, j)?, V: pt(&st.v, 0, m, j)? };
                        let responses = (0..nr).map(|k| sc(&st.rs, k, m, j)).collect::<Result<Vec<Scalar>, CredentialError>>()?;
                        let proof = CompactProof { challenge: sc(&st.ch, 0, m, j)?, responses };
                        let request = requests[*i].take().ok_or(CredentialError::MissingData)?;
                        Ok(CredentialIssuance { proof: ProofOfIssuance(proof), credential: AnonymousCredential { amac, attributes: request.attributes } })
                    })(),
                    ST_MAC_CREATION => Err(CredentialError::MacCreation),          // amacs.rs:285-287 -> errors.rs:141-142
                    _ => Err(CredentialError::CredentialIssuance),                 // a status this operation does not have: reported, never trusted
                });
            }
        }
        out.into_iter().map(|r| r.unwrap_or(Err(CredentialError::CredentialIssuance))).collect()
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
        let mut by_shape: BTreeMap<Vec<u8>, Vec<usize>> = BTreeMap::new();
        for (i, p) in batch.iter().enumerate() {
            if let Some(key) = shape_key(p) { by_shape.entry(key).or_insert_with(Vec::new).push(i); }   // else: stays a failure
        }
        // the staged columns, position lists and struct arrays of every group live until the call returns
        let staged: Vec<(AfxShape, Columns, Vec<u64>)> = by_shape.values().map(|members| {
            let items: Vec<&ProofOfValidCredential> = members.iter().map(|i| &batch[*i]).collect();
            let (shape, cols) = marshal(&items);
            (shape, cols, positions_of(members))
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
        if rc != 0 {
            self.last_rc.store(rc, Ordering::Relaxed);
            return (0..total).map(|_| Err(engine_error(rc, Op::Verify))).collect();   // fails closed
        }
        status.iter().map(|s| if *s == ST_OK { Ok(()) } else { Err(CredentialError::VerificationFailure) }).collect()
    }
}

impl Drop for GpuIssuer {
    // wipes every key copy, host and device (Zeroize + Drop of amacs::SecretKey, src/amacs.rs:64-82)
    fn drop(&mut self) { unsafe { if self.group.is_null() { afx_ctx_destroy(self.ctx) } else { afx_group_destroy(self.group) } } }
}

/// An `AnonymousCredential` paired with the engine that shows it: `self` of `show` is the credential, as in the crate.
pub struct GpuCredential<'a> { user: &'a GpuUser, credential: &'a AnonymousCredential }

/// A `CredentialIssuance` paired with the engine that checks it: `verify` consumes it, as in the crate.
pub struct GpuIssuance<'a> { user: &'a GpuUser, issuance: CredentialIssuance }

impl<'a> GpuCredential<'a> {
    /// `AnonymousCredential::show` (src/credential.rs:37-43), on the GPU: one credential through the engine's latency plan.
    /// The parameters must be the ones the engine was built for (`NoSystemParameters` otherwise).
    pub fn show(
        &self,
        system_parameters: &SystemParameters,
        issuer_parameters: &IssuerParameters,
        keypair: Option<&SymmetricKeypair>,
        mut csprng: impl CryptoRng + RngCore,
    ) -> Result<ProofOfValidCredential, CredentialError>
    {
        if !self.user.serves(system_parameters, issuer_parameters) { return Err(CredentialError::NoSystemParameters); }
        self.user.show_batch(core::slice::from_ref(self.credential), keypair.map(core::slice::from_ref), &mut csprng)
            .pop().unwrap_or(Err(CredentialError::MissingData))
    }
}

impl<'a> GpuIssuance<'a> {
    /// `CredentialIssuance::verify` (src/issuer.rs:48-52), on the GPU; moves the credential out on success.
    pub fn verify(
        self,
        system_parameters: &SystemParameters,
        issuer_parameters: &IssuerParameters,
    ) -> Result<AnonymousCredential, CredentialError>
    {
        if !self.user.serves(system_parameters, issuer_parameters) { return Err(CredentialError::NoSystemParameters); }
        self.user.verify_issuance_batch(vec![self.issuance]).pop().unwrap_or(Err(CredentialError::VerificationFailure))
    }
}

impl GpuUser {
    pub fn new(system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters, device: i32) -> Result<GpuUser, CredentialError> {
        let sp = system_parameters.to_bytes();
        let ip = issuer_params_bytes(issuer_parameters);
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { afx_ctx_create(&mut ctx, device, sp.as_ptr(), sp.len(), core::ptr::null(), 0, ip.as_ptr()) };
        if rc != 0 { return Err(engine_error(rc, Op::Create)); }
        Ok(GpuUser { ctx, group: core::ptr::null_mut(), n: system_parameters.NUMBER_OF_ATTRIBUTES as usize, params: sp, issuer_params: ip,
                     last_rc: AtomicI32::new(0) })
    }

    /// The same on several GPUs of the node: batches are split contiguously over `devices` inside the library.
    pub fn new_multi_user(system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters, devices: &[i32]) -> Result<GpuUser, CredentialError> {
        let sp = system_parameters.to_bytes();
        let ip = issuer_params_bytes(issuer_parameters);
        let mut group = core::ptr::null_mut();
        let rc = unsafe { afx_group_create(&mut group, devices.as_ptr(), devices.len() as u32, sp.as_ptr(), sp.len(), core::ptr::null(), 0, ip.as_ptr()) };
        if rc != 0 { return Err(engine_error(rc, Op::Create)); }
        Ok(GpuUser { ctx: core::ptr::null_mut(), group, n: system_parameters.NUMBER_OF_ATTRIBUTES as usize, params: sp, issuer_params: ip,
                     last_rc: AtomicI32::new(0) })
    }

    /// Where secrets are kept out of table addresses (`SecretAddressing`; the default covers every scalar of `show`: blindings, the
    /// credential's `t`, the symmetric key).
    pub fn set_secret_addressing(&self, mode: SecretAddressing) -> Result<(), CredentialError> { set_secret_addressing(self.ctx, self.group, mode) }

    /// The engine's return code (AFX_E_*, 0 = none) of the most recent call that failed as a whole.
    pub fn last_engine_code(&self) -> i32 { self.last_rc.load(Ordering::Relaxed) }

    /// Was this engine built for these parameters?
    pub fn serves(&self, system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters) -> bool {
        self.issuer_params == issuer_params_bytes(issuer_parameters) && self.params == system_parameters.to_bytes()
    }

    /// `credential` with this engine behind its `show` (the crate's signature: `GpuCredential::show`).
    pub fn credential<'a>(&'a self, credential: &'a AnonymousCredential) -> GpuCredential<'a> { GpuCredential { user: self, credential } }

    /// `issuance` with this engine behind its `verify` (the crate's signature: `GpuIssuance::verify`).
    pub fn issuance<'a>(&'a self, issuance: CredentialIssuance) -> GpuIssuance<'a> { GpuIssuance { user: self, issuance } }

    /// Batch `AnonymousCredential::show` (src/credential.rs:37-46 -> src/nizk/presentation.rs:139-321) over credentials in ANY
    /// state of their hide_attribute / reveal_attribute calls (src/credential.rs:53-97): grouped by layout, one engine call,
    /// results in the order given.  One keypair per credential (or `None`: a credential with a SecretPoint attribute then yields
    /// `NoSymmetricKey`, presentation.rs:150-157).
    pub fn show_batch<C: CryptoRng + RngCore>(&self, creds: &[AnonymousCredential], keypairs: Option<&[SymmetricKeypair]>, csprng: &mut C)
        -> Vec<Result<ProofOfValidCredential, CredentialError>>
    {
        let count = creds.len();
        if count == 0 { return Vec::new(); }
        if let Some(kps) = keypairs { if kps.len() != count { return (0..count).map(|_| Err(CredentialError::MissingData)).collect(); } }
        let mut out: Vec<Option<Result<ProofOfValidCredential, CredentialError>>> = (0..count).map(|_| None).collect();
        let mut by_layout: BTreeMap<Vec<u8>, Vec<usize>> = BTreeMap::new();
        let mut nsp_of = vec![0usize; count];
        for (i, c) in creds.iter().enumerate() {
            // more attribute positions than the parameters have generators for: the reference indexes G_y / G_m out of range (presentation.rs:169-180)
            let key = layout_key(&c.attributes).filter(|k| k.len() >= 1 && k.len() <= self.n);
            match key {
                Some(k) => {
                    nsp_of[i] = k.iter().filter(|x| **x == ATTR_SECRET_POINT).count();
                    by_layout.entry(k).or_insert_with(Vec::new).push(i);
                }
                None => out[i] = Some(Err(CredentialError::WrongNumberOfAttributes)),
            }
        }
        // the reference's csprng draws in its order, then the proofs' seeds (see the header of this file)
        let enc_total: usize = nsp_of.iter().sum();
        let (mut z_wide, mut seed, mut enc_seed) = (Wiped::new(64 * count), Wiped::new(32 * count), Wiped::new(32 * enc_total.max(1)));
        for i in 0..count { csprng.fill_bytes(&mut z_wide.0[64 * i..64 * i + 64]); }   // Scalar::random (presentation.rs:162)
        csprng.fill_bytes(&mut seed.0);                                               // in place of thread_rng() in the presentation proof's prove_compact (:284)
        if enc_total != 0 { csprng.fill_bytes(&mut enc_seed.0); }                       // ... and in each ProofOfEncryption's (:301): credential by credential, attribute order
        let mut enc_at = vec![0usize; count];                                          // where credential i's enc seeds start
        for i in 1..count { enc_at[i] = enc_at[i - 1] + nsp_of[i - 1]; }
        struct Stage { key: Vec<u8>, members: Vec<usize>, positions: Vec<u64>, hs: usize, nsp: usize,
                       values: Wiped, m2: Wiped, m3: Wiped, t: Vec<u8>, u: Vec<u8>, v: Vec<u8>, ka: Wiped, ka0: Wiped, ka1: Wiped, kpk: Vec<u8>,
                       z_wide: Wiped, seed: Wiped, enc_seeds: Wiped,
                       o_ch: Vec<u8>, o_rs: Vec<u8>, o_x0: Vec<u8>, o_x1: Vec<u8>, o_cv: Vec<u8>, o_cy: Vec<u8>, o_av: Vec<u8>, enc_cols: Vec<[Vec<u8>; 9]> }
        let mut stages: Vec<Stage> = grouped(by_layout).into_iter().map(|(key, members)| {
            let (m, na) = (members.len(), key.len());
            let hs = key.iter().filter(|x| **x == ATTR_SECRET_SCALAR).count();
            let nsp = key.iter().filter(|x| **x == ATTR_SECRET_POINT).count();
            let col = |k: usize| vec![0u8; 32 * k * m];
            // hidden attribute values are secrets of the user (amacs::Attribute zeroizes them, src/amacs.rs:184-200)
            let mut st = Stage { positions: positions_of(&members), hs, nsp, values: Wiped::new(32 * na * m), m2: Wiped::new(32 * na * m), m3: Wiped::new(32 * na * m),
                                 t: col(1), u: col(1), v: col(1), ka: Wiped::new(32 * m), ka0: Wiped::new(32 * m), ka1: Wiped::new(32 * m), kpk: col(1),
                                 z_wide: Wiped::new(64 * m), seed: Wiped::new(32 * m), enc_seeds: Wiped::new(32 * m * nsp.max(1)),
                                 o_ch: col(1), o_rs: col(3 + hs), o_x0: col(1), o_x1: col(1), o_cv: col(1), o_cy: col(na), o_av: col(na),
                                 enc_cols: (0..nsp).map(|_| [col(1), col(6), col(1), col(1), col(1), col(1), col(1), col(1), col(1)]).collect(),
                                 key, members };
            for (j, i) in st.members.iter().enumerate() {
                let c = &creds[*i];
                for (k, a) in c.attributes.iter().enumerate() {
                    let (_, val, plain) = attribute_cells(a);
                    put(&mut st.values.0, k, m, j, &val);
                    if let Some((p2, s3)) = plain { put(&mut st.m2.0, k, m, j, &p2); put(&mut st.m3.0, k, m, j, &s3); }
                }
                put(&mut st.t, 0, m, j, c.amac.t.as_bytes());
                put(&mut st.u, 0, m, j, c.amac.U.compress().as_bytes());
                put(&mut st.v, 0, m, j, c.amac.V.compress().as_bytes());
                if let Some(kps) = keypairs {                                          // symmetric::Keypair, src/symmetric.rs:52-81
                    put(&mut st.ka.0, 0, m, j, kps[*i].secret.a.as_bytes());
                    put(&mut st.ka0.0, 0, m, j, kps[*i].secret.a0.as_bytes());
                    put(&mut st.ka1.0, 0, m, j, kps[*i].secret.a1.as_bytes());
                    put(&mut st.kpk, 0, m, j, kps[*i].public.pk.compress().as_bytes());
                }
                st.z_wide.0[64 * j..64 * j + 64].copy_from_slice(&z_wide.0[64 * i..64 * i + 64]);
                st.seed.0[32 * j..32 * j + 32].copy_from_slice(&seed.0[32 * i..32 * i + 32]);
                for e in 0..nsp {                                                      // engine layout: [secret point][credential]
                    let from = 32 * (enc_at[*i] + e);
                    st.enc_seeds.0[32 * (e * m + j)..32 * (e * m + j) + 32].copy_from_slice(&enc_seed.0[from..from + 32]);
                }
            }
            st
        }).collect();
        let kp_soas: Vec<AfxKeypairsSoa> = stages.iter().map(|st| AfxKeypairsSoa { a: st.ka.0.as_ptr(), a0: st.ka0.0.as_ptr(), a1: st.ka1.0.as_ptr(), pk: st.kpk.as_ptr() }).collect();
        let enc_outs: Vec<Vec<AfxEncProofOut>> = stages.iter_mut().map(|st| st.enc_cols.iter_mut().map(|e| AfxEncProofOut {
            challenge: e[0].as_mut_ptr(), responses: e[1].as_mut_ptr(), pk: e[2].as_mut_ptr(), E1: e[3].as_mut_ptr(), E2: e[4].as_mut_ptr(),
            C_y_1: e[5].as_mut_ptr(), C_y_2: e[6].as_mut_ptr(), C_y_3: e[7].as_mut_ptr(), C_y_2p: e[8].as_mut_ptr() }).collect()).collect();
        let no_shape = AfxShape { n_attributes: 0, kinds: [0; 32], n_responses: 0, n_hidden_scalars: 0, hidden_scalar_indices: [0; 32], n_enc_proofs: 0, enc_indices: [0; 32] };
        let mut groups: Vec<AfxShowGroup> = stages.iter_mut().enumerate().map(|(g, st)| {
            let mut cs = AfxCredentialsSoa { n_attributes: st.key.len() as u32, kinds: [0; AFX_MAX_ATTRIBUTES], values: st.values.0.as_ptr(), M2: st.m2.0.as_ptr(),
                                             m3: st.m3.0.as_ptr(), t: st.t.as_ptr(), U: st.u.as_ptr(), V: st.v.as_ptr() };
            cs.kinds[..st.key.len()].copy_from_slice(&st.key);
            AfxShowGroup {
                creds: cs,
                keypairs: if keypairs.is_some() { &kp_soas[g] as *const AfxKeypairsSoa } else { core::ptr::null() },
                rnd: AfxShowRandomness { z_wide: st.z_wide.0.as_ptr(), rng_seed: st.seed.0.as_ptr(), enc_seeds: st.enc_seeds.0.as_ptr() },
                out: AfxPresentationOut { challenge: st.o_ch.as_mut_ptr(), responses: st.o_rs.as_mut_ptr(), C_x_0: st.o_x0.as_mut_ptr(), C_x_1: st.o_x1.as_mut_ptr(),
                                          C_V: st.o_cv.as_mut_ptr(), C_y: st.o_cy.as_mut_ptr(), attr_values: st.o_av.as_mut_ptr(), enc: enc_outs[g].as_ptr() },
                shape_out: no_shape,
                count: st.members.len(),
                positions: st.positions.as_ptr(),
            }
        }).collect();
        let mut status = vec![ST_VERIFICATION_FAILURE; count];
        let rc = unsafe {
            if self.group.is_null() { afx_show_mixed(self.ctx, groups.as_mut_ptr(), groups.len(), status.as_mut_ptr(), count) }
            else { afx_group_show_mixed(self.group, groups.as_mut_ptr(), groups.len(), status.as_mut_ptr(), count) }
        };
        if rc != 0 {
            self.last_rc.store(rc, Ordering::Relaxed);
            return (0..count).map(|_| Err(engine_error(rc, Op::Show))).collect();
        }
        // rebuild ProofOfValidCredential (src/nizk/presentation.rs:118-127) per item
        for (g, st) in stages.iter().enumerate() {
            let (m, na, shape) = (st.members.len(), st.key.len(), &groups[g].shape_out);
            for (j, i) in st.members.iter().enumerate() {
                out[*i] = Some(match status[*i] {
                    ST_OK => (|| -> Result<ProofOfValidCredential, CredentialError> {
                        let proof = CompactProof { challenge: sc(&st.o_ch, 0, m, j)?,
                                                   responses: (0..3 + st.hs).map(|k| sc(&st.o_rs, k, m, j)).collect::<Result<Vec<Scalar>, CredentialError>>()? };
                        let encrypted_attributes = (0..na).map(|k| -> Result<EncryptedAttribute, CredentialError> { Ok(match shape.kinds[k] {
                            0 => EncryptedAttribute::PublicScalar(sc(&st.o_av, k, m, j)?),
                            1 => EncryptedAttribute::SecretScalar,
                            2 => EncryptedAttribute::PublicPoint(pt(&st.o_av, k, m, j)?),
                            _ => EncryptedAttribute::SecretPoint,
                        }) }).collect::<Result<Vec<EncryptedAttribute>, CredentialError>>()?;
                        let proofs_of_encryption = (0..st.nsp).map(|e| -> Result<(u16, ProofOfEncryption), CredentialError> {
                            let c = &st.enc_cols[e];
                            let index = shape.enc_indices[e];
                            Ok((index, ProofOfEncryption {
                                proof: CompactProof { challenge: sc(&c[0], 0, m, j)?,
                                                      responses: (0..6).map(|k| sc(&c[1], k, m, j)).collect::<Result<Vec<Scalar>, CredentialError>>()? },
                                public_key: SymmetricPublicKey { pk: pt(&c[2], 0, m, j)? },
                                ciphertext: Ciphertext { E1: pt(&c[3], 0, m, j)?, E2: pt(&c[4], 0, m, j)? },
                                index,
                                C_y_1: pt(&c[5], 0, m, j)?, C_y_2: pt(&c[6], 0, m, j)?, C_y_3: pt(&c[7], 0, m, j)?, C_y_2_prime: pt(&c[8], 0, m, j)?,
                            }))
                        }).collect::<Result<Vec<(u16, ProofOfEncryption)>, CredentialError>>()?;
                        Ok(ProofOfValidCredential {
                            proof, proofs_of_encryption, encrypted_attributes,
                            hidden_scalar_indices: shape.hidden_scalar_indices[..shape.n_hidden_scalars as usize].to_vec(),
                            C_x_0: pt(&st.o_x0, 0, m, j)?, C_x_1: pt(&st.o_x1, 0, m, j)?, C_V: pt(&st.o_cv, 0, m, j)?,
                            C_y: (0..na).map(|k| pt(&st.o_cy, k, m, j)).collect::<Result<Vec<RistrettoPoint>, CredentialError>>()?,
                        })
                    })(),
                    ST_NO_SYMMETRIC_KEY => Err(CredentialError::NoSymmetricKey),
                    _ => Err(CredentialError::MissingData),                        // a status this operation does not have: no presentation is made up
                });
            }
        }
        out.into_iter().map(|r| r.unwrap_or(Err(CredentialError::MissingData))).collect()
    }

    /// Batch `CredentialIssuance::verify` (src/issuer.rs:48-57) over issuances of ANY attribute layouts: consumes the issuances
    /// and moves each credential out on success.
    pub fn verify_issuance_batch(&self, issuances: Vec<CredentialIssuance>) -> Vec<Result<AnonymousCredential, CredentialError>> {
        let count = issuances.len();
        if count == 0 { return Vec::new(); }
        // layout = attribute kinds + the proof's response count (zkp rejects a wrong count; the engine fails such a group whole)
        let mut by_layout: BTreeMap<Vec<u8>, Vec<usize>> = BTreeMap::new();
        for (i, iss) in issuances.iter().enumerate() {
            let nr = iss.proof.0.responses.len();
            if let Some(mut k) = layout_key(&iss.credential.attributes) {
                if nr <= AFX_MAX_ATTRIBUTES + 5 {
                    k.extend_from_slice(&(nr as u32).to_le_bytes());
                    by_layout.entry(k).or_insert_with(Vec::new).push(i);
                }
            }                                                                          // else: stays a failure
        }
        struct Stage { key: Vec<u8>, nr: usize, members: Vec<usize>, positions: Vec<u64>, values: Vec<u8>, t: Vec<u8>, u: Vec<u8>, v: Vec<u8>, ch: Vec<u8>, rs: Vec<u8> }
        let mut stages: Vec<Stage> = grouped(by_layout).into_iter().map(|(mut key, members)| {
            let mut nrb = [0u8; 4];
            nrb.copy_from_slice(&key[key.len() - 4..]);
            key.truncate(key.len() - 4);
            let (m, na, nr) = (members.len(), key.len(), u32::from_le_bytes(nrb) as usize);
            let col = |k: usize| vec![0u8; 32 * k * m];
            let mut st = Stage { positions: positions_of(&members), nr, values: col(na), t: col(1), u: col(1), v: col(1), ch: col(1), rs: col(nr), key, members };
            for (j, i) in st.members.iter().enumerate() {
                let iss = &issuances[*i];
                for (k, a) in iss.credential.attributes.iter().enumerate() { put(&mut st.values, k, m, j, &attribute_cells(a).1); }
                put(&mut st.t, 0, m, j, iss.credential.amac.t.as_bytes());
                put(&mut st.u, 0, m, j, iss.credential.amac.U.compress().as_bytes());
                put(&mut st.v, 0, m, j, iss.credential.amac.V.compress().as_bytes());
                put(&mut st.ch, 0, m, j, iss.proof.0.challenge.as_bytes());
                for (k, r) in iss.proof.0.responses.iter().enumerate() { put(&mut st.rs, k, m, j, r.as_bytes()); }
            }
            st
        }).collect();
        let groups: Vec<AfxIssuanceGroup> = stages.iter_mut().map(|st| AfxIssuanceGroup {
            attrs: attributes_soa(&st.key, st.values.as_ptr()),
            issuances: AfxIssuanceSoa { t: st.t.as_mut_ptr(), U: st.u.as_mut_ptr(), V: st.v.as_mut_ptr(), challenge: st.ch.as_mut_ptr(), responses: st.rs.as_mut_ptr() },
            n_responses: st.nr as u32,
            count: st.members.len(),
            positions: st.positions.as_ptr(),
        }).collect();
        let mut status = vec![ST_VERIFICATION_FAILURE; count];
        let rc = unsafe {
            if self.group.is_null() { afx_verify_issuances_mixed(self.ctx, groups.as_ptr(), groups.len(), status.as_mut_ptr(), count) }
            else { afx_group_verify_issuances_mixed(self.group, groups.as_ptr(), groups.len(), status.as_mut_ptr(), count) }
        };
        if rc != 0 {
            self.last_rc.store(rc, Ordering::Relaxed);
            return (0..count).map(|_| Err(engine_error(rc, Op::VerifyIssuance))).collect();   // fails closed
        }
        issuances.into_iter().enumerate().map(|(i, iss)| if status[i] == ST_OK { Ok(iss.credential) } else { Err(CredentialError::VerificationFailure) }).collect()
    }
}

impl Drop for GpuUser {
    fn drop(&mut self) { unsafe { if self.group.is_null() { afx_ctx_destroy(self.ctx) } else { afx_group_destroy(self.group) } } }
}

// ---- process-wide engines, for the three-line delegation patches of INTEGRATION.md section 1 --------------------------------
// `install_issuer` / `install_user` hand an engine over for the life of the process (it is leaked on purpose: references to it
// are `'static`, and the device copy of the key is wiped when the process ends with the context's memory).  The crate's own
// `Issuer::issue` / `Issuer::verify` / `AnonymousCredential::show` / `CredentialIssuance::verify` then ask `issuer_engine` /
// `user_engine` whether an engine for THEIR parameters is installed and delegate to it; otherwise they run as before.
static ISSUER_ENGINE: AtomicPtr<GpuIssuer> = AtomicPtr::new(core::ptr::null_mut());
static USER_ENGINE: AtomicPtr<GpuUser> = AtomicPtr::new(core::ptr::null_mut());

pub fn install_issuer(engine: GpuIssuer) -> &'static GpuIssuer {
    let p = Box::into_raw(Box::new(engine));
    ISSUER_ENGINE.store(p, Ordering::Release);
    unsafe { &*p }
}
pub fn install_user(engine: GpuUser) -> &'static GpuUser {
    let p = Box::into_raw(Box::new(engine));
    USER_ENGINE.store(p, Ordering::Release);
    unsafe { &*p }
}
/// The installed engine, if it was built from this issuer's parameters.
pub fn issuer_engine(issuer: &Issuer) -> Option<&'static GpuIssuer> {
    let p = ISSUER_ENGINE.load(Ordering::Acquire);
    if p.is_null() { return None; }
    let e = unsafe { &*p };
    if e.serves(issuer) { Some(e) } else { None }
}
/// The installed user-side engine, if it was built for these parameters.
pub fn user_engine(system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters) -> Option<&'static GpuUser> {
    let p = USER_ENGINE.load(Ordering::Acquire);
    if p.is_null() { return None; }
    let e = unsafe { &*p };
    if e.serves(system_parameters, issuer_parameters) { Some(e) } else { None }
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

/// Presentations of ONE shape (equal `shape_key`s: `verify_batch` groups by it and nothing else calls this) -> shape + columns.
/// Lives inside the crate because the struct's fields (src/nizk/presentation.rs:118-127) are private.  An item whose vectors are
/// shorter than the group's shape (it cannot happen behind `shape_key`) leaves zero cells, which fail verification.
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
    for (i, p) in batch.iter().enumerate() {
        put(&mut c.challenge, 0, count, i, p.proof.challenge.as_bytes());
        for (k, r) in p.proof.responses.iter().take(nr).enumerate() { put(&mut c.responses, k, count, i, r.as_bytes()); }
        put(&mut c.c_x_0, 0, count, i, p.C_x_0.compress().as_bytes());
        put(&mut c.c_x_1, 0, count, i, p.C_x_1.compress().as_bytes());
        put(&mut c.c_v, 0, count, i, p.C_V.compress().as_bytes());
        for (k, y) in p.C_y.iter().take(n).enumerate() { put(&mut c.c_y, k, count, i, y.compress().as_bytes()); }
        for (k, a) in p.encrypted_attributes.iter().take(n).enumerate() {
            match a {
                EncryptedAttribute::PublicScalar(m) => put(&mut c.attr_values, k, count, i, m.as_bytes()),
                EncryptedAttribute::PublicPoint(M) => put(&mut c.attr_values, k, count, i, M.compress().as_bytes()),
                _ => {}
            }
        }
        for (e, (_, q)) in p.proofs_of_encryption.iter().take(ne).enumerate() {
            put(&mut c.enc[e][0], 0, count, i, q.proof.challenge.as_bytes());
            for (k, r) in q.proof.responses.iter().take(6).enumerate() { put(&mut c.enc[e][1], k, count, i, r.as_bytes()); }
            put(&mut c.enc[e][2], 0, count, i, q.public_key.pk.compress().as_bytes());
            put(&mut c.enc[e][3], 0, count, i, q.ciphertext.E1.compress().as_bytes());
            put(&mut c.enc[e][4], 0, count, i, q.ciphertext.E2.compress().as_bytes());
            put(&mut c.enc[e][5], 0, count, i, q.C_y_1.compress().as_bytes());
            put(&mut c.enc[e][6], 0, count, i, q.C_y_2.compress().as_bytes());
            put(&mut c.enc[e][7], 0, count, i, q.C_y_3.compress().as_bytes());
            put(&mut c.enc[e][8], 0, count, i, q.C_y_2_prime.compress().as_bytes());
        }
    }
    (shape, c)
}
