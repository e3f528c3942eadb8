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
// 3. The engine's fast door: bytes in, bytes out.  `presentation_to_bytes` / `presentation_from_bytes`, `encryption_proof_to_bytes` /
//    `_from_bytes`, `issuance_to_bytes` / `issuance_from_bytes`, `issuer_parameters_to_bytes` / `_from_bytes` write and read exactly the
//    library's AFXP / AFXI v1 records (include/aeonflux_gpu.h "wire format"; the crate's own XXX: "the commitments should be
//    compressed", src/nizk/presentation.rs:117; style of src/parameters.rs:155-184).  `CompressedPresentation` keeps the 32-byte cells
//    it was parsed from: `GpuIssuer::verify_wire(&[u8])` / `verify_compressed` hand a server's network bytes to the engine without one
//    point decompressed or compressed on the host, `GpuUser::show_batch_wire` returns what `show` made as such bytes, and
//    `GpuUser::verify_issuances_wire` checks an issuer's AFXI answer.
//
// Errors are values (src/errors.rs:73-89): a per-item engine status becomes the `CredentialError` the crate returns for that
// outcome; a batch-level code about the caller's data (AFX_E_BAD_ARGS ...) becomes an `Err` for every item of the call
// (`engine_error` below maps it).  An ENGINE FAULT - AFX_E_NO_DEVICE, AFX_E_HIP, AFX_E_NO_MEMORY: the accelerator, not the data - is
// never turned into a cryptographic verdict: the `try_*` forms return it as `Err(EngineFault)`, and the crate-signature methods and
// the batch forms then run the crate's own body on the CPU for that call (`fall_throughs()` counts how often; `last_fault_code()` has
// the latest code).  A GPU reset therefore slows a server down; it does not turn honest users away.  Nothing here panics on caller
// data or on an engine fault.
//
// Threads.  One engine serves every thread of a server, as `&self` does in the crate: concurrent small calls on one context are
// COLLECTED inside the library into shared launch sets (afx_ctx_set_coalescing: 64 threads x 1 presentation = one pass of 64), so
// `install_issuer` with ONE engine scales with the caller threads.
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
use core::sync::atomic::{AtomicI32, AtomicPtr, AtomicUsize, Ordering};

use curve25519_dalek::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek::scalar::Scalar;
use curve25519_dalek::traits::Identity;
use rand_core::{CryptoRng, RngCore};
use zeroize::Zeroize;
use zkp::CompactProof;

use crate::amacs::{Amac, Attribute, EncryptedAttribute, SecretKey as AmacsSecretKey};
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
const E_NO_DEVICE: i32 = -3;
const E_HIP: i32 = -4;
const E_NO_KEY: i32 = -5;
const E_NO_MEMORY: i32 = -6;
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
    fn afx_verify_presentations_mixed_wire(ctx: *mut c_void, blob: *const u8, len: usize, status: *mut u8, status_cap: usize, count_out: *mut usize) -> i32;
    fn afx_group_verify_presentations_mixed_wire(group: *mut c_void, blob: *const u8, len: usize, status: *mut u8, status_cap: usize, count_out: *mut usize) -> i32;
    fn afx_verify_issuances_wire(ctx: *mut c_void, blob: *const u8, len: usize, status: *mut u8, status_cap: usize, count_out: *mut usize) -> i32;
}

/// An engine-level failure - AFX_E_NO_DEVICE, AFX_E_HIP or AFX_E_NO_MEMORY (the raw code is inside): the accelerator, not the data.
/// The only thing that makes a call fall through to the crate's own body.
#[derive(Clone, Copy, Debug)]
pub struct EngineFault(pub i32);
fn fault_of(rc: i32) -> Option<EngineFault> {
    if rc == E_NO_DEVICE || rc == E_HIP || rc == E_NO_MEMORY { Some(EngineFault(rc)) } else { None }
}

/// Which of the crate's operations an engine failure interrupted: decides the nearest `CredentialError`.
#[derive(Clone, Copy)]
enum Op { Create, Issue, Verify, Show, VerifyIssuance }

/// A batch-level engine return code (AFX_E_*, include/aeonflux_gpu.h) as the crate's error type (src/errors.rs:73-89).  Codes about
/// the caller's data map to what the crate would say.  The crate has no variant for "the accelerator failed": an engine fault
/// (`fault_of`) never reaches this function on the serving paths - those fall through to the CPU - and maps to the operation's own
/// failure variant where there is nothing to fall back to (engine creation; the wire forms' `EngineFault` is returned as it is).
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
/// `fallback`: the engine's own copy of the crate's `Issuer` (the key is wiped with it, src/amacs.rs:64-82): what a call runs on when
/// the accelerator fails.
pub struct GpuIssuer { ctx: *mut c_void, group: *mut c_void, n: usize, params: Vec<u8>, issuer_params: [u8; 64], fallback: Issuer,
                       fall_throughs: AtomicUsize, last_fault: AtomicI32 }

/// The user's side (no issuer key): `AnonymousCredential::show` and `CredentialIssuance::verify`.
pub struct GpuUser { ctx: *mut c_void, group: *mut c_void, n: usize, params: Vec<u8>, issuer_params: [u8; 64],
                     fallback_sp: SystemParameters, fallback_ip: IssuerParameters, fall_throughs: AtomicUsize, last_fault: AtomicI32 }

// The handles are plain pointers into the library, which may be called from any number of threads at once (include/aeonflux_gpu.h
// afx_ctx_create: small calls that arrive together share launch sets, everything else takes the context in turn); a group's
// members likewise.
unsafe impl Send for GpuIssuer {}
unsafe impl Sync for GpuIssuer {}
unsafe impl Send for GpuUser {}
unsafe impl Sync for GpuUser {}

/// The crate's `Issuer` again, field by field (it is not `Clone`; its parts are: src/parameters.rs:61,340, src/amacs.rs:51).
fn issuer_copy(issuer: &Issuer) -> Issuer {
    let k: &AmacsSecretKey = &issuer.amacs_key;
    Issuer { system_parameters: issuer.system_parameters.clone(), issuer_parameters: issuer.issuer_parameters.clone(), amacs_key: k.clone() }
}

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
                       fallback: issuer_copy(issuer), fall_throughs: AtomicUsize::new(0), last_fault: AtomicI32::new(0) })
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
                       fallback: issuer_copy(issuer), fall_throughs: AtomicUsize::new(0), last_fault: AtomicI32::new(0) })
    }

    /// Where secrets are kept out of table addresses (`SecretAddressing`; the default covers `issue`, `Everywhere` adds the issuer
    /// key's terms of `verify`).
    pub fn set_secret_addressing(&self, mode: SecretAddressing) -> Result<(), CredentialError> { set_secret_addressing(self.ctx, self.group, mode) }

    /// How many calls ran the crate's CPU body because the accelerator failed under them (a health counter: it decides nothing).
    pub fn fall_throughs(&self) -> usize { self.fall_throughs.load(Ordering::Relaxed) }
    /// The AFX_E_* code of the latest such failure (0: none yet).
    pub fn last_fault_code(&self) -> i32 { self.last_fault.load(Ordering::Relaxed) }
    fn fell_through(&self, f: EngineFault) { self.fall_throughs.fetch_add(1, Ordering::Relaxed); self.last_fault.store(f.0, Ordering::Relaxed); }

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
        match self.try_issue(request, csprng) {
            Ok(r) => r,
            Err((f, request)) => { self.fell_through(f); self.issue_on_cpu(request, csprng) }
        }
    }

    /// The same, with an engine fault handed back (together with the request, which was not consumed): what the delegating patch
    /// of INTEGRATION.md section 1 calls before it falls through to the crate's own body.
    pub fn try_issue<C: CryptoRng + RngCore>(&self, request: CredentialRequest, csprng: &mut C)
        -> Result<Result<CredentialIssuance, CredentialError>, (EngineFault, CredentialRequest)>
    {
        match self.try_issue_batch(vec![request], csprng) {
            Ok(mut v) => Ok(v.pop().unwrap_or(Err(CredentialError::CredentialIssuance))),
            Err((f, mut back)) => match back.pop() {
                Some(request) => Err((f, request)),
                None => Ok(Err(CredentialError::CredentialIssuance)),            // (cannot happen: a fault returns every request)
            },
        }
    }

    /// The body of `Issuer::issue` (src/issuer.rs:119-123) on the engine's copy of the issuer: where a call goes when the
    /// accelerator fails.  (Not `self.fallback.issue(..)`: with the delegating patch installed that would come back here.)
    fn issue_on_cpu<C: CryptoRng + RngCore>(&self, request: CredentialRequest, csprng: &mut C) -> Result<CredentialIssuance, CredentialError> {
        let amac = Amac::tag(csprng, &self.fallback.system_parameters, &self.fallback.amacs_key, &request.attributes)?;
        let cred = AnonymousCredential { amac, attributes: request.attributes };
        let proof = ProofOfIssuance::prove(&self.fallback, &cred);
        Ok(CredentialIssuance { proof: proof, credential: cred })
    }

    /// `Issuer::verify` (src/issuer.rs:141-145), on the GPU: one presentation through the engine's latency plan.
    pub fn verify(
        &self,
        presentation: &ProofOfValidCredential,
    ) -> Result<(), CredentialError>
    {
        match self.try_verify(presentation) {
            Ok(r) => r,
            Err(f) => { self.fell_through(f); presentation.verify(&self.fallback) }      // the crate's own body (src/issuer.rs:146)
        }
    }

    /// The same, with an engine fault handed back instead of a verdict (the delegating patch falls through on it).
    pub fn try_verify(&self, presentation: &ProofOfValidCredential) -> Result<Result<(), CredentialError>, EngineFault> {
        Ok(self.try_verify_batch(core::slice::from_ref(presentation))?.pop().unwrap_or(Err(CredentialError::VerificationFailure)))
    }

    /// Batch `Issuer::issue` (src/issuer.rs:111-124) over requests of ANY attribute layouts: consumes the requests like the
    /// reference does and returns one `Result` per request, in order.  The requests are grouped by layout (`layout_key`: the kind
    /// of every attribute, src/amacs.rs:168-179) and all groups go to the engine in one call.
    pub fn issue_batch<C: CryptoRng + RngCore>(&self, requests: Vec<CredentialRequest>, csprng: &mut C)
        -> Vec<Result<CredentialIssuance, CredentialError>>
    {
        match self.try_issue_batch(requests, csprng) {
            Ok(v) => v,
            Err((f, requests)) => { self.fell_through(f); requests.into_iter().map(|r| self.issue_on_cpu(r, &mut *csprng)).collect() }
        }
    }

    /// `issue_batch` with an engine fault handed back together with the (unconsumed) requests.
    pub fn try_issue_batch<C: CryptoRng + RngCore>(&self, requests: Vec<CredentialRequest>, csprng: &mut C)
        -> Result<Vec<Result<CredentialIssuance, CredentialError>>, (EngineFault, Vec<CredentialRequest>)>
    {
        let count = requests.len();
        if count == 0 { return Ok(Vec::new()); }
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
            drop(groups);                                             // (they point into `stages`, which borrows nothing of `requests`)
            if let Some(f) = fault_of(rc) { return Err((f, requests)); }
            return Ok((0..count).map(|_| Err(engine_error(rc, Op::Issue))).collect());
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
        Ok(out.into_iter().map(|r| r.unwrap_or(Err(CredentialError::CredentialIssuance))).collect())
    }

    /// Batch `Issuer::verify` (src/issuer.rs:141-147) over ANY presentations: like the reference, which reads the shape from
    /// each presentation's own fields (src/nizk/presentation.rs:293-309, :324-443), the batch is grouped by the full shape -
    /// attribute kinds, hidden scalar indices, response count, the indices of the attached proofs of encryption - and every
    /// group is verified under its own statement.  Results come back in the order given.  A presentation whose vectors do not
    /// fit together (lengths the reference would index out of range on, `presentation.rs:346,351,407`; a proof of encryption
    /// without its six responses, which zkp rejects) is answered with `VerificationFailure` without reaching the engine.
    pub fn verify_batch(&self, batch: &[ProofOfValidCredential]) -> Vec<Result<(), CredentialError>> {
        match self.try_verify_batch(batch) {
            Ok(v) => v,
            Err(f) => { self.fell_through(f); batch.iter().map(|p| p.verify(&self.fallback)).collect() }
        }
    }

    /// `verify_batch` with an engine fault handed back instead of verdicts.
    pub fn try_verify_batch(&self, batch: &[ProofOfValidCredential]) -> Result<Vec<Result<(), CredentialError>>, EngineFault> {
        let total = batch.len();
        if total == 0 { return Ok(Vec::new()); }
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
            if let Some(f) = fault_of(rc) { return Err(f); }                           // the accelerator: no verdict is made up
            return Ok((0..total).map(|_| Err(engine_error(rc, Op::Verify))).collect());
        }
        Ok(status.iter().map(|s| if *s == ST_OK { Ok(()) } else { Err(CredentialError::VerificationFailure) }).collect())
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
        match self.try_show(keypair, &mut csprng) {
            Ok(r) => r,
            Err(f) => {                                                                  // the crate's own body (src/credential.rs:45)
                self.user.fell_through(f);
                ProofOfValidCredential::prove(&system_parameters, &issuer_parameters, self.credential, keypair, &mut csprng)
            }
        }
    }

    /// The same without the parameter check, with an engine fault handed back (the delegating patch falls through on it).
    pub fn try_show<C: CryptoRng + RngCore>(&self, keypair: Option<&SymmetricKeypair>, csprng: &mut C)
        -> Result<Result<ProofOfValidCredential, CredentialError>, EngineFault>
    {
        Ok(self.user.try_show_batch(core::slice::from_ref(self.credential), keypair.map(core::slice::from_ref), csprng)?
            .pop().unwrap_or(Err(CredentialError::MissingData)))
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
        match self.user.try_verify_issuance_batch(vec![self.issuance]) {
            Ok(mut v) => v.pop().unwrap_or(Err(CredentialError::VerificationFailure)),
            Err((f, mut back)) => {                                                      // the crate's own body (src/issuer.rs:54-56)
                self.user.fell_through(f);
                match back.pop() {
                    Some(iss) => iss.proof.verify(system_parameters, issuer_parameters, &iss.credential).and(Ok(iss.credential)),
                    None => Err(CredentialError::VerificationFailure),
                }
            }
        }
    }

    /// The same, with an engine fault handed back together with the (unconsumed) issuance.
    pub fn try_verify(self) -> Result<Result<AnonymousCredential, CredentialError>, (EngineFault, CredentialIssuance)> {
        match self.user.try_verify_issuance_batch(vec![self.issuance]) {
            Ok(mut v) => Ok(v.pop().unwrap_or(Err(CredentialError::VerificationFailure))),
            Err((f, mut back)) => match back.pop() {
                Some(iss) => Err((f, iss)),
                None => Ok(Err(CredentialError::VerificationFailure)),
            },
        }
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
                     fallback_sp: system_parameters.clone(), fallback_ip: issuer_parameters.clone(), fall_throughs: AtomicUsize::new(0), last_fault: AtomicI32::new(0) })
    }

    /// The same on several GPUs of the node: batches are split contiguously over `devices` inside the library.
    pub fn new_multi_user(system_parameters: &SystemParameters, issuer_parameters: &IssuerParameters, devices: &[i32]) -> Result<GpuUser, CredentialError> {
        let sp = system_parameters.to_bytes();
        let ip = issuer_params_bytes(issuer_parameters);
        let mut group = core::ptr::null_mut();
        let rc = unsafe { afx_group_create(&mut group, devices.as_ptr(), devices.len() as u32, sp.as_ptr(), sp.len(), core::ptr::null(), 0, ip.as_ptr()) };
        if rc != 0 { return Err(engine_error(rc, Op::Create)); }
        Ok(GpuUser { ctx: core::ptr::null_mut(), group, n: system_parameters.NUMBER_OF_ATTRIBUTES as usize, params: sp, issuer_params: ip,
                     fallback_sp: system_parameters.clone(), fallback_ip: issuer_parameters.clone(), fall_throughs: AtomicUsize::new(0), last_fault: AtomicI32::new(0) })
    }

    /// Where secrets are kept out of table addresses (`SecretAddressing`; the default covers every scalar of `show`: blindings, the
    /// credential's `t`, the symmetric key).
    pub fn set_secret_addressing(&self, mode: SecretAddressing) -> Result<(), CredentialError> { set_secret_addressing(self.ctx, self.group, mode) }

    /// How many calls ran the crate's CPU body because the accelerator failed under them (a health counter: it decides nothing).
    pub fn fall_throughs(&self) -> usize { self.fall_throughs.load(Ordering::Relaxed) }
    /// The AFX_E_* code of the latest such failure (0: none yet).
    pub fn last_fault_code(&self) -> i32 { self.last_fault.load(Ordering::Relaxed) }
    fn fell_through(&self, f: EngineFault) { self.fall_throughs.fetch_add(1, Ordering::Relaxed); self.last_fault.store(f.0, Ordering::Relaxed); }

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
        match self.try_show_batch(creds, keypairs, csprng) {
            Ok(v) => v,
            Err(f) => { self.fell_through(f); self.show_on_cpu(creds, keypairs, csprng) }
        }
    }

    /// `show_batch` with an engine fault handed back instead of presentations.
    pub fn try_show_batch<C: CryptoRng + RngCore>(&self, creds: &[AnonymousCredential], keypairs: Option<&[SymmetricKeypair]>, csprng: &mut C)
        -> Result<Vec<Result<ProofOfValidCredential, CredentialError>>, EngineFault>
    {
        let count = creds.len();
        match self.run_show(creds, keypairs, csprng) {
            Ok(run) => Ok(finish_show(&run, count, rebuild_presentation)),
            Err(ShowStop::Fault(f)) => Err(f),
            Err(ShowStop::All(e)) => Ok((0..count).map(|_| Err(e)).collect()),
        }
    }

    /// The same presentations as the bytes a user sends to the issuer (one AFXP v1 section each: `CompressedPresentation`), written
    /// from the 32-byte cells the engine returned - no point is decompressed on the way.
    pub fn show_batch_wire<C: CryptoRng + RngCore>(&self, creds: &[AnonymousCredential], keypairs: Option<&[SymmetricKeypair]>, csprng: &mut C)
        -> Vec<Result<CompressedPresentation, CredentialError>>
    {
        let count = creds.len();
        match self.run_show(creds, keypairs, csprng) {
            Ok(run) => finish_show(&run, count, record_of_shown),
            Err(ShowStop::All(e)) => (0..count).map(|_| Err(e)).collect(),
            Err(ShowStop::Fault(f)) => {
                self.fell_through(f);
                self.show_on_cpu(creds, keypairs, csprng).into_iter().map(|r| r.and_then(|p| CompressedPresentation::from_presentation(&p))).collect()
            }
        }
    }

    /// The crate's own `ProofOfValidCredential::prove` per credential (src/credential.rs:45): where a call goes when the accelerator fails.
    fn show_on_cpu<C: CryptoRng + RngCore>(&self, creds: &[AnonymousCredential], keypairs: Option<&[SymmetricKeypair]>, csprng: &mut C)
        -> Vec<Result<ProofOfValidCredential, CredentialError>>
    {
        if let Some(kps) = keypairs { if kps.len() != creds.len() { return (0..creds.len()).map(|_| Err(CredentialError::MissingData)).collect(); } }
        creds.iter().enumerate().map(|(i, c)| {
            let kp: Option<&SymmetricKeypair> = match keypairs { Some(kps) => Some(&kps[i]), None => None };
            ProofOfValidCredential::prove(&self.fallback_sp, &self.fallback_ip, c, kp, &mut *csprng)
        }).collect()
    }

    /// The engine part of `show`: group by layout, stage, draw, call.
    fn run_show<C: CryptoRng + RngCore>(&self, creds: &[AnonymousCredential], keypairs: Option<&[SymmetricKeypair]>, csprng: &mut C) -> Result<ShowRun, ShowStop> {
        let count = creds.len();
        if count == 0 { return Ok(ShowRun { stages: Vec::new(), shapes: Vec::new(), status: Vec::new(), early: Vec::new() }); }
        if let Some(kps) = keypairs { if kps.len() != count { return Err(ShowStop::All(CredentialError::MissingData)); } }
        let mut early: Vec<Option<CredentialError>> = (0..count).map(|_| None).collect();
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
                None => early[i] = Some(CredentialError::WrongNumberOfAttributes),
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
        let mut stages: Vec<ShowStage> = grouped(by_layout).into_iter().map(|(key, members)| {
            let (m, na) = (members.len(), key.len());
            let hs = key.iter().filter(|x| **x == ATTR_SECRET_SCALAR).count();
            let nsp = key.iter().filter(|x| **x == ATTR_SECRET_POINT).count();
            let col = |k: usize| vec![0u8; 32 * k * m];
            // hidden attribute values are secrets of the user (amacs::Attribute zeroizes them, src/amacs.rs:184-200)
            let mut st = ShowStage { positions: positions_of(&members), hs, nsp, values: Wiped::new(32 * na * m), m2: Wiped::new(32 * na * m), m3: Wiped::new(32 * na * m),
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
            if let Some(f) = fault_of(rc) { return Err(ShowStop::Fault(f)); }
            return Err(ShowStop::All(engine_error(rc, Op::Show)));
        }
        let shapes: Vec<AfxShape> = groups.iter().map(|g| g.shape_out).collect();
        Ok(ShowRun { stages, shapes, status, early })
    }

    /// Batch `CredentialIssuance::verify` (src/issuer.rs:48-57) over issuances of ANY attribute layouts: consumes the issuances
    /// and moves each credential out on success.
    pub fn verify_issuance_batch(&self, issuances: Vec<CredentialIssuance>) -> Vec<Result<AnonymousCredential, CredentialError>> {
        match self.try_verify_issuance_batch(issuances) {
            Ok(v) => v,
            Err((f, issuances)) => {                                                   // the crate's own body (src/issuer.rs:54-56)
                self.fell_through(f);
                issuances.into_iter().map(|iss| iss.proof.verify(&self.fallback_sp, &self.fallback_ip, &iss.credential).and(Ok(iss.credential))).collect()
            }
        }
    }

    /// `verify_issuance_batch` with an engine fault handed back together with the (unconsumed) issuances.
    pub fn try_verify_issuance_batch(&self, issuances: Vec<CredentialIssuance>)
        -> Result<Vec<Result<AnonymousCredential, CredentialError>>, (EngineFault, Vec<CredentialIssuance>)>
    {
        let count = issuances.len();
        if count == 0 { return Ok(Vec::new()); }
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
            if let Some(f) = fault_of(rc) { return Err((f, issuances)); }              // the accelerator: no verdict is made up
            return Ok((0..count).map(|_| Err(engine_error(rc, Op::VerifyIssuance))).collect());
        }
        Ok(issuances.into_iter().enumerate().map(|(i, iss)| if status[i] == ST_OK { Ok(iss.credential) } else { Err(CredentialError::VerificationFailure) }).collect())
    }

    /// `CredentialIssuance::verify` over an issuer's serialized answer (an AFXI v1 batch, `issuance_to_bytes` /
    /// afx_issuance_wire_pack): one verdict per record, nothing decompressed on the host.  `Err(MissingData)`: the bytes are no AFXI
    /// batch.  On an engine fault the records are rebuilt and checked by the crate's own `ProofOfIssuance::verify`.
    pub fn verify_issuances_wire(&self, blob: &[u8]) -> Result<Vec<Result<(), CredentialError>>, CredentialError> {
        match self.try_verify_issuances_wire(blob) {
            Ok(r) => r,
            Err(f) => {
                self.fell_through(f);
                let (kinds, nr, count, off) = afxi_parse(blob).ok_or(CredentialError::MissingData)?;
                let cells = 4 + nr + kinds.len();
                Ok((0..count).map(|i| {
                    let iss = issuance_from_record(&kinds, nr, &blob[off + 32 * cells * i..off + 32 * cells * (i + 1)], None)?;
                    iss.proof.verify(&self.fallback_sp, &self.fallback_ip, &iss.credential)
                }).collect())
            }
        }
    }

    /// The same, with an engine fault handed back.
    pub fn try_verify_issuances_wire(&self, blob: &[u8]) -> Result<Result<Vec<Result<(), CredentialError>>, CredentialError>, EngineFault> {
        let cap = blob.len() / 32 + 1;
        let mut status = vec![ST_VERIFICATION_FAILURE; cap];
        let mut count = 0usize;
        let ctx = if self.group.is_null() { self.ctx } else { unsafe { afx_group_member(self.group, 0) } };   // (no group form of this call: member 0 takes it)
        let rc = unsafe { afx_verify_issuances_wire(ctx, blob.as_ptr(), blob.len(), status.as_mut_ptr(), cap, &mut count) };
        if rc != 0 {
            if let Some(f) = fault_of(rc) { return Err(f); }
            return Ok(Err(engine_error(rc, Op::VerifyIssuance)));
        }
        Ok(Ok(status.iter().take(count).map(|s| if *s == ST_OK { Ok(()) } else { Err(CredentialError::VerificationFailure) }).collect()))
    }
}

/// One layout group of a `show` call: its inputs staged column-major and the arrays the engine writes.
struct ShowStage { key: Vec<u8>, members: Vec<usize>, positions: Vec<u64>, hs: usize, nsp: usize,
                   values: Wiped, m2: Wiped, m3: Wiped, t: Vec<u8>, u: Vec<u8>, v: Vec<u8>, ka: Wiped, ka0: Wiped, ka1: Wiped, kpk: Vec<u8>,
                   z_wide: Wiped, seed: Wiped, enc_seeds: Wiped,
                   o_ch: Vec<u8>, o_rs: Vec<u8>, o_x0: Vec<u8>, o_x1: Vec<u8>, o_cv: Vec<u8>, o_cy: Vec<u8>, o_av: Vec<u8>, enc_cols: Vec<[Vec<u8>; 9]> }
/// What the engine made of a `show` call: per layout group its arrays and the presentation shape, per credential its status;
/// `early`: credentials that never reached the engine.
struct ShowRun { stages: Vec<ShowStage>, shapes: Vec<AfxShape>, status: Vec<u8>, early: Vec<Option<CredentialError>> }
/// Why a `show` call made no presentations at all: the accelerator failed, or one answer fits every credential.
enum ShowStop { Fault(EngineFault), All(CredentialError) }

/// Per credential, in the caller's order: `make(group, shape, group size, index in the group)` for the ones the engine showed.
fn finish_show<T>(run: &ShowRun, count: usize, make: fn(&ShowStage, &AfxShape, usize, usize) -> Result<T, CredentialError>) -> Vec<Result<T, CredentialError>> {
    let mut out: Vec<Option<Result<T, CredentialError>>> = (0..count).map(|i| run.early.get(i).and_then(|e| *e).map(Err)).collect();
    for (g, st) in run.stages.iter().enumerate() {
        let m = st.members.len();
        for (j, i) in st.members.iter().enumerate() {
            out[*i] = Some(match run.status[*i] {
                ST_OK => make(st, &run.shapes[g], m, j),
                ST_NO_SYMMETRIC_KEY => Err(CredentialError::NoSymmetricKey),
                _ => Err(CredentialError::MissingData),                            // a status this operation does not have: no presentation is made up
            });
        }
    }
    out.into_iter().map(|r| r.unwrap_or(Err(CredentialError::MissingData))).collect()
}

/// `ProofOfValidCredential` (src/nizk/presentation.rs:118-127) from item `j` of a group's output arrays: every point is decompressed.
fn rebuild_presentation(st: &ShowStage, shape: &AfxShape, m: usize, j: usize) -> Result<ProofOfValidCredential, CredentialError> {
    let na = st.key.len();
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
}

/// The same item as the bytes that go on the wire: the engine's 32-byte cells in AFXP record order, nothing decompressed.
fn record_of_shown(st: &ShowStage, shape: &AfxShape, m: usize, j: usize) -> Result<CompressedPresentation, CredentialError> {
    let mut bytes = afxp_header(shape, 1).ok_or(CredentialError::MissingData)?;
    afxp_record(&mut bytes, shape, m, j, &st.o_ch, &st.o_rs, &st.o_x0, &st.o_x1, &st.o_cv, &st.o_cy, &st.o_av, &st.enc_cols);
    Ok(CompressedPresentation { bytes })
}

impl Drop for GpuUser {
    fn drop(&mut self) { unsafe { if self.group.is_null() { afx_ctx_destroy(self.ctx) } else { afx_group_destroy(self.group) } } }
}

// ---- the wire format: AFXP v1 (presentations) and AFXI v1 (issuances), include/aeonflux_gpu.h ----------------------------------
// The crate has no byte form for these messages (`// XXX the commitments should be compressed`, src/nizk/presentation.rs:117; its
// `IssuerParameters::{to,from}_bytes` are `unimplemented!()`, src/parameters.rs:365-372).  The functions below write and read exactly
// what the library's C packers and parsers do (afx_wire_pack_presentations, afx_wire_parse, afx_issuance_wire_pack,
// afx_issuance_wire_parse; style of src/parameters.rs:155-184: little-endian u32 counts, then 32-byte items), so the bytes a Rust
// client writes are the bytes the engine verifies: tests/golden/wire.json holds C-packed messages and integration/pin_against_crate.rs
// asserts these writers reproduce them byte for byte.
//   AFXP: "AFXP" | u32 1 | u32 count | u32 cells | u32 n | u32 nr | u32 hs | u32 ne | kinds[n] | hidden[hs] u16 | enc_indices[ne] u16 | 0-pad to 32
//         then per record: challenge | responses[nr] | C_x_0 | C_x_1 | C_V | C_y[n] | value of every public attribute | per proof of
//         encryption: challenge | responses[6] | pk | E1 | E2 | C_y_1 | C_y_2 | C_y_3 | C_y_2'
//   AFXI: "AFXI" | u32 1 | u32 count | u32 cells | u32 n | u32 nr | kinds[n] | 0-pad to 32
//         then per record: t | U | V | challenge | responses[nr] | value of every attribute (a scalar, a point, or a plaintext's M1)

fn is_public_kind(k: u8) -> bool { k == 0 || k == 2 }
fn afxp_cells(shape: &AfxShape) -> usize {
    let n = shape.n_attributes as usize;
    let public = shape.kinds[..n.min(AFX_MAX_ATTRIBUTES)].iter().filter(|k| is_public_kind(**k)).count();
    1 + shape.n_responses as usize + 3 + n + public + 14 * shape.n_enc_proofs as usize
}
/// Header of a batch of `count` presentations of this shape; `None`: a shape the format does not carry.
fn afxp_header(shape: &AfxShape, count: u32) -> Option<Vec<u8>> {
    let (n, nr, hs, ne) = (shape.n_attributes as usize, shape.n_responses as usize, shape.n_hidden_scalars as usize, shape.n_enc_proofs as usize);
    if n > AFX_MAX_ATTRIBUTES || nr > 3 + AFX_MAX_ATTRIBUTES || hs > AFX_MAX_ATTRIBUTES || ne > AFX_MAX_ATTRIBUTES { return None; }
    if shape.kinds[..n].iter().any(|k| *k > 3) { return None; }
    let mut h = Vec::with_capacity(96);
    h.extend_from_slice(b"AFXP");
    for v in [1u32, count, afxp_cells(shape) as u32, n as u32, nr as u32, hs as u32, ne as u32].iter() { h.extend_from_slice(&v.to_le_bytes()); }
    h.extend_from_slice(&shape.kinds[..n]);
    for i in 0..hs { h.extend_from_slice(&shape.hidden_scalar_indices[i].to_le_bytes()); }
    for i in 0..ne { h.extend_from_slice(&shape.enc_indices[i].to_le_bytes()); }
    while h.len() % 32 != 0 { h.push(0); }
    Some(h)
}
/// Appends the record of item `j` of column arrays over `m` items (the engine's struct-of-arrays layout, `Columns` / `ShowStage`).
fn afxp_record(out: &mut Vec<u8>, shape: &AfxShape, m: usize, j: usize, challenge: &[u8], responses: &[u8], c_x_0: &[u8], c_x_1: &[u8], c_v: &[u8],
               c_y: &[u8], attr_values: &[u8], enc: &[[Vec<u8>; 9]]) {
    let n = shape.n_attributes as usize;
    out.extend_from_slice(&cell(challenge, 0, m, j));
    for k in 0..shape.n_responses as usize { out.extend_from_slice(&cell(responses, k, m, j)); }
    out.extend_from_slice(&cell(c_x_0, 0, m, j));
    out.extend_from_slice(&cell(c_x_1, 0, m, j));
    out.extend_from_slice(&cell(c_v, 0, m, j));
    for k in 0..n { out.extend_from_slice(&cell(c_y, k, m, j)); }
    for k in 0..n { if is_public_kind(shape.kinds[k]) { out.extend_from_slice(&cell(attr_values, k, m, j)); } }
    for e in enc.iter().take(shape.n_enc_proofs as usize) {
        out.extend_from_slice(&cell(&e[0], 0, m, j));
        for k in 0..6 { out.extend_from_slice(&cell(&e[1], k, m, j)); }
        for f in 2..9 { out.extend_from_slice(&cell(&e[f], 0, m, j)); }
    }
}
fn rd_u32(b: &[u8], at: usize) -> u32 { u32::from_le_bytes([b[at], b[at + 1], b[at + 2], b[at + 3]]) }
/// The header of an AFXP section at the start of `b` (afx_wire_parse): shape, count, offset of the records, cells per record, and
/// the section's length.  `None`: not an AFXP v1 section, or one that runs past the end of `b`.
fn afxp_parse(b: &[u8]) -> Option<(AfxShape, usize, usize, usize, usize)> {
    if b.len() < 32 || &b[..4] != b"AFXP" || rd_u32(b, 4) != 1 { return None; }
    let (count, cells) = (rd_u32(b, 8) as usize, rd_u32(b, 12) as usize);
    let (n, nr, hs, ne) = (rd_u32(b, 16) as usize, rd_u32(b, 20) as usize, rd_u32(b, 24) as usize, rd_u32(b, 28) as usize);
    if n > AFX_MAX_ATTRIBUTES || nr > 3 + AFX_MAX_ATTRIBUTES || hs > AFX_MAX_ATTRIBUTES || ne > AFX_MAX_ATTRIBUTES { return None; }
    let hdr = (32 + n + 2 * hs + 2 * ne + 31) & !31usize;
    if b.len() < hdr { return None; }
    let mut shape = AfxShape { n_attributes: n as u32, kinds: [0; AFX_MAX_ATTRIBUTES], n_responses: nr as u32, n_hidden_scalars: hs as u32,
                               hidden_scalar_indices: [0; AFX_MAX_ATTRIBUTES], n_enc_proofs: ne as u32, enc_indices: [0; AFX_MAX_ATTRIBUTES] };
    let mut at = 32;
    for i in 0..n { shape.kinds[i] = b[at]; at += 1; if shape.kinds[i] > 3 { return None; } }
    for i in 0..hs { shape.hidden_scalar_indices[i] = u16::from_le_bytes([b[at], b[at + 1]]); at += 2; }
    for i in 0..ne { shape.enc_indices[i] = u16::from_le_bytes([b[at], b[at + 1]]); at += 2; }
    if cells != afxp_cells(&shape) { return None; }
    let len = hdr.checked_add(count.checked_mul(cells)?.checked_mul(32)?)?;
    if len > b.len() { return None; }
    Some((shape, count, hdr, cells, len))
}

/// A presentation as it travels: one AFXP v1 section with one record - the 32-byte cells exactly as they were received (or as the
/// engine made them).  Parsing checks the framing only; no point is decompressed until `decompress` is asked for, and
/// `GpuIssuer::verify_compressed` / `verify_wire` never ask.
#[derive(Clone)]
pub struct CompressedPresentation { bytes: Vec<u8> }

impl CompressedPresentation {
    /// `ProofOfValidCredential::from_bytes` without the decompression: `WrongNumberOfBytes` unless `bytes` is exactly one section
    /// holding exactly one record.
    pub fn from_bytes(bytes: &[u8]) -> Result<CompressedPresentation, CredentialError> {
        match afxp_parse(bytes) {
            Some((_, 1, _, _, len)) if len == bytes.len() => Ok(CompressedPresentation { bytes: bytes.to_vec() }),
            _ => Err(CredentialError::WrongNumberOfBytes),
        }
    }
    pub fn as_bytes(&self) -> &[u8] { &self.bytes }
    pub fn to_bytes(&self) -> Vec<u8> { self.bytes.clone() }
    /// The compressed form of a presentation the crate holds (every point is compressed: the one-off cost the crate's XXX is about).
    pub fn from_presentation(p: &ProofOfValidCredential) -> Result<CompressedPresentation, CredentialError> {
        if shape_key(p).is_none() { return Err(CredentialError::MissingData); }       // vectors that do not fit together have no byte form
        let (shape, c) = marshal(&[p]);
        let mut bytes = afxp_header(&shape, 1).ok_or(CredentialError::MissingData)?;
        afxp_record(&mut bytes, &shape, 1, 0, &c.challenge, &c.responses, &c.c_x_0, &c.c_x_1, &c.c_v, &c.c_y, &c.attr_values, &c.enc);
        Ok(CompressedPresentation { bytes })
    }
    /// The crate's struct again: scalars must be canonical and points must decompress (`ScalarFormatError` /
    /// `PointDecompressionError` otherwise - the checks `from_bytes` of the crate's other types make, src/parameters.rs:92-153).
    pub fn decompress(&self) -> Result<ProofOfValidCredential, CredentialError> {
        let (shape, _, off, cells, _) = afxp_parse(&self.bytes).ok_or(CredentialError::WrongNumberOfBytes)?;
        presentation_from_record(&shape, &self.bytes[off..off + 32 * cells])
    }
}

/// `rec`: one AFXP record (cells x 32 bytes) of shape `shape`.
fn presentation_from_record(shape: &AfxShape, rec: &[u8]) -> Result<ProofOfValidCredential, CredentialError> {
    let (n, nr, ne) = (shape.n_attributes as usize, shape.n_responses as usize, shape.n_enc_proofs as usize);
    if rec.len() != 32 * afxp_cells(shape) { return Err(CredentialError::WrongNumberOfBytes); }
    // a record is a column of `cells` rows over one item
    let s = |k: usize| sc(rec, k, 1, 0);
    let p = |k: usize| pt(rec, k, 1, 0);
    let proof = CompactProof { challenge: s(0)?, responses: (0..nr).map(|k| s(1 + k)).collect::<Result<Vec<Scalar>, CredentialError>>()? };
    let at = 1 + nr;
    let (c_x_0, c_x_1, c_v) = (p(at)?, p(at + 1)?, p(at + 2)?);
    let c_y = (0..n).map(|k| p(at + 3 + k)).collect::<Result<Vec<RistrettoPoint>, CredentialError>>()?;
    let mut at = at + 3 + n;
    let mut encrypted_attributes = Vec::with_capacity(n);
    for k in 0..n {
        encrypted_attributes.push(match shape.kinds[k] {
            0 => { at += 1; EncryptedAttribute::PublicScalar(s(at - 1)?) }
            1 => EncryptedAttribute::SecretScalar,
            2 => { at += 1; EncryptedAttribute::PublicPoint(p(at - 1)?) }
            _ => EncryptedAttribute::SecretPoint,
        });
    }
    let mut proofs_of_encryption = Vec::with_capacity(ne);
    for e in 0..ne {
        let index = shape.enc_indices[e];
        proofs_of_encryption.push((index, encryption_proof_from_cells(index, &rec[32 * at..32 * (at + 14)])?));
        at += 14;
    }
    Ok(ProofOfValidCredential { proof, proofs_of_encryption, encrypted_attributes,
                                hidden_scalar_indices: shape.hidden_scalar_indices[..shape.n_hidden_scalars as usize].to_vec(),
                                C_x_0: c_x_0, C_x_1: c_x_1, C_V: c_v, C_y: c_y })
}
/// 14 cells: challenge | responses[6] | pk | E1 | E2 | C_y_1 | C_y_2 | C_y_3 | C_y_2'
fn encryption_proof_from_cells(index: u16, c: &[u8]) -> Result<ProofOfEncryption, CredentialError> {
    if c.len() != 32 * 14 { return Err(CredentialError::WrongNumberOfBytes); }
    Ok(ProofOfEncryption {
        proof: CompactProof { challenge: sc(c, 0, 1, 0)?, responses: (1..7).map(|k| sc(c, k, 1, 0)).collect::<Result<Vec<Scalar>, CredentialError>>()? },
        public_key: SymmetricPublicKey { pk: pt(c, 7, 1, 0)? },
        ciphertext: Ciphertext { E1: pt(c, 8, 1, 0)?, E2: pt(c, 9, 1, 0)? },
        index,
        C_y_1: pt(c, 10, 1, 0)?, C_y_2: pt(c, 11, 1, 0)?, C_y_3: pt(c, 12, 1, 0)?, C_y_2_prime: pt(c, 13, 1, 0)?,
    })
}

/// `ProofOfValidCredential::to_bytes` (what src/nizk/presentation.rs:117 asks for): one AFXP v1 section with one record.
pub fn presentation_to_bytes(p: &ProofOfValidCredential) -> Result<Vec<u8>, CredentialError> { Ok(CompressedPresentation::from_presentation(p)?.bytes) }
/// `ProofOfValidCredential::from_bytes`.
pub fn presentation_from_bytes(bytes: &[u8]) -> Result<ProofOfValidCredential, CredentialError> { CompressedPresentation::from_bytes(bytes)?.decompress() }

/// `ProofOfEncryption::to_bytes`: u32le index, then its 14 cells in the order they have inside a presentation's record.
pub fn encryption_proof_to_bytes(q: &ProofOfEncryption) -> Result<Vec<u8>, CredentialError> {
    if q.proof.responses.len() != 6 { return Err(CredentialError::MissingData); }
    let mut b = Vec::with_capacity(4 + 32 * 14);
    b.extend_from_slice(&(q.index as u32).to_le_bytes());
    b.extend_from_slice(q.proof.challenge.as_bytes());
    for r in q.proof.responses.iter() { b.extend_from_slice(r.as_bytes()); }
    for point in [&q.public_key.pk, &q.ciphertext.E1, &q.ciphertext.E2, &q.C_y_1, &q.C_y_2, &q.C_y_3, &q.C_y_2_prime].iter() { b.extend_from_slice(point.compress().as_bytes()); }
    Ok(b)
}
/// `ProofOfEncryption::from_bytes`.
pub fn encryption_proof_from_bytes(bytes: &[u8]) -> Result<ProofOfEncryption, CredentialError> {
    if bytes.len() != 4 + 32 * 14 { return Err(CredentialError::WrongNumberOfBytes); }
    let index = rd_u32(bytes, 0);
    if index > 0xffff { return Err(CredentialError::WrongNumberOfBytes); }
    encryption_proof_from_cells(index as u16, &bytes[4..])
}

/// `IssuerParameters::to_bytes` as the crate intends it (src/issuer.rs:155,163: 64 bytes): C_W || I.
pub fn issuer_parameters_to_bytes(ip: &IssuerParameters) -> Vec<u8> { issuer_params_bytes(ip).to_vec() }
/// `IssuerParameters::from_bytes` (the crate's is `unimplemented!()`, src/parameters.rs:365-367).
pub fn issuer_parameters_from_bytes(bytes: &[u8]) -> Result<IssuerParameters, CredentialError> {
    if bytes.len() != 64 { return Err(CredentialError::WrongNumberOfBytes); }
    Ok(IssuerParameters { C_W: pt(bytes, 0, 1, 0)?, I: pt(bytes, 1, 1, 0)? })
}

/// `CredentialIssuance::to_bytes`: one AFXI v1 batch with one record (t | U | V | challenge | responses | the value of every
/// attribute: a scalar, a point, or a plaintext's M1 - what the tag and the proof are made of, src/amacs.rs:225-243).
pub fn issuance_to_bytes(iss: &CredentialIssuance) -> Result<Vec<u8>, CredentialError> {
    let kinds = layout_key(&iss.credential.attributes).ok_or(CredentialError::WrongNumberOfAttributes)?;
    let (n, nr) = (kinds.len(), iss.proof.0.responses.len());
    if nr > AFX_MAX_ATTRIBUTES + 5 { return Err(CredentialError::MissingData); }
    let mut b = Vec::with_capacity(64 + 32 * (4 + nr + n));
    b.extend_from_slice(b"AFXI");
    for v in [1u32, 1u32, (4 + nr + n) as u32, n as u32, nr as u32].iter() { b.extend_from_slice(&v.to_le_bytes()); }
    b.extend_from_slice(&kinds);
    while b.len() % 32 != 0 { b.push(0); }
    b.extend_from_slice(iss.credential.amac.t.as_bytes());
    b.extend_from_slice(iss.credential.amac.U.compress().as_bytes());
    b.extend_from_slice(iss.credential.amac.V.compress().as_bytes());
    b.extend_from_slice(iss.proof.0.challenge.as_bytes());
    for r in iss.proof.0.responses.iter() { b.extend_from_slice(r.as_bytes()); }
    for a in iss.credential.attributes.iter() { b.extend_from_slice(&attribute_cells(a).1); }
    Ok(b)
}
/// The header of an AFXI batch (afx_issuance_wire_parse): attribute kinds, response count, record count, offset of the records.
fn afxi_parse(b: &[u8]) -> Option<(Vec<u8>, usize, usize, usize)> {
    if b.len() < 24 || &b[..4] != b"AFXI" || rd_u32(b, 4) != 1 { return None; }
    let (count, cells, n, nr) = (rd_u32(b, 8) as usize, rd_u32(b, 12) as usize, rd_u32(b, 16) as usize, rd_u32(b, 20) as usize);
    if n > AFX_MAX_ATTRIBUTES || nr > AFX_MAX_ATTRIBUTES + 5 || cells != 4 + nr + n { return None; }
    let hdr = (24 + n + 31) & !31usize;
    if b.len() < hdr { return None; }
    let kinds = b[24..24 + n].to_vec();
    if kinds.iter().any(|k| *k > ATTR_SECRET_POINT) { return None; }
    if hdr.checked_add(count.checked_mul(cells)?.checked_mul(32)?)? != b.len() { return None; }
    Some((kinds, nr, count, hdr))
}
/// One AFXI record -> the crate's `CredentialIssuance`.  `attributes`: the user's own (from the request they sent: the message
/// carries only what the issuer tagged - a plaintext's M2 and m3 never leave the user); each must be of the record's kind and
/// value, else `BadAttribute`.  `None`: stand-ins with the record's values (plaintext kinds get M2 = identity, m3 = 0), good for
/// exactly one thing - `ProofOfIssuance::verify`, which reads M1 only (src/amacs.rs:234-241).
fn issuance_from_record(kinds: &[u8], nr: usize, rec: &[u8], attributes: Option<Vec<Attribute>>) -> Result<CredentialIssuance, CredentialError> {
    let n = kinds.len();
    if rec.len() != 32 * (4 + nr + n) { return Err(CredentialError::WrongNumberOfBytes); }
    let amac = Amac { t: sc(rec, 0, 1, 0)?, U: pt(rec, 1, 1, 0)?, V: pt(rec, 2, 1, 0)? };
    let proof = CompactProof { challenge: sc(rec, 3, 1, 0)?, responses: (0..nr).map(|k| sc(rec, 4 + k, 1, 0)).collect::<Result<Vec<Scalar>, CredentialError>>()? };
    let attributes = match attributes {
        Some(mine) => {
            if mine.len() != n { return Err(CredentialError::WrongNumberOfAttributes); }
            for (k, a) in mine.iter().enumerate() {
                let (kind, value, _) = attribute_cells(a);
                if kind != kinds[k] || value != cell(rec, 4 + nr + k, 1, 0) { return Err(CredentialError::BadAttribute); }
            }
            mine
        }
        None => (0..n).map(|k| -> Result<Attribute, CredentialError> { Ok(match kinds[k] {
            ATTR_PUBLIC_SCALAR => Attribute::PublicScalar(sc(rec, 4 + nr + k, 1, 0)?),
            ATTR_SECRET_SCALAR => Attribute::SecretScalar(sc(rec, 4 + nr + k, 1, 0)?),
            ATTR_PUBLIC_POINT => Attribute::PublicPoint(pt(rec, 4 + nr + k, 1, 0)?),
            ATTR_EITHER_POINT => Attribute::EitherPoint(crate::symmetric::Plaintext { M1: pt(rec, 4 + nr + k, 1, 0)?, M2: RistrettoPoint::identity(), m3: Scalar::zero() }),
            _ => Attribute::SecretPoint(crate::symmetric::Plaintext { M1: pt(rec, 4 + nr + k, 1, 0)?, M2: RistrettoPoint::identity(), m3: Scalar::zero() }),
        }) }).collect::<Result<Vec<Attribute>, CredentialError>>()?,
    };
    Ok(CredentialIssuance { proof: ProofOfIssuance(proof), credential: AnonymousCredential { amac, attributes } })
}
/// `CredentialIssuance::from_bytes`: the issuer's answer (one AFXI v1 record) joined with the attributes of the request it answers.
pub fn issuance_from_bytes(bytes: &[u8], attributes: Vec<Attribute>) -> Result<CredentialIssuance, CredentialError> {
    let (kinds, nr, count, off) = afxi_parse(bytes).ok_or(CredentialError::WrongNumberOfBytes)?;
    if count != 1 { return Err(CredentialError::WrongNumberOfBytes); }
    issuance_from_record(&kinds, nr, &bytes[off..], Some(attributes))
}

impl GpuIssuer {
    /// `Issuer::verify` over presentations as they come off the network: `stream` = AFXP v1 sections back to back, in arrival order
    /// (one per presentation - `CompressedPresentation::as_bytes`, `presentation_to_bytes` - or per same-shape run); the library groups
    /// them by shape, verifies every group on the GPU and answers in stream order.  Nothing is decompressed or compressed on the
    /// host.  `Err(WrongNumberOfBytes)`: the stream does not parse (nothing was verified).  On an engine fault the stream is parsed
    /// here and every presentation checked by the crate's own `verify`.
    pub fn verify_wire(&self, stream: &[u8]) -> Result<Vec<Result<(), CredentialError>>, CredentialError> {
        match self.try_verify_wire(stream) {
            Ok(r) => r,
            Err(f) => {
                self.fell_through(f);
                let mut out = Vec::new();
                let mut at = 0usize;
                while at < stream.len() {
                    let (shape, count, off, cells, len) = afxp_parse(&stream[at..]).ok_or(CredentialError::WrongNumberOfBytes)?;
                    for i in 0..count {
                        let rec = &stream[at + off + 32 * cells * i..at + off + 32 * cells * (i + 1)];
                        out.push(presentation_from_record(&shape, rec).and_then(|p| p.verify(&self.fallback)));
                    }
                    at += len;
                }
                Ok(out)
            }
        }
    }

    /// The same, with an engine fault handed back.
    pub fn try_verify_wire(&self, stream: &[u8]) -> Result<Result<Vec<Result<(), CredentialError>>, CredentialError>, EngineFault> {
        let cap = stream.len() / 32 + 1;                              // a record is at least one cell
        let mut status = vec![ST_VERIFICATION_FAILURE; cap];
        let mut count = 0usize;
        let rc = unsafe {
            if self.group.is_null() { afx_verify_presentations_mixed_wire(self.ctx, stream.as_ptr(), stream.len(), status.as_mut_ptr(), cap, &mut count) }
            else { afx_group_verify_presentations_mixed_wire(self.group, stream.as_ptr(), stream.len(), status.as_mut_ptr(), cap, &mut count) }
        };
        if rc != 0 {
            if let Some(f) = fault_of(rc) { return Err(f); }
            return Ok(Err(if rc == E_BAD_ARGS { CredentialError::WrongNumberOfBytes } else { engine_error(rc, Op::Verify) }));
        }
        Ok(Ok(status.iter().take(count).map(|s| if *s == ST_OK { Ok(()) } else { Err(CredentialError::VerificationFailure) }).collect()))
    }

    /// `verify_wire` over presentations that were parsed (framing only) on arrival.
    pub fn verify_compressed(&self, batch: &[CompressedPresentation]) -> Result<Vec<Result<(), CredentialError>>, CredentialError> {
        let mut stream = Vec::with_capacity(batch.iter().map(|p| p.bytes.len()).sum());
        for p in batch.iter() { stream.extend_from_slice(&p.bytes); }
        self.verify_wire(&stream)
    }
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
