// Pins the engine's CPU oracle (and through it the GPU engine) against the aeonflux crate ITSELF, in both directions.
// SOURCE ONLY, like aeonflux_gpu.rs: this image has no Rust toolchain; a maintainer with the crate's nightly toolchain runs it.
//
// How to run:
//   1. copy this file to the crate as src/pin.rs and add to src/lib.rs:      #[cfg(test)] mod pin;
//      (the `pub(crate)` widening of INTEGRATION.md section 1 must be in: the fields of ProofOfValidCredential, ProofOfEncryption and
//      the tuple field of ProofOfIssuance are private to their modules today; `mod gpu` (aeonflux_gpu.rs) is needed only for the
//      `pin_wire_*` test - build with `--features gpu`, or delete that test)
//   2. AFX_PIN_FIXTURE=/path/to/aeonflux-amd/tests/golden/flows.pin.txt AFX_PIN_EXPORT=/path/to/aeonflux-amd/tests/golden/flows_from_crate.json \
//        cargo +nightly test pin_
//   3. back in the engine's repository: python -m pytest tests/test_pin_fixture.py            (and with -m gpu on an MI355X)
//
// What it pins:
//   pin_crate_verdicts_on_oracle_made_flows   (a) every flow of tests/golden/flows.json - issued, shown and (some) damaged by the ORACLE with
//       recorded randomness - is rebuilt as the crate's own structs from its hex and handed to the crate's own
//       `CredentialIssuance::verify` / `Issuer::verify`: the crate's accept / reject must equal the recorded one, damaged flows
//       included.  The oracle's PROVERS are then pinned: what they make, the reference accepts (and rejects when it should).
//   pin_export_crate_made_flows               (b) the crate's own `issue` -> `verify` -> `show` with thread_rng(), the layouts of its own tests
//       (src/nizk/presentation.rs:461-638, src/nizk/issuance.rs:233-295), written out in the schema of flows.json together with
//       the crate's verdicts on them and on damaged copies.  tests/test_pin_fixture.py feeds that file to the oracle's verifiers and
//       (-m gpu) to the engine's: the oracle's and the GPU's VERIFIERS are then pinned on proofs the reference made.
//   pin_wire_bytes                            the Rust writers of aeonflux_gpu.rs (`presentation_to_bytes`, `issuance_to_bytes`) reproduce, byte for
//       byte, what the library's C packers wrote for the same flow (the `*.afxp` / `*.afxi` lines of the fixture), and the
//       readers take them back.
//
// Together (a) and (b) move SURVEY.md section 8c from "parity unpinned" to pinned: nothing in this file trusts the oracle.

extern crate std;

use std::collections::HashMap;
use std::prelude::v1::*;
use std::string::String;

use curve25519_dalek::ristretto::{CompressedRistretto, RistrettoPoint};
use curve25519_dalek::scalar::Scalar;
use rand::thread_rng;
use zkp::CompactProof;

use crate::amacs::{Amac, Attribute, EncryptedAttribute, SecretKey};
use crate::credential::AnonymousCredential;
use crate::issuer::{CredentialIssuance, Issuer};
use crate::nizk::encryption::ProofOfEncryption;
use crate::nizk::issuance::ProofOfIssuance;
use crate::nizk::presentation::ProofOfValidCredential;
use crate::parameters::{IssuerParameters, SystemParameters};
use crate::symmetric::{Ciphertext, Keypair as SymmetricKeypair, Plaintext, PublicKey as SymmetricPublicKey};
use crate::user::CredentialRequestConstructor;

// ---- hex and the line fixture -----------------------------------------------------------------------------------------------------
fn unhex(s: &str) -> Vec<u8> {
    let b = s.as_bytes();
    let nib = |c: u8| -> u8 { match c { b'0'..=b'9' => c - b'0', b'a'..=b'f' => c - b'a' + 10, b'A'..=b'F' => c - b'A' + 10, _ => panic!("not hex") } };
    assert!(b.len() % 2 == 0);
    (0..b.len() / 2).map(|i| (nib(b[2 * i]) << 4) | nib(b[2 * i + 1])).collect()
}
fn hex(b: &[u8]) -> String {
    let mut s = String::with_capacity(2 * b.len());
    for x in b { s.push_str(&format!("{:02x}", x)); }
    s
}
fn arr32(b: &[u8]) -> [u8; 32] { let mut a = [0u8; 32]; a.copy_from_slice(&b[..32]); a }
fn scalar(b: &[u8]) -> Scalar { Scalar::from_canonical_bytes(arr32(b)).expect("canonical scalar") }
fn point(b: &[u8]) -> RistrettoPoint { CompressedRistretto(arr32(b)).decompress().expect("valid point") }
// a field of a damaged flow may hold bytes the crate's types cannot (a non-canonical scalar, an undecodable point): such a flow
// cannot be rebuilt as crate structs at all - the crate would have rejected it while parsing - and counts as "rejected"
fn try_scalar(b: &[u8]) -> Option<Scalar> { Scalar::from_canonical_bytes(arr32(b)) }
fn try_point(b: &[u8]) -> Option<RistrettoPoint> { CompressedRistretto(arr32(b)).decompress() }

/// One flow of the fixture: key -> the rest of its line.
struct Flow { name: String, f: HashMap<String, String> }
impl Flow {
    fn has(&self, k: &str) -> bool { self.f.contains_key(k) }
    fn s(&self, k: &str) -> &str { self.f.get(k).unwrap_or_else(|| panic!("flow {}: no {}", self.name, k)).as_str() }
    fn num(&self, k: &str) -> usize { self.s(k).trim().parse().unwrap() }
    fn nums(&self, k: &str) -> Vec<usize> { self.s(k).split_whitespace().map(|x| x.parse().unwrap()).collect() }
    fn bytes(&self, k: &str) -> Vec<u8> { unhex(self.s(k).trim()) }
    fn list(&self, k: &str) -> Vec<Vec<u8>> { self.s(k).split_whitespace().map(unhex).collect() }
}
fn load_flows() -> Vec<Flow> {
    let path = std::env::var("AFX_PIN_FIXTURE").expect("set AFX_PIN_FIXTURE to aeonflux-amd's tests/golden/flows.pin.txt");
    let text = std::fs::read_to_string(&path).expect("read the fixture");
    let mut flows = Vec::new();
    let mut cur: Option<Flow> = None;
    for line in text.lines() {
        if line.starts_with('#') || line.trim().is_empty() { continue; }
        let (key, rest) = match line.find(' ') { Some(i) => (&line[..i], &line[i + 1..]), None => (line, "") };
        if key == "flow" { cur = Some(Flow { name: rest.trim().to_string(), f: HashMap::new() }); continue; }
        if key == "end" { flows.push(cur.take().expect("end without flow")); continue; }
        cur.as_mut().expect("line outside a flow").f.insert(key.to_string(), rest.to_string());
    }
    assert!(!flows.is_empty());
    flows
}

// ---- the crate's structs from bytes ------------------------------------------------------------------------------------------------
/// amacs::SecretKey::to_bytes layout (src/amacs.rs:110-125): u32 n | w | w' | x_0 | x_1 | y[n] | W.  Built field by field: the crate's
/// own `from_bytes` re-reads ONE chunk for every y_i (src/amacs.rs:148-150), so it cannot load a key whose y_i differ.
fn secret_key(b: &[u8]) -> SecretKey {
    let n = u32::from_le_bytes([b[0], b[1], b[2], b[3]]) as usize;
    assert_eq!(b.len(), 4 + 32 * (5 + n));
    let c = |i: usize| &b[4 + 32 * i..4 + 32 * (i + 1)];
    SecretKey { w: scalar(c(0)), w_prime: scalar(c(1)), x_0: scalar(c(2)), x_1: scalar(c(3)), y: (0..n).map(|i| scalar(c(4 + i))).collect(), W: point(c(4 + n)) }
}
fn issuer_of(f: &Flow) -> Issuer {
    let system_parameters = SystemParameters::from_bytes(&f.bytes("params")).expect("SystemParameters::from_bytes");
    let ip = f.bytes("issuer_params");                                    // C_W || I (src/issuer.rs:155,163; the crate's from_bytes is unimplemented!())
    Issuer { system_parameters, issuer_parameters: IssuerParameters { C_W: point(&ip[..32]), I: point(&ip[32..]) }, amacs_key: secret_key(&f.bytes("key")) }
}
/// kinds + 96-byte records (value | M2 | m3) -> amacs::Attribute (src/amacs.rs:168-179); None: a value the crate's types cannot hold
fn attributes_of(kinds: &[usize], records: &[Vec<u8>]) -> Option<Vec<Attribute>> {
    kinds.iter().zip(records.iter()).map(|(k, r)| Some(match *k {
        0 => Attribute::PublicScalar(try_scalar(&r[..32])?),
        1 => Attribute::SecretScalar(try_scalar(&r[..32])?),
        2 => Attribute::PublicPoint(try_point(&r[..32])?),
        3 => Attribute::EitherPoint(Plaintext { M1: try_point(&r[..32])?, M2: try_point(&r[32..64])?, m3: try_scalar(&r[64..96])? }),
        4 => Attribute::SecretPoint(Plaintext { M1: try_point(&r[..32])?, M2: try_point(&r[32..64])?, m3: try_scalar(&r[64..96])? }),
        _ => panic!("attribute kind"),
    })).collect()
}
fn issuance_of(f: &Flow) -> Option<CredentialIssuance> {
    let attributes = attributes_of(&f.nums("issue.kinds"), &f.list("issue.values"))?;
    let responses = f.list("issue.responses").iter().map(|r| try_scalar(r)).collect::<Option<Vec<Scalar>>>()?;
    Some(CredentialIssuance {
        proof: ProofOfIssuance(CompactProof { challenge: try_scalar(&f.bytes("issue.challenge"))?, responses }),
        credential: AnonymousCredential { amac: Amac { t: try_scalar(&f.bytes("issue.t"))?, U: try_point(&f.bytes("issue.U"))?, V: try_point(&f.bytes("issue.V"))? }, attributes },
    })
}
fn presentation_of(f: &Flow) -> Option<ProofOfValidCredential> {
    let kinds = f.nums("present.kinds");
    let values = f.list("present.attr_values");
    let mut encrypted_attributes = Vec::new();
    for (k, v) in kinds.iter().zip(values.iter()) {
        encrypted_attributes.push(match *k {
            0 => EncryptedAttribute::PublicScalar(try_scalar(v)?),
            1 => EncryptedAttribute::SecretScalar,
            2 => EncryptedAttribute::PublicPoint(try_point(v)?),
            3 => EncryptedAttribute::SecretPoint,
            _ => panic!("encrypted attribute kind"),
        });
    }
    let mut proofs_of_encryption = Vec::new();
    for e in 0..f.num("present.enc") {
        let g = |name: &str| f.bytes(&format!("present.enc.{}.{}", e, name));
        let index = f.num(&format!("present.enc.{}.index", e)) as u16;
        let responses = f.list(&format!("present.enc.{}.responses", e)).iter().map(|r| try_scalar(r)).collect::<Option<Vec<Scalar>>>()?;
        proofs_of_encryption.push((index, ProofOfEncryption {
            proof: CompactProof { challenge: try_scalar(&g("challenge"))?, responses },
            public_key: SymmetricPublicKey { pk: try_point(&g("pk"))? },
            ciphertext: Ciphertext { E1: try_point(&g("E1"))?, E2: try_point(&g("E2"))? },
            index,
            C_y_1: try_point(&g("C_y_1"))?, C_y_2: try_point(&g("C_y_2"))?, C_y_3: try_point(&g("C_y_3"))?, C_y_2_prime: try_point(&g("C_y_2p"))?,
        }));
    }
    Some(ProofOfValidCredential {
        proof: CompactProof { challenge: try_scalar(&f.bytes("present.challenge"))?,
                              responses: f.list("present.responses").iter().map(|r| try_scalar(r)).collect::<Option<Vec<Scalar>>>()? },
        proofs_of_encryption, encrypted_attributes,
        hidden_scalar_indices: f.nums("present.hidden").iter().map(|h| *h as u16).collect(),
        C_x_0: try_point(&f.bytes("present.C_x_0"))?, C_x_1: try_point(&f.bytes("present.C_x_1"))?, C_V: try_point(&f.bytes("present.C_V"))?,
        C_y: f.list("present.C_y").iter().map(|y| try_point(y)).collect::<Option<Vec<RistrettoPoint>>>()?,
    })
}

// ---- (a) the crate's verdicts on what the oracle made ---------------------------------------------------------------------------------
#[test]
fn pin_crate_verdicts_on_oracle_made_flows() {
    let (mut issuances, mut presentations) = (0, 0);
    for f in load_flows().iter() {
        let issuer = issuer_of(f);
        if f.has("issuance_verify") {
            // CredentialIssuance::verify (src/issuer.rs:48-57); an identity among the allocated points panics nothing: zkp rejects it
            let crate_says = match issuance_of(f) {
                Some(iss) => if iss.verify(&issuer.system_parameters, &issuer.issuer_parameters).is_ok() { 0 } else { 1 },
                None => 1,
            };
            assert_eq!(crate_says, f.num("issuance_verify"), "flow {}: CredentialIssuance::verify", f.name);
            issuances += 1;
        }
        if f.has("verify") {
            // Issuer::verify (src/issuer.rs:141-147).  The reference PANICS on shapes it indexes out of range on (presentation.rs:81,100,407);
            // the engine answers those with a failure status: a panic here counts as "rejected"
            let crate_says = match presentation_of(f) {
                Some(p) => match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| issuer.verify(&p).is_ok())) { Ok(true) => 0, _ => 1 },
                None => 1,
            };
            assert_eq!(crate_says, f.num("verify"), "flow {}: Issuer::verify", f.name);
            presentations += 1;
        }
    }
    assert!(issuances >= 10 && presentations >= 10, "the fixture is thinner than expected: {} issuances, {} presentations", issuances, presentations);
}

// ---- the Rust wire writers against the library's C packers ---------------------------------------------------------------------------
#[cfg(feature = "gpu")]
#[test]
fn pin_wire_bytes() {
    use crate::gpu::{issuance_from_bytes, issuance_to_bytes, presentation_from_bytes, presentation_to_bytes, CompressedPresentation};
    let (mut n_p, mut n_i) = (0, 0);
    for f in load_flows().iter() {
        if f.has("present.afxp") {
            if let Some(p) = presentation_of(f) {
                let want = f.bytes("present.afxp");
                assert_eq!(hex(&presentation_to_bytes(&p).expect("to_bytes")), hex(&want), "flow {}: AFXP bytes", f.name);
                let back = presentation_from_bytes(&want).expect("from_bytes");
                assert_eq!(hex(&presentation_to_bytes(&back).expect("to_bytes")), hex(&want), "flow {}: AFXP round trip", f.name);
                assert_eq!(hex(CompressedPresentation::from_bytes(&want).expect("framing").as_bytes()), hex(&want));
                assert!(CompressedPresentation::from_bytes(&want[..want.len() - 1]).is_err() && presentation_from_bytes(&want[32..]).is_err());
                n_p += 1;
            }
        }
        if f.has("issue.afxi") {
            if let Some(iss) = issuance_of(f) {
                let want = f.bytes("issue.afxi");
                assert_eq!(hex(&issuance_to_bytes(&iss).expect("to_bytes")), hex(&want), "flow {}: AFXI bytes", f.name);
                let attributes = attributes_of(&f.nums("issue.kinds"), &f.list("issue.values")).expect("attributes");
                let back = issuance_from_bytes(&want, attributes).expect("from_bytes");
                assert_eq!(hex(&issuance_to_bytes(&back).expect("to_bytes")), hex(&want), "flow {}: AFXI round trip", f.name);
                // somebody else's attributes do not fit the record
                let mut other = attributes_of(&f.nums("issue.kinds"), &f.list("issue.values")).expect("attributes");
                other[0] = match &other[0] { Attribute::PublicScalar(s) => Attribute::PublicScalar(s + Scalar::one()), _ => Attribute::PublicScalar(Scalar::one()) };
                assert!(issuance_from_bytes(&want, other).is_err());
                n_i += 1;
            }
        }
    }
    assert!(n_p >= 10 && n_i >= 10);
}

// ---- (b) what the crate makes, for the oracle and the GPU to verify -----------------------------------------------------------------------
fn record_of(a: &Attribute) -> (usize, Vec<u8>) {
    let mut r = vec![0u8; 96];
    let kind = match a {
        Attribute::PublicScalar(m) => { r[..32].copy_from_slice(m.as_bytes()); 0 }
        Attribute::SecretScalar(m) => { r[..32].copy_from_slice(m.as_bytes()); 1 }
        Attribute::PublicPoint(M) => { r[..32].copy_from_slice(M.compress().as_bytes()); 2 }
        Attribute::EitherPoint(p) | Attribute::SecretPoint(p) => {
            r[..32].copy_from_slice(p.M1.compress().as_bytes());
            r[32..64].copy_from_slice(p.M2.compress().as_bytes());
            r[64..].copy_from_slice(p.m3.as_bytes());
            if let Attribute::EitherPoint(_) = a { 3 } else { 4 }
        }
    };
    (kind, r)
}
fn jlist(items: &[String]) -> String { format!("[{}]", items.join(", ")) }
fn jhex(b: &[u8]) -> String { format!("\"{}\"", hex(b)) }
fn presentation_json(p: &ProofOfValidCredential) -> String {
    let kinds: Vec<String> = p.encrypted_attributes.iter().map(|a| match a {
        EncryptedAttribute::PublicScalar(_) => "0", EncryptedAttribute::SecretScalar => "1", EncryptedAttribute::PublicPoint(_) => "2", EncryptedAttribute::SecretPoint => "3" }.to_string()).collect();
    let values: Vec<String> = p.encrypted_attributes.iter().map(|a| match a {
        EncryptedAttribute::PublicScalar(m) => jhex(m.as_bytes()), EncryptedAttribute::PublicPoint(M) => jhex(M.compress().as_bytes()), _ => jhex(&[0u8; 32]) }).collect();
    let enc: Vec<String> = p.proofs_of_encryption.iter().map(|(_, q)| format!(
        "{{\"index\": {}, \"challenge\": {}, \"responses\": {}, \"pk\": {}, \"E1\": {}, \"E2\": {}, \"C_y_1\": {}, \"C_y_2\": {}, \"C_y_3\": {}, \"C_y_2p\": {}}}",
        q.index, jhex(q.proof.challenge.as_bytes()), jlist(&q.proof.responses.iter().map(|r| jhex(r.as_bytes())).collect::<Vec<String>>()),
        jhex(q.public_key.pk.compress().as_bytes()), jhex(q.ciphertext.E1.compress().as_bytes()), jhex(q.ciphertext.E2.compress().as_bytes()),
        jhex(q.C_y_1.compress().as_bytes()), jhex(q.C_y_2.compress().as_bytes()), jhex(q.C_y_3.compress().as_bytes()), jhex(q.C_y_2_prime.compress().as_bytes()))).collect();
    format!("{{\"n_attributes\": {}, \"n_responses\": {}, \"challenge\": {}, \"responses\": {}, \"C_x_0\": {}, \"C_x_1\": {}, \"C_V\": {}, \"C_y\": {}, \"kinds\": {}, \"attr_values\": {}, \
             \"hidden_scalar_indices\": {}, \"enc\": {}}}",
            p.encrypted_attributes.len(), p.proof.responses.len(), jhex(p.proof.challenge.as_bytes()),
            jlist(&p.proof.responses.iter().map(|r| jhex(r.as_bytes())).collect::<Vec<String>>()),
            jhex(p.C_x_0.compress().as_bytes()), jhex(p.C_x_1.compress().as_bytes()), jhex(p.C_V.compress().as_bytes()),
            jlist(&p.C_y.iter().map(|y| jhex(y.compress().as_bytes())).collect::<Vec<String>>()), jlist(&kinds), jlist(&values),
            jlist(&p.hidden_scalar_indices.iter().map(|h| h.to_string()).collect::<Vec<String>>()), jlist(&enc))
}

/// One layout of the crate's own tests, run with thread_rng(); `layout`: S scalar, P point, T plaintext ("tsunami"); `hide`: positions
/// hidden before the show.  Emits the honest flow and a damaged copy of its presentation and of its issuance, each with the
/// crate's own verdict.
fn crate_flow(name: &str, layout: &str, hide: &[usize], out: &mut Vec<String>) {
    let mut rng = thread_rng();
    let n = layout.len() as u32;
    let system_parameters = SystemParameters::generate(&mut rng, n).unwrap();
    let issuer = Issuer::new(&system_parameters, &mut rng);
    let mut request = CredentialRequestConstructor::new(&system_parameters);
    for c in layout.chars() {
        match c {
            'S' => request.append_revealed_scalar(Scalar::random(&mut rng)),
            'P' => request.append_revealed_point(RistrettoPoint::random(&mut rng)),
            _ => { let _ = request.append_plaintext(&String::from("This is a tsunami alert test..").into_bytes()); }
        }
    }
    let issuance = issuer.issue(request.finish(), &mut rng).unwrap();
    // the issuance's bytes, before `verify` consumes it
    let (kinds, records): (Vec<usize>, Vec<Vec<u8>>) = issuance.credential.attributes.iter().map(record_of).unzip();
    let amac = issuance.credential.amac.clone();
    let (ich, irs) = (issuance.proof.0.challenge, issuance.proof.0.responses.clone());
    let mut credential = issuance.verify(&system_parameters, &issuer.issuer_parameters).unwrap();
    for i in hide { credential.hide_attribute(*i).unwrap(); }
    let (keypair, _) = SymmetricKeypair::generate(&system_parameters, &mut rng);
    let shown = credential.show(&system_parameters, &issuer.issuer_parameters, Some(&keypair), &mut rng).unwrap();
    let key = issuer.amacs_key.to_bytes();
    let mut ip = issuer.issuer_parameters.C_W.compress().as_bytes().to_vec();
    ip.extend_from_slice(issuer.issuer_parameters.I.compress().as_bytes());
    let mut emit = |tag: &str, p: &ProofOfValidCredential, irs: &Vec<Scalar>| {
        let p_ok = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| issuer.verify(p).is_ok())).unwrap_or(false);
        let attributes = attributes_of(&kinds, &records).unwrap();
        let iss = CredentialIssuance { proof: ProofOfIssuance(CompactProof { challenge: ich, responses: irs.clone() }),
                                       credential: AnonymousCredential { amac: amac.clone(), attributes } };
        let i_ok = iss.verify(&system_parameters, &issuer.issuer_parameters).is_ok();
        out.push(format!(
            "{{\"name\": \"{}{}\", \"n\": {}, \"params\": {}, \"key\": {}, \"issuer_params\": {}, \
              \"issue\": {{\"kinds\": {}, \"values\": {}, \"status\": 0, \"t\": {}, \"U\": {}, \"V\": {}, \"challenge\": {}, \"responses\": {}}}, \
              \"issuance_verify\": {}, \"presentation\": {}, \"verify\": {}}}",
            name, tag, n, jhex(&system_parameters.to_bytes()), jhex(&key), jhex(&ip),
            jlist(&kinds.iter().map(|k| k.to_string()).collect::<Vec<String>>()), jlist(&records.iter().map(|r| jhex(r)).collect::<Vec<String>>()),
            jhex(amac.t.as_bytes()), jhex(amac.U.compress().as_bytes()), jhex(amac.V.compress().as_bytes()), jhex(ich.as_bytes()),
            jlist(&irs.iter().map(|r| jhex(r.as_bytes())).collect::<Vec<String>>()),
            if i_ok { 0 } else { 1 }, presentation_json(p), if p_ok { 0 } else { 1 }));
    };
    emit("", &shown, &irs);
    // damaged copies: a response of the presentation proof, and a response of the issuance proof, off by one
    let mut bad_p = presentation_of_json_roundtrip(&shown);
    bad_p.proof.responses[0] += Scalar::one();
    let mut bad_irs = irs.clone();
    bad_irs[1] += Scalar::one();
    emit("_damaged", &bad_p, &bad_irs);
}
/// a field-by-field copy (ProofOfValidCredential is not Clone)
fn presentation_of_json_roundtrip(p: &ProofOfValidCredential) -> ProofOfValidCredential {
    ProofOfValidCredential {
        proof: CompactProof { challenge: p.proof.challenge, responses: p.proof.responses.clone() },
        proofs_of_encryption: p.proofs_of_encryption.iter().map(|(i, q)| (*i, ProofOfEncryption {
            proof: CompactProof { challenge: q.proof.challenge, responses: q.proof.responses.clone() },
            public_key: SymmetricPublicKey { pk: q.public_key.pk }, ciphertext: Ciphertext { E1: q.ciphertext.E1, E2: q.ciphertext.E2 }, index: q.index,
            C_y_1: q.C_y_1, C_y_2: q.C_y_2, C_y_3: q.C_y_3, C_y_2_prime: q.C_y_2_prime })).collect(),
        encrypted_attributes: p.encrypted_attributes.clone(),
        hidden_scalar_indices: p.hidden_scalar_indices.clone(),
        C_x_0: p.C_x_0, C_x_1: p.C_x_1, C_V: p.C_V, C_y: p.C_y.clone(),
    }
}

#[test]
fn pin_export_crate_made_flows() {
    let path = match std::env::var("AFX_PIN_EXPORT") { Ok(p) => p, Err(_) => { std::eprintln!("AFX_PIN_EXPORT not set: nothing exported"); return; } };
    let mut flows = Vec::new();
    // src/nizk/presentation.rs:461-638 and src/nizk/issuance.rs:233-295, plus the benchmark shapes of BASELINE.json (hidden points trailing)
    crate_flow("crate_10_attributes", "PPSSPSPSSP", &[], &mut flows);
    crate_flow("crate_10_attributes_with_plaintext", "TPSSPSPSSP", &[], &mut flows);
    crate_flow("crate_1_plaintext_hidden", "T", &[0], &mut flows);
    crate_flow("crate_1_scalar_revealed", "S", &[], &mut flows);
    crate_flow("crate_readme_sSPe", "SSPT", &[0, 3], &mut flows);
    crate_flow("crate_c3_SSPPeeee", "SSPPTTTT", &[4, 5, 6, 7], &mut flows);
    crate_flow("crate_hidden_scalars", "SSSSPP", &[0, 2, 3], &mut flows);
    let doc = format!("{{\"_source\": \"made by the aeonflux crate itself (integration/pin_against_crate.rs, thread_rng()): inputs for the oracle's and the GPU's verifiers\", \
                        \"flows\": [\n{}\n]}}\n", flows.join(",\n"));
    std::fs::write(&path, doc).expect("write the export");
}
