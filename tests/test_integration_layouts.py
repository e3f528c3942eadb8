"""CPU-only: the reference-side binding (integration/aeonflux_gpu.rs, source only - no Rust toolchain in this image) must
mirror the C ABI.  Parses every `#[repr(C)]` struct and every `extern "C"` declaration of the .rs file and the matching
typedef / prototype of include/aeonflux_gpu.h, and compares field names, order and widths, and argument counts and kinds."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RUST_TO_C = {"AfxShape": "afx_shape", "AfxEncProofSoa": "afx_encproof_soa", "AfxPresentationSoa": "afx_presentation_soa",
             "AfxAttributesSoa": "afx_attributes_soa", "AfxIssueRandomness": "afx_issue_randomness", "AfxIssuanceSoa": "afx_issuance_soa",
             "AfxCredentialsSoa": "afx_credentials_soa", "AfxKeypairsSoa": "afx_keypairs_soa", "AfxShowRandomness": "afx_show_randomness",
             "AfxEncProofOut": "afx_encproof_out", "AfxPresentationOut": "afx_presentation_out",
             "AfxPresentationGroup": "afx_presentation_group", "AfxIssueGroup": "afx_issue_group", "AfxIssuanceGroup": "afx_issuance_group",
             "AfxShowGroup": "afx_show_group"}


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def c_structs():
    src = strip_c_comments(open(os.path.join(ROOT, "include", "aeonflux_gpu.h")).read())
    out = {}
    for body, name in re.findall(r"typedef\s+struct(?:\s+\w+)?\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(\w+(?:\s*,\s*\w+)*)(?:\[(\w+)\])?$", decl)
            assert m, decl
            const, ty, ptr, fnames, arr = m.groups()
            if ptr:
                kind = "ptr"
            else:
                # a scalar, or another struct of this header held by value
                kind = {"uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "uint64_t": "u64", "size_t": "usize"}.get(ty, "struct " + ty)
                if arr:
                    kind = "[%s;%s]" % (kind, {"AFX_MAX_ATTRIBUTES": "32"}.get(arr, arr))
            for fname in fnames.split(","):
                fields.append((fname.strip(), kind))
        out[name] = fields
    return out


def rust_structs():
    src = re.sub(r"//.*", "", open(os.path.join(ROOT, "integration", "aeonflux_gpu.rs")).read())
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for decl in body.split(","):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"pub (\w+): (.+)$", decl)
            assert m, decl
            fname, ty = m.groups()
            if ty.startswith("*"):
                kind = "ptr"
            else:
                a = re.match(r"\[(\w+); (\w+)\]$", ty)
                kind = "[%s;%s]" % (a.group(1), {"AFX_MAX_ATTRIBUTES": "32"}.get(a.group(2), a.group(2))) if a else ("struct " + RUST_TO_C[ty] if ty in RUST_TO_C else ty)
            fields.append((fname, kind))
        out[name] = fields
    return out


def test_repr_c_structs_match_the_header():
    cs, rs = c_structs(), rust_structs()
    assert set(rs) == set(RUST_TO_C), sorted(set(rs) ^ set(RUST_TO_C))
    for rname, cname in RUST_TO_C.items():
        assert cname in cs, cname
        assert rs[rname] == cs[cname], (rname, rs[rname], cs[cname])
    # every data struct of the header that a batch call takes is bound (afx_plan_stats is a measurement aid)
    assert set(cs) - set(RUST_TO_C.values()) <= {"afx_plan_stats", "afx_coalescing_stats", "afx_plan_cache_stats"}, set(cs) - set(RUST_TO_C.values())


def c_prototypes():
    src = strip_c_comments(open(os.path.join(ROOT, "include", "aeonflux_gpu.h")).read())
    out = {}
    for ret, name, args in re.findall(r"\b(int|void|uint32_t|size_t|const char\*|afx_ctx\*)\s+(afx_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        kinds = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a in ("", "void"):
                continue
            if "*" in a or "[" in a:
                kinds.append("ptr")
            else:
                kinds.append({"int": "i32", "uint32_t": "u32", "uint16_t": "u16", "size_t": "usize"}[a.split()[-2] if len(a.split()) > 1 else a])
        out[name] = (ret, kinds)
    return out


def rust_externs():
    src = re.sub(r"//.*", "", open(os.path.join(ROOT, "integration", "aeonflux_gpu.rs")).read())
    block = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"fn (afx_\w+)\s*\((.*?)\)\s*(->\s*[\w*]+(?:\s+\w+)?)?\s*;", block, flags=re.S):
        kinds = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a:
                continue
            ty = a.split(":", 1)[1].strip()
            kinds.append("ptr" if ty.startswith("*") else ty)
        out[name] = (ret.replace("->", "").strip() if ret else "void", kinds)
    return out


def test_extern_declarations_match_the_header():
    cp, rx = c_prototypes(), rust_externs()
    assert {"afx_ctx_create", "afx_group_create", "afx_verify_presentations_mixed", "afx_group_verify_presentations_mixed", "afx_issue_mixed",
            "afx_group_issue_mixed", "afx_show_mixed", "afx_group_show_mixed", "afx_verify_issuances_mixed", "afx_group_verify_issuances_mixed",
            # the wire door: serialized presentations and issuances straight to the engine
            "afx_verify_presentations_mixed_wire", "afx_group_verify_presentations_mixed_wire", "afx_verify_issuances_wire"} <= set(rx)
    for name, (ret, kinds) in rx.items():
        assert name in cp, name
        cret, ckinds = cp[name]
        assert kinds == ckinds, (name, kinds, ckinds)
        assert {"int": "i32", "void": "void", "uint32_t": "u32", "afx_ctx*": "ptr"}[cret] == ("ptr" if ret.startswith("*") else ret), (name, cret, ret)


def rust_code():
    """the shim without comments (line comments only: the file has no block comments)"""
    src = open(os.path.join(ROOT, "integration", "aeonflux_gpu.rs")).read()
    assert "/*" not in src
    return re.sub(r"//.*", "", src)


def squeeze(text):
    """whitespace-insensitive form of a signature: every run of whitespace removed around punctuation, single spaces elsewhere"""
    return re.sub(r"\s*([(),<>:&])\s*", r"\1", " ".join(text.split()))


# The four entry points of the drop-in boundary, copied from the reference (cited lines; /root/reference is not read at test time).
REFERENCE_SIGNATURES = {
    # /root/reference/src/issuer.rs:111-118
    "Issuer::issue": """pub fn issue<C>(
        &self,
        request: CredentialRequest,
        csprng: &mut C,
    ) -> Result<CredentialIssuance, CredentialError>
    where
        C: CryptoRng + RngCore,
    {""",
    # /root/reference/src/issuer.rs:141-145
    "Issuer::verify": """pub fn verify(
        &self,
        presentation: &ProofOfValidCredential,
    ) -> Result<(), CredentialError>
    {""",
    # /root/reference/src/credential.rs:37-44
    "AnonymousCredential::show": """pub fn show(
        &self,
        system_parameters: &SystemParameters,
        issuer_parameters: &IssuerParameters,
        keypair: Option<&SymmetricKeypair>,
        mut csprng: impl CryptoRng + RngCore,
    ) -> Result<ProofOfValidCredential, CredentialError>
    {""",
    # /root/reference/src/issuer.rs:48-53
    "CredentialIssuance::verify": """pub fn verify(
        self,
        system_parameters: &SystemParameters,
        issuer_parameters: &IssuerParameters,
    ) -> Result<AnonymousCredential, CredentialError>
    {""",
}
# which impl block of the shim carries each of them (`self` there plays the part it plays in the crate)
SHIM_IMPL = {"Issuer::issue": "impl GpuIssuer {", "Issuer::verify": "impl GpuIssuer {", "AnonymousCredential::show": "impl<'a> GpuCredential<'a> {",
             "CredentialIssuance::verify": "impl<'a> GpuIssuance<'a> {"}


def test_the_four_entry_points_have_the_references_signatures_text_for_text():
    code = rust_code()
    blocks = {}
    for m in re.finditer(r"^impl[^\n]*\{$", code, flags=re.M):
        end = code.index("\n}\n", m.end())
        blocks.setdefault(m.group(0), "")
        blocks[m.group(0)] += code[m.end():end]
    for name, sig in REFERENCE_SIGNATURES.items():
        body = blocks[SHIM_IMPL[name]]
        assert squeeze(sig) in squeeze(body), name
    ref_root = "/root/reference/src"
    if os.path.isdir(ref_root):   # in the build container only: the strings above ARE the reference's text
        for name, (path, sig) in {"Issuer::issue": ("issuer.rs", REFERENCE_SIGNATURES["Issuer::issue"]), "Issuer::verify": ("issuer.rs", REFERENCE_SIGNATURES["Issuer::verify"]),
                                  "AnonymousCredential::show": ("credential.rs", REFERENCE_SIGNATURES["AnonymousCredential::show"]),
                                  "CredentialIssuance::verify": ("issuer.rs", REFERENCE_SIGNATURES["CredentialIssuance::verify"])}.items():
            assert squeeze(sig) in squeeze(open(os.path.join(ref_root, path)).read()), name


def test_errors_are_values_and_the_shim_is_no_std():
    code = rust_code()
    # nothing a caller or a GPU fault can trigger aborts the process (the crate returns Result everywhere, src/errors.rs:73-89)
    for needle in ("assert!(", "assert_eq!(", "debug_assert!(", "panic!(", ".expect(", ".unwrap()", "unreachable!(", "unimplemented!(", "todo!("):
        assert needle not in code, needle
    # every engine return code is looked at and mapped
    assert code.count("if rc != 0") >= 8 and code.count("engine_error(rc, Op::") >= 8
    for op in ("Op::Create", "Op::Issue", "Op::Verify", "Op::Show", "Op::VerifyIssuance"):
        assert op in code, op
    # #![no_std] + alloc like the crate (src/lib.rs:11-35)
    assert "std::" not in code and "extern crate alloc;" in code and "alloc::collections::BTreeMap" in code
    # the staged issuer key is wiped like the crate's own copy (src/amacs.rs:64-82), and the handles may cross threads
    assert code.count("Wiped(issuer.amacs_key.to_bytes())") == 2 and "impl Drop for Wiped" in code
    for t in ("GpuIssuer", "GpuUser"):
        assert "unsafe impl Send for %s {}" % t in code and "unsafe impl Sync for %s {}" % t in code
    # mixed layouts go to the engine as groups; nothing requires one layout per call any more
    for fn in ("afx_issue_mixed", "afx_show_mixed", "afx_verify_issuances_mixed", "afx_verify_presentations_mixed"):
        assert "unsafe" in code and fn + "(self.ctx" in code, fn


def test_binding_covers_the_call_sites_and_documents_the_draw_order():
    src = open(os.path.join(ROOT, "integration", "aeonflux_gpu.rs")).read()
    for needle in ("pub fn verify_batch", "pub fn issue_batch", "pub fn show_batch", "pub fn verify_issuance_batch", "pub fn new_multi",
                   "pub fn install_issuer", "pub fn install_user", "pub fn issuer_engine", "pub fn user_engine",
                   "src/amacs.rs:289", "src/amacs.rs:290", "presentation.rs:162", "thread_rng()", "fn shape_key", "fn layout_key", "zeroize"):
        assert needle in src, needle
    # the shim draws from the caller's csprng only (`rand` is a dev-dependency of the crate) and never trusts batch[0]'s shape
    code = rust_code()
    assert "rand::" not in code and "thread_rng" not in code
    assert "afx_verify_presentations_mixed" in code and "by_shape" in code and "by_layout" in code
    # the draw order stated in the shim is the engine's input order: t_wide, U_wide, rng_seed / z_wide, rng_seed, enc_seeds
    cs = c_structs()
    assert [f for f, _ in cs["afx_issue_randomness"]] == ["t_wide", "U_wide", "rng_seed"]
    assert [f for f, _ in cs["afx_show_randomness"]] == ["z_wide", "rng_seed", "enc_seeds"]
    # INTEGRATION.md shows the delegation patch for each of the four methods
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for needle in ("crate::gpu::issuer_engine(self)", "crate::gpu::user_engine(system_parameters, issuer_parameters)", "pub(crate)"):
        assert needle in doc, needle


def test_engine_faults_fall_through_to_the_crates_own_body():
    """AFX_E_NO_DEVICE / AFX_E_HIP / AFX_E_NO_MEMORY never become a cryptographic verdict: the try_* forms hand them back, every other
    form runs the crate's own code for that call (Missing 3)"""
    code = rust_code()
    assert "pub struct EngineFault(pub i32);" in code
    assert "if rc == E_NO_DEVICE || rc == E_HIP || rc == E_NO_MEMORY { Some(EngineFault(rc)) } else { None }" in code
    hdr = open(os.path.join(ROOT, "include", "aeonflux_gpu.h")).read()
    for name, val in (("E_NO_DEVICE", -3), ("E_HIP", -4), ("E_NO_MEMORY", -6), ("E_BAD_ARGS", -1), ("E_BAD_PARAMS", -2), ("E_NO_KEY", -5)):
        assert "const %s: i32 = %d;" % (name, val) in code and re.search(r"#define AFX_%s \(%d\)" % (name, val), hdr), name
    # every engine call looks for a fault before it maps anything else
    assert code.count("fault_of(rc)") >= 6   # issue, verify, show, verify_issuance, verify_wire, verify_issuances_wire
    for fn in ("pub fn try_issue<", "pub fn try_issue_batch<", "pub fn try_verify(", "pub fn try_verify_batch(", "pub fn try_show<", "pub fn try_show_batch<",
               "pub fn try_verify_issuance_batch(", "pub fn try_verify_wire(", "pub fn try_verify_issuances_wire(", "pub fn fall_throughs(", "pub fn last_fault_code("):
        assert fn in code, fn
    # the crate's own bodies, as the reference has them (src/issuer.rs:119-123, :146, :54-56; src/credential.rs:45)
    for body in ("presentation.verify(&self.fallback)", "Amac::tag(csprng, &self.fallback.system_parameters, &self.fallback.amacs_key, &request.attributes)?",
                 "ProofOfIssuance::prove(&self.fallback, &cred)", "ProofOfValidCredential::prove(&system_parameters, &issuer_parameters, self.credential, keypair, &mut csprng)",
                 "iss.proof.verify(system_parameters, issuer_parameters, &iss.credential).and(Ok(iss.credential))"):
        assert squeeze(body) in squeeze(code), body
    # a fault is never mapped to a verification verdict on the serving paths: the only engine_error(.., Op::Verify*) sites follow a fault_of check
    for m in re.finditer(r"engine_error\(rc, Op::(Verify|VerifyIssuance|Issue|Show)\)", code):
        assert "fault_of(rc)" in code[max(0, m.start() - 400):m.start()], code[m.start() - 200:m.end()]
    assert "last_engine_code" not in code
    # INTEGRATION.md's patches call the try_* forms and fall through
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for needle in ("engine.try_verify(presentation)", "engine.try_issue(request, csprng)", "engine.issuance(self).try_verify()", "try_show(keypair, &mut csprng)",
                   "presentation.verify(&self)"):
        assert needle in doc, needle
    if os.path.isdir("/root/reference/src"):   # build container only: the bodies quoted above are the reference's
        ref = squeeze(open("/root/reference/src/issuer.rs").read())
        assert squeeze("presentation.verify(&self)") in ref and squeeze("let proof = ProofOfIssuance::prove(&self, &cred);") in ref
        assert squeeze(".verify(system_parameters, issuer_parameters, &self.credential) .and(Ok(self.credential))") in ref


def test_the_wire_door_is_bound_and_pinned():
    """to_bytes / from_bytes of the four message types write AFXP / AFXI v1, the wire entry points are bound, and the crate-side pin test
    reads the committed fixture (Missing 2 and 4)"""
    code = rust_code()
    for fn in ("pub fn presentation_to_bytes(", "pub fn presentation_from_bytes(", "pub fn encryption_proof_to_bytes(", "pub fn encryption_proof_from_bytes(",
               "pub fn issuance_to_bytes(", "pub fn issuance_from_bytes(", "pub fn issuer_parameters_to_bytes(", "pub fn issuer_parameters_from_bytes(",
               "pub struct CompressedPresentation", "pub fn verify_wire(", "pub fn verify_compressed(", "pub fn show_batch_wire<", "pub fn verify_issuances_wire("):
        assert fn in code, fn
    for call in ("afx_verify_presentations_mixed_wire(self.ctx", "afx_group_verify_presentations_mixed_wire(self.group", "afx_verify_issuances_wire(ctx"):
        assert call in code, call
    # the format constants are the header's
    assert 'b"AFXP"' in code and 'b"AFXI"' in code and "1 + shape.n_responses as usize + 3 + n + public + 14 * shape.n_enc_proofs as usize" in code
    hdr = open(os.path.join(ROOT, "include", "aeonflux_gpu.h")).read()
    assert "cells_per_record = 4 + n_responses + n_attributes" in hdr and "(4 + nr + n) as u32" in code
    # the wire paths decompress nothing: no pt( / decompress( between the engine call and the status mapping of try_verify_wire
    body = code[code.index("pub fn try_verify_wire("):code.index("pub fn verify_compressed(")]
    assert "decompress" not in body and "pt(" not in body and "compress()" not in body
    shown = code[code.index("fn record_of_shown("):code.index("fn record_of_shown(") + 700]
    assert "decompress" not in shown and "pt(" not in shown
    # the pin file: reads the committed line fixture, judges with the crate's own verify, exports crate-made flows
    pin = open(os.path.join(ROOT, "integration", "pin_against_crate.rs")).read()
    for needle in ("AFX_PIN_FIXTURE", "AFX_PIN_EXPORT", "fn pin_crate_verdicts_on_oracle_made_flows()", "fn pin_export_crate_made_flows()", "fn pin_wire_bytes()",
                   "iss.verify(&issuer.system_parameters, &issuer.issuer_parameters)", "issuer.verify(&p)", "thread_rng()", "presentation_to_bytes(&p)", "issuance_to_bytes(&iss)"):
        assert needle in pin, needle
    # every key the pin file reads is a key the fixture has
    fixture = open(os.path.join(ROOT, "tests", "golden", "flows.pin.txt")).read()
    keys = set(ln.split(" ", 1)[0] for ln in fixture.splitlines() if ln and not ln.startswith("#"))
    for k in re.findall(r'f\.(?:s|num|nums|bytes|list|has)\("([a-z_.A-Z0-9]+)"\)', pin):
        assert k in keys, k
    for k in ("challenge", "responses", "pk", "E1", "E2", "C_y_1", "C_y_2", "C_y_3", "C_y_2p", "index"):
        assert "present.enc.0." + k in keys
    assert "flows_from_crate.json" in open(os.path.join(ROOT, "INTEGRATION.md")).read()
