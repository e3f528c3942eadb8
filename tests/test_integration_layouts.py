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
    assert set(cs) - set(RUST_TO_C.values()) <= {"afx_plan_stats"}, set(cs) - set(RUST_TO_C.values())


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
            "afx_group_issue_mixed", "afx_show_mixed", "afx_group_show_mixed", "afx_verify_issuances_mixed", "afx_group_verify_issuances_mixed"} <= set(rx)
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
