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
             "AfxPresentationGroup": "afx_presentation_group"}


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
    assert {"afx_ctx_create", "afx_verify_presentations", "afx_issue", "afx_show", "afx_verify_issuances", "afx_group_create",
            "afx_group_verify_presentations", "afx_group_issue", "afx_group_show", "afx_group_verify_issuances",
            "afx_verify_presentations_mixed", "afx_group_verify_presentations_mixed"} <= set(rx)
    for name, (ret, kinds) in rx.items():
        assert name in cp, name
        cret, ckinds = cp[name]
        assert kinds == ckinds, (name, kinds, ckinds)
        assert {"int": "i32", "void": "void", "uint32_t": "u32", "afx_ctx*": "ptr"}[cret] == ("ptr" if ret.startswith("*") else ret), (name, cret, ret)


def test_binding_covers_the_three_call_sites_and_documents_the_draw_order():
    src = open(os.path.join(ROOT, "integration", "aeonflux_gpu.rs")).read()
    for needle in ("pub fn verify_batch", "pub fn issue_batch", "pub fn show_batch", "pub fn verify_issuance_batch", "pub fn new_multi",
                   "src/amacs.rs:289", "src/amacs.rs:290", "presentation.rs:162", "thread_rng()", "fn shape_key", "zeroize"):
        assert needle in src, needle
    # the shim draws from the caller's csprng only (`rand` is a dev-dependency of the crate) and never trusts batch[0]'s shape
    code = re.sub(r"//.*", "", src)
    assert "rand::" not in code and "thread_rng" not in code
    assert "afx_verify_presentations_mixed" in code and "by_shape" in code
    # the draw order stated in the shim is the engine's input order: t_wide, U_wide, rng_seed / z_wide, rng_seed, enc_seeds
    cs = c_structs()
    assert [f for f, _ in cs["afx_issue_randomness"]] == ["t_wide", "U_wide", "rng_seed"]
    assert [f for f, _ in cs["afx_show_randomness"]] == ["z_wide", "rng_seed", "enc_seeds"]
