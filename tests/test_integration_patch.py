"""CPU-only: integration/aeonflux_gpu.patch - the reference-side patch of INTEGRATION.md section 1 (the `gpu` feature and build script,
`mod gpu`, the four delegating prologues, the pub(crate) widening) - applies cleanly to the crate and leaves the crate's own bodies in
place as the fall-through.  Needs the crate's sources (/root/reference, or AFX_REFERENCE_DIR): skipped where they are absent (the GPU
box).  Nothing here compiles Rust (no toolchain in this image); what is checked is that step 1 of the maintainer's procedure is
`patch -p1 < integration/aeonflux_gpu.patch` and not an afternoon of hand edits."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AFX_REFERENCE_DIR", "/root/reference")
PATCH = os.path.join(ROOT, "integration", "aeonflux_gpu.patch")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src", "issuer.rs")), reason="the crate's sources are not here")


@pytest.fixture()
def crate(tmp_path):
    dst = tmp_path / "aeonflux"
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git", "target"))
    return dst


def test_the_patch_applies_cleanly_and_keeps_the_crates_bodies(crate):
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=crate, capture_output=True, text=True)
    assert dry.returncode == 0 and "FAILED" not in dry.stdout and "fuzz" not in dry.stdout and "offset" not in dry.stdout, dry.stdout + dry.stderr
    before = {f: (crate / f).read_text() for f in ("src/issuer.rs", "src/credential.rs", "src/lib.rs", "Cargo.toml")}
    r = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=crate, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    issuer, cred = (crate / "src/issuer.rs").read_text(), (crate / "src/credential.rs").read_text()
    # the crate's own bodies are still there, after the delegation: what runs without an engine, and when the accelerator fails
    for body in ("self.proof\n            .verify(system_parameters, issuer_parameters, &self.credential)\n            .and(Ok(self.credential))",
                 "let amac = Amac::tag(csprng, &self.system_parameters, &self.amacs_key, &request.attributes)?;",
                 "let proof = ProofOfIssuance::prove(&self, &cred);",
                 "presentation.verify(&self)"):
        assert body in before["src/issuer.rs"] and body in issuer, body
        assert issuer.index("crate::gpu::") < issuer.rindex(body)
    show_body = "ProofOfValidCredential::prove(&system_parameters, &issuer_parameters, &self, keypair, &mut csprng)"
    assert show_body in before["src/credential.rs"] and show_body in cred and cred.index("crate::gpu::user_engine") < cred.index(show_body)
    # only additions in the four methods' files: every line of the originals survives, in order
    for f in ("src/issuer.rs", "src/credential.rs", "src/lib.rs", "Cargo.toml"):
        old, new = before[f].split("\n"), (crate / f).read_text().split("\n")
        it = iter(new)
        assert all(any(l == m for m in it) for l in old), f
    # every delegation is behind the feature, and the feature, the link name and the build script exist
    assert issuer.count('#[cfg(feature = "gpu")]') == 3 and cred.count('#[cfg(feature = "gpu")]') == 1
    assert '#[cfg(feature = "gpu")]\npub mod gpu;' in (crate / "src/lib.rs").read_text()
    cargo = (crate / "Cargo.toml").read_text()
    assert 'gpu = [ "alloc" ]' in cargo and 'links = "aeonflux_gpu"' in cargo and 'build = "build.rs"' in cargo
    assert "rustc-link-lib=dylib=aeonflux_gpu" in (crate / "build.rs").read_text()
    # the widening: every field the shim touches is pub(crate) now, nothing became pub
    pres, enc, iss = ((crate / "src/nizk" / f).read_text() for f in ("presentation.rs", "encryption.rs", "issuance.rs"))
    s = pres[pres.index("pub struct ProofOfValidCredential {"):]
    s = s[:s.index("}")]
    assert len(re.findall(r"^    pub\(crate\) \w+:", s, re.M)) == 8 and not re.findall(r"^    (pub )?\w+:", s, re.M)
    s = enc[enc.index("pub struct ProofOfEncryption {"):]
    s = s[:s.index("}")]
    assert len(re.findall(r"^    pub\(crate\) \w+:", s, re.M)) == 8 and not re.findall(r"^    (pub )?\w+:", s, re.M)
    assert "pub struct ProofOfIssuance(pub(crate) CompactProof);" in iss
    # the patch reverses cleanly (a maintainer can back it out)
    back = subprocess.run(["patch", "-p1", "-R", "-i", PATCH], cwd=crate, capture_output=True, text=True)
    assert back.returncode == 0 and (crate / "src/issuer.rs").read_text() == before["src/issuer.rs"] and not (crate / "build.rs").exists()


def test_the_patch_is_what_the_generator_writes(crate, tmp_path):
    """integration/make_patch.py reproduces the committed patch from the crate's sources (the patch was not edited by hand)"""
    out = tmp_path / "regen"
    out.mkdir()
    shutil.copy(os.path.join(ROOT, "integration", "make_patch.py"), out / "make_patch.py")
    r = subprocess.run(["python3", str(out / "make_patch.py"), str(crate)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (out / "aeonflux_gpu.patch").read_text() == open(PATCH).read()


def test_the_patch_calls_what_the_shim_offers():
    """every crate::gpu:: path and engine method the patch uses exists in integration/aeonflux_gpu.rs with the shape the patch assumes"""
    patch = open(PATCH).read()
    shim = open(os.path.join(ROOT, "integration", "aeonflux_gpu.rs")).read()
    for fn in set(re.findall(r"crate::gpu::(\w+)\(", patch)):
        assert re.search(r"pub fn %s\(" % fn, shim), fn
    for m in ("try_verify", "try_issue", "try_show", "issuance", "credential"):
        assert ("." + m + "(") in patch and re.search(r"pub fn %s[<(]" % m, shim), m
    # the fault arms take back what the call consumed
    assert "Err((_fault, request)) => request" in patch and "Result<Result<CredentialIssuance, CredentialError>, (EngineFault, CredentialRequest)>" in re.sub(r"\s+", " ", shim)
    assert "Err((_fault, back))" in patch and "(EngineFault, CredentialIssuance)>" in shim
