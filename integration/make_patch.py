#!/usr/bin/env python3
"""Writes integration/aeonflux_gpu.patch: the reference-side patch of INTEGRATION.md section 1 as a unified diff against the crate
(/root/reference, or the checkout given as argv[1]) - the `gpu` feature and its build script, `mod gpu`, the four delegating method
prologues (the crate's own bodies stay, as the fall-through) and the pub(crate) widening the shim needs.  The shim itself is not in
the diff: integration/aeonflux_gpu.rs is copied to src/gpu.rs.

    python integration/make_patch.py [/path/to/aeonflux]        # rewrites integration/aeonflux_gpu.patch

The edits are insertions in front of the existing bodies wherever that is possible, so that the diff carries as little of the crate's
text as a patch can (context lines and the eleven field declarations whose visibility changes)."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def edit(path, old, new):
    s = open(path).read()
    if s.count(old) != 1:
        raise SystemExit("%s: expected exactly one occurrence of %r" % (path, old[:60]))
    open(path, "w").write(s.replace(old, new))


def apply_edits(root):
    j = lambda *p: os.path.join(root, *p)
    # ---- Cargo.toml: the feature, the link name, the build script
    edit(j("Cargo.toml"), 'autobenches = false\n', 'autobenches = false\nbuild = "build.rs"\nlinks = "aeonflux_gpu"\n')
    edit(j("Cargo.toml"), 'simd_backend = [ "curve25519-dalek/simd_backend", "zkp/simd_backend" ]\n',
         'simd_backend = [ "curve25519-dalek/simd_backend", "zkp/simd_backend" ]\n'
         '# the MI355X batch engine behind Issuer::issue / Issuer::verify / AnonymousCredential::show (src/gpu.rs; libaeonflux_gpu.so)\n'
         'gpu = [ "alloc" ]\n')
    open(j("build.rs"), "w").write(
        '// Links libaeonflux_gpu.so when the `gpu` feature is on (AEONFLUX_GPU_LIB_DIR: where the library lies, if not on the linker\'s path).\n'
        'fn main() {\n'
        '    if std::env::var_os("CARGO_FEATURE_GPU").is_some() {\n'
        '        if let Some(dir) = std::env::var_os("AEONFLUX_GPU_LIB_DIR") {\n'
        '            println!("cargo:rustc-link-search=native={}", dir.to_string_lossy());\n'
        '        }\n'
        '        println!("cargo:rustc-link-lib=dylib=aeonflux_gpu");\n'
        '    }\n'
        '    println!("cargo:rerun-if-env-changed=AEONFLUX_GPU_LIB_DIR");\n'
        '}\n')
    # ---- src/lib.rs
    edit(j("src", "lib.rs"), 'pub mod errors;\n', 'pub mod errors;\n#[cfg(feature = "gpu")]\npub mod gpu;\n')
    # ---- src/issuer.rs: three prologues; the existing bodies follow them unchanged
    edit(j("src", "issuer.rs"),
         '    ) -> Result<AnonymousCredential, CredentialError>\n    {\n',
         '    ) -> Result<AnonymousCredential, CredentialError>\n    {\n'
         '        #[cfg(feature = "gpu")]\n'
         '        {\n'
         '            if let Some(engine) = crate::gpu::user_engine(system_parameters, issuer_parameters) {\n'
         '                return match engine.issuance(self).try_verify() {\n'
         '                    Ok(result) => result,\n'
         '                    // (an accelerator fault hands the issuance back: the check of the body below, on it)\n'
         '                    Err((_fault, back)) => back.proof.verify(system_parameters, issuer_parameters, &back.credential).and(Ok(back.credential)),\n'
         '                };\n'
         '            }\n'
         '        }\n')
    edit(j("src", "issuer.rs"),
         '        C: CryptoRng + RngCore,\n    {\n        let amac = Amac::tag(',
         '        C: CryptoRng + RngCore,\n    {\n'
         '        #[cfg(feature = "gpu")]\n'
         '        let request = match crate::gpu::issuer_engine(self) {\n'
         '            Some(engine) => match engine.try_issue(request, csprng) { Ok(result) => return result, Err((_fault, request)) => request },\n'
         '            None => request,\n'
         '        };\n'
         '        let amac = Amac::tag(')
    edit(j("src", "issuer.rs"),
         '    ) -> Result<(), CredentialError>\n    {\n',
         '    ) -> Result<(), CredentialError>\n    {\n'
         '        #[cfg(feature = "gpu")]\n'
         '        {\n'
         '            if let Some(engine) = crate::gpu::issuer_engine(self) {\n'
         '                if let Ok(verdict) = engine.try_verify(presentation) { return verdict; }   // (an accelerator fault: the body below)\n'
         '            }\n'
         '        }\n')
    # ---- src/credential.rs
    edit(j("src", "credential.rs"),
         '    ) -> Result<ProofOfValidCredential, CredentialError>\n    {\n',
         '    ) -> Result<ProofOfValidCredential, CredentialError>\n    {\n'
         '        #[cfg(feature = "gpu")]\n'
         '        {\n'
         '            if let Some(engine) = crate::gpu::user_engine(system_parameters, issuer_parameters) {\n'
         '                if let Ok(result) = engine.credential(self).try_show(keypair, &mut csprng) { return result; }\n'
         '            }\n'
         '        }\n')
    # ---- visibility: the shim reads and rebuilds these (INTEGRATION.md section 1 "Visibility")
    for f in ("proof: CompactProof,\n    proofs_of_encryption:", "proofs_of_encryption: Vec<(u16, ProofOfEncryption)>,", "encrypted_attributes: Vec<EncryptedAttribute>,",
              "hidden_scalar_indices: Vec<u16>,", "C_x_0: RistrettoPoint,", "C_x_1: RistrettoPoint,", "C_V:   RistrettoPoint,", "C_y: Vec<RistrettoPoint>,"):
        edit(j("src", "nizk", "presentation.rs"), "    " + f, "    pub(crate) " + f)
    for f in ("proof: CompactProof,\n    public_key", "public_key: SymmetricPublicKey,", "index: u16,\n    C_y_1", "C_y_1: RistrettoPoint,", "C_y_2: RistrettoPoint,", "C_y_3: RistrettoPoint,",
              "C_y_2_prime: RistrettoPoint,"):
        edit(j("src", "nizk", "encryption.rs"), "    " + f, "    pub(crate) " + f)
    edit(j("src", "nizk", "issuance.rs"), "pub struct ProofOfIssuance(CompactProof);", "pub struct ProofOfIssuance(pub(crate) CompactProof);")


def main():
    ref = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "a"), os.path.join(d, "b")
        ignore = shutil.ignore_patterns(".git", "target")
        shutil.copytree(ref, a, ignore=ignore)
        shutil.copytree(ref, b, ignore=ignore)
        apply_edits(b)
        r = subprocess.run(["diff", "-ruN", "a", "b"], cwd=d, capture_output=True, text=True)
        if r.returncode not in (0, 1):
            raise SystemExit(r.stderr)
        lines = [l for l in r.stdout.split("\n") if not l.startswith("diff -ruN")]
        # (timestamps out of the file headers: the patch is the same whenever it is made)
        out = []
        for l in lines:
            if l.startswith("--- ") or l.startswith("+++ "):
                l = l.split("\t")[0]
                if l.startswith("--- a/") and not os.path.exists(os.path.join(d, l[4:])):
                    l = "--- /dev/null"          # a new file (build.rs): backing the patch out removes it
            out.append(l)
    header = ("The reference-side patch of INTEGRATION.md section 1: apply inside the aeonflux crate with `patch -p1 < aeonflux_gpu.patch`, copy\n"
              "integration/aeonflux_gpu.rs to src/gpu.rs, build with `--features gpu` (AEONFLUX_GPU_LIB_DIR = where libaeonflux_gpu.so lies).\n"
              "Made by integration/make_patch.py; tests/test_integration_patch.py applies it to a copy of the crate.\n\n")
    open(os.path.join(HERE, "aeonflux_gpu.patch"), "w").write(header + "\n".join(out))
    print("integration/aeonflux_gpu.patch: %d lines" % (len(out) + 4))


if __name__ == "__main__":
    main()
