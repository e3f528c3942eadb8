"""TEST INFRASTRUCTURE.  A second restatement of the path's algorithm, written independently of oracle/ (pure Python integers,
own Keccak - no hashlib, no ctypes): ristretto255 (RFC 9496), merlin transcripts over STROBE-128, zkp 0.7's compact Schnorr
prover / verifier, and the aeonflux statements (/root/reference/src/nizk/{issuance,presentation,encryption}.rs, src/amacs.rs).
tests/gen_golden.py and tests/test_pyref_cross_check.py replay tests/golden/flows.json through it: every challenge, response,
commitment and accept / reject decision the oracle produced must come out of this code too.  Two independently written
restatements agreeing is weaker than a pin by the reference (which cannot be built here and holds no vectors), and stronger
than one restatement checked against itself."""
