"""zkp 0.7's toolbox (Prover / Verifier over a merlin transcript, compact proofs) [3P], restated from the crate's published
behaviour.  Scalars are ints mod l, points are ristretto.py tuples; a CompactProof is (challenge, [responses])."""
from . import ristretto as R
from .keccak import Transcript


class ProofError(Exception):
    pass


def _domain_sep(t, label):
    t.append_message(b"dom-sep", b"schnorrzkp/1.0/ristretto255")
    t.append_message(b"dom-sep", label)


def _append_point(t, kind, label, enc):
    t.append_message(kind, label)
    t.append_message(b"val", enc)


def _challenge(t):
    return R.sc_from_wide(t.challenge_bytes(b"chal", 64))


class Prover:
    def __init__(self, proof_label, transcript):
        self.t = transcript
        _domain_sep(self.t, proof_label)
        self.scalars, self.points, self.labels, self.constraints = [], [], [], []

    def allocate_scalar(self, label, value):
        self.t.append_message(b"scvar", label)
        self.scalars.append(value % R.L)
        return len(self.scalars) - 1

    def allocate_point(self, label, point):
        enc = R.encode(point)
        _append_point(self.t, b"ptvar", label, enc)
        self.points.append(point)
        self.labels.append(label)
        return len(self.points) - 1

    def constrain(self, lhs, terms):
        self.constraints.append((lhs, list(terms)))

    def prove_compact(self, external_random32):
        """returns (challenge, responses, commitment encodings).  external_random32 = the 32 bytes merlin draws from
        thread_rng() in TranscriptRngBuilder::finalize."""
        rb = self.t.build_rng()
        for s in self.scalars:
            rb.rekey_with_witness_bytes(b"", R.sc_bytes(s))
        rng = rb.finalize(external_random32)
        blindings = [R.sc_from_wide(rng.fill_bytes(64)) for _ in self.scalars]
        coms = []
        for lhs, terms in self.constraints:
            c = R.msm([blindings[s] for s, _ in terms], [self.points[p] for _, p in terms])
            enc = R.encode(c)
            _append_point(self.t, b"blindcom", self.labels[lhs], enc)
            coms.append(enc)
        ch = _challenge(self.t)
        responses = [(s * ch + b) % R.L for s, b in zip(self.scalars, blindings)]
        return ch, responses, coms


class Verifier:
    def __init__(self, proof_label, transcript):
        self.t = transcript
        _domain_sep(self.t, proof_label)
        self.n_scalars = 0
        self.points, self.labels, self.constraints = [], [], []

    def allocate_scalar(self, label):
        self.t.append_message(b"scvar", label)
        self.n_scalars += 1
        return self.n_scalars - 1

    def allocate_point(self, label, enc):
        if enc == bytes(32):
            raise ProofError("identity point in the statement")      # validate_and_append_point_var
        _append_point(self.t, b"ptvar", label, enc)
        self.points.append(enc)
        self.labels.append(label)
        return len(self.points) - 1

    def constrain(self, lhs, terms):
        self.constraints.append((lhs, list(terms)))

    def verify_compact(self, challenge, responses):
        """raises ProofError, or returns the recomputed commitment encodings"""
        if len(responses) != self.n_scalars:
            raise ProofError("wrong number of responses")
        pts = []
        for enc in self.points:
            p = R.decode(enc)
            if p is None:
                raise ProofError("point does not decompress")
            pts.append(p)
        coms = []
        for lhs, terms in self.constraints:
            c = R.msm([responses[s] for s, _ in terms] + [(-challenge) % R.L], [pts[p] for _, p in terms] + [pts[lhs]])
            enc = R.encode(c)
            if enc == bytes(32):
                raise ProofError("identity commitment")              # validate_and_append_blinding_commitment
            _append_point(self.t, b"blindcom", self.labels[lhs], enc)
            coms.append(enc)
        if _challenge(self.t) != challenge % R.L:
            raise ProofError("challenge mismatch")
        return coms
