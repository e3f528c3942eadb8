"""A REHEARSAL input for tests/test_pin_fixture.py: the oracle-made golden flows rewritten in the schema integration/pin_against_crate.rs
exports (no randomness fields), so that the acceptance tests for crate-made flows can be run end to end before anybody has run the
crate.  It pins nothing (the flows are the oracle's own).  python tools/simulate_crate_flows.py out.json; AFX_CRATE_FLOWS=out.json pytest tests/test_pin_fixture.py"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "tests", "golden", "flows.json")))
out = []
for r in d["flows"]:
    if r["show"]["status"] == 0 and r["issue"]["status"] == 0:
        f = {k: r[k] for k in ("name", "n", "params", "key", "issuer_params", "issuance_verify", "presentation", "verify")}
        f["issue"] = {k: r["issue"][k] for k in ("kinds", "values", "status", "t", "U", "V", "challenge", "responses")}
        out.append(f)
json.dump({"_source": "REHEARSAL: oracle-made flows in the crate-export schema (tools/simulate_crate_flows.py)", "flows": out}, open(sys.argv[1], "w"))
print(len(out), "flows")
