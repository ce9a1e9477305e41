"""Generates tests/golden/l2_digests.json from the CPU oracle (run in the authoring container).

The reference holds no fixture for the transform level (its twiddle tables are placeholders), so these
are digests of the oracle's outputs on seeded inputs; tests/test_oracle.py ties the oracle itself to
the direct O(n^2) mathematics and to the reference's primitive-level known answers."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ntt_math as nm                      # noqa: E402
from workload import rns_poly              # noqa: E402
from oracle import pyoracle as orc         # noqa: E402

CASES = [
    dict(n=2048, moduli=[40961], batch=2, seed_a=101, seed_b=102),
    dict(n=8192, moduli=nm.ntt_primes(30, 8192, 4), batch=2, seed_a=103, seed_b=104),
    dict(n=16384, moduli=nm.ntt_primes(40, 16384, 6), batch=1, seed_a=105, seed_b=106),
    dict(n=4096, moduli=nm.ntt_primes(60, 4096, 2), batch=1, seed_a=107, seed_b=108),
    dict(n=1024, moduli=[12289], batch=1, seed_a=109, seed_b=110),
    dict(n=256, moduli=nm.ntt_primes(250, 256, 2), batch=1, seed_a=111, seed_b=112),
]

out = {"_provenance": "oracle/fhe_oracle.c outputs on tests/workload.py seeded inputs; see make_golden.py", "cases": []}
for c in CASES:
    rp = orc.RnsPlan(c["n"], c["moduli"])
    a = rns_poly(c["seed_a"], c["moduli"], c["n"], c["batch"]); b = rns_poly(c["seed_b"], c["moduli"], c["n"], c["batch"])
    fa = rp.forward(a, threads=8); pm = rp.polymul(a, b, threads=8)
    out["cases"].append(dict(n=c["n"], moduli=[str(q) for q in c["moduli"]], batch=c["batch"], seed_a=c["seed_a"],
                             seed_b=c["seed_b"], forward_sha256=hashlib.sha256(fa.tobytes()).hexdigest(),
                             forward_first8=[int(v) for v in fa[0, 0, :8, 0]],
                             polymul_sha256=hashlib.sha256(pm.tobytes()).hexdigest(),
                             polymul_first8=[int(v) for v in pm[0, 0, :8, 0]]))
with open(os.path.join(HERE, "l2_digests.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out["cases"]), "cases")
