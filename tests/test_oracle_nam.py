"""CPU: pins the NAM-chaining oracle (oracle/chain_port.c: oracle_nam_quadratic / oracle_nam_nlogn,
SURVEY.md section 8 row f1) against golden vectors produced by the reference's own solvers
(tests/golden/make_nam_golden.py): optimal value and every co-optimal chain, in the reference's order."""
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nam_golden.json")


def test_nam_solvers_match_reference_vectors():
    from oracle import cport
    cases = json.load(open(GOLD))["cases"]
    nq = nt = multi = 0
    for rec in cases:
        for key, want in rec.items():
            if key.startswith("quad_"):
                C, tr = cport.nam_quadratic(rec["y"], rec["c"], rec["d"], int(key[5:]))
                nq += 1
            elif key == "nlogn":
                C, tr = cport.nam_nlogn(rec["c"], rec["d"])
                nt += 1
            else:
                continue
            chains, value = cport.all_optimal_chains(C, tr)
            assert value == want["value"], (rec["n"], rec["style"], key)
            assert chains == want["chains"], (rec["n"], rec["style"], key)
            multi += len(chains) > 1
    assert nq > 150 and nt > 100 and multi > 20
