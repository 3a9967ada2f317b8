"""CPU: the exposure measurement (oracle/parity_exposure.py) runs and brackets the oracle's own rule: solving an
instance with the fewest / the most insertions every optimal path allows is a different computation only where the
range is not a point, and the figures the script reports are consistent with each other."""
from oracle import parity_exposure
from ultra_amd import synth


def test_exposure_figures_are_consistent():
    w = synth.make_world(seed=11, n_genes=60)
    insts = synth.make_instances(w, 80, seed=3, preset="ont_cdna")
    r = parity_exposure.measure(w.index(), insts, 0.6, workers=1)
    assert r["instances"] == 80
    g = r["stored"]
    assert g["hits"] > 300 and 0 < g["range"] < g["hits"]
    assert g["alpha_not_min"] <= g["range"] and g["alpha_not_max"] <= g["range"] and g["gate_flips"] <= g["range"]
    assert g["alpha_not_min"] + g["alpha_not_max"] >= g["range"]          # a range that is not a point excludes one end at least
    assert 0 < g["max_acc_spread"] < 0.2
    assert r["chain_differs_any"] <= r["chain_differs_min"] + r["chain_differs_max"]
    assert r["value_differs_min"] >= r["chain_differs_min"] and r["value_differs_max"] >= r["chain_differs_max"]
    assert r["hirschberg"]["hits"] == 0            # ~1 kb reads against exons: 20 * ceil(m / 64) * L stays far below 1 MiB
