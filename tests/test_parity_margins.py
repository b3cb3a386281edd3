"""CPU: the committed GPU parity measurements keep a margin to their bars (VERDICT r03 "do this" 4).

profiles/parity_report.json is the record of the last GPU run of the suite.  Every mscores0 figure in it must sit at least
MIN_MARGIN (1.25x) under the bar that applies to it - 2e-2 against the oracle, 3e-2 between two fp16 paths
(tests/_lgcmp.py says why) - so a bar that is being grazed fails HERE, on CPU, instead of turning the driver's GPU run or
bench.py's exit status red after the next harmless reordering of a kernel."""
import json
import os
import re

import _lgcmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_committed_mscores_measurement_has_margin():
    rep = json.load(open(os.path.join(ROOT, "profiles", "parity_report.json")))
    m = _lgcmp.margins(rep)
    assert len(m) >= 8, sorted(m)
    assert any(k.startswith("batch128_vs_per_frame.") for k in m) and any(k.startswith("lg_batch64_vs_single.") for k in m)
    thin = {k: v for k, v in m.items() if v[2] < _lgcmp.MIN_MARGIN}
    assert not thin, f"measurement within {_lgcmp.MIN_MARGIN}x of its bar (measured, bar, margin): {thin}"
    # the path-vs-oracle figures are judged against the SURVEY's 2e-2, never the wider bar
    assert all(v[1] == _lgcmp.PATH_VS_ORACLE_BAR for k, v in m.items() if "vs_oracle" in k or k.startswith("lg_layers_"))


def test_bars_are_what_the_docstring_derives():
    assert _lgcmp.PATH_VS_ORACLE_BAR == 2e-2 and _lgcmp.MSCORE_BAR == 2e-2       # not ours to move
    assert _lgcmp.PATH_VS_ORACLE_BAR < _lgcmp.PATH_VS_PATH_BAR <= 2 * _lgcmp.PATH_VS_ORACLE_BAR   # at most the triangle inequality
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert float(re.search(r"^PATH_VS_PATH_BAR = ([0-9.e-]+)", src, re.M).group(1)) == _lgcmp.PATH_VS_PATH_BAR
    assert "<= 2e-2)" not in src            # self_check no longer compares two fp16 paths against the oracle's bar


def test_margins_and_annotation_on_a_synthetic_report():
    rep = {"batch128_vs_per_frame": {"mscores_maxd": 0.02}, "x_vs_oracle": {"mscores_maxd": 0.01, "mscores_maxd_all": 0.5},
           "lg_batch64_vs_single": {"mscores_maxd": 0.02, "mscores_maxd_vs_oracle": 0.0125}, "scalar": 0.99}
    m = _lgcmp.margins(rep)
    assert m["batch128_vs_per_frame.mscores_maxd"] == (0.02, 3e-2, 1.5)
    assert m["x_vs_oracle.mscores_maxd"] == (0.01, 2e-2, 2.0) and "x_vs_oracle.mscores_maxd_all" not in m
    assert m["lg_batch64_vs_single.mscores_maxd_vs_oracle"][1] == 2e-2 and m["lg_batch64_vs_single.mscores_maxd"][1] == 3e-2
    _lgcmp.annotate(rep)
    assert rep["batch128_vs_per_frame"]["mscores_maxd_margin"] == 1.5 and rep["x_vs_oracle"]["mscores_maxd_bar"] == 2e-2
