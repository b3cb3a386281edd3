"""LightGlue output comparison shared by the GPU tests.

matches0 must agree on >= 99 % of the rows.  mscores0 is exp(max_j S_ij) for MUTUAL rows and exactly 0 otherwise
(filter_matches): a near-tie whose mutual flag differs between the fp16 path and the oracle moves the score by its whole
value although nothing is wrong numerically.  Rows whose flag (`score > 0`) differs are therefore taken OUT of the mscores0
comparison and judged by their size: a flip that moves the score by more than the bar counts against the flip budget (<= 0.5 % of
the rows, at least 1 allowed); a smaller one is within tolerance by definition (its whole effect is below the bar) and is only
counted (`small_flips`).  `mscores_maxd` is the arithmetic difference on the rows whose flag agrees.  (Rounds 1-3 left the small
flips inside `mscores_maxd`: the 0.0199 / 0.0194 / 0.0204 "grazing" figures of round 3 were flipped rows whose score sat just under
the bar, not arithmetic noise.  Same pass / fail decisions as before; the reported number now means what its name says.)

Two bars, and why they differ (VERDICT r03 "do this" 4):
  * PATH_VS_ORACLE_BAR = 2e-2: an fp16 path against the fp64 oracle / the golden fixtures / the transformers port - SURVEY 8(c)'s
    tolerance.  Worst case measured over every oracle comparison of the suite: 0.0133 (profiles/parity_report.json); an fp16 residual
    stream alone costs 0.0138 on the 600 x 600 fixture (DESIGN item 25), so this bar has ~1.45x of margin and is not ours to move.
  * PATH_VS_PATH_BAR = 3e-2: two DIFFERENT fp16 evaluations of the same problem (batch kernels vs per-pair kernels, translated /
    permuted inputs, different batch slots).  Each is an independent fp16 perturbation of the oracle's answer; each may sit up to
    2e-2 from it, so the triangle inequality allows 4e-2 between them; with the measured path-vs-oracle worst case e = 0.0133-0.0170
    the realistic spread is 2e = 0.027-0.034.  Round 3 compared two paths against the ORACLE's 2e-2 and measured 0.01993 / 0.0194 /
    0.02038: the noise floor of the comparison, not a margin.  3e-2 is the sum rule tests/test_gpu_lightglue_layers.py has used
    since round 2; bench.py's self_check uses the same constant (tests/test_parity_margins.py pins the two together).
MIN_MARGIN: every committed measurement must sit at least 1.25x under its bar (bar / measured), checked on CPU against
profiles/parity_report.json - a bar that is being grazed fails the CPU suite before it can turn a GPU run red for a non-bug."""
import numpy as np

AGREEMENT_BAR = 0.99
PATH_VS_ORACLE_BAR = 2e-2
PATH_VS_PATH_BAR = 3e-2
MSCORE_BAR = PATH_VS_ORACLE_BAR          # historical name
FLIP_FRACTION = 0.005
MIN_MARGIN = 1.25

# entries of the parity report that compare two fp16 paths with each other (everything else with an mscores figure is path vs oracle)
PATH_VS_PATH_ENTRIES = {"batch128_vs_per_frame": "mscores_maxd", "lg_batch64_vs_single": "mscores_maxd", "bench_self_check": "mscores_maxd",
                        "lg_translation_invariance": "mscores_maxd", "lg_permutation_equivariance": "mscores_maxd",
                        "lg_ffn_kernel_variants": "mscores_maxd", "config4_multicam_at_size": "mscores_maxd_one_pair_vs_batch"}


def compare(m, s, m_ref, s_ref, bar=PATH_VS_ORACLE_BAR):
    m, s, m_ref, s_ref = (np.asarray(a) for a in (m, s, m_ref, s_ref))
    ds = np.abs(s - s_ref)
    flagdiff = (s > 0) != (s_ref > 0)
    same = ~flagdiff
    return {"agreement": float((m == m_ref).mean()), "mismatched_rows": int((m != m_ref).sum()),
            "mutual_flips": int((flagdiff & (ds > bar)).sum()), "small_flips": int((flagdiff & (ds <= bar)).sum()),
            "mscores_maxd": float(ds[same].max()) if same.any() else 0.0, "mscores_maxd_all": float(ds.max()) if len(ds) else 0.0,
            "rows": int(len(m)), "mscores_bar": bar}


def check(c, mscore_bar=None):
    bar = c.get("mscores_bar", PATH_VS_ORACLE_BAR) if mscore_bar is None else mscore_bar
    assert c["agreement"] >= AGREEMENT_BAR or c["mismatched_rows"] <= 1, c
    assert c["mutual_flips"] <= max(1, int(FLIP_FRACTION * c["rows"])), c
    assert c["mscores_maxd"] <= bar, c


def margins(report):
    """{entry.key: (measured, bar, bar / measured)} for every mscores0 figure of a parity report."""
    out = {}
    for name, e in report.items():
        if not isinstance(e, dict):
            continue
        for key, v in e.items():
            if not key.startswith("mscores_maxd") or key == "mscores_maxd_all" or key.endswith(("_bar", "_margin")) or not isinstance(v, (int, float)):
                continue
            paths = PATH_VS_PATH_ENTRIES.get(name) == key
            bar = PATH_VS_PATH_BAR if paths else PATH_VS_ORACLE_BAR
            out[f"{name}.{key}"] = (float(v), bar, float("inf") if v == 0 else bar / float(v))
    return out


def annotate(report):
    """Write bar and margin next to every mscores0 figure (called when the GPU session's report is saved)."""
    for k, (v, bar, m) in margins(report).items():
        name, key = k.rsplit(".", 1)
        report[name][key + "_bar"] = bar
        report[name][key + "_margin"] = round(m, 3) if m != float("inf") else None
    return report
