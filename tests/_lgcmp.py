"""LightGlue output comparison shared by the GPU tests.

matches0 must agree on >= 99 % of the rows.  mscores0 is exp(max_j S_ij) for MUTUAL rows and exactly 0 otherwise
(filter_matches): a near-tie whose mutual flag differs between the fp16 path and the oracle moves the score by its whole
value although nothing is wrong numerically - such rows are counted (`flips`, <= 0.5 % of the rows, at least 1 allowed) and
the 2e-2 bar applies to the rows whose flag agrees."""
import numpy as np

AGREEMENT_BAR = 0.99
MSCORE_BAR = 2e-2
FLIP_FRACTION = 0.005


def compare(m, s, m_ref, s_ref):
    m, s, m_ref, s_ref = (np.asarray(a) for a in (m, s, m_ref, s_ref))
    ds = np.abs(s - s_ref)
    # a flip = the mutual flag differs AND the score moved by more than the bar (exp(max) of a hopeless row underflows to 0 on
    # the GPU and to a denormal in the fp64 oracle: same flag for every practical purpose)
    same = ~(((s > 0) != (s_ref > 0)) & (ds > MSCORE_BAR))
    return {"agreement": float((m == m_ref).mean()), "mismatched_rows": int((m != m_ref).sum()), "mutual_flips": int((~same).sum()),
            "mscores_maxd": float(ds[same].max()) if same.any() else 0.0, "mscores_maxd_all": float(ds.max()) if len(ds) else 0.0,
            "rows": int(len(m))}


def check(c, mscore_bar=MSCORE_BAR):
    assert c["agreement"] >= AGREEMENT_BAR or c["mismatched_rows"] <= 1, c
    assert c["mutual_flips"] <= max(1, int(FLIP_FRACTION * c["rows"])), c
    assert c["mscores_maxd"] <= mscore_bar, c
