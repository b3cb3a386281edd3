import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(1, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is a build product (git-ignored): compile it in-tree when a fresh checkout runs the tests
    before `__graft_entry__.build()` (hipcc cross-compiles gfx950 without a GPU; a few minutes, once)."""
    from superslam_amd import build as B

    if not os.path.exists(B.LIB):
        B.build()
    return B.LIB


@pytest.fixture(scope="session")
def weights_dir(tmp_path_factory):
    """Seeded synthetic weights written as safetensors (regenerated from the seed on every machine)."""
    from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors

    d = tmp_path_factory.mktemp("weights")
    sp = make_superpoint_weights(0)
    lg = make_lightglue_weights(1)
    save_safetensors(sp, str(d / "superpoint.safetensors"))
    save_safetensors(lg, str(d / "lightglue.safetensors"))
    return {"dir": str(d), "sp_path": str(d / "superpoint.safetensors"), "lg_path": str(d / "lightglue.safetensors"),
            "sp": sp, "lg": lg}


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def parity_report():
    """Measured parity numbers of the GPU suite (IoU, flips, ulps, agreement, |d|), written at session end to
    gpurun_out/parity_report.json (merged back from the GPU box; the copy kept for the record is profiles/parity_report.json)."""
    import json

    rep = {}
    yield rep
    if not rep:
        return
    import _lgcmp

    _lgcmp.annotate(rep)     # bar and margin (bar / measured) next to every mscores0 figure
    for d in (os.path.join(ROOT, "gpurun_out"),):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "parity_report.json"), "w") as f:
                json.dump(rep, f, indent=1, sort_keys=True)
        except OSError:
            pass
