"""GPU: the fused front-end over unusual image sizes, keypoint budgets (1 .. 4096) and odd pair counts -
scripts/stress_shapes.py checks index ranges, score ranges, sortedness and unit-norm descriptors on every output."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_front_end_invariants_over_shapes_and_budgets():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_shapes.py")], capture_output=True, text=True,
                         timeout=600)
    print(out.stdout[-3000:])
    assert out.returncode == 0 and "stress ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
