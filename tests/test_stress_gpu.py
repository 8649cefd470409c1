"""The randomised differentials of tests/stress_gpu.py (marching cubes vs the C oracle, decode / wide decode / voxeliser / fusion /
UNet3D / first-layer skip / hand branch / winding number vs the oracle, on random shapes) as a collected test: a fixed seed and a
60-second budget, in a process of its own (the script walks module-level state and exits with its verdict)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_differentials_fixed_seed_60s():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_gpu.py"), "60"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, STRESS_SEED="20261003", PYTHONPATH=root))
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"cases: (\{.*\}) failures: (\d+)", r.stdout)
    assert m and int(m.group(2)) == 0, tail
    counts = eval(m.group(1), {"__builtins__": {}})
    # every family ran (the heavier ones every 4th / 40th round)
    for name in ("mc", "decode", "voxel", "fusion", "hand", "winding", "skip", "wide"):
        assert counts.get(name, 0) >= 1, (name, counts)
