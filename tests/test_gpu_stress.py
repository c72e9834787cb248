"""Short run of tools/stress_reduce.py: the cross-workgroup hand-off of the single-launch reduction (agent-scope
release -> ticket -> acquire) under concurrent contexts, random grid sizes and back-to-back async launches."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_reduction_handoff_under_stress():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "stress_reduce.py"), "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 errors" in r.stdout
