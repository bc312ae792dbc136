"""GPU: random shapes through the launch forms the library chooses by itself (walking entropy launch, two chunk classes, value-only and
gradient matrix-core log joint, lane kernel) against the uniform chunk grid + VALU log joint of the same library on the same draws
(tools/fuzz_launch_forms.py: the entropy pieces to 1e-12, the log-joint pieces to 1e-9 -- its worst cases are the surrogate's conditioning)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", [3, 11])
def test_random_shapes_agree_across_launch_forms(seed):
    env = {k: v for k, v in os.environ.items() if k not in ("VBMC_ENT_CHUNKS", "VBMC_LJ_KERNEL", "VBMC_ENT_WALK", "VBMC_ENT_KERNEL")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_launch_forms.py"), "60", str(seed)], env=env, cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "cases 60" in r.stdout
