import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    import glob

    drop = glob.glob(os.path.join(ROOT, "tests", "golden", "matlab", "matlab_*.json"))
    return ["oracle pinned-by-MATLAB: %s" % ("present (%d reference dumps under tests/golden/matlab/, checked by tests/test_matlab_pin.py)" % len(drop)
                                             if drop else "absent (no MATLAB here; tools/dump_golden.m + tools/compare_matlab_golden.py close it)")]
