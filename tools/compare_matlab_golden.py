#!/usr/bin/env python
"""Compare the MATLAB dump of the golden inputs (tests/golden/matlab/matlab_*.json, written by tools/dump_golden.m with the real
reference) against the committed mpmath vectors tests/golden/mp_*.json.  The oracle and the HIP path are pinned to the mpmath
vectors by the test-suite, so agreement here pins all of them to the reference itself.  tests/test_matlab_pin.py runs the same
comparison on every test run once the folder is committed.

usage: python tools/compare_matlab_golden.py [rtol]      (default 1e-9; north_star asks 1e-6)
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def flat(x):
    return np.asarray(x, dtype=np.float64).reshape(-1)


def dump_files():
    """The MATLAB owner's drop: tests/golden/matlab/ (older dumps sat next to the mp files)."""
    return sorted(glob.glob(os.path.join(GOLD, "matlab", "matlab_*.json")) + glob.glob(os.path.join(GOLD, "matlab_*.json")))


def compare(path):
    """-> list of (key, relative error or None if missing / mis-shaped) for one dump file against its mp_ twin."""
    ref = json.load(open(os.path.join(GOLD, os.path.basename(path).replace("matlab_", "mp_"))))["expected"]
    got = json.load(open(path))
    rows = []
    for key, val in ref.items():
        if key == "pos":      # nearest-neighbour indices of the acquisition fixtures: bookkeeping, 0-based here
            continue
        if key not in got:
            rows.append((key, None))
            continue
        a, b = flat(got[key]), flat(val)
        if a.size != b.size:
            rows.append((key, None))
            continue
        # the dump keeps MATLAB's array layout; the fixtures are row-major lists of the same mathematical object: compare in
        # order and, for transposed nestings, as sorted multisets
        err = min(np.max(np.abs(a - b)), np.max(np.abs(np.sort(a) - np.sort(b)))) / max(1.0, np.max(np.abs(b)))
        rows.append((key, float(err)))
    return rows


def main(rtol):
    files = dump_files()
    if not files:
        print("no tests/golden/matlab/matlab_*.json found: run tools/dump_golden.m in MATLAB first")
        return 2
    worst, bad = 0.0, 0
    for f in files:
        for key, err in compare(f):
            if err is None:
                print("%-28s %-14s missing / wrong size in the MATLAB dump" % (os.path.basename(f), key))
                bad += 1
                continue
            worst = max(worst, err)
            bad += err > rtol
            print("%-28s %-14s rel err %.3e%s" % (os.path.basename(f), key, err, "" if err <= rtol else "   <-- exceeds rtol"))
    print("worst relative error %.3e over %d files; %d entries outside rtol = %g" % (worst, len(files), bad, rtol))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 1e-9))
