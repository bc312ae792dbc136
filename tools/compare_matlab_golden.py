#!/usr/bin/env python
"""Compare tests/golden/matlab_*.json (written by tools/dump_golden.m with the real MATLAB reference) against the
committed mpmath vectors tests/golden/mp_*.json.  The oracle and the HIP path are pinned to the mpmath vectors by the
test-suite, so agreement here pins all of them to the reference itself.

usage: python tools/compare_matlab_golden.py [rtol]      (default 1e-9; north_star asks 1e-6)
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def flat(x):
    return np.asarray(x, dtype=np.float64).reshape(-1)


def main(rtol):
    gold = os.path.join(ROOT, "tests", "golden")
    files = sorted(glob.glob(os.path.join(gold, "matlab_*.json")))
    if not files:
        print("no tests/golden/matlab_*.json found: run tools/dump_golden.m in MATLAB first")
        return 2
    worst = 0.0
    bad = 0
    for f in files:
        ref = json.load(open(f.replace("matlab_", "mp_")))["expected"]
        got = json.load(open(f))
        for key, val in ref.items():
            if key == "pos":      # nearest-neighbour indices of the acquisition fixtures: bookkeeping, 0-based here
                continue
            if key not in got:
                print("%-28s %-14s missing in the MATLAB dump" % (os.path.basename(f), key))
                bad += 1
                continue
            a, b = flat(got[key]), flat(val)
            if a.size != b.size:
                # J_sjk / L come back in MATLAB's column-major nesting: compare as sorted multisets only as a last resort
                print("%-28s %-14s size %d vs %d" % (os.path.basename(f), key, a.size, b.size))
                bad += 1
                continue
            # the dump keeps MATLAB's array layout; the fixtures are row-major lists of the same mathematical object
            err = min(np.max(np.abs(a - b)), np.max(np.abs(np.sort(a) - np.sort(b)))) / max(1.0, np.max(np.abs(b)))
            worst = max(worst, err)
            flag = "" if err <= rtol else "   <-- exceeds rtol"
            bad += err > rtol
            print("%-28s %-14s rel err %.3e%s" % (os.path.basename(f), key, err, flag))
    print("worst relative error %.3e over %d files; %d entries outside rtol = %g" % (worst, len(files), bad, rtol))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 1e-9))
