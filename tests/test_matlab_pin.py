"""Is the oracle pinned by the reference itself?

The reference is MATLAB and cannot run in the development image or on the GPU box, and it ships no function-level vectors, so the
NumPy / C oracles and the HIP path are pinned to 50-digit mpmath evaluations of the reference's formulas (tests/golden/mp_*.json).
What closes the loop is ONE command for someone with MATLAB (tools/dump_golden.m): it evaluates the same inputs with the real
reference and writes tests/golden/matlab/.  Once that folder is committed this test compares every entry with the mpmath vectors
on every run ("pinned-by-MATLAB: present"); until then it reports "absent" -- explicitly, as a skip, and in the session header
(tests/conftest.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_matlab_golden as cmg  # noqa: E402

RTOL = 1e-9     # fp64 round-off of two different evaluation orders; north_star asks 1e-6


def test_reference_dump_agrees_with_the_mpmath_vectors():
    files = cmg.dump_files()
    if not files:
        pytest.skip("oracle pinned-by-MATLAB: absent -- run tools/dump_golden.m with MATLAB + VBMC and commit tests/golden/matlab/")
    families = {os.path.basename(f).split("case")[0] for f in files}
    assert {"matlab_", "matlab_nlz_", "matlab_pred_", "matlab_pen_", "matlab_acq_"} <= families, families
    bad = []
    for f in files:
        for key, err in cmg.compare(f):
            if err is None or err > RTOL:
                bad.append((os.path.basename(f), key, err))
    assert not bad, bad


def test_every_fixture_family_is_dumped_by_the_matlab_script():
    """tools/dump_golden.m covers all five families (and with mp_case4.json the Rosenbrock target of BASELINE configs[0]), and
    writes where this test looks."""
    src = open(os.path.join(ROOT, "tools", "dump_golden.m")).read()
    for pat in ("mp_case*.json", "mp_nlz_case*.json", "mp_pred_case*.json", "mp_pen_case*.json", "mp_acq_case*.json"):
        assert pat in src, pat
    assert "fullfile(gold,'matlab')" in src and src.count("write_json(fullfile(outdir,") == 5
    import json

    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "mp_case4.json")))
    assert rec["inputs"].get("target") == "rosenbrock" and rec["inputs"]["D"] == 2 and rec["inputs"]["K"] == 2 and rec["inputs"]["Mh"] == 50
