"""Writes vbmc_amd/csrc/exp2_tab1k.h: 2^(j/1024), j = 0..1023, correctly rounded to fp64 (50-digit mpmath), the table of the
entropy kernel's exp (device_math.h: vb_exp_tab1k)."""
import os

import mpmath as mp

mp.mp.dps = 50
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for j in range(0, 1024, 4):
    rows.append("    " + ", ".join(repr(float(mp.power(2, mp.mpf(j + t) / 1024))) for t in range(4)) + ",")
with open(os.path.join(ROOT, "vbmc_amd", "csrc", "exp2_tab1k.h"), "w") as f:
    f.write("// 2^(j/1024), j = 0..1023, correctly rounded (tools/gen_exp2_tab.py, mpmath 50 digits)\n#pragma once\n")
    f.write("static __constant__ double c_exp2_tab1k[1024] = {\n" + "\n".join(rows) + "\n};\n")
