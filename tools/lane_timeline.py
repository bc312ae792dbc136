"""Per-wave timeline of one k_entropy_lane launch (library built with -DVBMC_INSTRUMENT: tools/lane_build.py inst:-DVBMC_INSTRUMENT,
VBMC_HIP_LIB=vbmc_amd/lib/tune/lib_inst.so).  Usage: python tools/lane_timeline.py [D N K Ns S R]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402
from vbmc_amd import _lib  # noqa: E402

D, N, K, Ns, S, R = (int(x) for x in (sys.argv[1:7] if len(sys.argv) >= 7 else (6, 200, 10, 1000, 8, 64)))
eng = vbmc_amd.default_engine()
inp = synth_inputs(0, D, N, K, S, "student" if D == 6 else "lumpy", False)
gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
th0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
th = np.asfortranarray(th0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((th0.size, R)))
for i in range(5):
    vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=50 + i, engine=eng, outputs=("F", "dF"))
lib = _lib.load()
dt = 2 * ((D + 1) // 2)
fn = getattr(lib, "vbmc_dbg_lane_read_dt%d" % dt)
fn.argtypes = [C.c_void_p, C.c_size_t]
NWV = 16384
buf = np.zeros(8 * NWV, dtype=np.uint64)
assert fn(buf.ctypes.data, buf.size) == 0
g = buf.reshape(NWV, 8)
g = g[g[:, 0] > 0]
t = g[:, :7].astype(np.int64)
t0 = t[:, 0].min()
us = np.where(t > 0, (t - t0) / 100.0, np.nan)
print("waves with a stamp: %d (%d with sample tiles, %d with a role)" % (len(g), np.sum(t[:, 4] > 0), np.sum(t[:, 3] > t[:, 2])))
names = ["entry", "staged", "role begin", "role end", "tiles begin", "tiles end", "exit"]
for i, n in enumerate(names):
    col = us[:, i][~np.isnan(us[:, i])]
    if len(col):
        print("%-11s min %6.2f  median %6.2f  max %6.2f us" % (n, col.min(), np.median(col), col.max()))
def ph(name, a, b, sel=None):
    d = us[:, b] - us[:, a]
    d = d[~np.isnan(d)]
    if sel is not None:
        d = (us[:, b] - us[:, a])[sel]
        d = d[~np.isnan(d)]
    if len(d):
        print("phase %-14s median %6.2f  mean %6.2f  max %6.2f us  (n = %d)" % (name, np.median(d), d.mean(), d.max(), len(d)))
ph("stage", 0, 1)
hasrole = t[:, 3] > t[:, 2]
ph("role", 2, 3, hasrole)
first = t[:, 4] < t[:, 2]
ph("role (2nd)", 2, 3, hasrole & first)
ph("role (1st)", 2, 3, hasrole & ~first)
ph("tiles+table", 4, 5)
ph("tiles (1st)", 4, 5, first)
ph("tiles (2nd)", 4, 5, ~first)
ph("whole wave", 0, 6)
hw = g[:, 7] & 0xffffffff
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
xcc = (g[:, 7] >> 32) & 15
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs used: %d; waves per SIMD min %d median %d max %d; tiles-first share %.2f" % (len(u), cnt.min(), int(np.median(cnt)), cnt.max(), first.mean()))
both = 0
for k in u[cnt == 2]:
    f = first[key == k]
    both += int(f[0] != f[1])
print("SIMDs with two waves: %d, of which in opposite order: %d" % (np.sum(cnt == 2), both))
