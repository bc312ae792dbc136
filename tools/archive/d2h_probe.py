"""D2H of a gp.post(s).L-sized block (S N^2 doubles) into pageable host memory: the plain hipMemcpyAsync path of the ABI helper
against the double-buffered pinned bounce path gplite_post uses (common.h: d2h_bounced).  Usage: python tools/d2h_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

eng = vbmc_amd.Engine(0)
ctx = eng.ctx
n = 20 * 400 * 400
dst = np.empty(n)
p = C.c_void_p()
ctx.check(ctx.lib.vbmc_device_alloc(ctx.h, C.c_size_t(8 * n), C.byref(p)))
for _ in range(3):
    ctx.check(ctx.lib.vbmc_memcpy_d2h(ctx.h, dst.ctypes.data_as(C.c_void_p), p, C.c_size_t(8 * n)))
t = time.perf_counter()
for _ in range(10):
    ctx.check(ctx.lib.vbmc_memcpy_d2h(ctx.h, dst.ctypes.data_as(C.c_void_p), p, C.c_size_t(8 * n)))
dt = (time.perf_counter() - t) / 10
print("plain hipMemcpyAsync to pageable memory: %.2f ms for %.1f MB = %.1f GB/s" % (dt * 1e3, 8 * n / 1e6, 8 * n / dt / 1e9))
t = time.perf_counter()
for _ in range(10):
    fresh = np.empty(n)      # untouched pages: every copy pays the first-touch faults
    ctx.check(ctx.lib.vbmc_memcpy_d2h(ctx.h, fresh.ctypes.data_as(C.c_void_p), p, C.c_size_t(8 * n)))
dt = (time.perf_counter() - t) / 10
print("same copy into a FRESH numpy array each time: %.2f ms = %.1f GB/s" % (dt * 1e3, 8 * n / dt / 1e9))
inp = synth_inputs(0, 10, 400, 50, 20)
f = lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)  # noqa: E731
f()
t = time.perf_counter()
for _ in range(5):
    f()
print("gplite_post (returns alpha, L, sW for S = 20, N = 400): %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3))
