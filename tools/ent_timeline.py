"""Per-wave timeline of one headline launch of the entropy kernel (library built with -DVBMC_INSTRUMENT: ENT_AB="inst:-DVBMC_INSTRUMENT" python tools/ent_ab.py build: every wave records entry / loop start /
loop end / exit on the 100 MHz counter and its HW_ID / XCC_ID).  Run on the GPU box:
    VBMC_HIP_LIB=vbmc_amd/lib/exp/libvbmc_hip_inst.so python tools/ent_timeline.py [R] [Ns]
Prints: kernel span, per-phase totals in slot-time, the gaps between consecutive waves on the same (XCC, SE, CU, SIMD, wave slot)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    import vbmc_amd
    from bench import synth_inputs
    from vbmc_amd import _lib

    D, N, K, S = int(os.environ.get("EXP_D", "10")), int(os.environ.get("EXP_N", "400")), int(os.environ.get("EXP_K", "50")), int(os.environ.get("EXP_S", "20"))
    inp = synth_inputs(0, D, N, K, S)
    eng = vbmc_amd.Engine(0)
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    th = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((theta0.size, R)))
    eng.ctx.set_profiling(True)
    for i in range(4):
        vbmc_amd.negelcbo_batch(th, 0, vp, gp, Ns, True, 0, seed=7 + i, engine=eng, outputs=("F",))
    ms = eng.ctx.last_kernel_ms()[0]
    lib = _lib.load()
    n = 6 * 32768
    buf = (ctypes.c_ulonglong * n)()
    rc = getattr(lib, "vbmc_dbg_ent_read_qs%d" % ((D + 2 + 3) // 4))(buf, ctypes.c_size_t(n))
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    ent, ls, le, ex = (a[:, i] - t0 for i in range(4))
    hw, xcc = a[:, 4], a[:, 5] & 15
    wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    print("waves recorded %d; kernel (HIP events) %.3f ms; first entry -> last exit %.3f ms" % (len(a), ms, ex.max() / 1e5))
    print("mean per wave [us]: set-up %.2f  loop %.2f  epilogue %.2f  (total %.2f)" % (np.mean(ls - ent) / 100, np.mean(le - ls) / 100, np.mean(ex - le) / 100, np.mean(ex - ent) / 100))
    q = np.percentile(le - ls, [1, 25, 50, 75, 99]) / 100
    print("loop duration percentiles 1/25/50/75/99 [us]: " + " ".join("%.1f" % v for v in q))
    span = ex.max()
    slot = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd) * 16 + wave_id
    simdk = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print("distinct (xcc,se,sh,cu): %d; distinct SIMDs %d; distinct wave slots %d" % (len(np.unique(simdk >> 2)), len(np.unique(simdk)), len(np.unique(slot))))
    # per SIMD: busy time with >= 1, >= 2 waves resident, and idle
    tot = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    loop_conc = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    for k in np.unique(simdk):
        m = simdk == k
        ev = sorted([(t, +1) for t in ent[m]] + [(t, -1) for t in ex[m]])
        cur, last = 0, 0
        for t, d in ev:
            tot[min(cur, 3)] += t - last
            cur += d; last = t
        tot[0] += span - last
        ev = sorted([(t, +1) for t in ls[m]] + [(t, -1) for t in le[m]])
        cur, last = 0, 0
        for t, d in ev:
            loop_conc[min(cur, 3)] += t - last
            cur += d; last = t
        loop_conc[0] += span - last
    ns = len(np.unique(simdk))
    print("SIMD time by resident waves (fraction of span): " + "  ".join("%d: %.3f" % (k, v / (ns * span)) for k, v in tot.items()))
    print("SIMD time by waves IN THE TILE LOOP (fraction):  " + "  ".join("%d: %.3f" % (k, v / (ns * span)) for k, v in loop_conc.items()))
    waves_per_simd = np.bincount(np.unique(simdk, return_inverse=True)[1])
    print("waves per SIMD: min %d mean %.2f max %d" % (waves_per_simd.min(), waves_per_simd.mean(), waves_per_simd.max()))
    # finishing time per SIMD
    fin = np.array([ex[simdk == k].max() for k in np.unique(simdk)])
    print("last exit per SIMD [ms]: min %.3f median %.3f max %.3f" % (fin.min() / 1e5, np.median(fin) / 1e5, fin.max() / 1e5))
    # where do the slow waves sit?  loop duration by XCC, by the number of waves of this launch on the wave's SIMD, by chunk index
    dur = (le - ls) / 100.0
    print("loop us by XCC: " + "  ".join("%d: %.0f/%.0f" % (x, np.median(dur[xcc == x]), dur[xcc == x].max()) for x in np.unique(xcc)) + "   (median/max)")
    inv = np.unique(simdk, return_inverse=True)[1]
    wps = waves_per_simd[inv]
    print("loop us by waves on the SIMD: " + "  ".join("%d waves: n=%d median %.0f p90 %.0f" % (k, (wps == k).sum(), np.median(dur[wps == k]), np.percentile(dur[wps == k], 90)) for k in np.unique(wps)))
    cuk = simdk >> 2
    wpc = np.bincount(np.unique(cuk, return_inverse=True)[1])[np.unique(cuk, return_inverse=True)[1]]
    print("loop us by waves on the CU: " + "  ".join("%d: n=%d med %.0f" % (k, (wpc == k).sum(), np.median(dur[wpc == k])) for k in np.unique(wpc)))
    print("entry time us percentiles 1/50/99: " + " ".join("%.1f" % v for v in np.percentile(ent, [1, 50, 99]) / 100.0))
    print("entry time us percentiles 60/70/80/90/95/98: " + " ".join("%.1f" % v for v in np.percentile(ent, [60, 70, 80, 90, 95, 98]) / 100.0))
    late = ent > 200        # entered more than 2 us after the first wave
    if late.any():
        # is a late wave the successor of an earlier one in the same wave slot (it waited for registers / LDS), or did a free slot wait for it?
        first_exit = {}
        for k_, e_ in zip(slot, ex):
            first_exit[k_] = min(first_exit.get(k_, 1 << 60), e_)
        wait_for_slot = np.array([first_exit[k_] <= t_ for k_, t_ in zip(slot[late], ent[late])])
        print("late waves (> 2 us): %d; of them %d entered a wave slot an earlier wave of this launch had left" % (late.sum(), wait_for_slot.sum()))
        cuk_ = simdk >> 2
        nres = np.array([np.sum((cuk_ == c_) & (ent <= t_) & (ex > t_)) for c_, t_ in zip(cuk_[late], ent[late])])
        print("waves resident on the late wave's compute unit at its entry (incl. itself): min %d median %d max %d" % (nres.min(), np.median(nres), nres.max()))
    tiles = (Ns // 2 + 15) // 16
    print("tile-signs per wave-loop: ~%.1f; loop ticks(10ns)/tile-sign median %.2f" % (2.0 * tiles * K * R / len(a), np.median(le - ls) / (2.0 * tiles * K * R / len(a))))


if __name__ == "__main__":
    main()
