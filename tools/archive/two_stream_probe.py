"""Would the pipelined form gain from one STREAM per slot?  Headline shape, 64 restarts per batch, 40 batches:
(a) vbmc_elbo_submit / collect as shipped (two slots, one stream); (b) two contexts (a stream and a scratch set each), one slot
each, batches alternating between them.   python tools/two_stream_probe.py
Measured (round 3): 64 restarts per batch 25 470 -> 25 810 evals/s (+1.4 %: the step is the sum of the work on the chip, not a chain of
latencies), 8 restarts 19 600 -> 22 400 (+14 %).  Built INSIDE the library as well (slot 1 on a twin context with its own stream and
scratch): 25 470 -> 25 200 and 18 800 -> 19 100 -- the kernel trace shows why: beside the other batch's entropy kernel (16 000
workgroups queued) a batch's small kernels wait for wave slots (k_finalize_ws 11 -> 78 us, the 5 us read-back copy kernel 232 us), so
each pass gets longer by what the overlap was meant to save; it would take the entropy kernel on a LOWER-priority stream than the
rest of its own pass (a fork / join per pass, as the log joint has).  Not kept."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbmc_amd  # noqa: E402
from bench import synth_inputs  # noqa: E402

D, N, K, Ns, S, R = 10, 400, 50, 10000, 20, int(os.environ.get("PROBE_R", "64"))
inp = synth_inputs(0, D, N, K, S)
vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
T = theta0.size
thetas = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((T, R)))
engs = [vbmc_amd.Engine(0), vbmc_amd.Engine(0)]
objs = []
for e in engs:
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=e)
    objs.append(vbmc_amd.PreparedObjective(T, R, 0, vp, gp, Ns, 0, None, engine=e))


def run_one(n):
    o, pend = objs[0], []
    for i in range(n):
        o.submit(thetas, seed=i, slot=i & 1)
        pend.append(i & 1)
        if len(pend) == 2:
            o.collect(pend.pop(0))
    while pend:
        o.collect(pend.pop(0))


def run_two(n):
    pend = []
    for i in range(n):
        objs[i & 1].submit(thetas, seed=i, slot=0)
        pend.append(i & 1)
        if len(pend) == 2:
            objs[pend.pop(0)].collect(0)
    while pend:
        objs[pend.pop(0)].collect(0)


if os.environ.get("PROBE_SHARE_GP"):
    objs[1].dgp = objs[0].dgp        # both contexts read ONE uploaded surrogate
for name, fn in (("one stream ", run_one), ("two streams", run_two), ("one stream ", run_one), ("two streams", run_two)):
    fn(6)
    t = time.perf_counter()
    fn(40)
    dt = time.perf_counter() - t
    print("%s: %.3f ms per batch, %.0f evals/s" % (name, 1e3 * dt / 40, 40 * R / dt), flush=True)
