#!/usr/bin/env python
"""bench.py -- ELBO+grad evals/sec of the MI355X-native negelcbo path (BASELINE.json metric).

One "step" = one batched pass of the hot path: R = 64 independent negelcbo_vbmc(theta_r, 0, vp, gp,
Ns, 1, 0) evaluations (value + gradient; the Adam-loop call misc/vpoptimize_vbmc.m:71) on the
headline shape D=10, N=400, K=50, Ns=1e4 per component, S=20 GP hyper-samples (BASELINE.json
configs[2]), fresh Monte-Carlo draws every step (device Philox stream), theta H2D and (F, dF) D2H
inside the timed region, GP upload outside it.  N GPUs: one process per GPU, each rank owns its own
R restarts (weak scaling), and the per-step exchange is an all-gather of the R ELCBO values over RCCL
(misc/vpsieve_vbmc.m:81 sorts them; every rank ends with the full vector).

Launch forms (both give one process per GPU over torch.distributed, backend nccl == RCCL):
  * `python bench.py --gpus N ...`            -- this process re-executes itself N times (RANK / LOCAL_RANK / WORLD_SIZE /
                                                MASTER_ADDR=127.0.0.1 / MASTER_PORT set per child) and relays rank 0's JSON line;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` -- the ranks are already there (WORLD_SIZE in
                                                the environment) and are used as they are.
The JSON line carries what was actually observed: `n_gpus` = the torch.distributed world size, `ranks` = per-rank device,
evals/s and wall time, `backend`.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime initialises (see vbmc_amd/_lib.py)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # dense fp64 peak of MI355X, MFMA f64 = vector f64 (256 CU x 128 flop/clk x 2.4 GHz); see DESIGN.md


def synth_inputs(seed, D, N, K, S, target="lumpy", noisy=False):
    """Synthetic GP + VP of the headline shape: lumpy 12-component target (configs[2..4]) or the multivariate
    Student-t log-density of configs[1] (nu = 5, scale diag(1:D)/D); `noisy` adds N(0,1) observation noise (configs[4])."""
    rng = np.random.default_rng(seed)
    X = 1.5 * rng.standard_normal((N, D))
    if target == "student":
        y = -0.5 * (5.0 + D) * np.log1p(np.sum((X / (np.arange(1, D + 1) / D)) ** 2, axis=1) / 5.0)
    else:
        nc = 12
        mus = rng.uniform(-2, 2, size=(nc, D))
        sig = rng.uniform(0.3, 1.0, size=nc)
        wts = rng.dirichlet(np.ones(nc))
        lp = np.stack([np.log(wts[i]) - 0.5 * np.sum(((X - mus[i]) / sig[i]) ** 2, axis=1) - D * np.log(sig[i])
                       - 0.5 * D * np.log(2 * np.pi) for i in range(nc)])
        mx = lp.max(axis=0)
        y = mx + np.log(np.sum(np.exp(lp - mx), axis=0))
    if noisy:
        y = y + rng.standard_normal(N)
    hyp = np.zeros((D + 2 + 2 * D + 1, S))
    for s in range(S):
        hyp[:D, s] = np.log(0.8) + 0.2 * rng.standard_normal(D)
        hyp[D, s] = np.log(np.std(y)) + 0.1 * rng.standard_normal()
        hyp[D + 1, s] = np.log(1e-3)
        hyp[D + 2, s] = np.max(y)
        hyp[D + 3: D + 3 + D, s] = 0.2 * rng.standard_normal(D)
        hyp[D + 3 + D:, s] = np.log(2.0) + 0.1 * rng.standard_normal(D)
    order = np.argsort(-y, kind="stable")
    hpd = X[order[: int(round(0.8 * N))]]
    mu = hpd[rng.permutation(hpd.shape[0])[np.arange(K) % hpd.shape[0]]].T.copy()
    sigma = np.sqrt(np.mean(np.var(mu, axis=1, ddof=1)) / K) * np.exp(0.2 * rng.standard_normal(K))
    lam = np.std(hpd, axis=0, ddof=1)
    lam = lam * np.sqrt(D / np.sum(lam ** 2))
    eta = 0.3 * rng.standard_normal(K)
    return dict(X=X, y=y, hyp=hyp, mu=mu, sigma=sigma, lam=lam, eta=eta)


def algorithmic_flops(D, K, M, S, N):
    """entmc value+grad flops and gplogjoint flops per evaluation."""
    P = K * M * K
    ent = P * (5 * D + 7) + K * M * (5 * D + 4)
    lj = S * K * (16 * D * N + 5 * N)
    return ent, lj, P


def cpu_baseline(inp, gp, D, K, Ns_full, budget_s=20.0):
    """The compiled C port of the MATLAB path (oracle/vbmc_oracle.c, reference loop structure) timed on
    this host: one thread (the reported baseline) and all cores with OpenMP (extra).  Bounded sample:
    the log-joint part is timed in full; the entropy part at a reduced Ns and scaled linearly in Ns
    (its cost is exactly linear: K*Ns samples x K densities)."""
    from oracle import c_oracle  # checker / baseline leg only

    theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    alpha = np.stack([p["alpha"] for p in gp["post"]], axis=1)
    rng = np.random.default_rng(0)

    def run(Ns, omp):
        eps = rng.standard_normal((K, Ns // 2, D))
        t0 = time.perf_counter()
        c_oracle.negelcbo(theta, inp["X"], inp["hyp"], alpha, eps, openmp=omp)
        return time.perf_counter() - t0

    out = {}
    for omp in (False, True):
        run(2, omp)                          # untimed: the first call into the port loads (or rebuilds, -march=native) the library
        t_lj = run(2, omp)                   # entropy negligible: K*2 samples
        Ns1 = 400
        t1 = run(Ns1, omp) - t_lj
        Ns2 = int(min(Ns_full, max(Ns1, Ns1 * (budget_s / 2) / max(t1, 1e-6))))
        Ns2 -= Ns2 % 2
        reps = [run(Ns2, omp)]
        while len(reps) < 10 and sum(reps) + reps[0] < budget_s / 2:    # median of up to 10 repeats inside the time budget
            reps.append(run(Ns2, omp))
        t2 = float(np.median(reps))
        t_full = t_lj + (t2 - t_lj) * (Ns_full / Ns2)
        out[omp] = (1.0 / t_full, Ns2, t2, t_lj, len(reps))
    lib = c_oracle.load(True)
    v1, Ns2, t2, t_lj, nrep = out[False]
    vo = out[True]
    return {"value": v1, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "C port of the MATLAB loop nest (oracle/vbmc_oracle.c), 1 thread: median of %d evaluations at Ns=%d of %d per "
                      "component, %.2fs each; entropy part scaled linearly in Ns, log-joint part (%.3fs) timed in full" % (nrep, Ns2, Ns_full, t2, t_lj),
            "all_cores": {"value": vo[0], "cores": int(lib.oracle_num_threads()), "sample": "same port with OpenMP, median of %d evaluations at Ns=%d" % (vo[4], vo[1])}}


def interpreted_baseline(D, K, Ns_full, budget_s=20.0):
    """"if matlab or octave is discovered on the GPU box at bench time, additionally time the build's own .m restatement".
    tools/cpu_ref_entmc.m is that restatement (written from the formulas of ent/entmc_vbmc.m:28-128, vectorised over samples the way
    the reference is); it is run at a reduced sample count inside a time budget and scaled linearly in Ns.  Returns {"found": None}
    when neither interpreter is on the PATH -- the usual case: the image ships neither."""
    import shutil
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    script = os.path.join(here, "tools", "cpu_ref_entmc.m")
    for exe, argv in (("matlab", ["-batch"]), ("octave-cli", ["--no-gui", "--quiet", "--eval"]), ("octave", ["--no-gui", "--quiet", "--eval"])):
        path = shutil.which(exe)
        if not path:
            continue
        Ns1 = 400
        cmd = "addpath('%s'); cpu_ref_entmc(%d, %d, %d, %g);" % (os.path.join(here, "tools"), D, K, Ns1, budget_s / 2)
        try:
            r = subprocess.run([path] + argv + [cmd], capture_output=True, text=True, timeout=6 * budget_s)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("ENTMC_SECONDS_PER_EVAL")]
            if r.returncode != 0 or not line:
                return {"found": exe, "error": (r.stderr or r.stdout)[-300:]}
            sec = float(line[-1].split()[1]) * (Ns_full / Ns1)
            return {"found": exe, "value": 1.0 / sec, "unit": "evals/s (entropy + gradient only)", "kind": "port, interpreted",
                    "sample": "tools/cpu_ref_entmc.m (the build's own restatement of ent/entmc_vbmc.m, not the reference's file) at Ns=%d of %d per "
                              "component, scaled linearly in Ns" % (Ns1, Ns_full)}
        except Exception as e:  # noqa: BLE001
            return {"found": exe, "error": str(e)[:300]}
    return {"found": None, "note": "neither matlab nor octave on the PATH of this host: no interpreted baseline (tools/cpu_ref_entmc.m is what would run)"}


def newest_profile_with(marker):
    """the newest profiles/rNN_*_summary.md that holds a section for `marker` (e.g. "configs[4]"), or None"""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    best = None
    try:
        for f in sorted(os.listdir(here)):
            if f.endswith("_summary.md") and marker in open(os.path.join(here, f), errors="replace").read():
                best = "profiles/" + f
    except OSError:
        pass
    return best


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one rank per GPU, and wait for them.
    Rank 0 prints the JSON line on the inherited stdout.  Children are killed by PID if one of them fails."""
    import subprocess

    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VBMC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:   # one rank died: the others would wait in a collective forever
                        q.kill()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def kernel_source_hash():
    """Identifies the CODE of the entropy kernel a PMC figure (profiles/*_pmc.json) belongs to: sha256 over its sources with
    // comments and all white space removed, so that editing a comment does not make a measured figure look stale."""
    import hashlib
    import re

    here = os.path.dirname(os.path.abspath(__file__))
    hh = hashlib.sha256()
    for fn in ("entropy_mfma.h", "ent_mfma_inst.hip", "device_math.h", "elbo_types.h"):
        txt = open(os.path.join(here, "vbmc_amd", "csrc", fn)).read()
        txt = re.sub(r"//[^\n]*", "", txt)
        hh.update(re.sub(r"\s+", "", txt).encode())
    return hh.hexdigest()[:16]


def entropy_kernel_label(D, K):
    """the instantiation the library's own policy (mfma_entropy_fits, vbmc_amd/csrc/abi_elbo.hip) picks for this shape, asked
    through the reporting hook vbmc_entropy_plan: k-tiles per wave, component tail, waves per workgroup"""
    import ctypes

    from vbmc_amd import _lib

    qs, kt, hv, tl = (ctypes.c_int() for _ in range(4))
    kind = _lib.load().vbmc_entropy_plan(int(D), int(K), ctypes.byref(qs), ctypes.byref(kt), ctypes.byref(hv), ctypes.byref(tl))
    if not kind:
        return "k_entropy<D=%d,grad> (VALU)" % D
    if kind == 2:
        return "k_entropy_lane<DT=%d,KP=%d,grad> (+ log-joint role)" % (qs.value, kt.value)
    Kh = (K + hv.value - 1) // hv.value
    return "k_entropy_mfma<QS=%d,KT=%d%s,grad%s>" % (qs.value, kt.value, "+tail%d" % (Kh - 16 * kt.value) if tl.value else "",
                                                     "" if hv.value == 1 else ",HV=%d" % hv.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--restarts", type=int, default=64, help="R: ELBO+grad evaluations batched per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--D", type=int, default=10)
    ap.add_argument("--N", type=int, default=400)
    ap.add_argument("--K", type=int, default=50)
    ap.add_argument("--Ns", type=int, default=10000)
    ap.add_argument("--S", type=int, default=20)
    ap.add_argument("--eps-stream", action="store_true", help="(kept for compatibility: the parity-mode leg now runs by default, see --no-aux)")
    ap.add_argument("--extras", action="store_true", help="also report the host-loop single-chain rate and block-sparse mode")
    ap.add_argument("--no-aux", action="store_true",
                    help="skip the auxiliary legs (parity-mode eps stream, single-chain device Adam, GP-side entry points) reported under 'aux'")
    ap.add_argument("--shard-s", action="store_true",
                    help="fewer restarts than GPUs (SURVEY 8e): every step is ONE batch of --restarts evaluations sharded over the ranks "
                         "along the GP hyper-sample axis and the entropy sample chunks (strong scaling, bit-identical to 1 GPU)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking vbmc_elbo_batch call per step instead of the pipelined submit / collect")
    ap.add_argument("--check-launch", action="store_true",
                    help="rendezvous only: every rank reports (rank, pid, device) through an all-gather and rank 0 prints them; no GPU work")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("VBMC_DIST_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only for single-GPU smoke tests
    # the N > 1 code path: process group, library communicator, all-gather per step, strong leg.  VBMC_BENCH_FORCE_COMM=1 takes it with
    # ONE rank too (under torch.distributed.run --nproc-per-node 1): the only way to execute it on a one-GPU box (tests/test_gpu_comm.py)
    multi = world > 1 or (os.environ.get("VBMC_BENCH_FORCE_COMM") == "1" and "RANK" in os.environ)
    if args.gpus != world and not args.check_launch:
        # (round 5) a line whose n_gpus is not the N that was asked for is not a measurement of N GPUs: refuse
        raise SystemExit("bench.py: --gpus %d but the launcher created WORLD_SIZE=%d ranks (rank %d): start it with --nproc-per-node %d"
                         % (args.gpus, world, rank, args.gpus))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher created WORLD_SIZE=%d ranks; reporting the %d ranks that exist"
              % (args.gpus, world, world), file=sys.stderr)
    ndev_real = torch.cuda.device_count()
    ndev = max(ndev_real, 1)
    gpu = local_rank % ndev   # fewer devices than ranks (a 1-GPU box): ranks share a device -- gloo only, RCCL refuses duplicates
    if ndev_real:
        torch.cuda.set_device(gpu)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            if world > ndev_real:
                raise SystemExit("bench.py: %d ranks but %d GPU(s): RCCL needs one device per rank (set VBMC_DIST_BACKEND=gloo "
                                 "to share a device for a functional check)" % (world, ndev_real))
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", gpu)
    cdev = dev if (backend == "nccl" and ndev_real) else torch.device("cpu")  # where collective buffers live

    if args.check_launch:
        # (restart_offset, restart_stride): the device-RNG keys of this rank's share of the dealt batch -- restart r of the undivided batch
        # sits on rank r mod world (vbmc_elbo_batch_multi; misc/vpsieve_vbmc.m:74-83 is the loop that is dealt)
        me = torch.tensor([rank, os.getpid(), gpu if ndev_real else -1, rank % world, world], dtype=torch.int64, device=cdev)
        NF = 5
        if multi:
            allr = torch.empty(world * NF, dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(allr, me)
            dist.barrier()
        else:
            allr = me
        if rank == 0:
            rows = allr.cpu().numpy().reshape(world, NF)
            print(json.dumps({"check_launch": True, "n_gpus": world, "world_size_observed": int(rows.shape[0]), "backend": backend if world > 1 else None,
                              "spawned_by_bench": os.environ.get("VBMC_BENCH_SPAWNED") == "1",
                              "ranks": [{"rank": int(a), "pid": int(b), "device": int(c), "restart_offset": int(d), "restart_stride": int(e)}
                                        for a, b, c, d, e in rows]}), flush=True)
        if multi:
            dist.destroy_process_group()
        return

    import vbmc_amd

    D, N, K, Ns, S, Rr = args.D, args.N, args.K, args.Ns, args.S, args.restarts
    inp = synth_inputs(0, D, N, K, S)  # same GP on every rank (replicated, 25.6 MB with L)
    eng = vbmc_amd.Engine(gpu)
    # GP posterior (alpha, L per hyper-sample) from the build's own device gplite_post -- outside the timed region
    gp = vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    T = theta0.size
    rng = np.random.default_rng(100 + rank)
    thetas = np.asfortranarray(theta0[:, None] + 0.05 * rng.standard_normal((T, Rr)))  # R jittered restarts
    eng.device_gp(gp)  # one-off upload (already resident after gplite_post), outside the timed region
    gathered = torch.empty(world * Rr, dtype=torch.float64, device=cdev) if multi else None

    # the optimiser-loop objective [F,dF] = negelcbo_vbmc(theta,0,vp,gp,Ns,1,0) as the closure vpoptimize_vbmc.m:71 builds: vp, gp, flags
    # and buffers resolved once; every step still moves theta H2D and (F, dF) D2H
    objective = vbmc_amd.PreparedObjective(T, Rr, 0, vp, gp, Ns, 0, None, engine=eng)

    # N > 1: the restarts of ALL ranks form one batch of world x R candidates, identical on every rank, dealt r = g (mod world) by
    # the library (vbmc_elbo_batch_multi): rank g evaluates its R restarts and the ELCBO values of all world x R are all-gathered
    # device to device by RCCL reached from inside libvbmc_hip.so (no host hop, no torch tensor on the data path; torch.distributed
    # only carries the 128-byte RCCL id at start-up and the timing rows at the end).  VBMC_BENCH_EXCHANGE=torch forces the round-2
    # path (torch.distributed all_gather_into_tensor of the host vector) for A/B runs; it is also the fall-back if librccl cannot be
    # opened from the library.
    comm, gps_all, thetas_all, exchange = None, None, None, None
    if multi and not args.shard_s:
        exchange = "torch.distributed all_gather_into_tensor"
        # (gloo = ranks sharing one device for a functional check: RCCL refuses two ranks on a device)
        if os.environ.get("VBMC_BENCH_EXCHANGE", "library") != "torch" and backend == "nccl":
            try:
                from vbmc_amd.multi import Comm

                comm = Comm.from_torch(eng.ctx)
                gps_all = comm.upload_gp(gp)
                thetas_all = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((T, Rr * world)))
                exchange = "ncclAllGather inside libvbmc_hip.so (vbmc_elbo_batch_multi)"
            except Exception as e:  # noqa: BLE001
                comm = None
                exchange += " (library communicator unavailable: %s)" % str(e)[:120]

    # the N > 1 objective with everything resolved once (vbmc_amd.multi.PreparedMulti: the argument struct, the per-device sub-plans'
    # staging and the exchange blocks live across calls); pipelined like the one-GPU step: vbmc_elbo_multi_submit / _collect, two
    # batches in flight
    po_multi = comm.prepare(T, Rr * world, 0, vp, gps_all, Ns) if comm is not None else None

    def multi_finish(po, slot, ncols):
        F_, dF_ = po.collect(slot)
        np.argsort(F_, kind="stable")         # every rank: the identical sieve order (misc/vpsieve_vbmc.m:82)
        mine = np.arange(ncols)[rank::world]
        return {"F": F_, "dF": dF_[:, mine]}

    def multi_step(i, batch, po=None):
        po = po or po_multi
        po.submit(batch, seed=i, slot=0)
        return multi_finish(po, 0, batch.shape[1])

    def multi_run(po, batch, i0, n, pipe):
        o, pend = None, []
        for i in range(n):
            if not pipe:
                o = multi_step(i0 + i, batch, po)
                continue
            po.submit(batch, seed=i0 + i, slot=i % DEPTH)
            pend.append(i % DEPTH)
            if len(pend) == DEPTH:
                o = multi_finish(po, pend.pop(0), batch.shape[1])
        while pend:
            o = multi_finish(po, pend.pop(0), batch.shape[1])
        return o

    shard_ex = None
    if args.shard_s and multi:
        from vbmc_amd.dist import ShardExchange

        thetas = np.asfortranarray(theta0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((T, Rr)))  # the SAME batch on every rank
        shard_ex = ShardExchange(device=dev)

    def step(i):
        if shard_ex is not None:
            o = vbmc_amd.negelcbo_shard(thetas, 0, vp, gp, Ns, True, None, rank=rank, world=world, exchange=shard_ex, seed=i, engine=eng)
            return {"F": o["F"], "dF": o["dF"]}, None
        if comm is not None:
            return multi_step(i, thetas_all), None
        F_, dF_ = objective(thetas, seed=(rank << 32) + i)
        out = {"F": F_, "dF": dF_}
        if multi:
            f = torch.from_numpy(out["F"]).to(cdev)
            dist.all_gather_into_tensor(gathered, f)
            order = torch.argsort(gathered, stable=True)  # every rank: identical sieve order
            return out, order
        return out, None

    # The steps are independent batches (the sieve's candidates, misc/vpsieve_vbmc.m:74-78: batch i + 1 does not depend on the
    # result of batch i), so by default they go through the pipelined form of the same ABI call -- vbmc_elbo_submit /
    # vbmc_elbo_collect, four batches in flight on two streams: the host stages theta of step i + 1 while the device works on step i.  Every
    # step still moves its theta H2D, runs the full pass and moves (F, dF) D2H, and every step's results are consumed (the
    # all-gather + sort of the sieve when world > 1) before the timed region ends.  --sync-steps: one blocking call per step.
    pipelined = shard_ex is None and not args.sync_steps
    DEPTH = 2 if os.environ.get("VBMC_SLOT_STREAMS") == "0" else 4      # batches in flight: four slots on two streams (include/vbmc_hip.h, vbmc_elbo_submit)

    def finish(slot):
        F_, dF_ = objective.collect(slot)
        o = {"F": F_, "dF": dF_}
        if multi:
            dist.all_gather_into_tensor(gathered, torch.from_numpy(F_).to(cdev))
            torch.argsort(gathered, stable=True)  # every rank: identical sieve order
        return o

    def run_steps(i0, n):
        o, pend = None, []
        if comm is not None:
            return multi_run(po_multi, thetas_all, i0, n, pipelined)
        if not pipelined:
            for i in range(n):
                o, _ = step(i0 + i)
            return o
        for i in range(n):
            objective.submit(thetas, seed=(rank << 32) + i0 + i, slot=i % DEPTH)
            pend.append(i % DEPTH)
            if len(pend) == DEPTH:
                o = finish(pend.pop(0))
        while pend:
            o = finish(pend.pop(0))
        return o

    # untimed: a fresh box runs its first launches with cold code pages, first-touch pinned blocks and ramping clocks; a few steps
    # beyond the caller's --warmup keep a single slow step of that kind out of the timed region (observed once in ~25 runs on a
    # fresh box: one 7 ms step among twenty)
    WARM_EXTRA = 8                # reported as warmup_untimed_extra
    if multi:
        # the process group's first collective builds its RCCL communicator (hundreds of ms with the device idle): here, not in the
        # barrier that opens the timed region -- the warm-up steps below would otherwise be followed by an idle gap and the timed steps
        # start on a device that has clocked down (seen as ~1.5 ms on the first timed steps, forced one-rank runs of this path)
        dist.barrier()
    run_steps(0, WARM_EXTRA)
    run_steps(0, args.warmup)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run_steps(args.warmup, args.steps)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # ---- strong scaling (BASELINE configs[3]: the SAME 64 restarts over the GPUs): R restarts in total, R / world per rank
    strong = None
    if comm is not None and Rr >= world:
        batch = np.ascontiguousarray(thetas_all[:, :Rr], dtype=np.float64)
        batch = np.asfortranarray(batch)
        po_strong = comm.prepare(T, Rr, 0, vp, gps_all, Ns)
        multi_run(po_strong, batch, 5000, 4, pipelined)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        multi_run(po_strong, batch, 6000, args.steps, pipelined)
        torch.cuda.synchronize()
        dist.barrier()
        strong_s = time.perf_counter() - t1
        ts = torch.tensor([strong_s], dtype=torch.float64, device=cdev)
        alls = torch.empty(world, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(alls, ts)
        strong_s = float(alls.max())
        strong = {"value": Rr * args.steps / strong_s, "unit": "evals/s", "scaling": "strong", "restarts_total": Rr,
                  "restarts_per_gpu": Rr / world, "ms_per_step": 1e3 * strong_s / args.steps, "steps": args.steps}
    rank_rows = [[float(rank), float(gpu), elapsed]]
    if multi:
        mine = torch.tensor(rank_rows[0], dtype=torch.float64, device=cdev)
        allr = torch.empty(world * 3, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine)
        rank_rows = allr.cpu().numpy().reshape(world, 3).tolist()
        elapsed = max(r[2] for r in rank_rows)   # MAX over ranks
    assert np.all(np.isfinite(out["F"])) and np.all(np.isfinite(out["dF"]))

    # ---- everything below runs AFTER the timed region.  Each leg is guarded: a failing side measurement is recorded as
    # {"error": ...} in its place and never suppresses the headline line.
    roof = None
    extra = {}
    leg_errors = {}

    def leg(name, fn):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001
            leg_errors[name] = "%s: %s" % (type(e).__name__, str(e)[:300])
            try:
                eng.ctx.set_profiling(False)
            except Exception:  # noqa: BLE001
                pass
            return None

    M = Ns + (Ns % 2)
    aux_on = rank == 0 and world == 1 and not args.no_aux

    def roofline_leg():
        """HIP-event duration of the dominant kernel (rank 0): the kernel ALONE on the device (vbmc_ctx_set_profiling(ctx, 2): the
        expected log joint runs before it on the same stream).  In the timed region the kernel shares the chip with the other
        slot stream's small kernels and log joint, and in a blocking call with the log joint forked beside it: those durations
        overlap each other and do not price the kernel -- the blocking call's is reported next to it."""
        ent_ms, lj_ms, beside, walk_ms = [], [], [], []
        # (round 6) the launch FORM of the timed region: the steps of a pipeline run the chunk grid; a blocking call of this width would take
        # the walking launch (entropy_mfma.h: WALK) -- plan_restarts = R keeps the chunk grid (same shapes, same bits as the pipelined steps)
        eng.ctx.set_profiling(1)
        for i in range(10):
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=757 + i, engine=eng, plan_restarts=Rr)
            beside.append(eng.ctx.last_kernel_ms()[0])
        eng.ctx.set_profiling(2)
        for i in range(20):      # twenty launches: one disturbed launch in five moved the average by several per cent
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=777 + i, engine=eng, plan_restarts=Rr)
            a, b = eng.ctx.last_kernel_ms()
            ent_ms.append(a)
            lj_ms.append(b)
        for i in range(10):      # ... and the walking launch of the blocking call, alone on the device likewise
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=797 + i, engine=eng)
            walk_ms.append(eng.ctx.last_kernel_ms()[0])
        eng.ctx.set_profiling(False)
        extra["entropy_kernel_ms_beside_forked_logjoint"] = float(np.mean(beside))
        extra["entropy_kernel_ms_median_min_max"] = [float(np.median(ent_ms)), float(np.min(ent_ms)), float(np.max(ent_ms))]
        extra["blocking_call_entropy_kernel_ms"] = float(np.mean(walk_ms))     # (the walking launch where the library chose it)
        ent_ms, lj_ms = float(np.mean(ent_ms)), float(np.mean(lj_ms))
        f_ent, f_lj, P = algorithmic_flops(D, K, M, S, N)
        achieved = Rr * f_ent / (ent_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: hardware counters cannot be read from inside the run; they come from the
        # committed rocprofv3 --pmc passes of this same command (FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
        # MI355X_MICROARCH.md prescribes), stamped with the commit and a hash of the kernel sources they were taken at; a figure
        # whose hash no longer matches the sources in this tree is reported as stale
        traffic, traffic_src, traffic_stale = None, None, None
        here = os.path.dirname(os.path.abspath(__file__))
        pdir = os.path.join(here, "profiles")
        pmc_files = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc.json")) if os.path.isdir(pdir) else []
        if pmc_files and (D, N, K, S, Rr) == (10, 400, 50, 20, 64):
            with open(os.path.join(pdir, pmc_files[-1])) as f:
                pmc = json.load(f)
            traffic = pmc["hbm_bytes_per_launch"]
            if "walk_hbm_bytes_per_launch" in pmc:
                extra["blocking_call_entropy_kernel_traffic_bytes"] = pmc["walk_hbm_bytes_per_launch"]
            traffic_stale = pmc.get("kernel_source_sha256_16") != kernel_source_hash()
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; taken at commit %s)" % (pmc_files[-1], pmc.get("commit", "?"))
        extra["logjoint_kernel_ms"] = lj_ms
        return {"bound": "mfma", "kernel": entropy_kernel_label(D, K), "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                "algorithmic_bytes_per_launch": Rr * 8 * (2 * (D * K + K + D + K) + 2),
                "kernel_ms": ent_ms, "flops_per_launch": Rr * f_ent, "exp_per_launch": Rr * P,
                "note": "fp64 pipe bound: v_mfma_f64_16x16x4_f64 and fp64 VALU share one pipe on gfx950 (measured: no overlap), "
                        "dense fp64 peak 78.6 TFLOP/s for either.  achieved = algorithmic flops P(5D+7)+K*Ns(5D+4) per eval "
                        "(SURVEY 8d) x R / kernel time; the P fp64 exp evaluations (8 fp64 + 3 int VALU ops each here) are NOT counted. "
                        "traffic: device-RNG mode reads no O(Ns) data from HBM; the bytes are per-chunk partial records "
                        "(written once, reduced by k_ent_reduce) and the packed mixture parameters"}

    def adam_leg():
        """single-chain latency: the on-device Adam loop (vbmc_adam_batch)"""
        out_ = {}
        for Rc in (1, 2):
            x0 = thetas[:, :Rc].copy()
            vbmc_amd.fminadam_device(x0, 0, vp, gp, Ns, None, 0.0, 40, seed=5, engine=eng)  # warm-up
            t1 = time.perf_counter()
            _, _, _, _, its = vbmc_amd.fminadam_device(x0, 0, vp, gp, Ns, None, 0.0, 200, seed=6, engine=eng)
            out_["device_adam_R%d_evals_per_s" % Rc] = float(np.sum(its)) / (time.perf_counter() - t1)
        # ... and at the sample count VBMC itself runs its chains with (NSent = 100 K^(2/3): 28 per component at K = 50,
        # misc/vpsieve_vbmc.m:28 with the defaults of vbmc.m), soft bounds as misc/vpoptimize_vbmc.m passes them
        ns_v = int(np.ceil(100 * K ** (2.0 / 3.0) / K))
        vpb, tb = vbmc_amd.vpbounds(vp, {"X": inp["X"], "y": inp["y"]}, {"TolConLoss": 0.01, "TolWeight": 1e-2, "WeightPenalty": 0.1, "TolLength": 1e-6}, K)
        x0 = thetas[:, :1].copy()
        vbmc_amd.fminadam_device(x0, 0, vpb, gp, ns_v, tb, 0.0, 40, seed=5, engine=eng)
        t1 = time.perf_counter()
        _, _, _, _, its = vbmc_amd.fminadam_device(x0, 0, vpb, gp, ns_v, tb, 0.0, 200, seed=6, engine=eng)
        dt_ = time.perf_counter() - t1
        out_["device_adam_R1_vbmc_Ns"] = {"Ns_per_component": ns_v, "evals_per_s": float(np.sum(its)) / dt_, "us_per_iteration": 1e6 * dt_ / float(np.sum(its))}
        return out_

    def extras_leg():
        out_ = {}
        # one host round trip per evaluation (what utils/fminadam.m does through the shim), arguments resolved once
        obj1 = vbmc_amd.PreparedObjective(thetas.shape[0], 1, 0.0, vp, gp, Ns, 0, None, engine=eng)
        for i in range(10):
            obj1(thetas[:, :1], seed=800 + i)
        t1 = time.perf_counter()
        for i in range(200):
            obj1(thetas[:, :1], seed=900 + i)
        out_["host_loop_R1_evals_per_s"] = 200 / (time.perf_counter() - t1)
        # opt-in block-sparse mode (vbmc_elbo_args.sparse_cutoff = 100): same outputs to < 1e-13, component tiles whose
        # terms are < e^-100 of q(x) are skipped -- data dependent, NOT the headline value
        for i in range(2):
            sp = vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=40 + i, engine=eng, sparse_cutoff=100.0)
        dn = vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=41, engine=eng)
        t1 = time.perf_counter()
        for i in range(5):
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=50 + i, engine=eng, sparse_cutoff=100.0)
        out_["block_sparse"] = {"evals_per_s": 5 * Rr / (time.perf_counter() - t1), "cutoff": 100.0,
                                "max_rel_diff_vs_dense": float(np.max(np.abs(sp["dF"] - dn["dF"])) / np.max(np.abs(dn["dF"])))}
        return out_

    def eps_leg():
        """parity mode: every restart reads its own D x Ns/2 x K block of standard normals from HBM (the reference's randn
        stream, entmc_vbmc.m:53), already resident on the device -- R x 20 MB per launch at the headline shape"""
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        eps_d = torch.randn((Rr, K, M // 2, D), dtype=torch.float64, device=dev, generator=g)
        torch.cuda.synchronize()
        kw = dict(eps_device_ptr=eps_d.data_ptr(), eps_shared=False, engine=eng, outputs=("F", "dF"))
        # twelve untimed calls first: producing the draws (1.28 GB of randn, a synchronise) leaves the device idle long enough to
        # clock down, and two warm-up calls measured the ramp, not the kernel (2.47 ms; warmed and interleaved with the device-RNG
        # kernel, tools/eps_probe.py: 2.28 ms against 2.30)
        for _ in range(12):
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, **kw)
        eng.ctx.set_profiling(2)
        t1 = time.perf_counter()
        ems, rng_ms = [], []
        NE = 8
        for _ in range(NE):
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, **kw)
            ems.append(eng.ctx.last_kernel_ms()[0])
        dt_ = time.perf_counter() - t1
        for i in range(NE):      # the device-RNG kernel right after it, same clocks
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=900 + i, engine=eng, outputs=("F", "dF"))
            rng_ms.append(eng.ctx.last_kernel_ms()[0])
        eng.ctx.set_profiling(False)
        eps_bytes = Rr * K * (M // 2) * D * 8
        return {"evals_per_s": NE * Rr / dt_, "entropy_kernel_ms": float(np.median(ems)), "device_rng_kernel_ms": float(np.median(rng_ms)),
                "eps_bytes_per_launch": eps_bytes, "eps_stream_GBps": eps_bytes / (float(np.median(ems)) * 1e-3) / 1e9}

    def timeit(f, n=5, warm=1):
        """median wall time of n calls after `warm` warm-up calls (a single slow call -- a first-use allocation -- is not the rate)"""
        for _ in range(warm):
            f()
        ts = []
        for _ in range(n):
            t1 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t1)
        return float(np.median(ts))

    def gp_legs(aux):
        """the other entry points of the path at the same GP shape (wall time per call incl. H2D / D2H)"""
        Xs = 1.5 * np.random.default_rng(0).standard_normal((8192, D))
        st = {"ymax": float(np.max(inp["y"])), "VarianceRegularizedAcqFcn": True, "TolGPVar": 1e-4}
        gl = np.exp(np.mean(inp["hyp"][:D], axis=1))
        gpn = dict(gp, X_rescaled=inp["X"] / gl[None, :], sn2new=np.full(N, 0.05))
        stv = dict(st, gplengthscale=gl, ActiveImportanceSampling={"Xa": 1.2 * np.random.default_rng(2).standard_normal((100, D))})
        gpd = {"X": inp["X"], "y": inp["y"], "s2": None, "covfun": 1, "Ncov": D + 1, "noisefun": (1, 0, 0), "Nnoise": 1, "meanfun": 4,
               "Nmean": 2 * D + 1, "intmeanfun": 0}
        legs = [
            ("gplite_post_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, engine=eng), 5, warm=6)),
            ("gplite_post_resident_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.gplite_post(inp["hyp"], inp["X"], inp["y"], 1, 4, (1, 0, 0), None, need_L=False, engine=eng), 5, warm=6)),
            ("gplite_pred_8192_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.gplite_pred(gp, Xs, None, None, False, engine=eng), 5, warm=3)),
            ("eval_fullelcbo_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.negelcbo_vbmc(theta0, 0, vp, gp, 4096, 0, 1, nargout=11, engine=eng), 5)),
            ("diagvar_grad_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.negelcbo_vbmc(theta0, 1.0, vp, gp, 128, 1, 2, nargout=2, engine=eng), 5)),
            ("entlb_sieve_R250_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.negelcbo_batch(np.tile(theta0[:, None], (1, 250)), 0, vp, gp, 0, False, 0, engine=eng), 5)),
            ("acqwrapper_acqf_8192_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.acqwrapper_vbmc(Xs, vp, gp, st, False, "acqf_vbmc", None, engine=eng), 3)),
            ("acqwrapper_acqviqr_8192_Na100_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.acqwrapper_vbmc(Xs, vp, gpn, stv, False, "acqviqr_vbmc", None, engine=eng), 3)),
            # the importance-point sampler behind acqimiqr_vbmc (private/activeimportancesampling_vbmc.m:103-246): resampling + MCMC for
            # every hyper-sample, VBMC's default 100 + 100 + 100 points, all ensembles in lock-step with batched device predictions
            ("activeimportancesampling_imiqr_ms", lambda: 1e3 * timeit(lambda: vbmc_amd.activeimportancesampling_vbmc(
                vp, gpn, "acqimiqr_vbmc", None, {}, rng=np.random.default_rng(7), engine=eng), 3)),
        ]
        for B in (1, 64, 256):
            H = np.tile(inp["hyp"], (1, (B + S - 1) // S))[:, :B] + 0.01 * np.random.default_rng(1).standard_normal((inp["hyp"].shape[0], B))
            legs.append(("nlz_grad_B%d_evals_per_s" % B, lambda H=H, B=B: B / timeit(lambda: vbmc_amd.gplite_nlZ(H, gpd, engine=eng), 3)))
        for name, fn in legs:
            v = leg("aux." + name, fn)
            if v is not None:
                aux[name] = v
        aux["gplite_pred_8192_gflop"] = S * 8192.0 * N * N / 1e9    # S N* N^2 flops: two per multiply-add of the triangle inv(L') (sW Ks)

    def shape_leg(D_, N_, K_, Ns_, S_, R_, target, noisy, nsteps):
        """another BASELINE configuration on this GPU, same stepping as the headline (pipelined independent batches), with the
        HIP-event duration of its entropy kernel and the fraction of the fp64 peak that makes"""
        inp_ = synth_inputs(0, D_, N_, K_, S_, target, noisy)
        s2_ = np.ones(N_) if noisy else None
        gp_ = vbmc_amd.gplite_post(inp_["hyp"], inp_["X"], inp_["y"], 1, 4, (1, 1, 0) if noisy else (1, 0, 0), s2_, need_L=False, engine=eng)
        vp_ = vbmc_amd.make_vp(inp_["mu"], inp_["sigma"], inp_["lam"], eta=inp_["eta"])
        vp_["w"] = np.exp(inp_["eta"]) / np.sum(np.exp(inp_["eta"]))
        th0 = np.concatenate([inp_["mu"].reshape(-1, order="F"), np.log(inp_["sigma"]), np.log(inp_["lam"]), inp_["eta"]])
        th = np.asfortranarray(th0[:, None] + 0.05 * np.random.default_rng(100).standard_normal((th0.size, R_)))
        obj = vbmc_amd.PreparedObjective(th0.size, R_, 0, vp_, gp_, Ns_, 0, None, engine=eng)
        for _ in obj.stream([th] * 3, seeds=[1, 2, 3]):
            pass
        t1 = time.perf_counter()
        for F_, dF_ in obj.stream([th] * nsteps, seeds=list(range(10, 10 + nsteps))):
            pass
        dt_ = time.perf_counter() - t1
        assert np.all(np.isfinite(F_)) and np.all(np.isfinite(dF_))
        eng.ctx.set_profiling(2)
        ems = []
        for i in range(5):
            vbmc_amd.negelcbo_batch(th, 0, vp_, gp_, Ns_, True, 0, seed=50 + i, engine=eng, outputs=("F", "dF"))
            ems.append(eng.ctx.last_kernel_ms()[0])
        eng.ctx.set_profiling(False)
        M_ = Ns_ + (Ns_ % 2)
        f_ent, _, _ = algorithmic_flops(D_, K_, M_, S_, N_)
        ach = R_ * f_ent / (float(np.mean(ems)) * 1e-3) / 1e12
        return {"workload": "D=%d N=%d K=%d Ns=%d/component S=%d, R=%d, %s target%s" % (D_, N_, K_, Ns_, S_, R_, target, ", noisy (s2 = 1)" if noisy else ""),
                "evals_per_s": R_ * nsteps / dt_, "ms_per_step": 1e3 * dt_ / nsteps, "kernel": entropy_kernel_label(D_, K_),
                "entropy_kernel_ms": float(np.mean(ems)), "achieved_TFLOPs": ach, "frac": ach / FP64_PEAK_TFLOPS,
                # rocprofv3 evidence for this configuration (kernel trace + one PMC pass of this command line), newest committed summary
                "profile": newest_profile_with("configs[%d]" % (1 if D_ == 6 else 4))}

    def small_batch_leg():
        """BASELINE configs[3] is the SAME 64 restarts over 8 GPUs: 64 / G per device.  The rate of one device at R = 32, 16, 8 says
        what G = 2, 4, 8 devices can reach before a node is there (strong-scaling bound: G x rate(64 / G) / rate(64))."""
        out_ = {}
        f_ent, _, _ = algorithmic_flops(D, K, M, S, N)
        for Rc in (32, 16, 8, 4, 2, 1):
            obj = vbmc_amd.PreparedObjective(T, Rc, 0, vp, gp, Ns, 0, None, engine=eng)
            th = np.asfortranarray(thetas[:, :Rc])
            for _ in obj.stream([th] * 4, seeds=[1, 2, 3, 4]):
                pass
            t1 = time.perf_counter()
            for _ in obj.stream([th] * 20, seeds=list(range(10, 30))):
                pass
            dt_ = time.perf_counter() - t1
            out_["restarts_%d_evals_per_s" % Rc] = Rc * 20 / dt_
            eng.ctx.set_profiling(2)
            ems = []
            for i in range(5):
                obj(th, seed=40 + i)
                ems.append(eng.ctx.last_kernel_ms()[0])
            eng.ctx.set_profiling(False)
            out_["restarts_%d" % Rc] = {"ms_per_step": 1e3 * dt_ / 20, "entropy_kernel_ms": float(np.median(ems)),
                                        "frac": Rc * f_ent / (float(np.median(ems)) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
        return out_

    def comm_one_rank_leg():
        """The N > 1 step's own cost on ONE device: the same R = 64 and R = 8 steps through a one-rank communicator -- Comm.create_all(1):
        a real ncclAllGather on the context's stream, k_comm_pick, the gathered read-back -- pipelined like the headline step, beside
        the plain PreparedObjective step of the same R on the same context.  What the strong-scaling prediction 8 x rate(R = 8) /
        rate(R = 64) has to be discounted by before a node is there."""
        from vbmc_amd.multi import Comm

        out_ = {}
        comm1 = Comm.create_all(1)
        try:
            gps1 = comm1.upload_gp(gp)
            for Rc in (64, 8):
                if Rc > thetas.shape[1]:
                    continue
                th = np.asfortranarray(thetas[:, :Rc])
                po = comm1.prepare(T, Rc, 0, vp, gps1, Ns)
                obj = vbmc_amd.PreparedObjective(T, Rc, 0, vp, gp, Ns, 0, None, engine=eng)

                def run_comm(n, i0):
                    pend = []
                    for i in range(n):
                        po.submit(th, seed=i0 + i, slot=i & 3)
                        pend.append(i & 3)
                        if len(pend) == 4:
                            F_, _ = po.collect(pend.pop(0))
                            np.argsort(F_, kind="stable")
                    while pend:
                        F_, _ = po.collect(pend.pop(0))
                        np.argsort(F_, kind="stable")

                def run_plain(n, i0):
                    for _ in obj.stream([th] * n, seeds=list(range(i0, i0 + n))):
                        pass

                nst = 20 if Rc >= 32 else 40
                res = {}
                for name, fn in (("plain", run_plain), ("comm", run_comm)):
                    fn(4, 1)
                    ts = []
                    for rep_ in range(3):
                        t1 = time.perf_counter()
                        fn(nst, 100 * (rep_ + 1))
                        ts.append((time.perf_counter() - t1) / nst)
                    res[name] = 1e3 * float(np.median(ts))
                out_["R%d" % Rc] = {"plain_ms_per_step": res["plain"], "comm_ms_per_step": res["comm"],
                                    "overhead_pct": 100.0 * (res["comm"] - res["plain"]) / res["plain"],
                                    "comm_evals_per_s": Rc / (res["comm"] * 1e-3),
                                    # the N = 1 value of the multi-GPU code path against the plain line: within 3 %
                                    "within_3pct_of_plain": bool(res["comm"] <= 1.03 * res["plain"])}
            comm1.free_gp(gps1)
        finally:
            comm1.close()
        out_["path"] = "Comm.create_all(1): vbmc_elbo_multi_submit / _collect, ncclAllGather of [F | varG] on the communicator's exchange stream, four batches in flight on two pass streams"
        return out_

    def sync_leg():
        """the same steps as one blocking call each (round 1's stepping): what a DEPENDENT sequence of batches gets"""
        n = max(5, min(args.steps, 20))
        step(0)
        t1 = time.perf_counter()
        for i in range(n):
            step(1000 + i)
        return 1e3 * (time.perf_counter() - t1) / n

    if rank == 0 and pipelined and world == 1:
        v = leg("sync_steps", sync_leg)
        if v is not None:
            extra["blocking_call_ms_per_step"] = v
            extra["blocking_call_evals_per_s"] = Rr / (v * 1e-3)
    if rank == 0:
        roof = leg("roofline", roofline_leg)
        if args.extras:
            extra.update(leg("extras", extras_leg) or {})
    if aux_on:
        aux = {}
        aux.update(leg("aux.device_adam", adam_leg) or {})
        v = leg("aux.eps_streamed", eps_leg)
        if v is not None:
            aux["eps_streamed"] = v
        gp_legs(aux)
        aux.update(leg("aux.small_batches", small_batch_leg) or {})
        v = leg("aux.comm_one_rank", comm_one_rank_leg)
        if isinstance(v, dict) and isinstance(v.get("R64"), dict) and not v["R64"]["within_3pct_of_plain"]:
            leg_errors["aux.comm_one_rank.assert"] = "the one-rank communicator step at R = 64 is %.1f %% slower than the plain step (> 3 %%)" % v["R64"]["overhead_pct"]
        if v is not None:
            aux["comm_one_rank"] = v
        if (D, N, K, Ns, S) == (10, 400, 50, 10000, 20):     # the other single-GPU configurations of BASELINE.json, beside the headline
            v = leg("aux.config1", lambda: shape_leg(6, 200, 10, 1000, 8, 64, "student", False, 20))
            if v is not None:
                aux["config1"] = v
            v = leg("aux.config4", lambda: shape_leg(20, 800, 100, 20000, 20, 16, "lumpy", True, 6))
            if v is not None:
                aux["config4"] = v
                # configs[4]'s label "fp32 vs fp64 tolerance stress": the device-side A/B (S-step on v_mfma_f32_16x16x4_f32, tools/archive/f32s_eval.py)
                # is not part of the product build; its measured error and kernel time are read from the committed record
                try:
                    f32 = json.load(open(os.path.join(ROOT, "profiles", "r05_fp32_exponent.json")))
                    v["fp32_exponent"] = {
                        "source": "profiles/r05_fp32_exponent.json (tools/archive/f32s_eval.py on the GPU box: full configs[4] shape, both builds against the C port on the dumped device stream)",
                        **{sc: {"kernel_ms_fp64": r_["fp64"]["kernel_ms_R16"], "kernel_ms_fp32_exponent": r_["fp32_exponent"]["kernel_ms_R16"],
                                "F_relerr": r_["fp32_exponent"]["F_relerr_vs_c_port"], "H_relerr": r_["fp32_exponent"]["H_relerr_vs_c_port"],
                                "dF_block_relerr": r_["fp32_exponent"]["dF_block_relerr_vs_c_port"],
                                "fp64_product_H_relerr": r_["fp64"]["H_relerr_vs_c_port"]} for sc, r_ in f32.items()},
                        "adopted": False}
                except Exception as e_:  # noqa: BLE001
                    v["fp32_exponent"] = {"error": str(e_)[:200]}
        extra["aux"] = aux
    elif rank == 0 and args.eps_stream:
        extra["eps_streamed"] = leg("eps_streamed", eps_leg)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = leg("cpu_baseline", lambda: cpu_baseline(inp, gp, D, K, Ns))
        if cpu is not None:
            cpu["interpreted"] = leg("cpu_baseline.interpreted", lambda: interpreted_baseline(D, K, Ns))
    if leg_errors:
        extra["leg_errors"] = leg_errors

    if multi:
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
    if rank == 0:
        evals = (1 if shard_ex is not None else world) * Rr * args.steps
        line = {
            "metric": "ELBO+grad evals/sec (Ns=1e4, K=50, D=10, N=400)" if (D, N, K, Ns) == (10, 400, 50, 10000)
            else "ELBO+grad evals/sec (Ns=%d, K=%d, D=%d, N=%d)" % (Ns, K, D, N),
            "value": evals / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_untimed_extra": WARM_EXTRA,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if shard_ex is not None else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded lumpy 12-component target, SURVEY 8d); device Philox MC draws",
            "config": {"workload": "BASELINE configs[%d]: D=%d N=%d K=%d Ns=%d/component S=%d, R=%d restarts batched per GPU per step, "
                                   "value+gradient, beta=0, no variance" % (3 if world > 1 else 2, D, N, K, Ns, S, Rr),
                       "restarts_per_gpu": Rr,
                       "parallelism": ("hyper-sample x sample-chunk sharded x%d (one batch of %d), all-gather of the partial records" % (world, Rr))
                       if shard_ex is not None else ("restart-sharded x%d (restart r on rank r mod %d), all-gather of ELCBO" % (world, world)
                                                     if world > 1 else "one GPU"),
                       "stepping": (("pipelined: independent batches through vbmc_elbo_multi_submit / vbmc_elbo_multi_collect (the restarts dealt over "
                                     "the ranks, ncclAllGather of the ELCBO values inside the library), four in flight on two streams per device" if comm is not None else
                                     "pipelined: independent batches through vbmc_elbo_submit / vbmc_elbo_collect, four in flight on two streams") +
                                    "; every step moves its theta H2D and its (F, dF) D2H" if pipelined else "one blocking call per step")},
            "backend": ({"nccl": "nccl (RCCL)"}.get(backend, backend) if multi else None),
            "world_size_observed": (dist.get_world_size() if multi else 1),     # (== --gpus, asserted above and again here)
            "ranks": [{"rank": int(r[0]), "device": int(r[1]), "wall_s": r[2], "evals_per_s": Rr * args.steps / r[2]} for r in rank_rows],
            "roofline": roof, "cpu_baseline": cpu,
        }
        if multi:
            line["exchange"] = exchange
            line["strong"] = strong
        line.update(extra)
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
