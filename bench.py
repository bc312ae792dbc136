#!/usr/bin/env python
"""bench.py -- ELBO+grad evals/sec of the MI355X-native negelcbo path (BASELINE.json metric).

One "step" = one batched pass of the hot path: R = 64 independent negelcbo_vbmc(theta_r, 0, vp, gp,
Ns, 1, 0) evaluations (value + gradient; the Adam-loop call misc/vpoptimize_vbmc.m:71) on the
headline shape D=10, N=400, K=50, Ns=1e4 per component, S=20 GP hyper-samples (BASELINE.json
configs[2]), fresh Monte-Carlo draws every step (device Philox stream), theta H2D and (F, dF) D2H
inside the timed region, GP upload outside it.  N GPUs: one process per GPU, each rank owns its own
R restarts (weak scaling), and the per-step exchange is an all-gather of the R ELCBO values over RCCL
(misc/vpsieve_vbmc.m:81 sorts them; every rank ends with the full vector).

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6  # vendor vector-fp64 peak of MI355X (256 CU x 128 flop/clk x 2.4 GHz); see DESIGN.md


def synth_inputs(seed, D, N, K, S):
    """Synthetic GP + VP of the headline shape: lumpy 12-component target."""
    rng = np.random.default_rng(seed)
    X = 1.5 * rng.standard_normal((N, D))
    nc = 12
    mus = rng.uniform(-2, 2, size=(nc, D))
    sig = rng.uniform(0.3, 1.0, size=nc)
    wts = rng.dirichlet(np.ones(nc))
    lp = np.stack([np.log(wts[i]) - 0.5 * np.sum(((X - mus[i]) / sig[i]) ** 2, axis=1) - D * np.log(sig[i])
                   - 0.5 * D * np.log(2 * np.pi) for i in range(nc)])
    mx = lp.max(axis=0)
    y = mx + np.log(np.sum(np.exp(lp - mx), axis=0))
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


def synth_gp_posterior(inp, D):
    """alpha / L per hyper-sample for the synthetic GP (input generation, outside the timed path)."""
    X, y, hyp = inp["X"], inp["y"], inp["hyp"]
    N = X.shape[0]
    post = []
    for s in range(hyp.shape[1]):
        h = hyp[:, s]
        ell = np.exp(h[:D])
        sf2 = np.exp(2 * h[D])
        sn2 = np.exp(2 * h[D + 1])
        Z = X / ell
        d2 = np.maximum(np.sum(Z * Z, 1)[:, None] + np.sum(Z * Z, 1)[None, :] - 2 * Z @ Z.T, 0)
        Kmat = sf2 * np.exp(-0.5 * d2)
        m = h[D + 2] - 0.5 * np.sum(((X - h[D + 3: D + 3 + D]) / np.exp(h[D + 3 + D:])) ** 2, axis=1)
        Lc = np.linalg.cholesky(Kmat / sn2 + np.eye(N))
        alpha = np.linalg.solve(Lc.T, np.linalg.solve(Lc, y - m)) / sn2
        post.append({"hyp": h.copy(), "alpha": alpha, "sW": np.ones(N) / np.sqrt(sn2), "L": Lc.T.copy(),
                     "sn2_mult": 1.0, "Lchol": True})
    return {"X": X, "y": y, "s2": None, "covfun": 1, "meanfun": 4, "noisefun": (1, 0, 0), "Ncov": D + 1,
            "Nnoise": 1, "Nmean": 2 * D + 1, "post": post}


def algorithmic_flops(D, K, M, S, N):
    """entmc value+grad flops and gplogjoint flops per evaluation."""
    P = K * M * K
    ent = P * (5 * D + 7) + K * M * (5 * D + 4)
    lj = S * K * (16 * D * N + 5 * N)
    return ent, lj, P


def cpu_baseline(inp, gp, D, K, Ns_full, budget_s=20.0):
    """The oracle (NumPy restatement of the MATLAB path) timed on this host, bounded sample."""
    from oracle import vbmc_ref as R  # checker / baseline leg only

    vp = R.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    Ns = 200  # bounded sample: same shape, Ns reduced; entmc cost is linear in Ns
    rng = np.random.default_rng(0)
    eps = rng.standard_normal((K, Ns // 2, D))
    t0 = time.perf_counter()
    R.negelcbo_vbmc(theta, 0, vp, gp, Ns, True, 0, eps=eps)
    t_small = time.perf_counter() - t0
    # gplogjoint part does not scale with Ns: time it alone to extrapolate honestly
    t0 = time.perf_counter()
    R.gplogjoint(vp, gp, (1, 1, 1, 1), True, True, 0)
    t_lj = time.perf_counter() - t0
    t_ent = max(t_small - t_lj, 1e-9)
    reps = 1
    Ns2 = int(min(Ns_full, max(200, Ns * (budget_s - t_small) / (2 * t_ent))))
    Ns2 -= Ns2 % 2
    eps = rng.standard_normal((K, Ns2 // 2, D))
    t0 = time.perf_counter()
    for _ in range(reps):
        R.negelcbo_vbmc(theta, 0, vp, gp, Ns2, True, 0, eps=eps)
    t_run = (time.perf_counter() - t0) / reps
    t_full = t_lj + (t_run - t_lj) * (Ns_full / Ns2)
    return {"value": 1.0 / t_full, "unit": "ELBO+grad evals/s", "cores": 1, "kind": "port",
            "sample": "NumPy oracle (line-by-line restatement of the MATLAB path), 1 eval at Ns=%d of %d per component, "
                      "entropy part scaled linearly in Ns, gplogjoint part (%.2fs) unscaled; measured %.2fs" % (Ns2, Ns_full, t_lj, t_run)}


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
    ap.add_argument("--eps-stream", action="store_true", help="also time the parity mode (eps streamed from HBM)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import vbmc_amd

    D, N, K, Ns, S, Rr = args.D, args.N, args.K, args.Ns, args.S, args.restarts
    inp = synth_inputs(0, D, N, K, S)  # same GP on every rank (replicated, 25.6 MB with L)
    gp = synth_gp_posterior(inp, D)
    vp = vbmc_amd.make_vp(inp["mu"], inp["sigma"], inp["lam"], eta=inp["eta"])
    vp["w"] = np.exp(inp["eta"]) / np.sum(np.exp(inp["eta"]))
    theta0 = np.concatenate([inp["mu"].reshape(-1, order="F"), np.log(inp["sigma"]), np.log(inp["lam"]), inp["eta"]])
    T = theta0.size
    rng = np.random.default_rng(100 + rank)
    thetas = np.asfortranarray(theta0[:, None] + 0.05 * rng.standard_normal((T, Rr)))  # R jittered restarts
    eng = vbmc_amd.Engine(local_rank)
    eng.device_gp(gp)  # one-off upload, outside the timed region
    gathered = torch.empty(world * Rr, dtype=torch.float64, device=dev) if world > 1 else None

    def step(i):
        out = vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=(rank << 32) + i, engine=eng)
        if world > 1:
            f = torch.from_numpy(out["F"]).to(dev)
            dist.all_gather_into_tensor(gathered, f)
            order = torch.argsort(gathered, stable=True)  # every rank: identical sieve order
            return out, order
        return out, None

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out, _ = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert np.all(np.isfinite(out["F"])) and np.all(np.isfinite(out["dF"]))

    # ---- roofline leg (rank 0): HIP-event duration of the dominant kernel, outside the timed region
    roof = None
    extra = {}
    if rank == 0:
        eng.ctx.set_profiling(True)
        ent_ms, lj_ms = [], []
        for i in range(5):
            vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, seed=777 + i, engine=eng)
            a, b = eng.ctx.last_kernel_ms()
            ent_ms.append(a)
            lj_ms.append(b)
        eng.ctx.set_profiling(False)
        ent_ms, lj_ms = float(np.mean(ent_ms)), float(np.mean(lj_ms))
        M = Ns + (Ns % 2)
        f_ent, f_lj, P = algorithmic_flops(D, K, M, S, N)
        achieved = Rr * f_ent / (ent_ms * 1e-3) / 1e12
        roof = {"bound": "valu_f64", "kernel": "k_entropy<%d,grad>" % D, "achieved": achieved, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP64_VALU_PEAK_TFLOPS, "traffic": None,
                "kernel_ms": ent_ms, "flops_per_launch": Rr * f_ent, "exp_per_launch": Rr * P,
                "note": "fp64 vector-ALU bound (SURVEY 8d): algorithmic flops P(5D+7)+KM(5D+4) per eval, the "
                        "P fp64 exp evaluations (~20 flop-equivalents each) are NOT counted in achieved; "
                        "HBM is not the limiter (device RNG: 0 B/sample; eps-streamed mode: 8*D*Ns/2*K B/eval)"}
        extra["logjoint_kernel_ms"] = lj_ms
        if args.eps_stream:
            g = torch.Generator(device=dev)
            g.manual_seed(1)
            eps_d = torch.randn((K, M // 2, D), dtype=torch.float64, device=dev, generator=g)
            torch.cuda.synchronize()
            for _ in range(2):
                vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, eps_device_ptr=eps_d.data_ptr(), eps_shared=True, engine=eng)
            t1 = time.perf_counter()
            for _ in range(5):
                vbmc_amd.negelcbo_batch(thetas, 0, vp, gp, Ns, True, 0, eps_device_ptr=eps_d.data_ptr(), eps_shared=True, engine=eng)
            extra["eps_streamed_evals_per_s"] = 5 * Rr / (time.perf_counter() - t1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(inp, gp, D, K, Ns)

    if rank == 0:
        evals = world * Rr * args.steps
        line = {
            "metric": "ELBO+grad evals/sec (Ns=1e4, K=50, D=10, N=400)" if (D, N, K, Ns) == (10, 400, 50, 10000)
            else "ELBO+grad evals/sec (Ns=%d, K=%d, D=%d, N=%d)" % (Ns, K, D, N),
            "value": evals / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seeded lumpy 12-component target, SURVEY 8d); device Philox MC draws",
            "config": {"workload": "BASELINE configs[2]: D=%d N=%d K=%d Ns=%d/component S=%d, R=%d restarts batched per GPU per step, "
                                   "value+gradient, beta=0, no variance" % (D, N, K, Ns, S, Rr),
                       "restarts_per_gpu": Rr, "parallelism": "restart-sharded x%d, all-gather of ELCBO" % world},
            "roofline": roof, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
