// C-ABI entry points of libvbmc_hip.so for the ELBO objective (include/vbmc_hip.h).
// Host side: argument validation, device scratch, launches, one packed D2H per call.
#include <algorithm>
#include <cmath>
#include <mutex>

#include "elbo_kernels.h"
#include "var_kernels.h"
#include "gp_kernels.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
extern "C" int vbmc_abi_version(void) { return VBMC_ABI_VERSION; }

// with_aux: the second stream is created WITH the context (a caller's context: its blocking calls fork the expected log joint onto it);
// a child behind the pipeline slots never forks and holds one stream.  Created beside the first stream, not on first use: a
// low-priority stream that comes into being after the slot streams no longer yields to the first stream's kernel (the blocking call
// at the headline shape 2.50 -> 2.64 ms, tools/archive/r4_blocking_probe.py).
static vbmc_status ctx_create_impl(int device, void* stream, vbmc_ctx** out, bool with_aux) {
  if (!out) return VBMC_ERR_INVALID;
  *out = nullptr;
  // The pipelined step keeps up to five streams of a context busy (the two pass streams behind the four slots, the exchange stream of a
  // communicator, the context's own and its forked log joint); the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
  // queues (four by default), and two busy streams that share a queue run one after the other (profiles/r04_experiments.md section 11:
  // the R = 8 step 0.34 -> 0.51 ms when that happens).  The runtime reads the variable when it initialises -- at the first HIP call of the
  // process -- so it is raised HERE, ahead of this function's own first HIP call, for every host alike (the MEX gateway, ctypes, a C
  // caller).  A value already in the environment is kept; it has no effect when another HIP user (torch, say) initialised the runtime
  // first -- the streams are then placed among the queues that exist (stream_beside) -- and VBMC_HW_QUEUES=0 leaves the variable alone.
  // ONCE per process (it ran on every context creation -- setenv from a library races with getenv in the host's other
  // threads, the MATLAB JVM's among them -- so the window is the first creation only, before this library has made any HIP call).
  {
    static std::once_flag hq_once;
    std::call_once(hq_once, [] {
      const char* hq = getenv("VBMC_HW_QUEUES");
      if (!(hq && !strcmp(hq, "0"))) (void)setenv("GPU_MAX_HW_QUEUES", hq && atoi(hq) > 0 ? hq : "8", 0);
    });
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VBMC_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return VBMC_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return VBMC_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return VBMC_ERR_NO_DEVICE;  // gfx950 only, no fallback
  vbmc_ctx* ctx = new vbmc_ctx();
  ctx->device = device;
  ctx->num_cu = prop.multiProcessorCount;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return VBMC_ERR_HIP; }
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return VBMC_ERR_HIP; }
    ctx->own_stream = true;
  }
  for (auto& e : ctx->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete ctx; return VBMC_ERR_HIP; }
  {
    ctx->overlap = with_aux;
    if (with_aux) (void)ctx_aux(ctx);
    if (hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
      delete ctx;
      return VBMC_ERR_HIP;
    }
  }
  *out = ctx;
  return VBMC_OK;
}
extern "C" vbmc_status vbmc_ctx_create(int device, void* stream, vbmc_ctx** out) { return ctx_create_impl(device, stream, out, true); }

static void elbo_plan_free(void* plan);   // (ElboPlan is defined further down)

extern "C" void vbmc_ctx_destroy(vbmc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  DevBuf* bufs[] = {&ctx->theta, &ctx->prep, &ctx->entp, &ctx->ljpart, &ctx->entpart, &ctx->out,
                    &ctx->eps, &ctx->bnd, &ctx->vpfix, &ctx->misc, &ctx->varbuf, &ctx->zbuf};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto& b : ctx->pool)
    if (b.p) (void)hipFree(b.p);
  for (vbmc_gp* g : ctx->null_gp) vbmc_gp_free(ctx, g);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  for (int sl = 0; sl < 2; ++sl) {
    if (ctx->slot_pin[sl]) (void)hipHostFree(ctx->slot_pin[sl]);
    if (ctx->slot_ev[sl]) (void)hipEventDestroy(ctx->slot_ev[sl]);
    elbo_plan_free(ctx->slot_plan[sl]);
  }
  if (ctx->bounce) (void)hipHostFree(ctx->bounce);
  for (auto& e : ctx->bounce_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (vbmc_ctx* sc : ctx->slot_sub)
    if (sc) vbmc_ctx_destroy(sc);
  for (int sl = 0; sl < VBMC_SLOTS; ++sl) {
    if (ctx->slot_xev[sl]) (void)hipEventDestroy(ctx->slot_xev[sl]);
    if (ctx->slot_yev[sl]) (void)hipEventDestroy(ctx->slot_yev[sl]);
    if (ctx->slot_zev[sl]) (void)hipEventDestroy(ctx->slot_zev[sl]);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char* vbmc_last_error(const vbmc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" vbmc_status vbmc_ctx_synchronize(vbmc_ctx* ctx) {
  if (!ctx) return VBMC_ERR_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (vbmc_ctx* sc : ctx->slot_sub)
    if (sc) HIP_TRY(ctx, hipStreamSynchronize(sc->stream));
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_ctx_set_profiling(vbmc_ctx* ctx, int enable) {
  if (!ctx) return VBMC_ERR_INVALID;
  ctx->profiling = enable != 0;
  ctx->prof_alone = enable == 2;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_ctx_last_kernel_ms(vbmc_ctx* ctx, double* ent_ms, double* logjoint_ms) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (ent_ms) *ent_ms = ctx->last_ent_ms;
  if (logjoint_ms) *logjoint_ms = ctx->last_lj_ms;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_device_alloc(vbmc_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return VBMC_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMalloc(dptr, bytes));
  return VBMC_OK;
}
extern "C" vbmc_status vbmc_device_free(vbmc_ctx* ctx, void* dptr) {
  if (!ctx) return VBMC_ERR_INVALID;
  HIP_TRY(ctx, hipFree(dptr));
  return VBMC_OK;
}
extern "C" vbmc_status vbmc_memcpy_h2d(vbmc_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return VBMC_ERR_INVALID;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VBMC_OK;
}
extern "C" vbmc_status vbmc_memcpy_d2h(vbmc_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return VBMC_ERR_INVALID;
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
// GP upload
// ------------------------------------------------------------------------------------------
// derived per-sample constants of a surrogate (gplogjoint.m:99-121): S x GPC_STRIDE(D)
static void gp_constants(int D, int S, int Nhyp, int Ncov, int Nnoise, int meanfun, const double* hyp, double* gpc) {
  for (int s = 0; s < S; ++s) {
    const double* h = hyp + (size_t)s * Nhyp;
    double* g = gpc + (size_t)s * GPC_STRIDE(D);
    double sum_lnell = 0.0;
    for (int d = 0; d < D; ++d) {
      double ell = std::exp(h[d]);
      g[d] = ell * ell;
      sum_lnell += h[d];
    }
    const int mo = Ncov + Nnoise;
    for (int d = 0; d < D; ++d) {
      if (meanfun == 4) {
        g[D + d] = h[mo + 1 + d];
        double om = std::exp(h[mo + D + 1 + d]);
        g[2 * D + d] = 1.0 / (om * om);
      } else {
        g[D + d] = 0.0;
        g[2 * D + d] = 0.0;
      }
    }
    g[3 * D] = 2.0 * h[D] + sum_lnell;
    g[3 * D + 1] = meanfun > 0 ? h[mo] : 0.0;
  }
}

// L comes either from the host (L) or, for a posterior just computed on the device (vbmc_gp_post), from device memory:
// dL_chol holds the Cholesky factors (all samples), dL_inv the solves L\(L'\I) of the low-noise samples (negated here).
static vbmc_status gp_upload_impl(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int Ncov, int Nnoise,
                                  int meanfun, const double* X, const double* hyp, const double* alpha,
                                  const double* L, const double* dL_chol, const double* dL_inv, const double* sW1,
                                  const uint8_t* Lchol, vbmc_gp** out) {
  if (!ctx || !out) return VBMC_ERR_INVALID;
  *out = nullptr;
  if (N <= 0 || D <= 0 || S <= 0 || !X || !hyp || !alpha || !sW1)
    return set_err(ctx, VBMC_ERR_INVALID, "vbmc_gp_upload: N, D, S must be positive and X/hyp/alpha/sW1 non-null");
  if (D > VBMC_LIM_D) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "vbmc_gp_upload: D = %d > %d not accelerated", D, VBMC_LIM_D);
  if (!(meanfun == 0 || meanfun == 1 || meanfun == 4))
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "gplogjoint:UnsupportedMeanFun: meanfun %d not accelerated (0,1,4 are)", meanfun);
  if (Ncov != D + 1) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "only the SE-ARD covariance (Ncov = D+1) is accelerated");
  const int Nmean = meanfun == 0 ? 0 : (meanfun == 1 ? 1 : 2 * D + 1);
  if (Nhyp < Ncov + Nnoise + Nmean)
    return set_err(ctx, VBMC_ERR_INVALID, "gplite_post:dimmismatch: Nhyp = %d < Ncov+Nnoise+Nmean = %d", Nhyp, Ncov + Nnoise + Nmean);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  vbmc_gp* gp = new vbmc_gp();
  gp->N = N; gp->D = D; gp->S = S; gp->Nhyp = Nhyp; gp->Ncov = Ncov; gp->Nnoise = Nnoise; gp->meanfun = meanfun;
  gp->hyp_host.assign(hyp, hyp + (size_t)Nhyp * S);
  gp->sn2_eff.resize(S);
  gp->Lchol.resize(S);
  for (int s = 0; s < S; ++s) {
    gp->sn2_eff[s] = 1.0 / (sW1[s] * sW1[s]);  // gplogjoint.m:160
    gp->Lchol[s] = Lchol ? Lchol[s] : 1;
  }
  std::vector<double> gpc((size_t)S * GPC_STRIDE(D));
  gp_constants(D, S, Nhyp, Ncov, Nnoise, meanfun, hyp, gpc.data());
  // device blocks of a surrogate come from the context's pool (hipMalloc / hipFree cost ~0.1 ms each and serialise the
  // device: an append or a re-upload per acquired point would pay for a dozen of them)
  gp->pooled = true;
  // uploads are queued on the context's stream and awaited once at the end (every source outlives this function's last
  // synchronisation): a blocking hipMemcpy per array cost ~50 us each, seven of them per surrogate
  auto up = [&](double** dst, const double* src, size_t n) -> hipError_t {
    hipError_t e = pool_get(ctx, n * sizeof(double), (void**)dst);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(*dst, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  };
  hipError_t e = up(&gp->X, X, (size_t)N * D);
  if (e == hipSuccess) e = up(&gp->alpha, alpha, (size_t)N * S);
  if (e == hipSuccess) e = up(&gp->gpc, gpc.data(), gpc.size());
  if (e == hipSuccess) e = up(&gp->hyp, hyp, (size_t)Nhyp * S);
  if (e == hipSuccess && L) {
    e = up(&gp->L, L, (size_t)N * N * S);
    gp->hasL = true;
  } else if (e == hipSuccess && dL_chol) {
    e = pool_get(ctx, (size_t)N * N * S * sizeof(double), (void**)&gp->L);
    if (e == hipSuccess && !dL_inv) {       // every sample on the Cholesky branch: one contiguous copy
      e = hipMemcpyAsync(gp->L, dL_chol, (size_t)S * N * N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
    } else {
      for (int s = 0; s < S && e == hipSuccess; ++s) {
        const size_t off = (size_t)s * N * N;
        if (gp->Lchol[s]) e = hipMemcpyAsync(gp->L + off, dL_chol + off, (size_t)N * N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
        else hipLaunchKernelGGL(k_negate_copy, dim3((unsigned)(((size_t)N * N + 255) / 256)), dim3(256), 0, ctx->stream, (size_t)N * N, dL_inv + off, gp->L + off);
      }
    }
    gp->hasL = true;
  }
  if (e == hipSuccess) e = up(&gp->d_sn2, gp->sn2_eff.data(), (size_t)S);
  std::vector<double> mx(D);          // function scope: its copy is queued, not awaited, below
  if (e == hipSuccess) {
    for (int d = 0; d < D; ++d) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n) acc += X[n + (size_t)N * d];
      mx[d] = acc / N;
    }
    e = up(&gp->d_meanX, mx.data(), (size_t)D);
  }
  if (e == hipSuccess) {
    e = pool_get(ctx, (size_t)S, (void**)&gp->d_lchol);
    if (e == hipSuccess) e = hipMemcpyAsync(gp->d_lchol, gp->Lchol.data(), (size_t)S, hipMemcpyHostToDevice, ctx->stream);
  }
  if (e == hipSuccess && gp->hasL) {
    e = pool_get(ctx, (size_t)S * TRSM_NBLK(N) * 256 * sizeof(double), (void**)&gp->d_finv);
    if (e == hipSuccess) hipLaunchKernelGGL(k_diag_inv, dim3(TRSM_NBLK(N), S), dim3(64), 0, ctx->stream, N, gp->L, gp->d_lchol, gp->d_finv);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // the one wait for everything queued above
  else (void)hipStreamSynchronize(ctx->stream);                   // sources of queued copies die with this frame
  if (e != hipSuccess) {
    vbmc_gp_free(ctx, gp);
    return set_err(ctx, VBMC_ERR_HIP, "vbmc_gp_upload: %s", hipGetErrorString(e));
  }
  *out = gp;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_gp_upload(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int Ncov, int Nnoise,
                                      int meanfun, const double* X, const double* hyp, const double* alpha,
                                      const double* L, const double* sW1, const uint8_t* Lchol,
                                      vbmc_gp** out) {
  return gp_upload_impl(ctx, N, D, S, Nhyp, Ncov, Nnoise, meanfun, X, hyp, alpha, L, nullptr, nullptr, sW1, Lchol, out);
}

extern "C" void vbmc_gp_free(vbmc_ctx* ctx, vbmc_gp* gp) {
  if (!gp) return;
  if (ctx && gp->pooled) ctx_drain_slots(ctx);   // a pass in flight on a slot stream may still read the blocks the pool is about to hand out again
  // pooled blocks go back to the context's pool; without a live context (destroyed first) they are already gone with it
  // in_views: X, hyp, gpc, d_sn2, d_lchol, d_mult, d_meanX are windows into blk_in (a posterior assembled by vbmc_gp_post from
  // its own packed upload, abi_gp.hip)
  void* blocks[] = {gp->in_views ? nullptr : gp->X, gp->alpha, gp->L, gp->in_views ? nullptr : gp->gpc, gp->in_views ? nullptr : gp->hyp,
                    gp->in_views ? nullptr : gp->d_sn2, gp->in_views ? nullptr : gp->d_lchol, gp->in_views ? nullptr : gp->d_mult,
                    gp->d_finv, gp->d_tinv, gp->in_views ? nullptr : gp->d_meanX, gp->blk_in};
  for (void* b : blocks) {
    if (!b) continue;
    if (gp->pooled) { if (ctx) pool_put(ctx, b); }
    else (void)hipFree(b);
  }
  delete gp;
}

// ------------------------------------------------------------------------------------------
// template dispatch on the padded dimension DT
// ------------------------------------------------------------------------------------------
static int pick_dt(int D) {
  static const int dts[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 20, 24, 28, 32};
  for (int t : dts)
    if (t >= D) return t;
  return -1;
}

#define DISPATCH_DT(DTV, ...)                                                                      \
  switch (DTV) {                                                                                   \
    case 1: { constexpr int DT = 1; __VA_ARGS__; } break;                                                 \
    case 2: { constexpr int DT = 2; __VA_ARGS__; } break;                                                 \
    case 3: { constexpr int DT = 3; __VA_ARGS__; } break;                                                 \
    case 4: { constexpr int DT = 4; __VA_ARGS__; } break;                                                 \
    case 5: { constexpr int DT = 5; __VA_ARGS__; } break;                                                 \
    case 6: { constexpr int DT = 6; __VA_ARGS__; } break;                                                 \
    case 7: { constexpr int DT = 7; __VA_ARGS__; } break;                                                 \
    case 8: { constexpr int DT = 8; __VA_ARGS__; } break;                                                 \
    case 9: { constexpr int DT = 9; __VA_ARGS__; } break;                                                 \
    case 10: { constexpr int DT = 10; __VA_ARGS__; } break;                                               \
    case 11: { constexpr int DT = 11; __VA_ARGS__; } break;                                               \
    case 12: { constexpr int DT = 12; __VA_ARGS__; } break;                                               \
    case 14: { constexpr int DT = 14; __VA_ARGS__; } break;                                               \
    case 16: { constexpr int DT = 16; __VA_ARGS__; } break;                                               \
    case 18: { constexpr int DT = 18; __VA_ARGS__; } break;                                               \
    case 20: { constexpr int DT = 20; __VA_ARGS__; } break;                                               \
    case 24: { constexpr int DT = 24; __VA_ARGS__; } break;                                               \
    case 28: { constexpr int DT = 28; __VA_ARGS__; } break;                                               \
    case 32: { constexpr int DT = 32; __VA_ARGS__; } break;                                               \
    default: return set_err(ctx, VBMC_ERR_UNSUPPORTED, "D = %d not accelerated", dm.D);            \
  }

template <int DT>
static void launch_entropy(bool grad, dim3 grid, size_t lds, hipStream_t st, const EntArgs& ea) {
  if (grad) hipLaunchKernelGGL((k_entropy<DT, true>), grid, dim3(WAVE), lds, st, ea);
  else hipLaunchKernelGGL((k_entropy<DT, false>), grid, dim3(WAVE), lds, st, ea);
}

template <int DT>
static hipError_t set_entropy_lds(bool grad, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  if (grad) return hipFuncSetAttribute((const void*)k_entropy<DT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  return hipFuncSetAttribute((const void*)k_entropy<DT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}


// ---- MFMA entropy kernel dispatch: QS = ceil((D+2)/4) in 1..9 (one translation unit each,
// ent_mfma_inst.hip), KT = ceil(K/16) in 1..8 (1..4 for QS > 6: register budget)
extern "C" {
int vbmc_launch_ent_mfma_qs1(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs2(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs3(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs4(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs5(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs6(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs7(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs8(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_mfma_qs9(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs1(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs2(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs3(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs4(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs5(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs6(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs7(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs8(int, int, int, const EntArgs*);
int vbmc_occupancy_ent_mfma_qs9(int, int, int, const EntArgs*);
}
// ---- lane-per-sample entropy kernel (entropy_lane.h; round 6): small mixtures, K <= 16 and D <= 12.  One translation unit per padded
// dimension DT = 2, 4, .., 12 (ent_lane_inst.hip), KP = K rounded up to even.
extern "C" {
int vbmc_launch_ent_lane_dt2(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_lane_dt4(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_lane_dt6(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_lane_dt8(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_lane_dt10(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_launch_ent_lane_dt12(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
int vbmc_occupancy_ent_lane_dt2(int, int, const EntArgs*);
int vbmc_occupancy_ent_lane_dt4(int, int, const EntArgs*);
int vbmc_occupancy_ent_lane_dt6(int, int, const EntArgs*);
int vbmc_occupancy_ent_lane_dt8(int, int, const EntArgs*);
int vbmc_occupancy_ent_lane_dt10(int, int, const EntArgs*);
int vbmc_occupancy_ent_lane_dt12(int, int, const EntArgs*);
}
#define ENT_LANE_WAVES_HOST 4     // = ENT_LANE_WAVES (entropy_lane.h)
// the role's staged inputs (entropy_lane.h: ent_lane_role_lds) must fit the launch's dynamic LDS; larger training sets keep the separate log-joint kernel
static bool lane_role_fits(int D, int K, int N, int S) {
  return ((size_t)((N + 63) & ~63) * (D + ENT_LANE_WAVES_HOST) + (size_t)S * GPC_STRIDE(D) + (size_t)VpLayout{D, K}.stride() + D) * sizeof(double) <= 48 * 1024;
}
// The class: K <= 16, D <= 12, dense -- minus the corner where the two signs' densities and the gradient accumulators of a lane do not
// fit 256 registers (DT = 12 with KP >= 10, DT = 10 with KP >= 12: built for one wave per SIMD those kernels take 2-3x the matrix-core
// kernel's time, tools/run_lane_sweep.sh; they are not instantiated).
static bool lane_entropy_fits(int D, int K, double cutoff) {
  const int dt = 2 * ((D + 1) / 2), kp = 2 * ((K + 1) / 2);
  return D >= 1 && D <= 12 && K >= 1 && K <= 16 && !(cutoff > 0.0) && !((dt >= 12 && kp >= 10) || (dt >= 10 && kp >= 12));
}
// ... and a batch wide enough to fill the chip with 64-sample tiles: a single chain (K R tiles < ~100) is bound by the launch's own
// latencies, where the matrix-core kernel's role splits the training set over more waves (6 % to 10 % faster there)
#define ENT_LANE_MIN_TILES 96
static bool launch_entropy_lane(int D, int K, bool grad, dim3 g, hipStream_t st, const EntArgs& ea) {
  typedef int (*fn_t)(int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
  static const fn_t fns[6] = {vbmc_launch_ent_lane_dt2, vbmc_launch_ent_lane_dt4, vbmc_launch_ent_lane_dt6,
                              vbmc_launch_ent_lane_dt8, vbmc_launch_ent_lane_dt10, vbmc_launch_ent_lane_dt12};
  return fns[(D + 1) / 2 - 1](2 * ((K + 1) / 2), grad ? 1 : 0, g.x, g.y, g.z, (void*)st, &ea) == 0;
}
static int entropy_lane_occupancy(int D, int K, bool grad) {
  typedef int (*fn_t)(int, int, const EntArgs*);
  static const fn_t fns[6] = {vbmc_occupancy_ent_lane_dt2, vbmc_occupancy_ent_lane_dt4, vbmc_occupancy_ent_lane_dt6,
                              vbmc_occupancy_ent_lane_dt8, vbmc_occupancy_ent_lane_dt10, vbmc_occupancy_ent_lane_dt12};
  static int cache[6][9][2];      // 0: not asked yet
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int& c = cache[(D + 1) / 2 - 1][(K + 1) / 2][grad ? 1 : 0];
  if (c == 0) { EntArgs q{}; const int nb = fns[(D + 1) / 2 - 1](2 * ((K + 1) / 2), grad ? 1 : 0, &q); c = nb > 0 ? nb : -1; }
  return c;
}

// Waves per workgroup for 64 < K <= 128 (tools/tune_sweep.py, round 2): two -- four are 10-30 % slower (more exchange and barrier
// coupling) -- EXCEPT where the two-wave kernel with four k-tiles per wave and a wide operand (D >= 15) spills its way down:
// there four waves with two k-tiles each fit their registers (D = 24, K = 128: 3.4 vs 5.7 ms; D = 20, K = 128: 3.9 vs 5.0;
// D = 20, K = 100 the other way: 50 vs 57 ms at configs[4]).
// K <= 64: one wave per workgroup.  Round 3 (tools/hv_small_sweep.py -> profiles/r03_hv_small.md) tried two waves with two k-tiles
// each where the one-wave kernel with four k-tiles spills: once those kernels were rebuilt for ONE wave per SIMD (512 registers,
// VBMC_ENT_ONE_WAVE in entropy_mfma.h: 11-23 % faster) the split only wins at K = 57..64 for D >= 31 (6 %) and costs 6-85 % everywhere else.
static int ent_hv_small(int qs, int K) { return (qs >= 9 && K > 52) ? 2 : 1; }   // (round 4, with the shared even part in the two-wave kernels: K = 53..56 at D >= 31 too, 0.77 of the one-wave time)
static int ent_hv_mid(int qs, int K) { return (K > 96 && (qs >= 7 || (qs >= 5 && K > 112))) ? 4 : 2; }
static bool launch_entropy_mfma(int qs, int kt, int hv, bool grad, dim3 g, hipStream_t st, const EntArgs& ea) {
  typedef int (*fn_t)(int, int, int, unsigned, unsigned, unsigned, void*, const EntArgs*);
  static const fn_t fns[9] = {vbmc_launch_ent_mfma_qs1, vbmc_launch_ent_mfma_qs2, vbmc_launch_ent_mfma_qs3,
                              vbmc_launch_ent_mfma_qs4, vbmc_launch_ent_mfma_qs5, vbmc_launch_ent_mfma_qs6,
                              vbmc_launch_ent_mfma_qs7, vbmc_launch_ent_mfma_qs8, vbmc_launch_ent_mfma_qs9};
  if (qs < 1 || qs > 9) return false;
  return fns[qs - 1](kt, grad ? 1 : 0, hv, g.x, g.y, g.z, (void*)st, &ea) == 0;
}
// workgroups of the chosen instantiation per compute unit (registers and LDS: hipOccupancyMaxActiveBlocksPerMultiprocessor), cached
static int entropy_mfma_occupancy(int qs, int kt, int hv, bool grad, const EntArgs& ea) {
  typedef int (*fn_t)(int, int, int, const EntArgs*);
  static const fn_t fns[9] = {vbmc_occupancy_ent_mfma_qs1, vbmc_occupancy_ent_mfma_qs2, vbmc_occupancy_ent_mfma_qs3,
                              vbmc_occupancy_ent_mfma_qs4, vbmc_occupancy_ent_mfma_qs5, vbmc_occupancy_ent_mfma_qs6,
                              vbmc_occupancy_ent_mfma_qs7, vbmc_occupancy_ent_mfma_qs8, vbmc_occupancy_ent_mfma_qs9};
  if (qs < 1 || qs > 9) return -1;
  struct Key { int qs, kt, hv, grad, co, sparse, ldsk; };
  static std::vector<std::pair<Key, int>> cache;
  static std::mutex mu;                           // contexts of several host threads share the table
  std::lock_guard<std::mutex> lock(mu);
  const Key k{qs, kt, hv, grad ? 1 : 0, ea.lj.rows > 0 ? 1 : 0, ea.cutoff > 0.0 ? 1 : 0, ea.K * (ea.D + ENTP_EXTRA)};
  for (const auto& c : cache)
    if (c.first.qs == k.qs && c.first.kt == k.kt && c.first.hv == k.hv && c.first.grad == k.grad && c.first.co == k.co &&
        c.first.sparse == k.sparse && c.first.ldsk == k.ldsk) return c.second;
  const int nb = fns[qs - 1](kt, grad ? 1 : 0, hv, &ea);
  cache.push_back({k, nb});
  return nb;
}
// K <= 64: one wave per (chunk, component, restart) with kt = ceil(K/16) k-tiles; larger mixtures split their components
// over the hv = 2 or 4 waves of a workgroup, kt = ceil(ceil(K/hv)/16) <= 4: two waves up to K = 128, four up to K = 256.
// VBMC_ENT_HV = 2 / 4 forces the split where both fit (A/B runs).  D <= 34 (qs <= 9).
// hv + 16: every wave runs kt = Kh / 16 full k-tiles and its Kh mod 16 <= 4 remaining components (Kh = components per wave) as a
// lane-layout TAIL instead of a k-tile of their own (entropy_mfma.h, TL = values per lane): one value for up to 4 components
// (K = 17..20, 33..36, 49..52 on one wave, 66..72 and 98..104 on two, 130..144 and 194..208 on four), two for 5..8 where that
// kernel keeps its registers (tail8_ok below).  VBMC_ENT_TAIL=0 keeps the padded k-tile (A/B runs);
// the block-sparse mode (cutoff > 0) always does.
static bool mfma_entropy_fits(int D, int K, double cutoff, int* qs_out, int* kt_out, int* hv_out) {
  const int qs = (D + 2 + 3) / 4;
  int hv = K <= 64 ? ent_hv_small(qs, K) : (K <= 128 ? ent_hv_mid(qs, K) : (K <= 256 ? 4 : 8));   // (round 5: eight waves for 256 < K <= 512, full k-tiles only)
  if (K > 64 && K <= 128)
    if (const char* f = getenv("VBMC_ENT_HV")) { const int v = atoi(f); if (v == 2 || v == 4) hv = v; }
  if (K > 32 && K <= 64)
    if (const char* f = getenv("VBMC_ENT_HV")) { const int v = atoi(f); if (v == 1 || v == 2) hv = v; }
  int kt = (((K + hv - 1) / hv) + 15) / 16;
  static const bool tail_on = [] { const char* e = getenv("VBMC_ENT_TAIL"); return !(e && !strcmp(e, "0")); }();
  {
    const int Kh = (K + hv - 1) / hv;      // components of the first waves (the last one may hold fewer: its tail lanes idle)
    static const int tail_max = [] { const char* e = getenv("VBMC_ENT_TAIL"); return e ? atoi(e) : 2; }();   // 0 / 1 / 2 values per lane at most (A/B runs)
    const int tl = (Kh % 16 + 3) / 4;         // tail values per lane that would be needed: 1 for 1..4 components, 2 for 5..8
    // two values per lane pay except where that kernel runs out of registers (tools/tail_sweep.py: 57 shapes x tail limit 0 / 1 / 2)
    const int ktf = Kh / 16, rem = Kh % 16;
    const bool tail8_ok = ktf == 1 || (hv == 1 && ktf == 2 && (qs >= 5 || rem <= 6)) || (hv == 1 && ktf == 3 && qs <= 4) ||
                          (hv > 1 && ktf == 2) || (hv > 1 && ktf == 3 && qs >= 5);
    if (tail_on && hv != 8 && Kh > 16 && tl >= 1 && tl <= tail_max && (tl == 1 || (tl == 2 && tail8_ok)) && !(cutoff > 0.0) && !(hv > 1 && ktf < 2)) {
      kt = ktf;
      hv += 16 * tl;
    }
  }
  *qs_out = qs; *kt_out = kt; *hv_out = hv;
  return qs >= 1 && qs <= 9 && K >= 1 && K <= 512 && kt >= 1 && kt <= 4 && !(hv == 2 && kt < 2) && !(hv == 4 && kt < 2) && !(hv == 8 && kt < 3) && !(hv > 16 && kt > 3);
}

// ------------------------------------------------------------------------------------------
// vbmc_elbo_batch
// ------------------------------------------------------------------------------------------
// Everything one evaluation pass needs, resolved once: dims, device pointers, launch geometry.
// Small transfers between the pinned staging block and device memory by a kernel instead of the copy engines: the pinned block
// (hipHostMalloc: mapped into the device's address space, coherent) is read / written by the kernel directly over the host link.
// A DMA-engine copy between two kernels of a stream costs 50-100 us of hand-over latency each way (measured: 190-215 us between
// k_finalize of one batch and k_prep of the next at the headline shape, where the copies themselves are 0.3 MB); a kernel
// launch costs ~10 us.  Larger transfers use the copy engines.
#define COPY_KERNEL_MAX_BYTES ((size_t)4 << 20)
__global__ void k_copy_f64(size_t n, const double* __restrict__ src, double* __restrict__ dst) {
  VB_SMALL_PRIO();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// rows of `width` doubles, row r at src + r * stride -> dst + r * stride (same stride on both sides)
__global__ void k_copy_rows_f64(int rows, size_t stride, size_t width, const double* __restrict__ src, double* __restrict__ dst) {
  VB_SMALL_PRIO();
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (size_t)gridDim.x * blockDim.x)
      dst[(size_t)r * stride + i] = src[(size_t)r * stride + i];
}
// separate_K outputs straight into the caller's layouts, written to the pinned block by the kernel: I_sk (S x K x R, s fastest)
// from the log-joint records, J_sjk (S x K x K x R) from the variance matrices (diagonal approximation: only J_kk is set,
// gplogjoint.m:283).  Replaces two pageable D2H copies and a host-side transposition (~0.15 ms of a 0.34 ms eval_fullelcbo).
__global__ void k_pack_sepk(int R, int S, int K, int LJS, int diag_only, const double* __restrict__ lj, const double* __restrict__ J,
                            double* __restrict__ hI, double* __restrict__ hJ) {
  const size_t nI = (size_t)R * S * K, nJ = J ? (size_t)R * S * K * K : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nI + nJ; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nI) {
      const int s = (int)(i % S), k = (int)((i / S) % K), r = (int)(i / ((size_t)S * K));
      hI[i] = lj[(((size_t)r * S + s) * K + k) * LJS];
    } else {
      const size_t q = i - nI;
      const int s = (int)(q % S), j = (int)((q / S) % K), k = (int)((q / ((size_t)S * K)) % K), r = (int)(q / ((size_t)S * K * K));
      hJ[q] = (diag_only && j != k) ? 0.0 : J[(((size_t)r * S + s) * K + k) * K + j];
    }
  }
}

static bool copy_by_kernel(size_t bytes) { return bytes <= COPY_KERNEL_MAX_BYTES; }

// inv(L') of the Lchol samples, once per surrogate (gplite_pred's triangular product and the full-variance path use it); null when
// a 16-row tile of it does not fit the prediction kernel's LDS (N > 1248: those paths fall back to substitutions)
static vbmc_status ensure_tinv(vbmc_ctx* ctx, const vbmc_gp* gp, bool* have) {
  const int N = gp->N, S = gp->S;
  const int Np = ((N + 15) >> 4) << 4, nblk = Np >> 4;
  *have = false;
  if (!gp->hasL || (size_t)16 * Np * 8 > PRED_LDS_MAX || nblk > PRED_MAXG || trsm_cw_for(N) != 16) return VBMC_OK;
  if (!gp->d_tinv) {
    double* t = nullptr;
    HIP_TRY(ctx, gp->pooled ? pool_get(ctx, (size_t)S * N * N * 8, (void**)&t) : hipMalloc((void**)&t, (size_t)S * N * N * 8));
    hipError_t e_ = tri_inverse_launch(ctx->stream, N, S, gp->L, gp->d_finv, gp->d_lchol, t, 0);
    if (e_ != hipSuccess) {
      if (gp->pooled) pool_put(ctx, t); else (void)hipFree(t);
      return set_err(ctx, VBMC_ERR_HIP, "inv(L') failed: %s", hipGetErrorString(e_));
    }
    gp->d_tinv = t;
  }
  *have = true;
  return VBMC_OK;
}

struct ElboPlan {
  ElboDims dm{};
  int compute_grad = 0, compute_var = 0, dt = 0;
  double beta = 0.0;
  bool mc = false, has_bnd = false, use_mfma = false, use_lane = false, vgrad = false, any_nochol = false, needX = false, fin_big = false, tri_gemm = false;
  double *d_finbig = nullptr, *d_gamma = nullptr;
  int Mh = 0, C = 1, tpc = 1, ncol = 1, qs = 0, kt = 0, hv = 1, var_stride = 0;
  int no_jacobian = 0;
  int Rp = 0;                // the restarts the launch shapes are chosen for: R, or the undivided batch's (vbmc_elbo_args.plan_restarts)
  int walk_tpw = 0, walk_nw = 0;   // > 0: the matrix-core entropy kernel WALKS (entropy_mfma.h): tiles per wave, waves of the launch
  int co_c1 = 0, co_c2 = 0, co_tpc2 = 0;   // co_c2 > 0: two chunk classes (EntArgs: the role's workgroups take the second, shorter one)
  double* d_dvs = nullptr;   // per-hyper-sample variance gradient block (dvarG_s), pooled for the call
  bool lj_records = false;   // the caller reads per-hyper-sample log-joint records (separate_K, G_s / varG_s, the variance kernels)
  int r0 = 0, rstride = 1;   // device-RNG key of restart r: r0 + r * rstride (vbmc_elbo_args.restart_offset / restart_stride)
  size_t n_theta = 0, n_up = 0, out_n = 0, ent_lds = 0, tlds = 0, n_sepk = 0;
  double *d_theta = nullptr, *d_fix = nullptr, *d_delta2 = nullptr, *d_bnd = nullptr;
  double *d_ljbar = nullptr;
  double *d_vpd = nullptr, *d_entp = nullptr, *d_lj = nullptr, *d_out = nullptr, *d_part = nullptr, *d_red = nullptr;
  double *d_Z = nullptr, *d_X = nullptr, *d_J = nullptr, *d_vg = nullptr, *d_var = nullptr, *d_vs = nullptr;   // d_vs: per-sample gradient vectors (k_var_sample)
  const double* d_eps = nullptr;
  double* out_direct = nullptr;   // the pinned result block itself: the finalize kernel writes the records there (small pipelined passes)
  long long eps_stride_r = 0;
  double TolCon = 0.0, WeightThreshold = 0.0, WeightPenalty = 0.0, cutoff = 0.0;
};

// Splits of the training set per cell group of the log-joint role: per-workgroup set-up (exp table, tau / log tau) against the length of the
// dependent loop over the training set.  Single chain at the headline shape (260 cell groups, 25 slabs of 16 points), us per Adam
// iteration: 1 split 46.6, 2: 41.1, 3: 41.2, 4: 43.0, 6: 45.4, 8: 52.2 (more workgroups than wave slots) -> about 640 role workgroups per restart
static int lj_co_nsplit(const vbmc_ctx* ctx, const ElboPlan& P) {
  const int K = P.dm.K, S = P.dm.S, R = P.dm.R;
  const long long cells = (long long)((K + 3) / 4) * S;   // per restart: a restart's bits do not depend on the batch it is in
  const int slabs = (P.dm.N + 15) / 16;
  int ns = (int)std::max<long long>(1, std::min<long long>(std::min(LJ_CO_SPLIT, slabs), (640 + cells / 2) / cells));
  // (round 5) a BATCH wide enough that the record buffer holds one record per hyper-sample (elbo_plan: ljrec): one role workgroup per
  // cell group -- the restarts supply the parallelism the splits supply to a single chain
  if ((long long)S * std::min(R, P.Rp) >= ctx->num_cu / 2) ns = 1;
  if (P.use_lane) ns = 1;      // (the lane kernel's role walks the LDS-staged training set whole)
  return ns;
}

// Does the expected log joint run as a role of the MFMA entropy launch (entropy_mfma.h CO = true; see elbo_enqueue)?  The part of the
// answer that elbo_plan needs too (the chunk model asks for the occupancy of the kernel that will run).
static bool lj_co_shape(const vbmc_ctx* ctx, const ElboPlan& P) {
  static const bool co_off = [] { const char* e = getenv("VBMC_LJ_CO"); return e && !strcmp(e, "0"); }();
  const char* ljf = getenv("VBMC_LJ_KERNEL");
  const int Rp = P.Rp > 0 ? P.Rp : P.dm.R;        // (plan_restarts: the undivided batch decides)
  const long long SR = (long long)P.dm.S * Rp;
  const bool lj_force_mfma = ljf && !strcmp(ljf, "mfma");
  // ... and only where the ENTROPY launch is small too (K R waves per sample chunk: a single chain has 50, a batch of restarts over one
  // hyper-sample -- or an entropy-only evaluation, whose surrogate is a one-point stand-in -- can fill the chip by itself and wants the
  // kernels built for occupancy, not these)
  const long long KR = (long long)P.dm.K * (P.Rp > P.dm.R ? P.Rp : (long long)P.dm.R * P.rstride);
  const double lim_sr = 0.5, lim_kr = 2.0;     // the two width limits in units of the chip's compute units (other values measured in round 5: profiles/r05_experiments.md section 8)
  return !co_off && !lj_force_mfma && P.mc && P.use_mfma && (P.hv & 15) == 1 && P.qs <= 8 && !(P.cutoff > 0.0) && P.compute_grad &&
         !P.lj_records && SR < lim_sr * ctx->num_cu && (P.Rp > P.dm.R || SR * P.rstride < lim_sr * ctx->num_cu) &&   // (the undivided batch's choice when the restarts are dealt over devices)
         KR < lim_kr * ctx->num_cu && P.dm.N > 1 &&
         // (round 4) ... and small in WORK, not only in width: with many sample tiles per wave (Ns = 1e4 per component: 313 tiles per
         // (component, restart)) the role's workgroups delay an entropy launch that fills the chip by itself -- R = 4 at the headline
         // shape: 0.242 ms with the role, 0.221 without; equal at R = 2 -- while at the optimiser's own sample counts (Ns = 28..400)
         // the role wins by 13-24 % for R <= 4.  The bound: sixteen sample tiles per resident wave slot.
         KR * ((P.Mh + 15) / 16) <= 16LL * 8 * ctx->num_cu;
}

// dynamic LDS of k_var_final: reduction scratch, two S-vectors, five T-vectors (only with a gradient), two K-vectors
#define VAR_FINAL_LDS(S_, K_, Tg_) ((VARFIN_THREADS + 2 * (size_t)(S_) + 5 * (size_t)(Tg_) + 2 * (size_t)(K_) + 8) * sizeof(double))
// Validation (reference error ids), one H2D of theta | fixed vp | delta^2 | bounds, scratch sizing.
static vbmc_status elbo_plan(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, ElboPlan& P, int chunk_world = 0, bool pipelined = false) {
  if (!gp || !a) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_batch: null gp/args");
  if (a->struct_size != sizeof(vbmc_elbo_args))
    return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_args.struct_size %u != %zu (ABI mismatch)", a->struct_size, sizeof(vbmc_elbo_args));
  ElboDims& dm = P.dm;
  dm.D = a->D; dm.K = a->K; dm.R = a->R; dm.S = gp->S; dm.N = gp->N;
  if (dm.D != gp->D) return set_err(ctx, VBMC_ERR_INVALID, "vp.D = %d but gp has D = %d", dm.D, gp->D);
  if (dm.K <= 0 || dm.R <= 0) return set_err(ctx, VBMC_ERR_INVALID, "K and R must be positive");
  if (dm.K > VBMC_LIM_K) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "K = %d > %d not accelerated", dm.K, VBMC_LIM_K);
  const int D = dm.D, K = dm.K, R = dm.R, S = dm.S;
  int T = 0;
  for (int g = 0; g < 4; ++g) dm.opt[g] = a->optimize[g] ? 1 : 0;
  dm.off_mu = dm.off_sigma = dm.off_lambda = dm.off_eta = -1;
  if (dm.opt[0]) { dm.off_mu = T; T += D * K; }
  if (dm.opt[1]) { dm.off_sigma = T; T += K; }
  if (dm.opt[2]) { dm.off_lambda = T; T += D; }
  if (dm.opt[3]) { dm.off_eta = T; T += K; }
  dm.T = T;
  if (T > 0 && !a->theta) return set_err(ctx, VBMC_ERR_INVALID, "theta is null");
  if ((!dm.opt[0] && !a->vp_mu) || (!dm.opt[1] && !a->vp_sigma) || (!dm.opt[2] && !a->vp_lambda) || (!dm.opt[3] && !a->vp_w))
    return set_err(ctx, VBMC_ERR_INVALID, "a vp field is required for every group that is not optimised");
  const int compute_grad = P.compute_grad = a->compute_grad ? 1 : 0;
  const int compute_var = P.compute_var = a->compute_var;
  if (compute_var < 0 || compute_var > 2) return set_err(ctx, VBMC_ERR_INVALID, "compute_var must be 0, 1 or 2");
  // negelcbo_vbmc.m:21-24
  if (compute_grad && a->beta != 0.0 && compute_var != 2)
    return set_err(ctx, VBMC_ERR_INVALID, "negelcbo_vbmc:vargrad Computation of the gradient of ELBO with full variance not supported.");
  // gplogjoint.m:29-32 (negelcbo requests dvarG whenever compute_var && compute_grad)
  if (compute_grad && compute_var == 1)
    return set_err(ctx, VBMC_ERR_INVALID, "gplogjoint:FullVarianceGradient gradient of the log joint variance needs compute_var == 2");
  if (a->separate_K && compute_grad)  // negelcbo_vbmc.m:57-59
    return set_err(ctx, VBMC_ERR_INVALID, "Computing the gradient of variational parameters and requesting per-component results at the same time.");
  if (compute_var != 0 && !gp->hasL)
    return set_err(ctx, VBMC_ERR_INVALID, "compute_var != 0 needs gp.post(s).L: upload the GP with L");
  if (compute_var != 0 && VAR_FINAL_LDS(S, K, compute_grad ? T : 0) > 160 * 1024)   // k_var_final: five T-vectors in LDS
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "variance gradient with %d variational parameters (> 4000) not accelerated", T);
  if (compute_var != 0 && trsm_cw_for(dm.N) == 0)
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "variance path with N = %d > %d not accelerated", dm.N, trsm_max_n());
  P.beta = (std::isfinite(a->beta)) ? a->beta : 0.0;  // negelcbo_vbmc.m:15: non-finite beta -> 0
  // theta must be finite (the device exp does not propagate NaN)
  for (size_t i = 0; i < (size_t)T * R; ++i)
    if (!std::isfinite(a->theta[i])) return set_err(ctx, VBMC_ERR_INVALID, "theta contains a non-finite value at linear index %zu", i);

  int M = a->Ns;
  if (M < 0) return set_err(ctx, VBMC_ERR_INVALID, "Ns must be >= 0");
  if (a->restart_offset < 0 || a->restart_stride < 0) return set_err(ctx, VBMC_ERR_INVALID, "restart_offset / restart_stride must be >= 0");
  if (a->no_jacobian && compute_grad) {
    if (a->bnd_lb) return set_err(ctx, VBMC_ERR_INVALID, "no_jacobian (JACOBIAN_FLAG = 0) is a form of the stand-alone functions: no soft bounds");
  }
  if (a->dvarG && !(compute_grad && compute_var == 2)) return set_err(ctx, VBMC_ERR_INVALID, "dvarG needs compute_grad with compute_var = 2");
  if (a->dG_s && !compute_grad) return set_err(ctx, VBMC_ERR_INVALID, "dG_s (per-hyper-sample gradients) needs compute_grad");
  if (a->dvarG_s && !(compute_grad && compute_var == 2)) return set_err(ctx, VBMC_ERR_INVALID, "dvarG_s (per-hyper-sample variance gradients) needs compute_grad with compute_var = 2");
  if (a->plan_restarts < 0 || (a->plan_restarts > 0 && a->plan_restarts < R)) return set_err(ctx, VBMC_ERR_INVALID, "plan_restarts must be 0 or the size (>= R) of the batch this call is a share of");
  P.Rp = a->plan_restarts > 0 ? a->plan_restarts : R;
  P.no_jacobian = a->no_jacobian && compute_grad ? 1 : 0;
  P.r0 = a->restart_offset; P.rstride = a->restart_stride > 0 ? a->restart_stride : 1;
  M = ((M + 1) / 2) * 2;  // entmc_vbmc.m:45
  const int Mh = P.Mh = M / 2;
  P.mc = M > 0;
  if (P.mc && a->eps_mode != 0 && !a->eps) return set_err(ctx, VBMC_ERR_INVALID, "eps_mode %d needs eps", a->eps_mode);
  if (a->eps_mode < 0 || a->eps_mode > 2) return set_err(ctx, VBMC_ERR_INVALID, "eps_mode must be 0, 1 or 2");
  P.dt = pick_dt(D);
  if (P.dt < 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "D = %d not accelerated", D);
  // k_finalize keeps three T-vectors, the D x K soft-bound table and its reduction scratch in LDS; beyond 160 KB
  // (4 D K + 9 K > 19400, e.g. D = 32 with K > 141) they move to a global scratch block
  P.fin_big = (FIN_THREADS + 3 * (size_t)K + (size_t)D * K + 3 * (size_t)T + 8) * sizeof(double) > 160 * 1024;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  VpLayout VL{D, K};

  // ---- uploads: theta, fixed vp, delta^2, bounds (one pinned staging buffer, one H2D)
  const size_t n_theta = P.n_theta = (size_t)T * R;
  const size_t n_fix = (size_t)D * K + 2 * K + D;
  const size_t n_delta = D;
  const int next_mu = dm.opt[0] ? D * K : 0;
  const int has_sc = (dm.opt[1] || dm.opt[2]) ? 1 : 0;
  const int Text = next_mu + has_sc * D * K + (dm.opt[3] ? K : 0);
  P.has_bnd = a->bnd_lb != nullptr && a->bnd_ub != nullptr;
  const size_t n_bnd = P.has_bnd ? 3 * (size_t)Text : 0;     // lb | ub | 1 / ((ub - lb) TolCon)^2  (round 5: the reciprocals from the host)
  const size_t n_up = P.n_up = n_theta + n_fix + n_delta + n_bnd;
  P.out_n = (size_t)R * (OUT_HDR + 3 * T);
  // pinned block: staged inputs | result records | (separate_K) I_sk and J_sjk in the caller's layouts, when they are small enough
  // to be packed by a kernel (k_pack_sepk)
  P.n_sepk = (a->separate_K && (a->I_sk || a->J_sjk)) ? (size_t)S * K * R * (1 + (size_t)K) : 0;
  if (P.n_sepk * sizeof(double) > COPY_KERNEL_MAX_BYTES * 4) P.n_sepk = 0;
  { vbmc_status s_ = ensure_pin(ctx, (n_up + P.out_n + (size_t)S * K * R + P.n_sepk) * sizeof(double)); if (s_) return s_; }
  { vbmc_status s_ = ensure(ctx, ctx->theta, n_up * sizeof(double)); if (s_) return s_; }
  double* hp = (double*)ctx->pin;
  if (n_theta) memcpy(hp, a->theta, n_theta * sizeof(double));
  double* hfix = hp + n_theta;
  for (size_t i = 0; i < n_fix; ++i) hfix[i] = 1.0;
  if (a->vp_mu) memcpy(hfix, a->vp_mu, (size_t)D * K * sizeof(double));
  if (a->vp_sigma) memcpy(hfix + D * K, a->vp_sigma, K * sizeof(double));
  if (a->vp_lambda) memcpy(hfix + D * K + K, a->vp_lambda, D * sizeof(double));
  if (a->vp_w) memcpy(hfix + D * K + K + D, a->vp_w, K * sizeof(double));
  double* hdel = hfix + n_fix;
  for (int d = 0; d < D; ++d) hdel[d] = a->vp_delta ? a->vp_delta[d] * a->vp_delta[d] : 0.0;
  if (P.has_bnd) {
    memcpy(hdel + n_delta, a->bnd_lb, Text * sizeof(double));
    memcpy(hdel + n_delta + Text, a->bnd_ub, Text * sizeof(double));
    // softbndloss.m's ell = (UB - LB) TolCon enters as 1 / ell^2: (x - LB) / ell^2 and ((LB - x) / ell)^2 = (LB - x)^2 / ell^2 -- two fp64
    // divisions per bounded element in k_finalize_ws otherwise (44 division sequences, the longest of its tasks)
    double* hinv = hdel + n_delta + 2 * (size_t)Text;
    for (int i = 0; i < Text; ++i) { const double ell = (a->bnd_ub[i] - a->bnd_lb[i]) * a->TolCon; hinv[i] = 1.0 / (ell * ell); }
  }
  P.d_theta = (double*)ctx->theta.p;
  P.d_fix = P.d_theta + n_theta;
  P.d_delta2 = P.d_fix + n_fix;
  P.d_bnd = P.has_bnd ? P.d_delta2 + n_delta : nullptr;
  // (upload + unpacking in one launch and the reductions folded into the finalize kernel were built and measured in round 5: the host's
  // submit 23.6 -> 17.1 us at BASELINE configs[1], the step unchanged -- docs/history.md)
  if (copy_by_kernel(n_up * sizeof(double))) {
    hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)std::min<size_t>((n_up + 255) / 256, 1024)), dim3(256), 0, st, n_up, (const double*)hp, P.d_theta);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(P.d_theta, hp, n_up * sizeof(double), hipMemcpyHostToDevice, st));
  }
  P.TolCon = a->TolCon; P.WeightThreshold = a->WeightThreshold; P.WeightPenalty = a->WeightPenalty;
  P.cutoff = a->sparse_cutoff > 0.0 ? a->sparse_cutoff : 0.0;

  // ---- scratch
  { vbmc_status s_ = ensure(ctx, ctx->prep, (size_t)R * VL.stride() * sizeof(double)); if (s_) return s_; }
  { vbmc_status s_ = ensure(ctx, ctx->entp, (size_t)R * K * (D + ENTP_EXTRA) * sizeof(double)); if (s_) return s_; }
  const int LJS = 2 * D + 2;
  // (LJ_CO_SPLIT records per hyper-sample where the log joint may run as a role of the entropy launch: small grids only, see elbo_enqueue)
  // (also set when the caller asks for the numbers a SHARDED evaluation gives -- chunk_world: the sharded path adds the log-joint
  // records per hyper-sample, so the unsharded evaluation it is compared with bit for bit must too)
  P.lj_records = a->separate_K || a->G_s || a->varG_s || a->dG_s || compute_var != 0 || chunk_world > 0 || a->chunk_world > 1;
  const size_t ljrec = (size_t)R * S * K * LJS * (((long long)S * std::min(R, P.Rp) < ctx->num_cu / 2 && !P.lj_records) ? LJ_CO_SPLIT : 1);
  { vbmc_status s_ = ensure(ctx, ctx->ljpart, (ljrec + (size_t)R * K * LJS) * sizeof(double)); if (s_) return s_; }
  { vbmc_status s_ = ensure(ctx, ctx->out, P.out_n * sizeof(double)); if (s_) return s_; }
  {
    const size_t nbig = P.fin_big ? (size_t)R * (3 * (size_t)T + (size_t)D * K) : 0;
    const size_t ngam = (!(a->Ns > 0) && K > 128) ? (size_t)R * K * K : 0;      // entlb's K x K table beyond the LDS
    if (nbig + ngam) {
      vbmc_status s_ = ensure(ctx, ctx->bnd, (nbig + ngam) * sizeof(double)); if (s_) return s_;
      P.d_finbig = nbig ? (double*)ctx->bnd.p : nullptr;
      P.d_gamma = ngam ? (double*)ctx->bnd.p + nbig : nullptr;
    }
  }
  P.d_vpd = (double*)ctx->prep.p;
  P.d_entp = (double*)ctx->entp.p;
  P.d_lj = (double*)ctx->ljpart.p;
  P.d_ljbar = P.d_lj + ljrec;
  P.d_out = (double*)ctx->out.p;

  if (P.mc) {
    const char* force = getenv("VBMC_ENT_KERNEL");  // "valu" (A/B testing); default: the MFMA kernel when it fits
    P.use_mfma = mfma_entropy_fits(D, K, P.cutoff, &P.qs, &P.kt, &P.hv);
    if (force && !strcmp(force, "valu")) P.use_mfma = false;
    // small mixtures: the lane-per-sample kernel (entropy_lane.h).  VBMC_ENT_KERNEL=mfma keeps the matrix-core kernel there (A/B runs, tests)
    P.use_lane = lane_entropy_fits(D, K, P.cutoff) && !(force && (!strcmp(force, "valu") || !strcmp(force, "mfma"))) &&
                 ((long long)K * P.Rp * ((Mh + 63) / 64) >= ENT_LANE_MIN_TILES || (force && !strcmp(force, "lane")));
    if (P.use_lane) P.use_mfma = false;
    const int tile_sz = P.use_lane ? 64 : (P.use_mfma ? 16 : 32);          // base samples per tile
    const int ntile = (Mh + tile_sz - 1) / tile_sz;
    // chunks per (component, restart): minimise  ceil(waves / resident slots) * (setup + tiles per wave),
    // i.e. whole rounds of resident waves, with the per-wave setup worth ~1.5 tiles
    {
      const int cw = chunk_world > 0 ? chunk_world : (a->chunk_world > 1 ? a->chunk_world : 1);   // sharded over cw devices
      // resident waves: what the chosen instantiation really holds per compute unit (registers AND LDS; round 3 -- rounds 1-2 assumed two
      // waves per SIMD for every MFMA kernel, which under-filled the chip for the small kernels that hold three or four)
      int waves_per_cu = P.use_mfma ? 8 : 5;
      if (P.use_mfma) {
        EntArgs q{};
        q.D = D; q.K = K; q.cutoff = P.cutoff; q.lj.rows = lj_co_shape(ctx, P) ? 1 : 0;
        const int nb = entropy_mfma_occupancy(P.qs, P.kt, P.hv, compute_grad != 0, q);
        if (nb > 0) waves_per_cu = nb * (P.hv & 15);
        // eight-wave workgroups (K > 256): the parameter block, the parked exponents of eight waves and the PV exchange can exceed the
        // 160 KB of a compute unit at large D -- then no workgroup fits and the shape is refused below (the VALU kernel's LDS does not
        // hold K > 256 either)
        if (nb <= 0 && (P.hv & 15) == 8) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "K = %d, D = %d: the eight-wave entropy kernel needs more than the 160 KiB of LDS", K, D);
        // wide operands (D >= 15): the kernels that COULD hold more than eight waves (one k-tile) do not gain from shorter chunks -- their
        // per-wave set-up grows with D (D = 20, K = 8: 0.27 -> 0.37 ms with twelve assumed) -- while the ones that hold fewer (LDS: seven)
        // are where the correction pays (D = 28, K = 40: 1.36 -> 1.10 ms): profiles/r03_shape_sweep.md
        if (P.qs >= 5 && waves_per_cu > 8) waves_per_cu = 8;
      }
      if (P.use_lane) {
        const int nb = entropy_lane_occupancy(D, K, compute_grad != 0);
        waves_per_cu = ENT_LANE_WAVES_HOST * (nb > 0 ? nb : 2);
      }
      const long long slots = (long long)ctx->num_cu * waves_per_cu * cw;
      const long long kr = (long long)K * P.Rp * (P.use_mfma ? (P.hv & 15) : 1);   // waves per chunk index (hv + 16 TL: with a component tail); Rp: R, or the undivided batch's (plan_restarts)
      const double setup = P.use_lane ? 1.0 : 1.5;   // (lane kernel: a tile of 64 samples is ~1.5 us, the set-up about that)  measured: C = 7 (45 tiles per wave) beats C = 5 (63) by 1 % at the headline shape once the setup loads are batched
      double best = 1e300;
      int bestC = 1;
      for (int c = 1; c <= ntile; ++c) {
        const int tpc = (ntile + c - 1) / c;
        const int ceff = (ntile + tpc - 1) / tpc;
        const long long rounds = (kr * ceff + slots - 1) / slots;
        const double cost = (double)rounds * (setup + tpc);
        if (cost < best - 1e-9) { best = cost; bestC = ceff; }
        if (tpc == 1) break;
      }
      if (const char* fc = getenv("VBMC_ENT_CHUNKS")) {   // A/B testing of the chunk model
        const int c = atoi(fc);
        if (c >= 1 && c <= ntile) bestC = c;
      }
      if (getenv("VBMC_DEBUG_OCC")) fprintf(stderr, "chunks: D %d K %d R %d qs %d kt %d hv %d waves/CU %d slots %lld kr %lld ntile %d -> C %d\n", D, K, R, P.qs, P.kt, P.hv, waves_per_cu, slots, kr, ntile, bestC);
      P.tpc = (ntile + bestC - 1) / bestC;
      P.C = (ntile + P.tpc - 1) / P.tpc;
      // Two chunk classes where the launch carries the log-joint role (one or two restarts at Ns = 1e4: ~2000 entropy waves + 500-1000 role
      // waves on 2048 slots -- the role's waves went first and a quarter of the entropy waves entered a role late): the role's workgroups go
      // on to a chunk of the entropy that is a role's length (~4 tiles: 18 us) shorter, every wave of the launch is resident from the
      // start and all leave together.  Not where the chunking must be another launch's (sharded, plan_restarts) or is forced.
      if (P.use_mfma && (P.hv & 15) == 1 && cw == 1 && chunk_world == 0 && a->plan_restarts == 0 && !getenv("VBMC_ENT_CHUNKS") && compute_grad &&
          lj_co_shape(ctx, P)) {
        const int role_per_r = ((K + 3) / 4) * S * lj_co_nsplit(ctx, P);
        const long long waves0 = slots - (long long)role_per_r * R;
        const int TROLE = 4;                                   // a role in tiles (role + its later set-up against 4.5 us per tile of a wave that shares its SIMD)
        const int c1 = (int)(waves0 / ((long long)K * R)), c2max = role_per_r / K;
        if (c1 >= 1 && c2max >= 1 && ntile > TROLE * c1 + c1 + c2max) {
          const int tpc2 = (ntile - TROLE * c1 + c1 + c2max - 1) / (c1 + c2max);
          const int tpc1 = (ntile - c2max * tpc2 + c1 - 1) / c1;
          const int rest = ntile - c1 * tpc1;
          if (tpc2 >= 1 && tpc1 > tpc2 && rest > 0) {
            P.co_c1 = c1; P.co_tpc2 = tpc2; P.co_c2 = (rest + tpc2 - 1) / tpc2;
            P.tpc = tpc1; P.C = P.co_c1 + P.co_c2;
            if (getenv("VBMC_DEBUG_OCC")) fprintf(stderr, "two chunk classes: %d x %d tiles + %d x %d tiles (role workgroups per restart %d)\n", c1, tpc1, P.co_c2, tpc2, role_per_r);
          }
        }
      }
      // The walk (entropy_mfma.h): once the chunk grid would hand every wave slot two or more waves, ONE wave per slot walks its share of
      // all the (restart, component) pairs' tiles instead -- a set-up per (wave, pair) instead of per chunk.  The device-RNG gradient kernels of
      // single-wave workgroups at D <= 14, K <= 56 (the instantiations that take the loop without spilling); not where
      // the bits must be those of another launch shape (sharded evaluation, plan_restarts) or the launch carries the log-joint role.
      // Not for the passes of a pipeline either (vbmc_elbo_submit): there the short kernels of the NEXT pass take the slots the chunk grid's
      // waves free as they finish, and the step is the sum of its kernels' work with no gap at all (2.19 ms at the headline shape) -- a launch
      // whose waves all end together leaves them nothing until it is over (2.24); a blocking call or an optimiser iteration has no next pass
      // to interleave (2.29 -> 2.25 ms).  VBMC_ENT_WALK=0: the chunk grid (A/B runs, tests).
      const char* walk_env = getenv("VBMC_ENT_WALK");
      const bool walk_off = walk_env && !strcmp(walk_env, "0");
      const long long total = (long long)K * R * ntile;
      if (P.use_mfma && (P.hv & 15) == 1 && P.qs <= 4 && P.kt <= 3 && a->eps_mode == 0 && compute_grad && !(P.cutoff > 0.0) && cw == 1 && chunk_world == 0 && !pipelined && a->plan_restarts == 0 && !walk_off && !getenv("VBMC_ENT_CHUNKS") &&
          kr * P.C >= 2 * slots && total < (1LL << 31) && !lj_co_shape(ctx, P)) {
        P.walk_tpw = (int)((total + slots - 1) / slots);
        P.walk_nw = (int)((total + P.walk_tpw - 1) / P.walk_tpw);
        P.C = ent_walk_max_slots(ntile, P.walk_tpw);
        if (getenv("VBMC_DEBUG_OCC")) fprintf(stderr, "walk: %d waves x %d tiles, %d record slots per pair\n", P.walk_nw, P.walk_tpw, P.C);
      }
    }
    P.ncol = compute_grad ? (2 + 2 * D + K) : 1;
    { vbmc_status s_ = ensure(ctx, ctx->entpart, ((size_t)R * K * P.C * P.ncol + (size_t)R * K * P.ncol) * sizeof(double)); if (s_) return s_; }
    P.d_part = (double*)ctx->entpart.p;
    P.d_red = P.d_part + (size_t)R * K * P.C * P.ncol;
    const size_t eps_block = (size_t)D * Mh * K;
    if (a->eps_mode == 1) {
      const size_t n_eps = eps_block * (a->eps_shared ? 1 : (size_t)R);
      { vbmc_status s_ = ensure(ctx, ctx->eps, n_eps * sizeof(double)); if (s_) return s_; }
      HIP_TRY(ctx, hipMemcpyAsync(ctx->eps.p, a->eps, n_eps * sizeof(double), hipMemcpyHostToDevice, st));
      P.d_eps = (const double*)ctx->eps.p; P.eps_stride_r = a->eps_shared ? 0 : (long long)eps_block;
    } else if (a->eps_mode == 2) {
      P.d_eps = a->eps; P.eps_stride_r = a->eps_shared ? 0 : (long long)eps_block;
    }
    if (!P.use_mfma && !P.use_lane) {
      P.ent_lds = ((size_t)K * (P.dt + ENTP_EXTRA) + WAVE + (compute_grad ? (size_t)K * 65 : 0)) * sizeof(double);
      if (P.ent_lds > 160 * 1024) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "K = %d, D = %d needs %zu B of LDS (> 160 KiB)", K, D, P.ent_lds);
    }
  } else {
    const size_t ebs = 1 + (size_t)D * K + 2 * K + D;
    { vbmc_status s_ = ensure(ctx, ctx->entpart, (size_t)R * ebs * sizeof(double)); if (s_) return s_; }
    P.d_part = (double*)ctx->entpart.p;
  }
  if (compute_var != 0) {
    const int N = dm.N;
    P.var_stride = 2 + T;
    P.vgrad = compute_grad && compute_var == 2;
    for (int s = 0; s < S; ++s) P.any_nochol |= (gp->Lchol[s] == 0);
    const size_t nz = (size_t)R * S * K * N;
    { vbmc_status s_ = ensure(ctx, ctx->zbuf, nz * sizeof(double)); if (s_) return s_; }
    const size_t nJ = (size_t)R * S * K * K, nvg = (size_t)R * S * K * (2 * D + 1), nvo = (size_t)R * P.var_stride;
    // full variance without its gradient (eval_fullelcbo): V = inv(L') Z as a product with the explicit inverse (k_tri_gemm, out of
    // place into the X block) instead of the substitution
    {
      bool any_chol = false;
      for (int s = 0; s < S; ++s) any_chol |= (gp->Lchol[s] != 0);
      if (compute_var == 1 && any_chol) {
        bool have = false;
        vbmc_status s_ = ensure_tinv(ctx, gp, &have);
        if (s_) return s_;
        P.tri_gemm = have;
      }
    }
    P.needX = P.vgrad || P.any_nochol || P.tri_gemm;
    const size_t nvs = P.vgrad ? (size_t)R * S * 2 * (size_t)dm.T : 0;
    { vbmc_status s_ = ensure(ctx, ctx->varbuf, ((P.needX ? nz : 0) + nJ + nvg + nvo + nvs) * sizeof(double)); if (s_) return s_; }
    P.d_Z = (double*)ctx->zbuf.p;
    P.d_X = (double*)ctx->varbuf.p;
    P.d_J = P.d_X + (P.needX ? nz : 0);
    P.d_vg = P.d_J + nJ;
    P.d_var = P.d_vg + nvg;
    P.d_vs = P.d_var + nvo;
  }
  return VBMC_OK;
}

// Enqueue one evaluation pass on the context's stream: reads theta from P.d_theta, leaves the packed
// results [F G H varG varGss | dF dG dH] per restart in P.d_out.  No host synchronisation.
// pend / pend_iter: inside the optimiser loop, the Adam update of iteration pend_iter that k_prep applies before unpacking.
// cheap (no synchronisation) attribution of a failed launch to its kernel
#define LAUNCH_CHECK(ctx_, what)                                                                            \
  do {                                                                                                      \
    hipError_t le_ = hipGetLastError();                                                                     \
    if (le_ != hipSuccess) return set_err(ctx_, VBMC_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(le_)); \
  } while (0)

// Sharding of ONE evaluation over `world` ranks when there are fewer restarts than GPUs: rank g computes the
// expected-log-joint records of the hyper-samples [g perS, (g+1) perS) (misc/gplogjoint.m:98: the iterations over s are
// independent until the averaging at :399-413) and the entropy partial records of the sample chunks [g perC, (g+1) perC)
// -- with the SAME chunking the unsharded evaluation uses -- into one contiguous block; after the all-gather every rank
// scatters the blocks back into the unsharded layouts and runs the unsharded reductions + k_finalize, so the result is
// bit-identical to the 1-GPU evaluation by construction.
struct ShardSpec {
  int mode = 0;          // 0: unsharded; 1: begin (prep + own records -> send); 2: finish (prep + scatter + reductions + finalize)
  int rank = 0, world = 1;
  double* send = nullptr;            // mode 1: device block of shard_doubles(P, world)
  const double* gathered = nullptr;  // mode 2: device, world blocks in rank order
};
static inline int shard_per(int n, int world) { return (n + world - 1) / world; }
static inline size_t shard_lj_doubles(const ElboPlan& P, int world) {
  return (size_t)P.dm.R * shard_per(P.dm.S, world) * P.dm.K * (2 * P.dm.D + 2);
}
static inline size_t shard_doubles(const ElboPlan& P, int world) {
  return shard_lj_doubles(P, world) + (P.mc ? (size_t)P.dm.R * P.dm.K * shard_per(P.C, world) * P.ncol : 0);
}

// gathered blocks -> the unsharded record layouts lj[r][s][k][LJS] and part[r][j][c][ncol]
__global__ void __launch_bounds__(256) k_shard_scatter(int R, int S, int K, int LJS, int C, int ncol, int world, int perS, int perC,
                                                       size_t blk, size_t ljn, const double* __restrict__ g,
                                                       double* __restrict__ lj, double* __restrict__ part) {
  const size_t nlj = (size_t)R * S * K * LJS, npe = part ? (size_t)R * K * C * ncol : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlj + npe; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nlj) {
      const int col = (int)(i % LJS);
      size_t t = i / LJS;
      const int k = (int)(t % K); t /= K;
      const int sidx = (int)(t % S), r = (int)(t / S);
      const int gk = sidx / perS, sl = sidx - gk * perS;
      lj[i] = g[(size_t)gk * blk + (((size_t)r * perS + sl) * K + k) * LJS + col];
    } else {
      const size_t e = i - nlj;
      const int col = (int)(e % ncol);
      size_t t = e / ncol;
      const int c = (int)(t % C); t /= C;
      const int j = (int)(t % K), r = (int)(t / K);
      const int gk = c / perC, cl = c - gk * perC;
      part[e] = g[(size_t)gk * blk + ljn + (((size_t)r * K + j) * perC + cl) * ncol + col];
    }
  }
}

// pend / pend_iter: the Adam update of iteration pend_iter rides on this pass's k_prep.  fuse / fuse_iter: this pass's
// finalize kernel applies the Adam update of iteration fuse_iter and unpacks the new theta for the NEXT pass, which is then
// enqueued with skip_prep (the on-device optimiser loop: one launch per iteration less, vbmc_adam_batch).
static vbmc_status elbo_enqueue(vbmc_ctx* ctx, const vbmc_gp* gp, const ElboPlan& P, unsigned long long seed,
                                const AdamState* pend = nullptr, int pend_iter = 0, const ShardSpec* shp = nullptr,
                                const AdamState* fuse = nullptr, int fuse_iter = 0, bool skip_prep = false) {
  const ElboDims& dm = P.dm;
  const int D = dm.D, K = dm.K, R = dm.R, S = dm.S, T = dm.T;
  const int dt = P.dt;
  hipStream_t st = ctx->stream;
  const ShardSpec sh = shp ? *shp : ShardSpec{};
  // this rank's slice of the hyper-samples and of the entropy chunks (everything when unsharded)
  const int perS = shard_per(S, sh.world), s0 = sh.mode == 1 ? std::min(S, sh.rank * perS) : 0;
  const int ns = sh.mode == 1 ? std::min(S, s0 + perS) - s0 : S;
  const int perC = shard_per(P.C, sh.world), c0 = sh.mode == 1 ? std::min(P.C, sh.rank * perC) : 0;
  const int nc = sh.mode == 1 ? std::min(P.C, c0 + perC) - c0 : P.C;
  double* const lj_out = sh.mode == 1 ? sh.send : P.d_lj;
  const size_t prep_lds = ((size_t)D * K + 3 * K + D + 8) * sizeof(double);
  if (prep_lds > 64 * 1024)
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prep_lds));
  if (!skip_prep) {
    hipLaunchKernelGGL(k_prep, dim3(R), dim3(256), prep_lds, st, dm, P.d_theta, P.d_fix, P.d_vpd, P.d_entp, pend ? *pend : AdamState{},
                       pend ? pend_iter : 0, (const double*)P.d_out);
    LAUNCH_CHECK(ctx, "k_prep");
  }

  // ---- expected log joint: enqueued on `ls` -- the context's stream, or the auxiliary one beside the entropy kernel
  // ... and so does a pass on a slot stream: the pass on the other slot stream is what fills in around its entropy kernel, and a log
  // joint forked off there is the last to be let onto the chip (profiles/r04_experiments.md section 11)
  const bool lane_role_possible = P.use_lane && P.compute_grad && !P.lj_records && dm.N > 1 && lane_role_fits(D, K, dm.N, S);     // (decided below: co_lane)
  const bool fork = sh.mode == 0 && P.mc && (long long)S * R >= ctx->num_cu / 2 && !ctx->prof_alone && !lane_role_possible && ctx_aux(ctx);   // a single chain: the fork / join events cost more than they hide
  // value + gradient: moments on the matrix cores (k_logjoint_mfma); value only: the VALU kernel.  VBMC_LJ_KERNEL=valu / mfma forces one of them.
  const char* ljf = getenv("VBMC_LJ_KERNEL");
  // one workgroup per (hyper-sample, restart): needs enough of them to fill the chip, otherwise (a single chain) the finer-grained VALU
  // kernel has the lower latency
  const bool lj_force = ljf && !strcmp(ljf, "mfma");   // tests: exercise the MFMA kernel on small grids too
  // (round 4: from S R = one workgroup per compute unit on -- below, the finer-grained VALU kernel is the faster one: R = 8 at the headline
  // shape, 160 (hyper-sample, restart) workgroups: 56 us against 34 alone, the step 0.394 -> 0.360 ms; equal at R = 16, 142 against 174 us at R = 64)
  // (round 6) the lane-per-sample kernel of small mixtures carries the role at every batch width: a pass of that class is ONE chip-wide launch
  static const bool co_off_env = [] { const char* e = getenv("VBMC_LJ_CO"); return e && !strcmp(e, "0"); }();
  const bool co_lane = lane_role_possible && sh.mode == 0 && !lj_force && !co_off_env && !(ljf && !strcmp(ljf, "valu"));
  const bool co_shape = co_lane || (sh.mode == 0 && !fork && !lj_force && lj_co_shape(ctx, P));      // (the role takes precedence over the matrix-core kernel where its limits admit the batch)
  // (round 5) ... and enough WAVES in each: with K <= 16 a workgroup of the matrix-core kernel is a single wave walking the whole training
  // set, and the VALU kernel's four waves per cell group are faster until the batch is several chips wide (BASELINE configs[1], K = 10,
  // S R = 512: 28.3 us against 19.9)
  const bool lj_wide = K > 16 || (long long)S * P.Rp >= 4LL * ctx->num_cu;
  // (round 6: value-only passes too -- the sieve's 250 candidates -- through the kernel's GRAD = false form, once the batch is four chips wide)
  const bool lj_mfma = !co_shape && (P.compute_grad || (long long)S * P.Rp >= 4LL * ctx->num_cu) && K <= 256 && (lj_force || ((long long)S * P.Rp >= ctx->num_cu && lj_wide)) && !(ljf && !strcmp(ljf, "valu"));
  // Small grids (a single chain, a handful of restarts): the VALU log joint runs as a ROLE of the entropy launch (single-wave
  // workgroups ahead of the entropy ones, entropy_mfma.h CO = true) -- two dependent-chain-bound kernels side by side instead of
  // one after the other, one launch less.  Its records are per (hyper-sample, split of the training set); the reduction over
  // hyper-samples adds the splits.  VBMC_LJ_CO=0 keeps the separate launch (A/B runs, tests).
  const bool co = co_shape && !lj_mfma;
  auto enqueue_logjoint = [&](hipStream_t ls) -> vbmc_status {
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ls));
    if (!co) {
      // VALU kernel: four waves per cell (training set split, lower latency) while the grid is small, one wave per cell (no
      // replicated per-wave setup) once there are enough cells to fill the chip several times over
      const bool lj_split = dm.N > 64 && (long long)((K + 3) / 4) * S * P.Rp < 8LL * ctx->num_cu;
      DISPATCH_DT(dt, {
        constexpr int NCT = (2 * DT + 1 + 15) / 16;
        const int nw = (K + 15) / 16;
        const size_t mom_lds = LJ_MFMA_DYN_LDS(DT, nw);   // feature rows / moment exchange; large K x D falls back to the VALU kernel
        // the kernels reach hyper-sample s only through alpha + s N and gpc + s GPC_STRIDE and use dm.S as the record
        // stride: a slice [s0, s0 + ns) is the same launch with shifted pointers (the variant choice above uses the FULL S,
        // so a sharded evaluation runs the kernel the unsharded one would)
        ElboDims dml = dm;
        dml.S = sh.mode == 1 ? perS : S;
        const double* al = gp->alpha + (size_t)s0 * dm.N;
        const double* gc = gp->gpc + (size_t)s0 * GPC_STRIDE(D);
        if (ns <= 0) {
        } else if (lj_mfma && mom_lds + LJ_MFMA_STATIC_LDS <= 64 * 1024) {
          if (P.compute_grad)
            hipLaunchKernelGGL((k_logjoint_mfma<DT, true>), dim3(ns, R), dim3(WAVE * nw), mom_lds, ls, dml, P.d_vpd,
                               gp->X, gp->d_meanX, al, gc, P.d_delta2, lj_out);
          else
            hipLaunchKernelGGL((k_logjoint_mfma<DT, false>), dim3(ns, R), dim3(WAVE * nw), mom_lds, ls, dml, P.d_vpd,
                               gp->X, gp->d_meanX, al, gc, P.d_delta2, lj_out);
          LAUNCH_CHECK(ctx, "k_logjoint_mfma");
        } else {
          hipLaunchKernelGGL((k_logjoint<DT>), dim3((K + 3) / 4, ns, R), dim3(lj_split ? WAVE * LJ_MAXW : WAVE), 0, ls, dml, P.d_vpd, gp->X, al, gc,
                             P.d_delta2, lj_out, P.compute_grad);
          LAUNCH_CHECK(ctx, "k_logjoint");
        }
      });
    }
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ls));
    // log-joint partials summed over hyper-samples (in sample order), one record per (r, k); on the main stream of an MC
    // evaluation this shares a launch with the entropy reduction below
    if (sh.mode == 0 && (fork || !P.mc)) hipLaunchKernelGGL(k_lj_reduce, dim3(K, R), dim3(64), 0, ls, S, K, 2 * D + 2, P.d_lj, P.d_ljbar);
    LAUNCH_CHECK(ctx, "k_lj_reduce");
    return VBMC_OK;
  };
  if (fork) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
  } else if (sh.mode != 2) {
    vbmc_status s_ = enqueue_logjoint(st);
    if (s_) return s_;
  }
  if (sh.mode == 2) {   // the gathered records of all ranks -> the unsharded layouts
    const size_t blk = shard_doubles(P, sh.world), ljn = shard_lj_doubles(P, sh.world);
    const size_t tot = (size_t)R * S * K * (2 * D + 2) + (P.mc ? (size_t)R * K * P.C * P.ncol : 0);
    hipLaunchKernelGGL(k_shard_scatter, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 4096)), dim3(256), 0, st, R, S, K, 2 * D + 2,
                       P.C, P.ncol, sh.world, perS, perC, blk, ljn, sh.gathered, P.d_lj, P.mc ? P.d_part : nullptr);
    LAUNCH_CHECK(ctx, "k_shard_scatter");
    if (!P.mc) hipLaunchKernelGGL(k_lj_reduce, dim3(K, R), dim3(64), 0, st, S, K, 2 * D + 2, P.d_lj, P.d_ljbar);
  }

  // ---- entropy
  FinArgs fa{};
  fa.dm = dm;
  if (P.mc) {
    EntArgs ea{};
    ea.entp = P.d_entp; ea.vpd = P.d_vpd;
    // sharded: this rank's chunks [c0, c0 + nc) of the unsharded chunking, written to slots 0 .. nc-1 of a perC-slot record
    ea.part = sh.mode == 1 ? sh.send + shard_lj_doubles(P, sh.world) : P.d_part;
    ea.D = D; ea.K = K; ea.Mh = P.Mh; ea.C = sh.mode == 1 ? perC : P.C; ea.c0 = c0; ea.tiles_per_chunk = P.tpc; ea.ncol = P.ncol; ea.seed = seed;
    ea.eps = P.d_eps; ea.eps_stride_r = P.eps_stride_r; ea.cutoff = P.cutoff; ea.r0 = P.r0; ea.rstride = P.rstride;
    ea.prio = 1;   // progress-ordered wave priorities (entropy_mfma.h)
    ea.co_c1 = P.co_c1; ea.co_c2 = P.co_c2; ea.co_tpc2 = P.co_tpc2;
    const bool walk = P.walk_tpw > 0;
    if (walk && (sh.mode != 0 || co)) return set_err(ctx, VBMC_ERR_INVALID, "internal: the walking entropy launch was planned for a sharded / role-carrying pass");
    if (walk) { ea.walk_tpw = P.walk_tpw; ea.walk_R = R; }
    int co_rows = 0;
    if (co) {
      LjCo& lc = ea.lj;
      lc.nsplit = lj_co_nsplit(ctx, P);
      lc.nwg = ((K + 3) / 4) * S * lc.nsplit;
      if (P.use_lane) {      // the role is dealt over the entropy waves themselves (entropy_lane.h): no rows of its own
        lc.rows = co_rows = 0;
      } else {
        // (two chunk classes: the grid's x extent is the first class, the role's workgroups -- and as many more as the second class needs -- follow in rows)
        const int gx = P.co_c2 > 0 ? P.co_c1 : nc;
        lc.rows = co_rows = (std::max(lc.nwg, P.co_c2 > 0 ? K * P.co_c2 : 0) + gx - 1) / gx;
      }
      lc.want_grad = P.compute_grad;
      lc.dm = dm; lc.X = gp->X; lc.alpha = gp->alpha; lc.gpc = gp->gpc; lc.delta2 = P.d_delta2; lc.lj = P.d_lj;
    }
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], st));
    if (sh.mode == 2 || nc <= 0) {
    } else if (P.use_lane) {
      ea.nc_launch = nc;
      // workgroups per restart: four (component, chunk) items each -- and, where the launch carries the role, at least one wave per cell
      // group (a single chain has ten entropy waves and twenty-four cell groups: two groups behind each other on a wave were most of its launch)
      int gx = (K * nc + ENT_LANE_WAVES_HOST - 1) / ENT_LANE_WAVES_HOST;
      if (co) gx = std::max(gx, (ea.lj.nwg + ENT_LANE_WAVES_HOST - 1) / ENT_LANE_WAVES_HOST);
      bool ok = launch_entropy_lane(D, K, P.compute_grad != 0, dim3(gx, 1 + co_rows, R), st, ea);
      if (!ok) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "no lane entropy kernel for D = %d, K = %d", D, K);
    } else if (P.use_mfma) {
      bool ok = launch_entropy_mfma(P.qs, P.kt, P.hv, P.compute_grad != 0, walk ? dim3(P.walk_nw, 1, 1) : dim3((co && P.co_c2 > 0) ? P.co_c1 : nc, K + co_rows, R), st, ea);
      if (!ok) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "no MFMA entropy kernel for D = %d", D);
    } else {
      const size_t lds = P.ent_lds;
      DISPATCH_DT(dt, {
        HIP_TRY(ctx, set_entropy_lds<DT>(P.compute_grad != 0, lds));
        launch_entropy<DT>(P.compute_grad != 0, dim3(nc, K, R), lds, st, ea);
      });
    }
    LAUNCH_CHECK(ctx, "the entropy kernel");
    if (ctx->profiling) HIP_TRY(ctx, hipEventRecord(ctx->ev[3], st));
    if (sh.mode == 1) return VBMC_OK;   // the records are in the send block; the exchange and the rest follow in mode 2
    // chunk partials -> one record per (r, j), summed in chunk order
    if (fork)
      hipLaunchKernelGGL(k_ent_reduce, dim3(K, R), dim3(P.ncol >= 192 ? 256 : (P.ncol >= 96 ? 128 : 64)), 0, st, P.C, P.ncol,
                         P.d_part, P.d_red, P.walk_tpw, (P.Mh + 15) / 16);
    else
      hipLaunchKernelGGL(k_reduce_both, dim3(K, R, 2), dim3(P.ncol >= 192 ? 256 : (P.ncol >= 96 ? 128 : 64)), 0, st, P.C, P.ncol,
                         P.d_part, P.d_red, co ? S * ea.lj.nsplit : S, 2 * D + 2, P.d_lj, P.d_ljbar, P.walk_tpw, (P.Mh + 15) / 16);
    LAUNCH_CHECK(ctx, "k_ent_reduce / k_reduce_both");
    fa.entpart = P.d_red; fa.entlb = nullptr; fa.M = P.Mh; fa.C = 1; fa.ncol = P.ncol;
  } else {
    if (sh.mode == 1) return VBMC_OK;   // the deterministic bound is O(K^2 D): every rank evaluates it in mode 2
    size_t lds = ((P.d_gamma ? 0 : (size_t)K * K) + K + 256) * sizeof(double);
    if (lds > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_entlb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_entlb, dim3(R), dim3(256), lds, st, dm, P.d_vpd, P.d_part, P.compute_grad, P.d_gamma);
    LAUNCH_CHECK(ctx, "k_entlb");
    fa.entpart = nullptr; fa.entlb = P.d_part;
  }

  if (fork) {   // the entropy kernel is already queued on the main stream: the log joint fills in around it
    {
      vbmc_status s_ = enqueue_logjoint(ctx->aux);
      if (s_) return s_;
      HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    }
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
  }

  // ---- variance of the expected log joint (gplogjoint.m:273-337,375-413)
  if (P.compute_var != 0) {
    const int N = dm.N;
    DISPATCH_DT(dt, {
      hipLaunchKernelGGL((k_var_z<DT>), dim3(K, S, R), dim3(WAVE), 0, st, dm, P.d_vpd, gp->X, gp->gpc, P.d_delta2, P.d_Z);
      LAUNCH_CHECK(ctx, "k_var_z");
    });
    if (P.any_nochol) hipLaunchKernelGGL(k_symm, dim3(32, S, R), dim3(256), 0, st, N, K, S, gp->L, gp->d_lchol, P.d_Z, P.d_X);
    LAUNCH_CHECK(ctx, "k_symm");
    if (P.tri_gemm) {
      hipLaunchKernelGGL(k_tri_gemm, dim3(((N + 15) / 16) * ((K + 15) / 16), S, R), dim3(64), 0, st, N, K, S, gp->d_tinv, gp->d_lchol,
                         (const double*)P.d_Z, P.d_X);
      LAUNCH_CHECK(ctx, "k_tri_gemm");
    } else {
      HIP_TRY(ctx, trsm_fwd_launch(st, N, K, S, R, gp->L, gp->d_finv, gp->d_lchol, P.d_Z));
    }
    if (P.compute_var == 1)   // full K x K matrix: Gram products on the matrix cores, one workgroup per (s, r)
      hipLaunchKernelGGL(k_var_gram_mfma, dim3(S, R), dim3(1024), 0, st, dm, P.d_vpd, gp->gpc, P.d_delta2, gp->d_sn2, gp->d_lchol,
                         P.d_Z, P.d_X, P.d_J, P.tri_gemm ? 1 : 0);
    else
      hipLaunchKernelGGL(k_var_gram, dim3(16, S, R), dim3(256), 0, st, dm, P.d_vpd, gp->gpc, P.d_delta2, gp->d_sn2, gp->d_lchol,
                       P.d_Z, P.d_X, P.d_J, P.compute_var == 1 ? 1 : 0);
    LAUNCH_CHECK(ctx, "k_var_gram");
    if (P.vgrad) {
      HIP_TRY(ctx, trsm_bwd_launch(st, N, K, S, R, gp->L, gp->d_finv, gp->d_lchol, P.d_Z, P.d_X));
      DISPATCH_DT(dt, {
        hipLaunchKernelGGL((k_vargrad<DT>), dim3(K, S, R), dim3(WAVE), 0, st, dm, P.d_vpd, gp->X, gp->gpc, P.d_delta2,
                           gp->d_sn2, gp->d_lchol, P.d_X, P.d_vg);
        LAUNCH_CHECK(ctx, "k_vargrad");
      });
    }
    VarFinArgs va{};
    va.dm = dm; va.vpd = P.d_vpd; va.gpc = gp->gpc; va.delta2 = P.d_delta2; va.lj = P.d_lj; va.J = P.d_J;
    va.vg = P.vgrad ? P.d_vg : nullptr; va.compute_var = P.compute_var; va.want_grad = P.compute_grad; va.stride = P.var_stride;
    va.out = P.d_var;
    va.no_jacobian = P.no_jacobian; va.dvs_out = P.d_dvs; va.vs = P.d_vs;
    if (P.vgrad) {
      const size_t slds = VAR_SAMPLE_LDS(K, T);
      if (slds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_var_sample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
      hipLaunchKernelGGL(k_var_sample, dim3(S, R), dim3(VARSMP_THREADS), slds, st, va);
      LAUNCH_CHECK(ctx, "k_var_sample");
    }
    const size_t vlds = VAR_FINAL_LDS(S, K, P.compute_grad ? T : 0);
    if (vlds > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_var_final, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlds));
    hipLaunchKernelGGL(k_var_final, dim3(R), dim3(VARFIN_THREADS), vlds, st, va);
    LAUNCH_CHECK(ctx, "k_var_final");
  }

  // ---- finalize
  fa.vpd = P.d_vpd; fa.theta = P.d_theta; fa.ljbar = P.d_ljbar; fa.var = P.d_var; fa.var_stride = P.d_var ? P.var_stride : 0;
  fa.bnd = P.d_bnd; fa.has_bnd = P.has_bnd ? 1 : 0;
  fa.TolCon = P.TolCon; fa.WeightThreshold = P.WeightThreshold; fa.WeightPenalty = P.WeightPenalty;
  fa.beta = P.beta; fa.want_grad = P.compute_grad; fa.out = P.out_direct ? P.out_direct : P.d_out; fa.no_jacobian = P.no_jacobian;
  fa.invS = 1.0 / S; fa.invM = fa.M > 0 ? 1.0 / (2.0 * fa.M) : 0.0;
  {
    size_t lds = (FIN_THREADS + 3 * (size_t)K + (P.fin_big ? 0 : (size_t)D * K + 3 * (size_t)T) + 8) * sizeof(double);
    fa.big = P.d_finbig;
    const int next_mu = dm.opt[0] ? D * K : 0;
    const int Text = next_mu + ((dm.opt[1] || dm.opt[2]) ? D * K : 0) + (dm.opt[3] ? K : 0);
    const size_t stage_rec = ((size_t)K * (2 * D + 2) + (fa.entpart ? (size_t)K * fa.C * fa.ncol : 0)) * sizeof(double);
    const size_t stage_vp = ((size_t)VpLayout{D, K}.stride() + (P.has_bnd ? 3 * (size_t)Text : 0)) * sizeof(double);
    fa.stage = 0;
    if (lds + stage_vp <= 96 * 1024) { fa.stage |= 2; lds += stage_vp; }
    if (lds + stage_rec <= 96 * 1024) { fa.stage |= 1; lds += stage_rec; }
    // k_finalize_ws: the wave-specialised form (one barrier; round 3 -- the sequential kernel with its ~18 barriers is in the history);
    // it carries the fused Adam update + unpacking of the next iteration
    if (fuse) {
      fa.next_iter = fuse_iter; fa.next_A = *fuse; fa.next_theta = P.d_theta; fa.next_vpfix = P.d_fix; fa.next_vpd = P.d_vpd; fa.next_entp = P.d_entp;
    }
    if (lds > 64 * 1024) {
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_finalize_ws<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_finalize_ws<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (fa.stage == 3 && !fa.big) hipLaunchKernelGGL(k_finalize_ws<true>, dim3(R), dim3(FIN_THREADS), lds, st, fa);
    else hipLaunchKernelGGL(k_finalize_ws<false>, dim3(R), dim3(FIN_THREADS), lds, st, fa);
    LAUNCH_CHECK(ctx, "k_finalize");
  }
  HIP_TRY(ctx, hipGetLastError());
  return VBMC_OK;
}

// The surrogate behind an entropy-only call: N = 1, alpha = 0, meanfun 0  =>  G = 0 and dG = 0 exactly, so that
// H, dH (and F = -H + penalties) are entmc_vbmc / entlb_vbmc on their own (ent/entmc_vbmc.m:1, ent/entlb_vbmc.m:1).
static vbmc_status null_gp_for(vbmc_ctx* ctx, int D, const vbmc_gp** out) {
  if (D <= 0 || D > 32) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "D = %d not accelerated", D);
  if (!ctx->null_gp[D]) {
    std::vector<double> X(D, 0.0), hyp(D + 2, 0.0);
    const double alpha = 0.0, sW1 = 1.0;
    const uint8_t lchol = 1;
    vbmc_status s_ = gp_upload_impl(ctx, 1, D, 1, D + 2, D + 1, 1, 0, X.data(), hyp.data(), &alpha, nullptr, nullptr, nullptr, &sW1,
                                    &lchol, &ctx->null_gp[D]);
    if (s_) return s_;
  }
  *out = ctx->null_gp[D];
  return VBMC_OK;
}

// The packed read-back of an enqueued pass into pinned memory (no synchronisation) ...
static vbmc_status elbo_enqueue_readback(vbmc_ctx* ctx, const ElboPlan& P, const vbmc_elbo_args* a, double* hout) {
  const int R = P.dm.R, T = P.dm.T;
  // record per restart: [F G H varG varGss | dF (T) | dG (T) | dH (T)]; without dG / dH only the leading part moves
  const size_t OSr = OUT_HDR + 3 * (size_t)T;
  const bool lead = P.compute_grad && !a->dG && !a->dH && R > 1;
  const size_t width = OUT_HDR + (size_t)T;
  if (copy_by_kernel((lead ? (size_t)R * width : P.out_n) * sizeof(double))) {
    if (lead)
      hipLaunchKernelGGL(k_copy_rows_f64, dim3((unsigned)((width + 255) / 256), (unsigned)std::min(R, 1024)), dim3(256), 0, ctx->stream, R, OSr,
                         width, (const double*)P.d_out, hout);
    else
      hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)std::min<size_t>((P.out_n + 255) / 256, 1024)), dim3(256), 0, ctx->stream, P.out_n,
                         (const double*)P.d_out, hout);
    HIP_TRY(ctx, hipGetLastError());
  } else if (lead) {
    HIP_TRY(ctx, hipMemcpy2DAsync(hout, OSr * sizeof(double), P.d_out, OSr * sizeof(double), width * sizeof(double), R,
                                  hipMemcpyDeviceToHost, ctx->stream));
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(hout, P.d_out, P.out_n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  return VBMC_OK;
}

// ... and, once it has landed, its unpacking into the caller's arrays (restart q of the pass is column col0 + q * cstride there:
// 0, 1 for a batch of its own; g, G for the share of rank g of a batch dealt over G ranks)
static void elbo_unpack(const ElboPlan& P, const vbmc_elbo_args* a, const double* hout, int col0 = 0, int cstride = 1) {
  const int R = P.dm.R, T = P.dm.T;
  const size_t OS = OUT_HDR + 3 * (size_t)T;
  for (int q = 0; q < R; ++q) {
    const double* o = hout + (size_t)q * OS;
    const size_t r = (size_t)col0 + (size_t)q * cstride;
    if (a->F) a->F[r] = o[0];
    if (a->G) a->G[r] = o[1];
    if (a->H) a->H[r] = o[2];
    if (a->varG) a->varG[r] = o[3];
    if (a->varGss) a->varGss[r] = o[4];
    if (P.compute_grad) {
      if (a->dF) memcpy(a->dF + r * T, o + OUT_HDR, T * sizeof(double));
      if (a->dG) memcpy(a->dG + r * T, o + OUT_HDR + T, T * sizeof(double));
      if (a->dH) memcpy(a->dH + r * T, o + OUT_HDR + 2 * T, T * sizeof(double));
    }
  }
}

// One packed D2H of the results of an enqueued pass (+ I_sk / J_sjk / per-sample outputs when requested); synchronises.
static vbmc_status elbo_read_results(vbmc_ctx* ctx, const ElboPlan& P, const vbmc_elbo_args* a) {
  const ElboDims& dm = P.dm;
  const int K = dm.K, R = dm.R, S = dm.S, D = dm.D;
  const int LJS = 2 * D + 2;
  hipStream_t st = ctx->stream;

  // ---- results: one packed D2H (+ I_sk / J_sjk when requested)
  double* hout = (double*)ctx->pin + P.n_up;
  { vbmc_status s_ = elbo_enqueue_readback(ctx, P, a, hout); if (s_) return s_; }
  std::vector<double> ljh;
  std::vector<double> Jh;
  double* hI = hout + P.out_n + (size_t)S * K * R;   // (behind the slack the block has always carried)
  double* hJ = hI + (size_t)S * K * R;
  const bool packed = P.n_sepk != 0 && copy_by_kernel(0);
  const bool wantJ = a->separate_K && a->J_sjk && P.d_J;
  if (packed) {
    const size_t tot = (size_t)R * S * K * (1 + (wantJ ? (size_t)K : 0));
    hipLaunchKernelGGL(k_pack_sepk, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 2048)), dim3(256), 0, st, R, S, K, LJS,
                       P.compute_var == 2 ? 1 : 0, (const double*)P.d_lj, wantJ ? (const double*)P.d_J : nullptr, hI, hJ);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    if (a->separate_K && a->I_sk) {
      ljh.resize((size_t)R * S * K * LJS);
      HIP_TRY(ctx, hipMemcpyAsync(ljh.data(), P.d_lj, ljh.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (wantJ) {
      Jh.resize((size_t)R * S * K * K);
      HIP_TRY(ctx, hipMemcpyAsync(Jh.data(), P.d_J, Jh.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    }
  }
  std::vector<double> psh;
  double* d_ps = nullptr;
  if (a->G_s || a->varG_s) {   // gplogjoint's avg_flag = 0 outputs
    if (a->varG_s && !P.compute_var) return set_err(ctx, VBMC_ERR_INVALID, "varG_s needs compute_var != 0");
    HIP_TRY(ctx, pool_get(ctx, 2 * (size_t)S * R * sizeof(double), (void**)&d_ps));
    hipLaunchKernelGGL(k_per_sample, dim3((S + 63) / 64, R), dim3(64), 0, st, dm, P.d_vpd, P.d_lj, P.d_J, P.compute_var, d_ps,
                       a->varG_s ? d_ps + (size_t)S * R : nullptr);
    psh.resize(2 * (size_t)S * R);
    hipError_t e_ = hipMemcpyAsync(psh.data(), d_ps, psh.size() * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e_ != hipSuccess) { pool_put(ctx, d_ps); return set_err(ctx, VBMC_ERR_HIP, "per-sample read-back: %s", hipGetErrorString(e_)); }
  }
  std::vector<double> dgsh;
  double* d_dgs = nullptr;
  if (a->dG_s) {               // gplogjoint's dF with avg_flag = 0: T x S per restart
    const size_t n = (size_t)dm.T * S * R;
    hipError_t e0 = pool_get(ctx, n * sizeof(double), (void**)&d_dgs);
    if (e0 != hipSuccess) { if (d_ps) pool_put(ctx, d_ps); return set_err(ctx, VBMC_ERR_HIP, "per-sample gradient block: %s", hipGetErrorString(e0)); }
    hipLaunchKernelGGL(k_per_sample_grad, dim3(S, R), dim3(256), 0, st, dm, P.d_vpd, P.d_lj, P.no_jacobian, d_dgs);
    dgsh.resize(n);
    hipError_t e_ = hipMemcpyAsync(dgsh.data(), d_dgs, n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e_ != hipSuccess) { pool_put(ctx, d_dgs); if (d_ps) pool_put(ctx, d_ps); return set_err(ctx, VBMC_ERR_HIP, "per-sample gradient read-back: %s", hipGetErrorString(e_)); }
  }
  {
    hipError_t e_ = hipStreamSynchronize(st);
    if (d_dgs) pool_put(ctx, d_dgs);
    if (d_ps) pool_put(ctx, d_ps);
    if (e_ != hipSuccess) { (void)hipGetLastError(); return set_err(ctx, VBMC_ERR_HIP, "vbmc_elbo_batch: %s", hipGetErrorString(e_)); }
  }
  if (a->dG_s) memcpy(a->dG_s, dgsh.data(), dgsh.size() * sizeof(double));
  if (a->G_s) memcpy(a->G_s, psh.data(), (size_t)S * R * sizeof(double));
  if (a->varG_s) memcpy(a->varG_s, psh.data() + (size_t)S * R, (size_t)S * R * sizeof(double));
  if (ctx->profiling) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) ctx->last_lj_ms = ms;
    if (P.mc && hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]) == hipSuccess) ctx->last_ent_ms = ms;
  }
  elbo_unpack(P, a, hout);
  if (a->dvarG && P.vgrad && P.d_var) {   // gplogjoint's dvarF: behind the two variance scalars of each restart's record
    HIP_TRY(ctx, hipMemcpy2DAsync(a->dvarG, (size_t)dm.T * sizeof(double), P.d_var + 2, (size_t)P.var_stride * sizeof(double),
                                  (size_t)dm.T * sizeof(double), R, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  if (packed) {
    if (a->I_sk) memcpy(a->I_sk, hI, (size_t)R * S * K * sizeof(double));
    if (wantJ) memcpy(a->J_sjk, hJ, (size_t)R * S * K * K * sizeof(double));
  } else if (a->separate_K && a->I_sk) {
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < K; ++k)
        for (int s = 0; s < S; ++s) a->I_sk[s + (size_t)S * (k + (size_t)K * r)] = ljh[(((size_t)r * S + s) * K + k) * LJS];
  }
  if (!Jh.empty()) {
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < K; ++k)
        for (int j = 0; j < K; ++j)
          for (int s = 0; s < S; ++s)
            a->J_sjk[s + (size_t)S * (j + (size_t)K * (k + (size_t)K * r))] =   // diagonal approximation: only J_kk is set (gplogjoint.m:283)
                (P.compute_var == 2 && j != k) ? 0.0 : Jh[(((size_t)r * S + s) * K + k) * K + j];
  }
  return VBMC_OK;
}


extern "C" vbmc_status vbmc_elbo_batch(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!gp && a) {  // entropy only
    if (a->compute_var != 0 || a->separate_K)
      return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_batch: an entropy-only call (gp == NULL) has no variance / per-component outputs");
    vbmc_status s_ = null_gp_for(ctx, a->D, &gp);
    if (s_) return s_;
  }
  ElboPlan P;
  { vbmc_status s_ = elbo_plan(ctx, gp, a, P); if (s_) return s_; }
  const size_t ndvs = a->dvarG_s ? (size_t)P.dm.T * P.dm.S * P.dm.R : 0;    // gplogjoint's dvarF with avg_flag = 0: T x S per restart
  if (ndvs) HIP_TRY(ctx, pool_get(ctx, ndvs * sizeof(double), (void**)&P.d_dvs));
  vbmc_status st = elbo_enqueue(ctx, gp, P, a->seed);
  if (!st) st = elbo_read_results(ctx, P, a);      // (waits for the stream)
  if (P.d_dvs) {
    if (!st) {
      hipError_t e_ = hipMemcpy(a->dvarG_s, P.d_dvs, ndvs * sizeof(double), hipMemcpyDeviceToHost);
      if (e_ != hipSuccess) { (void)hipGetLastError(); st = set_err(ctx, VBMC_ERR_HIP, "per-hyper-sample variance gradient read-back: %s", hipGetErrorString(e_)); }
    } else {
      (void)hipStreamSynchronize(ctx->stream);
    }
    pool_put(ctx, P.d_dvs);
  }
  return st;
}

// ---- pipelined form for streams of independent batches (include/vbmc_hip.h): submit enqueues and returns, collect waits
struct SlotPlan {
  ElboPlan P;
  double* hout = nullptr;
};
static void elbo_plan_free(void* plan) { delete (SlotPlan*)plan; }

// stage + enqueue + read-back of one batch into `slot` of the context, WITHOUT the event that marks its end (the caller may append work
// of its own to the stream first: the exchange of vbmc_elbo_multi_submit)
static vbmc_status elbo_submit_core(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, int slot, const char* who,
                                    bool allow_direct = false) {
  if (!a) return set_err(ctx, VBMC_ERR_INVALID, "%s: null args", who);
  if (slot < 0 || slot > 1) return set_err(ctx, VBMC_ERR_INVALID, "%s: slot must be 0 or 1", who);
  if (ctx->slot_busy[slot]) return set_err(ctx, VBMC_ERR_INVALID, "%s: slot %d holds an uncollected pass", who, slot);
  if (a->separate_K || a->I_sk || a->J_sjk || a->G_s || a->varG_s || a->dG_s || a->dvarG_s)
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "%s: per-component / per-hyper-sample outputs only through vbmc_elbo_batch", who);
  if (a->eps_mode == 1)
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "%s: host-resident draws (eps_mode 1) only through vbmc_elbo_batch", who);
  if (!gp) {
    if (a->compute_var != 0) return set_err(ctx, VBMC_ERR_INVALID, "%s: an entropy-only call (gp == NULL) has no variance", who);
    vbmc_status s_ = null_gp_for(ctx, a->D, &gp);
    if (s_) return s_;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->slot_ev[slot]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->slot_ev[slot], hipEventDisableTiming));
  if (!ctx->slot_plan[slot]) ctx->slot_plan[slot] = new SlotPlan();
  SlotPlan* sp = (SlotPlan*)ctx->slot_plan[slot];
  sp->P = ElboPlan{};
  // the slot's own pinned block stands in for the context's while the inputs are staged and the copies are enqueued
  std::swap(ctx->pin, ctx->slot_pin[slot]);
  std::swap(ctx->pin_cap, ctx->slot_pin_cap[slot]);
  vbmc_status s_ = elbo_plan(ctx, gp, a, sp->P, 0, true);
  // Result blocks of at most 256 KB are written by the finalize kernel straight into the pinned block (no read-back launch: BASELINE
  // configs[1] 52.1 -> 50.0 us per step, nothing elsewhere), only where nothing on the device reads the records afterwards (not under a
  // communicator: k_comm_pick does).  The mirror image -- the staging block read
  // by the pass's first kernel instead of a copy launch -- was built and measured too: no gain (49.8 us), removed.
  const size_t direct_kb = 256;
  if (!s_) {
    sp->hout = (double*)ctx->pin + sp->P.n_up;
    if (allow_direct && direct_kb && sp->P.compute_var == 0 && sp->P.out_n * sizeof(double) <= direct_kb * 1024) sp->P.out_direct = sp->hout;
  }
  if (!s_) s_ = elbo_enqueue(ctx, gp, sp->P, a->seed);
  if (!s_ && !sp->P.out_direct) s_ = elbo_enqueue_readback(ctx, sp->P, a, sp->hout);
  std::swap(ctx->pin, ctx->slot_pin[slot]);
  std::swap(ctx->pin_cap, ctx->slot_pin_cap[slot]);
  if (s_) { (void)hipStreamSynchronize(ctx->stream); return s_; }   // nothing of a failed submit stays in flight
  return VBMC_OK;
}
// ... and the event: the slot is busy from here on
static vbmc_status elbo_submit_mark(vbmc_ctx* ctx, int slot, const char* who) {
  ctx->slot_busy[slot] = true;
  hipError_t e_ = hipEventRecord(ctx->slot_ev[slot], ctx->stream);
  if (e_ != hipSuccess) {   // the pass is enqueued but cannot be waited for through the event: drain it and give the slot back
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);
    ctx->slot_busy[slot] = false;
    return set_err(ctx, VBMC_ERR_HIP, "%s: hipEventRecord: %s", who, hipGetErrorString(e_));
  }
  return VBMC_OK;
}

// ---- where a new stream lands.  The runtime gives a stream a hardware queue, and the k-th hardware queue a process creates sits on
// dispatch pipe k mod 4 (tools/stream_pipes.hip, profiles/r04_experiments.md section 11): a kernel of a stream on the SAME pipe as a
// stream whose large grid is being handed out waits until the last workgroup of that grid is out -- for the entropy kernel ten
// elevenths of its duration -- and one on the same queue until it has finished.  Which queues exist when the slot streams are created
// is the process's history (torch, RCCL, other contexts), so the placement is measured: a candidate stream is kept if a one-wave
// kernel on it finishes early in the life of a long, low-occupancy kernel on each of the streams it has to run beside.
__global__ void k_place_probe(long long ticks) {
  extern __shared__ char probe_lds[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (ticks < 0) probe_lds[threadIdx.x] = 0;
}
// when the small kernel on `b` finished, as a fraction of the long kernel on `a` (both streams idle before); 2 on any error
static double place_ratio(int device, int num_cu, hipStream_t a, hipStream_t b) {
  hipEvent_t e[3] = {nullptr, nullptr, nullptr};
  double best = 2.0;
  bool ok = true;
  for (auto& ev : e) ok = ok && hipEventCreate(&ev) == hipSuccess;
  int khz = 100000;
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device);
  const long long ticks = (long long)(15e-6 * khz * 1e3);     // 15 us per workgroup, three rounds of four one-wave workgroups per compute unit
  for (int rep = 0; ok && rep < 2; ++rep) {
    ok = ok && hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess;
    ok = ok && hipEventRecord(e[0], a) == hipSuccess;
    hipLaunchKernelGGL(k_place_probe, dim3(num_cu * 12), dim3(64), 36 * 1024, a, ticks);
    hipLaunchKernelGGL(k_place_probe, dim3(1), dim3(64), 0, b, ticks / 50);
    ok = ok && hipEventRecord(e[1], b) == hipSuccess && hipEventRecord(e[2], a) == hipSuccess;
    ok = ok && hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess;
    float ts = 0.f, tl = 0.f;
    ok = ok && hipEventElapsedTime(&ts, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&tl, e[0], e[2]) == hipSuccess;
    if (ok && tl > 0.f) best = std::min(best, (double)ts / tl);
  }
  for (auto& ev : e) if (ev) (void)hipEventDestroy(ev);
  if (!ok) (void)hipGetLastError();
  return ok ? best : 2.0;
}
// a new stream that dispatches beside every stream of `beside` (up to six candidates; the first one if none does)
static hipStream_t stream_beside(vbmc_ctx* ctx, const hipStream_t* beside, int nb, int priority = 0) {
  hipStream_t cand[6];
  int nc = 0;
  hipStream_t pick = nullptr;
  while (nc < 6 && !pick) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority) != hipSuccess) { (void)hipGetLastError(); break; }
    cand[nc++] = s;
    bool ok = true;
    for (int i = 0; i < nb && ok; ++i) ok = place_ratio(ctx->device, ctx->num_cu, beside[i], s) < 0.35;
    if (ok) pick = s;
  }
  if (!pick && nc) pick = cand[0];
  for (int i = 0; i < nc; ++i)
    if (cand[i] != pick) (void)hipStreamDestroy(cand[i]);
  return pick;
}

// The context (and its slot) a pass submitted into public slot `slot` runs on: child context slot & 1, its slot slot >> 1 (see
// vbmc_ctx.slot_sub) for the optimiser-loop / sieve form of the call; this context itself, slots 0 and 1 only, for the variance forms
// (their lazily built per-surrogate blocks live in the parent's pool) and under VBMC_SLOT_STREAMS=0 (A/B runs).  The child's stream is
// ordered after everything enqueued on the parent's stream so far (a surrogate uploaded, draws produced there).
static vbmc_status slot_ctx(vbmc_ctx* ctx, const vbmc_elbo_args* a, int slot, vbmc_ctx** out, int* inner) {
  static const bool off = [] { const char* e = getenv("VBMC_SLOT_STREAMS"); return e && !strcmp(e, "0"); }();
  *out = ctx; *inner = slot;
  if (slot < 0 || slot >= VBMC_SLOTS) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_submit: slot must be 0 .. %d", VBMC_SLOTS - 1);
  if (off || ctx->is_sub || !a || a->compute_var != 0) {
    if (slot > 1) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_submit: slots 2 and 3 exist for passes without a variance term (and not under VBMC_SLOT_STREAMS=0)");
    return VBMC_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int ch = slot & 1;
  if (!ctx->slot_sub[0] || !ctx->slot_sub[1]) {     // both children at once, before anything runs on either: the second beside the first
    for (int c2 = 0; c2 < 2; ++c2) {
      if (ctx->slot_sub[c2]) continue;
      hipStream_t other = ctx->slot_sub[1 - c2] ? ctx->slot_sub[1 - c2]->stream : nullptr;
      hipStream_t s = stream_beside(ctx, &other, other ? 1 : 0);
      if (!s) return set_err(ctx, VBMC_ERR_HIP, "vbmc_elbo_submit: no stream for slot %d", slot);
      vbmc_ctx* sc = nullptr;
      vbmc_status st = ctx_create_impl(ctx->device, s, &sc, false);     // no fork on a slot stream (elbo_enqueue): one stream per child
      if (st != VBMC_OK) { (void)hipStreamDestroy(s); return set_err(ctx, st, "vbmc_elbo_submit: no stream for slot %d", slot); }
      sc->own_stream = true;
      sc->is_sub = true;
      ctx->slot_sub[c2] = sc;
    }
  }
  if (!ctx->slot_xev[slot]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->slot_xev[slot], hipEventDisableTiming));
  vbmc_ctx* sc = ctx->slot_sub[ch];
  sc->profiling = false;
  // ... when there is anything: an event recorded on an idle stream and waited for on another is still a device-side dependency between
  // two queues, 15-30 us per step where a step is that short (tools/archive/r4_host_cost.py: one restart at VBMC's own sample count 58 -> 27 us
  // per step, BASELINE configs[1] 89 -> 52).
  if (hipStreamQuery(ctx->stream) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(ctx, hipEventRecord(ctx->slot_xev[slot], ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(sc->stream, ctx->slot_xev[slot], 0));
  }
  *out = sc; *inner = slot >> 1;
  return VBMC_OK;
}
// an error of the child context reported through the parent the caller holds
static vbmc_status slot_err(vbmc_ctx* ctx, vbmc_ctx* sc, vbmc_status st) {
  if (st != VBMC_OK && sc != ctx) ctx->err = sc->err;
  return st;
}
static bool slot_in_flight(const vbmc_ctx* ctx, int slot) {
  return slot >= 0 && slot < VBMC_SLOTS && ctx->slot_where[slot] && ctx->slot_where[slot]->slot_busy[ctx->slot_inner[slot]] &&
         (ctx->slot_where[slot] != ctx || ctx->slot_inner[slot] == slot);
}

extern "C" vbmc_status vbmc_elbo_submit(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, int slot) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (slot_in_flight(ctx, slot)) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_submit: slot %d holds an uncollected pass", slot);
  vbmc_ctx* sc = ctx;
  int inner = slot;
  { vbmc_status s_ = slot_ctx(ctx, a, slot, &sc, &inner); if (s_) return s_; }
  { vbmc_status s_ = elbo_submit_core(sc, gp, a, inner, "vbmc_elbo_submit", true); if (s_) return slot_err(ctx, sc, s_); }
  ctx->slot_where[slot] = sc; ctx->slot_inner[slot] = inner;
  return slot_err(ctx, sc, elbo_submit_mark(sc, inner, "vbmc_elbo_submit"));
}

// waits for the pass submitted in `slot`; *sp_out: its plan and the pinned block its results landed in
static vbmc_status elbo_collect_core(vbmc_ctx* ctx, const vbmc_elbo_args* a, int slot, const SlotPlan** sp_out, const char* who) {
  if (!a || slot < 0 || slot > 1) return set_err(ctx, VBMC_ERR_INVALID, "%s: null args / slot not 0 or 1", who);
  if (!ctx->slot_busy[slot]) return set_err(ctx, VBMC_ERR_INVALID, "%s: nothing submitted in slot %d", who, slot);
  const SlotPlan* sp = (const SlotPlan*)ctx->slot_plan[slot];
  // elbo_unpack copies T = plan.T doubles per restart into the caller's arrays: the layout must be the submitted one
  int T_now = 0;
  { const int n[4] = {a->D * a->K, a->K, a->D, a->K}; for (int g = 0; g < 4; ++g) if (a->optimize[g]) T_now += n[g]; }
  if (a->D != sp->P.dm.D || a->K != sp->P.dm.K || a->R != sp->P.dm.R || T_now != sp->P.dm.T ||
      (a->compute_grad ? 1 : 0) != sp->P.compute_grad)
    return set_err(ctx, VBMC_ERR_INVALID, "%s: args differ from the submitted ones (D, K, R, optimize flags, compute_grad)", who);
  ctx->slot_busy[slot] = false;
  hipError_t e_ = hipEventSynchronize(ctx->slot_ev[slot]);
  if (e_ != hipSuccess) { (void)hipGetLastError(); return set_err(ctx, VBMC_ERR_HIP, "%s: %s", who, hipGetErrorString(e_)); }
  *sp_out = sp;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_elbo_collect(vbmc_ctx* ctx, const vbmc_elbo_args* a, int slot) {
  if (!ctx) return VBMC_ERR_INVALID;
  const SlotPlan* sp = nullptr;
  if (slot < 0 || slot >= VBMC_SLOTS) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_collect: slot must be 0 .. %d", VBMC_SLOTS - 1);
  if (!slot_in_flight(ctx, slot)) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_collect: nothing submitted in slot %d", slot);
  vbmc_ctx* sc = ctx->slot_where[slot];
  { vbmc_status s_ = elbo_collect_core(sc, a, ctx->slot_inner[slot], &sp, "vbmc_elbo_collect"); if (s_) return slot_err(ctx, sc, s_); }
  elbo_unpack(sp->P, a, sp->hout);
  return VBMC_OK;
}

// gives a slot back without its results: waits for the pass in flight (if any) and clears the slot -- for a caller that will not collect
// (an exception between submit and collect, an abandoned generator).  No-op on an idle slot.
extern "C" vbmc_status vbmc_elbo_abandon(vbmc_ctx* ctx, int slot) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (slot < 0 || slot >= VBMC_SLOTS) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_abandon: slot must be 0 .. %d", VBMC_SLOTS - 1);
  if (!slot_in_flight(ctx, slot)) return VBMC_OK;
  vbmc_ctx* sc = ctx->slot_where[slot];
  const int inner = ctx->slot_inner[slot];
  hipError_t e_ = sc->slot_ev[inner] ? hipEventSynchronize(sc->slot_ev[inner]) : hipStreamSynchronize(sc->stream);
  sc->slot_busy[inner] = false;
  if (e_ != hipSuccess) { (void)hipGetLastError(); return set_err(ctx, VBMC_ERR_HIP, "vbmc_elbo_abandon: %s", hipGetErrorString(e_)); }
  return VBMC_OK;
}

// ---- one evaluation sharded over `world` ranks along the hyper-sample axis and the entropy sample chunks (see ShardSpec)
static vbmc_status shard_check(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, int rank, int world) {
  if (!gp || !a) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_shard_*: null gp / args");
  if (world < 1 || rank < 0 || rank >= world) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_shard_*: rank %d of world %d", rank, world);
  if (a->compute_var != 0 || a->separate_K || a->G_s || a->varG_s || a->dG_s || a->dvarG_s)
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "vbmc_elbo_shard_*: the sharded evaluation covers value + gradient without variance "
                                              "(the optimiser-loop call, misc/vpoptimize_vbmc.m:71)");
  if (a->eps_mode != 0 && a->Ns > 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "vbmc_elbo_shard_*: device RNG only (eps_mode 0)");
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_elbo_shard_size(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, int world, size_t* n_doubles) {
  if (!ctx || !n_doubles) return VBMC_ERR_INVALID;
  { vbmc_status s_ = shard_check(ctx, gp, a, 0, world); if (s_) return s_; }
  ElboPlan P;
  { vbmc_status s_ = elbo_plan(ctx, gp, a, P, world); if (s_) return s_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *n_doubles = shard_doubles(P, world);
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_elbo_shard_begin(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, int rank, int world,
                                             double* d_send) {
  if (!ctx || !d_send) return VBMC_ERR_INVALID;
  { vbmc_status s_ = shard_check(ctx, gp, a, rank, world); if (s_) return s_; }
  ElboPlan P;
  { vbmc_status s_ = elbo_plan(ctx, gp, a, P, world); if (s_) return s_; }
  ShardSpec sh;
  sh.mode = 1; sh.rank = rank; sh.world = world; sh.send = d_send;
  HIP_TRY(ctx, hipMemsetAsync(d_send, 0, shard_doubles(P, world) * sizeof(double), ctx->stream));   // unused slots of an uneven split
  { vbmc_status s_ = elbo_enqueue(ctx, gp, P, a->seed, nullptr, 0, &sh); if (s_) return s_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the exchange runs on the caller's stream / communicator
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_elbo_shard_finish(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, int world,
                                              const double* d_gathered) {
  if (!ctx || !d_gathered) return VBMC_ERR_INVALID;
  { vbmc_status s_ = shard_check(ctx, gp, a, 0, world); if (s_) return s_; }
  ElboPlan P;
  { vbmc_status s_ = elbo_plan(ctx, gp, a, P, world); if (s_) return s_; }
  ShardSpec sh;
  sh.mode = 2; sh.world = world; sh.gathered = d_gathered;
  { vbmc_status s_ = elbo_enqueue(ctx, gp, P, a->seed, nullptr, 0, &sh); if (s_) return s_; }
  return elbo_read_results(ctx, P, a);
}

// ------------------------------------------------------------------------------------------
// On-device Adam (utils/fminadam.m:42-102) for R chains in lock-step: the whole loop
// [ELBO+grad pass -> Adam update -> every 20 iterations the slope / random-walk stopping test]
// is enqueued on the stream without host round trips; the host only polls R "done" flags every
// 20 iterations (the reference tests termination on exactly those iterations, fminadam.m:65).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_adam_step(AdamState A, int iter, double* __restrict__ x /*T x R*/,
                                                   const double* __restrict__ out /*R x (5+3T)*/) {
  adam_update_chain(A, iter, x, out, blockIdx.x);
}

// stopping test at iter (a multiple of 20, >= 40): fminadam.m:65-81
__global__ void __launch_bounds__(256) k_adam_check(AdamState A, int iter) {
  __shared__ double red[256];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (A.done[r]) return;
  const int T = A.T, B = 20;
  const double* f = A.ftab + (size_t)r * A.MaxIter + (iter - B);
  // dx = sqrt(sum((mean(x last B) - mean(x previous B)).^2 / B))
  double part = 0.0;
  for (int i = tid; i < T; i += blockDim.x) {
    double a = 0.0, b = 0.0;
    for (int t = 0; t < B; ++t) {
      a += A.xtab[((size_t)r * A.MaxIter + (iter - B + t)) * T + i];
      b += A.xtab[((size_t)r * A.MaxIter + (iter - 2 * B + t)) * T + i];
    }
    double d = a / B - b / B;
    part += d * d / B;
  }
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  if (tid == 0) {
    const double dx = sqrt(red[0]);
    // degree-1 least squares on xxp = linspace(-(B-1)/2, (B-1)/2, B): slope and its variance A(1,1) (:66-69)
    double fm = 0.0;
    for (int t = 0; t < B; ++t) fm += f[t];
    fm /= B;
    double sxx = 0.0, sxy = 0.0;
    for (int t = 0; t < B; ++t) { double xx = -(B - 1) / 2.0 + t; sxx += xx * xx; sxy += xx * (f[t] - fm); }
    const double slope = sxy / sxx;
    double rss = 0.0;
    for (int t = 0; t < B; ++t) { double xx = -(B - 1) / 2.0 + t; double e = f[t] - (fm + slope * xx); rss += e * e; }
    const double svar = rss / (B - 2) / sxx;
    const double TolX = 0.001, TolX_max = 0.1, TolFun_max = A.TolFun * 100.0;
    const double slope_err = sqrt(svar + A.TolFun * A.TolFun), slope_err_max = sqrt(svar + TolFun_max * TolFun_max);
    if ((dx < TolX && fabs(slope) < slope_err_max) || (fabs(slope) < slope_err && dx < TolX_max)) A.done[r] = iter;  // :79
  }
}

extern "C" vbmc_status vbmc_adam_batch(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* a, double TolFun, int MaxIter,
                                       double step_min, double step_max, double step_decay, double* x_out, double* f_out,
                                       int32_t* iters_out, double* xtab_out, double* ftab_out, double* xmid_out) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!a || !a->compute_grad) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_adam_batch needs compute_grad = 1");
  if (MaxIter < 1) return set_err(ctx, VBMC_ERR_INVALID, "MaxIter must be >= 1");
  if (a->eps_mode != 0 && a->Ns > 0) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_adam_batch draws fresh device RNG every iteration (eps_mode 0)");
  if (a->separate_K || a->I_sk || a->J_sjk || a->G_s || a->varG_s || a->dG_s || a->dvarG_s || a->dvarG)      // (accepted by elbo_plan, never served here)
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "vbmc_adam_batch: per-component / per-hyper-sample outputs only through vbmc_elbo_batch");
  ElboPlan P;
  { vbmc_status s_ = elbo_plan(ctx, gp, a, P); if (s_) return s_; }
  const int T = P.dm.T, R = P.dm.R;
  hipStream_t st = ctx->stream;
  const size_t nT = (size_t)T * R, nx = (size_t)T * MaxIter * R, nf = (size_t)MaxIter * R;
  { vbmc_status s_ = ensure(ctx, ctx->misc, (2 * nT + nx + nf) * sizeof(double) + (size_t)R * sizeof(int)); if (s_) return s_; }
  AdamState A{};
  A.m = (double*)ctx->misc.p; A.v = A.m + nT; A.xtab = A.v + nT; A.ftab = A.xtab + nx; A.done = (int*)(A.ftab + nf);
  A.T = T; A.R = R; A.MaxIter = MaxIter; A.step_min = step_min; A.step_max = step_max; A.step_decay = step_decay; A.TolFun = TolFun;
  HIP_TRY(ctx, hipMemsetAsync(A.m, 0, 2 * nT * sizeof(double), st));
  HIP_TRY(ctx, hipMemsetAsync(A.done, 0, (size_t)R * sizeof(int), st));
  std::vector<int> done(R, 0);
  int iter = 0;
  // the factors of iteration `it`'s update (utils/fminadam.m:53-57), handed to the kernels with the state
  auto adam_set_iter = [&](AdamState& S_, int it) {
    S_.c1 = 1.0 - std::pow(0.9, (double)it); S_.c2 = 1.0 - std::pow(0.999, (double)it);
    S_.step = step_min + (step_max - step_min) * std::exp(-(double)it / step_decay);
  };
  // Where the NEXT iteration follows without a stopping test in between, this iteration's finalize kernel applies the Adam update
  // and unpacks the new theta itself (k_finalize_ws: fused), and the next pass starts at its log-joint kernel; before a stopping
  // test (every 20 iterations from the 40th, utils/fminadam.m:65) and at the end the update is a launch of its own.
  bool prepped = false;   // the previous pass has already applied its update and unpacked theta for this one
  for (iter = 1; iter <= MaxIter; ++iter) {
    const bool check = iter % 20 == 0 && iter >= 40;
    const bool last = check || iter == MaxIter;
    const bool fuse = !last;
    adam_set_iter(A, iter);
    { vbmc_status s_ = elbo_enqueue(ctx, gp, P, a->seed + (unsigned long long)iter, nullptr, 0, nullptr, fuse ? &A : nullptr, iter, prepped); if (s_) return s_; }
    prepped = fuse;
    // the stopping test and the final read-back need this iteration's update now; the next pass unpacks theta itself
    if (!fuse) hipLaunchKernelGGL(k_adam_step, dim3(R), dim3(256), 0, st, A, iter, P.d_theta, P.d_out);
    if (check) {
      hipLaunchKernelGGL(k_adam_check, dim3(R), dim3(256), 0, st, A, iter);
      HIP_TRY(ctx, hipMemcpyAsync(done.data(), A.done, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      bool all = true;
      for (int r = 0; r < R; ++r) all = all && done[r] != 0;
      if (all) break;
    }
  }
  if (iter > MaxIter) iter = MaxIter;
  HIP_TRY(ctx, hipGetLastError());
  // outputs: x = mean of the last 20 iterates, f = mean of the last 20 values (fminadam.m:96-97).  Only the iterations that were
  // run come back: `iter` rows of each chain's table (a strided copy), not the MaxIter the table was sized for -- with
  // MaxIter = 1e4 and chains that stop after a few hundred iterations that is the difference between megabytes and hundreds of them
  const size_t nit = (size_t)iter;
  std::vector<double> xt(nit * T * R), ft(nit * R);
  HIP_TRY(ctx, hipMemcpy2DAsync(xt.data(), nit * T * sizeof(double), A.xtab, (size_t)MaxIter * T * sizeof(double), nit * T * sizeof(double), R,
                                hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpy2DAsync(ft.data(), nit * sizeof(double), A.ftab, (size_t)MaxIter * sizeof(double), nit * sizeof(double), R,
                                hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(done.data(), A.done, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  for (int r = 0; r < R; ++r) {
    const int it = done[r] ? done[r] : iter;
    const int nb = it < 20 ? it : 20;
    const double* xr = xt.data() + (size_t)r * nit * T;
    const double* fr = ft.data() + (size_t)r * nit;
    if (iters_out) iters_out[r] = it;
    for (int i = 0; i < T; ++i) {
      double acc = 0.0;
      for (int t = it - nb; t < it; ++t) acc += xr[(size_t)t * T + i];
      if (x_out) x_out[(size_t)r * T + i] = acc / nb;
    }
    double fa = 0.0;
    for (int t = it - nb; t < it; ++t) fa += fr[t];
    if (f_out) f_out[r] = fa / nb;
    if (xmid_out) {   // the iterate with the smallest recorded objective: [~,idx_mid] = min(fval_lst) (misc/vpoptimize_vbmc.m:133; first minimum, NaN skipped)
      int best = 0;
      bool have = false;
      for (int t = 0; t < it; ++t)
        if (!std::isnan(fr[t]) && (!have || fr[t] < fr[best])) { best = t; have = true; }
      memcpy(xmid_out + (size_t)r * T, xr + (size_t)best * T, (size_t)T * sizeof(double));
    }
    if (xtab_out) memcpy(xtab_out + (size_t)r * MaxIter * T, xr, (size_t)it * T * sizeof(double));
    if (ftab_out) memcpy(ftab_out + (size_t)r * MaxIter, fr, (size_t)it * sizeof(double));
  }
  return VBMC_OK;
}

// test / reporting hook: the instantiation the Monte-Carlo entropy of a D-dimensional K-component mixture runs on (dense mode):
// qs = ceil((D + 2) / 4), kt = k-tiles per wave, hv = waves per workgroup, tail = tail values per lane (0: none).  Returns 0 when
// the matrix-core kernel does not serve the shape (the VALU kernel does).
// largest variational parameter count whose five T-vectors fit k_var_final's LDS (with S = 1, K = 1: the bound elbo_plan applies per call)
static int limit_T_vargrad() {
  int T = 1;
  while (VAR_FINAL_LDS(1, 1, T + 1) <= 160 * 1024) ++T;
  return T;
}
extern "C" vbmc_status vbmc_get_limits(vbmc_limits* out) {
  if (!out || out->struct_size != sizeof(vbmc_limits)) return VBMC_ERR_INVALID;
  out->max_D = VBMC_LIM_D;
  out->max_K = VBMC_LIM_K;
  out->max_N = trsm_max_n();
  out->max_Na = VBMC_LIM_NA;
  out->max_T_vargrad = limit_T_vargrad();
  out->delta_ok = 1;
  out->meanfun_mask = VBMC_LIM_MEANFUN_MASK;
  return VBMC_OK;
}

extern "C" int vbmc_entropy_plan(int D, int K, int* qs, int* kt, int* hv, int* tail) {
  if (lane_entropy_fits(D, K, 0.0)) {       // small mixtures: k_entropy_lane<DT, KP> (qs: DT, kt: KP, four waves per workgroup)
    if (qs) *qs = 2 * ((D + 1) / 2);
    if (kt) *kt = 2 * ((K + 1) / 2);
    if (hv) *hv = ENT_LANE_WAVES_HOST;
    if (tail) *tail = 0;
    return 2;
  }
  int q = 0, k = 0, h = 0;
  const bool ok = mfma_entropy_fits(D, K, 0.0, &q, &k, &h);
  if (qs) *qs = q;
  if (kt) *kt = k;
  if (hv) *hv = h & 15;
  if (tail) *tail = h >> 4;
  return ok ? 1 : 0;
}

#ifdef VBMC_INSTRUMENT
extern "C" int vbmc_dbg_fin_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_dbg), 64 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// test hook: y[i] = exp(x[i]) with the hot-loop implementations (variant 0: vb_exp, 1: vb_exp_tab<0>, 2: vb_exp_tab<1>)
__global__ void k_test_exp(int n, int variant, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double tab[VB_EXP_TAB_N];
  for (int t = threadIdx.x; t < VB_EXP_TAB_N; t += blockDim.x) tab[t] = c_exp2_tab[t];
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = variant == 2 ? vb_exp_tab<1>(x[i], tab) : (variant ? vb_exp_tab<0>(x[i], tab) : vb_exp(x[i]));
}

// variant 3: vb_exp_tab1k, the entropy kernel's exp, on y = x * 1024/ln2 (in the kernel the factor sits in the MFMA operands)
#include "exp2_tab1k.h"
// variant 3: as the entropy kernel is built (VB_EXP_TAB1K_QUAD), 4: the economised cubic, 5: the economised quadratic
__global__ void k_test_exp1k(int n, int variant, const double* __restrict__ x, double* __restrict__ y) {
  __shared__ double tab[VB_EXP_TAB1K_N];
  for (int t = threadIdx.x; t < VB_EXP_TAB1K_N; t += blockDim.x) tab[t] = c_exp2_tab1k[t];
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double ys;   // a ROUNDED product, as the MFMA delivers it: the compiler must not contract it into the reduction's subtraction
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(ys) : "v"(x[i]), "v"(VB_EXP_TAB1K_SCALE));
    const bool quad = variant == 3 ? VB_EXP_TAB1K_QUAD : variant == 5;
    y[i] = quad ? vb_exp_tab1k<true>(ys, tab) : vb_exp_tab1k<false>(ys, tab);
  }
}

extern "C" vbmc_status vbmc_test_exp(vbmc_ctx* ctx, int n, int variant, const double* x, double* y) {
  if (!ctx || n <= 0 || !x || !y) return VBMC_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  { vbmc_status s_ = ensure(ctx, ctx->misc, 2 * (size_t)n * sizeof(double)); if (s_) return s_; }
  double* dx = (double*)ctx->misc.p;
  HIP_TRY(ctx, hipMemcpyAsync(dx, x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (variant >= 3 && variant <= 5) hipLaunchKernelGGL(k_test_exp1k, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, variant, dx, dx + n);
  else if (variant < 0 || variant > 5) return VBMC_ERR_INVALID;
  else hipLaunchKernelGGL(k_test_exp, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, variant, dx, dx + n);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(y, dx + n, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_rng_dump(vbmc_ctx* ctx, int D, int K, int R, int Ns, uint64_t seed, double* eps_host) {
  if (!ctx || !eps_host || D <= 0 || K <= 0 || R <= 0 || Ns <= 0) return VBMC_ERR_INVALID;
  const int Mh = (Ns + 1) / 2;
  const size_t n = (size_t)D * Mh * K * R;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  { vbmc_status s_ = ensure(ctx, ctx->eps, n * sizeof(double)); if (s_) return s_; }
  const long long total = (long long)R * K * Mh;
  hipLaunchKernelGGL(k_rng_dump, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, D, K, R, Mh,
                     (unsigned long long)seed, (double*)ctx->eps.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(eps_host, ctx->eps.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VBMC_OK;
}
