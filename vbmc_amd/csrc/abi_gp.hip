// C-ABI entry points for the gplite GP surrogate: vbmc_sq_dist, vbmc_gp_post, vbmc_gp_pred
// (include/vbmc_hip.h).  Host side: O(N) hyper-parameter transforms, the Cholesky jitter-retry
// loop (gplite_core.m:77-80,91-94), launches, D2H of the posterior.
#include <functional>
#include <algorithm>
#include <cmath>
#include <limits>

#include "gp_kernels.h"

namespace {

struct TmpBuf {   // a pooled device block (common.h pool_get/pool_put), returned to the context's pool on scope exit
  void* p = nullptr;
  vbmc_ctx* owner = nullptr;
  bool is_view = false;   // a window into another block (the packed input block of gp_factorize): nothing to return
  ~TmpBuf() { if (p && !is_view) pool_put(owner, p); }
  hipError_t alloc(vbmc_ctx* ctx, size_t bytes) { owner = ctx; return pool_get(ctx, bytes, &p); }
  void view(void* q) { p = q; is_view = true; }
  template <typename T> T* as() { return (T*)p; }
};

// gplite_noisefun.m:176-210 on the host (O(N)): returns per-point sn2 (length N)
void noise_vector(const int32_t nf[3], const double* hn, int N, const double* y, const double* s2, std::vector<double>& sn2) {
  int idx = 0;
  double base;
  if (nf[0] == 0) base = 2.220446049250313e-16;
  else { base = std::exp(2.0 * hn[idx]); idx++; }
  sn2.assign(N, base);
  if (nf[1] == 1 && s2) { for (int n = 0; n < N; ++n) sn2[n] += s2[n]; }
  else if (nf[1] == 2 && s2) { double c = std::exp(hn[idx]); for (int n = 0; n < N; ++n) sn2[n] += c * s2[n]; idx++; }
  else if (nf[1] == 2) idx++;
  if (nf[2] == 1) {
    if (y) {
      double ythr = hn[idx], w2 = std::exp(2.0 * hn[idx + 1]);
      for (int n = 0; n < N; ++n) { double zz = std::max(0.0, ythr - y[n]); sn2[n] += w2 * zz * zz; }
    }
  }
}

int noise_nhyp(const int32_t nf[3]) { return (nf[0] == 1) + (nf[1] == 2) + 2 * (nf[2] == 1); }

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" vbmc_status vbmc_sq_dist(vbmc_ctx* ctx, int D, int n, int m, const double* a, const double* b, double* C) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (D <= 0 || n <= 0 || !a || !C) return set_err(ctx, VBMC_ERR_INVALID, "sq_dist: Wrong number of arguments.");
  const bool self = (b == nullptr);
  if (self) { b = a; m = n; }
  if (m <= 0) return set_err(ctx, VBMC_ERR_INVALID, "sq_dist: empty b");
  // mean used for stabilisation (sq_dist.m:26,36), computed in MATLAB's order on the host: O(D(n+m))
  std::vector<double> mu(D);
  for (int d = 0; d < D; ++d) {
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < n; ++i) sa += a[d + (size_t)D * i];
    if (self) mu[d] = sa / n;
    else {
      for (int j = 0; j < m; ++j) sb += b[d + (size_t)D * j];
      mu[d] = ((double)m / (n + m)) * (sb / m) + ((double)n / (n + m)) * (sa / n);
    }
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  TmpBuf da, db, dmu, dC;
  HIP_TRY(ctx, da.alloc(ctx, (size_t)D * n * 8));
  HIP_TRY(ctx, dmu.alloc(ctx, (size_t)D * 8));
  HIP_TRY(ctx, dC.alloc(ctx, (size_t)n * m * 8));
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(da.p, a, (size_t)D * n * 8, hipMemcpyHostToDevice, st));
  const double* dbp = da.as<double>();
  if (!self) {
    HIP_TRY(ctx, db.alloc(ctx, (size_t)D * m * 8));
    HIP_TRY(ctx, hipMemcpyAsync(db.p, b, (size_t)D * m * 8, hipMemcpyHostToDevice, st));
    dbp = db.as<double>();
  }
  HIP_TRY(ctx, hipMemcpyAsync(dmu.p, mu.data(), (size_t)D * 8, hipMemcpyHostToDevice, st));
  dim3 grid(((n + 15) / 16 + 3) / 4, (m + 15) / 16);
  hipLaunchKernelGGL(k_sq_dist_mfma, grid, dim3(256), 0, st, D, n, m, da.as<double>(), dbp, dmu.as<double>(), dC.as<double>());
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(C, dC.p, (size_t)n * m * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
// Shared by vbmc_gp_post and vbmc_gp_nlz: kernel matrices, jittered Cholesky and alpha for S hyper-parameter
// vectors, left on the device (gplite_core.m:33-102).
namespace {

struct GpFactor {
  int Ncov = 0, Nnoise = 0, Nmean = 0, cw = 16;      // cw: slab width of the triangular solves at this N (trsm_cw_for)
  std::vector<double> sn2all, scal, sn2min;         // host copies: S x N noise, S x 4 {sn2div, mult, lchol, sl}
  std::vector<unsigned char> lch, failed;           // Lchol flag; 1 = Cholesky still failing after 10 retries
  bool any_inv = false;
  size_t tlds = 0;
  double* pin_out = nullptr;
  // optimistic first try: the Cholesky's failure indices (as doubles, behind alpha on the device: d_pfd) are NOT waited for;
  // the caller copies them out with its own results, points h_pfd at them and looks after its own synchronisation (gp_factor_ok)
  double* d_pfd = nullptr;
  const double* h_pfd = nullptr;
  bool unchecked = false;
  bool alpha_event = false;    // alpha is being computed on the second stream: wait for ctx->ev_join before reading it
  // caller's extra inputs riding in the packed upload (set before gp_factorize): extra_in doubles, filled by fill_extra
  size_t extra_in = 0;
  std::function<void(double*)> fill_extra;
  bool defer_alpha = false;     // set by the caller: it launches alpha's backward solve itself (dz -> dal, k_tri_inverse2_alpha)
  TmpBuf dIn, dX, dy, dhyp, dXc, daa, dsn2, dscal, dact, dA, dpf, dr, dz, dones, dninv, dlch, dal, dfinv, dExtra;   // dIn: the packed inputs (dX .. dninv, dExtra are windows)
};

// fail_is_error: vbmc_gp_post refuses a matrix that is still not positive definite after the retries;
// vbmc_gp_nlz marks that hyper-parameter vector as failed (NaN result, gplite_train.m:542-546) and goes on.
//
// Device schedule (round 5): ONE upload of the packed inputs; k_gp_scale (scaled inputs, row norms AND the residual y - m);
// k_gp_build; k_chol2, which also leaves the inverses of the diagonal blocks (the triangular solves' Finv) and the forward
// solve z = R' \ (y - m) behind; the backward half of alpha's solve with the 1 / sl scaling folded in.  (Round 4: the
// residual, the block inverses, the two-sided solve and the scaling were four more launches after the factorisation.)
vbmc_status gp_factorize(vbmc_ctx* ctx, const char* who, int N, int D, int S, int Nhyp, int meanfun, const int32_t noisefun[3],
                         const double* X, const double* y, const double* s2, const double* hyp, bool fail_is_error,
                         GpFactor& f, size_t pin_extra_doubles = 0, bool optimistic = false, bool alpha_aside = false) {
  if (N <= 0 || D <= 0 || S <= 0 || !X || !y || !hyp || !noisefun)
    return set_err(ctx, VBMC_ERR_INVALID, "%s: bad arguments", who);
  if (D > VBMC_LIM_D) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "D = %d > %d not accelerated", D, VBMC_LIM_D);
  if (!(meanfun == 0 || meanfun == 1 || meanfun == 4))
    return set_err(ctx, VBMC_ERR_UNSUPPORTED, "gplite mean function %d not accelerated (0,1,4 are)", meanfun);
  const int Ncov = D + 1, Nnoise = noise_nhyp(noisefun);
  const int Nmean = meanfun == 0 ? 0 : (meanfun == 1 ? 1 : 2 * D + 1);
  f.Ncov = Ncov; f.Nnoise = Nnoise; f.Nmean = Nmean;
  if (Nhyp != Ncov + Nnoise + Nmean)
    return set_err(ctx, VBMC_ERR_INVALID, "%s:dimmismatch Number of hyperparameters mismatched with GP model specification.",
                   fail_is_error ? "gplite_post" : "gplite_nlZ");
  // right-hand-side slabs of the triangular solves live in LDS: 16 columns wide up to N = 1136, narrower beyond (trsm_cw_for);
  // the Cholesky panel moves to a global scratch block when it no longer fits
  if (trsm_cw_for(N) == 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "N = %d > %d not accelerated", N, trsm_max_n());
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  // host: noise vectors, Lchol flags (gplite_core.m:33-40,67)
  std::vector<double>&sn2all = f.sn2all, &scal = f.scal, &sn2min = f.sn2min;
  sn2all.assign((size_t)S * N, 0.0); scal.assign((size_t)S * 4, 0.0); sn2min.assign(S, 0.0);
  std::vector<unsigned char>& lch = f.lch;
  lch.assign(S, 0); f.failed.assign(S, 0);
  std::vector<unsigned char> active(S, 1), ones(S, 1), needinv(S);
  std::vector<double> tmp;
  for (int s = 0; s < S; ++s) {
    noise_vector(noisefun, hyp + (size_t)s * Nhyp + Ncov, N, y, s2, tmp);
    std::copy(tmp.begin(), tmp.end(), sn2all.begin() + (size_t)s * N);
    double mn = *std::min_element(tmp.begin(), tmp.end());
    sn2min[s] = mn;
    lch[s] = mn >= 1e-6 ? 1 : 0;
    needinv[s] = lch[s] ? 0 : 1;
    f.any_inv |= !lch[s];
    scal[s * 4 + 0] = lch[s] ? mn : 1.0;  // sn2div
    scal[s * 4 + 1] = 1.0;                // sn2_mult
    scal[s * 4 + 2] = lch[s];
    scal[s * 4 + 3] = lch[s] ? scal[s * 4 + 0] : 1.0;   // sl = sn2div * sn2_mult (:82,:96); rewritten below if a retry inflates the noise
  }
  TmpBuf &dX = f.dX, &dy = f.dy, &dhyp = f.dhyp, &dXc = f.dXc, &daa = f.daa, &dsn2 = f.dsn2, &dscal = f.dscal, &dact = f.dact,
         &dA = f.dA, &dpf = f.dpf, &dr = f.dr, &dz = f.dz, &dones = f.dones, &dninv = f.dninv;
  // Inputs: ONE pinned block [X | y | hyp | sn2 | scal | ones needinv active | caller's extras], one asynchronous copy.  (Round 2
  // issued eight hipMemcpyAsync from pageable memory -- each of them staged and waited for by the runtime, ~100 us of host time
  // in a call whose kernels take 0.36 ms.)
  const size_t nX = (size_t)N * D, nH = (size_t)Nhyp * S, nS = (size_t)S * N, nC = (size_t)S * 4;
  const size_t nB = (4 * (size_t)S + 7) / 8;             // ones | needinv | active | lchol
  const size_t in_doubles = nX + N + nH + nS + nC + nB + f.extra_in;
  { vbmc_status s_ = ensure_pin(ctx, (in_doubles + pin_extra_doubles) * 8 + 8); if (s_) return s_; }
  double* hin = (double*)ctx->pin;
  f.pin_out = hin + in_doubles;      // the caller's results come back through the same pinned block (pin_extra_doubles of it)
  memcpy(hin, X, nX * 8);
  memcpy(hin + nX, y, (size_t)N * 8);
  memcpy(hin + nX + N, hyp, nH * 8);
  memcpy(hin + nX + N + nH, sn2all.data(), nS * 8);
  memcpy(hin + nX + N + nH + nS, scal.data(), nC * 8);
  unsigned char* hb = (unsigned char*)(hin + nX + N + nH + nS + nC);
  memcpy(hb, ones.data(), S); memcpy(hb + S, needinv.data(), S); memcpy(hb + 2 * S, active.data(), S); memcpy(hb + 3 * S, lch.data(), S);
  if (f.extra_in && f.fill_extra) f.fill_extra(hin + nX + N + nH + nS + nC + nB);
  TmpBuf& dIn = f.dIn;
  HIP_TRY(ctx, dIn.alloc(ctx, in_doubles * 8));
  {
    double* din = dIn.as<double>();
    dX.view(din); dy.view(din + nX); dhyp.view(din + nX + N); dsn2.view(din + nX + N + nH); dscal.view(din + nX + N + nH + nS);
    unsigned char* db = (unsigned char*)(din + nX + N + nH + nS + nC);
    dones.view(db); dninv.view(db + S); dact.view(db + 2 * S); f.dlch.view(db + 3 * S);
    f.dExtra.view(din + nX + N + nH + nS + nC + nB);
  }
  TmpBuf &dal = f.dal, &dfinv = f.dfinv;
  HIP_TRY(ctx, dXc.alloc(ctx, (size_t)S * N * D * 8));
  HIP_TRY(ctx, daa.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, dA.alloc(ctx, (size_t)S * N * N * 8));
  HIP_TRY(ctx, dpf.alloc(ctx, (size_t)S * sizeof(int)));
  HIP_TRY(ctx, dr.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, dz.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, dal.alloc(ctx, ((size_t)S * N + S) * 8));       // alpha, and behind it the failure indices as doubles
  HIP_TRY(ctx, dfinv.alloc(ctx, (size_t)S * TRSM_NBLK(N) * 256 * 8));
  f.d_pfd = dal.as<double>() + (size_t)S * N;
  const int moff = Ncov + Nnoise;
  // (small blocks by a kernel that reads the pinned block over the host link instead of the copy engine: the hand-over from a DMA
  // copy to the first kernel of the stream is 8-9 us on top of the copy's 6.5 -- abi_elbo.hip: copy_by_kernel)
  if (copy_by_kernel(in_doubles * 8))
    hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)std::min<size_t>((in_doubles + 255) / 256, 1024)), dim3(256), 0, st, in_doubles, (const double*)hin,
                       dIn.as<double>());
  else
    HIP_TRY(ctx, hipMemcpyAsync(dIn.p, hin, in_doubles * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_gp_scale, dim3(4, S), dim3(256), 0, st, N, D, Nhyp, dX.as<double>(), dhyp.as<double>(), dXc.as<double>(), daa.as<double>(),
                     moff, meanfun, dy.as<double>(), dr.as<double>());

  // jittered Cholesky: up to 10 tries, noise multiplier x10 per failure (gplite_core.m:77-80,91-94)
  // the 16 x N panel of the Cholesky lives in LDS up to N = 1120, in a global scratch block beyond (chol_mfma.h)
  TmpBuf dPg;
  if (chol2_needs_gpanel(N, true)) HIP_TRY(ctx, dPg.alloc(ctx, (size_t)S * 16 * (size_t)(((N + 15) >> 4) << 4) * 8));
  std::vector<int> pf(S);
  bool pending = true, retried = false;
  for (int iter = 0; iter < 10 && pending; ++iter) {
    if (iter > 0) {   // (the first try's copies are part of the packed block)
      HIP_TRY(ctx, hipMemcpyAsync(dscal.p, scal.data(), (size_t)S * 4 * 8, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipMemcpyAsync(dact.p, active.data(), S, hipMemcpyHostToDevice, st));
    }
    DISPATCH_GPDT(D, hipLaunchKernelGGL((k_gp_build<DT>), dim3((N + GPB_T - 1) / GPB_T, (N + GPB_T - 1) / GPB_T, S), dim3(256), 0, st, N, D,
                                        Nhyp, dhyp.as<double>(), dXc.as<double>(), daa.as<double>(), dsn2.as<double>(), dscal.as<double>(),
                                        dact.as<unsigned char>(), dA.as<double>()));
    HIP_TRY(ctx, chol2_launch(N, S, dA.as<double>(), dpf.as<int>(), dact.as<unsigned char>(), dPg.p ? dPg.as<double>() : nullptr, st,
                              dfinv.as<double>(), f.d_pfd, dr.as<double>(), dz.as<double>()));
    if (optimistic) {
      // Almost every factorisation succeeds at the first try (the retries exist for hyper-parameter vectors at the edge of the
      // prior).  The flags are NOT waited for: everything downstream is enqueued as if the try had succeeded, they travel with
      // the caller's results (d_pfd -> h_pfd) and the caller looks at them at its own final synchronisation (gp_factor_ok); if
      // one is set it repeats the call with the retry loop -- one host round trip (25 us of an otherwise idle device) and one
      // copy less per call
      f.unchecked = true;
      std::fill(active.begin(), active.end(), 0);
      pending = false;
      break;
    }
    HIP_TRY(ctx, hipMemcpyAsync(pf.data(), dpf.p, (size_t)S * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    pending = false;
    for (int s = 0; s < S; ++s) {
      if (!active[s]) continue;
      if (pf[s] > 0) { scal[s * 4 + 1] *= 10.0; pending = true; retried = true; }  // sn2_mult = sn2_mult*10
      else active[s] = 0;
    }
  }
  if (pending) {
    // MATLAB leaves the loop with the last multiplier even when chol still fails; we refuse instead.
    if (fail_is_error)
      return set_err(ctx, VBMC_ERR_NOT_POSDEF, "gplite_core: Cholesky failed after 10 noise-inflation retries");
    for (int s = 0; s < S; ++s) f.failed[s] = active[s];
  }
  if (retried) {   // (without a retry the packed block already carries sl)
    for (int s = 0; s < S; ++s) scal[s * 4 + 3] = lch[s] ? scal[s * 4 + 0] * scal[s * 4 + 1] : 1.0;  // sl (:82,:96)
    HIP_TRY(ctx, hipMemcpyAsync(dscal.p, scal.data(), (size_t)S * 4 * 8, hipMemcpyHostToDevice, st));
  }

  // alpha = L\(L'\(y-m)) / sl  (:102): the forward half came out of the factorisation (dz).  alpha_aside (vbmc_gp_nlz with a
  // gradient, few matrices): the backward half runs on the context's second stream while the caller inverts the factor on the
  // first -- two latency-bound kernels that do not depend on each other; the caller waits for ctx->ev_join before it reads alpha.
  f.cw = trsm_cw_for(N);
  f.tlds = TRSM_LDS_BYTES_CW(N, f.cw);
  hipStream_t sa = st;
  f.alpha_event = false;
  if (f.defer_alpha) { HIP_TRY(ctx, hipGetLastError()); return VBMC_OK; }
  if (alpha_aside && ctx->ev_fork && ctx->ev_join && ctx_aux(ctx)) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    sa = ctx->aux;
    f.alpha_event = true;
  }
  if (N <= ASOLVE1_THREADS)
    hipLaunchKernelGGL(k_alpha_solve1, dim3(S), dim3(ASOLVE1_THREADS), 0, sa, N, dA.as<double>(), dfinv.as<double>(), dones.as<unsigned char>(),
                       dz.as<double>(), dal.as<double>(), 1, dscal.as<double>());
  else
    hipLaunchKernelGGL(k_alpha_solve, dim3(S), dim3(ASOLVE_THREADS), (size_t)((TRSM_NBLK(N) << 4) + 16) * sizeof(double), sa, N, dA.as<double>(),
                       dfinv.as<double>(), dones.as<unsigned char>(), dz.as<double>(), dal.as<double>(), 1, dscal.as<double>());
  if (f.alpha_event) HIP_TRY(ctx, hipEventRecord(ctx->ev_join, sa));
  HIP_TRY(ctx, hipGetLastError());
  return VBMC_OK;
}

// after the caller's synchronisation: did the optimistic first try succeed for every matrix?
bool gp_factor_ok(const GpFactor& f, int S) {
  if (!f.unchecked) return true;
  if (!f.h_pfd) return false;            // nobody brought the flags back: take the checked path
  for (int s = 0; s < S; ++s)
    if (f.h_pfd[s] > 0.0) return false;
  return true;
}

}  // namespace

static vbmc_status gp_post_impl(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int meanfun, const int32_t noisefun[3],
                                const double* X, const double* y, const double* s2, const double* hyp,
                                double* alpha, double* L, double* sW, double* sn2_mult, uint8_t* Lchol,
                                vbmc_gp** gp_out, bool optimistic);
#define VBMC_INTERNAL_RETRY 1000
extern "C" vbmc_status vbmc_gp_post(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int meanfun, const int32_t noisefun[3],
                                    const double* X, const double* y, const double* s2, const double* hyp,
                                    double* alpha, double* L, double* sW, double* sn2_mult, uint8_t* Lchol,
                                    vbmc_gp** gp_out) {
  vbmc_status st = gp_post_impl(ctx, N, D, S, Nhyp, meanfun, noisefun, X, y, s2, hyp, alpha, L, sW, sn2_mult, Lchol, gp_out, true);
  if (st == VBMC_INTERNAL_RETRY) st = gp_post_impl(ctx, N, D, S, Nhyp, meanfun, noisefun, X, y, s2, hyp, alpha, L, sW, sn2_mult, Lchol, gp_out, false);
  return st;
}

static vbmc_status gp_post_impl(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int meanfun, const int32_t noisefun[3],
                                const double* X, const double* y, const double* s2, const double* hyp,
                                double* alpha, double* L, double* sW, double* sn2_mult, uint8_t* Lchol,
                                vbmc_gp** gp_out, bool optimistic) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (gp_out) *gp_out = nullptr;
  GpFactor f;
  // Device-resident posterior, first (optimistic) try: what a vbmc_gp needs beyond the factorisation's own buffers -- the
  // per-sample constants of the log joint, sn2_eff, the column means of X, the noise multipliers -- rides in the packed upload
  // (they depend on the inputs only, the multipliers being 1 unless a retry inflates the noise), and the surrogate ADOPTS the
  // device blocks of the factorisation: no second round of uploads, no copy of the S N x N factors, one synchronisation per call.
  const bool resident = gp_out != nullptr && optimistic && D <= 32 && D > 0 && S > 0 && N > 0;
  const size_t nG = resident ? (size_t)S * GPC_STRIDE(D) : 0;
  std::vector<double> sW1(S > 0 ? S : 0), mult(S > 0 ? S : 0, 1.0);
  if (resident) {
    f.extra_in = nG + (size_t)S + (size_t)D + (size_t)S;            // gpc | sn2_eff | meanX | mult
    f.fill_extra = [&](double* e) {
      gp_constants(D, S, Nhyp, D + 1, noise_nhyp(noisefun), meanfun, hyp, e);
      for (int s = 0; s < S; ++s) {
        const double w = 1.0 / std::sqrt(f.sn2min[s]);               // post.sW(1) with sn2_mult = 1  (:281)
        e[nG + s] = 1.0 / (w * w);                                   // gplogjoint.m:160
        e[nG + S + D + s] = 1.0;
      }
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int n = 0; n < N; ++n) acc += X[n + (size_t)N * d];
        e[nG + S + d] = acc / N;
      }
    };
  }
  { vbmc_status s_ = gp_factorize(ctx, "vbmc_gp_post", N, D, S, Nhyp, meanfun, noisefun, X, y, s2, hyp, true, f, (size_t)S * N + S, optimistic); if (s_ != VBMC_OK) return s_; }
  hipStream_t st = ctx->stream;
  const int Ncov = f.Ncov, Nnoise = f.Nnoise;
  const bool any_inv = f.any_inv;
  std::vector<double>&scal = f.scal, &sn2min = f.sn2min;
  std::vector<unsigned char>& lch = f.lch;
  TmpBuf &dA = f.dA, &dal = f.dal, &dfinv = f.dfinv, &dninv = f.dninv, dXi;

  double* alh = f.pin_out;   // pinned: the copy is asynchronous, one synchronisation below; the failure indices ride behind alpha
  if (copy_by_kernel(((size_t)S * N + S) * 8))
    hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)std::min<size_t>(((size_t)S * N + S + 255) / 256, 1024)), dim3(256), 0, st, (size_t)S * N + S,
                       (const double*)dal.as<double>(), alh);
  else
    HIP_TRY(ctx, hipMemcpyAsync(alh, dal.p, ((size_t)S * N + S) * 8, hipMemcpyDeviceToHost, st));
  f.h_pfd = alh + (size_t)S * N;
  const bool wantL = L != nullptr || gp_out != nullptr;
  // Low-noise samples (:84-99): gp.post(s).L = -inv(K + sn2 I) = -T'T with T = inv(R').  Round 5, N <= 1024: the workgroup-per-slab
  // inverse of the factor and the rank-k product on the matrix cores (k_tri_inverse2 + k_syrk_tt, as in the marginal-likelihood
  // gradient), the product written NEGATED and in full over the factor it came from -- the factorisation's block then holds
  // gp.post(s).L for every sample, whichever branch it took (k_spd_inverse, 270 us at N = 400, wrote the inverse elsewhere and left the
  // sign and one copy per sample to the assembly).  alpha's solve (gp_factorize) read the factor before this point of the stream.
  const bool inv_in_place = wantL && any_inv && tri_inverse2_fits(N) && trsm2_wanted(N);
  TmpBuf dTTi;
  if (inv_in_place) {
    HIP_TRY(ctx, dTTi.alloc(ctx, (size_t)S * N * N * 8));
    HIP_TRY(ctx, tri_inverse_launch(st, N, S, dA.as<double>(), dfinv.as<double>(), dninv.as<unsigned char>(), dTTi.as<double>(), 1));
    syrk_tt_launch(st, N, S, dTTi.as<double>(), dninv.as<unsigned char>(), dA.as<double>(), true);
    HIP_TRY(ctx, hipGetLastError());
  } else if (wantL && any_inv) {
    // pL = -L\(L'\eye(N)) for low-noise samples (:98); the sign is applied where the matrix is consumed
    HIP_TRY(ctx, dXi.alloc(ctx, (size_t)S * N * N * 8));
    SPD_INVERSE_LAUNCH(ctx, N, S, st, dA.as<double>(), dfinv.as<double>(), dninv.as<unsigned char>(), dXi.as<double>());
    HIP_TRY(ctx, hipGetLastError());
  }
  if (L) {
    // the caller's copy of gp.post(s).L (D2H only when asked for)
    if (!any_inv || inv_in_place) {   // one contiguous block: every sample on the Cholesky branch, or the inverses already in their places
      vbmc_status s_ = d2h_bounced(ctx, L, dA.as<double>(), (size_t)S * N * N * 8);
      if (s_) return s_;
    } else {
      for (int s = 0; s < S; ++s) {
        const double* src = lch[s] ? dA.as<double>() + (size_t)s * N * N : dXi.as<double>() + (size_t)s * N * N;
        vbmc_status s_ = d2h_bounced(ctx, L + (size_t)s * N * N, src, (size_t)N * N * 8);
        if (s_) return s_;
      }
    }
  }
  HIP_TRY(ctx, stream_wait_latency(st, N > 1024));
  if (!gp_factor_ok(f, S)) return VBMC_INTERNAL_RETRY;     // a first try failed: once more with the noise-inflation loop
  if (L && any_inv && !inv_in_place)
    for (int s = 0; s < S; ++s)
      if (!lch[s]) for (size_t i = 0; i < (size_t)N * N; ++i) L[(size_t)s * N * N + i] = -L[(size_t)s * N * N + i];

  for (int s = 0; s < S; ++s) {
    mult[s] = scal[s * 4 + 1];
    sW1[s] = 1.0 / std::sqrt(sn2min[s] * mult[s]);  // post.sW = ones(N,1)./sqrt(min(sn2)*sn2_mult)  (:281)
  }
  if (alpha) memcpy(alpha, alh, (size_t)S * N * 8);
  if (sW) for (int s = 0; s < S; ++s) for (int n = 0; n < N; ++n) sW[(size_t)s * N + n] = sW1[s];
  if (sn2_mult) memcpy(sn2_mult, mult.data(), S * 8);
  if (Lchol) memcpy(Lchol, lch.data(), S);
  if (gp_out && resident && (!any_inv || inv_in_place)) {
    // adopt: the surrogate's device blocks ARE the factorisation's (d_lchol = the per-sample branch flags of the packed upload)
    vbmc_gp* gp = new vbmc_gp();
    gp->N = N; gp->D = D; gp->S = S; gp->Nhyp = Nhyp; gp->Ncov = Ncov; gp->Nnoise = Nnoise; gp->meanfun = meanfun;
    gp->hyp_host.assign(hyp, hyp + (size_t)Nhyp * S);
    gp->sn2_eff.resize(S); gp->Lchol.assign(lch.begin(), lch.end());
    for (int s = 0; s < S; ++s) gp->sn2_eff[s] = 1.0 / (sW1[s] * sW1[s]);
    gp->pooled = true; gp->in_views = true; gp->hasL = true;
    gp->blk_in = f.dIn.p; f.dIn.p = nullptr;
    gp->X = f.dX.as<double>(); gp->hyp = f.dhyp.as<double>(); gp->d_lchol = f.dlch.as<unsigned char>();
    double* e = f.dExtra.as<double>();
    gp->gpc = e; gp->d_sn2 = e + nG; gp->d_meanX = e + nG + S; gp->d_mult = e + nG + S + D;
    gp->alpha = dal.as<double>(); dal.p = nullptr;
    gp->L = dA.as<double>(); dA.p = nullptr;
    gp->d_finv = dfinv.as<double>(); dfinv.p = nullptr;
    for (int i = 0; i < 3; ++i) gp->noisefun[i] = noisefun[i];
    gp->has_noise = true;
    *gp_out = gp;
  } else if (gp_out) {
    // the device-resident posterior is assembled from the device buffers (no round trip of the S N x N matrices)
    vbmc_status st2 = gp_upload_impl(ctx, N, D, S, Nhyp, Ncov, Nnoise, meanfun, X, hyp, alh, nullptr, dA.as<double>(),
                                     any_inv ? dXi.as<double>() : nullptr, sW1.data(), lch.data(), gp_out);
    if (st2 != VBMC_OK) return st2;
    st2 = vbmc_gp_set_noise(ctx, *gp_out, noisefun, mult.data());
    if (st2 != VBMC_OK) return st2;
  }
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
namespace {
// gplite_noisefun.m:164-210, gradient part, on the host: dsn2 as Nnoise x N (constant models replicated over n)
void noise_grad(const int32_t nf[3], const double* hn, int N, const double* y, const double* s2, int Nnoise, double* dsn2) {
  std::fill(dsn2, dsn2 + (size_t)Nnoise * N, 0.0);
  int idx = 0;
  if (nf[0] == 1) { const double v = 2.0 * std::exp(2.0 * hn[idx]); for (int n = 0; n < N; ++n) dsn2[(size_t)idx * N + n] = v; idx++; }
  if (nf[1] == 2) { const double c = std::exp(hn[idx]); for (int n = 0; n < N; ++n) dsn2[(size_t)idx * N + n] = s2 ? c * s2[n] : 0.0; idx++; }
  if (nf[2] == 1 && y) {
    const double ythr = hn[idx], w2 = std::exp(2.0 * hn[idx + 1]);
    for (int n = 0; n < N; ++n) {
      const double zz = std::max(0.0, ythr - y[n]);
      dsn2[(size_t)idx * N + n] = zz > 0 ? 2.0 * w2 * (ythr - y[n]) : 0.0;
      dsn2[(size_t)(idx + 1) * N + n] = 2.0 * w2 * zz * zz;
    }
  }
}
}  // namespace

static vbmc_status gp_nlz_impl(vbmc_ctx* ctx, int N, int D, int B, int Nhyp, int meanfun, const int32_t noisefun[3], const double* X,
                               const double* y, const double* s2, const double* hyp, int compute_grad, double* nlZ, double* dnlZ,
                               bool optimistic);
extern "C" vbmc_status vbmc_gp_nlz(vbmc_ctx* ctx, int N, int D, int B, int Nhyp, int meanfun, const int32_t noisefun[3],
                                   const double* X, const double* y, const double* s2, const double* hyp, int compute_grad,
                                   double* nlZ, double* dnlZ) {
  vbmc_status st = gp_nlz_impl(ctx, N, D, B, Nhyp, meanfun, noisefun, X, y, s2, hyp, compute_grad, nlZ, dnlZ, true);
  if (st == VBMC_INTERNAL_RETRY) st = gp_nlz_impl(ctx, N, D, B, Nhyp, meanfun, noisefun, X, y, s2, hyp, compute_grad, nlZ, dnlZ, false);
  return st;
}

static vbmc_status gp_nlz_impl(vbmc_ctx* ctx, int N, int D, int B, int Nhyp, int meanfun, const int32_t noisefun[3], const double* X,
                               const double* y, const double* s2, const double* hyp, int compute_grad, double* nlZ, double* dnlZ,
                               bool optimistic) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!nlZ || (compute_grad && !dnlZ)) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_gp_nlz: null output");
  {
    // three B x N x N work matrices: keep each below 2 GiB by cutting the batch
    const size_t per = (size_t)N * N * 8;
    const int CHB = (int)std::max<size_t>(1, ((size_t)2 << 30) / std::max<size_t>(per, 1));
    if (N > 0 && B > CHB && hyp) {
      for (int b0 = 0; b0 < B; b0 += CHB) {
        const int nb = std::min(CHB, B - b0);
        vbmc_status st_ = vbmc_gp_nlz(ctx, N, D, nb, Nhyp, meanfun, noisefun, X, y, s2, hyp + (size_t)b0 * Nhyp, compute_grad, nlZ + b0,
                                      compute_grad ? dnlZ + (size_t)b0 * Nhyp : nullptr);
        if (st_ != VBMC_OK) return st_;
      }
      return VBMC_OK;
    }
  }
  GpFactor f;
  const int NnoiseH = noise_nhyp(noisefun);
  // the noise-model derivatives (host, O(B Nnoise N)) ride in the packed upload
  const size_t nds = compute_grad ? (size_t)B * std::max(NnoiseH, 1) * N : 0;
  if (compute_grad && N > 0 && B > 0 && hyp && y) {
    f.extra_in = nds;
    f.fill_extra = [&](double* e) {
      for (int b = 0; b < B; ++b)
        noise_grad(noisefun, hyp + (size_t)b * Nhyp + D + 1, N, y, s2, NnoiseH, e + (size_t)b * NnoiseH * N);
    };
  }
  // results: [nlZ B | failure indices B | dnlZ B x Nhyp], one block on the device, one copy back
  const size_t nout = (size_t)B * (2 + (compute_grad ? Nhyp : 0));
  // few matrices of moderate order with a gradient: the factor's inverse and alpha's backward solve share one launch
  const bool combined = compute_grad && N > 0 && N <= ASOLVE1_THREADS && (size_t)B * TRSM_NBLK(N) <= 1024 && tri_inverse2_fits(N);
  f.defer_alpha = combined;
  { vbmc_status s_ = gp_factorize(ctx, "vbmc_gp_nlz", N, D, B, Nhyp, meanfun, noisefun, X, y, s2, hyp, false, f, nout, optimistic, compute_grad && B <= 16); if (s_ != VBMC_OK) return s_; }
  hipStream_t st = ctx->stream;
  const int Nnoise = f.Nnoise, Nmean = f.Nmean;
  TmpBuf dout, dKi, dpart, dTT;
  // the closing kernel writes the (small) block of results straight into the pinned block over the host link: no copy behind it
  const bool out_direct = copy_by_kernel(nout * 8);
  if (!out_direct) HIP_TRY(ctx, dout.alloc(ctx, nout * 8));
  double* dnlz = out_direct ? f.pin_out : dout.as<double>();
  const int nt1 = (N + NLZ_T - 1) / NLZ_T, ntile = nt1 * nt1, P = D + 1 + Nnoise;
  if (compute_grad) {
    // Kinv*sl = L\(L'\eye(N)) for every hyper-parameter vector (:240) as T'T with T = inv(L'): one triangular solve of the
    // identity (k_tri_inverse2 / k_tri_inverse, from the column block's own rows down) and a rank-k update on the matrix cores
    // (k_syrk_tt); only the upper triangle is formed -- the part k_nlz_grad reads.  Neither needs alpha: `combined` runs alpha's
    // solve as one more workgroup of the inverse's launch; otherwise, with alpha on the second stream (f.alpha_event), they run
    // beside its solve and the streams join below.
    HIP_TRY(ctx, dKi.alloc(ctx, (size_t)B * N * N * 8));
    HIP_TRY(ctx, dTT.alloc(ctx, (size_t)B * N * N * 8));
    if (combined) {
      const int nblk = TRSM_NBLK(N);
      if (nblk <= TRI2_W * 4)
        hipLaunchKernelGGL((k_tri_inverse2_alpha<4>), dim3(nblk + 1, B), dim3(64 * TRI2_W), 0, st, N, f.dA.as<double>(), f.dfinv.as<double>(),
                           f.dones.as<unsigned char>(), dTT.as<double>(), f.dz.as<double>(), f.dal.as<double>(), f.dscal.as<double>());
      else
        hipLaunchKernelGGL((k_tri_inverse2_alpha<8>), dim3(nblk + 1, B), dim3(64 * TRI2_W), 0, st, N, f.dA.as<double>(), f.dfinv.as<double>(),
                           f.dones.as<unsigned char>(), dTT.as<double>(), f.dz.as<double>(), f.dal.as<double>(), f.dscal.as<double>());
    } else {
      HIP_TRY(ctx, tri_inverse_launch(st, N, B, f.dA.as<double>(), f.dfinv.as<double>(), f.dones.as<unsigned char>(), dTT.as<double>(), 1));
    }
    syrk_tt_launch(st, N, B, dTT.as<double>(), f.dones.as<unsigned char>(), dKi.as<double>());
    if (f.alpha_event) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    HIP_TRY(ctx, dpart.alloc(ctx, (size_t)B * ntile * P * 8));
    DISPATCH_GPDT(D, hipLaunchKernelGGL((k_nlz_grad<DT>), dim3(nt1, nt1, B), dim3(256), 0, st, N, D, Nhyp, Nnoise, f.dhyp.as<double>(),
                                        f.dXc.as<double>(), f.daa.as<double>(), dKi.as<double>(), f.dal.as<double>(), f.dscal.as<double>(),
                                        f.dExtra.as<double>(), dpart.as<double>()));
  }
  DISPATCH_GPDT(D, hipLaunchKernelGGL((k_nlz_final<DT>), dim3(B), dim3(256), 0, st, N, D, Nhyp, Nnoise, Nmean, meanfun, ntile, compute_grad ? 1 : 0,
                                      f.dX.as<double>(), f.dy.as<double>(), f.dhyp.as<double>(), f.dA.as<double>(), f.dal.as<double>(),
                                      f.dscal.as<double>(), compute_grad ? dpart.as<double>() : (const double*)nullptr,
                                      (const double*)f.d_pfd, dnlz));
  HIP_TRY(ctx, hipGetLastError());
  if (!out_direct) HIP_TRY(ctx, hipMemcpyAsync(f.pin_out, dnlz, nout * 8, hipMemcpyDeviceToHost, st));   // pinned: asynchronous
  f.h_pfd = f.pin_out + B;
  HIP_TRY(ctx, stream_wait_latency(st, N > 1024));
  if (!gp_factor_ok(f, B)) return VBMC_INTERNAL_RETRY;
  memcpy(nlZ, f.pin_out, (size_t)B * 8);
  if (compute_grad) memcpy(dnlZ, f.pin_out + 2 * (size_t)B, (size_t)B * Nhyp * 8);
  // a matrix still not positive definite after the retries: MATLAB errors downstream and the caller maps it to NaN
  // (gplite_train.m:542-546)
  const double qnan = std::numeric_limits<double>::quiet_NaN();
  for (int b = 0; b < B; ++b)
    if (f.failed[b]) {
      nlZ[b] = qnan;
      if (compute_grad) for (int i = 0; i < Nhyp; ++i) dnlZ[(size_t)b * Nhyp + i] = qnan;
    }
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_gp_set_noise(vbmc_ctx* ctx, vbmc_gp* gp, const int32_t noisefun[3], const double* sn2_mult) {
  if (!ctx || !gp || !noisefun || !sn2_mult) return VBMC_ERR_INVALID;
  if (gp->d_mult) ctx_drain_slots(ctx);    // a pass in flight on a slot stream may be reading the multipliers about to be overwritten
  for (int i = 0; i < 3; ++i) gp->noisefun[i] = noisefun[i];
  if (!gp->d_mult) HIP_TRY(ctx, gp->pooled ? pool_get(ctx, (size_t)gp->S * 8, (void**)&gp->d_mult) : hipMalloc((void**)&gp->d_mult, (size_t)gp->S * 8));
  HIP_TRY(ctx, hipMemcpy(gp->d_mult, sn2_mult, (size_t)gp->S * 8, hipMemcpyHostToDevice));
  gp->has_noise = true;
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
namespace {
struct PredBufs {
  TmpBuf dXs, ds2, dys, dmb, dout, dXc, daa, dmuv, dgrp, dpV, dpF, dKs;
  double *fmu = nullptr, *fs2 = nullptr, *ys2 = nullptr;   // Nstar x S each, inside dout
};

// gplite_pred for every hyper-sample, results left on the device (shared by vbmc_gp_pred and vbmc_acq_eval)
// want_ks: the caller reads the sW-scaled cross-kernel matrix itself (the IQR acquisition functions); otherwise the variance comes from
// k_pred_fused and that matrix is never written (round 6).  VBMC_PRED_FUSED=0 keeps the two-kernel form (A/B runs, tests).
vbmc_status pred_on_device(vbmc_ctx* ctx, const char* who, const vbmc_gp* gp, int Nstar, const double* Xstar, const double* ystar,
                           const double* s2star, PredBufs& pb, bool want_ks = false) {
  if (!gp || Nstar <= 0 || !Xstar) return set_err(ctx, VBMC_ERR_INVALID, "%s: bad arguments", who);
  if (!gp->hasL) return set_err(ctx, VBMC_ERR_INVALID, "%s needs gp.post(s).L on the device", who);
  if (!gp->has_noise) return set_err(ctx, VBMC_ERR_INVALID, "%s: call vbmc_gp_set_noise (noisefun, sn2_mult) first", who);
  // output-dependent noise (noisefun(3) = 1) enters ys2 only, and only with a non-empty ystar (gplite_noisefun.m:198-207);
  // fmu / fs2 -- all the acquisition functions read (acqwrapper_vbmc.m:17) -- never depend on it
  // an empty s2star counts as zero (gplite_noisefun.m:51): the kernels add nothing when the pointer is null
  const int N = gp->N, D = gp->D, S = gp->S;
  if (D > VBMC_LIM_D) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "D = %d > %d not accelerated", D, VBMC_LIM_D);
  const int Np = ((N + 15) >> 4) << 4, nblk = Np >> 4;
  // beyond N = 1248 a 16-row tile of inv(L') no longer fits the LDS: the variance then comes from slab solves (k_pred_slab)
  const bool slab_pred = (size_t)16 * Np * 8 > PRED_LDS_MAX || nblk > PRED_MAXG || trsm_cw_for(N) != 16;
  if (trsm_cw_for(N) == 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "N = %d too large for the prediction kernels", N);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (!slab_pred) {   // Tinv = inv(L') = L' \ I for the Lchol samples, once per GP (the kernels skip the others)
    bool have = false;
    vbmc_status s_ = ensure_tinv(ctx, gp, &have);
    if (s_) return s_;
  }
  // column means for sq_dist's centring (sq_dist.m:36), O((N + Nstar) D) on the host in MATLAB's order
  std::vector<double> mb(D);
  for (int d = 0; d < D; ++d) {
    double sb = 0.0;
    for (int j = 0; j < Nstar; ++j) sb += Xstar[j + (size_t)Nstar * d];
    mb[d] = sb / Nstar;
  }
  TmpBuf &dXs = pb.dXs, &ds2 = pb.ds2, &dmb = pb.dmb, &dout = pb.dout, &dXc = pb.dXc, &daa = pb.daa, &dmuv = pb.dmuv;
  HIP_TRY(ctx, dXc.alloc(ctx, (size_t)S * N * D * 8));
  HIP_TRY(ctx, daa.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, dmuv.alloc(ctx, (size_t)S * 2 * D * 8));
  HIP_TRY(ctx, dXs.alloc(ctx, (size_t)Nstar * D * 8));
  HIP_TRY(ctx, dmb.alloc(ctx, (size_t)D * 8));
  HIP_TRY(ctx, dout.alloc(ctx, (size_t)3 * Nstar * S * 8));
  HIP_TRY(ctx, hipMemcpyAsync(dXs.p, Xstar, (size_t)Nstar * D * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dmb.p, mb.data(), (size_t)D * 8, hipMemcpyHostToDevice, st));
  if (s2star) {
    HIP_TRY(ctx, ds2.alloc(ctx, (size_t)Nstar * 8));
    HIP_TRY(ctx, hipMemcpyAsync(ds2.p, s2star, (size_t)Nstar * 8, hipMemcpyHostToDevice, st));
  }
  if (ystar && gp->noisefun[2] == 1) {
    HIP_TRY(ctx, pb.dys.alloc(ctx, (size_t)Nstar * 8));
    HIP_TRY(ctx, hipMemcpyAsync(pb.dys.p, ystar, (size_t)Nstar * 8, hipMemcpyHostToDevice, st));
  }
  PredArgs pa{};
  pa.N = N; pa.D = D; pa.S = S; pa.Nhyp = gp->Nhyp; pa.Nstar = Nstar; pa.meanfun = gp->meanfun;
  pa.moff = gp->Ncov + gp->Nnoise; pa.noff = gp->Ncov; pa.nf0 = gp->noisefun[0]; pa.nf1 = gp->noisefun[1]; pa.nf2 = gp->noisefun[2];
  pa.X = gp->X; pa.Xs = dXs.as<double>(); pa.s2s = s2star ? ds2.as<double>() : nullptr; pa.hyp = gp->hyp;
  pa.ys = (ystar && gp->noisefun[2] == 1) ? pb.dys.as<double>() : nullptr;
  pa.alpha = gp->alpha; pa.L = gp->L; pa.sn2_eff = gp->d_sn2; pa.sn2_mult = gp->d_mult; pa.lchol = gp->d_lchol;
  pa.mean_a = gp->d_meanX; pa.mean_b = dmb.as<double>(); pa.finv = gp->d_finv; pa.tinv = gp->d_tinv;
  pa.fmu = dout.as<double>(); pa.fs2 = pa.fmu + (size_t)Nstar * S; pa.ys2 = pa.fs2 + (size_t)Nstar * S;
  pb.fmu = pa.fmu; pb.fs2 = pa.fs2; pb.ys2 = pa.ys2;
  hipLaunchKernelGGL(k_pred_prep, dim3(4, S), dim3(256), 0, st, pa, dXc.as<double>(), daa.as<double>(), dmuv.as<double>());
  // resident row blocks of Tinv per hyper-sample: consecutive 16-row tiles, balanced by area (tile b costs b + 1),
  // bounded by PRED_MAXR tiles and PRED_LDS_MAX bytes of LDS (two workgroups per CU)
  std::vector<int> grp((size_t)S + 2 * (size_t)S * PRED_MAXG, 0);
  int maxg = 0;
  size_t maxlds = 0;
  for (int s = 0; s < S; ++s) {
    int ng = 0;
    if (slab_pred) {
      ng = 1;                                  // k_pred_slab writes block partial 0
    } else if (gp->Lchol[s]) {
      int t0 = 0;
      for (int b = 0; b < nblk; ++b) {
        const int R = b - t0 + 1;
        const size_t lds_need = (size_t)R * 16 * (size_t)(b + 1) * 16 * 8;
        if (b > t0 && (R > PRED_MAXR || lds_need > PRED_LDS_MAX)) {
          grp[S + (size_t)s * PRED_MAXG + ng] = t0; grp[S + (size_t)S * PRED_MAXG + (size_t)s * PRED_MAXG + ng] = b; ++ng;
          maxlds = std::max(maxlds, (size_t)(b - t0) * 16 * (size_t)b * 16 * 8);
          t0 = b;
        }
      }
      grp[S + (size_t)s * PRED_MAXG + ng] = t0; grp[S + (size_t)S * PRED_MAXG + (size_t)s * PRED_MAXG + ng] = nblk; ++ng;
      maxlds = std::max(maxlds, (size_t)(nblk - t0) * 16 * (size_t)nblk * 16 * 8);
    } else {
      for (int b = 0; b < nblk; ++b) { grp[S + (size_t)s * PRED_MAXG + ng] = b; grp[S + (size_t)S * PRED_MAXG + (size_t)s * PRED_MAXG + ng] = b + 1; ++ng; }
      maxlds = std::max(maxlds, (size_t)16 * Np * 8);
    }
    grp[s] = ng;
    maxg = std::max(maxg, ng);
  }
  // one workgroup per CU (its LDS is full): slice the point tiles over gridDim.z until the chip is covered
  const int ntile_ = (Nstar + 15) / 16;
  int PZ = std::max(1, ctx->num_cu / std::max(1, maxg * S));
  PZ = std::min(PZ, std::max(1, ntile_ / (PRED_THREADS / 64)));
  static const bool fused_off = [] { const char* e = getenv("VBMC_PRED_FUSED"); return e && !strcmp(e, "0"); }();
  // point tiles resident per workgroup: as many N x 16 tiles as the 160 KB hold beside the 2 KB table (three at N = 400)
  const int fused_pt = (int)std::min<size_t>(PREDF_PT_FOR_QS((D + 3) / 4), ((size_t)160 * 1024 - 2048 - (PREDF_THREADS / 64) * PREDF_MAXPT * 16 * 8 - 256) / ((size_t)Np * 16 * 8));
  const bool fused = !want_ks && !slab_pred && !fused_off && fused_pt >= 1 && (D + 3) / 4 <= 8;
  // few points (the importance sampler's ~110 per call): RS workgroups per (hyper-sample, pass) unit share its row tiles (k_pred_fused)
  int fused_rs = 1;
  if (fused) {
    const int units = ((ntile_ + fused_pt - 1) / fused_pt) * S;
    fused_rs = std::max(1, std::min(std::min(4, nblk / 4), ctx->num_cu / std::max(1, units)));
    for (int s = 0; s < S; ++s) grp[s] = fused_rs;      // k_pred_final: fused_rs blocks of partial sums per hyper-sample
    maxg = fused_rs;
  }
  HIP_TRY(ctx, pb.dgrp.alloc(ctx, grp.size() * sizeof(int)));
  HIP_TRY(ctx, pb.dpV.alloc(ctx, (size_t)maxg * S * Nstar * 8));
  HIP_TRY(ctx, pb.dpF.alloc(ctx, (size_t)S * Nstar * 8));
  HIP_TRY(ctx, hipMemcpyAsync(pb.dgrp.p, grp.data(), grp.size() * sizeof(int), hipMemcpyHostToDevice, st));
  if (fused) {
    const size_t fl = (size_t)fused_pt * Np * 16 * 8;
    const int npass = (ntile_ + fused_pt - 1) / fused_pt;
    // one workgroup per compute unit (its LDS is full), each walking the (hyper-sample, pass) units b, b + grid, ...
    const int gxf = std::max(1, std::min(npass * S * fused_rs, ctx->num_cu));
#define PRED_FUSED_PT(QSV, PTV) { \
      if (fl > 64 * 1024) HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_pred_fused<QSV, PTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl)); \
      hipLaunchKernelGGL((k_pred_fused<QSV, PTV>), dim3(gxf), dim3(PREDF_THREADS), fl, st, pa, dXc.as<double>(), daa.as<double>(), \
                         dmuv.as<double>(), pb.dpV.as<double>(), pb.dpF.as<double>(), fused_rs); }
#define PRED_FUSED(QSV) case QSV: \
      if (fused_pt >= 3) { if constexpr (PREDF_PT_FOR_QS(QSV) >= 3) PRED_FUSED_PT(QSV, 3) } \
      else if (fused_pt == 2) { if constexpr (PREDF_PT_FOR_QS(QSV) >= 2) PRED_FUSED_PT(QSV, 2) } \
      else PRED_FUSED_PT(QSV, 1) \
      break;
    switch ((D + 3) / 4) { PRED_FUSED(1) PRED_FUSED(2) PRED_FUSED(3) PRED_FUSED(4) PRED_FUSED(5) PRED_FUSED(6) PRED_FUSED(7) PRED_FUSED(8) default: break; }
#undef PRED_FUSED_PT
#undef PRED_FUSED
    hipLaunchKernelGGL(k_pred_final, dim3((Nstar + 255) / 256, S), dim3(256), 0, st, pa, pb.dgrp.as<int>(), pb.dpV.as<double>(), pb.dpF.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    return VBMC_OK;
  }
  HIP_TRY(ctx, pb.dKs.alloc(ctx, (size_t)S * N * (((size_t)Nstar + 15) / 16) * 16 * 8));   // tiled by 16 points
  // inner dimension of the MFMA distance blocks: QS = ceil(D / 4) steps
#define PRED_KS(QSV) case QSV: hipLaunchKernelGGL((k_pred_ks<QSV>), dim3((Nstar + 15) / 16, S), dim3(64), 0, st, pa, dXc.as<double>(), \
                                                  daa.as<double>(), dmuv.as<double>(), pb.dKs.as<double>(), pb.dpF.as<double>()); break;
  switch ((D + 3) / 4) { PRED_KS(1) PRED_KS(2) PRED_KS(3) PRED_KS(4) PRED_KS(5) PRED_KS(6) PRED_KS(7) PRED_KS(8) default: break; }
#undef PRED_KS
  if (slab_pred) {
    TRSM_DISPATCH_CW(trsm_cw_for(N), {
      const size_t sl = TRSM_LDS_BYTES_CW(N, CW);
      if (sl > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_pred_slab<CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sl));
      hipLaunchKernelGGL((k_pred_slab<CW>), dim3((Nstar + CW - 1) / CW, S), dim3(64), sl, st, pa, pb.dKs.as<double>(), pb.dpV.as<double>());
    });
  } else {
    if (maxlds > 64 * 1024)
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_gp_pred, hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxlds));
    hipLaunchKernelGGL(k_gp_pred, dim3(maxg, S, PZ), dim3(PRED_THREADS), maxlds, st, pa, pb.dKs.as<double>(), pb.dgrp.as<int>(), pb.dpV.as<double>());
  }
  hipLaunchKernelGGL(k_pred_final, dim3((Nstar + 255) / 256, S), dim3(256), 0, st, pa, pb.dgrp.as<int>(), pb.dpV.as<double>(), pb.dpF.as<double>());
  HIP_TRY(ctx, hipGetLastError());
  return VBMC_OK;
}
}  // namespace

namespace {
// The prediction pipeline materialises S x N x Nstar cross-kernel values; very large sweeps are cut into chunks of
// test points whose matrix stays below 1 GiB (the sq_dist centring constant then differs per chunk: rounding only).
int pred_chunk_points(const vbmc_gp* gp, int Nstar) {
  if (!gp) return Nstar;
  const size_t per_point = (size_t)gp->S * gp->N * 8;
  const size_t cap = ((size_t)1 << 30) / std::max<size_t>(per_point, 1);
  const size_t ch = std::max<size_t>(256, (cap / 16) * 16);
  return (size_t)Nstar <= ch ? Nstar : (int)ch;
}
// rows [i0, i0 + n) of a column-major Ntot x C matrix -> contiguous n x C
void gather_rows(const double* src, int Ntot, int C, int i0, int n, std::vector<double>& dst) {
  dst.resize((size_t)n * C);
  for (int c = 0; c < C; ++c) memcpy(dst.data() + (size_t)c * n, src + (size_t)c * Ntot + i0, (size_t)n * 8);
}
void scatter_rows(const std::vector<double>& src, int Ntot, int C, int i0, int n, double* dst) {
  if (!dst) return;
  for (int c = 0; c < C; ++c) memcpy(dst + (size_t)c * Ntot + i0, src.data() + (size_t)c * n, (size_t)n * 8);
}
}  // namespace

extern "C" vbmc_status vbmc_gp_pred(vbmc_ctx* ctx, const vbmc_gp* gp, int Nstar, const double* Xstar, const double* ystar,
                                    const double* s2star, int ssflag, double* ymu, double* ys2, double* fmu, double* fs2) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (gp && Xstar && Nstar > pred_chunk_points(gp, Nstar)) {
    const int CH = pred_chunk_points(gp, Nstar), D = gp->D;
    const int nc = (ssflag && gp->S > 1) ? gp->S : 1;
    std::vector<double> xs, s2c, ysc, o[4];
    for (int i0 = 0; i0 < Nstar; i0 += CH) {
      const int n = std::min(CH, Nstar - i0);
      gather_rows(Xstar, Nstar, D, i0, n, xs);
      if (s2star) s2c.assign(s2star + i0, s2star + i0 + n);
      if (ystar) ysc.assign(ystar + i0, ystar + i0 + n);
      for (auto& v : o) v.assign((size_t)n * nc, 0.0);
      vbmc_status st_ = vbmc_gp_pred(ctx, gp, n, xs.data(), ystar ? ysc.data() : nullptr, s2star ? s2c.data() : nullptr, ssflag, o[0].data(), o[1].data(), o[2].data(), o[3].data());
      if (st_ != VBMC_OK) return st_;
      scatter_rows(o[0], Nstar, nc, i0, n, ymu); scatter_rows(o[1], Nstar, nc, i0, n, ys2);
      scatter_rows(o[2], Nstar, nc, i0, n, fmu); scatter_rows(o[3], Nstar, nc, i0, n, fs2);
    }
    return VBMC_OK;
  }
  PredBufs pb;
  { vbmc_status s_ = pred_on_device(ctx, "vbmc_gp_pred", gp, Nstar, Xstar, ystar, s2star, pb); if (s_ != VBMC_OK) return s_; }
  hipStream_t st = ctx->stream;
  const int S = gp->S;
  TmpBuf davg;
  TmpBuf& dout = pb.dout;
  struct { double *fmu, *fs2, *ys2; } pa{pb.fmu, pb.fs2, pb.ys2};
  const size_t ns = (size_t)Nstar * S;
  if (S > 1 && !ssflag) {
    HIP_TRY(ctx, davg.alloc(ctx, (size_t)4 * Nstar * 8));
    hipLaunchKernelGGL(k_pred_avg, dim3((Nstar + 255) / 256), dim3(256), 0, st, Nstar, S, pa.fmu, pa.fs2, pa.ys2, davg.as<double>());
    std::vector<double> h((size_t)4 * Nstar);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), davg.p, h.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ymu) memcpy(ymu, h.data(), (size_t)Nstar * 8);
    if (ys2) memcpy(ys2, h.data() + Nstar, (size_t)Nstar * 8);
    if (fmu) memcpy(fmu, h.data() + 2 * (size_t)Nstar, (size_t)Nstar * 8);
    if (fs2) memcpy(fs2, h.data() + 3 * (size_t)Nstar, (size_t)Nstar * 8);
  } else {
    std::vector<double> h(3 * ns);
    HIP_TRY(ctx, hipMemcpyAsync(h.data(), dout.p, h.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (fmu) memcpy(fmu, h.data(), ns * 8);
    if (ymu) memcpy(ymu, h.data(), ns * 8);
    if (fs2) memcpy(fs2, h.data() + ns, ns * 8);
    if (ys2) memcpy(ys2, h.data() + 2 * ns, ns * 8);
  }
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" vbmc_status vbmc_acq_eval(vbmc_ctx* ctx, const vbmc_gp* gp, int Nstar, const double* Xs, int acq_id, int K,
                                     const double* vp_mu, const double* vp_sigma, const double* vp_lambda, const double* vp_w,
                                     double ymax, int var_regularized, double TolGPVar, const double* gplengthscale,
                                     const double* X_rescaled, const double* sn2new, double* acq, double* fbar, double* vtot) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!acq || K <= 0 || !vp_mu || !vp_sigma || !vp_lambda || !vp_w) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_eval: bad arguments");
  if (acq_id < 0 || acq_id > 3) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "acquisition function id %d not accelerated (0 acqf, 1 acqflog, 2 acqus, 3 acqfsn2)", acq_id);
  if (acq_id == 3 && (!gplengthscale || !X_rescaled || !sn2new))
    return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_eval: acqfsn2 needs gplengthscale, X_rescaled and sn2new");
  if (gp && Xs && Nstar > pred_chunk_points(gp, Nstar)) {
    const int CH = pred_chunk_points(gp, Nstar);
    std::vector<double> xs, o[3];
    for (int i0 = 0; i0 < Nstar; i0 += CH) {
      const int n = std::min(CH, Nstar - i0);
      gather_rows(Xs, Nstar, gp->D, i0, n, xs);
      for (auto& v : o) v.assign(n, 0.0);
      vbmc_status st_ = vbmc_acq_eval(ctx, gp, n, xs.data(), acq_id, K, vp_mu, vp_sigma, vp_lambda, vp_w, ymax, var_regularized, TolGPVar,
                                      gplengthscale, X_rescaled, sn2new, o[0].data(), o[1].data(), o[2].data());
      if (st_ != VBMC_OK) return st_;
      scatter_rows(o[0], Nstar, 1, i0, n, acq); scatter_rows(o[1], Nstar, 1, i0, n, fbar); scatter_rows(o[2], Nstar, 1, i0, n, vtot);
    }
    return VBMC_OK;
  }
  PredBufs pb;
  { vbmc_status s_ = pred_on_device(ctx, "vbmc_acq_eval", gp, Nstar, Xs, nullptr, nullptr, pb); if (s_ != VBMC_OK) return s_; }
  hipStream_t st = ctx->stream;
  const int N = gp->N, D = gp->D, S = gp->S;
  if ((size_t)(2 * K * D + K) * 8 > 64 * 1024) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "vbmc_acq_eval: K*D = %d too large", K * D);
  // host: O(K D) constants of vbmc_pdf.m:57-62 in the reference's order of operations
  double prodl = 1.0;
  for (int d = 0; d < D; ++d) prodl *= vp_lambda[d];
  const double nf = 1.0 / std::pow(2.0 * 3.14159265358979323846, D / 2.0) / prodl;
  std::vector<double> hb((size_t)K * D + K);
  for (int k = 0; k < K; ++k) {
    for (int d = 0; d < D; ++d) hb[(size_t)k * D + d] = 1.0 / (vp_sigma[k] * vp_lambda[d]);
    hb[(size_t)K * D + k] = nf * vp_w[k] / std::pow(vp_sigma[k], D);
  }
  TmpBuf dmu, dhb, dgl, dXr, dsn, dsx, dres;
  HIP_TRY(ctx, dmu.alloc(ctx, (size_t)D * K * 8));
  HIP_TRY(ctx, dhb.alloc(ctx, hb.size() * 8));
  HIP_TRY(ctx, dres.alloc(ctx, (size_t)3 * Nstar * 8));
  HIP_TRY(ctx, hipMemcpyAsync(dmu.p, vp_mu, (size_t)D * K * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dhb.p, hb.data(), hb.size() * 8, hipMemcpyHostToDevice, st));
  AcqArgs a{};
  a.Nstar = Nstar; a.S = S; a.D = D; a.K = K; a.N = N; a.acq_id = acq_id; a.reg = var_regularized ? 1 : 0;
  a.ymax = ymax; a.TolVar = TolGPVar;
  a.Xs = pb.dXs.as<double>(); a.fmu = pb.fmu; a.fs2 = pb.fs2;
  a.mu = dmu.as<double>(); a.isl = dhb.as<double>(); a.coef = dhb.as<double>() + (size_t)K * D;
  if (acq_id == 3) {
    HIP_TRY(ctx, dgl.alloc(ctx, (size_t)D * 8));
    HIP_TRY(ctx, dXr.alloc(ctx, (size_t)N * D * 8));
    HIP_TRY(ctx, dsn.alloc(ctx, (size_t)N * 8));
    HIP_TRY(ctx, hipMemcpyAsync(dgl.p, gplengthscale, (size_t)D * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dXr.p, X_rescaled, (size_t)N * D * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dsn.p, sn2new, (size_t)N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, dsx.alloc(ctx, (size_t)Nstar * 8));
    switch ((D + 3) / 4) {
#define NN_CASE(QSV) case QSV: hipLaunchKernelGGL((k_nn_noise<QSV>), dim3((Nstar + 15) / 16), dim3(64), 0, st, Nstar, N, D, pb.dXs.as<double>(), \
                                                 dgl.as<double>(), dXr.as<double>(), dsn.as<double>(), dsx.as<double>()); break;
      NN_CASE(1) NN_CASE(2) NN_CASE(3) NN_CASE(4) NN_CASE(5) NN_CASE(6) NN_CASE(7) NN_CASE(8)
#undef NN_CASE
      default: break;
    }
    a.sn2x = dsx.as<double>();
  }
  a.acq = dres.as<double>(); a.fbar = a.acq + Nstar; a.vtot = a.fbar + Nstar;
  hipLaunchKernelGGL(k_acq, dim3((Nstar + 255) / 256), dim3(256), (size_t)(2 * K * D + K) * 8, st, a);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<double> h((size_t)3 * Nstar);
  HIP_TRY(ctx, hipMemcpyAsync(h.data(), dres.p, h.size() * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  memcpy(acq, h.data(), (size_t)Nstar * 8);
  if (fbar) memcpy(fbar, h.data() + Nstar, (size_t)Nstar * 8);
  if (vtot) memcpy(vtot, h.data() + 2 * (size_t)Nstar, (size_t)Nstar * 8);
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
// Importance-sampling state of the IQR acquisition functions (optimState.ActiveImportanceSampling)
struct vbmc_acq_is {
  int Na = 0, Nap = 0, per_s = 0, S = 0, N = 0, D = 0;
  bool has_lnw = false;
  double *Xa = nullptr, *CT = nullptr, *fs2a = nullptr, *lnw = nullptr;
};

extern "C" void vbmc_acq_is_free(vbmc_ctx* ctx, vbmc_acq_is* h) {
  if (!h) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  for (double* p : {h->Xa, h->CT, h->fs2a, h->lnw})
    if (p) (void)hipFree(p);
  delete h;
}

extern "C" vbmc_status vbmc_acq_is_create(vbmc_ctx* ctx, const vbmc_gp* gp, int Na, const double* Xa, int per_sample_inputs,
                                          const double* lnw, const double* fs2a, const double* Ctmp, vbmc_acq_is** out) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (out) *out = nullptr;
  if (!gp || !out || Na <= 0 || !Xa) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_is_create: bad arguments");
  if (!gp->hasL) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_is_create needs gp.post(s).L on the device");
  if (per_sample_inputs && !fs2a)
    return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_is_create: per-hyper-sample importance points need fs2a from the caller");
  const int N = gp->N, D = gp->D, S = gp->S;
  const int Nap = ((Na + 15) / 16) * 16;
  if (Nap > VBMC_LIM_NA) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "Na = %d > %d importance points not accelerated", Na, VBMC_LIM_NA);
  if (trsm_cw_for(N) == 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "N = %d too large", N);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  vbmc_acq_is* h = new vbmc_acq_is();
  h->Na = Na; h->Nap = Nap; h->per_s = per_sample_inputs ? 1 : 0; h->S = S; h->N = N; h->D = D;
  const size_t nxa = (size_t)Na * D * (per_sample_inputs ? S : 1);
  auto fail = [&](vbmc_status st_) { vbmc_acq_is_free(ctx, h); return st_; };
#define IS_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(set_err(ctx, VBMC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_))); } while (0)
  IS_TRY(hipMalloc((void**)&h->Xa, nxa * 8));
  IS_TRY(hipMalloc((void**)&h->CT, (size_t)S * N * Nap * 8));
  IS_TRY(hipMalloc((void**)&h->fs2a, (size_t)S * Nap * 8));
  IS_TRY(hipMemcpyAsync(h->Xa, Xa, nxa * 8, hipMemcpyHostToDevice, st));
  // lnw (S x Na, column-major as in MATLAB) -> S x Nap rows, -inf in the padding
  std::vector<double> hb((size_t)S * Nap);
  if (lnw) {
    h->has_lnw = true;
    IS_TRY(hipMalloc((void**)&h->lnw, (size_t)S * Nap * 8));
    for (int s = 0; s < S; ++s)
      for (int a = 0; a < Nap; ++a) hb[(size_t)s * Nap + a] = a < Na ? lnw[s + (size_t)S * a] : -INFINITY;
    IS_TRY(hipMemcpyAsync(h->lnw, hb.data(), hb.size() * 8, hipMemcpyHostToDevice, st));
    IS_TRY(hipStreamSynchronize(st));
  }
  // fs2a (Na x S): given, or gplite_pred at the (shared) importance points
  std::vector<double> f2((size_t)S * Nap, 0.0);
  if (fs2a) {
    for (int s = 0; s < S; ++s)
      for (int a = 0; a < Na; ++a) f2[(size_t)s * Nap + a] = fs2a[a + (size_t)Na * s];
  } else {
    std::vector<double> tmp((size_t)Na * S);
    vbmc_status ps = vbmc_gp_pred(ctx, gp, Na, Xa, nullptr, nullptr, 1, nullptr, nullptr, nullptr, tmp.data());
    if (ps != VBMC_OK) return fail(ps);
    for (int s = 0; s < S; ++s)
      for (int a = 0; a < Na; ++a) f2[(size_t)s * Nap + a] = tmp[a + (size_t)Na * s];
  }
  IS_TRY(hipMemcpyAsync(h->fs2a, f2.data(), f2.size() * 8, hipMemcpyHostToDevice, st));
  // Ctmp (N x Na x S): given, or (L\(L'\Kax'))/sn2_eff | L*Kax' computed here (activeimportancesampling_vbmc.m:255-275)
  TmpBuf dZ, dU;
  IS_TRY(dZ.alloc(ctx, (size_t)S * N * Na * 8));
  IS_TRY(dU.alloc(ctx, (size_t)S * N * Na * 8));
  if (Ctmp) {
    IS_TRY(hipMemcpyAsync(dU.p, Ctmp, (size_t)S * N * Na * 8, hipMemcpyHostToDevice, st));
    // already scaled: pack with unit scale (flag array of zeros -> sc = 1)
    TmpBuf dzero;
    IS_TRY(dzero.alloc(ctx, S));
    IS_TRY(hipMemsetAsync(dzero.p, 0, S, st));
    hipLaunchKernelGGL(k_ctmp_pack, dim3(64, S), dim3(256), 0, st, N, Na, Nap, dU.as<double>(), gp->d_sn2, dzero.as<unsigned char>(), h->CT);
    IS_TRY(hipGetLastError());
    IS_TRY(hipStreamSynchronize(st));
  } else {
    hipLaunchKernelGGL(k_cross_kernel, dim3(64, S), dim3(256), 0, st, N, D, gp->Nhyp, Na, h->per_s, gp->X, h->Xa, gp->hyp, dZ.as<double>());
    hipLaunchKernelGGL(k_symm, dim3(64, S, 1), dim3(256), 0, st, N, Na, S, gp->L, gp->d_lchol, dZ.as<double>(), dU.as<double>());
    IS_TRY(trsm_fwd_launch(st, N, Na, S, 1, gp->L, gp->d_finv, gp->d_lchol, dZ.as<double>()));
    IS_TRY(trsm_bwd_launch(st, N, Na, S, 1, gp->L, gp->d_finv, gp->d_lchol, dZ.as<double>(), dU.as<double>()));
    hipLaunchKernelGGL(k_ctmp_pack, dim3(64, S), dim3(256), 0, st, N, Na, Nap, dU.as<double>(), gp->d_sn2, gp->d_lchol, h->CT);
    IS_TRY(hipGetLastError());
    IS_TRY(hipStreamSynchronize(st));
  }
#undef IS_TRY
  *out = h;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_acq_iqr_eval(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_acq_is* is, int Nstar, const double* Xs,
                                         const double* gplengthscale, const double* X_rescaled, const double* sn2new,
                                         int var_regularized, double TolGPVar, double* acq, double* fbar, double* vtot) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!is || !acq || !gplengthscale || !X_rescaled || !sn2new) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_iqr_eval: bad arguments");
  if (gp && (is->N != gp->N || is->S != gp->S || is->D != gp->D))
    return set_err(ctx, VBMC_ERR_INVALID, "vbmc_acq_iqr_eval: importance-sampling state belongs to a different GP");
  if (gp && Xs && Nstar > pred_chunk_points(gp, Nstar)) {
    const int CH = pred_chunk_points(gp, Nstar);
    std::vector<double> xs, o[3];
    for (int i0 = 0; i0 < Nstar; i0 += CH) {
      const int n = std::min(CH, Nstar - i0);
      gather_rows(Xs, Nstar, gp->D, i0, n, xs);
      for (auto& v : o) v.assign(n, 0.0);
      vbmc_status st_ = vbmc_acq_iqr_eval(ctx, gp, is, n, xs.data(), gplengthscale, X_rescaled, sn2new, var_regularized, TolGPVar,
                                          o[0].data(), o[1].data(), o[2].data());
      if (st_ != VBMC_OK) return st_;
      scatter_rows(o[0], Nstar, 1, i0, n, acq); scatter_rows(o[1], Nstar, 1, i0, n, fbar); scatter_rows(o[2], Nstar, 1, i0, n, vtot);
    }
    return VBMC_OK;
  }
  PredBufs pb;
  { vbmc_status s_ = pred_on_device(ctx, "vbmc_acq_iqr_eval", gp, Nstar, Xs, nullptr, nullptr, pb, true); if (s_ != VBMC_OK) return s_; }
  hipStream_t st = ctx->stream;
  const int N = gp->N, D = gp->D, S = gp->S;
  TmpBuf dgl, dXr, dsn, dsx, dacqs, dres;
  HIP_TRY(ctx, dgl.alloc(ctx, (size_t)D * 8));
  HIP_TRY(ctx, dXr.alloc(ctx, (size_t)N * D * 8));
  HIP_TRY(ctx, dsn.alloc(ctx, (size_t)N * 8));
  HIP_TRY(ctx, dsx.alloc(ctx, (size_t)Nstar * 8));
  HIP_TRY(ctx, dacqs.alloc(ctx, (size_t)Nstar * S * 8));
  HIP_TRY(ctx, dres.alloc(ctx, (size_t)3 * Nstar * 8));
  HIP_TRY(ctx, hipMemcpyAsync(dgl.p, gplengthscale, (size_t)D * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dXr.p, X_rescaled, (size_t)N * D * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(dsn.p, sn2new, (size_t)N * 8, hipMemcpyHostToDevice, st));
  switch ((D + 3) / 4) {
#define NN_CASE(QSV) case QSV: hipLaunchKernelGGL((k_nn_noise<QSV>), dim3((Nstar + 15) / 16), dim3(64), 0, st, Nstar, N, D, pb.dXs.as<double>(), \
                                                 dgl.as<double>(), dXr.as<double>(), dsn.as<double>(), dsx.as<double>()); break;
    NN_CASE(1) NN_CASE(2) NN_CASE(3) NN_CASE(4) NN_CASE(5) NN_CASE(6) NN_CASE(7) NN_CASE(8)
#undef NN_CASE
    default: break;
  }
  IqrArgs a{};
  a.N = N; a.D = D; a.S = S; a.Nhyp = gp->Nhyp; a.Nstar = Nstar; a.Na = is->Na; a.Nap = is->Nap; a.per_s = is->per_s;
  a.Xs = pb.dXs.as<double>(); a.Xa = is->Xa; a.hyp = gp->hyp; a.Xc = pb.dXc.as<double>(); a.muv = pb.dmuv.as<double>();
  a.CT = is->CT; a.fs2a = is->fs2a; a.lnw = is->has_lnw ? is->lnw : nullptr; a.fs2 = pb.fs2; a.sn2x = dsx.as<double>();
  a.lchol = gp->d_lchol; a.acqs = dacqs.as<double>(); a.KsW = pb.dKs.as<double>(); a.sn2_eff = gp->d_sn2;
  dim3 grid((Nstar + IQR_PTS - 1) / IQR_PTS, S);
  switch (is->Nap / 16) {
#define IQR_CASE(NT)                                                                                                                   \
  case NT:                                                                                                                             \
    if (IQR_LDS_BYTES(NT) > 64 * 1024)                                                                                                 \
      HIP_TRY(ctx, hipFuncSetAttribute((const void*)k_acq_iqr<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)IQR_LDS_BYTES(NT))); \
    hipLaunchKernelGGL((k_acq_iqr<NT>), grid, dim3(IQR_THREADS), IQR_LDS_BYTES(NT), st, a);                                                     \
    break;
    IQR_CASE(1) IQR_CASE(2) IQR_CASE(3) IQR_CASE(4) IQR_CASE(5) IQR_CASE(6) IQR_CASE(7) IQR_CASE(8)
    IQR_CASE(9) IQR_CASE(10) IQR_CASE(11) IQR_CASE(12) IQR_CASE(13) IQR_CASE(14) IQR_CASE(15) IQR_CASE(16)
#undef IQR_CASE
    default: return set_err(ctx, VBMC_ERR_UNSUPPORTED, "Na = %d not accelerated", is->Na);
  }
  double* r = dres.as<double>();
  hipLaunchKernelGGL(k_iqr_final, dim3((Nstar + 255) / 256), dim3(256), 0, st, Nstar, S, var_regularized ? 1 : 0, TolGPVar,
                     dacqs.as<double>(), pb.fmu, pb.fs2, r, r + Nstar, r + 2 * (size_t)Nstar);
  HIP_TRY(ctx, hipGetLastError());
  std::vector<double> hres((size_t)3 * Nstar);
  HIP_TRY(ctx, hipMemcpyAsync(hres.data(), dres.p, hres.size() * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  memcpy(acq, hres.data(), (size_t)Nstar * 8);
  if (fbar) memcpy(fbar, hres.data() + Nstar, (size_t)Nstar * 8);
  if (vtot) memcpy(vtot, hres.data() + 2 * (size_t)Nstar, (size_t)Nstar * 8);
  return VBMC_OK;
}

// ------------------------------------------------------------------------------------------
// The O(N^2) pieces of the rank-1 append (gplite_post.m:210-237) for every hyper-sample:
//   Ks = k(X, x*);  Lchol: v = L' \ Ks, x = L \ v  (alpha_update = x / sn2_eff, new column = v / sn2_eff)
//                   else : x = L * Ks               (alpha_update = -x)
// The O(N) assembly of the new alpha / L / sW stays with the caller (vbmc_amd/gplite.py, the MEX shim).
namespace {
// Ks = k(X, xstar), v = L' \ Ks, x = L \ v per hyper-sample on the device (gplite_post.m:226-237); for the samples that store
// -inv(K + sn2 I) instead of a factor, x = L Ks (k_symm) and v is unused.
// up to six small device-to-device copies in one launch (the per-sample constants a surrogate inherits from the one it extends)
struct CopySegs { const double* src[6]; double* dst[6]; int n[6]; };
__global__ void k_copy_segs(CopySegs c) {
  const int g = blockIdx.x;
  for (int i = threadIdx.x; i < c.n[g]; i += blockDim.x) c.dst[g][i] = c.src[g][i];
}

vbmc_status rank1_solves_dev(vbmc_ctx* ctx, const vbmc_gp* gp, const double* xstar, TmpBuf& dKs, TmpBuf& dV, TmpBuf& dXo) {
  const int N = gp->N, D = gp->D, S = gp->S;
  if (trsm_cw_for(N) == 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "N = %d too large", N);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  TmpBuf dxs;
  HIP_TRY(ctx, dxs.alloc(ctx, (size_t)D * 8));
  HIP_TRY(ctx, dKs.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, dV.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, dXo.alloc(ctx, (size_t)S * N * 8));
  HIP_TRY(ctx, hipMemcpyAsync(dxs.p, xstar, (size_t)D * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_gp_ks, dim3(4, S), dim3(256), 0, st, N, D, gp->Nhyp, gp->X, dxs.as<double>(), gp->hyp, gp->d_meanX, dKs.as<double>());
  HIP_TRY(ctx, hipMemcpyAsync(dV.p, dKs.p, (size_t)S * N * 8, hipMemcpyDeviceToDevice, st));
  // Lchol samples: triangular solves; the others (flag 0) are skipped by the kernels and handled by k_symm
  hipLaunchKernelGGL(k_symm, dim3(8, S, 1), dim3(256), 0, st, N, 1, S, gp->L, gp->d_lchol, dKs.as<double>(), dXo.as<double>());
  HIP_TRY(ctx, trsm_fwd_launch(st, N, 1, S, 1, gp->L, gp->d_finv, gp->d_lchol, dV.as<double>()));
  HIP_TRY(ctx, trsm_bwd_launch(st, N, 1, S, 1, gp->L, gp->d_finv, gp->d_lchol, dV.as<double>(), dXo.as<double>()));
  HIP_TRY(ctx, hipGetLastError());
  return VBMC_OK;
}
}  // namespace

extern "C" vbmc_status vbmc_gp_rank1_solves(vbmc_ctx* ctx, const vbmc_gp* gp, const double* xstar, double* Ks,
                                            double* v, double* x) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!gp || !xstar || !Ks || !v || !x) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_gp_rank1_solves: null argument");
  if (!gp->hasL) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_gp_rank1_solves needs gp.post(s).L on the device");
  const int N = gp->N, S = gp->S;
  TmpBuf dKs, dV, dXo;
  { vbmc_status s_ = rank1_solves_dev(ctx, gp, xstar, dKs, dV, dXo); if (s_) return s_; }
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemcpyAsync(Ks, dKs.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(v, dV.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(x, dXo.p, (size_t)S * N * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_gp_rank1_update(vbmc_ctx* ctx, const vbmc_gp* gp, const double* X_new, double ystar, const double* mstar,
                                            const double* vstar, const double* sn2_eff, double* alpha_new, double* L_new,
                                            vbmc_gp** out) {
  if (!ctx) return VBMC_ERR_INVALID;
  if (!gp || !X_new || !sn2_eff || !out || ((mstar == nullptr) != (vstar == nullptr)))
    return set_err(ctx, VBMC_ERR_INVALID, "vbmc_gp_rank1_update: null argument (mstar and vstar go together)");
  *out = nullptr;
  if (!gp->hasL) return set_err(ctx, VBMC_ERR_INVALID, "vbmc_gp_rank1_update needs gp.post(s).L on the device");
  const int N = gp->N, D = gp->D, S = gp->S, N1 = N + 1;
  if (trsm_cw_for(N1) == 0) return set_err(ctx, VBMC_ERR_UNSUPPORTED, "N = %d too large", N1);
  std::vector<double> xs(D);
  for (int d = 0; d < D; ++d) xs[d] = X_new[(size_t)N + (size_t)N1 * d];   // the appended row of the (N+1) x D matrix
  TmpBuf dKs, dV, dXo, dsc, dLn, dan;
  { vbmc_status s_ = rank1_solves_dev(ctx, gp, xs.data(), dKs, dV, dXo); if (s_) return s_; }
  hipStream_t st = ctx->stream;
  // per-sample scalars: sn2_eff (:207), Kss = sf2 (:213), (mstar - ystar)/vstar (:245), vstar
  std::vector<double> sc((size_t)S * 4);
  for (int s = 0; s < S; ++s) {
    sc[s * 4 + 0] = sn2_eff[s];
    sc[s * 4 + 1] = std::exp(2.0 * gp->hyp_host[(size_t)s * gp->Nhyp + D]);
    sc[s * 4 + 2] = mstar ? (mstar[s] - ystar) / vstar[s] : 0.0;
    sc[s * 4 + 3] = mstar ? vstar[s] : 0.0;
  }
  HIP_TRY(ctx, dsc.alloc(ctx, sc.size() * 8));
  HIP_TRY(ctx, hipMemcpyAsync(dsc.p, sc.data(), sc.size() * 8, hipMemcpyHostToDevice, st));
  if (!mstar) {   // the prediction at the new point from the solves at hand (no separate gplite_pred pass)
    TmpBuf dx1;
    HIP_TRY(ctx, dx1.alloc(ctx, (size_t)D * 8));
    HIP_TRY(ctx, hipMemcpyAsync(dx1.p, xs.data(), (size_t)D * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rank1_stats, dim3(S), dim3(256), 0, st, N, D, gp->Nhyp, gp->Ncov + gp->Nnoise, gp->meanfun, gp->hyp, dx1.as<double>(),
                       gp->alpha, gp->d_lchol, dKs.as<double>(), dV.as<double>(), dXo.as<double>(), ystar, dsc.as<double>());
  }
  HIP_TRY(ctx, dLn.alloc(ctx, (size_t)S * N1 * N1 * 8));
  HIP_TRY(ctx, dan.alloc(ctx, (size_t)S * N1 * 8));
  hipLaunchKernelGGL(k_rank1_assemble, dim3(N1, S), dim3(256), 0, st, N, gp->L, gp->alpha, gp->d_lchol, dV.as<double>(), dXo.as<double>(),
                     dsc.as<double>(), dLn.as<double>(), dan.as<double>());
  HIP_TRY(ctx, hipGetLastError());
  // The new surrogate ADOPTS the assembled factors and alpha (round 5; before: alpha to the host, a second round of six pageable
  // uploads, a device-to-device copy of the S factors, two more synchronisations).  Its small blocks are windows of ONE pooled
  // block [X_new | meanX | hyp | gpc | sn2_eff | mult | lchol]: X_new and its column means come up through the pinned block, the
  // per-sample constants -- unchanged by an append -- are copied from the surrogate it extends in one launch.
  const int Nhyp = gp->Nhyp;
  const size_t nXn = (size_t)N1 * D, nG = (size_t)S * GPC_STRIDE(D), nH = (size_t)Nhyp * S;
  const size_t nsmall = nXn + D + nH + nG + 2 * (size_t)S + ((size_t)S + 7) / 8;
  { vbmc_status s_ = ensure_pin(ctx, ((nXn + D) + (size_t)S * N1) * 8 + 8); if (s_) return s_; }
  double* hin = (double*)ctx->pin;
  memcpy(hin, X_new, nXn * 8);
  for (int d = 0; d < D; ++d) {
    double acc = 0.0;
    for (int n = 0; n < N1; ++n) acc += X_new[n + (size_t)N1 * d];
    hin[nXn + d] = acc / N1;
  }
  double* ah = hin + nXn + D;              // alpha comes back here
  vbmc_gp* ng = new vbmc_gp();
  ng->N = N1; ng->D = D; ng->S = S; ng->Nhyp = Nhyp; ng->Ncov = gp->Ncov; ng->Nnoise = gp->Nnoise; ng->meanfun = gp->meanfun;
  ng->hyp_host = gp->hyp_host; ng->sn2_eff = gp->sn2_eff; ng->Lchol = gp->Lchol;     // post.sW(1) is unchanged by the append (:239)
  ng->pooled = true; ng->in_views = true; ng->hasL = true;
  {
    hipError_t e = pool_get(ctx, nsmall * 8, &ng->blk_in);
    if (e == hipSuccess) e = pool_get(ctx, (size_t)S * TRSM_NBLK(N1) * 256 * sizeof(double), (void**)&ng->d_finv);
    if (e != hipSuccess) { vbmc_gp_free(ctx, ng); return set_err(ctx, VBMC_ERR_HIP, "vbmc_gp_rank1_update: %s", hipGetErrorString(e)); }
  }
  double* blk = (double*)ng->blk_in;
  ng->X = blk; ng->d_meanX = blk + nXn; ng->hyp = blk + nXn + D; ng->gpc = ng->hyp + nH; ng->d_sn2 = ng->gpc + nG; ng->d_mult = ng->d_sn2 + S;
  ng->d_lchol = (unsigned char*)(ng->d_mult + S);
  ng->L = dLn.as<double>(); dLn.p = nullptr;
  ng->alpha = dan.as<double>(); dan.p = nullptr;
  bool ok = true;
  if (copy_by_kernel((nXn + D) * 8))
    hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)std::min<size_t>((nXn + D + 255) / 256, 1024)), dim3(256), 0, st, nXn + D, (const double*)hin, blk);
  else
    ok = hipMemcpyAsync(blk, hin, (nXn + D) * 8, hipMemcpyHostToDevice, st) == hipSuccess;
  {
    CopySegs cs{};
    cs.src[0] = gp->hyp; cs.dst[0] = ng->hyp; cs.n[0] = (int)nH;
    cs.src[1] = gp->gpc; cs.dst[1] = ng->gpc; cs.n[1] = (int)nG;
    cs.src[2] = gp->d_sn2; cs.dst[2] = ng->d_sn2; cs.n[2] = S;
    cs.src[3] = gp->has_noise ? gp->d_mult : gp->d_sn2; cs.dst[3] = ng->d_mult; cs.n[3] = S;     // (without a noise model the slot is just initialised)
    cs.src[4] = (const double*)gp->d_lchol; cs.dst[4] = (double*)ng->d_lchol; cs.n[4] = 0;
    hipLaunchKernelGGL(k_copy_segs, dim3(4), dim3(256), 0, st, cs);
    ok = ok && hipMemcpyAsync(ng->d_lchol, gp->d_lchol, (size_t)S, hipMemcpyDeviceToDevice, st) == hipSuccess;     // bytes: not a whole number of doubles
  }
  hipLaunchKernelGGL(k_diag_inv, dim3(TRSM_NBLK(N1), S), dim3(64), 0, st, N1, ng->L, ng->d_lchol, ng->d_finv);
  if (copy_by_kernel((size_t)S * N1 * 8))
    hipLaunchKernelGGL(k_copy_f64, dim3((unsigned)std::min<size_t>(((size_t)S * N1 + 255) / 256, 1024)), dim3(256), 0, st, (size_t)S * N1,
                       (const double*)ng->alpha, ah);
  else
    ok = ok && hipMemcpyAsync(ah, ng->alpha, (size_t)S * N1 * 8, hipMemcpyDeviceToHost, st) == hipSuccess;
  ok = ok && hipGetLastError() == hipSuccess;
  {
    const hipError_t e = stream_wait_latency(st, N1 > 1024);
    if (!ok || e != hipSuccess) { vbmc_gp_free(ctx, ng); (void)hipGetLastError(); return set_err(ctx, VBMC_ERR_HIP, "vbmc_gp_rank1_update: %s", hipGetErrorString(e)); }
  }
  if (gp->has_noise) {
    for (int i = 0; i < 3; ++i) ng->noisefun[i] = gp->noisefun[i];
    ng->has_noise = true;
  }
  if (alpha_new) memcpy(alpha_new, ah, (size_t)S * N1 * 8);
  if (L_new) { vbmc_status s_ = d2h_bounced(ctx, L_new, ng->L, (size_t)S * N1 * N1 * 8); if (s_) { vbmc_gp_free(ctx, ng); return s_; } }
  *out = ng;
  return VBMC_OK;
}

