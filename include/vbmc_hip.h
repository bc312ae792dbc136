/*
 * vbmc_hip.h -- C ABI of libvbmc_hip.so: the MI355X (gfx950) implementation of the VBMC
 * ELBO inner loop.  This is the drop-in boundary: the reference (acerbilab/vbmc v1.0.12)
 * is pure MATLAB with no FFI seam, so these entry points are what a MEX
 * gateway (matlab/vbmc_hip_mex.cpp) or a ctypes binding (vbmc_amd/_lib.py) binds, one per
 * reference function on the path.  Each declaration cites the reference interface it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - all arrays are column-major IEEE fp64 exactly as MATLAB stores them, caller-owned,
 *     non-aliasing; the library never keeps a host pointer after the call returns;
 *   - every function returns a vbmc_status (0 = OK) and never throws across the ABI;
 *     vbmc_last_error(ctx) gives the message of the last failure on that context;
 *   - a context owns one HIP stream and all device scratch; calls on one context are
 *     serialised by the caller (the MATLAB interpreter thread / one Python process per GPU);
 *   - the library FAILS (VBMC_ERR_NO_DEVICE) when no gfx950 device is present: there is no
 *     CPU fallback anywhere behind this ABI.
 */
#ifndef VBMC_HIP_H
#define VBMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBMC_ABI_VERSION 5

typedef int vbmc_status;
enum {
  VBMC_OK = 0,
  VBMC_ERR_INVALID = 1,      /* bad argument (message says which)                         */
  VBMC_ERR_NO_DEVICE = 2,    /* no HIP device / not gfx950                                */
  VBMC_ERR_HIP = 3,          /* a HIP runtime call failed                                 */
  VBMC_ERR_UNSUPPORTED = 4,  /* option outside the accelerated path: caller must fall     */
                             /* through to the reference .m (e.g. meanfun not in {0,1,4}) */
  VBMC_ERR_NOT_POSDEF = 5    /* Cholesky failed after the reference's 10 jitter retries   */
};

typedef struct vbmc_ctx vbmc_ctx; /* device context: stream + scratch                      */
typedef struct vbmc_gp vbmc_gp;   /* device-resident gp.post(1..S) (gplite_post.m:94-157)  */

/* ---- library / context ------------------------------------------------------------- */
int vbmc_abi_version(void);
/* The shapes the library accepts -- the numbers its own validation enforces (anything beyond is VBMC_ERR_UNSUPPORTED and falls
 * through to the reference), so that a binding asks instead of restating them: matlab/vbmc_hip_supported.m reads them through
 * vbmc_hip_mex('limits').  A pure host function: no device, no context.  Replaces nothing in the reference (it has no limits);
 * the fall-through edges are listed in INTEGRATION.md section 4. */
typedef struct vbmc_limits {
  uint32_t struct_size;      /* sizeof(vbmc_limits) of the caller */
  int32_t max_D;             /* dimensions (every entry point) */
  int32_t max_K;             /* mixture components (misc/negelcbo_vbmc.m, ent/entmc_vbmc.m) */
  int32_t max_N;             /* training points of the surrogate on the factor paths: gplite_post / _pred / _nlZ, the variance of the
                                expected log joint, the acquisition sweep (the plain expected log joint has no limit) */
  int32_t max_Na;            /* importance points of acqviqr / acqimiqr */
  int32_t max_T_vargrad;     /* variational parameters with the gradient of the variance (compute_var = 2) */
  int32_t delta_ok;          /* 1: vp.delta ~= 0 is accelerated */
  int32_t meanfun_mask;      /* bit i set: gplite mean function id i is accelerated (0, 1, 4) */
} vbmc_limits;
vbmc_status vbmc_get_limits(vbmc_limits* out);
/* stream: a hipStream_t to launch on (e.g. torch's current stream) or NULL for a private one.  Side effect of the FIRST call in a process (std::call_once;
 * later contexts do not touch the environment): setenv("GPU_MAX_HW_QUEUES", "8", no overwrite) -- the pipelined forms keep five
 * streams busy and the runtime's default of four hardware queues serialises two of them (VBMC_HW_QUEUES=0: leave it alone,
 * VBMC_HW_QUEUES=n: that many).  It takes effect only if no other HIP user (torch, say) initialised the runtime earlier; a host with
 * threads that read the environment concurrently should set the variable itself and pass VBMC_HW_QUEUES=0. */
vbmc_status vbmc_ctx_create(int device, void* stream, vbmc_ctx** out);
void vbmc_ctx_destroy(vbmc_ctx* ctx);
const char* vbmc_last_error(const vbmc_ctx* ctx);
vbmc_status vbmc_ctx_synchronize(vbmc_ctx* ctx);
/* When enabled, HIP events bracket the dominant kernel (entropy MC) of every elbo call on the
 * context's stream; vbmc_ctx_last_kernel_ms returns its duration (bench.py roofline leg).
 * enable = 1: the call as it normally runs (a blocking call forks the expected log joint onto
 * the context's second, low-priority stream, where it shares the chip with the dominant kernel
 * and stretches its duration); enable = 2: nothing is forked beside it -- the duration of the
 * kernel alone, the figure a roofline prices. */
vbmc_status vbmc_ctx_set_profiling(vbmc_ctx* ctx, int enable);
vbmc_status vbmc_ctx_last_kernel_ms(vbmc_ctx* ctx, double* ent_ms, double* logjoint_ms);

/* ---- GP surrogate state -------------------------------------------------------------- */
/*
 * Upload gp.X and gp.post(s).{hyp,alpha,L,sW,Lchol} once per vpoptimize_vbmc call
 * (misc/vpoptimize_vbmc.m:71 closes over a constant gp).  Replaces the reads at
 * misc/gplogjoint.m:97-160.
 *   X      N x D      training inputs
 *   hyp    Nhyp x S   [log ell(D); log sf; noise(Nnoise); mean(Nmean)]
 *   alpha  N x S
 *   L      N x N x S  upper Cholesky factor (Lchol=1) or -inv(K+sn2 I) (Lchol=0); may be NULL
 *                     when only value/gradient without variance will be requested
 *   sW1    S          gp.post(s).sW(1)   (sn2_eff = 1/sW1^2, gplogjoint.m:160)
 *   Lchol  S          uint8 flags
 *   meanfun           gplite mean-function id; only 0 (zero), 1 (const), 4 (negquad) are
 *                     accelerated -- others return VBMC_ERR_UNSUPPORTED
 */
vbmc_status vbmc_gp_upload(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int Ncov, int Nnoise,
                           int meanfun, const double* X, const double* hyp, const double* alpha,
                           const double* L, const double* sW1, const uint8_t* Lchol,
                           vbmc_gp** out);
/* Releases a surrogate.  Its device blocks belong to the creating context's pool: pass that context while it is alive
 * (after vbmc_ctx_destroy the blocks are already gone and ctx may be NULL). */
void vbmc_gp_free(vbmc_ctx* ctx, vbmc_gp* gp);
/* Noise model needed by vbmc_gp_pred: gp.noisefun (3 ids, gplite_noisefun.m:176-210) and
 * gp.post(s).sn2_mult (S). */
vbmc_status vbmc_gp_set_noise(vbmc_ctx* ctx, vbmc_gp* gp, const int32_t noisefun[3], const double* sn2_mult);

/*
 * gp = gplite_post(hyp, X, y, covfun=SE-ARD, meanfun, noisefun, s2)   (gplite/gplite_post.m:1,
 * gplite/private/gplite_core.m:1-102,278-291; full posterior, no rank-1): per hyper-sample the
 * ARD-SE kernel matrix via sq_dist, the jittered Cholesky with the reference's x10 noise
 * inflation retry (<= 10 tries), alpha, and L (upper factor, or -inv(K+sn2 I) when
 * min(sn2) < 1e-6).  Outputs (any may be NULL): alpha N x S, L N x N x S, sW N x S, sn2_mult S,
 * Lchol S; *gp_out (optional) receives the device-resident posterior for vbmc_elbo_batch /
 * vbmc_gp_pred without a second upload: it takes over the device blocks the factorisation worked in (no copy of the S
 * N x N factors) and is ready when the call returns -- one upload, one synchronisation per call.
 * This call and vbmc_gp_nlz end with a polling wait on the context's stream (at most 300 us of hipStreamQuery before the
 * blocking wait, none where the enqueued work is known to be large: their callers issue the next evaluation at once).
 */
vbmc_status vbmc_gp_post(vbmc_ctx* ctx, int N, int D, int S, int Nhyp, int meanfun, const int32_t noisefun[3],
                         const double* X, const double* y, const double* s2, const double* hyp,
                         double* alpha, double* L, double* sW, double* sn2_mult, uint8_t* Lchol,
                         vbmc_gp** gp_out);

/*
 * acq = acqwrapper_vbmc(Xs,vp,gp,optimState,0,acqFun,acqInfo)   (acq/acqwrapper_vbmc.m:11-46) for the
 * density-based acquisition functions, with vp.delta = 0: gplite_pred for every hyper-sample (:17), fbar / vtot
 * (:21-29), p = max(vbmc_pdf(vp,Xs,0),realmin), then
 *   acq_id 0  acqf_vbmc     -vtot .* exp(fbar - ymax) .* p                       (acq/acqf_vbmc.m:9-10)
 *   acq_id 1  acqflog_vbmc  -(log(vtot) + fbar - ymax + log(p))                  (acq/acqflog_vbmc.m:17-18)
 *   acq_id 2  acqus_vbmc    -vtot .* p.^2                                        (acq/acqus_vbmc.m:9)
 *   acq_id 3  acqfsn2_vbmc  -vtot .* (1 - sn2./(vtot+sn2)) .* exp(fbar - ymax) .* p, sn2 = gp.sn2new at the nearest
 *             row of gp.X_rescaled to Xs ./ optimState.gplengthscale               (acq/acqfsn2_vbmc.m:9-17)
 * followed by the variance regularisation (:35-45, if var_regularized) and max(acq,-realmax) (:46).  NOT done
 * here: the integer mapping (:8) and the hard-bound test in the ORIGINAL parameter space (:49-51), which need the
 * caller's warpvars_vbmc; the caller sets acq(outside) = Inf.  Xs is Nstar x D column-major in transformed
 * coordinates; vp_mu D x K.  Optional outputs fbar, vtot (Nstar each, may be NULL).
 */
vbmc_status vbmc_acq_eval(vbmc_ctx* ctx, const vbmc_gp* gp, int Nstar, const double* Xs, int acq_id, int K,
                          const double* vp_mu, const double* vp_sigma, const double* vp_lambda, const double* vp_w,
                          double ymax, int var_regularized, double TolGPVar, const double* gplengthscale,
                          const double* X_rescaled, const double* sn2new, double* acq, double* fbar, double* vtot);

/*
 * Importance-sampled IQR acquisition functions: acqviqr_vbmc (acq/acqviqr_vbmc.m:36-109) and acqimiqr_vbmc
 * (acq/acqimiqr_vbmc.m:30-95) behind acqwrapper_vbmc (acq/acqwrapper_vbmc.m:11-46, log-valued).
 *
 * vbmc_acq_is_create uploads optimState.ActiveImportanceSampling for one gp: the Na importance points Xa
 * (Na x D, or Na x D x S if per_sample_inputs), lnw (S x Na; NULL = zeros, i.e. VIQR), fs2a (Na x S; NULL = computed
 * here with gplite_pred, shared inputs only) and Ctmp_mat (N x Na x S; NULL = computed here as in
 * private/activeimportancesampling_vbmc.m:248-276: (L\(L'\Kax'))/sn2_eff for Lchol samples, L*Kax' otherwise).
 * IMIQR's per-call solve (acqimiqr_vbmc.m:77-79) is the same matrix, so one state serves both functions.
 *
 * vbmc_acq_iqr_eval: per hyper-sample C = Ka -/+ Ks'*Ctmp (:84-90), tau2 = C.^2 ./ ys2 with ys2 = fs2 + sn2new at
 * the nearest row of X_rescaled (:42-45), s_pred = sqrt(max(fs2a' - tau2, 0)), log-sum-exp of
 * lnw + u*s_pred + log1p(-exp(-2*u*s_pred)) over the importance points (:97-102), log-mean-exp over hyper-samples
 * (:104-107), the wrapper's variance regulariser and clamp.  The integer mapping and hard-bound test stay with the
 * caller (see vbmc_acq_eval).  Xs: Nstar x D column-major, transformed coordinates.
 */
typedef struct vbmc_acq_is vbmc_acq_is;
vbmc_status vbmc_acq_is_create(vbmc_ctx* ctx, const vbmc_gp* gp, int Na, const double* Xa, int per_sample_inputs,
                               const double* lnw, const double* fs2a, const double* Ctmp, vbmc_acq_is** out);
void vbmc_acq_is_free(vbmc_ctx* ctx, vbmc_acq_is* is);
vbmc_status vbmc_acq_iqr_eval(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_acq_is* is, int Nstar, const double* Xs,
                              const double* gplengthscale, const double* X_rescaled, const double* sn2new,
                              int var_regularized, double TolGPVar, double* acq, double* fbar, double* vtot);

/*
 * [nlZ,dnlZ] = gplite_nlZ(hyp,gp,[])   (gplite/gplite_nlZ.m:1-72 -> gplite/private/gplite_core.m:1-102,128-275)
 * for B hyper-parameter vectors at once (the walkers / restarts of gplite_train.m:181,251,292,330): negative
 * log marginal likelihood nlZ (B) and, if compute_grad, its gradient dnlZ (Nhyp x B, column-major).  SE-ARD
 * covariance, mean functions 0/1/4, noise models of gplite_noisefun.m:176-210, no integrated mean, no output
 * warping, no hyper-prior (gplite_hypprior is O(Nhyp) host work; see vbmc_amd/gplite.py).  A matrix that is
 * still not positive definite after the 10 noise-inflation retries yields NaN for that vector
 * (gplite_train.m:542-546), not an error.
 */
vbmc_status vbmc_gp_nlz(vbmc_ctx* ctx, int N, int D, int B, int Nhyp, int meanfun, const int32_t noisefun[3],
                        const double* X, const double* y, const double* s2, const double* hyp, int compute_grad,
                        double* nlZ, double* dnlZ);

/*
 * [ymu,ys2,fmu,fs2] = gplite_pred(gp, Xstar, ystar, s2star, ssflag)   (gplite/gplite_pred.m:1-165).
 * Xstar is Nstar x D.  ssflag = 0: outputs are Nstar vectors averaged over hyper-samples with
 * the between-sample variance added (:154-165); ssflag = 1: Nstar x S per-sample outputs.
 * ystar (Nstar, may be NULL = []) only enters the noise at the test points for output-dependent noise models
 * (gp.noisefun(3) = 1: sn2 += w^2 max(0, ythresh - ystar)^2, gplite_noisefun.m:198-207; skipped when ystar is empty, as in
 * the reference); fmu / fs2 never depend on it.
 */
vbmc_status vbmc_gp_pred(vbmc_ctx* ctx, const vbmc_gp* gp, int Nstar, const double* Xstar, const double* ystar,
                         const double* s2star, int ssflag, double* ymu, double* ys2, double* fmu, double* fs2);

/*
 * The O(N^2) pieces of gplite_post's rank-1 append of one training point x* (gplite/gplite_post.m:173-251),
 * for every hyper-sample: Ks = k(X, x*) (N x S); for Lchol samples v = L' \ Ks and x = L \ v, so that
 * alpha_update = x / sn2_eff (:227) and the new column of L is v / sn2_eff (:228); for low-noise samples
 * x = L * Ks (alpha_update = -x, :234).  The O(N) assembly of the enlarged posterior is the caller's.
 */
vbmc_status vbmc_gp_rank1_solves(vbmc_ctx* ctx, const vbmc_gp* gp, const double* xstar, double* Ks, double* v,
                                 double* x);

/* C = sq_dist(a, b)   (utils/sq_dist.m:14-50): a is D x n, b is D x m (NULL -> b = a), C is n x m.
 * The a'b contraction runs on v_mfma_f64_16x16x4_f64. */
/* gp = gplite_post(gp, xstar, ystar, [], [], [], [], 1): rank-one append of one observation, entirely on the device
 * (gplite/gplite_post.m:173-251).  X_new is the (N+1) x D training matrix with xstar as its last row; mstar, vstar (S each)
 * are [mstar, vstar] = gplite_pred(gp, xstar, ystar, [], 1) (:189) -- or both NULL, in which case they are computed here from
 * the solves of the append itself -- and sn2_eff (S) the noise at the new point times sn2_mult (:196-207), which the caller has.  *out is a NEW surrogate handle with N+1 points (the old one stays valid);
 * alpha_new ((N+1) x S) and L_new ((N+1) x (N+1) x S) are optional host copies of the new gp.post(s).alpha / .L. */
vbmc_status vbmc_gp_rank1_update(vbmc_ctx* ctx, const vbmc_gp* gp, const double* X_new, double ystar, const double* mstar,
                                 const double* vstar, const double* sn2_eff, double* alpha_new, double* L_new, vbmc_gp** out);
vbmc_status vbmc_sq_dist(vbmc_ctx* ctx, int D, int n, int m, const double* a, const double* b, double* C);

/* ---- the ELBO objective ---------------------------------------------------------------
 * One call evaluates R independent negelcbo_vbmc(theta_r, beta, vp, gp, Ns, compute_grad,
 * compute_var, ~, thetabnd) (misc/negelcbo_vbmc.m:1) -- R = 1 is the Adam-loop call
 * (misc/vpoptimize_vbmc.m:71, utils/fminadam.m:48), R > 1 is the sieve batch
 * (misc/vpsieve_vbmc.m:74-78).
 * gp == NULL evaluates the entropy term alone: H, dH (and F = -H + penalties) are then
 * [H,dH] = entmc_vbmc(vp,Ns,grad_flags,1) (ent/entmc_vbmc.m:1) for Ns > 0 and entlb_vbmc(vp,grad_flags,1)
 * (ent/entlb_vbmc.m:1) for Ns = 0, with grad_flags = optimize[]; G = 0, dG = 0; compute_var and
 * separate_K must be 0.  With Ns = 0 and a surrogate, G, dG, varG, varGss, I_sk, J_sjk are the outputs of
 * gplogjoint(vp,gp,grad_flags,1,1,compute_var,separate_K) (misc/gplogjoint.m:1) on its own; G_s / varG_s are its
 * avg_flag = 0 outputs.
 */
typedef struct vbmc_elbo_args {
  uint32_t struct_size;      /* = sizeof(vbmc_elbo_args), for ABI versioning                */
  int32_t D, K, R;
  int32_t optimize[4];       /* vp.optimize_{mu,sigma,lambda,weights}                       */
  const double* theta;       /* T x R, T = sum of optimised groups, negelcbo_vbmc.m:33-48   */
  /* current vp fields; used for every group that is NOT optimised (may be NULL otherwise)  */
  const double* vp_mu;       /* D x K */
  const double* vp_sigma;    /* K     */
  const double* vp_lambda;   /* D     */
  const double* vp_w;        /* K     */
  const double* vp_delta;    /* D or NULL (= 0), gplogjoint.m:85-89                         */
  int32_t Ns;                /* MC samples per component; forced even (entmc_vbmc.m:45);    */
                             /* 0 -> deterministic bound entlb_vbmc (negelcbo_vbmc.m:104-110) */
  int32_t eps_mode;          /* 0 device Philox RNG; 1 eps on host; 2 eps already on device */
  const double* eps;         /* D x Ns/2 x K (x R unless eps_shared): the K consecutive     */
                             /* randn(D,1,Ns/2) blocks of entmc_vbmc.m:53                   */
  int32_t eps_shared;        /* 1: one eps block reused for all R restarts                  */
  uint64_t seed;             /* eps_mode 0: Philox key; stream position = (r, j, sample)    */
  int32_t compute_grad;      /* negelcbo arg 6                                              */
  int32_t compute_var;       /* 0 none, 1 full K x K, 2 diagonal (gplogjoint.m:273-337)     */
  int32_t separate_K;        /* also return I_sk (and J_sjk when compute_var)               */
  double beta;               /* ELCBO weight (negelcbo arg 2)                               */
  /* soft bounds (misc/vpbounds.m, misc/vpbndloss.m); lb == NULL -> thetabnd = []           */
  const double* bnd_lb;      /* length of theta_ext: [mu(:); lnscale(:) (D x K); eta]       */
  const double* bnd_ub;
  double TolCon, WeightThreshold, WeightPenalty;
  double sparse_cutoff;      /* 0: dense (every sample x component term, as the reference).  c > 0: block-sparse -- a
                              * 16-component tile is skipped for a 16-sample tile when every one of its terms is
                              * provably < exp(-c) relative to q(x) (c = 100 changes results by < 1e-40 relative)   */
  /* outputs (any may be NULL) */
  double* F;                 /* R      negative EL(C)BO incl. penalties                     */
  double* dF;                /* T x R                                                       */
  double* G;                 /* R      expected log joint                                   */
  double* H;                 /* R      entropy                                              */
  double* dG;                /* T x R                                                       */
  double* dH;                /* T x R                                                       */
  double* varG;              /* R                                                           */
  double* varGss;            /* R                                                           */
  double* I_sk;              /* S x K x R                                                   */
  double* J_sjk;             /* S x K x K x R                                               */
  /* per-hyper-sample outputs of gplogjoint(vp,gp,0,0,...) (avg_flag = 0, misc/gplogjoint.m:399: no averaging) -- what
   * private/activesample_vbmc.m:155 ([~,~,varF] = gplogjoint(vp,gp,0,0,0,1)) and misc/vpoptimizeweights_vbmc.m:42 read */
  double* G_s;               /* S x R  F(s) = sum_k w_k I_sk  (:203)                        */
  double* varG_s;            /* S x R  varF(s), each max(.,eps) (:283,:329-332,:350); needs compute_var */
  int32_t chunk_world;       /* 0 / 1: the Monte-Carlo samples of a (restart, component) are split into as many chunks as fill
                              * THIS device once.  W > 1: as many as fill W devices -- the chunking vbmc_elbo_shard_* use for a
                              * world of W ranks, so that an unsharded evaluation with chunk_world = W is their bit-exact
                              * reference (the chunk count only moves the summation order of the entropy partials) */
  int32_t restart_offset;    /* eps_mode 0: the device stream of restart r (column r of theta) is keyed by                    */
  int32_t restart_stride;    /* restart_offset + r * restart_stride (stride 0 is read as 1).  0 / 1: the position in this      */
                             /* batch.  A batch dealt over G devices (vbmc_elbo_batch_multi: device g gets restarts g, g+G,   */
                             /* ...) passes g / G, so that every restart draws what it would draw in the undivided batch      */
  int32_t no_jacobian;       /* 1: gradients with respect to sigma, lambda and the weights w themselves -- the JACOBIAN_FLAG = 0  */
                             /* form of gplogjoint / entmc_vbmc / entlb_vbmc (misc/gplogjoint.m:352-373, ent/entmc_vbmc.m:110-125, */
                             /* ent/entlb_vbmc.m:132-143) -- instead of log sigma, log lambda, eta.  Stand-alone forms only: no soft  */
                             /* bounds (bnd_lb NULL); the variance gradient follows suit (round 5: misc/gplogjoint.m:375-396          */
                             /* skipped).  0 (default): the transformed gradients negelcbo uses                                     */
  double* dvarG;             /* T x R  gradient of the diagonal variance of the expected log joint, gplogjoint's 4th output     */
                             /* (compute_var = 2 with compute_grad; misc/gplogjoint.m:375-413), or NULL                          */
  double* dG_s;              /* T x S x R  gradient of the expected log joint PER hyper-sample: gplogjoint's dF with avg_flag = 0  */
                             /* (misc/gplogjoint.m:206-271 per s, Jacobians :352-373, the averaging of :411 skipped); needs      */
                             /* compute_grad; honours no_jacobian.  ABI version 4.  NULL: not wanted                               */
  /* ---- ABI version 5 ---- */
  double* dvarG_s;           /* T x S x R  gradient of the diagonal variance PER hyper-sample: gplogjoint's dvarF with avg_flag = 0   */
                             /* (misc/gplogjoint.m:286-304 per s, Jacobians :375-396, the averaging of :407-409 skipped); needs      */
                             /* compute_grad with compute_var = 2; honours no_jacobian.  NULL: not wanted                           */
  int32_t plan_restarts;     /* 0: launch shapes follow THIS call's R.  P > 0: this call is a share of a batch of P restarts (dealt   */
                             /* over devices: restart_offset / restart_stride): the sample chunking, the log-joint kernel and its     */
                             /* splits are chosen as for a batch of P, so that every restart's results are BIT-IDENTICAL to the ones   */
                             /* the undivided batch evaluated with plan_restarts = P computes (a blocking call with plan_restarts 0    */
                             /* may take the walking entropy launch of wide batches: same draws, another order of summation over a      */
                             /* component's partial records, 1e-13) -- the opt-in exact mode of                                         */
                             /* vbmc_elbo_batch_multi / vbmc_elbo_multi_submit, which pass the field through.  Costs throughput where   */
                             /* a share is much smaller than the batch (launch shapes of a full chip on an eighth of the work).        */
} vbmc_elbo_args;

vbmc_status vbmc_elbo_batch(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* args);

/*
 * The same pass, pipelined, for a stream of INDEPENDENT batches -- the candidates of the sieve are evaluated one after the
 * other with no dependence between them (misc/vpsieve_vbmc.m:74-78), and so are the full-ELCBO re-evaluations of
 * misc/vpoptimize_vbmc.m:134-165:
 *   vbmc_elbo_submit   validates and stages the inputs of one batch in the slot's own pinned block, enqueues the H2D, the
 *                      kernels and the packed D2H on the context's stream and returns WITHOUT waiting;
 *   vbmc_elbo_collect  waits for that slot's pass and fills the outputs named in args (F, dF, G, H, dG, dH, varG, varGss).
 * Four slots (0 .. 3; round 4: slot s runs on slot stream s & 1 of two streams the context creates at the first submit, two passes
 * deep each -- the head and tail of one pass overlap the entropy kernel of another and a stream never runs dry while the host collects
 * and re-submits; slots 2 and 3 exist for passes without a variance term; the variance forms run on the context's own stream, slots 0
 * and 1).  The two slot streams are chosen among candidates that the runtime places on different hardware queues and dispatch pipes
 * (measured at creation; vbmc_ctx_create raises GPU_MAX_HW_QUEUES to 8 ahead of its own first HIP call so that there are queues to choose
 * from -- unless the environment holds a value, VBMC_HW_QUEUES=0 forbids it, or another HIP user initialised the runtime first: see
 * INTEGRATION.md), and a slot stream is ordered after whatever the context's own stream still holds at submit (a surrogate being
 * uploaded, draws being produced).  COST OF THE FIRST SUBMIT of a context: the placement is MEASURED -- up to six candidate streams
 * per slot stream, each timed with two probe launches beside every stream it has to share the device with (a 45 us low-occupancy
 * kernel and a one-wave kernel, twice, with a synchronisation each): 1-6 ms once per context, device otherwise idle, before the
 * first batch is enqueued; VBMC_PLACE=0 skips the measurement and takes the first candidate (VBMC_DEBUG_PLACE=1 prints the ratios).
 * Create the context, and submit a first (warm-up) batch, outside a timed region.  While the device works on one batch the host stages the next, so that the device never waits for
 * the host between batches.  Measured at the headline shape: 2.49 ms per blocking call of 64 restarts, 2.39-2.42 ms per pipelined batch;
 * 0.37 / 0.33 ms for 8 restarts, 102 / 65 us for one.  Passes of one slot stream execute in submission order; each
 * slot must be collected before it is submitted again.  Results are bit-identical to vbmc_elbo_batch with the same args.
 * Not offered here: separate_K / I_sk / J_sjk / G_s / varG_s outputs and host-resident draws (eps_mode 1) -- their copies
 * go through pageable memory; use vbmc_elbo_batch.  The surrogate handle and the arrays named in args must stay valid until the
 * slot is collected (the inputs are copied at submit, the outputs are written at collect).  Other entry points of the same context may be called between a submit and
 * its collect (they run on the context's own stream, beside the passes in flight) -- except those that change or free the surrogate
 * a pass in flight reads (vbmc_gp_set_noise, vbmc_gp_free): collect first.  (Round 5, ABI 5: those two now wait for the slot streams
 * that hold a pass before they touch the surrogate -- the pass completes on the old data and stays collectable -- so a forgotten
 * collect costs a synchronisation, not a read of recycled memory.)
 *   vbmc_elbo_abandon  gives a slot back WITHOUT its results: waits for the pass in flight, if any, and clears the slot (a caller that
 *                      will not collect: an exception between submit and collect, an abandoned generator).  VBMC_OK on an idle slot.
 */
vbmc_status vbmc_elbo_submit(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* args, int slot);
vbmc_status vbmc_elbo_collect(vbmc_ctx* ctx, const vbmc_elbo_args* args, int slot);
vbmc_status vbmc_elbo_abandon(vbmc_ctx* ctx, int slot);

/*
 * ONE evaluation (or a batch with fewer restarts than GPUs) sharded over `world` ranks, one process per GPU, each with a
 * full replica of the surrogate: rank g evaluates the expected-log-joint records of its hyper-samples (the iterations of
 * misc/gplogjoint.m:98 are independent until the averaging at :399-413) and the Monte-Carlo entropy partials of its share
 * of the sample chunks (ent/entmc_vbmc.m:49-104: the samples are independent) into one contiguous device block;
 *   vbmc_elbo_shard_size    number of doubles of that block (equal on every rank);
 *   vbmc_elbo_shard_begin   fills d_send (device memory of the caller, e.g. a torch tensor) and synchronises;
 *   -- the caller all-gathers the blocks in rank order (RCCL ncclAllGather over xGMI: torch.distributed
 *      all_gather_into_tensor; ~0.3 MB per rank at the headline shape) --
 *   vbmc_elbo_shard_finish  scatters the gathered blocks into the unsharded record layouts, runs the unsharded
 *                           fixed-order reductions and k_finalize on every rank and returns the outputs of
 *                           vbmc_elbo_batch: BIT-IDENTICAL to the 1-GPU evaluation with args.chunk_world = world (the
 *                           samples are split into `world` times as many chunks as one device needs, so that every rank's
 *                           share still fills its device; same chunking, same summation order => same bits).
 * Covers value + gradient without variance (compute_var = 0, separate_K = 0), device RNG (eps_mode 0): the optimiser-loop
 * call of misc/vpoptimize_vbmc.m:71.  args must be identical on all ranks and in all three calls.
 */
vbmc_status vbmc_elbo_shard_size(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* args, int world, size_t* n_doubles);
vbmc_status vbmc_elbo_shard_begin(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* args, int rank, int world, double* d_send);
vbmc_status vbmc_elbo_shard_finish(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* args, int world, const double* d_gathered);

/* ---- more than one GPU: the communicator inside the library -----------------------------------------------------------
 * The path shards over the independent restarts of the sieve / the optimiser (misc/vpsieve_vbmc.m:74-78,
 * misc/vpoptimize_vbmc.m:49): rank g of G evaluates restarts g, g + G, g + 2G, ... against its own replica of the surrogate, and
 * the ONE exchange is an all-gather of the restarts' ELCBO values (RCCL ncclAllGather, device to device over xGMI), after which
 * every rank holds the identical vector and performs the identical stable sort (:82) -- index-identical order with no broadcast.
 * RCCL is reached from inside the library (dlopen of librccl on first use), so that a host without a distributed runtime of its
 * own -- the MATLAB process behind matlab/vbmc_hip_mex.cpp -- can use every GPU of a node.
 *
 *   vbmc_comm_create_all    ONE process drives ndev devices (devices == NULL: 0 .. ndev-1): creates a context per device and an
 *                           RCCL rank per device (ncclCommInitAll).  vbmc_comm_ctx(c, i) is the context of local device i.
 *   vbmc_comm_unique_id /   one process PER device (python -m torch.distributed.run, mpirun): rank 0 obtains the 128-byte id and
 *   vbmc_comm_create_rank   hands it to the others by any means (a file, torch.distributed, MPI); every process then joins with its
 *                           own context (ncclCommInitRank).  The context stays the caller's.
 *   vbmc_allgather_f64      every local device contributes `count` doubles (d_send[i], device memory on local device i) and
 *                           receives size * count doubles in rank order (d_recv[i]); enqueued on the contexts' streams inside
 *                           ncclGroupStart / ncclGroupEnd, then synchronised.  All-gather, never all-reduce: the exchange moves
 *                           the values it is given without arithmetic, and every rank holds the identical vector afterwards
 *                           (whether the VALUES equal the one-GPU batch's bit for bit is vbmc_elbo_batch_multi's business: see
 *                           plan_restarts there).
 *   vbmc_allgather_host_f64 the same for host blocks (send: local * count doubles, recv: size * count), staged through the
 *                           communicator's own device blocks.
 *   vbmc_gp_upload_all      vbmc_gp_upload on every local device (gps[local]): the surrogate is replicated, not sharded (25.6 MB
 *                           at the headline shape against 288 GB of HBM per device).
 *   vbmc_elbo_batch_multi   vbmc_elbo_batch for R restarts dealt over the ranks.  args is the UNDIVIDED batch (theta T x R, outputs
 *                           sized for R), identical on every rank.  On return F and varG hold ALL R values on every rank (the
 *                           all-gathered vectors as device memory of local device 0 received them); the other outputs are filled
 *                           for the restarts this process evaluated (all of them for vbmc_comm_create_all) and left untouched
 *                           elsewhere.  The device stream of restart r is keyed by its index r in the undivided batch
 *                           (restart_offset / restart_stride): every rank evaluates exactly the Monte-Carlo estimator the one-device
 *                           batch evaluates, sample for sample.  The values agree with vbmc_elbo_batch of the whole batch on one
 *                           device to the order of summation (relative 1e-13): launch shapes -- the number of sample chunks per
 *                           component, which sets the order in which the entropy partials are added -- are chosen for the restarts a
 *                           device actually holds, which is what strong scaling needs (8 restarts per device want other chunks than
 *                           64).  Where the two launches coincide (small batches) the values are bit-identical.  EXACT MODE (round
 *                           5, opt-in): args.plan_restarts = R makes every device choose its launch shapes for the undivided batch
 *                           -- then every value is BIT-IDENTICAL to vbmc_elbo_batch of the whole batch on one device, whatever the
 *                           number of devices, and a sieve sorted on 1 GPU and on 8 orders even exact ties the same way
 *                           (tests/test_gpu_comm.py: shares of 2 / 3 / 8 at a shape where the two modes choose different chunkings).
 *                           All RANKS of one call always see the identical gathered vectors, hence the identical sieve order, in
 *                           either mode.  eps_mode 0, or one shared host block of draws (eps_mode 1 with eps_shared).
 * Errors of these calls are reported by vbmc_comm_last_error.
 */
typedef struct vbmc_comm vbmc_comm;
vbmc_status vbmc_comm_create_all(int ndev, const int* devices, vbmc_comm** out);
vbmc_status vbmc_comm_unique_id(void* id128);
vbmc_status vbmc_comm_create_rank(vbmc_ctx* ctx, int rank, int size, const void* id128, vbmc_comm** out);
void vbmc_comm_destroy(vbmc_comm* comm);
int vbmc_comm_size(const vbmc_comm* comm);    /* ranks of the communicator                         */
int vbmc_comm_local(const vbmc_comm* comm);   /* devices this process drives                        */
int vbmc_comm_rank(const vbmc_comm* comm);    /* rank of local device 0                             */
vbmc_ctx* vbmc_comm_ctx(vbmc_comm* comm, int local);
const char* vbmc_comm_last_error(const vbmc_comm* comm);
vbmc_status vbmc_allgather_f64(vbmc_comm* comm, const double* const* d_send, double* const* d_recv, size_t count);
vbmc_status vbmc_allgather_host_f64(vbmc_comm* comm, const double* send, double* recv, size_t count);
vbmc_status vbmc_gp_upload_all(vbmc_comm* comm, int N, int D, int S, int Nhyp, int Ncov, int Nnoise, int meanfun,
                               const double* X, const double* hyp, const double* alpha, const double* L, const double* sW1,
                               const uint8_t* Lchol, vbmc_gp** gps);
void vbmc_gp_free_all(vbmc_comm* comm, vbmc_gp** gps);
vbmc_status vbmc_elbo_batch_multi(vbmc_comm* comm, const vbmc_gp* const* gps, const vbmc_elbo_args* args);
/* The pipelined form of vbmc_elbo_batch_multi for streams of INDEPENDENT batches (the candidates of misc/vpsieve_vbmc.m:74-78), as
 * vbmc_elbo_submit / vbmc_elbo_collect are for one device: submit stages this process's restarts, enqueues the passes, the one
 * ncclAllGather and the copy of the gathered vectors to pinned host memory and returns; collect waits and fills the caller's arrays
 * (same contents as vbmc_elbo_batch_multi).  slot = 0 .. 3: up to four batches in flight, on two streams per device (the exchange itself
 * stays on each context's own stream, ordered after the pass).  Device RNG (eps_mode 0), no per-component or
 * per-hyper-sample outputs.  A steady-state call allocates nothing: the per-device argument structs, staging vectors, exchange
 * blocks and the pinned landing block live in the communicator's slot.  A failure local to one rank (resource error, missing
 * surrogate) is reported AFTER the rank has entered the collective with an all-NaN block, so that the other ranks are never left
 * waiting (vbmc_elbo_batch_multi does the same). */
vbmc_status vbmc_elbo_multi_submit(vbmc_comm* comm, const vbmc_gp* const* gps, const vbmc_elbo_args* args, int slot);
vbmc_status vbmc_elbo_multi_collect(vbmc_comm* comm, const vbmc_elbo_args* args, int slot);

/*
 * [x,f,xtab,ftab,iter] = fminadam(@(t) negelcbo_vbmc(t,beta,vp,gp,Ns,1,compute_var,~,thetabnd), x0, [], [],
 *                                 TolFun, MaxIter, master_stepsize)          (utils/fminadam.m:1-104,
 * call site misc/vpoptimize_vbmc.m:127) for R chains in lock-step, entirely on the device: per
 * iteration one batched ELBO+grad pass and the Adam update (:48-61), every 20 iterations the slope /
 * random-walk stopping test (:65-81); the host only polls R flags on those iterations.  args->theta
 * holds the R starting points x0 (T x R); args->seed + iter keys the MC draws of iteration iter
 * (eps_mode must be 0).  Outputs (any may be NULL): x T x R (mean of the last 20 iterates, :96),
 * f R (:97), iters R, xtab T x MaxIter x R and ftab MaxIter x R (first iters(r) entries filled), xmid T x R: per chain the
 * iterate with the smallest recorded objective, theta_lst(idx_mid,:) with [~,idx_mid] = min(fval_lst) of
 * misc/vpoptimize_vbmc.m:133 -- the one thing that caller reads from the tables, so that it need not ask for them (T x MaxIter x R
 * doubles with MaxIter = 1e4).
 */
vbmc_status vbmc_adam_batch(vbmc_ctx* ctx, const vbmc_gp* gp, const vbmc_elbo_args* args, double TolFun, int MaxIter,
                            double step_min, double step_max, double step_decay, double* x, double* f, int32_t* iters,
                            double* xtab, double* ftab, double* xmid);

/* Writes the exact standard-normal block eps (D x Ns/2 x K x R) that eps_mode 0 consumes for
 * `seed`, so that a host oracle can be fed the same draws (test hook; entmc_vbmc.m:53). */
vbmc_status vbmc_rng_dump(vbmc_ctx* ctx, int D, int K, int R, int Ns, uint64_t seed, double* eps_host);

/* Reporting hook: which instantiation of the Monte-Carlo entropy kernel (vbmc_amd/csrc/entropy_mfma.h) a D-dimensional,
 * K-component mixture runs on in dense mode: qs = ceil((D+2)/4), kt = 16-component k-tiles per wave, hv = waves per workgroup,
 * tail = values per lane of the component tail (0: none).  Returns 1; 2 when the shape is in the small class (K <= 16, D <= 12) and
 * runs on the lane-per-sample kernel (vbmc_amd/csrc/entropy_lane.h: qs = padded dimension DT, kt = padded component count KP,
 * hv = waves per workgroup); 0 when the plain VALU kernel serves it.  bench.py labels the kernel it measures with it. */
int vbmc_entropy_plan(int D, int K, int* qs, int* kt, int* hv, int* tail);

/* Test hook: y = exp(x) evaluated by the hot-loop device implementations (0: polynomial, 1 / 2: 256-entry table with the two- /
 * one-constant reduction, 3: the Monte-Carlo entropy kernel's exponential as built, 4 / 5: its cubic / quadratic form). */
vbmc_status vbmc_test_exp(vbmc_ctx* ctx, int n, int variant, const double* x, double* y);

/* Device-memory helpers for callers without their own allocator (MEX). */
vbmc_status vbmc_device_alloc(vbmc_ctx* ctx, size_t bytes, void** dptr);
vbmc_status vbmc_device_free(vbmc_ctx* ctx, void* dptr);
vbmc_status vbmc_memcpy_h2d(vbmc_ctx* ctx, void* dst, const void* src, size_t bytes);
vbmc_status vbmc_memcpy_d2h(vbmc_ctx* ctx, void* dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* VBMC_HIP_H */
