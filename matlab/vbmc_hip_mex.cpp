// vbmc_hip_mex.cpp -- MEX gateway from MATLAB to libvbmc_hip.so (include/vbmc_hip.h).
//
// Build (on a machine with MATLAB + ROCm; cannot be built in the development container, which has
// neither MATLAB nor mex.h -- this file is deliberately free of numerics):
//     mex -R2018a vbmc_hip_mex.cpp -I../include -L../vbmc_amd/lib -lvbmc_hip
//
// Usage from the .m shims in this directory:
//     vbmc_hip_mex('open', device)                         -> (context kept in a persistent, mexLock'ed)
//     h  = vbmc_hip_mex('gp_upload', gpstruct)             -> uint64 handle of a device-resident gp.post
//          vbmc_hip_mex('gp_free', h)
//     [F,dF,G,H,varG,dH,varGss,I_sk,J_sjk,dG,G_s,varG_s,dvarG,dG_s,dvarG_s] = vbmc_hip_mex('elbo', h, theta, vp, Ns, compute_grad,
//                                     compute_var, separate_K, beta, thetabnd_or_empty, eps_or_empty, seed, numel(gp.post), no_jacobian)
//                                     (G_s, varG_s, dG_s (T x S): the per-hyper-sample outputs of gplogjoint(...,avg_flag = 0); no_jacobian,
//                                     optional: 1 = gradients with respect to sigma, lambda, w themselves, the JACOBIAN_FLAG = 0
//                                     form of entmc_vbmc / entlb_vbmc / gplogjoint)
//     [F,dF,varG,G,H,varGss,I_sk,J_sjk] = vbmc_hip_mex('elbo_batch', h, Theta /*T x R*/, vp, Ns, compute_grad, compute_var, beta,
//                                thetabnd_or_empty, seed, separate_K, numel(gp.post))
//                                (the R candidates of vpsieve_vbmc.m:74-78, or the 2*Nslowopts eval_fullelcbo calls of
//                                 vpoptimize_vbmc.m:134,165, in one pass; I_sk is S x K x R, J_sjk S x K x K x R)
//     [x,f,iters,xmid,xtab,ftab] = vbmc_hip_mex('adam', h, Theta0 /*T x R*/, vp, Ns, compute_var, beta, thetabnd_or_empty, seed,
//                                TolFun, MaxIter, [step_min step_max step_decay])   (fminadam.m on the device; xmid T x R: each
//                                chain's iterate of smallest recorded objective, vpoptimize_vbmc.m:133; the tables only on
//                                request: xtab T x MaxIter x R, ftab MaxIter x R, the first iters(r) entries of chain r filled)
//     [alpha,L,sW,sn2_mult,Lchol,h] = vbmc_hip_mex('gp_post', hyp, X, y, s2, meanfun, noisefun)
//     [ymu,ys2,fmu,fs2] = vbmc_hip_mex('gp_pred', h, Xstar, s2star, ssflag)
//     [acq,fbar,vtot] = vbmc_hip_mex('acq', h, Xs, acq_id, vp, ymax, var_regularized, TolGPVar, gplengthscale, X_rescaled, sn2new)
//     his = vbmc_hip_mex('is_create', h, Xa, lnw_or_empty, fs2a_or_empty, Ctmp_or_empty)   (ActiveImportanceSampling state)
//           vbmc_hip_mex('is_free', his)
//     [acq,fbar,vtot] = vbmc_hip_mex('acq_iqr', h, his, Xs, gplengthscale, X_rescaled, sn2new, var_regularized, TolGPVar)
//     [nlZ,dnlZ] = vbmc_hip_mex('gp_nlz', Hyp /*Nhyp x B*/, X, y, s2, meanfun, noisefun)   (gplite_nlZ for B vectors)
//     C = vbmc_hip_mex('sq_dist', a, b)
//     lim = vbmc_hip_mex('limits')                          -> struct max_D, max_K, max_N, max_Na, max_T_vargrad, delta_ok, meanfun: the shapes the
//                                                             library accepts (vbmc_get_limits; no device needed) -- matlab/vbmc_hip_supported.m
//     n3 = vbmc_hip_mex('stats')                            -> [host uploads of a surrogate, device posteriors kept, rank-one appends kept]
// Every GPU of the node from ONE MATLAB process (the communicator inside the library, include/vbmc_hip.h):
//     n  = vbmc_hip_mex('comm_open', ndev)                 -> must be the first command of the session: a context and an RCCL rank
//                                                             per device; the single-device commands then run on device 0.
//                                                             VBMC_HIP_DEVICES=ndev in the environment does the same implicitly.
//     n  = vbmc_hip_mex('comm_size')                       -> devices of the session (1 without a communicator)
//     hs = vbmc_hip_mex('gp_upload_all', gpstruct)         -> 1 x n uint64 handles, one replica per device (hs(1): device 0)
//          vbmc_hip_mex('gp_free_all', hs)
//     [F,dF,varG,G,H,varGss,I_sk,J_sjk] = vbmc_hip_mex('elbo_batch_multi', hs, Theta, vp, Ns, compute_grad, compute_var, beta,
//                                thetabnd_or_empty, seed, separate_K, numel(gp.post))
//                                ('elbo_batch' with the R candidates dealt r = g (mod n) over the devices and their ELCBO values
//                                 all-gathered over xGMI; every value bit-identical to 'elbo_batch' on one device)
//
// Errors: mexErrMsgIdAndTxt long-jumps out of the MEX function without running C++ destructors, so it is called from
// exactly one place -- mexFunction itself, which owns no C++ object -- after dispatch() has RETURNED (all its
// std::vector / std::string temporaries destroyed) with the id and message parked in static character buffers.
// VBMC_ERR_UNSUPPORTED becomes the id 'vbmc_hip:unsupported' which the shims catch to fall through
// to the reference .m implementation (SURVEY.md 8b "Errors").
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "vbmc_hip.h"

static vbmc_ctx* g_ctx = nullptr;
static long long g_stats[3] = {0, 0, 0};   // 'stats': gp_upload(_all) commands, gp_post handles kept, gp_rank1 appends
static vbmc_comm* g_comm = nullptr;   // 'comm_open': owns one context per device; g_ctx is then its device-0 context

static void at_exit() {
  if (g_comm) { vbmc_comm_destroy(g_comm); g_comm = nullptr; g_ctx = nullptr; }
  if (g_ctx) { vbmc_ctx_destroy(g_ctx); g_ctx = nullptr; }
}

// pending error of the current call (plain static storage: survives the long jump, owns nothing)
static char g_err_id[96];
static char g_err_msg[640];

static int raise(const char* id, const char* msg) {
  snprintf(g_err_id, sizeof g_err_id, "%s", id);
  snprintf(g_err_msg, sizeof g_err_msg, "%s", msg);
  return 1;
}

// library status -> pending MATLAB error; returns nonzero so that call sites read `if (st != VBMC_OK) return fail(st);`
static int fail(vbmc_status st) {
  const char* msg = g_ctx ? vbmc_last_error(g_ctx) : "no context";
  if (st == VBMC_ERR_UNSUPPORTED) return raise("vbmc_hip:unsupported", msg);
  // messages of INVALID errors start with the reference's own error id where one exists ("gplogjoint:FullVarianceGradient ...")
  const char* sp = strchr(msg, ' ');
  const char* col = strchr(msg, ':');
  if (st == VBMC_ERR_INVALID && sp && col && col < sp && (size_t)(sp - msg) < sizeof g_err_id) {
    memcpy(g_err_id, msg, (size_t)(sp - msg));
    g_err_id[sp - msg] = 0;
    snprintf(g_err_msg, sizeof g_err_msg, "%s", sp + 1);
    return 1;
  }
  return raise("vbmc_hip:error", msg);
}

// status of a communicator call -> pending MATLAB error
static int fail_comm(vbmc_status st) {
  const char* msg = g_comm ? vbmc_comm_last_error(g_comm) : "no communicator";
  return raise(st == VBMC_ERR_UNSUPPORTED ? "vbmc_hip:unsupported" : "vbmc_hip:error", msg);
}

static int ensure_ctx(int device) {
  if (g_ctx) return 0;
  // VBMC_HIP_DEVICES=n (n >= 2) in the environment: the session opens on n devices, as 'comm_open' n would
  const char* nd = getenv("VBMC_HIP_DEVICES");
  if (nd && atoi(nd) >= 2) {
    vbmc_status st = vbmc_comm_create_all(atoi(nd), nullptr, &g_comm);
    if (st != VBMC_OK) { g_comm = nullptr; return raise("vbmc_hip:nodevice", "libvbmc_hip: VBMC_HIP_DEVICES asks for more gfx950 devices than could be opened (or librccl is missing)"); }
    g_ctx = vbmc_comm_ctx(g_comm, 0);
    mexLock();
    mexAtExit(at_exit);
    return 0;
  }
  vbmc_status st = vbmc_ctx_create(device, nullptr, &g_ctx);
  if (st != VBMC_OK) return raise("vbmc_hip:nodevice", "libvbmc_hip: no gfx950 (MI355X) device available");
  mexLock();
  mexAtExit(at_exit);
  return 0;
}

// device handles travel as uint64 scalars; anything else (a double that lost its class on the way) must not be dereferenced
static bool is_handle(const mxArray* a) { return a && mxGetClassID(a) == mxUINT64_CLASS && mxGetNumberOfElements(a) >= 1; }
static const double* dbl(const mxArray* a) { return (a && !mxIsEmpty(a)) ? mxGetDoubles(a) : nullptr; }
static const mxArray* field(const mxArray* s, const char* name) { return mxGetField(s, 0, name); }
static double scalar_field(const mxArray* s, const char* name, double dflt) {
  const mxArray* f = field(s, name);
  return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}

// vp struct + thetabnd -> the parameter-layout part of vbmc_elbo_args (shared by 'elbo', 'elbo_batch', 'adam')
static void fill_vp_args(vbmc_elbo_args& a, const mxArray* vp, const mxArray* tb, std::vector<double>& delta) {
  memset(&a, 0, sizeof a);
  a.struct_size = sizeof a;
  a.D = (int)scalar_field(vp, "D", 0); a.K = (int)scalar_field(vp, "K", 0); a.R = 1;
  a.optimize[0] = scalar_field(vp, "optimize_mu", 1) != 0; a.optimize[1] = scalar_field(vp, "optimize_sigma", 1) != 0;
  a.optimize[2] = scalar_field(vp, "optimize_lambda", 1) != 0; a.optimize[3] = scalar_field(vp, "optimize_weights", 0) != 0;
  a.vp_mu = dbl(field(vp, "mu")); a.vp_sigma = dbl(field(vp, "sigma")); a.vp_lambda = dbl(field(vp, "lambda")); a.vp_w = dbl(field(vp, "w"));
  const mxArray* dl = field(vp, "delta");
  if (dl && !mxIsEmpty(dl)) {  // scalar or D-vector (gplogjoint.m:85-89)
    delta.assign(a.D, mxGetDoubles(dl)[0]);
    if ((int)mxGetNumberOfElements(dl) == a.D) memcpy(delta.data(), mxGetDoubles(dl), a.D * sizeof(double));
    a.vp_delta = delta.data();
  }
  if (tb && !mxIsEmpty(tb)) {
    a.bnd_lb = dbl(field(tb, "lb")); a.bnd_ub = dbl(field(tb, "ub")); a.TolCon = scalar_field(tb, "TolCon", 0.01);
    a.WeightThreshold = scalar_field(tb, "WeightThreshold", 0); a.WeightPenalty = scalar_field(tb, "WeightPenalty", 0);
  }
  { const char* sc = getenv("VBMC_HIP_SPARSE_CUTOFF"); a.sparse_cutoff = sc ? atof(sc) : 0.0; }
}

// gp struct (gplite_post.m:94-157) -> the flat arrays of vbmc_gp_upload; returns the sizes
struct GpArrays {
  int N = 0, D = 0, S = 0, Nhyp = 0, Ncov = 0, Nnoise = 0, meanfun = 0;
  const double* X = nullptr;
  std::vector<double> hyp, alpha, L, sW1, mult;
  std::vector<uint8_t> lch;
  int32_t nf[3] = {1, 0, 0};
};
static void read_gp(const mxArray* gp, GpArrays& g) {
  const mxArray* X = field(gp, "X");
  const mxArray* post = field(gp, "post");
  g.N = (int)mxGetM(X); g.D = (int)mxGetN(X); g.S = (int)mxGetNumberOfElements(post);
  g.Nhyp = (int)mxGetNumberOfElements(mxGetField(post, 0, "hyp"));
  g.X = mxGetDoubles(X);
  const int N = g.N, S = g.S, Nhyp = g.Nhyp;
  g.hyp.resize((size_t)Nhyp * S); g.alpha.resize((size_t)N * S); g.L.resize((size_t)N * N * S); g.sW1.resize(S); g.mult.resize(S);
  g.lch.resize(S);
  for (int s = 0; s < S; ++s) {
    memcpy(&g.hyp[(size_t)s * Nhyp], mxGetDoubles(mxGetField(post, s, "hyp")), Nhyp * sizeof(double));
    memcpy(&g.alpha[(size_t)s * N], mxGetDoubles(mxGetField(post, s, "alpha")), N * sizeof(double));
    memcpy(&g.L[(size_t)s * N * N], mxGetDoubles(mxGetField(post, s, "L")), (size_t)N * N * sizeof(double));
    g.sW1[s] = mxGetDoubles(mxGetField(post, s, "sW"))[0];
    g.mult[s] = mxGetScalar(mxGetField(post, s, "sn2_mult"));
    g.lch[s] = mxIsLogicalScalarTrue(mxGetField(post, s, "Lchol")) ? 1 : 0;
  }
  const mxArray* nfa = field(gp, "noisefun");
  for (int i = 0; nfa && i < 3 && i < (int)mxGetNumberOfElements(nfa); ++i) g.nf[i] = (int32_t)mxGetDoubles(nfa)[i];
  g.Ncov = (int)scalar_field(gp, "Ncov", g.D + 1); g.Nnoise = (int)scalar_field(gp, "Nnoise", 1);
  g.meanfun = (int)scalar_field(gp, "meanfun", 4);
}

// Every command; returns 0 on success, nonzero with g_err_id / g_err_msg set.  All C++ objects live in here.
static int dispatch(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 1 || !mxIsChar(prhs[0])) return raise("vbmc_hip:usage", "first argument must be a command string");
  char cmd[32];
  mxGetString(prhs[0], cmd, sizeof cmd);

  if (!strcmp(cmd, "limits")) {     // the shapes the library accepts (vbmc_get_limits: a host function, no device): struct for vbmc_hip_supported.m
    vbmc_limits lim;
    lim.struct_size = sizeof lim;
    if (vbmc_get_limits(&lim) != VBMC_OK) return raise("vbmc_hip:abi", "vbmc_get_limits refused the struct (ABI mismatch)");
    const char* names[] = {"max_D", "max_K", "max_N", "max_Na", "max_T_vargrad", "delta_ok", "meanfun"};
    plhs[0] = mxCreateStructMatrix(1, 1, 7, names);
    const double v[6] = {(double)lim.max_D, (double)lim.max_K, (double)lim.max_N, (double)lim.max_Na, (double)lim.max_T_vargrad, (double)lim.delta_ok};
    for (int i = 0; i < 6; ++i) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); mxGetDoubles(a)[0] = v[i]; mxSetField(plhs[0], 0, names[i], a); }
    int nm = 0;
    for (int i = 0; i < 31; ++i) nm += (lim.meanfun_mask >> i) & 1;
    mxArray* mf = mxCreateDoubleMatrix(1, nm, mxREAL);
    for (int i = 0, j = 0; i < 31; ++i) if ((lim.meanfun_mask >> i) & 1) mxGetDoubles(mf)[j++] = i;
    mxSetField(plhs[0], 0, "meanfun", mf);
    return 0;
  }
  if (!strcmp(cmd, "stats")) {      // [surrogates uploaded from the host, posteriors built on the device, rank-one appends] since the gateway was loaded
    plhs[0] = mxCreateDoubleMatrix(1, 3, mxREAL);
    for (int i = 0; i < 3; ++i) mxGetDoubles(plhs[0])[i] = (double)g_stats[i];
    return 0;
  }
  if (!strcmp(cmd, "open")) return ensure_ctx(nrhs > 1 ? (int)mxGetScalar(prhs[1]) : 0);
  if (!strcmp(cmd, "comm_open")) {
    if (g_comm || g_ctx) return raise("vbmc_hip:usage", "comm_open must be the first command of the session");
    const int ndev = nrhs > 1 ? (int)mxGetScalar(prhs[1]) : 1;
    vbmc_status st = vbmc_comm_create_all(ndev, nullptr, &g_comm);
    if (st != VBMC_OK) { g_comm = nullptr; return raise("vbmc_hip:nodevice", "libvbmc_hip: could not open the requested gfx950 devices / librccl"); }
    g_ctx = vbmc_comm_ctx(g_comm, 0);
    mexLock();
    mexAtExit(at_exit);
    plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
    mxGetDoubles(plhs[0])[0] = vbmc_comm_size(g_comm);
    return 0;
  }
  if (ensure_ctx(0)) return 1;
  {  // commands whose first argument is a device handle (the IQR evaluation takes two)
    const char* with_handle[] = {"gp_free", "elbo", "elbo_batch", "elbo_batch_multi", "adam", "gp_rank1", "acq", "is_create", "is_free",
                                 "acq_iqr", "gp_pred", "gp_free_all"};
    for (const char* w : with_handle)
      if (!strcmp(cmd, w) && (nrhs < 2 || !is_handle(prhs[1]) || (!strcmp(cmd, "acq_iqr") && (nrhs < 3 || !is_handle(prhs[2])))))
        return raise("vbmc_hip:usage", "this command takes a uint64 device handle as its first argument");
  }

  if (!strcmp(cmd, "gp_upload")) {
    GpArrays g;
    read_gp(prhs[1], g);
    vbmc_gp* h = nullptr;
    vbmc_status st = vbmc_gp_upload(g_ctx, g.N, g.D, g.S, g.Nhyp, g.Ncov, g.Nnoise, g.meanfun, g.X, g.hyp.data(), g.alpha.data(),
                                    g.L.data(), g.sW1.data(), g.lch.data(), &h);
    if (st == VBMC_OK) st = vbmc_gp_set_noise(g_ctx, h, g.nf, g.mult.data());
    if (st != VBMC_OK) return fail(st);
    ++g_stats[0];
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)h;
    return 0;
  }
  if (!strcmp(cmd, "comm_size")) { plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL); mxGetDoubles(plhs[0])[0] = g_comm ? vbmc_comm_size(g_comm) : 1; return 0; }
  if (!strcmp(cmd, "gp_upload_all")) {
    if (!g_comm) return raise("vbmc_hip:usage", "gp_upload_all needs 'comm_open' first");
    GpArrays g;
    read_gp(prhs[1], g);
    const int n = vbmc_comm_local(g_comm);
    std::vector<vbmc_gp*> hs(n, nullptr);
    vbmc_status st = vbmc_gp_upload_all(g_comm, g.N, g.D, g.S, g.Nhyp, g.Ncov, g.Nnoise, g.meanfun, g.X, g.hyp.data(), g.alpha.data(),
                                        g.L.data(), g.sW1.data(), g.lch.data(), hs.data());
    if (st != VBMC_OK) return fail_comm(st);
    for (int i = 0; i < n && st == VBMC_OK; ++i) st = vbmc_gp_set_noise(vbmc_comm_ctx(g_comm, i), hs[i], g.nf, g.mult.data());
    if (st != VBMC_OK) { vbmc_gp_free_all(g_comm, hs.data()); return raise("vbmc_hip:error", "vbmc_gp_set_noise failed on a replica"); }
    ++g_stats[0];
    plhs[0] = mxCreateNumericMatrix(1, n, mxUINT64_CLASS, mxREAL);
    for (int i = 0; i < n; ++i) ((uint64_t*)mxGetData(plhs[0]))[i] = (uint64_t)(uintptr_t)hs[i];
    return 0;
  }
  if (!strcmp(cmd, "gp_free_all")) {
    if (!g_comm) return 0;
    const int n = vbmc_comm_local(g_comm);
    std::vector<vbmc_gp*> hs(n, nullptr);
    for (int i = 0; i < n && i < (int)mxGetNumberOfElements(prhs[1]); ++i) hs[i] = (vbmc_gp*)(uintptr_t)((uint64_t*)mxGetData(prhs[1]))[i];
    vbmc_gp_free_all(g_comm, hs.data());
    return 0;
  }
  if (!strcmp(cmd, "gp_free")) { vbmc_gp_free(g_ctx, (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]))); return 0; }

  if (!strcmp(cmd, "elbo")) {
    // (h, theta, vp, Ns, compute_grad, compute_var, separate_K, beta, thetabnd, eps)
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    const mxArray* theta = prhs[2];
    const mxArray* vp = prhs[3];
    vbmc_elbo_args a;
    std::vector<double> delta;
    fill_vp_args(a, vp, nrhs > 9 ? prhs[9] : nullptr, delta);
    a.theta = mxGetDoubles(theta);
    a.Ns = (int)mxGetScalar(prhs[4]);
    a.compute_grad = (int)mxGetScalar(prhs[5]); a.compute_var = (int)mxGetScalar(prhs[6]); a.separate_K = (int)mxGetScalar(prhs[7]);
    a.beta = mxGetScalar(prhs[8]);
    const mxArray* eps = nrhs > 10 ? prhs[10] : nullptr;  // D x Ns/2 x K block drawn by the shim with randn, or []
    if (eps && !mxIsEmpty(eps)) { a.eps_mode = 1; a.eps = mxGetDoubles(eps); a.eps_shared = 1; }
    else { a.eps_mode = 0; a.seed = (uint64_t)(nrhs > 11 ? mxGetScalar(prhs[11]) : 0); }
    a.no_jacobian = (nrhs > 13 && !mxIsEmpty(prhs[13])) ? (mxGetScalar(prhs[13]) != 0) : 0;
    const size_t T = mxGetNumberOfElements(theta);
    mxArray *F = mxCreateDoubleMatrix(1, 1, mxREAL), *dF = mxCreateDoubleMatrix(a.compute_grad ? T : 0, a.compute_grad ? 1 : 0, mxREAL);
    mxArray *G = mxCreateDoubleMatrix(1, 1, mxREAL), *H = mxCreateDoubleMatrix(1, 1, mxREAL), *vG = mxCreateDoubleMatrix(1, 1, mxREAL);
    mxArray *dH = mxCreateDoubleMatrix(a.compute_grad ? T : 0, a.compute_grad ? 1 : 0, mxREAL), *vss = mxCreateDoubleMatrix(1, 1, mxREAL);
    a.F = mxGetDoubles(F); a.G = mxGetDoubles(G); a.H = mxGetDoubles(H); a.varG = mxGetDoubles(vG); a.varGss = mxGetDoubles(vss);
    mxArray* dG = mxCreateDoubleMatrix(a.compute_grad ? T : 0, a.compute_grad ? 1 : 0, mxREAL);  // 10th output (gplogjoint shim)
    if (a.compute_grad) { a.dF = mxGetDoubles(dF); a.dH = mxGetDoubles(dH); a.dG = mxGetDoubles(dG); }
    mxArray *Isk = nullptr, *Jsjk = nullptr;
    if (a.separate_K) {
      // S is known to the library; query through a first call would cost a launch, so the shim passes numel(gp.post)
      const int S = (int)mxGetScalar(prhs[12]);
      Isk = mxCreateDoubleMatrix(S, a.K, mxREAL);
      a.I_sk = mxGetDoubles(Isk);
      if (a.compute_var) { mwSize dims[3] = {(mwSize)S, (mwSize)a.K, (mwSize)a.K}; Jsjk = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxREAL); a.J_sjk = mxGetDoubles(Jsjk); }
    }
    mxArray *Gs = nullptr, *vGs = nullptr;
    if (nlhs > 10 && nrhs > 12) {   // per-hyper-sample values (gplogjoint avg_flag = 0)
      const int S = (int)mxGetScalar(prhs[12]);
      Gs = mxCreateDoubleMatrix(1, S, mxREAL);
      a.G_s = mxGetDoubles(Gs);
      if (nlhs > 11 && a.compute_var) { vGs = mxCreateDoubleMatrix(1, S, mxREAL); a.varG_s = mxGetDoubles(vGs); }
    }
    mxArray* dvG = nullptr;         // 13th output: gradient of the diagonal variance (dvarF of misc/gplogjoint.m:27)
    if (nlhs > 12 && a.compute_grad && a.compute_var == 2) { dvG = mxCreateDoubleMatrix(T, 1, mxREAL); a.dvarG = mxGetDoubles(dvG); }
    mxArray* dGs = nullptr;         // 14th output: the gradient per hyper-sample, T x S (gplogjoint's dF with avg_flag = 0, misc/gplogjoint.m:411 skipped)
    if (nlhs > 13 && a.compute_grad && nrhs > 12) {
      const int S = (int)mxGetScalar(prhs[12]);
      dGs = mxCreateDoubleMatrix(T, S, mxREAL);
      a.dG_s = mxGetDoubles(dGs);
    }
    mxArray* dvGs = nullptr;        // 15th output (ABI 5): the variance gradient per hyper-sample, T x S (gplogjoint's dvarF with avg_flag = 0, :407-409 skipped)
    if (nlhs > 14 && a.compute_grad && a.compute_var == 2 && nrhs > 12) {
      const int S = (int)mxGetScalar(prhs[12]);
      dvGs = mxCreateDoubleMatrix(T, S, mxREAL);
      a.dvarG_s = mxGetDoubles(dvGs);
    }
    vbmc_status st = vbmc_elbo_batch(g_ctx, h, &a);
    if (st != VBMC_OK) return fail(st);  // MATLAB frees the mxArrays created above on error
    mxArray* outs[15] = {F, dF, G, H, vG, dH, vss, Isk, Jsjk, dG, Gs, vGs, dvG, dGs, dvGs};
    for (int i = 0; i < 15 && (i < nlhs || i == 0); ++i) plhs[i] = outs[i] ? outs[i] : mxCreateDoubleMatrix(0, 0, mxREAL);
    return 0;
  }

  const bool multi = !strcmp(cmd, "elbo_batch_multi");
  if (multi && !g_comm) return raise("vbmc_hip:usage", "elbo_batch_multi needs 'comm_open' first");
  if (multi || !strcmp(cmd, "elbo_batch")) {
    // (h, Theta, vp, Ns, compute_grad, compute_var, beta, thetabnd, seed): R = size(Theta,2) candidates that share
    // vp's flags and its non-optimised groups; device MC stream keyed by (seed, r).  'elbo_batch_multi': h is the 1 x n handle
    // vector of 'gp_upload_all' and the candidates are dealt over the n devices.
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    std::vector<const vbmc_gp*> hs;
    if (multi) {
      if ((int)mxGetNumberOfElements(prhs[1]) != vbmc_comm_local(g_comm)) return raise("vbmc_hip:usage", "elbo_batch_multi: one handle per device");
      for (int i = 0; i < vbmc_comm_local(g_comm); ++i) hs.push_back((const vbmc_gp*)(uintptr_t)((uint64_t*)mxGetData(prhs[1]))[i]);
    }
    const mxArray* Theta = prhs[2];
    vbmc_elbo_args a;
    std::vector<double> delta;
    fill_vp_args(a, prhs[3], nrhs > 8 ? prhs[8] : nullptr, delta);
    const size_t T = mxGetM(Theta);
    a.R = (int)mxGetN(Theta);
    a.theta = mxGetDoubles(Theta);
    a.Ns = (int)mxGetScalar(prhs[4]); a.compute_grad = (int)mxGetScalar(prhs[5]); a.compute_var = (int)mxGetScalar(prhs[6]);
    a.beta = mxGetScalar(prhs[7]);
    a.eps_mode = 0; a.seed = (uint64_t)(nrhs > 9 ? mxGetScalar(prhs[9]) : 0);
    mxArray *F = mxCreateDoubleMatrix(1, a.R, mxREAL), *dF = mxCreateDoubleMatrix(a.compute_grad ? T : 0, a.compute_grad ? a.R : 0, mxREAL);
    mxArray* vG = mxCreateDoubleMatrix(1, a.R, mxREAL);
    a.F = mxGetDoubles(F); a.varG = mxGetDoubles(vG);
    if (a.compute_grad) a.dF = mxGetDoubles(dF);
    mxArray *G = nullptr, *H = nullptr, *vss = nullptr, *Isk = nullptr, *Jsjk = nullptr;
    if (nlhs > 3) { G = mxCreateDoubleMatrix(1, a.R, mxREAL); a.G = mxGetDoubles(G); }
    if (nlhs > 4) { H = mxCreateDoubleMatrix(1, a.R, mxREAL); a.H = mxGetDoubles(H); }
    if (nlhs > 5) { vss = mxCreateDoubleMatrix(1, a.R, mxREAL); a.varGss = mxGetDoubles(vss); }
    a.separate_K = (nrhs > 10 && nlhs > 6) ? (int)mxGetScalar(prhs[10]) : 0;
    if (a.separate_K) {
      if (nrhs < 12) return raise("vbmc_hip:usage", "elbo_batch with separate_K needs numel(gp.post) as its 12th argument");
      const mwSize S = (mwSize)mxGetScalar(prhs[11]);
      mwSize d3[3] = {S, (mwSize)a.K, (mwSize)a.R}, d4[4] = {S, (mwSize)a.K, (mwSize)a.K, (mwSize)a.R};
      Isk = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL);
      a.I_sk = mxGetDoubles(Isk);
      if (a.compute_var && nlhs > 7) { Jsjk = mxCreateNumericArray(4, d4, mxDOUBLE_CLASS, mxREAL); a.J_sjk = mxGetDoubles(Jsjk); }
    }
    if (multi) {
      vbmc_status st = vbmc_elbo_batch_multi(g_comm, hs.data(), &a);
      if (st != VBMC_OK) return fail_comm(st);
    } else {
      vbmc_status st = vbmc_elbo_batch(g_ctx, h, &a);
      if (st != VBMC_OK) return fail(st);
    }
    mxArray* outs[8] = {F, dF, vG, G, H, vss, Isk, Jsjk};
    for (int i = 0; i < 8 && (i < nlhs || i == 0); ++i) plhs[i] = outs[i] ? outs[i] : mxCreateDoubleMatrix(0, 0, mxREAL);
    return 0;
  }

  if (!strcmp(cmd, "adam")) {
    // (h, Theta0, vp, Ns, compute_var, beta, thetabnd, seed, TolFun, MaxIter, [step_min step_max step_decay])
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    const mxArray* Theta = prhs[2];
    vbmc_elbo_args a;
    std::vector<double> delta;
    fill_vp_args(a, prhs[3], nrhs > 7 ? prhs[7] : nullptr, delta);
    const size_t T = mxGetM(Theta);
    a.R = (int)mxGetN(Theta);
    a.theta = mxGetDoubles(Theta);
    a.Ns = (int)mxGetScalar(prhs[4]); a.compute_grad = 1; a.compute_var = (int)mxGetScalar(prhs[5]); a.beta = mxGetScalar(prhs[6]);
    a.eps_mode = 0; a.seed = (uint64_t)mxGetScalar(prhs[8]);
    const double TolFun = mxGetScalar(prhs[9]);
    const int MaxIter = (int)mxGetScalar(prhs[10]);
    double step[3] = {0.001, 0.1, 200.0};  // fminadam.m:28-31 defaults
    for (int i = 0; nrhs > 11 && i < 3 && i < (int)mxGetNumberOfElements(prhs[11]); ++i) step[i] = mxGetDoubles(prhs[11])[i];
    mxArray *x = mxCreateDoubleMatrix(T, a.R, mxREAL), *f = mxCreateDoubleMatrix(1, a.R, mxREAL);
    mxArray* it = mxCreateNumericMatrix(1, a.R, mxINT32_CLASS, mxREAL);
    mxArray *xmid = nullptr, *xtab = nullptr, *ftab = nullptr;
    if (nlhs > 3) xmid = mxCreateDoubleMatrix(T, a.R, mxREAL);
    if (nlhs > 4) { mwSize d3[3] = {(mwSize)T, (mwSize)MaxIter, (mwSize)a.R}; xtab = mxCreateNumericArray(3, d3, mxDOUBLE_CLASS, mxREAL); }
    if (nlhs > 5) ftab = mxCreateDoubleMatrix(MaxIter, a.R, mxREAL);
    vbmc_status st = vbmc_adam_batch(g_ctx, h, &a, TolFun, MaxIter, step[0], step[1], step[2], mxGetDoubles(x), mxGetDoubles(f),
                                     (int32_t*)mxGetData(it), xtab ? mxGetDoubles(xtab) : nullptr, ftab ? mxGetDoubles(ftab) : nullptr,
                                     xmid ? mxGetDoubles(xmid) : nullptr);
    if (st != VBMC_OK) return fail(st);
    plhs[0] = x;
    if (nlhs > 1) plhs[1] = f;
    if (nlhs > 2) plhs[2] = it;
    if (nlhs > 3) plhs[3] = xmid;
    if (nlhs > 4) plhs[4] = xtab;
    if (nlhs > 5) plhs[5] = ftab;
    return 0;
  }

  if (!strcmp(cmd, "gp_post")) {
    // (hyp, X, y, s2, meanfun, noisefun) -> alpha, L, sW, sn2_mult, Lchol, handle
    const mxArray *hyp = prhs[1], *X = prhs[2], *y = prhs[3], *s2 = prhs[4];
    const int N = (int)mxGetM(X), D = (int)mxGetN(X), Nhyp = (int)mxGetM(hyp), S = (int)mxGetN(hyp);
    int32_t nf[3] = {1, 0, 0};
    for (int i = 0; i < 3 && i < (int)mxGetNumberOfElements(prhs[6]); ++i) nf[i] = (int32_t)mxGetDoubles(prhs[6])[i];
    mwSize ld[3] = {(mwSize)N, (mwSize)N, (mwSize)S};
    plhs[0] = mxCreateDoubleMatrix(N, S, mxREAL);
    mxArray* L = mxCreateNumericArray(3, ld, mxDOUBLE_CLASS, mxREAL);
    mxArray* sW = mxCreateDoubleMatrix(N, S, mxREAL);
    mxArray* mult = mxCreateDoubleMatrix(S, 1, mxREAL);
    mxArray* lch = mxCreateNumericMatrix(S, 1, mxUINT8_CLASS, mxREAL);
    vbmc_gp* h = nullptr;
    vbmc_status st = vbmc_gp_post(g_ctx, N, D, S, Nhyp, (int)mxGetScalar(prhs[5]), nf, mxGetDoubles(X), mxGetDoubles(y), dbl(s2),
                                  mxGetDoubles(hyp), mxGetDoubles(plhs[0]), mxGetDoubles(L), mxGetDoubles(sW), mxGetDoubles(mult),
                                  (uint8_t*)mxGetData(lch), &h);
    if (st != VBMC_OK) return fail(st);
    if (nlhs > 1) plhs[1] = L;
    if (nlhs > 2) plhs[2] = sW;
    if (nlhs > 3) plhs[3] = mult;
    if (nlhs > 4) plhs[4] = lch;
    if (nlhs > 5) { plhs[5] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t*)mxGetData(plhs[5]) = (uint64_t)(uintptr_t)h; ++g_stats[1]; }
    else vbmc_gp_free(g_ctx, h);
    return 0;
  }

  if (!strcmp(cmd, "gp_rank1")) {
    // (h, Xnew, ystar, mstar | [], vstar | [], sn2_eff) -> alpha (N+1 x S), L (N+1 x N+1 x S), new handle
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    const mxArray* Xn = prhs[2];
    const int N1 = (int)mxGetM(Xn), S = (int)mxGetNumberOfElements(prhs[6]);
    mwSize ld[3] = {(mwSize)N1, (mwSize)N1, (mwSize)S};
    plhs[0] = mxCreateDoubleMatrix(N1, S, mxREAL);
    mxArray* L = mxCreateNumericArray(3, ld, mxDOUBLE_CLASS, mxREAL);
    vbmc_gp* hn = nullptr;
    vbmc_status st = vbmc_gp_rank1_update(g_ctx, h, mxGetDoubles(Xn), mxGetScalar(prhs[3]), dbl(prhs[4]), dbl(prhs[5]),
                                          mxGetDoubles(prhs[6]), mxGetDoubles(plhs[0]), nlhs > 1 ? mxGetDoubles(L) : nullptr, &hn);
    if (st != VBMC_OK) return fail(st);
    if (nlhs > 1) plhs[1] = L;
    if (nlhs > 2) { plhs[2] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t*)mxGetData(plhs[2]) = (uint64_t)(uintptr_t)hn; ++g_stats[2]; }
    else vbmc_gp_free(g_ctx, hn);
    return 0;
  }

  if (!strcmp(cmd, "acq")) {
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    const mxArray *Xs = prhs[2], *vp = prhs[4];
    const int Nstar = (int)mxGetM(Xs);
    plhs[0] = mxCreateDoubleMatrix(Nstar, 1, mxREAL);
    mxArray *fb = mxCreateDoubleMatrix(Nstar, 1, mxREAL), *vt = mxCreateDoubleMatrix(Nstar, 1, mxREAL);
    vbmc_status st = vbmc_acq_eval(g_ctx, h, Nstar, mxGetDoubles(Xs), (int)mxGetScalar(prhs[3]), (int)scalar_field(vp, "K", 0),
                                   dbl(field(vp, "mu")), dbl(field(vp, "sigma")), dbl(field(vp, "lambda")), dbl(field(vp, "w")),
                                   mxGetScalar(prhs[5]), (int)mxGetScalar(prhs[6]), mxGetScalar(prhs[7]),
                                   nrhs > 8 ? dbl(prhs[8]) : nullptr, nrhs > 9 ? dbl(prhs[9]) : nullptr, nrhs > 10 ? dbl(prhs[10]) : nullptr,
                                   mxGetDoubles(plhs[0]), mxGetDoubles(fb), mxGetDoubles(vt));
    if (st != VBMC_OK) return fail(st);
    if (nlhs > 1) plhs[1] = fb;
    if (nlhs > 2) plhs[2] = vt;
    return 0;
  }

  if (!strcmp(cmd, "is_create")) {
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    const mxArray* Xa = prhs[2];
    const mwSize nd = mxGetNumberOfDimensions(Xa);
    const int Na = (int)mxGetDimensions(Xa)[0];
    vbmc_acq_is* is = nullptr;
    vbmc_status st = vbmc_acq_is_create(g_ctx, h, Na, mxGetDoubles(Xa), nd > 2 ? 1 : 0, nrhs > 3 ? dbl(prhs[3]) : nullptr,
                                        nrhs > 4 ? dbl(prhs[4]) : nullptr, nrhs > 5 ? dbl(prhs[5]) : nullptr, &is);
    if (st != VBMC_OK) return fail(st);
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)is;
    return 0;
  }
  if (!strcmp(cmd, "is_free")) { vbmc_acq_is_free(g_ctx, (vbmc_acq_is*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]))); return 0; }

  if (!strcmp(cmd, "acq_iqr")) {
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    vbmc_acq_is* is = (vbmc_acq_is*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[2]));
    const mxArray* Xs = prhs[3];
    const int Nstar = (int)mxGetM(Xs);
    plhs[0] = mxCreateDoubleMatrix(Nstar, 1, mxREAL);
    mxArray *fb = mxCreateDoubleMatrix(Nstar, 1, mxREAL), *vt = mxCreateDoubleMatrix(Nstar, 1, mxREAL);
    vbmc_status st = vbmc_acq_iqr_eval(g_ctx, h, is, Nstar, mxGetDoubles(Xs), dbl(prhs[4]), dbl(prhs[5]), dbl(prhs[6]),
                                       (int)mxGetScalar(prhs[7]), mxGetScalar(prhs[8]), mxGetDoubles(plhs[0]), mxGetDoubles(fb), mxGetDoubles(vt));
    if (st != VBMC_OK) return fail(st);
    if (nlhs > 1) plhs[1] = fb;
    if (nlhs > 2) plhs[2] = vt;
    return 0;
  }

  if (!strcmp(cmd, "gp_nlz")) {
    const mxArray *hyp = prhs[1], *X = prhs[2], *y = prhs[3], *s2 = prhs[4];
    const int N = (int)mxGetM(X), D = (int)mxGetN(X), Nhyp = (int)mxGetM(hyp), B = (int)mxGetN(hyp);
    int32_t nf[3] = {1, 0, 0};
    for (int i = 0; i < 3 && i < (int)mxGetNumberOfElements(prhs[6]); ++i) nf[i] = (int32_t)mxGetDoubles(prhs[6])[i];
    plhs[0] = mxCreateDoubleMatrix(1, B, mxREAL);
    mxArray* g = nlhs > 1 ? mxCreateDoubleMatrix(Nhyp, B, mxREAL) : nullptr;
    vbmc_status st = vbmc_gp_nlz(g_ctx, N, D, B, Nhyp, (int)mxGetScalar(prhs[5]), nf, mxGetDoubles(X), mxGetDoubles(y), dbl(s2),
                                 mxGetDoubles(hyp), g ? 1 : 0, mxGetDoubles(plhs[0]), g ? mxGetDoubles(g) : nullptr);
    if (st != VBMC_OK) return fail(st);
    if (g) plhs[1] = g;
    return 0;
  }

  if (!strcmp(cmd, "gp_pred")) {
    vbmc_gp* h = (vbmc_gp*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[1]));
    const mxArray* Xs = prhs[2];
    // vbmc_hip_mex('gp_pred', h, Xstar, ystar, s2star, ssflag, numel(gp.post))
    const int Nstar = (int)mxGetM(Xs), ss = (int)mxGetScalar(prhs[5]), S = (int)mxGetScalar(prhs[6]);
    const int nc = (ss && S > 1) ? S : 1;
    for (int i = 0; i < 4; ++i) plhs[i] = mxCreateDoubleMatrix(Nstar, nc, mxREAL);
    vbmc_status st = vbmc_gp_pred(g_ctx, h, Nstar, mxGetDoubles(Xs), dbl(prhs[3]), dbl(prhs[4]), ss || S == 1, mxGetDoubles(plhs[0]), mxGetDoubles(plhs[1]),
                                  mxGetDoubles(plhs[2]), mxGetDoubles(plhs[3]));
    if (st != VBMC_OK) return fail(st);
    return 0;
  }

  if (!strcmp(cmd, "sq_dist")) {
    const mxArray* a = prhs[1];
    const mxArray* b = (nrhs > 2 && !mxIsEmpty(prhs[2])) ? prhs[2] : nullptr;
    const int D = (int)mxGetM(a), n = (int)mxGetN(a), m = b ? (int)mxGetN(b) : n;
    if (b && (int)mxGetM(b) != D) return raise("vbmc_hip:sq_dist", "Error: column lengths must agree.");
    plhs[0] = mxCreateDoubleMatrix(n, m, mxREAL);
    vbmc_status st = vbmc_sq_dist(g_ctx, D, n, m, mxGetDoubles(a), b ? mxGetDoubles(b) : nullptr, mxGetDoubles(plhs[0]));
    if (st != VBMC_OK) return fail(st);
    return 0;
  }
  return raise("vbmc_hip:usage", "unknown command");
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  g_err_id[0] = 0;
  if (dispatch(nlhs, plhs, nrhs, prhs)) mexErrMsgIdAndTxt(g_err_id, "%s", g_err_msg);   // no C++ object alive here
}
