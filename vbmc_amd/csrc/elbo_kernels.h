// ELBO inner loop on gfx950: negelcbo_vbmc -> gplogjoint + entmc_vbmc / entlb_vbmc.
//
// Reference (acerbilab/vbmc v1.0.12, MATLAB): misc/negelcbo_vbmc.m:1-165,
// misc/gplogjoint.m:1-415, ent/entmc_vbmc.m:1-128, ent/entlb_vbmc.m:1-148,
// misc/vpbndloss.m:1-73, utils/softbndloss.m:1-30.  Nothing here is translated from the
// reference's vectorised bsxfun code: the kernels are organised around wave64 lanes that own
// Monte-Carlo samples (entropy) or GP training points (log-joint), LDS-staged mixture
// parameters, and fixed-order two-level reductions so that results are run-to-run identical.
//
// Kernel inventory (DESIGN.md has the data layout and rooflines):
//   k_prep       theta -> vp fields + packed per-component parameters        (1 WG / restart)
//   k_logjoint   closed-form BQ expected log joint + gradient partials       (1 wave / (k,s,r))
//   k_entropy    antithetic MC entropy + reparameterisation-gradient partials (1 wave / chunk)
//   k_entlb      Gershman lower bound on the entropy + gradient               (1 WG / restart)
//   k_finalize   fixed-order reduction of the partials, Jacobians, penalties  (1 WG / restart)
#pragma once
#include "common.h"
#include "device_math.h"
#include "elbo_types.h"
#include "exp2_tab1k.h"
#include "logjoint_body.h"

// Sum over the workgroup (blockDim.x a multiple of 64; `red` holds at least one double per wave): a fixed-order butterfly
// inside each wave, then the wave totals in wave order -- two barriers instead of a log2(blockDim) LDS tree.
__device__ inline double block_sum(double v, double* red) {
  const int tid = threadIdx.x, nw = blockDim.x >> 6;
  v = wave_sum(v);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int i = 1; i < nw; ++i) r += red[i];
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------------------
// On-device Adam state (utils/fminadam.m:42-102, R chains in lock-step) and the update of one chain by its workgroup.
// ------------------------------------------------------------------------------------------
struct AdamState {
  double *m, *v, *xtab, *ftab;  // T x R, T x R, T x MaxIter x R, MaxIter x R
  int* done;                    // R: 0 = running, otherwise the iteration at which the chain stopped
  int T, R, MaxIter;
  double step_min, step_max, step_decay, TolFun;
  // factors of the update this launch applies (functions of the iteration alone: set by the host per launch, adam_set_iter in
  // abi_elbo.hip -- two pow and an exp per thread on the critical path of a single chain otherwise)
  double c1, c2, step;          // 1 - 0.9^iter, 1 - 0.999^iter (fminadam.m:53-54), step size (:56-57)
};

// one element of the update (utils/fminadam.m:51-59): g = dF_i of this iteration; returns the new x_i
__device__ __forceinline__ double adam_elem(const AdamState& A, int iter, int r, int i, double g, double* __restrict__ x) {
  const int T = A.T;
  const double b1 = 0.9, b2 = 0.999, fudge = 1.4901161193847656e-08;  // sqrt(eps)  (fminadam.m:20-22)
  double m = b1 * A.m[(size_t)r * T + i] + (1.0 - b1) * g;      // :51
  double v = b2 * A.v[(size_t)r * T + i] + (1.0 - b2) * g * g;  // :52
  A.m[(size_t)r * T + i] = m;
  A.v[(size_t)r * T + i] = v;
  const double mhat = m / A.c1, vhat = v / A.c2;
  const double xn = x[(size_t)r * T + i] - A.step * mhat / (sqrt(vhat) + fudge);  // :59 (LB/UB are [] at the call site)
  x[(size_t)r * T + i] = xn;
  A.xtab[((size_t)r * A.MaxIter + (iter - 1)) * T + i] = xn;
  return xn;
}

__device__ inline void adam_update_chain(const AdamState& A, int iter, double* __restrict__ x /*T x R*/,
                                         const double* __restrict__ out /*R x (5+3T)*/, int r) {
  if (A.done[r]) return;
  const int T = A.T;
  const double* o = out + (size_t)r * (OUT_HDR + 3 * T);
  if (threadIdx.x == 0) A.ftab[(size_t)r * A.MaxIter + (iter - 1)] = o[0];
  for (int i = threadIdx.x; i < T; i += blockDim.x) adam_elem(A, iter, r, i, o[OUT_HDR + i], x);
}

// ------------------------------------------------------------------------------------------
// k_prep: unpack theta exactly as misc/negelcbo_vbmc.m:33-48 does.  Inside the on-device optimiser loop the Adam update
// of the previous iteration (adam_iter > 0) is applied first by the same workgroup, which saves one launch per iteration.
// ------------------------------------------------------------------------------------------
// the body of k_prep for restart r, by whichever workgroup calls it (k_prep, or k_finalize_ws for the NEXT iteration of the
// on-device optimiser loop); sh: dynamic LDS of D K + 3 K + D doubles + one per wave
// TH_LDS: th_row is given and lives in LDS (the fused call at the end of k_finalize_ws) -- a compile-time fact, so that the reads of theta
// are ds_read and not flat_load (round 5)
template <bool TH_LDS = false>
__device__ inline void prep_body(const ElboDims& dm, const double* __restrict__ theta, const double* __restrict__ vpfix,
                                 double* __restrict__ vpd, double* __restrict__ entp, int r, double* sh,
                                 const double* th_row = nullptr /* restart r's theta where the caller already holds it (LDS) */) {
  // Latency form (round 3): no workgroup barrier and no LDS.  The two sums every element needs -- sum_k exp(eta_k) of the
  // softmax (:45-47) and sum_d log lambda_d of the normalisation constant -- are computed by EVERY wave for itself (K / 64
  // exponentials per lane and one butterfly: less than a barrier costs), in an order that does not depend on the number of
  // threads, so k_prep (256 threads) and the fused call at the end of k_finalize_ws (1024) give the same bits.  Every output
  // element is then a function of theta alone; the packed entropy parameters recompute the exponentials they need
  // instead of waiting for another thread's result.
  (void)sh;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const int D = dm.D, K = dm.K;
  VpLayout L{D, K};
  const double* th = TH_LDS ? th_row : (th_row ? th_row : theta + (size_t)r * dm.T);
  double* v = vpd + (size_t)r * L.stride();
  const double* fmu = vpfix;
  const double* fsig = vpfix + D * K;
  const double* flam = fsig + K;
  const double* fw = flam + D;
  double s_sum = 0.0, sum_loglam = 0.0;
  if (dm.opt[3]) {
    for (int k = lane; k < K; k += 64) s_sum += exp(th[dm.off_eta + k]);
    s_sum = wave_sum(s_sum);
  }
  for (int d = lane; d < D; d += 64) sum_loglam += dm.opt[2] ? th[dm.off_lambda + d] : log(flam[d]);   // log(exp(x)) = x (:41)
  sum_loglam = wave_sum(sum_loglam);
  auto mu_of = [&](int i) { return dm.opt[0] ? th[dm.off_mu + i] : fmu[i]; };
  auto lnsig_of = [&](int k) { return dm.opt[1] ? th[dm.off_sigma + k] : log(fsig[k]); };
  auto sig_of = [&](int k) { return dm.opt[1] ? exp(th[dm.off_sigma + k]) : fsig[k]; };   // vp.sigma = exp(theta) (:40)
  auto lam_of = [&](int d) { return dm.opt[2] ? exp(th[dm.off_lambda + d]) : flam[d]; };
  auto w_of = [&](int k) { return dm.opt[3] ? exp(th[dm.off_eta + k]) / s_sum : fw[k]; };   // no max-shift (:45-47)
  for (int i = tid; i < D * K; i += nt) v[L.mu() + i] = mu_of(i);
  for (int k = tid; k < K; k += nt) {
    v[L.lnsigma() + k] = lnsig_of(k);
    v[L.sigma() + k] = sig_of(k);
    v[L.eta() + k] = dm.opt[3] ? th[dm.off_eta + k] : log(fw[k]);
    v[L.w() + k] = w_of(k);
  }
  for (int d = tid; d < D; d += nt) {
    v[L.lnlambda() + d] = dm.opt[2] ? th[dm.off_lambda + d] : log(flam[d]);
    v[L.lambda() + d] = lam_of(d);
  }
  if (tid == 0) v[L.lognf()] = -0.5 * D * 1.8378770664093454835606594728112 - sum_loglam;
  // packed entropy parameters [m_dk = mu_dk / lambda_d (D), -1/(2 sigma_k^2), -D ln sigma_k, w_k, w_k / sigma_k^2]
  double* ep = entp + (size_t)r * K * (D + ENTP_EXTRA);
  const int PS = D + ENTP_EXTRA;
  for (int i = tid; i < K * PS; i += nt) {
    const int k = i / PS, c = i - k * PS;
    double val;
    if (c < D) val = mu_of(c + D * k) / lam_of(c);
    else {
      const double sg = sig_of(k);
      if (c == D) val = -0.5 / (sg * sg);
      else if (c == D + 1) val = -(double)D * lnsig_of(k);
      else if (c == D + 2) val = w_of(k);
      else val = w_of(k) / (sg * sg);
    }
    ep[i] = val;
  }
}

__global__ void __launch_bounds__(256) k_prep(ElboDims dm, double* __restrict__ theta,
                                              const double* __restrict__ vpfix,  // mu sigma lambda w (fixed vp) packed
                                              double* __restrict__ vpd, double* __restrict__ entp, AdamState A, int adam_iter,
                                              const double* __restrict__ prev_out) {
  VB_SMALL_PRIO();
  extern __shared__ double sh[];
  if (adam_iter > 0) {
    adam_update_chain(A, adam_iter, theta, prev_out, blockIdx.x);
    __syncthreads();
  }
  prep_body(dm, theta, vpfix, vpd, entp, blockIdx.x, sh);
}

// ------------------------------------------------------------------------------------------
// k_logjoint: misc/gplogjoint.m:162-271 (body: logjoint_body.h).  One workgroup = four (component k, hyper-sample s) cells;
// with LJ_MAXW waves each wave takes every LJ_MAXW-th 16-point slab of the training set.
// partial layout LJ[r][s][k][2D+2] = I_k, w_k*dmu[D], w_k*dsigma (no Jacobian), w_k*dlambda[D]
// ------------------------------------------------------------------------------------------
#ifndef LJ_MAXW
#define LJ_MAXW 4   // waves per workgroup of k_logjoint (eight: 59 vs 52 us per single-chain Adam iteration, round 3)
#endif

template <int DT>
__global__ void __launch_bounds__(WAVE * LJ_MAXW) k_logjoint(ElboDims dm, const double* __restrict__ vpd,
                                                             const double* __restrict__ X,      // N x D col-major
                                                             const double* __restrict__ alpha,  // N x S
                                                             const double* __restrict__ gpc,    // S x GPC_STRIDE
                                                             const double* __restrict__ delta2,  // D (delta.^2)
                                                             double* __restrict__ lj, int want_grad) {
  VB_SMALL_PRIO();
  constexpr int NC = 2 * DT + 2;                 // I, M[DT], S, L[DT]
  __shared__ double TAB[VB_EXP_TAB_N];
  __shared__ double PART[LJ_MAXW][4][NC];        // per wave and component: the 16-lane row sums
  const int s = blockIdx.y, r = blockIdx.z, wv = threadIdx.x >> 6, NW = blockDim.x >> 6;
  for (int t = threadIdx.x; t < VB_EXP_TAB_N; t += blockDim.x) TAB[t] = c_exp2_tab[t];
  VpLayout L{dm.D, dm.K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = gpc + (size_t)s * GPC_STRIDE(dm.D);
  lj_wave_sums<DT>(dm, v, X, alpha + (size_t)s * dm.N, g, delta2, TAB, blockIdx.x, wv, NW, want_grad, &PART[wv][0][0],
                   [] { __syncthreads(); });
  __syncthreads();
  // ---- epilogue by wave 0, one lane per output column of its component; wave partials added in wave order
  if (wv != 0) return;
  lj_write_record<DT>(dm, v, g, delta2, &PART[0][0][0], NW, 4 * NC, blockIdx.x, want_grad, true,
                      lj + ((size_t)r * dm.S + s) * dm.K * (2 * dm.D + 2));
}

// ------------------------------------------------------------------------------------------
// k_logjoint_mfma: the value + gradient form of k_logjoint with the O(N D) gradient sums moved to the fp64 matrix
// cores.  Every gradient of gplogjoint.m:206-252 is a combination of the moments of za_n = z_k(n) alpha_n,
//     M0 = sum_n za_n,   M1_d = sum_n za_n x'_nd,   M2_d = sum_n za_n x'_nd^2        (x' = x - column mean),
// since delta = (mu' - x') / tau:  sum za delta_d = (mu'_d M0 - M1_d)/tau_d,  sum za delta_d^2 = (mu'_d^2 M0 -
// 2 mu'_d M1_d + M2_d)/tau_d^2.  A workgroup = one (restart, hyper-sample); wave w owns the 16 components 16w + li.
// Per k-step (4 training points) a lane evaluates ONE z (its component, point 4q + lg: the MFMA A operand) and the
// 2D + 1 moment columns are NCT = ceil((2D+1)/16) MFMAs against the feature rows [x', x'^2, 1] of the LDS-staged
// chunk of X.  The exponent itself stays on the VALU as a sum of squares of (mu' - x')/tau (no cancellation);
// centring keeps the moment recombination at ~1e-14 relative.
// ------------------------------------------------------------------------------------------
#define LJ_CH 64   // training points staged per chunk (32: 113 against 116 us at the headline shape, twice the barriers for large N; profiles/r06_experiments.md)
// dynamic LDS of the kernel: the staged feature rows, and -- in the same bytes, once the loop is over -- the moment exchange of its nw waves
#define LJ_MFMA_NF(DT_) (16 * ((2 * (DT_) + 1 + 15) / 16))
#define LJ_MFMA_NFP(DT_) ((LJ_MFMA_NF(DT_) % 32 == 16) ? LJ_MFMA_NF(DT_) : LJ_MFMA_NF(DT_) + 16)
#define LJ_MFMA_DYN_LDS(DT_, nw_) (std::max<size_t>((size_t)LJ_CH * LJ_MFMA_NFP(DT_), (size_t)(nw_) * 16 * LJ_MFMA_NF(DT_)) * sizeof(double))
#define LJ_MFMA_STATIC_LDS ((size_t)(VB_EXP_TAB1K_N + LJ_CH) * sizeof(double))
// GRAD = false (round 6: the sieve's value-only passes, R = 250 candidates x S hyper-samples): the same walk without the moments -- the
// lane sums its own z alpha over its points, the four point lanes of a component are added at the end; no MFMA, no feature columns
// beyond x'.  (The VALU kernel's four-cells-per-wave form is built for latency: 575 us for that batch against ~180 here.)
template <int DT, bool GRAD = true>
__global__ void __launch_bounds__(1024) k_logjoint_mfma(ElboDims dm, const double* __restrict__ vpd,
                                                        const double* __restrict__ X,       // N x D col-major
                                                        const double* __restrict__ meanX,   // D column means of X
                                                        const double* __restrict__ alpha,   // N x S
                                                        const double* __restrict__ gpc,     // S x GPC_STRIDE
                                                        const double* __restrict__ delta2,  // D (delta.^2)
                                                        double* __restrict__ lj) {
  VB_SMALL_PRIO();
  constexpr int NCT = (2 * DT + 1 + 15) / 16;
  constexpr int NF = LJ_MFMA_NF(DT);      // feature columns of a staged point: [x'_1..x'_D, x'_1^2..x'_D^2, 1, 0..]  (x' first: 16-byte aligned reads)
  constexpr int CH = LJ_CH;
  __shared__ double TAB[VB_EXP_TAB1K_N];
  // (round 6) the chunk is staged as the FEATURE rows the moment MFMAs multiply by: a lane's B operand is one ds_read_b64 at [point][16 t + li]
  // (it was a three-way select on the column index per MFMA -- a v_cndmask_b32 costs four fp64 operations on this chip), and the exponent
  // reads x' from the same row (columns 0..D-1)
  // (row stride = 16 mod 32 doubles: the two rows a half-wave's ds_read_b64 touches -- points 4 q + lg, lg = 0, 1 or 2, 3 -- then lie in
  // different halves of the banks, for the B operand's sixteen consecutive columns and for the exponent's broadcast reads alike; with
  // NF = 32 the four rows sat on the same banks and the kernel was slower than the selects it replaced: 161 against 138 us)
  constexpr int NFP = LJ_MFMA_NFP(DT);
  extern __shared__ __attribute__((aligned(16))) double MOM[];         // CH x NFP feature rows; after the loop nw x 16 x NF: moments per wave, [cell][column]
  double (*PHI)[NFP] = reinterpret_cast<double (*)[NFP]>(MOM);
  __shared__ double ALC[CH];
  const int s = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int D = dm.D, K = dm.K, N = dm.N;
  const int kk = 16 * wv + li;
  const bool kvalid = kk < K;
  const int k = kvalid ? kk : K - 1;
  for (int t = tid; t < VB_EXP_TAB1K_N; t += nthr) TAB[t] = c_exp2_tab1k[t];
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = gpc + (size_t)s * GPC_STRIDE(D);
  const double sig = v[L.sigma() + k];
  const double wk = v[L.w() + k];
  // The exponent as the table exp wants it, y = (ln nf - sum_d delta_d^2 / 2) 1024/ln2 = lnf' - sum_d (c_d - x'_d t_d)^2 with
  // t_d = sqrt(512/ln2) / tau_d and c_d = mu'_d t_d: two operations per dimension where (mu' - x') / tau, squared and summed, took three, and
  // no scaling of the argument.  (c_d carries one rounding of mu'_d t_d: an absolute error of 1e-16 |mu'_d| / tau_d in delta_d, i.e.
  // 2e-16 |delta_d mu'_d| / tau_d in the exponent -- the reference's own (mu - x) / tau has the rounding of x - mean(x) against it.)
  const double SQH = 27.17829760922398;      // sqrt(1024 / (2 ln 2))
  double tt[DT], cc[DT];
  double sumlogtau = 0.0;
  // (the four point lanes of a component share this set-up: lane group lg takes the dimensions d = lg, lg + 4, .. -- a square root, a
  // logarithm and a division each, ~120 instructions -- and hands t_d, c_d to the other three through the LDS the feature rows will
  // occupy; all ten dimensions on every lane were a third of the kernel's VALU instructions)
  {
    double* ex = MOM + (size_t)(wv * 16 + li) * NF;      // [t_0..t_DT-1 | c_0..c_DT-1] of this component (2 DT <= NF)
#pragma unroll
    for (int u = 0; u < (DT + 3) / 4; ++u) {
      const int d = 4 * u + lg;
      double it = 0.0, m = 0.0;
      if (d < D) {
        const double lam_d = v[L.lambda() + d];
        const double tau = sqrt(sig * sig * lam_d * lam_d + g[d] + delta2[d]);  // :164
        sumlogtau += log(tau);
        it = 1.0 / tau;
        m = v[L.mu() + d + D * k] - meanX[d];
      }
      if (d < DT) { ex[d] = SQH * it; ex[DT + d] = m * (SQH * it); }
    }
    sumlogtau += __shfl_xor(sumlogtau, 16, 64);
    sumlogtau += __shfl_xor(sumlogtau, 32, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int d = 0; d < DT; ++d) { tt[d] = ex[d]; cc[d] = ex[DT + d]; }
  }
  __syncthreads();       // (the feature rows' initialisation follows in the same bytes)
  for (int idx = tid; idx < CH * NFP; idx += nthr) { const int col = idx % NFP; PHI[idx / NFP][col] = col == 2 * D ? 1.0 : 0.0; }   // the constant column and the padding: once
  const double lnnf = g[3 * D] - sumlogtau;  // ln_sf2 + sum_lnell - sum(log(tau_k))  :165
  const double lnf = kvalid ? VB_EXP_TAB1K_SCALE * lnnf : -1.0e300;     // (a padded row: exp -> 0, no select in the loop)
  typedef double lj4 __attribute__((ext_vector_type(4)));
  lj4 acc[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) acc[t] = (lj4){0.0, 0.0, 0.0, 0.0};
  const double* al = alpha + (size_t)s * N;
  for (int c0 = 0; c0 < N; c0 += CH) {
    __syncthreads();
    // (consecutive threads take consecutive dimensions of one point: consecutive LDS words.  Consecutive POINTS of one dimension -- the
    // coalesced order for X -- are 384 B apart in the rows: two banks for a wave's store, 1.4e7 conflict cycles per launch; X is 32 KB in the L2)
    for (int idx = tid; idx < CH * DT; idx += nthr) {
      const int nl = idx / DT, d = idx - nl * DT, n = c0 + nl;
      const double xv = (d < D && n < N) ? X[n + (size_t)N * d] - meanX[d] : 0.0;
      if (d < D) { PHI[nl][d] = xv; if (GRAD) PHI[nl][D + d] = xv * xv; }
    }
    for (int nl = tid; nl < CH; nl += nthr) ALC[nl] = (c0 + nl < N) ? al[c0 + nl] : 0.0;
    __syncthreads();
#pragma unroll 2
    for (int q = 0; q < CH / 4; ++q) {
      const int nl = 4 * q + lg;
      const double* xr = &PHI[nl][0];
      double a2 = 0.0;
#pragma unroll
      for (int d = 0; d < DT; ++d) { const double dl = fma(-xr[d], tt[d], cc[d]); a2 = fma(dl, dl, a2); }   // delta_k :167 (scaled)
      const double z = vb_exp_tab1k(lnf - a2, TAB);                                                         // z_k :168
      const double za = z * ALC[nl];       // alpha is zero beyond N
      if (GRAD) {
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(za, PHI[nl][16 * t + li], acc[t], 0, 0, 0);
      } else {
        acc[0][0] += za;
      }
    }
  }
  if (!GRAD) {      // I_k alone (:169-174): the component's four point lanes, in lane-group order
    double M0 = acc[0][0];
    M0 += __shfl_xor(M0, 16, 64);
    M0 += __shfl_xor(M0, 32, 64);
    if (lg == 0 && kvalid) {
      double nu = 0.0;
      for (int d = 0; d < D; ++d) {
        const double xm = g[D + d], iom2 = g[2 * D + d];
        const double lam_d = v[L.lambda() + d], mu_d = v[L.mu() + d + D * k];
        nu += iom2 * (mu_d * mu_d + sig * sig * lam_d * lam_d - 2.0 * mu_d * xm + xm * xm + delta2[d]);
      }
      lj[(((size_t)r * dm.S + s) * K + k) * (2 * D + 2)] = M0 + g[3 * D + 1] + (-0.5 * nu);
    }
    return;
  }
  __syncthreads();     // (every wave is done with the feature rows: the moments take their place)
  // moments -> LDS: accumulator (row = cell lg + 4 reg, column = 16 t + li)
  double* mw = MOM + (size_t)wv * 16 * (16 * NCT);
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) mw[(lg + 4 * rr) * (16 * NCT) + 16 * t + li] = acc[t][rr];
  __syncthreads();
  if (lg == 0 && kvalid) {
    const double* m = mw + li * (16 * NCT);
    const double M0 = m[2 * D];
    double* o = lj + (((size_t)r * dm.S + s) * K + k) * (2 * D + 2);
    double nu = 0.0, sl2 = 0.0, accS = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      if (d < D) {
        const double xm = g[D + d], iom2 = g[2 * D + d];
        const double lam_d = v[L.lambda() + d], mu_d = v[L.mu() + d + D * k];
        const double M1 = m[d], M2 = m[D + d];
        const double muc = mu_d - meanX[d], it = tt[d] * (1.0 / SQH);      // 1 / tau_d back from t_d (one rounding: the moments carry 1e-14 already)
        const double S1 = (muc * M0 - M1) * it;                                      // sum za delta_d
        const double S2 = ((muc * muc) * M0 - 2.0 * muc * M1 + M2) * (it * it);      // sum za delta_d^2
        const double lit = lam_d * it, sit = sig * it;
        const double accM = -it * S1;                            // dz_dmu*alpha      :207-208
        const double accL = (sit * sit * lam_d) * (S2 - M0);     // dz_dlambda*alpha  :249-250
        accS = fma(lit * lit, S2 - M0, accS);                    // :228
        nu += iom2 * (mu_d * mu_d + sig * sig * lam_d * lam_d - 2.0 * mu_d * xm + xm * xm + delta2[d]);
        sl2 += iom2 * lam_d * lam_d;
        o[1 + d] = wk * accM - wk * iom2 * (mu_d - xm);                    // :208-210
        o[2 + D + d] = wk * accL - wk * sig * sig * iom2 * lam_d;          // :250-252
      }
    }
    o[0] = M0 + g[3 * D + 1] + (-0.5 * nu);                                // :169-174
    o[1 + D] = wk * (accS * sig) - wk * sig * sl2;                         // :229-231
  }
}

// ------------------------------------------------------------------------------------------
// k_entropy: ent/entmc_vbmc.m:49-104.  One wave per (chunk of samples, component j, restart r).
// Lanes 0-31 own base samples with +eps, lanes 32-63 the antithetic -eps (:53-54).
// partial layout PE[r][j][c][NCOL]: sum log q | G[D] | SG | LG[D] | W[K]   (NCOL = 1 if !GRAD)
// ------------------------------------------------------------------------------------------

template <int DT, bool GRAD>
__global__ void __launch_bounds__(WAVE) k_entropy(EntArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int c = blockIdx.x, j = blockIdx.y, r = blockIdx.z;
  const int D = a.D, K = a.K;
  constexpr int PS = DT + ENTP_EXTRA;
  double* P = lds;                 // K * PS  packed params, d padded to DT with zeros
  double* rqb = P + K * PS;        // 64
  double* Tn = rqb + WAVE;         // K * 65 (GRAD)
  {
    const double* gp = a.entp + (size_t)r * K * (D + ENTP_EXTRA);
    for (int idx = lane; idx < K * PS; idx += WAVE) {
      int k = idx / PS, cc = idx - k * PS;
      double val;
      if (cc < DT) val = (cc < D) ? gp[k * (D + ENTP_EXTRA) + cc] : 0.0;
      else val = gp[k * (D + ENTP_EXTRA) + D + (cc - DT)];
      P[idx] = val;
    }
  }
  __syncthreads();
  VpLayout L{D, K};
  const double sigj = a.vpd[(size_t)r * L.stride() + L.sigma() + j];
  double mj[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) mj[d] = P[j * PS + d];
  const double cKj = P[j * PS + DT + 1];

  double accH = 0.0, accSG = 0.0;
  double accG[DT], accLG[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { accG[d] = 0.0; accLG[d] = 0.0; }
  constexpr int KW = 4;  // K <= 256
  double accW[KW] = {0.0, 0.0, 0.0, 0.0};

  const int nt = (a.Mh + 31) / 32;
  const int t0 = (c + a.c0) * a.tiles_per_chunk;
  const int t1 = min(t0 + a.tiles_per_chunk, nt);
  const double sgn = (lane < 32) ? 1.0 : -1.0;
  const double* epsr = a.eps ? a.eps + (size_t)r * a.eps_stride_r + (size_t)j * a.Mh * D : nullptr;

  for (int tile = t0; tile < t1; ++tile) {
    const int b = tile * 32 + (lane & 31);
    const bool valid = b < a.Mh;
    double e[DT], u[DT];
    if (epsr) {
#pragma unroll
      for (int d = 0; d < DT; ++d) e[d] = (valid && d < D) ? sgn * epsr[(size_t)b * D + d] : 0.0;
    } else {
#pragma unroll
      for (int q4 = 0; q4 < (DT + 3) / 4; ++q4) {
        double z4[4];
        vb_normal4(a.seed, (unsigned)b, (unsigned)j, (unsigned)(a.r0 + r * a.rstride), (unsigned)q4, z4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          int d = q4 * 4 + t;
          if (d < DT) e[d] = (valid && d < D) ? sgn * z4[t] : 0.0;
        }
      }
    }
    double e2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      u[d] = fma(e[d], sigj, mj[d]);  // x/lambda = eps*sigma_j + mu_j/lambda   (:55)
      e2 = fma(e[d], e[d], e2);
    }
    const double shift = cKj - 0.5 * e2;  // exponent of the sample's own component
    double qp = 0.0, Ap = 0.0;
    double Bp[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) Bp[d] = 0.0;

#pragma unroll 2
    for (int k = 0; k < K; ++k) {
      const double* Pk = P + k * PS;
      double acc = 0.0;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        double df = u[d] - Pk[d];
        acc = fma(df, df, acc);
      }
      double ex = fma(acc, Pk[DT], Pk[DT + 1] - shift);  // -d2/2 - D log sigma_k - shift  (:61-63)
      double n = vb_exp(ex);
      qp = fma(Pk[DT + 2], n, qp);                       // q' += w_k n_k   (:64)
      if (GRAD) {
        double ta = n * Pk[DT + 3];                      // w_k n_k / sigma_k^2
        Ap += ta;
#pragma unroll
        for (int d = 0; d < DT; ++d) Bp[d] = fma(ta, Pk[d], Bp[d]);
        Tn[k * 65 + lane] = n;
      }
    }
    // log q = log nf + shift + log q'  (log nf added in k_finalize)
    const double lq = valid ? (shift + log(qp)) : 0.0;
    accH += lq;
    if (GRAD) {
      const double rq = valid ? 1.0 / qp : 0.0;
      double sg = 0.0;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        // lambda_d * lsum_d / q = u_d * A - B_d   (:77-79)
        double gd = (u[d] * Ap - Bp[d]) * rq;
        accG[d] += gd;                         // -> mu_grad (:82)
        double eg = e[d] * gd;
        sg += eg;                              // -> sigma_grad (:87-88)
        accLG[d] += eg;                        // -> lambda_grad (:93)
      }
      accSG += sg;
      rqb[lane] = rq;
      __syncthreads();
      // w_grad(l) -= w_j * sum_i N_l(x_i)/q(x_i)  (:100): lane l sweeps the 64 samples of the tile
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
        int l = lane + kk * WAVE;
        if (l < K) {
          double ws = 0.0;
          const double* Tl = Tn + l * 65;
#pragma unroll 8
          for (int i = 0; i < WAVE; ++i) ws = fma(Tl[i], rqb[i], ws);
          accW[kk] += ws;
        }
      }
      __syncthreads();
    }
  }
  // wave reduction, fixed order
  double* o = a.part + (((size_t)r * K + j) * a.C + c) * a.ncol;
  accH = wave_sum(accH);
  if (GRAD) {
    accSG = wave_sum(accSG);
#pragma unroll
    for (int d = 0; d < DT; ++d) { accG[d] = wave_sum(accG[d]); accLG[d] = wave_sum(accLG[d]); }
  }
  if (lane == 0) {
    o[0] = accH;
    if (GRAD) {
      o[1 + D] = accSG;
#pragma unroll
      for (int d = 0; d < DT; ++d)
        if (d < D) { o[1 + d] = accG[d]; o[2 + D + d] = accLG[d]; }
    }
  }
  if (GRAD) {
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      int l = lane + kk * WAVE;
      if (l < K) o[2 + 2 * D + l] = accW[kk];
    }
  }
}

// eps dump for the test hook (same generator, same counters as k_entropy)
__global__ void k_rng_dump(int D, int K, int R, int Mh, unsigned long long seed, double* __restrict__ out) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)R * K * Mh;
  if (idx >= total) return;
  int b = (int)(idx % Mh);
  int j = (int)((idx / Mh) % K);
  int r = (int)(idx / ((long long)Mh * K));
  double* o = out + idx * D;
  for (int q4 = 0; q4 < (D + 3) / 4; ++q4) {
    double z4[4];
    vb_normal4(seed, (unsigned)b, (unsigned)j, (unsigned)r, (unsigned)q4, z4);
    for (int t = 0; t < 4; ++t)
      if (q4 * 4 + t < D) o[q4 * 4 + t] = z4[t];
  }
}

// ------------------------------------------------------------------------------------------
// k_entlb: ent/entlb_vbmc.m:1-148 -- deterministic lower bound, O(K^2 D); one WG per restart.
// Writes H and the *untransformed-to-theta-order* gradient pieces straight into the same
// "entropy result" slots k_finalize reads: EB[r] = H | mu_grad[D*K] | sigma_grad[K] (Jacobian
// applied) | lambda_grad[D] | w_grad_raw[K] (softmax Jacobian applied in k_finalize).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_entlb(ElboDims dm, const double* __restrict__ vpd,
                                               double* __restrict__ eb, int want_grad, double* __restrict__ gamma_g) {
  VB_SMALL_PRIO();
  extern __shared__ double lds[];
  const int r = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int D = dm.D, K = dm.K;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* mu = v + L.mu();
  const double* sigma = v + L.sigma();
  const double* lam = v + L.lambda();
  const double* w = v + L.w();
  double* o = eb + (size_t)r * (1 + D * K + 2 * K + D);
  // K*K  gamma[j + K*k]: in LDS up to K = 128; larger mixtures keep it in a global scratch block of the restart (the
  // workgroup barriers below order its writes before the reads)
  double* gamma = gamma_g ? gamma_g + (size_t)r * K * K : lds;
  double* gammasum = gamma_g ? lds : lds + K * K; // K
  double* red = gammasum + K;       // nt
  if (K == 1) {  // :32-47 exact entropy
    if (tid == 0) {
      double sl = 0.0;
      for (int d = 0; d < D; ++d) sl += log(lam[d]);
      o[0] = 0.5 * D * (1.0 + 1.8378770664093454835606594728112) + D * log(sigma[0]) + sl;
      if (want_grad) {
        for (int d = 0; d < D; ++d) o[1 + d] = 0.0;
        o[1 + D] = D / sigma[0] * sigma[0];   // sigma_grad = D./sigma, then Jacobian *sigma (:36,131)
        for (int d = 0; d < D; ++d) o[1 + D + 1 + d] = 1.0;  // :41
        o[1 + D + 1 + D] = 0.0;
      }
    }
    return;
  }
  const double lognf = v[L.lognf()];
  for (int p = tid; p < K * K; p += nt) {
    int j = p % K, k = p / K;
    double ss2 = sigma[j] * sigma[j] + sigma[k] * sigma[k];
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) {
      double t = (mu[d + D * j] - mu[d + D * k]) / lam[d];
      d2 = fma(t, t, d2);
    }
    d2 /= ss2;
    gamma[p] = exp(lognf - 0.5 * D * log(ss2) - 0.5 * d2);  // :75
  }
  __syncthreads();
  for (int k = tid; k < K; k += nt) {
    double gs = 0.0;
    for (int j = 0; j < K; ++j) gs += w[j] * gamma[j + K * k];  // :76
    gammasum[k] = gs;
  }
  __syncthreads();
  if (tid == 0) {
    double H = 0.0;
    for (int k = 0; k < K; ++k) H -= w[k] * log(gammasum[k]);  // :78
    o[0] = H;
  }
  if (!want_grad) return;
  // mu_grad(:,j) = -w_j * sum_k dmu_jk * (w_k gamma_jk/gammasum_k + gamma_jk w_k / gammasum_j)  (:96-98)
  for (int p = tid; p < D * K; p += nt) {
    int d = p % D, j = p / D;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      double ss2 = sigma[j] * sigma[j] + sigma[k] * sigma[k];
      double dmu = (mu[d + D * k] - mu[d + D * j]) / (ss2 * lam[d] * lam[d]);  // :87
      double gj = gamma[j + K * k] * w[k];
      acc += dmu * (gj / gammasum[k] + gj / gammasum[j]);
    }
    o[1 + p] = -w[j] * acc;
  }
  for (int j = tid; j < K; j += nt) {  // :103-105, Jacobian :131
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      double ss2 = sigma[j] * sigma[j] + sigma[k] * sigma[k];
      double m2 = 0.0;
      for (int d = 0; d < D; ++d) {
        double t = (mu[d + D * j] - mu[d + D * k]) / lam[d];
        m2 = fma(t, t, m2);
      }
      double ds = -D / ss2 + m2 / (ss2 * ss2);  // :90
      double gj = gamma[j + K * k] * w[k];
      acc += ds * (gj / gammasum[k] + gj / gammasum[j]);
    }
    o[1 + D * K + j] = -w[j] * sigma[j] * acc * sigma[j];
    // w_grad = -log(gammasum) - sum_k w_k gamma_jk / gammasum_k  (:118)
    double ws = 0.0;
    for (int k = 0; k < K; ++k) ws += w[k] * gamma[j + K * k] / gammasum[k];
    o[1 + D * K + K + D + j] = -log(gammasum[j]) - ws;
  }
  for (int d = tid; d < D; d += nt) {  // :110-113
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      double inner = 0.0;
      for (int j = 0; j < K; ++j) {
        double ss2 = sigma[j] * sigma[j] + sigma[k] * sigma[k];
        double t = mu[d + D * k] - mu[d + D * j];
        double dmu2 = t * t / (ss2 * lam[d] * lam[d]);
        inner += (dmu2 - 1.0) * gamma[j + K * k] * w[j];
      }
      acc += w[k] * inner / gammasum[k];
    }
    o[1 + D * K + K + d] = -acc;
  }
  (void)red;
}

// ------------------------------------------------------------------------------------------
// k_ent_reduce: sum the per-chunk entropy partials over chunks, in chunk order, one thread per
// column: red[r][j][col] = sum_c part[r][j][c][col].  Keeps k_finalize's latency independent of the
// chunk count (which is large when few restarts must still fill the chip).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void ent_reduce_body(int j, int r, int K, int C, int ncol, const double* __restrict__ part,
                                                double* __restrict__ red, int walk_tpw, int walk_ntile) {
  const double* p = part + ((size_t)r * K + j) * C * ncol;
  if (walk_tpw > 0) C = ent_walk_slots((long long)r * K + j, walk_ntile, walk_tpw);   // the walk: the slots this pair's waves filled, of its C
  double* o = red + ((size_t)r * K + j) * ncol;
  for (int col = threadIdx.x; col < ncol; col += blockDim.x) {
    double acc = 0.0;
    int c = 0;
    // (round 5) 32 loads in flight, then 8: every batch is one round trip to memory, and a single chain waits for each of them
    // (k_reduce_both 4.9 -> 3.x us at VBMC's own sample count); the sum stays in chunk order
    for (; c + 32 <= C; c += 32) {
      double t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = p[(size_t)(c + u) * ncol + col];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += t[u];
    }
    for (; c + 8 <= C; c += 8) {
      double v0 = p[(size_t)(c + 0) * ncol + col], v1 = p[(size_t)(c + 1) * ncol + col];
      double v2 = p[(size_t)(c + 2) * ncol + col], v3 = p[(size_t)(c + 3) * ncol + col];
      double v4 = p[(size_t)(c + 4) * ncol + col], v5 = p[(size_t)(c + 5) * ncol + col];
      double v6 = p[(size_t)(c + 6) * ncol + col], v7 = p[(size_t)(c + 7) * ncol + col];
      acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
    }
    for (; c < C; ++c) acc += p[(size_t)c * ncol + col];
    o[col] = acc;
  }
}

__global__ void __launch_bounds__(256) k_ent_reduce(int C, int ncol, const double* __restrict__ part,
                                                    double* __restrict__ red, int walk_tpw, int walk_ntile) {
  VB_SMALL_PRIO();
  ent_reduce_body(blockIdx.x, blockIdx.y, gridDim.x, C, ncol, part, red, walk_tpw, walk_ntile);
}

// k_lj_reduce: sum the log-joint partials over hyper-samples in sample order, one thread per column:
// ljbar[r][k][col] = sum_s lj[r][s][k][col]   (gplogjoint.m:399-413 averages are linear in these sums)
__device__ __forceinline__ void lj_reduce_body(int k, int r, int S, int K, int LJS, const double* __restrict__ lj,
                                               double* __restrict__ ljbar) {
  for (int col = threadIdx.x; col < LJS; col += blockDim.x) {
    double acc = 0.0;
    const double* p = lj + ((size_t)r * S * K + k) * LJS + col;
    const size_t st = (size_t)K * LJS;
    int s = 0;
    for (; s + 32 <= S; s += 32) {   // (round 5) 32 loads in flight: one round trip where there were four; summed in sample order
      double t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = p[(size_t)(s + u) * st];
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += t[u];
    }
    for (; s + 8 <= S; s += 8) {   // eight loads in flight; summed in sample order
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(s + u) * st];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; s < S; ++s) acc += p[(size_t)s * st];
    ljbar[((size_t)r * K + k) * LJS + col] = acc;
  }
}

__global__ void __launch_bounds__(64) k_lj_reduce(int S, int K, int LJS, const double* __restrict__ lj,
                                                  double* __restrict__ ljbar) {
  VB_SMALL_PRIO();
  lj_reduce_body(blockIdx.x, blockIdx.y, S, K, LJS, lj, ljbar);
}

// both reductions in one launch (blockIdx.z = 0: entropy chunks, 1: hyper-samples) when they sit on the same stream
__global__ void __launch_bounds__(256) k_reduce_both(int C, int ncol, const double* __restrict__ part, double* __restrict__ red,
                                                     int S, int LJS, const double* __restrict__ lj, double* __restrict__ ljbar,
                                                     int walk_tpw, int walk_ntile) {
  VB_SMALL_PRIO();
  if (blockIdx.z == 0) ent_reduce_body(blockIdx.x, blockIdx.y, gridDim.x, C, ncol, part, red, walk_tpw, walk_ntile);
  else lj_reduce_body(blockIdx.x, blockIdx.y, S, gridDim.x, LJS, lj, ljbar);
}

// ------------------------------------------------------------------------------------------
// k_finalize: reduce partials in a fixed order, apply Jacobians (gplogjoint.m:352-373,
// entmc_vbmc.m:106-125), average over hyper-samples (gplogjoint.m:399-413), add the soft-bound
// and weight penalties (negelcbo_vbmc.m:116-164), pack to theta order.
// ------------------------------------------------------------------------------------------
struct FinArgs {
  ElboDims dm;
  const double* vpd;
  const double* theta;
  const double* ljbar;    // R x K x (2D+2): log-joint partials summed over hyper-samples
  const double* entpart;  // MC partials or null
  const double* entlb;    // entlb block or null
  const double* var;      // R x 2 (varG, varGss) + optional dvarG[T] per restart, or null
  const double* bnd;      // lb[Text] ub[Text] or null
  double TolCon, WeightThreshold, WeightPenalty, beta;
  int M, C, ncol, want_grad, has_bnd, var_stride;
  int no_jacobian;        // 1: gradients with respect to sigma, lambda, w themselves (JACOBIAN_FLAG = 0 of the stand-alone forms; k_finalize_ws only)
  double invS, invM;      // 1 / S and 1 / (2 M) from the host (the same IEEE quotients; a division per thread in k_finalize_ws's preamble otherwise)
  int stage;              // 1: the host sized the LDS so that the log-joint and entropy records of a restart are staged in it
  double* big;            // null, or R x (3T + DK) doubles of global scratch for dG | dH | dP | gsc when they exceed the LDS
  double* out;            // R x (OUT_HDR + 3T)
  // on-device optimiser loop (k_finalize_ws): after this iteration's results, the same workgroup applies the Adam update
  // (next_iter > 0: utils/fminadam.m:48-61 for iteration next_iter) and unpacks the NEW theta for the next iteration (k_prep's
  // body) -- one launch per iteration less, and the next iteration's first kernel starts from records that are already there
  int next_iter;
  AdamState next_A;
  double* next_theta;
  const double* next_vpfix;
  double* next_vpd;
  double* next_entp;
};

#define FIN_THREADS 1024
// global -> LDS copy with eight independent loads in flight per thread (the plain loop waits for each load in turn)
__device__ __forceinline__ void stage_copy(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nt) {
  int i = tid;
  for (; i + 7 * nt < n; i += 8 * nt) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[i + u * nt];
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * nt] = t[u];
  }
  for (; i < n; i += nt) dst[i] = src[i];
}

// up to four blocks, every load of all of them issued before the first store (round 5: four calls of stage_copy were four -- with the
// single-element tail loop up to a dozen -- round trips to memory one after the other: 3.3 us of k_finalize_ws's 15, tools/fin_timeline.py).
// Blocks of at most 4 nt elements each; longer ones fall back to stage_copy.
struct StageBlk { double* dst; const double* src; int n; };
__device__ __forceinline__ void stage_copy4(const StageBlk (&b)[4], int tid, int nt) {
  double t[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < 4; ++u) t[q][u] = (b[q].n <= 4 * nt && tid + u * nt < b[q].n) ? b[q].src[tid + u * nt] : 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (b[q].n > 4 * nt) { stage_copy(b[q].dst, b[q].src, b[q].n, tid, nt); continue; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (tid + u * nt < b[q].n) b[q].dst[tid + u * nt] = t[q][u];
  }
}


// ------------------------------------------------------------------------------------------
// k_finalize_ws: the same arithmetic as k_finalize, organised for LATENCY (round 3).  k_finalize runs its sections one after the
// other with every thread of the workgroup and a workgroup barrier (or a two-barrier block sum) between them: ~18 barriers of
// 1024 threads, 16 us for one restart -- the second-longest kernel of a single Adam chain.  Here every section is ONE WAVE's
// task: the inputs are staged in LDS once, each wave computes the outputs of its task from them with wave-level sums only
// (the few scalars a task needs from another section -- G, the entropy means, the weight-gradient dot product -- it
// recomputes itself: O(K) or O(K^2 / 64) work), and the waves meet at ONE barrier before the assembly.  Sums over components
// run over the 64 lanes of a wave in a fixed order (butterfly), so results stay run-to-run identical; they differ from
// k_finalize's in the last bits only where the order of a sum differs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef VBMC_INSTRUMENT   // phase timeline of the finalize kernel (tools/fin_timeline.py): restart 0's workgroup stamps the 100 MHz counter
__device__ unsigned long long g_fin_dbg[64];   // (in-loop iterations only: the Adam tail is part of the picture)
#define FIN_STAMP(i_) do { if (a.next_iter > 0 && blockIdx.x == 0 && threadIdx.x == 0) g_fin_dbg[i_] = wall_clock64(); } while (0)
#define FIN_STAMP_W(i_) do { if (a.next_iter > 0 && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_fin_dbg[i_] = wall_clock64(); } while (0)
#else
#define FIN_STAMP(i_) do { } while (0)
#define FIN_STAMP_W(i_) do { } while (0)
#endif
// FAST (round 5): everything staged (a.stage == 3) and nothing in the global fall-back block (a.big == null) -- the case of every shape but
// the very largest.  Known at compile time, the records, the vp block, the bounds and the gradient vectors are LDS pointers and nothing
// else: ds_read / ds_write instead of the flat_load / flat_store (130 + 78 of them, each waiting on BOTH memory counters) the compiler
// must emit for a pointer that is global memory or LDS depending on a run-time flag.
template <bool FAST>
__global__ void __launch_bounds__(FIN_THREADS) k_finalize_ws(FinArgs a) {
  VB_SMALL_PRIO();
  FIN_STAMP(0);
  extern __shared__ double lds[];
  const int r = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
  const ElboDims& dm = a.dm;
  const int D = dm.D, K = dm.K, S = dm.S, T = dm.T;
  VpLayout L{D, K};
  const double* v = a.vpd + (size_t)r * L.stride();
  const double* bnd = a.bnd;
  double* red = lds;           // nt   (unused here; same layout as k_finalize so that the host sizes one LDS block)
  double* Ibar = red + nt;     // K    (unused)
  double* Hj = Ibar + K;       // K    (unused)
  double* wraw = Hj + K;       // K    raw w-gradient of H (task 5's own scratch)
  double* bigr = (!FAST && a.big) ? a.big + (size_t)r * (3 * (size_t)T + (size_t)D * K) : nullptr;
  double* dG = bigr ? bigr : wraw + K;       // T (packed)
  double* dH = dG + T;         // T
  double* dP = dH + T;         // T   penalty gradient
  double* scal = bigr ? wraw + K : dP + T;   // 8 scalars: G, H, three partial penalties
  double* gsc = bigr ? dP + T : scal + 8;    // D x K   soft-bound gradient of the lnscale block per (d, k)  (task 7's own scratch)
  double* stg = bigr ? scal + 8 : gsc + D * K;
  StageBlk sb[4] = {{nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
  if (FAST || (a.stage & 2)) {
    sb[0] = StageBlk{stg, v, L.stride()};
    v = stg;
    stg += L.stride();
    if (FAST) bnd = stg;          // (an LDS pointer whether or not there are bounds: never read without)
    if (a.has_bnd) {
      const int next_mu = dm.opt[0] ? D * K : 0;
      const int Text = next_mu + ((dm.opt[1] || dm.opt[2]) ? D * K : 0) + (dm.opt[3] ? K : 0);
      sb[1] = StageBlk{stg, a.bnd, 3 * Text};
      bnd = stg;
      stg += 3 * Text;
    }
  }
  const double* w = v + L.w();
  const double* sigma = v + L.sigma();
  const double* lam = v + L.lambda();
  double* o = a.out + (size_t)r * (OUT_HDR + 3 * T);
  const int LJS = 2 * D + 2;
  const double invS = a.invS;
  const double* lb = a.ljbar + (size_t)r * K * LJS;
  const double* pe = a.entpart ? a.entpart + (size_t)r * K * a.C * a.ncol : nullptr;   // (C = 1: reduced records)
  if (FAST || (a.stage & 1)) {
    double* lbL = stg;
    double* peL = lbL + K * LJS;
    sb[2] = StageBlk{lbL, lb, K * LJS};
    if (pe) sb[3] = StageBlk{peL, pe, K * a.C * a.ncol};
    lb = lbL;
    if (pe) pe = peL;
  }
  stage_copy4(sb, tid, nt);      // (every load of the four blocks in flight before the first store)
  for (int i = tid; i < T; i += nt) { dG[i] = 0.0; dH[i] = 0.0; dP[i] = 0.0; }
  if (tid < 8) scal[tid] = 0.0;
  __syncthreads();
  FIN_STAMP(1);

  const bool grad = a.want_grad != 0;
  const bool jac = a.no_jacobian == 0;   // the Jacobians of sigma = exp(.), lambda = exp(.), w = softmax(eta) (gplogjoint.m:352-373, entmc_vbmc.m:110-125)
  const double lognf = v[L.lognf()];
  const double invM = a.entpart ? a.invM : 0.0;
  const int ncol = a.ncol;
  const double* eb = a.entpart ? nullptr : a.entlb + (size_t)r * (1 + D * K + 2 * K + D);
  const int next_mu = dm.opt[0] ? D * K : 0;
  const int has_sc = (dm.opt[1] || dm.opt[2]) ? 1 : 0;
  const int Text = next_mu + has_sc * D * K + (dm.opt[3] ? K : 0);
  const double* blo = bnd;
  const double* bup = (FAST || bnd) ? bnd + Text : nullptr;
  const double* binv = (FAST || bnd) ? bnd + 2 * Text : nullptr;     // 1 / ell^2, ell = (ub - lb) TolCon (softbndloss.m), from the host

  // Eleven tasks (round 5; nine through round 4): the two lambda blocks that were D sequential 64-lane butterflies (tasks 0 and 4: 3.8 and
  // 3.6 us of a 5.4 us phase, tools/fin_timeline.py) sum their K terms per lane (lane = dimension + 32 x half of the components) and meet in
  // ONE exchange; the lambda sums of the lnscale bounds get a wave of their own and recompute their terms instead of reading the other
  // wave's table; the K x K weight gradient issues its LDS reads eight at a time.
  for (int task = wave; task < 11; task += nw) {
    FIN_STAMP_W(16 + 2 * task);
    switch (task) {
      case 0: {   // G (:203,:400), the sigma / lambda / eta blocks of its gradient (:356,:362,:366-368)
        double part = 0.0;
        for (int k = lane; k < K; k += 64) part += w[k] * (lb[(size_t)k * LJS] * invS);
        const double G = wave_sum(part);
        if (lane == 0) scal[0] = G;
        if (grad) {
          if (dm.opt[1])
            for (int k = lane; k < K; k += 64) dG[dm.off_sigma + k] = lb[(size_t)k * LJS + 1 + D] * (jac ? sigma[k] : 1.0) * invS;
          if (dm.opt[3])
            for (int k = lane; k < K; k += 64) { const double Ib = lb[(size_t)k * LJS] * invS; dG[dm.off_eta + k] = jac ? w[k] * Ib - w[k] * G : Ib; }
        }
      } break;
      case 9:     // lambda block of dG (:250,:362): lane <-> (dimension d = lane & 31, components of parity lane >> 5)
        if (grad && dm.opt[2])
          for (int d0 = 0; d0 < D; d0 += 32) {
            const int d = d0 + (lane & 31);
            double acc = 0.0;
            if (d < D) {
#pragma unroll 8
              for (int k = lane >> 5; k < K; k += 2) acc += lb[(size_t)k * LJS + 2 + D + d];
            }
            acc += __shfl_xor(acc, 32, 64);
            if (lane < 32 && d < D) dG[dm.off_lambda + d] = acc * (jac ? lam[d] : 1.0) * invS;
          }
        break;
      case 1:     // mu block of dG
        if (grad && dm.opt[0])
          for (int p = lane; p < D * K; p += 64) dG[dm.off_mu + p] = lb[(size_t)(p / D) * LJS + 1 + p % D] * invS;
        break;
      case 2: {   // H (:67) and the sigma block of its gradient (:88,:113)
        if (pe) {
          double part = 0.0;
          for (int j = lane; j < K; j += 64) part -= w[j] * (lognf + pe[(size_t)j * ncol] * invM);
          const double H = wave_sum(part);
          if (lane == 0) scal[1] = H;
          if (grad && dm.opt[1])
            for (int j = lane; j < K; j += 64) dH[dm.off_sigma + j] = w[j] * pe[(size_t)j * ncol + 1 + D] * invM * (jac ? sigma[j] : 1.0);
        } else {
          if (lane == 0) scal[1] = eb[0];
          // k_entlb's sigma and lambda blocks carry their Jacobians (entlb_vbmc.m:132-137): divided back out for JACOBIAN_FLAG = 0
          if (grad && dm.opt[1]) for (int k = lane; k < K; k += 64) dH[dm.off_sigma + k] = jac ? eb[1 + D * K + k] : eb[1 + D * K + k] / sigma[k];
        }
      } break;
      case 3:     // mu block of dH (:82)
        if (grad && dm.opt[0]) {
          if (pe)
            for (int p = lane; p < D * K; p += 64) {
              const int d = p % D, j = p / D;
              dH[dm.off_mu + p] = w[j] * pe[(size_t)j * ncol + 1 + d] * invM / lam[d];
            }
          else
            for (int p = lane; p < D * K; p += 64) dH[dm.off_mu + p] = eb[1 + p];
        }
        break;
      case 4:     // lambda block of dH (:93; the /lambda of lsum cancels the *lambda of :107)
        if (grad && dm.opt[2]) {
          if (pe)
            for (int d0 = 0; d0 < D; d0 += 32) {
              const int d = d0 + (lane & 31);
              double acc = 0.0;
              if (d < D) {
#pragma unroll 8
                for (int j = lane >> 5; j < K; j += 2) acc += w[j] * sigma[j] * pe[(size_t)j * ncol + 2 + D + d] * invM;
              }
              acc += __shfl_xor(acc, 32, 64);
              if (lane < 32 && d < D) dH[dm.off_lambda + d] = jac ? acc : acc / lam[d];      // (:116-118)
            }
          else
            for (int d = lane; d < D; d += 64) dH[dm.off_lambda + d] = jac ? eb[1 + D * K + K + d] : eb[1 + D * K + K + d] / lam[d];
        }
        break;
      case 5:     // eta block of dH: raw weight gradient (:97-100), then the softmax Jacobian (:121-123)
        if (grad && dm.opt[3]) {
          double dpart = 0.0;
          for (int l0 = 0; l0 < K; l0 += 64) {
            const int l = l0 + lane;
            double wr = 0.0;
            if (l < K) {
              if (pe) {
                double acc = 0.0;
                int j = 0;
                for (; j + 8 <= K; j += 8) {     // lane <-> l: no cross-lane sum; eight LDS reads in flight, summed in component order
                  double t8[8];
#pragma unroll
                  for (int u = 0; u < 8; ++u) t8[u] = w[j + u] * pe[(size_t)(j + u) * ncol + 2 + 2 * D + l] * invM;
#pragma unroll
                  for (int u = 0; u < 8; ++u) acc += t8[u];
                }
                for (; j < K; ++j) acc += w[j] * pe[(size_t)j * ncol + 2 + 2 * D + l] * invM;
                wr = -(lognf + pe[(size_t)l * ncol] * invM) - acc;
              } else {
                wr = eb[1 + D * K + K + D + l];
              }
              wraw[l] = wr;
              dpart += w[l] * wr;
            }
          }
          const double dot = wave_sum(dpart);
          wave_fence();
          for (int l = lane; l < K; l += 64) dH[dm.off_eta + l] = jac ? w[l] * wraw[l] - w[l] * dot : wraw[l];
        }
        break;
      case 6:     // soft bounds, mu block (vpbndloss.m, softbndloss.m)
        if (a.has_bnd && dm.opt[0]) {
          double part = 0.0;
          for (int p = lane; p < D * K; p += 64) {
            const double x = v[L.mu() + p], l = blo[p], u = bup[p], i2 = binv[p];
            double g = 0.0;
            if (x < l) { const double t = x - l; part += 0.5 * t * t * i2; g += t * i2; }
            if (x > u) { const double t = x - u; part += 0.5 * t * t * i2; g += t * i2; }
            dP[dm.off_mu + p] = g;
          }
          part = wave_sum(part);
          if (lane == 0) scal[2] = part;
        }
        break;
      case 7:     // soft bounds, lnscale block D x K: ln sigma_k + ln lambda_d (vpbndloss.m:36); gradient summed over d / k
        if (a.has_bnd && has_sc) {
          double part = 0.0;
          for (int p = lane; p < D * K; p += 64) {
            const int d = p % D, k = p / D;
            const double x = v[L.lnsigma() + k] + v[L.lnlambda() + d];
            const double l = blo[next_mu + p], u = bup[next_mu + p], i2 = binv[next_mu + p];
            double g = 0.0;
            if (x < l) { const double t = x - l; part += 0.5 * t * t * i2; g += t * i2; }
            if (x > u) { const double t = x - u; part += 0.5 * t * t * i2; g += t * i2; }
            gsc[p] = g;
          }
          part = wave_sum(part);
          if (lane == 0) scal[3] = part;
          wave_fence();
          if (grad) {
            if (dm.opt[1])
              for (int k = lane; k < K; k += 64) {
                double acc = 0.0;
                for (int d = 0; d < D; ++d) acc += gsc[d + D * k];
                dP[dm.off_sigma + k] = acc;
              }
          }
        }
        break;
      case 10:    // ... its lambda sums: a wave of its own that recomputes the (rarely non-zero) terms, lane <-> (d, parity of k)
        if (a.has_bnd && has_sc && grad && dm.opt[2])
          for (int d0 = 0; d0 < D; d0 += 32) {
            const int d = d0 + (lane & 31);
            double acc = 0.0;
            if (d < D) {
              const double lnl = v[L.lnlambda() + d];
              for (int k = lane >> 5; k < K; k += 2) {
                const double x = v[L.lnsigma() + k] + lnl;
                const double l = blo[next_mu + d + D * k], u = bup[next_mu + d + D * k], i2 = binv[next_mu + d + D * k];
                double g = 0.0;
                if (x < l) g += (x - l) * i2;
                if (x > u) g += (x - u) * i2;
                acc += g;
              }
            }
            acc += __shfl_xor(acc, 32, 64);
            if (lane < 32 && d < D) dP[dm.off_lambda + d] = acc;
          }
        break;
      case 8:     // soft bounds on eta and the weight-size penalty (negelcbo_vbmc.m:146-162)
        if (a.has_bnd && dm.opt[3]) {
          const int o3 = next_mu + has_sc * D * K;
          double part = 0.0, pd = 0.0;
          for (int k = lane; k < K; k += 64) {
            const double x = v[L.eta() + k], l = blo[o3 + k], u = bup[o3 + k], i2 = binv[o3 + k];
            if (x < l) { const double t = x - l; part += 0.5 * t * t * i2; }
            if (x > u) { const double t = x - u; part += 0.5 * t * t * i2; }
            part += a.WeightPenalty * ((w[k] < a.WeightThreshold) ? w[k] : a.WeightThreshold);
            pd += (w[k] < a.WeightThreshold) ? w[k] * a.WeightPenalty : 0.0;
          }
          part = wave_sum(part);
          const double dot = wave_sum(pd);
          if (lane == 0) scal[4] = part;
          if (grad)
            for (int k = lane; k < K; k += 64) {
              const double x = v[L.eta() + k], l = blo[o3 + k], u = bup[o3 + k], i2 = binv[o3 + k];
              double g = 0.0;
              if (x < l) g += (x - l) * i2;
              if (x > u) g += (x - u) * i2;
              const double gk = (w[k] < a.WeightThreshold) ? a.WeightPenalty : 0.0;
              dP[dm.off_eta + k] = g + (w[k] * gk - w[k] * dot);
            }
        }
        break;
      default: break;
    }
    FIN_STAMP_W(17 + 2 * task);
  }
  __syncthreads();
  FIN_STAMP(2);
  // ---- assemble
  double varG = 0.0, varGss = 0.0;
  const double* vr = a.var ? a.var + (size_t)r * a.var_stride : nullptr;
  if (vr) { varG = vr[0]; varGss = vr[1]; }
  if (tid == 0) {
    const double G = scal[0], H = scal[1];
    double F = -G - H;                                  // :116
    if (a.beta != 0.0) F += a.beta * sqrt(varG);        // :127 (varH = 0)
    F += (scal[2] + scal[3]) + scal[4];
    o[0] = F; o[1] = G; o[2] = H; o[3] = varG; o[4] = varGss;
  }
  // ---- on-device optimiser loop: this iteration's Adam update and the unpacking of the new theta, by the same workgroup.  Every
  // thread updates the elements whose gradient it has just assembled (no barrier, no trip through global memory in between); the
  // new theta reaches the unpacking through LDS (xl: the gradient blocks are dead once every thread has assembled its elements)
  const bool adam = a.next_iter > 0 && grad;
  const bool running = adam && a.next_A.done[r] == 0;
  double xn_keep[4];                                     // T <= 4 * 1024 elements per restart pass through here (else: global)
  const bool keep = adam && T <= 4 * nt;
  if (running && tid == 0) a.next_A.ftab[(size_t)r * a.next_A.MaxIter + (a.next_iter - 1)] = o[0];
  if (grad) {
    auto elem = [&](int i) -> double {
      double g = -dG[i] - dH[i];                        // :117
      if (a.beta != 0.0 && vr) g += 0.5 * a.beta * vr[2 + i] / sqrt(varG);  // :129
      const double gF = g + dP[i];
      o[OUT_HDR + i] = gF;
      o[OUT_HDR + T + i] = dG[i];
      o[OUT_HDR + 2 * T + i] = dH[i];
      if (!adam) return 0.0;
      return running ? adam_elem(a.next_A, a.next_iter, r, i, gF, a.next_theta) : a.next_theta[(size_t)r * T + i];
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * nt;
      xn_keep[u] = i < T ? elem(i) : 0.0;
    }
    for (int i = tid + 4 * nt; i < T; i += nt) elem(i);
  }
  FIN_STAMP(3);
  if (adam) {
    __syncthreads();                                     // every thread is done with dG / dH / dP: their LDS becomes theta's
    double* xl = lds;
    if (keep) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (tid + u * nt < T) xl[tid + u * nt] = xn_keep[u];
      __syncthreads();
    }
    FIN_STAMP(4);
    if (keep) prep_body<true>(dm, a.next_theta, a.next_vpfix, a.next_vpd, a.next_entp, r, lds, xl);
    else prep_body<false>(dm, a.next_theta, a.next_vpfix, a.next_vpd, a.next_entp, r, lds, nullptr);
  }
  FIN_STAMP(5);
}
