// Bayesian-quadrature variance of the expected log joint on gfx950.
//
// Reference: misc/gplogjoint.m:273-337 (diag and full K x K variance), :375-413 (gradient of the
// diagonal variance, hyper-sample averaging).  The reference re-solves two N x N triangular systems
// for every (j,k) pair -- O(S K^2 N^2).  Here every hyper-sample does ONE blocked triangular solve
// V = L' \ Z for all K right-hand sides and then a K x K Gram matrix, O(S (N^2 K + K^2 N)); results
// agree to summation order.
//
//   k_var_z       Z[r][s][k][:] = z_k  (gplogjoint.m:164-168)
//   k_trsm_fwd    V = L' \ Z   (L upper, MATLAB chol convention): trsm_mfma.h, one wave per 16 columns, MFMA f64
//   k_trsm_bwd    X = L \ V    (only for the variance gradient: invKzk = X / sn2_eff, :277)
//   k_symm        U = L * Z    (Lchol == false: L = -inv(K + sn2 I), :279,:321)
//   k_var_gram    J[r][s][j][k] (:281,:318-322) and the raw varF(s) terms
//   k_vargrad     raw dots  dz_dtheta * invKzk  (:289-301)
//   k_var_final   varF, varss, dvarF with Jacobians and averaging (:350,:375-413)
#pragma once
#include "common.h"
#include "device_math.h"
#include "elbo_kernels.h"
#include "trsm_mfma.h"

#define TR_CB 16   // right-hand-side columns per workgroup
#define TR_B 16    // row block

// ------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(WAVE) k_var_z(ElboDims dm, const double* __restrict__ vpd,
                                                const double* __restrict__ X, const double* __restrict__ gpc,
                                                const double* __restrict__ delta2, double* __restrict__ Z) {
  const int k = blockIdx.x, s = blockIdx.y, r = blockIdx.z, lane = threadIdx.x;
  const int D = dm.D, K = dm.K, N = dm.N;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = gpc + (size_t)s * GPC_STRIDE(D);
  const double sig = v[L.sigma() + k];
  double mu[DT], itau[DT];
  double sumlogtau = 0.0;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    if (d < D) {
      double lam = v[L.lambda() + d];
      mu[d] = v[L.mu() + d + D * k];
      double tau = sqrt(sig * sig * lam * lam + g[d] + delta2[d]);
      sumlogtau += log(tau);
      itau[d] = 1.0 / tau;
    } else { mu[d] = 0.0; itau[d] = 0.0; }
  }
  const double lnnf = g[3 * D] - sumlogtau;
  double* z = Z + (((size_t)r * dm.S + s) * K + k) * N;
  for (int n = lane; n < N; n += WAVE) {
    double a2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      double x = (d < D) ? X[n + (size_t)N * d] : 0.0;
      double dl = (mu[d] - x) * itau[d];
      a2 = fma(dl, dl, a2);
    }
    z[n] = vb_exp(lnnf - 0.5 * a2);
  }
}

// U = Lm * Z for hyper-samples with Lchol == false (Lm = -inv(K + sn2 I), full symmetric N x N)
__global__ void __launch_bounds__(256) k_symm(int N, int K, int S, const double* __restrict__ Lall,
                                              const unsigned char* __restrict__ lchol,
                                              const double* __restrict__ Z, double* __restrict__ U) {
  const int s = blockIdx.y, r = blockIdx.z;
  if (lchol[s]) return;
  const double* Lm = Lall + (size_t)s * N * N;
  const double* Zs = Z + ((size_t)r * S + s) * (size_t)K * N;
  double* Us = U + ((size_t)r * S + s) * (size_t)K * N;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N * K; idx += gridDim.x * blockDim.x) {
    int i = idx % N, k = idx / N;
    double acc = 0.0;
    for (int j = 0; j < N; ++j) acc = fma(Lm[(size_t)j * N + i], Zs[(size_t)k * N + j], acc);  // symmetric: L[i][j] = L[j][i]
    Us[idx] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// J[r][s][j][k] for j <= k (mirrored), diag only if !full.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_var_gram(ElboDims dm, const double* __restrict__ vpd,
                                                  const double* __restrict__ gpc, const double* __restrict__ delta2,
                                                  const double* __restrict__ sn2_eff, const unsigned char* __restrict__ lchol,
                                                  const double* __restrict__ ZV,   // L'\Z in place (Lchol) / raw z (!Lchol)
                                                  const double* __restrict__ XU,   // L*Z for !Lchol samples (else unused)
                                                  double* __restrict__ J, int full) {
  const int s = blockIdx.y, r = blockIdx.z;
  const int D = dm.D, K = dm.K, N = dm.N;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = gpc + (size_t)s * GPC_STRIDE(D);
  const double* Zs = ZV + ((size_t)r * dm.S + s) * (size_t)K * N;
  const bool lc = lchol[s] != 0;
  const double* Vs = (lc ? ZV : XU) + ((size_t)r * dm.S + s) * (size_t)K * N;
  double* Js = J + ((size_t)r * dm.S + s) * (size_t)K * K;
  const int lane = threadIdx.x & 63, wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  const int npair = full ? K * K : K;
  for (int p = wv; p < npair; p += nw) {
    int j, k;
    if (full) { j = p % K; k = p / K; if (j > k) continue; } else { j = k = p; }
    double dot = 0.0;
    if (lc) {
      for (int n = lane; n < N; n += 64) dot = fma(Vs[(size_t)j * N + n], Vs[(size_t)k * N + n], dot);
    } else {
      for (int n = lane; n < N; n += 64) dot = fma(Zs[(size_t)k * N + n], Vs[(size_t)j * N + n], dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) {
      double sj = v[L.sigma() + j], sk = v[L.sigma() + k];
      double slt = 0.0, d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        double lam = v[L.lambda() + d];
        double t2 = (sj * sj + sk * sk) * lam * lam + g[d] + 2.0 * delta2[d];  // tau_jk^2 (:313), tau_kk (:274)
        slt += log(sqrt(t2));
        double dm_ = v[L.mu() + d + D * j] - v[L.mu() + d + D * k];
        d2 += dm_ * dm_ / t2;
      }
      double nf = exp(g[3 * D] - slt - 0.5 * d2);
      double val = lc ? nf - dot / sn2_eff[s] : nf + dot;  // :318-322
      Js[j + (size_t)K * k] = val;
      Js[k + (size_t)K * j] = val;
    }
  }
}

// V = L' \ Z as a PRODUCT with the explicit triangular inverse T = inv(L') (S x N x N, element (i, n) at n N + i, formed once per
// surrogate for gplite_pred: ||inv(L')|| <= 1 since L'L = K/sl + I >= I) instead of a substitution: k_trsm_fwd walks the 25 block
// rows of N = 400 one after the other on 80 waves (142 us, latency-bound); the product has no dependence between output tiles.
// One wave per 16 x 16 tile of V (components k x rows i), inner index n <= i in steps of four: Vout[k][i] = sum_n T[i][n] Z[k][n].
// Lchol samples only (the others go through k_symm).  Out of place.
__global__ void __launch_bounds__(64) k_tri_gemm(int N, int K, int S, const double* __restrict__ Tall, const unsigned char* __restrict__ lchol,
                                                 const double* __restrict__ Z, double* __restrict__ Vout) {
  typedef double vg4 __attribute__((ext_vector_type(4)));
  const int nkb = (K + 15) >> 4;
  const int ib = blockIdx.x / nkb, kb = blockIdx.x - ib * nkb, s = blockIdx.y, r = blockIdx.z;
  if (!lchol[s]) return;
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const double* T = Tall + (size_t)s * N * N;
  const double* Zs = Z + ((size_t)r * S + s) * (size_t)K * N;
  double* Vs = Vout + ((size_t)r * S + s) * (size_t)K * N;
  const int i0 = 16 * ib, k0 = 16 * kb;
  const int ka = min(k0 + li, K - 1), ii = min(i0 + li, N - 1);
  const double* pa = Zs + (size_t)ka * N + lg;          // A operand: row m = component k0 + li, inner index n0 + lg
  const double* pb = T + (size_t)lg * N + ii;           // B operand: inner index n0 + lg, column = row i0 + li of V
  vg4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
  const int nend = min(i0 + 16, N);                     // T[i][n] = 0 for n > i
  int n = 0;
  for (; n + 32 <= nend; n += 32) {     // sixteen loads in flight
    double av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { av[u] = pa[n + 4 * u]; bv[u] = pb[(size_t)(n + 4 * u) * N]; }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bv[u + 1], acc2, 0, 0, 0);
    }
  }
  for (; n + 8 <= nend; n += 8) {
    const double a0 = pa[n], b0 = pb[(size_t)n * N], a1 = pa[n + 4], b1 = pb[(size_t)(n + 4) * N];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc2, 0, 0, 0);
  }
  for (; n < nend; n += 4) {
    const bool in = n + lg < N;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? pa[n] : 0.0, in ? pb[(size_t)n * N] : 0.0, acc, 0, 0, 0);
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {                      // D[m = lg + 4 rr -> component][n = li -> row i]
    const int k = k0 + lg + 4 * rr, i = i0 + li;
    if (k < K && i < N) Vs[(size_t)k * N + i] = acc[rr] + acc2[rr];
  }
}

// The full K x K variance matrix with the Gram products on the matrix cores (round 2): k_var_gram spends one wave per pair on
// a 7-trip dot product and lane 0 on the pair's scalar term (194 us at K = 50, N = 400, S = 20: 39 % of eval_fullelcbo).  Here one
// 1024-thread workgroup per (hyper-sample, restart): phase A, the 16 x 16 tiles of V'V (or Z'U for Lchol == false) on or
// above the diagonal dealt to the 16 waves, inner dimension N in steps of four, raw dots parked in J; phase B, one THREAD per
// pair (j <= k) for nf_jk (gplogjoint.m:313-317) and the combination (:318-322), mirrored.
__global__ void __launch_bounds__(1024) k_var_gram_mfma(ElboDims dm, const double* __restrict__ vpd,
                                                        const double* __restrict__ gpc, const double* __restrict__ delta2,
                                                        const double* __restrict__ sn2_eff, const unsigned char* __restrict__ lchol,
                                                        const double* __restrict__ ZV, const double* __restrict__ XU,
                                                        double* __restrict__ J, int lc_in_x) {   // lc_in_x: V of the Lchol samples is in XU (k_tri_gemm)
  typedef double vg4 __attribute__((ext_vector_type(4)));
  const int s = blockIdx.x, r = blockIdx.y;
  const int D = dm.D, K = dm.K, N = dm.N;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = gpc + (size_t)s * GPC_STRIDE(D);
  const double* Zs = ZV + ((size_t)r * dm.S + s) * (size_t)K * N;
  const bool lc = lchol[s] != 0;
  const double* Vs = ((lc && !lc_in_x) ? ZV : XU) + ((size_t)r * dm.S + s) * (size_t)K * N;
  if (lc) Zs = Vs;                                  // Lchol: both operands are V
  double* Js = J + ((size_t)r * dm.S + s) * (size_t)K * K;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int nt = (K + 15) >> 4, ntile = nt * (nt + 1) / 2;
  // ---- phase A: dot[j][k] = sum_n A_j[n] B_k[n] for j <= k (tile-wise), A = V (Lchol) or U, B = V or Z
  for (int t = wv; t < ntile; t += 16) {
    int tk = 0;
    while ((tk + 1) * (tk + 2) / 2 <= t) ++tk;     // tile (tj, tk), tj <= tk: t = tk (tk + 1) / 2 + tj
    const int tj = t - tk * (tk + 1) / 2;
    const int ja = min(16 * tj + li, K - 1), kb = min(16 * tk + li, K - 1);
    const double* pa = Vs + (size_t)ja * N + lg;   // MFMA A operand: row j = li, inner index lg
    const double* pb = Zs + (size_t)kb * N + lg;   //      B operand: inner index lg, column k = li   (Lchol: Zs == Vs)
    vg4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    int n = 0;
    for (; n + 32 <= N; n += 32) {      // sixteen strided loads in flight: the loop is bound by their latency, not by the MFMAs
      double av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { av[u] = pa[n + 4 * u]; bv[u] = pb[n + 4 * u]; }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u + 1], bv[u + 1], acc2, 0, 0, 0);
      }
    }
    for (; n + 8 <= N; n += 8) {
      const double a0 = pa[n], b0 = pb[n], a1 = pa[n + 4], b1 = pb[n + 4];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc2, 0, 0, 0);
    }
    for (; n < N; n += 4) {
      const bool in = n + lg < N;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? pa[n] : 0.0, in ? pb[n] : 0.0, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {               // D[m = lg + 4 rr][n = li]
      const int j = 16 * tj + lg + 4 * rr, k = 16 * tk + li;
      if (j < K && k < K && j <= k) Js[j + (size_t)K * k] = acc[rr] + acc2[rr];
    }
  }
  __syncthreads();
  // ---- phase B: one thread per pair
  const double sn2 = sn2_eff[s];
  for (int p = tid; p < K * K; p += 1024) {
    const int j = p % K, k = p / K;
    if (j > k) continue;
    const double dot = Js[j + (size_t)K * k];
    const double sj = v[L.sigma() + j], sk = v[L.sigma() + k];
    double slt = 0.0, d2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double lam = v[L.lambda() + d];
      const double t2 = (sj * sj + sk * sk) * lam * lam + g[d] + 2.0 * delta2[d];  // tau_jk^2 (:313), tau_kk (:274)
      slt += log(sqrt(t2));
      const double dm_ = v[L.mu() + d + D * j] - v[L.mu() + d + D * k];
      d2 += dm_ * dm_ / t2;
    }
    const double nf = exp(g[3 * D] - slt - 0.5 * d2);
    const double val = lc ? nf - dot / sn2 : nf + dot;  // :318-322
    Js[j + (size_t)K * k] = val;
    if (j != k) Js[k + (size_t)K * j] = val;
  }
}

// raw dots of the variance gradient for compute_var == 2: one wave per (k, s, r)
// VG[r][s][k][2D+1] = dz_dmu*x [D], dz_dsigma*x, dz_dlambda*x [D]   with x = invKzk
template <int DT>
__global__ void __launch_bounds__(WAVE) k_vargrad(ElboDims dm, const double* __restrict__ vpd,
                                                  const double* __restrict__ X, const double* __restrict__ gpc,
                                                  const double* __restrict__ delta2, const double* __restrict__ sn2_eff,
                                                  const unsigned char* __restrict__ lchol,
                                                  const double* __restrict__ Xsol,  // L\(L'\z) (Lchol) or L*z (!Lchol)
                                                  double* __restrict__ vg) {
  const int k = blockIdx.x, s = blockIdx.y, r = blockIdx.z, lane = threadIdx.x;
  const int D = dm.D, K = dm.K, N = dm.N;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = gpc + (size_t)s * GPC_STRIDE(D);
  const double sig = v[L.sigma() + k];
  const double xs = lchol[s] ? 1.0 / sn2_eff[s] : -1.0;  // invKzk = X/sn2_eff  or  -L z
  double mu[DT], itau[DT], lam[DT];
  double sumlogtau = 0.0;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    if (d < D) {
      lam[d] = v[L.lambda() + d];
      mu[d] = v[L.mu() + d + D * k];
      double tau = sqrt(sig * sig * lam[d] * lam[d] + g[d] + delta2[d]);
      sumlogtau += log(tau);
      itau[d] = 1.0 / tau;
    } else { lam[d] = 0.0; mu[d] = 0.0; itau[d] = 0.0; }
  }
  const double lnnf = g[3 * D] - sumlogtau;
  double accS = 0.0, accM[DT], accL[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { accM[d] = 0.0; accL[d] = 0.0; }
  const double* xv = Xsol + (((size_t)r * dm.S + s) * K + k) * N;
  for (int n = lane; n < N; n += WAVE) {
    double dl[DT];
    double a2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      double x = (d < D) ? X[n + (size_t)N * d] : 0.0;
      dl[d] = (mu[d] - x) * itau[d];
      a2 = fma(dl[d], dl[d], a2);
    }
    double za = vb_exp(lnnf - 0.5 * a2) * (xv[n] * xs);
    double ssum = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      double li = lam[d] * itau[d], si = sig * itau[d];
      double q = fma(dl[d], dl[d], -1.0);
      accM[d] = fma(-dl[d] * itau[d], za, accM[d]);
      ssum = fma(li * li, q, ssum);
      accL[d] = fma(si * si * q * lam[d], za, accL[d]);
    }
    accS = fma(ssum * sig, za, accS);
  }
  accS = wave_sum(accS);
#pragma unroll
  for (int d = 0; d < DT; ++d) { accM[d] = wave_sum(accM[d]); accL[d] = wave_sum(accL[d]); }
  if (lane == 0) {
    double* o = vg + (((size_t)r * dm.S + s) * K + k) * (2 * D + 1);
#pragma unroll
    for (int d = 0; d < DT; ++d)
      if (d < D) { o[d] = accM[d]; o[D + 1 + d] = accL[d]; }
    o[D] = accS;
  }
}

// ------------------------------------------------------------------------------------------
// k_per_sample: the avg_flag = 0 outputs of gplogjoint -- F(s) = sum_k w_k I_k (gplogjoint.m:203) and, with the
// variance, varF(s) accumulated over (j <= k) in the reference's order with the diagonal clamp (:283, :329-332) and the
// final max(varF, eps) (:350).  One thread per (hyper-sample, restart); O(K^2) each.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_per_sample(ElboDims dm, const double* __restrict__ vpd, const double* __restrict__ lj,
                                                    const double* __restrict__ J, int compute_var, double* __restrict__ Gs,
                                                    double* __restrict__ vGs) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  const int D = dm.D, K = dm.K, S = dm.S;
  if (s >= S) return;
  const double EPS = 2.220446049250313e-16;
  VpLayout L{D, K};
  const double* w = vpd + (size_t)r * L.stride() + L.w();
  const int LJS = 2 * D + 2;
  const double* l = lj + ((size_t)r * S + s) * K * LJS;
  double F = 0.0;
  for (int k = 0; k < K; ++k) F += w[k] * l[(size_t)k * LJS];
  Gs[s + (size_t)S * r] = F;
  if (compute_var && vGs) {
    const double* Js = J + ((size_t)r * S + s) * K * K;
    double v = 0.0;
    for (int k = 0; k < K; ++k) {
      if (compute_var == 2) { v += w[k] * w[k] * fmax(EPS, Js[k + (size_t)K * k]); continue; }
      for (int j = 0; j <= k; ++j) {
        const double Jjk = Js[j + (size_t)K * k];
        v += (j == k) ? w[k] * w[k] * fmax(EPS, Jjk) : 2.0 * w[j] * w[k] * Jjk;
      }
    }
    vGs[s + (size_t)S * r] = fmax(v, EPS);
  }
}

// ------------------------------------------------------------------------------------------
// k_per_sample_grad: gplogjoint's dF with avg_flag = 0 -- the gradient of F(s) for every hyper-sample on its own, T x S per restart
// (misc/gplogjoint.m:206-271 hold the per-sample pieces; :352-373 the Jacobians; the averaging of :411 is what is skipped).  The
// per-(s, k) records carry I_k | w_k dI_k/dmu (D) | w_k dI_k/dsigma | w_k dI_k/dlambda (D): mu and sigma blocks are copies (times
// sigma_k for log sigma), lambda sums over the components (times lambda_d for log lambda), the weight gradient is I_k itself, through
// the softmax Jacobian diag(w) - w w' for eta (:366-368): w_k (I_k - F(s)).  One workgroup per (hyper-sample, restart).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_per_sample_grad(ElboDims dm, const double* __restrict__ vpd, const double* __restrict__ lj,
                                                         int no_jacobian, double* __restrict__ dGs) {
  const int s = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
  const int D = dm.D, K = dm.K, S = dm.S, T = dm.T;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double *w = v + L.w(), *sigma = v + L.sigma(), *lam = v + L.lambda();
  const int LJS = 2 * D + 2;
  const double* l = lj + ((size_t)r * S + s) * K * LJS;
  double* o = dGs + ((size_t)r * S + s) * T;
  const bool jac = no_jacobian == 0;
  __shared__ double Fs;
  if (tid == 0) {
    double F = 0.0;
    for (int k = 0; k < K; ++k) F += w[k] * l[(size_t)k * LJS];     // :203, the order of k_per_sample
    Fs = F;
  }
  __syncthreads();
  if (dm.opt[0])
    for (int p = tid; p < D * K; p += nt) o[dm.off_mu + p] = l[(size_t)(p / D) * LJS + 1 + p % D];
  if (dm.opt[1])
    for (int k = tid; k < K; k += nt) o[dm.off_sigma + k] = l[(size_t)k * LJS + 1 + D] * (jac ? sigma[k] : 1.0);        // :356
  if (dm.opt[2])
    for (int d = tid; d < D; d += nt) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += l[(size_t)k * LJS + 2 + D + d];                                               // :250
      o[dm.off_lambda + d] = acc * (jac ? lam[d] : 1.0);                                                                // :362
    }
  if (dm.opt[3])
    for (int k = tid; k < K; k += nt) o[dm.off_eta + k] = jac ? w[k] * l[(size_t)k * LJS] - w[k] * Fs : l[(size_t)k * LJS];   // :366-368
}

// ------------------------------------------------------------------------------------------
// k_var_final: one workgroup per restart.  out VR[r] = varG, varGss, dvarG[T]
// ------------------------------------------------------------------------------------------
struct VarFinArgs {
  ElboDims dm;
  const double* vpd;
  const double* gpc;
  const double* delta2;
  const double* lj;   // R x S x K x (2D+2)  (per-sample I_k and gradient pieces)
  const double* J;    // R x S x K x K
  const double* vg;   // R x S x K x (2D+1) or null
  int compute_var, want_grad, stride;
  double* out;
  int no_jacobian;     // 1: the variance gradient with respect to sigma, lambda, w themselves (misc/gplogjoint.m:375-396 skipped)
  double* dvs_out;     // null, or R x S x T: the per-hyper-sample variance gradient dvarF(:, s) (avg_flag = 0: :407-409 skipped)
  double* vs;          // R x S x 2 x T scratch: per hyper-sample dF(:, s) and dvarF(:, s) after the Jacobians (k_var_sample -> k_var_final)
};

// k_var_sample (round 5): the per-hyper-sample gradients dF(:, s) and dvarF(:, s) with their Jacobians (misc/gplogjoint.m:286-303,
// 352-390), one workgroup per (hyper-sample, restart).  They were a loop over s inside k_var_final's single workgroup per restart --
// seven barriers and half a dozen dependent round trips to memory per sample, S = 20 times in a row: 464 us of a 0.95 ms
// evaluation at the headline GP shape.  k_var_final now only adds the S vectors up (in sample order).
#define VARSMP_THREADS 256
__global__ void __launch_bounds__(VARSMP_THREADS) k_var_sample(VarFinArgs a) {
  extern __shared__ double lds[];
  const int s = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
  const ElboDims& dm = a.dm;
  const int D = dm.D, K = dm.K, S = dm.S, T = dm.T;
  const double EPS = 2.220446049250313e-16;
  VpLayout L{D, K};
  const double* v = a.vpd + (size_t)r * L.stride();
  const double* w = v + L.w();
  const double* sigma = v + L.sigma();
  const double* lam = v + L.lambda();
  double* red = lds;          // nt
  double* dFs = red + nt;     // T
  double* dvs = dFs + T;      // T
  double* tmpK = dvs + T;     // K
  double* tmpK2 = tmpK + K;   // K
  const int LJS = 2 * D + 2;
  const double* lj = a.lj + (size_t)r * S * K * LJS;
  const double* Js = a.J + ((size_t)r * S + s) * K * K;
  const bool jac = a.no_jacobian == 0;
  const double* g = a.gpc + (size_t)s * GPC_STRIDE(D);
  const double* vg = a.vg + ((size_t)r * S + s) * (size_t)K * (2 * D + 1);
  for (int i = tid; i < T; i += nt) { dFs[i] = 0.0; dvs[i] = 0.0; }
  __syncthreads();
  // per-sample value gradient dF(:,s) after Jacobians (:352-373)
  if (dm.opt[0]) for (int p = tid; p < D * K; p += nt) dFs[dm.off_mu + p] = lj[((size_t)s * K + p / D) * LJS + 1 + p % D];
  if (dm.opt[1]) for (int k = tid; k < K; k += nt) dFs[dm.off_sigma + k] = lj[((size_t)s * K + k) * LJS + 1 + D] * (jac ? sigma[k] : 1.0);
  if (dm.opt[2])
    for (int d = tid; d < D; d += nt) {
      double ls = 0.0;
      for (int k = 0; k < K; ++k) ls += lj[((size_t)s * K + k) * LJS + 2 + D + d];
      dFs[dm.off_lambda + d] = ls * (jac ? lam[d] : 1.0);
    }
  // variance gradient pieces (:286-303)
  if (dm.opt[0])
    for (int p = tid; p < D * K; p += nt) {
      int d = p % D, k = p / D;
      dvs[dm.off_mu + p] = -w[k] * w[k] * (2.0 * vg[(size_t)k * (2 * D + 1) + d]);  // :289
    }
  // nf_kk and sum_d lambda_d^2 / tau_kk,d^2 once per component (:274-275,293)
  for (int k = tid; k < K; k += nt) {
    double slt = 0.0, sl2 = 0.0;
    for (int d = 0; d < D; ++d) {
      double t2 = 2.0 * sigma[k] * sigma[k] * lam[d] * lam[d] + g[d] + 2.0 * a.delta2[d];
      slt += log(sqrt(t2));
      sl2 += lam[d] * lam[d] / t2;
    }
    tmpK[k] = exp(g[3 * D] - slt);
    tmpK2[k] = sl2;
  }
  __syncthreads();
  if (dm.opt[1])
    for (int k = tid; k < K; k += nt)
      dvs[dm.off_sigma + k] = -2.0 * w[k] * w[k] * (sigma[k] * tmpK[k] * tmpK2[k] + vg[(size_t)k * (2 * D + 1) + D]) * (jac ? sigma[k] : 1.0);  // :293, Jacobian :382
  if (dm.opt[2])
    for (int d = tid; d < D; d += nt) {
      double accd = 0.0;
      for (int k = 0; k < K; ++k) {
        double t2 = 2.0 * sigma[k] * sigma[k] * lam[d] * lam[d] + g[d] + 2.0 * a.delta2[d];
        accd -= 2.0 * w[k] * w[k] * (sigma[k] * sigma[k] * tmpK[k] * lam[d] / t2 + vg[(size_t)k * (2 * D + 1) + D + 1 + d]);  // :297
      }
      dvs[dm.off_lambda + d] = accd * (jac ? lam[d] : 1.0);  // Jacobian :386
    }
  __syncthreads();
  if (dm.opt[3]) {
    // softmax Jacobian on w_grad = I_k and on w_vargrad = 2 w_k max(eps, J_kk)  (:301, :366-372, :390)
    double p1 = 0.0, p2 = 0.0;
    for (int k = tid; k < K; k += nt) {
      double ik = lj[((size_t)s * K + k) * LJS];
      double wv = 2.0 * w[k] * fmax(EPS, Js[k + (size_t)K * k]);
      tmpK[k] = wv;
      p1 += w[k] * ik;
      p2 += w[k] * wv;
    }
    double d1 = block_sum(p1, red);
    double d2 = block_sum(p2, red);
    for (int k = tid; k < K; k += nt) {
      double ik = lj[((size_t)s * K + k) * LJS];
      dFs[dm.off_eta + k] = jac ? w[k] * ik - w[k] * d1 : ik;
      dvs[dm.off_eta + k] = jac ? w[k] * tmpK[k] - w[k] * d2 : tmpK[k];
    }
  }
  __syncthreads();
  double* o = a.vs + ((size_t)r * S + s) * 2 * (size_t)T;
  for (int i = tid; i < T; i += nt) { o[i] = dFs[i]; o[T + i] = dvs[i]; }
  if (a.dvs_out) for (int i = tid; i < T; i += nt) a.dvs_out[((size_t)r * S + s) * T + i] = dvs[i];
}
#define VAR_SAMPLE_LDS(K, T) ((size_t)(VARSMP_THREADS + 2 * (size_t)(T) + 2 * (size_t)(K)) * sizeof(double))

#define VARFIN_THREADS 1024
__global__ void __launch_bounds__(VARFIN_THREADS) k_var_final(VarFinArgs a) {
  extern __shared__ double lds[];
  const int r = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const ElboDims& dm = a.dm;
  const int D = dm.D, K = dm.K, S = dm.S, T = dm.T;
  const double EPS = 2.220446049250313e-16;
  VpLayout L{D, K};
  const double* v = a.vpd + (size_t)r * L.stride();
  const double* w = v + L.w();
  double* red = lds;          // nt
  double* Fs = red + nt;      // S
  double* vFs = Fs + S;       // S
  const int Tg = a.want_grad ? T : 0;   // the gradient vectors exist only when a gradient is wanted (host sizing follows: five T-vectors)
  double* acc1 = vFs + S;     // T  sum_s dvarF(:,s)
  double* acc2 = acc1 + Tg;   // T  sum_s F(s) dF(:,s)
  double* acc3 = acc2 + Tg;   // T  sum_s dF(:,s)
  const int LJS = 2 * D + 2;
  const double* lj = a.lj + (size_t)r * S * K * LJS;
  const double* Jr = a.J + (size_t)r * S * K * K;
  double* o = a.out + (size_t)r * a.stride;
  const bool vgrad = a.want_grad && a.compute_var == 2;
  // ---- per-hyper-sample F(s) and varF(s) (:203, :283, :329-332, :350): the S samples are independent, one WAVE each (16 at a
  // time), lanes along the component pairs, fixed-order wave butterfly -- instead of S sequential workgroup-wide reductions
  // (a single full-ELCBO evaluation spent 0.26-0.39 ms of its 0.72 ms here)
  double* wl = red;            // the weights in LDS for the pair loop below (red is free until the reductions further down)
  for (int k = tid; k < K; k += nt) wl[k] = w[k];
  __syncthreads();
  {
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    for (int s = wave; s < S; s += nw) {
      const double* Js = Jr + (size_t)s * K * K;
      double part = 0.0;
      if (a.compute_var == 2) {
        for (int k = lane; k < K; k += 64) part += w[k] * w[k] * fmax(EPS, Js[k + (size_t)K * k]);
      } else {
        // all K^2 entries of the (mirrored) symmetric matrix, coalesced along j, eight loads in flight per lane: the loop over the
        // upper triangle column by column was a chain of K dependent load latencies (64 us of a 0.39 ms evaluation)
        const int KK = K * K;
        for (int p0 = 0; p0 < KK; p0 += 8 * 64) {
          double jv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int p = p0 + 64 * u + lane; jv[u] = p < KK ? Js[p] : 0.0; }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int p = p0 + 64 * u + lane;
            if (p < KK) {
              const int k = p / K, j = p - k * K;
              part = fma(wl[j] * wl[k], (j == k) ? fmax(EPS, jv[u]) : jv[u], part);
            }
          }
        }
      }
      const double vf = wave_sum(part);
      part = 0.0;
      for (int k = lane; k < K; k += 64) part += w[k] * lj[((size_t)s * K + k) * LJS];
      const double fs = wave_sum(part);
      if (lane == 0) { vFs[s] = fmax(vf, EPS); Fs[s] = fs; }
    }
  }
  __syncthreads();
  if (vgrad) {
    // sum_s dvarF(:, s), sum_s F(s) dF(:, s), sum_s dF(:, s) from k_var_sample's vectors, in sample order, eight loads in flight
    const double* vsr = a.vs + (size_t)r * S * 2 * (size_t)T;
    for (int i = tid; i < T; i += nt) {
      double a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int s0 = 0; s0 < S; s0 += 4) {
        double df[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = s0 + u < S;
          df[u] = ok ? vsr[(size_t)(s0 + u) * 2 * T + i] : 0.0;
          dv[u] = ok ? vsr[(size_t)(s0 + u) * 2 * T + T + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s0 + u < S) { a1 += dv[u]; a2 += Fs[s0 + u] * df[u]; a3 += df[u]; }
      }
      acc1[i] = a1; acc2[i] = a2; acc3[i] = a3;
    }
    __syncthreads();
  }
  // averaging (:399-413)
  if (tid == 0) {
    double varF, varss = 0.0;
    if (S > 1) {
      double Fbar = 0.0;
      for (int s = 0; s < S; ++s) Fbar += Fs[s];
      Fbar /= S;
      double vss = 0.0, vm = 0.0;
      for (int s = 0; s < S; ++s) { vss += (Fs[s] - Fbar) * (Fs[s] - Fbar); vm += vFs[s]; }
      vss /= (S - 1);
      double mean = vm / S, sd = 0.0;
      for (int s = 0; s < S; ++s) sd += (vFs[s] - mean) * (vFs[s] - mean);
      sd = sqrt(sd / (S - 1));        // MATLAB std()
      varss = vss + sd;               // :404 (variance + std, as in the reference)
      varF = vm / S + vss;            // :405
      red[0] = Fbar;
    } else {
      varF = vFs[0];
      red[0] = Fs[0];
    }
    o[0] = varF;
    o[1] = varss;
  }
  __syncthreads();
  if (vgrad) {
    const double Fbar = red[0];
    for (int i = tid; i < T; i += nt) {
      double dv;
      if (S > 1) {
        double dvv = 2.0 * acc2[i] / (S - 1) - 2.0 * Fbar * acc3[i] / (S - 1);  // :408
        dv = acc1[i] / S + dvv;                                                  // :409
      } else {
        dv = acc1[i];
      }
      o[2 + i] = dv;
    }
  }
}
