/*
 * vbmc_oracle.c -- plain-C restatement of the metric path, TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 *
 * negelcbo_vbmc(theta, 0, vp, gp, Ns, 1, 0) = -gplogjoint - entmc_vbmc, value + gradient, with the
 * reference's loop structure (outer loop over source components j, inner loop over mixture components
 * k, hyper-sample loop s; reference: ent/entmc_vbmc.m:49-125, misc/gplogjoint.m:92-271,352-373,
 * 399-413, misc/negelcbo_vbmc.m:33-48,116-117; acerbilab/vbmc v1.0.12).  It exists so that
 * bench.py's cpu_baseline leg can time a compiled port (1 thread, or all cores with -fopenmp) next to
 * the NumPy restatement, and so that tests can check larger shapes quickly.  Nothing under
 * vbmc_amd/ links or calls it.  PARITY UNPINNED by the reference (no MATLAB here); this file is pinned
 * against oracle/vbmc_ref.py and the mpmath golden vectors by tests/test_oracle_c.py.
 *
 * Layouts are MATLAB's: mu D x K column-major, X N x D column-major, eps D x Mh x K (d fastest).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* team size for the OpenMP build (the caller passes the cores the process may actually use: cgroup quota / affinity) */
void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* softmax Jacobian applied to a vector: (diag(w) - w w') g, w = exp(eta)/sum  (entmc_vbmc.m:121-123) */
static void softmax_jac_apply(int K, const double* eta, const double* g, double* out) {
  double* e = (double*)malloc(sizeof(double) * K);
  double ssum = 0.0;
  for (int k = 0; k < K; ++k) { e[k] = exp(eta[k]); ssum += e[k]; }
  for (int a = 0; a < K; ++a) {
    double acc = 0.0;
    for (int b = 0; b < K; ++b) acc += (-e[a] * e[b] / (ssum * ssum) + (a == b ? e[a] / ssum : 0.0)) * g[b];
    out[a] = acc;
  }
  free(e);
}

/* ent/entmc_vbmc.m:1-128 with explicit eps; dH = [mu(:); sigma; lambda; eta] (all four groups).
   As in the reference, everything that depends on the mixture only is formed once outside the sample loops (sigmalambda :34,
   nf ./ sigma.^D :61-63, sigmalambda.^2 :78); the samples of one component are split over the OpenMP team, one parallel region
   for the whole evaluation. */
void oracle_entmc(int D, int K, int Mh, const double* mu, const double* sigma, const double* lambda, const double* w,
                  const double* eta, const double* eps, int grad, double* H_out, double* dH) {
  const int Ns = 2 * Mh;
  double prodlam = 1.0;
  for (int d = 0; d < D; ++d) prodlam *= lambda[d];
  const double nf = 1.0 / pow(2.0 * PI, D / 2.0) / prodlam; /* :40 */
  double H = 0.0;
  double* mu_g = (double*)calloc((size_t)D * K, sizeof(double));
  double* sg_g = (double*)calloc(K, sizeof(double));
  double* lam_g = (double*)calloc(D, sizeof(double));
  double* w_g = (double*)calloc(K, sizeof(double));
  double* sl = (double*)malloc(sizeof(double) * D * K);   /* sigma_k lambda_d        :34 */
  double* sl2 = (double*)malloc(sizeof(double) * D * K);  /* (sigma_k lambda_d)^2    :78 */
  double* nfk = (double*)malloc(sizeof(double) * K);      /* nf / sigma_k^D          :61-63 */
  for (int k = 0; k < K; ++k) {
    nfk[k] = nf / pow(sigma[k], D);
    for (int d = 0; d < D; ++d) { sl[d + (size_t)D * k] = sigma[k] * lambda[d]; sl2[d + (size_t)D * k] = sl[d + (size_t)D * k] * sl[d + (size_t)D * k]; }
  }
  /* per-component accumulators, shared by the team */
  double Hj = 0.0, sgj = 0.0, wjlog = 0.0;
  double* muj = (double*)calloc(D, sizeof(double));
  double* lamj = (double*)calloc(D, sizeof(double));
  double* wl = (double*)calloc(K, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    double* x = (double*)malloc(sizeof(double) * D);
    double* e = (double*)malloc(sizeof(double) * D);
    double* nrm = (double*)malloc(sizeof(double) * K);
    double* muj_p = (double*)malloc(sizeof(double) * D);
    double* lamj_p = (double*)malloc(sizeof(double) * D);
    double* wl_p = (double*)malloc(sizeof(double) * K);
    for (int j = 0; j < K; ++j) { /* :49 */
      double Hj_p = 0.0, sgj_p = 0.0, wjlog_p = 0.0;
      for (int d = 0; d < D; ++d) { muj_p[d] = 0.0; lamj_p[d] = 0.0; }
      for (int l = 0; l < K; ++l) wl_p[l] = 0.0;
#ifdef _OPENMP
#pragma omp for schedule(static) nowait
#endif
      for (int i = 0; i < Ns; ++i) {
        const int b = i < Mh ? i : i - Mh;
        const double sgn = i < Mh ? 1.0 : -1.0; /* antithetic :53-54 */
        for (int d = 0; d < D; ++d) {
          e[d] = sgn * eps[((size_t)j * Mh + b) * D + d];
          x[d] = e[d] * lambda[d] * sigma[j] + mu[d + (size_t)D * j]; /* :55 */
        }
        double q = 0.0;
        for (int k = 0; k < K; ++k) { /* :60-65 */
          double d2 = 0.0;
          for (int d = 0; d < D; ++d) {
            double t = (x[d] - mu[d + (size_t)D * k]) / sl[d + (size_t)D * k];
            d2 += t * t;
          }
          nrm[k] = nfk[k] * exp(-0.5 * d2);
          q += w[k] * nrm[k];
        }
        Hj_p += log(q);
        if (grad) {
          double isum = 0.0;
          for (int d = 0; d < D; ++d) { /* :77-79 */
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += (x[d] - mu[d + (size_t)D * k]) / sl2[d + (size_t)D * k] * nrm[k] * w[k];
            muj_p[d] += acc / q;                         /* :82 */
            isum += acc * e[d] * lambda[d];              /* :87 */
            lamj_p[d] += acc * e[d] / q;                 /* :93 (w_j sigma_j applied below) */
          }
          sgj_p += isum / q;
          wjlog_p += log(q);                             /* :97 */
          for (int l = 0; l < K; ++l) wl_p[l] += nrm[l] / q; /* :100 */
        }
      }
#ifdef _OPENMP
#pragma omp critical
#endif
      {
        Hj += Hj_p; sgj += sgj_p; wjlog += wjlog_p;
        for (int d = 0; d < D; ++d) { muj[d] += muj_p[d]; lamj[d] += lamj_p[d]; }
        for (int l = 0; l < K; ++l) wl[l] += wl_p[l];
      }
#ifdef _OPENMP
#pragma omp barrier
#pragma omp single
#endif
      {
        H -= w[j] * Hj / Ns; /* :67 */
        if (grad) {
          for (int d = 0; d < D; ++d) {
            mu_g[d + (size_t)D * j] = w[j] * muj[d] / Ns;
            lam_g[d] += w[j] * sigma[j] * lamj[d] / Ns;
          }
          sg_g[j] = w[j] * sgj / Ns;
          w_g[j] -= wjlog / Ns;
          for (int l = 0; l < K; ++l) w_g[l] -= w[j] * wl[l] / Ns;
        }
        Hj = 0.0; sgj = 0.0; wjlog = 0.0;
        for (int d = 0; d < D; ++d) { muj[d] = 0.0; lamj[d] = 0.0; }
        for (int l = 0; l < K; ++l) wl[l] = 0.0;
      } /* implicit barrier of the single: the next component starts from zeroed accumulators */
    }
    free(x); free(e); free(nrm); free(muj_p); free(lamj_p); free(wl_p);
  }
  *H_out = H;
  if (grad) {
    for (int i = 0; i < D * K; ++i) dH[i] = mu_g[i];
    for (int k = 0; k < K; ++k) dH[D * K + k] = sg_g[k] * sigma[k];          /* :113 */
    for (int d = 0; d < D; ++d) dH[D * K + K + d] = lam_g[d] * lambda[d];    /* :107 */
    softmax_jac_apply(K, eta, w_g, dH + D * K + K + D);
  }
  free(mu_g); free(sg_g); free(lam_g); free(w_g); free(sl); free(sl2); free(nfk); free(muj); free(lamj); free(wl);
}

/* misc/gplogjoint.m value + gradient, negquad/const/zero mean (ids 4/1/0), no variance, averaged over S.
   hyp is Nhyp x S: [log ell(D); log sf; noise(Nnoise); m0; xm(D); log omega(D)] */
void oracle_gplogjoint(int D, int K, int N, int S, int Nhyp, int Nnoise, int meanfun, const double* mu, const double* sigma,
                       const double* lambda, const double* w, const double* eta, const double* X, const double* hyp,
                       const double* alpha, int grad, double* F_out, double* dF) {
  const int T = D * K + K + D + K;
  double Fsum = 0.0;
  double* acc = (double*)calloc(T, sizeof(double));
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    double* accp = (double*)calloc(T, sizeof(double));
    double* wg = (double*)calloc(K, sizeof(double));
    double* tmp = (double*)calloc(K, sizeof(double));
    double* tau = (double*)malloc(sizeof(double) * D);
    double* dl = (double*)malloc(sizeof(double) * D);
    double* dmu = (double*)malloc(sizeof(double) * D);
    double* dlam = (double*)malloc(sizeof(double) * D);
    double Fp = 0.0;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int s = 0; s < S; ++s) { /* :98 */
      const double* h = hyp + (size_t)s * Nhyp;
      const int mo = D + 1 + Nnoise;
      double sum_lnell = 0.0;
      for (int d = 0; d < D; ++d) sum_lnell += h[d];
      const double ln_sf2 = 2.0 * h[D];
      const double m0 = meanfun > 0 ? h[mo] : 0.0;
      const double* al = alpha + (size_t)s * N;
      double Fs = 0.0;
      for (int k = 0; k < K; ++k) { /* :162 */
        double sumlogtau = 0.0;
        for (int d = 0; d < D; ++d) {
          double ell = exp(h[d]);
          tau[d] = sqrt(sigma[k] * sigma[k] * lambda[d] * lambda[d] + ell * ell);
          sumlogtau += log(tau[d]);
          dmu[d] = 0.0; dlam[d] = 0.0;
        }
        const double lnnf = ln_sf2 + sum_lnell - sumlogtau;
        double I = 0.0, dsig = 0.0;
        for (int n = 0; n < N; ++n) {
          double a2 = 0.0;
          for (int d = 0; d < D; ++d) {
            dl[d] = (mu[d + (size_t)D * k] - X[n + (size_t)N * d]) / tau[d];
            a2 += dl[d] * dl[d];
          }
          double za = exp(lnnf - 0.5 * a2) * al[n];
          I += za;
          if (grad) {
            double ssum = 0.0;
            for (int d = 0; d < D; ++d) {
              double q = dl[d] * dl[d] - 1.0;
              dmu[d] += -dl[d] / tau[d] * za;
              ssum += (lambda[d] / tau[d]) * (lambda[d] / tau[d]) * q;
              dlam[d] += (sigma[k] / tau[d]) * (sigma[k] / tau[d]) * q * lambda[d] * za;
            }
            dsig += ssum * sigma[k] * za;
          }
        }
        I += m0;
        double sl2 = 0.0;
        if (meanfun == 4) {
          double nu = 0.0;
          for (int d = 0; d < D; ++d) {
            double xm = h[mo + 1 + d], om = exp(h[mo + D + 1 + d]);
            double m = mu[d + (size_t)D * k];
            nu += (m * m + sigma[k] * sigma[k] * lambda[d] * lambda[d] - 2.0 * m * xm + xm * xm) / (om * om);
            sl2 += lambda[d] * lambda[d] / (om * om);
          }
          I += -0.5 * nu;
        }
        Fs += w[k] * I;
        if (grad) {
          for (int d = 0; d < D; ++d) {
            double g = w[k] * dmu[d], gl = w[k] * dlam[d];
            if (meanfun == 4) {
              double xm = h[mo + 1 + d], om = exp(h[mo + D + 1 + d]);
              g -= w[k] / (om * om) * (mu[d + (size_t)D * k] - xm);
              gl -= w[k] * sigma[k] * sigma[k] / (om * om) * lambda[d];
            }
            accp[d + D * k] += g;
            accp[D * K + K + d] += gl * lambda[d];                               /* Jacobian :362 */
          }
          accp[D * K + k] += (w[k] * dsig - w[k] * sigma[k] * sl2) * sigma[k];   /* :229-231, :356 */
          wg[k] = I;
        }
      }
      Fp += Fs;
      if (grad) {
        softmax_jac_apply(K, eta, wg, tmp);
        for (int k = 0; k < K; ++k) accp[D * K + K + D + k] += tmp[k];
      }
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      Fsum += Fp;
      for (int i = 0; i < T; ++i) acc[i] += accp[i];
    }
    free(accp); free(wg); free(tmp); free(tau); free(dl); free(dmu); free(dlam);
  }
  *F_out = Fsum / S;
  if (grad) for (int i = 0; i < T; ++i) dF[i] = acc[i] / S;
  free(acc);
}

/* misc/negelcbo_vbmc.m:33-48,97,106,116-117 for all four groups optimised, beta = 0, no bounds */
void oracle_negelcbo(int D, int K, int N, int S, int Nhyp, int Nnoise, int meanfun, int Mh, const double* theta,
                     const double* X, const double* hyp, const double* alpha, const double* eps, int grad, double* F, double* dF,
                     double* G_out, double* H_out) {
  const int T = D * K + K + D + K;
  double* sigma = (double*)malloc(sizeof(double) * K);
  double* lambda = (double*)malloc(sizeof(double) * D);
  double* w = (double*)malloc(sizeof(double) * K);
  double* dG = (double*)calloc(T, sizeof(double));
  double* dH = (double*)calloc(T, sizeof(double));
  const double* mu = theta;
  const double* eta = theta + D * K + K + D;
  double ssum = 0.0;
  for (int k = 0; k < K; ++k) sigma[k] = exp(theta[D * K + k]);
  for (int d = 0; d < D; ++d) lambda[d] = exp(theta[D * K + K + d]);
  for (int k = 0; k < K; ++k) { w[k] = exp(eta[k]); ssum += w[k]; }
  for (int k = 0; k < K; ++k) w[k] /= ssum;
  double G, H;
  oracle_gplogjoint(D, K, N, S, Nhyp, Nnoise, meanfun, mu, sigma, lambda, w, eta, X, hyp, alpha, grad, &G, dG);
  oracle_entmc(D, K, Mh, mu, sigma, lambda, w, eta, eps, grad, &H, dH);
  *F = -G - H;
  if (grad) for (int i = 0; i < T; ++i) dF[i] = -dG[i] - dH[i];
  if (G_out) *G_out = G;
  if (H_out) *H_out = H;
  free(sigma); free(lambda); free(w); free(dG); free(dH);
}
