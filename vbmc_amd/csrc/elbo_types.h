// Plain-data layouts shared by every kernel translation unit (no kernels in here).
#pragma once
#include <hip/hip_runtime.h>

#define WAVE 64

// ------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------
// vpd block per restart (doubles): mu[D*K] sigma[K] lambda[D] w[K] eta[K] lognf lnsigma[K] lnlambda[D]
struct VpLayout {
  int D, K;
  __host__ __device__ int mu() const { return 0; }
  __host__ __device__ int sigma() const { return D * K; }
  __host__ __device__ int lambda() const { return D * K + K; }
  __host__ __device__ int w() const { return D * K + K + D; }
  __host__ __device__ int eta() const { return D * K + 2 * K + D; }
  __host__ __device__ int lognf() const { return D * K + 3 * K + D; }
  __host__ __device__ int lnsigma() const { return D * K + 3 * K + D + 1; }
  __host__ __device__ int lnlambda() const { return D * K + 4 * K + D + 1; }
  __host__ __device__ int stride() const { return D * K + 4 * K + 2 * D + 2; }
};
// packed per-component entropy parameters: [m_dk = mu_dk/lambda_d (D), h_k = -1/(2 sigma_k^2),
// cK_k = -D log sigma_k, w_k, wi_k = w_k / sigma_k^2]
#define ENTP_EXTRA 4
// per-hyper-sample GP constants: ell2[D] xm[D] iom2[D] lnsf2_plus_sumlnell m0
#define GPC_STRIDE(D) (3 * (D) + 2)
// output block per restart: F G H varG varGss | dF[T] dG[T] dH[T]
#define OUT_HDR 5

struct ElboDims {
  int D, K, R, S, N, T;
  int opt[4];
  int off_mu, off_sigma, off_lambda, off_eta;  // offsets into theta (or -1)
};


#define LJ_CO_SPLIT 8   // most splits of the training set per cell group in the log-joint role (one single-wave workgroup each; buffer sizing)
// The log-joint role of the MFMA entropy kernel's launch (logjoint_body.h: lj_co_role; entropy_mfma.h, CO = true)
struct LjCo {
  int rows;              // grid rows (blockIdx.y) taken by the role, ahead of the entropy kernel's K; 0: no role
  int nwg;               // role workgroups per restart = G4 * S * nsplit
  int nsplit, want_grad;
  ElboDims dm;
  const double *X, *alpha, *gpc, *delta2;
  double* lj;
};

// k_entropy / k_entropy_mfma arguments.
// partial layout PE[r][j][c][NCOL]: sum log q | G[D] | SG | LG[D] | W[K]   (NCOL = 1 if !GRAD); c = blockIdx.x is the
// slot in the output record (C slots per (r, j)), c0 + blockIdx.x the chunk of samples it covers
struct EntArgs {
  const double* entp;    // R x K x (D+4)
  const double* vpd;     // R x VpLayout
  const double* eps;     // D x Mh x K (x R) or null -> Philox
  long long eps_stride_r;
  double* part;
  int D, K, Mh, C, tiles_per_chunk, ncol;
  int c0;                // first chunk of this launch (blockIdx.x + c0 is the chunk index; 0 unless the chunks are sharded over ranks)
  int prio;              // 1: a wave lowers its issue priority as it progresses (entropy_mfma.h: the one behind on a SIMD is served first)
  unsigned long long seed;
  int r0, rstride;       // the device-RNG stream of restart r is keyed by r0 + r * rstride: the restart's index in the WHOLE batch when
                         // the batch is dealt over several devices (vbmc_elbo_batch_multi: r0 = g, rstride = G); 0, 1 otherwise
  double cutoff;         // > 0: skip k-tiles whose terms are provably < exp(-cutoff) relative to q (block-sparse mode)
  int nc_launch;         // k_entropy_lane: chunk slots this launch covers (its items are (j, slot) pairs, four per workgroup)
  int co_c1, co_c2, co_tpc2;   // k_entropy_mfma: TWO chunk classes per (component, restart) -- slots 0 .. co_c1 - 1 hold tiles_per_chunk tiles each, slots
                         // co_c1 .. co_c1 + co_c2 - 1 co_tpc2 (fewer) tiles each; in a launch that carries the log-joint role the role's workgroups
                         // take the second class AFTER their role (they enter the tile loop a role later and leave it with the others).  co_c2 = 0: one class
  int walk_tpw, walk_R;  // k_entropy_mfma, single-wave workgroups: > 0: the WALK -- grid (waves, 1, 1), wave w owns tiles [w tpw, (w + 1) tpw) of the
                         // sequence of all walk_R x K (restart, component) pairs' tiles; C = record slots per pair (ent_walk_slots)
  LjCo lj;               // CO kernels only: the expected log joint as extra workgroups of this launch (lj.rows = 0: none)
};

// The walk's record slots (entropy_mfma.h): pair p holds tiles [p ntile, (p + 1) ntile) of the sequence, wave w owns [w tpw, (w + 1) tpw): the
// pair's records come from waves first .. last, in that order, in slots 0 .. last - first of its C slots.
__host__ __device__ inline int ent_walk_first(long long p, int ntile, int tpw) { return (int)((p * ntile) / tpw); }
__host__ __device__ inline int ent_walk_slots(long long p, int ntile, int tpw) {
  return (int)((p * ntile + ntile - 1) / tpw) - ent_walk_first(p, ntile, tpw) + 1;
}
__host__ __device__ inline int ent_walk_max_slots(int ntile, int tpw) { return (ntile - 1) / tpw + 2; }

// The short kernels of a pass (copies, k_prep, the log joint, the reductions, the finalize kernel) ask for the highest issue priority: in
// the pipelined step they share SIMDs with the other pass's entropy kernel, whose waves run at priorities 3 -> 0 (entropy_mfma.h), and
// they are what the next pass waits for.
#define VB_SMALL_PRIO() __builtin_amdgcn_s_setprio(3)

