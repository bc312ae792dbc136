// gplite GP surrogate on gfx950: SE-ARD kernel matrix (sq_dist), jittered Cholesky posterior,
// predictive mean / variance.
//
// Reference: utils/sq_dist.m:14-50, gplite/private/gplite_core.m:33-102,278-291,
// gplite/gplite_post.m:167-251, gplite/gplite_pred.m:52-165, gplite/gplite_meanfun.m:400-436,
// gplite/gplite_noisefun.m:176-210.
//
//   k_sq_dist_mfma  C = max(|a|^2 + |b|^2 - 2 a'b, 0), the a'b contraction on v_mfma_f64_16x16x4_f64
//   k_gp_build      A_s = K/(sn2div*mult) + diag(sn2/sn2div)   (Lchol)   or  K + mult*diag(sn2)
//   k_chol2         (chol_mfma.h) in-place blocked (16) right-looking Cholesky, upper factor, one WG per sample,
//                   reports MATLAB's p (> 0 = not positive definite) for the jitter retry
//   k_gp_scale      scaled, centred inputs and row norms per hyper-sample; r = y - m(X)
//   k_pred_prep     ell-scaled, sq_dist-centred training inputs per hyper-sample
//   k_gp_pred       four waves per 16 test points: cross-kernel slab -> fmu = m* + Ks'alpha, V = inv(L')*(sW.*Ks) as
//                   MFMA products against the precomputed triangular inverse, fs2 = kss - |V|^2
//   k_pred_avg      hyper-sample averaging with between-sample variance (gplite_pred.m:154-165)
#pragma once
#include "common.h"
#include "device_math.h"
#include "var_kernels.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// sq_dist on MFMA.  a: D x n, b: D x m (column-major, already mean-centred by the caller kernel
// through `mu`), C: n x m.  One wave computes a 16(i) x 16(j) tile as the TRANSPOSED product
// (rows = j from b, cols = i from a) so that the four accumulator registers of a lane are four
// consecutive-j rows of ONE column i... stores of a fixed register are 16 consecutive i (128 B).
// A operand: lane l holds b[d = 4q + (l>>4)][j0 + (l&15)];  B operand: a[d = 4q + (l>>4)][i0 + (l&15)].
// C/D layout of v_mfma_f64_16x16x4_f64: col = l&15, row = (l>>4) + 4*reg.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sq_dist_mfma(int D, int n, int m, const double* __restrict__ a,
                                                      const double* __restrict__ b, const double* __restrict__ mu,
                                                      double* __restrict__ C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ti = blockIdx.x * 4 + wv;     // tile along i (a)
  const int tj = blockIdx.y;              // tile along j (b)
  const int i0 = ti * 16, j0 = tj * 16;
  if (i0 >= n) return;
  const int li = lane & 15, lq = lane >> 4;
  const int ia = i0 + li, jb = j0 + li;
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  double aa = 0.0, bb = 0.0;  // partial |a_i|^2 (for i = ia) and |b_j|^2 (j = jb) over this lane's d's
  for (int q = 0; q < (D + 3) / 4; ++q) {
    const int d = 4 * q + lq;
    double av = 0.0, bv = 0.0;
    if (d < D) {
      if (ia < n) av = a[d + (size_t)D * ia] - mu[d];
      if (jb < m) bv = b[d + (size_t)D * jb] - mu[d];
    }
    aa = fma(av, av, aa);
    bb = fma(bv, bv, bb);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc, 0, 0, 0);
  }
  // complete the squared norms across the 4 lanes sharing (l&15)
  aa += __shfl_xor(aa, 16, 64); aa += __shfl_xor(aa, 32, 64);
  bb += __shfl_xor(bb, 16, 64); bb += __shfl_xor(bb, 32, 64);
  // lane holds column i = ia, rows j = j0 + lq + 4*reg ; needs bb of those j (held by lanes with l&15 == lq+4reg)
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int jr = lq + 4 * reg;
    const double bbj = __shfl(bb, jr, 64);
    const int j = j0 + jr;
    if (ia < n && j < m) {
      double c = aa + (bbj - 2.0 * acc[reg]);   // sq_dist.m:45  sum(a.*a)' + (sum(b.*b) - 2*a'*b)
      C[ia + (size_t)n * j] = fmax(c, 0.0);     // :49
    }
  }
}

// ------------------------------------------------------------------------------------------
// device mean function (ids 0, 1, 4) at a point x (strided access), hyp_mean points at the block
// ------------------------------------------------------------------------------------------
__device__ inline double gp_meanfun(int meanfun, int D, const double* hm, const double* x, size_t stride) {
  if (meanfun == 0) return 0.0;
  if (meanfun == 1) return hm[0];
  double z2 = 0.0;
  for (int d = 0; d < D; ++d) {
    double t = (x[d * stride] - hm[1 + d]) / exp(hm[D + 1 + d]);
    z2 = fma(t, t, z2);
  }
  return hm[0] - 0.5 * z2;  // gplite_meanfun.m:425-431
}

// A_s for the Cholesky.  Xc[s] holds the scaled, mean-centred inputs a = (X' ./ ell) - mean  (D x N),
// aa[s][n] = |a_n|^2.  K = sf2 * exp(-max(aa_i + aa_j - 2 a_i.a_j, 0)/2)   (gplite_core.m:52-56)
// rout != null: the residual r = y - m(X) of the same hyper-sample as well (gplite_core.m:58-65; round 4 had a kernel of its own for it).
__global__ void __launch_bounds__(256) k_gp_scale(int N, int D, int Nhyp, const double* __restrict__ X,
                                                  const double* __restrict__ hyp, double* __restrict__ Xc,
                                                  double* __restrict__ aa, int moff, int meanfun, const double* __restrict__ y,
                                                  double* __restrict__ rout) {
  const int s = blockIdx.y;
  const double* h = hyp + (size_t)s * Nhyp;
  __shared__ double mean[32], sell[32], som[32];
  const int tid = threadIdx.x;
  if (rout && meanfun == 4 && tid < D) som[tid] = exp(h[moff + D + 1 + tid]);
  {
    // mean over n of X(n,d)/ell_d  (mean(a,2), sq_dist.m:36): one wave per dimension at a time, lanes along n with the loads
    // of eight rows in flight, fixed-order butterfly.  (Round 5: the first version walked N / 8 dependent load + divide steps
    // per lane and called exp(h[d]) for every element of the scaled copy -- 18 us of a 0.55 ms gplite_nlZ call.)
    const int wave = tid >> 6, lane = tid & 63;
    // wave w takes the dimensions w, w + 4, ...: all of them at once, eight rows per lane and dimension in flight (the ragged end
    // masked: adds + 0), so that the whole reduction is one round trip to memory -- the inputs have just been uploaded
    double ell[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int d = wave + 4 * j; ell[j] = d < D ? exp(h[d]) : 1.0; acc[j] = 0.0; }
    for (int n = lane; n < N; n += 8 * 64) {
      double v[8][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = wave + 4 * j;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[j][u] = (d < D && n + 64 * u < N) ? X[(size_t)N * d + n + 64 * u] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[j] += v[j][u] / ell[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = wave + 4 * j;
      if (d < D) {                       // wave-uniform
        const double t = wave_sum(acc[j]);
        if (lane == 0) { mean[d] = t / N; sell[d] = ell[j]; }
      }
    }
  }
  __syncthreads();
  for (int n = blockIdx.x * blockDim.x + tid; n < N; n += gridDim.x * blockDim.x) {
    double acc = 0.0;
    double xv[32];                                        // the point's coordinates: every load in flight at once
#pragma unroll
    for (int d = 0; d < 32; ++d) xv[d] = d < D ? X[n + (size_t)N * d] : 0.0;
#pragma unroll
    for (int d = 0; d < 32; ++d)
      if (d < D) {
        double v = xv[d] / sell[d] - mean[d];
        Xc[((size_t)s * N + n) * D + d] = v;
        acc = fma(v, v, acc);
      }
    aa[(size_t)s * N + n] = acc;
    if (rout) {
      const double* hm = h + moff;
      double m = 0.0;
      if (meanfun == 1) m = hm[0];
      else if (meanfun == 4) {
        double z2 = 0.0;
#pragma unroll
        for (int d = 0; d < 32; ++d)
          if (d < D) {
            const double t = (xv[d] - hm[1 + d]) / som[d];
            z2 = fma(t, t, z2);
          }
        m = hm[0] - 0.5 * z2;            // gplite_meanfun.m:425-431
      }
      rout[(size_t)s * N + n] = y[n] - m;
    }
  }
}

// One workgroup per 64 x 64 tile of the UPPER triangle (k_chol2 reads i <= j only; it zeroes the strict lower part
// itself): both 64-point slabs of the scaled inputs are staged in LDS dimension-major, a lane keeps its own point in
// registers and walks 16 columns whose coordinates are wave-uniform LDS broadcasts; exponent by the table exp.
#define GPB_T 64
template <int DT>   // D padded to DT with zero coordinates: the dimension loops are branch-free
__global__ void __launch_bounds__(256) k_gp_build(int N, int D, int Nhyp, const double* __restrict__ hyp,
                                                  const double* __restrict__ Xc, const double* __restrict__ aa,
                                                  const double* __restrict__ sn2,      // S x N  noise variance per point
                                                  const double* __restrict__ scal,     // S x 4: sn2div, mult, lchol, _
                                                  const unsigned char* __restrict__ active, double* __restrict__ A) {
  const int s = blockIdx.z;
  if (!active[s]) return;
  const int i0 = blockIdx.x * GPB_T, j0 = blockIdx.y * GPB_T;
  if (i0 > j0) return;                               // tile entirely below the diagonal
  __shared__ double TAB[VB_EXP_TAB_N];
  __shared__ double XI[DT * GPB_T], XJ[DT * GPB_T];  // [d][point]
  __shared__ double AJ[GPB_T];
  const int tid = threadIdx.x, ti = tid & 63, tj = tid >> 6;
  const double* h = hyp + (size_t)s * Nhyp;
  const double sf2 = exp(2.0 * h[D]);
  const double sn2div = scal[s * 4 + 0], mult = scal[s * 4 + 1];
  const bool lchol = scal[s * 4 + 2] != 0.0;
  const double* xs = Xc + (size_t)s * N * D;
  const double* as = aa + (size_t)s * N;
  double* As = A + (size_t)s * N * N;
  TAB[tid] = c_exp2_tab[tid];
  // the rows i0 .. i0+63 of the point-major Xc are one contiguous block: coalesced reads, transposed into LDS
  for (int e = tid; e < GPB_T * (DT - D); e += 256) { XI[D * GPB_T + e] = 0.0; XJ[D * GPB_T + e] = 0.0; }   // padded dimensions
  for (int e = tid; e < GPB_T * D; e += 256) {
    const int pnt = e / D, d = e - pnt * D;
    XI[d * GPB_T + pnt] = (i0 + pnt < N) ? xs[(size_t)i0 * D + e] : 0.0;
    XJ[d * GPB_T + pnt] = (j0 + pnt < N) ? xs[(size_t)j0 * D + e] : 0.0;
  }
  if (tid < GPB_T) AJ[tid] = (j0 + tid < N) ? as[j0 + tid] : 0.0;
  __syncthreads();
  const int i = i0 + ti;
  if (i >= N) return;
  double xi[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) xi[d] = XI[d * GPB_T + ti];
  const double ai = as[i];
  const double sdiv = sn2div * mult;
  for (int t = 0; t < 16; ++t) {
    const int jl = tj * 16 + t, j = j0 + jl;
    if (j >= N) break;
    double dot = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) dot = fma(xi[d], XJ[d * GPB_T + jl], dot);
    const double c = fmax(ai + (AJ[jl] - 2.0 * dot), 0.0);
    const double k = sf2 * vb_exp_tab<0>(-0.5 * c, TAB);
    double v;
    if (lchol) v = k / sdiv + (i == j ? sn2[(size_t)s * N + i] / sn2div : 0.0);  // :78
    else v = k + (i == j ? mult * sn2[(size_t)s * N + i] : 0.0);               // :92
    As[(size_t)j * N + i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, upper factor R (R'R = A) in place, strict lower part zeroed: k_chol2 in chol_mfma.h.
// pfail[s] = 0 on success, j+1 when the j-th pivot is not positive (MATLAB's [R,p] = chol(A)).
// ------------------------------------------------------------------------------------------
#ifdef CHOL_TS   // tools/chol_bench.hip only: per-step phase stamps of matrix 0 (wall_clock64, 100 MHz)
__device__ long long g_chol_ts[4 * 512];
__device__ long long g_chol_clk[2 * 512];   // shader-clock readings next to slot 0 / slot 3: cycles per microsecond = the clock the CU really ran at
#define CHOL_STAMP(slot, step) do { if (blockIdx.x == 0 && (step) < 512) { g_chol_ts[4 * (step) + (slot)] = wall_clock64(); if ((slot) == 0) g_chol_clk[2 * (step)] = clock64(); if ((slot) == 3) g_chol_clk[2 * (step) + 1] = clock64(); } } while (0)
#else
#define CHOL_STAMP(slot, step) do { } while (0)
#endif

__device__ __forceinline__ double chol_readlane(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

#include "chol_mfma.h"

// alpha-type solve x = R \ (R' \ z) for ONE right-hand side per matrix, latency-shaped: one 1024-thread workgroup per
// matrix, the vector in LDS.  Per 16-row block step the 16 dot products with the already solved part are one wave each
// (lanes along the long dimension, fixed-order butterfly), then wave 0 applies the precomputed inverse of the diagonal
// block (k_diag_inv).  ~2 us per step instead of ~5 us for the 16-column MFMA slab solve with 15 zero columns.
#define ASOLVE_THREADS 1024
// skip_fwd != 0: Zin already holds R' \ z (the factorisation solved it on the way, chol_mfma.h): backward half only.
__global__ void __launch_bounds__(ASOLVE_THREADS) k_alpha_solve(int N, const double* __restrict__ Lall, const double* __restrict__ Finv,
                                                                const unsigned char* __restrict__ on, const double* __restrict__ Zin,
                                                                double* __restrict__ Xo, int skip_fwd, const double* __restrict__ scal) {
  extern __shared__ double lds[];   // Np + 16
  const int s = blockIdx.x;
  if (!on[s]) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nblk = (N + 15) >> 4, Np = nblk << 4;
  double* v = lds;
  double* tmp = v + Np;
  const double* R = Lall + (size_t)s * N * N;
  const double* Fi = Finv + (size_t)s * nblk * 256;
  for (int i = tid; i < Np; i += ASOLVE_THREADS) v[i] = i < N ? Zin[(size_t)s * N + i] : 0.0;
  __syncthreads();
  // ---- forward: R' v = z
  for (int bi = skip_fwd ? nblk : 0; bi < nblk; ++bi) {
    const int b0 = bi << 4, c = b0 + wave;
    double dot = 0.0;
    if (c < N)
      for (int row = lane; row < b0; row += 64) dot = fma(R[(size_t)c * N + row], v[row], dot);
    dot = wave_sum(dot);
    if (lane == 0) tmp[wave] = v[c < Np ? c : 0] - dot;
    __syncthreads();
    if (tid < 16) {
      double a = 0.0;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) a = fma(Fi[(size_t)bi * 256 + tid * 16 + cc], tmp[cc], a);
      v[b0 + tid] = (b0 + tid < N) ? a : 0.0;
    }
    __syncthreads();
  }
  // ---- backward: R x = v
  for (int bi = nblk - 1; bi >= 0; --bi) {
    const int b0 = bi << 4, e0 = b0 + 16, i = b0 + wave;
    double dot = 0.0;
    if (i < N)
      for (int j = e0 + lane; j < N; j += 64) dot = fma(R[(size_t)j * N + i], v[j], dot);
    dot = wave_sum(dot);
    if (lane == 0) tmp[wave] = v[i < Np ? i : 0] - dot;
    __syncthreads();
    if (tid < 16) {
      double a = 0.0;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) a = fma(Fi[(size_t)bi * 256 + cc * 16 + tid], tmp[cc], a);   // (R_bb^{-1})[i][c] = Finv[c][i]
      v[b0 + tid] = (b0 + tid < N) ? a : 0.0;
    }
    __syncthreads();
  }
  const double sl = scal ? scal[s * 4 + 3] : 1.0;     // alpha = x / sl (gplite_core.m:102); no divisor, no division
  for (int i = tid; i < N; i += ASOLVE_THREADS) Xo[(size_t)s * N + i] = scal ? v[i] / sl : v[i];
}

// The same solve for N <= ASOLVE1_THREADS, right-looking: thread e owns element e of the vector in a register.  Per 16-row block step the
// 16 owner lanes (one wave) exchange their entries through LDS and apply the inverse of the diagonal block -- no workgroup
// barrier inside that --, publish x_b, and after ONE barrier every thread behind (forward) / ahead of (backward) the block
// subtracts its 16-term product with x_b.  The 16 matrix entries a thread needs for the NEXT step (and the owner lanes' row
// of the block inverse) are fetched before the barrier: no L2 round trip on the critical path, no 64-lane reduction.
// (k_alpha_solve above stays as the path for larger N; 512 threads: the 32 prefetched values need the 256-VGPR budget.)
#define ASOLVE1_THREADS 512
__device__ __forceinline__ void alpha_solve1_body(int N, int s, const double* __restrict__ Lall, const double* __restrict__ Finv,
                                                  const double* __restrict__ Zin, double* __restrict__ Xo, int skip_fwd,
                                                  const double* __restrict__ scal) {
  __shared__ double xb[2][16], tb[16];
  const int e = threadIdx.x, lane = e & 63, k = e & 15, myblk = e >> 4;
  const int nblk = (N + 15) >> 4;
  const double* R = Lall + (size_t)s * N * N;
  const double* Fi = Finv + (size_t)s * nblk * 256;
  double z = e < N ? Zin[(size_t)s * N + e] : 0.0;
  double pre[16], fi[16];
  // ---- forward: R' v = z.  Step b needs R[b0 + r][e] (column e, 16 contiguous rows) for e >= b0 + 16.
  auto fetch_fwd = [&](int b) {
    const int b0 = b << 4;
    const bool mine = e >= b0 + 16 && e < N && b < nblk;
    // 16 contiguous rows of column e: 16-byte loads (a lane per column means every load instruction touches 64 cache lines;
    // half as many instructions as with 8-byte loads).  The last block of a ragged N goes element by element.
    typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
    if (b0 + 16 <= N) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        d2u v2 = {0.0, 0.0};
        if (mine) v2 = *reinterpret_cast<const d2u*>(R + (size_t)e * N + b0 + r);
        pre[r] = v2[0]; pre[r + 1] = v2[1];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) pre[r] = (mine && b0 + r < N) ? R[(size_t)e * N + b0 + r] : 0.0;
    }
    if (myblk == b + 1 && b + 1 < nblk) {
#pragma unroll
      for (int c = 0; c < 16; ++c) fi[c] = Fi[(size_t)(b + 1) * 256 + k * 16 + c];      // row k of (R_bb')^{-1}
    }
  };
  if (!skip_fwd) {
  if (myblk == 0) {
#pragma unroll
    for (int c = 0; c < 16; ++c) fi[c] = Fi[k * 16 + c];
  }
  fetch_fwd(0);
  }
  for (int b = skip_fwd ? nblk : 0; b < nblk; ++b) {
    if (myblk == b) {                     // 16 lanes of one wave
      tb[k] = z;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double a4[4] = {0.0, 0.0, 0.0, 0.0};          // four short chains instead of one of 16 dependent FMAs
#pragma unroll
      for (int c = 0; c < 16; ++c) a4[c & 3] = fma(fi[c], tb[c], a4[c & 3]);
      const double a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
      z = e < N ? a : 0.0;
      xb[b & 1][k] = z;
    }
    // LDS-only barrier: __syncthreads() would also wait for the global loads just issued for the next step (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    double p4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; ++r) p4[r & 3] = fma(pre[r], xb[b & 1][r], p4[r & 3]);   // zero for the threads at or before the block
    z -= (p4[0] + p4[1]) + (p4[2] + p4[3]);
    fetch_fwd(b + 1);
  }
  __syncthreads();
  // ---- backward: R x = v.  Step b needs R[e][b0 + c] (row e, 16 columns) for e < b0 -- fetched FOUR steps ahead into a ring
  // (round 5: a step is ~0.25 us of exchange and two 16-term products, an L2 round trip ~1 us; with one step of look-ahead every
  // step waited for its row, 1.2 us per step)
  double prq[4][16];
  auto fetch_bwd = [&](int b, double (&pr)[16]) {
    const int b0 = b << 4;
    const bool mine = b >= 0 && e < b0;
#pragma unroll
    for (int c = 0; c < 16; ++c) pr[c] = (mine && b0 + c < N) ? R[(size_t)(b0 + c) * N + e] : 0.0;
    if (b >= 1 && myblk == b - 1) {
#pragma unroll
      for (int c = 0; c < 16; ++c) fi[c] = Fi[(size_t)(b - 1) * 256 + c * 16 + k];      // row k of R_bb^{-1} = column k of (R_bb')^{-1}
    }
  };
  if (myblk == nblk - 1) {
#pragma unroll
    for (int c = 0; c < 16; ++c) fi[c] = Fi[(size_t)(nblk - 1) * 256 + c * 16 + k];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) fetch_bwd(nblk - 1 - q, prq[q]);
  for (int bb = nblk - 1; bb >= 0; bb -= 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = bb - q;
      if (b >= 0) {                          // uniform
        if (myblk == b) {
          tb[k] = z;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          double a4[4] = {0.0, 0.0, 0.0, 0.0};          // four short chains instead of one of 16 dependent FMAs
#pragma unroll
          for (int c = 0; c < 16; ++c) a4[c & 3] = fma(fi[c], tb[c], a4[c & 3]);
          const double a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
          z = e < N ? a : 0.0;
          xb[b & 1][k] = z;
        }
        // LDS-only barrier: __syncthreads() would also wait for the global loads in flight for the next steps (vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        double p4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 16; ++c) p4[c & 3] = fma(prq[q][c], xb[b & 1][c], p4[c & 3]);
        z -= (p4[0] + p4[1]) + (p4[2] + p4[3]);
        fetch_bwd(b - 4, prq[q]);
      }
    }
  }
  if (e < N) Xo[(size_t)s * N + e] = scal ? z / scal[s * 4 + 3] : z;
}
__global__ void __launch_bounds__(ASOLVE1_THREADS, 2) k_alpha_solve1(int N, const double* __restrict__ Lall, const double* __restrict__ Finv,
                                                                 const unsigned char* __restrict__ on, const double* __restrict__ Zin,
                                                                 double* __restrict__ Xo, int skip_fwd, const double* __restrict__ scal) {
  const int s = blockIdx.x;
  if (!on[s]) return;
  alpha_solve1_body(N, s, Lall, Finv, Zin, Xo, skip_fwd, scal);
}

// T' = inv(R')' for the marginal-likelihood gradient AND alpha's backward solve in ONE launch (N <= ASOLVE1_THREADS, few
// matrices): workgroups 0 .. nblk - 1 of a matrix invert the factor slab by slab (tri_inverse2_body), workgroup nblk runs the
// single-vector solve.  Neither needs the other; as two kernels on two streams they cost an event record, two stream waits
// and their idle gaps (~12 us of a 0.46 ms gplite_nlZ call) on top of the longer of the two.
static_assert(ASOLVE1_THREADS == 64 * TRI2_W, "the combined launch runs both bodies with one block size");
template <int MAXS>
__global__ void __launch_bounds__(64 * TRI2_W, 2) k_tri_inverse2_alpha(int N, const double* __restrict__ Lall, const double* __restrict__ Finv,
                                                                    const unsigned char* __restrict__ on, double* __restrict__ TT,
                                                                    const double* __restrict__ Zin, double* __restrict__ Xo,
                                                                    const double* __restrict__ scal) {
  const int s = blockIdx.y;
  if (!on[s]) return;
  if ((int)blockIdx.x == (int)gridDim.x - 1) alpha_solve1_body(N, s, Lall, Finv, Zin, Xo, 1, scal);
  else tri_inverse2_body<MAXS, (MAXS <= 4 ? 4 : 2)>(N, blockIdx.x, s, Lall, Finv, TT, 1);
}

// [mstar, vstar] = gplite_pred(gp, xstar, ystar, [], 1, 1) (gplite_post.m:189) from the solves the append needs anyway:
//   mstar = m(xstar) + Ks' alpha                                                         (gplite_pred.m:83)
//   vstar = max(fs2, 0) + sn2_eff,  fs2 = kss - |v|^2/sn2_eff (factor) or kss + Ks' x (stored inverse)   (:99-104,120-121)
// and the coefficient (mstar - ystar)/vstar of the alpha update; writes them over sc[s][2], sc[s][3].
__global__ void __launch_bounds__(256) k_rank1_stats(int N, int D, int Nhyp, int moff, int meanfun, const double* __restrict__ hyp,
                                                     const double* __restrict__ xstar, const double* __restrict__ alpha,
                                                     const unsigned char* __restrict__ lchol, const double* __restrict__ Ks,
                                                     const double* __restrict__ V, const double* __restrict__ Xs, double ystar,
                                                     double* __restrict__ sc) {
  __shared__ double red[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const bool ch = lchol[s] != 0;
  const double* ks = Ks + (size_t)s * N;
  double d1 = 0.0, d2 = 0.0;
  for (int n = tid; n < N; n += 256) {
    d1 = fma(ks[n], alpha[(size_t)s * N + n], d1);
    const double t = ch ? V[(size_t)s * N + n] : 0.0;
    d2 = ch ? fma(t, t, d2) : fma(ks[n], Xs[(size_t)s * N + n], d2);
  }
  d1 = block_sum(d1, red);
  d2 = block_sum(d2, red);
  if (tid == 0) {
    const double sn2 = sc[s * 4 + 0], kss = sc[s * 4 + 1];
    const double mstar = gp_meanfun(meanfun, D, hyp + (size_t)s * Nhyp + moff, xstar, 1) + d1;
    const double fs2 = ch ? kss - d2 / sn2 : kss + d2;
    const double vstar = fmax(fs2, 0.0) + sn2;
    sc[s * 4 + 2] = (mstar - ystar) / vstar;
    sc[s * 4 + 3] = vstar;
  }
}

// Rank-one append of one observation (gplite/gplite_post.m:226-245): column j of the new (N+1) x (N+1) matrix per workgroup.
//   factor samples:   Lnew = [L, v/sn2_eff; 0, sqrt(1 + Kss/sn2_eff - |v/sn2_eff|^2)]            (:228-232)
//   inverse samples:  Lnew = [L + vv*au', -vv; -vv', -1/vstar],  au = -x, vv = -au/vstar          (:234-236)
//   alpha_new = [alpha; 0] + (mstar - ystar)/vstar * [au; -1],   au = x/sn2_eff (factor) or -x     (:227,234,245)
// sc[s] = {sn2_eff, Kss, (mstar - ystar)/vstar, vstar}.
__global__ void __launch_bounds__(256) k_rank1_assemble(int N, const double* __restrict__ Lall, const double* __restrict__ alpha,
                                                        const unsigned char* __restrict__ lchol, const double* __restrict__ V,
                                                        const double* __restrict__ Xs, const double* __restrict__ sc,
                                                        double* __restrict__ Lnew, double* __restrict__ anew) {
  __shared__ double red[256];
  const int j = blockIdx.x, s = blockIdx.y, tid = threadIdx.x, N1 = N + 1;
  const double sn2 = sc[s * 4 + 0], Kss = sc[s * 4 + 1], coef = sc[s * 4 + 2], vstar = sc[s * 4 + 3];
  const bool ch = lchol[s] != 0;
  const double* L = Lall + (size_t)s * N * N;
  const double* v = V + (size_t)s * N;
  const double* x = Xs + (size_t)s * N;
  double* o = Lnew + (size_t)s * N1 * N1 + (size_t)j * N1;
  if (j < N) {
    const double xj = x[j];
    for (int i = tid; i < N; i += 256) o[i] = ch ? L[(size_t)j * N + i] : L[(size_t)j * N + i] - x[i] * xj / vstar;
    if (tid == 0) o[N] = ch ? 0.0 : -xj / vstar;
    // alpha, one entry per column block
    if (tid == 1) anew[(size_t)s * N1 + j] = alpha[(size_t)s * N + j] + coef * (ch ? xj / sn2 : -xj);
  } else {
    double part = 0.0;
    for (int i = tid; i < N; i += 256) {
      const double c = ch ? v[i] / sn2 : -x[i] / vstar;
      o[i] = c;
      part = fma(c, c, part);
    }
    const double cc = block_sum(part, red);
    if (tid == 0) {
      o[N] = ch ? sqrt(1.0 + Kss / sn2 - cc) : -1.0 / vstar;
      anew[(size_t)s * N1 + N] = -coef;
    }
  }
}

__global__ void k_negate_copy(size_t n, const double* __restrict__ src, double* __restrict__ dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = -src[i];
}

// ------------------------------------------------------------------------------------------
// Prediction: one wave = 16 test points x one hyper-sample.
// ------------------------------------------------------------------------------------------
struct PredArgs {
  int N, D, S, Nhyp, Nstar, meanfun, moff, noff, nf0, nf1, nf2;
  const double* X;       // N x D
  const double* Xs;      // Nstar x D (column-major)
  const double* s2s;     // Nstar or null
  const double* ys;      // Nstar or null: ystar, only for the output-dependent noise term
  const double* hyp;     // Nhyp x S
  const double* alpha;   // N x S
  const double* L;       // N x N x S
  const double* sn2_eff; // S  (1/sW^2)
  const double* sn2_mult;// S
  const unsigned char* lchol;
  const double* mean_a;  // D  column means of X
  const double* mean_b;  // D  column means of Xstar
  const double* finv;    // S x nblk x 256 inverse diagonal blocks (k_diag_inv)
  const double* tinv;    // S x N x N  inv(L') per Lchol sample, element (row, col) at col*N + row
  double* fmu;           // Nstar x S
  double* fs2;
  double* ys2;
};

// k_pred_prep: per hyper-sample the ell-scaled, sq_dist-centred training inputs and their squared norms:
// Xc[s][i][d] = X(i,d)/ell_d - mu_d,  aa[s][i] = |Xc_i|^2,  mu = (m/(n+m)) mean(b) + (n/(n+m)) mean(a)  (sq_dist.m:36)
__global__ void __launch_bounds__(256) k_pred_prep(PredArgs a, double* __restrict__ Xc, double* __restrict__ aa,
                                                   double* __restrict__ muv /* S x 2D: mu, 1/ell */) {
  const int s = blockIdx.y, N = a.N, D = a.D;
  const double* h = a.hyp + (size_t)s * a.Nhyp;
  __shared__ double mu[32], iell[32];
  if (threadIdx.x < D) {
    const int d = threadIdx.x;
    const double ie = 1.0 / exp(h[d]);   // diag(1./ell) * X'   (gplite_pred.m:73)
    const double n = (double)N, m = (double)a.Nstar;
    iell[d] = ie;
    mu[d] = (m / (n + m)) * (a.mean_b[d] * ie) + (n / (n + m)) * (a.mean_a[d] * ie);
    if (blockIdx.x == 0) { muv[(size_t)s * 2 * D + d] = mu[d]; muv[(size_t)s * 2 * D + D + d] = ie; }
  }
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int d = 0; d < D; ++d) {
      double v = a.X[i + (size_t)N * d] * iell[d] - mu[d];
      Xc[((size_t)s * N + i) * D + d] = v;
      acc = fma(v, v, acc);
    }
    aa[(size_t)s * N + i] = acc;
  }
}

// k_pred_ks: the sW-scaled cross-kernel matrix for every hyper-sample, KsW[s][i/16][n][i%16] = sW_s * k_s(X_n, Xstar_i)
// (tiled by 16 points), each element computed exactly once, and fmu's data term Ks' alpha (gplite_pred.m:74,83).
// One wave per (16 test points, hyper-sample).  The inner products of sq_dist's expansion |a|^2 + |b|^2 - 2 a.b
// (sq_dist.m:45) for a 16 x 16 block of (training point, test point) pairs are QS MFMAs (inner dimension D in steps
// of 4); the accumulator layout (row n = lg + 4 reg, column point = li) is exactly the coalesced store pattern.
template <int QS>
__global__ void __launch_bounds__(64) k_pred_ks(PredArgs a, const double* __restrict__ Xc, const double* __restrict__ aa,
                                                const double* __restrict__ muv, double* __restrict__ KsW, double* __restrict__ partF) {
  __shared__ double tab[VB_EXP_TAB_N];
  const int pt = blockIdx.x, s = blockIdx.y, lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int N = a.N, D = a.D;
  for (int t = lane; t < VB_EXP_TAB_N; t += 64) tab[t] = c_exp2_tab[t];
  const double* h = a.hyp + (size_t)s * a.Nhyp;
  const double* mu = muv + (size_t)s * 2 * D;
  const double* iell = mu + D;
  const double sf2 = exp(2.0 * h[D]);
  const double lsf2 = 2.0 * h[D];
  const double sW = a.lchol[s] ? 1.0 / sqrt(a.sn2_eff[s]) : 1.0;
  const int jc = pt * 16 + li;
  const bool cv = jc < a.Nstar;
  // B operand: this lane's test point li, dimensions 4q + lg; |b|^2 of the point by a 4-lane-group reduction
  double xb[QS];
  double bb = 0.0;
#pragma unroll
  for (int q = 0; q < QS; ++q) {
    const int d = 4 * q + lg;
    xb[q] = (d < D && cv) ? a.Xs[jc + (size_t)a.Nstar * d] * iell[d] - mu[d] : 0.0;
    bb = fma(xb[q], xb[q], bb);
  }
  bb += __shfl_xor(bb, 16, 64);
  bb += __shfl_xor(bb, 32, 64);
  __syncthreads();
  const double* al = a.alpha + (size_t)s * N;
  const double* xcs = Xc + (size_t)s * N * D;
  const double* aas = aa + (size_t)s * N;
  // tiled layout KsW[s][point tile][n][16]: each 16-point tile is one contiguous N x 16 block, written here and streamed by
  // k_gp_pred / k_acq_iqr front to back (a row-major N x Nstar matrix made every consumer hop 8 * Nstar bytes per step)
  double* out = KsW + ((size_t)s * gridDim.x + pt) * (size_t)N * 16;
  double fm = 0.0;
  (void)sf2;
  for (int n0 = 0; n0 < N; n0 += 16) {
    // A operand: training point n0 + li, dimensions 4q + lg
    const int na = min(n0 + li, N - 1);
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int d = 4 * q + lg;
      const double av = d < D ? xcs[(size_t)na * D + d] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, xb[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + lg + 4 * r;
      if (n < N) {
        const double cdist = fmax(aas[n] + (bb - 2.0 * acc[r]), 0.0);      // sq_dist.m:45,49
        const double ks = vb_exp_tab<0>(lsf2 - cdist / 2.0, tab);          // sf2 * exp(-K/2)  (gplite_pred.m:74)
        fm = fma(ks, al[n], fm);
        if (cv) out[(size_t)n * 16 + li] = ks * sW;                        // sW .* Ks (:99); plain Ks when !Lchol
      }
    }
  }
  fm += __shfl_xor(fm, 16, 64);
  fm += __shfl_xor(fm, 32, 64);
  if (lg == 0 && cv) partF[(size_t)s * a.Nstar + jc] = fm;
}

// k_gp_pred: V = L' \ (sW .* Ks) as the product Tinv * (sW .* Ks), Tinv = inv(L') precomputed once per GP
// (k_trsm_fwd on the identity; |inv(L')| <= 1 because L'L = K/sl + I >= I, so the explicit inverse is as well
// conditioned as the substitution).  The operand that is REUSED is made resident:
//   * a workgroup owns a block of up to 8 consecutive 16-row tiles of Tinv (rows [r0, r1), columns [0, r1): lower
//     triangle), as large as one CU's LDS allows (156 KB), and keeps it there for its whole life;
//   * its 16 waves stream over (a z-slice of) the test-point tiles: per k-step (4 training points) a lane reads ONE cross-kernel value
//     of k_pred_ks's matrix (the MFMA B operand, coalesced over the 16 points, two steps ahead) and issues one MFMA per
//     resident row tile (A from LDS);
//   * per point tile it writes sum_rows V^2 to a partial slot; k_pred_final adds the partials in block order:
//     fs2 = kss - sum(V.^2) (gplite_pred.m:99-100).
// Low-noise samples (Lchol = false, L = -inv(K + sn2 I), :103-104) use single-tile blocks over all columns and
// accumulate Ks .* (L*Ks) instead.  Tinv / L are read from HBM/L2 once per workgroup, not once per point tile.
#define PRED_THREADS 1024
#define PRED_MAXG 80
#define PRED_MAXR 8
#define PRED_LDS_MAX (156 * 1024)
// group table (ints): ng[s] at [s]; t0 at [S + s*PRED_MAXG + g]; t1 at [S + S*PRED_MAXG + s*PRED_MAXG + g]
// One point tile against RR resident row tiles: sum over the rows of V^2 (Lchol) or Ks .* (L Ks) (low-noise samples).
// The B operand (one cross-kernel value per lane and k-step, coalesced over the 16 points) is loaded four k-steps
// ahead of its MFMAs; the A operands come from the LDS-resident block T[col][row].
template <int RR>
__device__ __forceinline__ double pred_tile(const double* __restrict__ T, int RB, const double* __restrict__ kcol, size_t ldk, int N,
                                            int ncol, bool cv, bool lc, int tb, int li, int lg) {
  tmf4 acc[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) acc[r] = (tmf4){0.0, 0.0, 0.0, 0.0};
  auto ldb = [&](int n0) { const int n = n0 + lg; return (cv && n < N && n0 < ncol) ? kcol[(size_t)n * ldk] : 0.0; };
  double bq[4] = {ldb(0), ldb(4), ldb(8), ldb(12)};
  for (int n0 = 0; n0 < ncol; n0 += 16) {
    double bn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bn[u] = ldb(n0 + 16 + 4 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (n0 + 4 * u < ncol) {
        const double* trow = T + (size_t)(n0 + 4 * u + lg) * RB + li;
#pragma unroll
        for (int r = 0; r < RR; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(trow[16 * r], bq[u], acc[r], 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) bq[u] = bn[u];
  }
  // C layout: lane (col = li = point, row = lg + 4 reg) of each resident tile
  double part = 0.0;
  if (lc) {
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fma(acc[r][q], acc[r][q], part);   // sum(V.*V)
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                            // sum(Ks .* (L*Ks)), one row tile per block
      const int row = tb * 16 + lg + 4 * q;
      const double kv = (cv && row < N) ? kcol[(size_t)row * ldk] : 0.0;
      part = fma(kv, acc[0][q], part);
    }
  }
  return part;
}

__global__ void __launch_bounds__(PRED_THREADS, 4) k_gp_pred(PredArgs a, const double* __restrict__ KsW, const int* __restrict__ grp,
                                                             double* __restrict__ partV /* G x S x Nstar */) {
  extern __shared__ double lds[];
  const int g = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  if (g >= grp[s]) return;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int N = a.N;
  const int tb = grp[a.S + s * PRED_MAXG + g], te = grp[a.S + a.S * PRED_MAXG + s * PRED_MAXG + g];
  const int R = te - tb;                       // resident row tiles
  const int RB = R * 16;                       // resident rows
  const bool lc = a.lchol[s] != 0;
  const int Np = ((N + 15) >> 4) << 4;
  const int ncol = lc ? te * 16 : Np;          // columns that matter (lower triangle / full)
  double* T = lds;                             // ncol x RB: T[col * RB + row_local]
  const double* Am = (lc ? a.tinv : a.L) + (size_t)s * N * N;   // element (row, col) at col*N + row
  for (int idx = tid; idx < ncol * RB; idx += PRED_THREADS) {
    const int col = idx / RB, rl = idx % RB, row = tb * 16 + rl;
    T[idx] = (col < N && row < N) ? Am[(size_t)col * N + row] : 0.0;
  }
  __syncthreads();
  const int ntile = (a.Nstar + 15) >> 4;
  const double* ksw = KsW + (size_t)s * ntile * N * 16;   // [point tile][n][16]
  for (int pt = wave + (PRED_THREADS / 64) * blockIdx.z; pt < ntile; pt += (PRED_THREADS / 64) * gridDim.z) {
    const int jc = pt * 16 + li;
    const bool cv = jc < a.Nstar;
    const double* kcol = ksw + (size_t)pt * N * 16 + li;
    double part = 0.0;
    switch (R) {   // one straight-line instantiation per number of resident row tiles
#define PRED_CASE(RR) case RR: part = pred_tile<RR>(T, RB, kcol, (size_t)16, N, ncol, cv, lc, tb, li, lg); break;
      PRED_CASE(1) PRED_CASE(2) PRED_CASE(3) PRED_CASE(4) PRED_CASE(5) PRED_CASE(6) PRED_CASE(7) PRED_CASE(8)
#undef PRED_CASE
      default: break;
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lg == 0 && cv) partV[((size_t)g * a.S + s) * a.Nstar + jc] = part;
  }
}

// k_pred_fused (round 6): gplite_pred's variance WITHOUT the cross-kernel matrix in memory (gplite_pred.m:74-104).  k_pred_ks wrote
// sW .* Ks for every (hyper-sample, training point, test point) -- 524 MB at N* = 8192, S = 20, N = 400 -- and every row group of
// k_gp_pred streamed its columns of it again (2.2 GB against ~16 MB of inputs and outputs).  Here the roles of the two operands are
// swapped: a workgroup computes the N x 16 cross-kernel tiles of PT point tiles ONCE, into LDS (each value exactly once on the whole
// device: the distances as QS MFMAs per 16 x 16 block and the table exponential, as k_pred_ks), and its sixteen waves then walk the
// row tiles of inv(L') -- 640 KB per hyper-sample at N = 400, read through the L2 (the workgroups in flight share one or two
// hyper-samples: x fastest in the grid) -- with every A operand loaded once per k-step and used against the PT resident tiles.
// Row tiles are dealt to the waves in snake order over the four SIMD classes (tile b costs b + 1 k-steps x 4: the sums per SIMD, not
// per wave, are what the matrix pipe sees).  Low-noise samples (Lchol = false, :103-104) take all columns and accumulate
// Ks .* (L Ks).  fmu's data term Ks' alpha (:83) is a dot product over the resident tile.  Partial sums: one per hyper-sample and
// point (block 0 of partV; k_pred_final is told there is one block).
// dynamic LDS: PT x Np x 16 doubles (beside 2 KB of table and NWV x PT x 16 doubles of per-wave sums).
#define PREDF_THREADS 768      // twelve waves, three per SIMD (168 registers: sixteen at 128 spilled 37; eight left the matrix pipe half idle behind the L2)
#define PREDF_MAXPT 3
// resident point tiles the registers allow at QS dim-blocks (168 registers at three waves per SIMD; beyond: spills -- tests/test_lane_build.py)
#define PREDF_PT_FOR_QS(QS_) ((QS_) <= 3 ? 3 : ((QS_) <= 5 ? 2 : 1))
template <int QS, int PT>
__global__ void __launch_bounds__(PREDF_THREADS, 1) k_pred_fused(PredArgs a, const double* __restrict__ Xc, const double* __restrict__ aa,
                                                                 const double* __restrict__ muv, double* __restrict__ partV,
                                                                 double* __restrict__ partF, const int RS) {
  // RS > 1 (few points: the units do not cover the chip): RS workgroups share a unit -- each computes the unit's tiles (cheap) and takes
  // every RS-th row tile of inv(L'); their partial sums are blocks rs of partV, which k_pred_final adds in block order.  The importance
  // sampler's predictions of ~110 points were 60 workgroups walking 25 row tiles each: 74 us a call, 187 dependent calls.
  constexpr int NWV = PREDF_THREADS / 64;
  extern __shared__ double KsL[];              // [p][n][16], sW-scaled, zero for n >= N and for points beyond Nstar
  __shared__ double tab[VB_EXP_TAB_N];
  __shared__ double FMW[NWV][PT][16];          // fmu's data term per wave (its row blocks), added in wave order
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int N = a.N, D = a.D, Np = ((N + 15) >> 4) << 4, nblk = Np >> 4;
  const int ntile = (a.Nstar + 15) >> 4, npass = (ntile + PT - 1) / PT, nunit = npass * a.S * RS;
  for (int t = tid; t < VB_EXP_TAB_N; t += PREDF_THREADS) tab[t] = c_exp2_tab[t];
  __syncthreads();
  // A workgroup's LDS is full -- two never share a compute unit -- so the grid is one workgroup per unit of the chip, each walking the
  // units (hyper-sample, pass) b, b + gridDim.x, ...: what a workgroup does once (launch, table) is paid once, the test points of the
  // next unit are in flight behind the current one, and at any moment the workgroups in flight share one or two hyper-samples' factors
  for (int unit = blockIdx.x; unit < nunit; unit += gridDim.x) {
    const int rs = unit % RS, u2 = unit / RS;
    const int s = u2 / npass, pass = u2 - s * npass;
    const int pt0 = pass * PT;
    const int npt = min(PT, ntile - pt0);
    const double* h = a.hyp + (size_t)s * a.Nhyp;
    const double* mu = muv + (size_t)s * 2 * D;
    const double* iell = mu + D;
    const double lsf2 = 2.0 * h[D];
    const bool lc = a.lchol[s] != 0;
    const double sW = lc ? 1.0 / sqrt(a.sn2_eff[s]) : 1.0;
    const double* xcs = Xc + (size_t)s * N * D;
    const double* aas = aa + (size_t)s * N;
    const double* al = a.alpha + (size_t)s * N;
    const double* Am = (lc ? a.tinv : a.L) + (size_t)s * N * N;   // element (row, col) at col * N + row
    const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)Am, 0, (int)((size_t)N * N * 8), 0x00020000);
    // ---- the cross-kernel tiles: 16 x 16 blocks (training points n0 .. n0 + 15 x the tile's points).  A wave takes the row blocks
    // n0 = 16 (wave + NWV i) of EVERY resident point tile; the training-point fragment, |a|^2 and alpha of a row block are loaded once per
    // block and one block ahead; fmu's data term Ks' alpha (:83) is accumulated on the way
    {
      double xb[PT][QS], bb[PT], fmacc[PT];
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const bool cv = p < npt && (pt0 + p) * 16 + li < a.Nstar;
        bb[p] = 0.0; fmacc[p] = 0.0;
#pragma unroll
        for (int q = 0; q < QS; ++q) {
          const int d = 4 * q + lg;
          xb[p][q] = (cv && d < D) ? fma(a.Xs[(pt0 + p) * 16 + li + (size_t)a.Nstar * d], iell[d], -mu[d]) : 0.0;
          bb[p] = fma(xb[p][q], xb[p][q], bb[p]);
        }
        bb[p] = xor_sum16(bb[p]);
        bb[p] = xor_sum32(bb[p]);
      }
      auto ld_a = [&](int nb, double (&av)[QS], double (&an)[4]) {
        const int n0 = nb * 16, na = min(n0 + li, N - 1);
#pragma unroll
        for (int q = 0; q < QS; ++q) { const int d = 4 * q + lg; av[q] = (d < D && nb < nblk) ? xcs[(size_t)na * D + d] : 0.0; }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = n0 + lg + 4 * r; an[r] = (n < N && nb < nblk) ? aas[n] : 0.0; }
      };
      double av[QS], an[4];
      ld_a(wave, av, an);
      for (int nb = wave; nb < nblk; nb += NWV) {
        double avn[QS], ann[4], ah[4];
        ld_a(nb + NWV, avn, ann);
        const int n0 = nb * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = n0 + lg + 4 * r; ah[r] = n < N ? al[n] : 0.0; }     // (alpha: consumed after the exponentials)
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < QS; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], xb[p][q], acc, 0, 0, 0);
          const bool cv = p < npt && (pt0 + p) * 16 + li < a.Nstar;
          double* out = KsL + ((size_t)p * Np + n0) * 16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = n0 + lg + 4 * r;
            double v = 0.0;
            if (n < N && cv) {
              const double cdist = fmax(an[r] + (bb[p] - 2.0 * acc[r]), 0.0);      // sq_dist.m:45,49
              v = vb_exp_tab<0>(lsf2 - cdist / 2.0, tab);                          // sf2 exp(-K/2)  (gplite_pred.m:74)
            }
            fmacc[p] = fma(v, ah[r], fmacc[p]);
            out[(lg + 4 * r) * 16 + li] = v * sW;                                  // sW .* Ks  (:99)
          }
        }
#pragma unroll
        for (int q = 0; q < QS; ++q) av[q] = avn[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) an[r] = ann[r];
      }
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        double fm = xor_sum16(fmacc[p]);
        fm = xor_sum32(fm);
        if (lg == 0) FMW[wave][p][li] = fm;
      }
    }
    __syncthreads();
    if (tid < npt * 16) {
      const int p = tid >> 4, i = tid & 15, jc = (pt0 + p) * 16 + i;
      double fm = 0.0;
      for (int w = 0; w < NWV; ++w) fm += FMW[w][p][i];
      if (jc < a.Nstar && rs == 0) partF[(size_t)s * a.Nstar + jc] = fm;
    }
    // ---- the product: this wave's row tiles against the PT resident tiles
    double part[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) part[p] = 0.0;
    for (int k0 = rs; k0 < nblk; k0 += RS) {
      // this workgroup's row tiles k0 = rs, rs + RS, .. (descending cost); the i-th of them -> SIMD class in snake order, wave within the class in turn
      const int k = k0 / RS;
      const int kq = k & 7, cls = kq < 4 ? kq : 7 - kq, wsel = cls + 4 * ((k >> 2) % (NWV / 4));
      if (wsel != wave) continue;
      const int rt = nblk - 1 - k0;
      const int ncol = lc ? (rt + 1) * 16 : Np;
      const int row = rt * 16 + li;
      const bool rv = row < N;
      tmf4 acc[PT];
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[p] = (tmf4){0.0, 0.0, 0.0, 0.0};
      // Operands in chunks of four k-steps, BOTH loaded one chunk ahead of their MFMAs into two register sets used in turn: the A operands
      // (inv(L'), from the L2) and the B operands (the resident tiles, from LDS).  The scheduling barriers keep the next chunk's loads
      // ABOVE the current chunk's MFMAs: left to itself the compiler sank half of the global loads and every LDS read down to right in
      // front of their consumers (s_waitcnt vmcnt / lgkmcnt directly ahead of every second MFMA: half of the waves' cycles were such
      // waits, SQ_WAIT_INST_ANY in profiles/r06_aux.md).
      // (A through a buffer descriptor: base and extent in scalar registers, ONE 32-bit offset per lane and a scalar add per load -- the
      // 64-bit address, the compare and the select per element were 3 VALU instructions per MFMA of this loop, and on this chip every VALU
      // instruction of a SIMD waits while an fp64 MFMA executes.  Columns beyond N lie past the extent and read 0; the lanes of rows
      // beyond N start 2 GB up and stay past it)
      const unsigned vo = rv ? (unsigned)(((size_t)row + (size_t)lg * N) * 8) : 0x80000000u;
      const double* bl = KsL + (size_t)lg * 16 + li;
      auto lda4 = [&](int c0, double (&d)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          d[u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Ar, (int)(vo + (unsigned)((c0 + 4 * u) * N * 8)), 0, 0));
      };
      auto ldb4 = [&](int c0, double (&b)[4][PT]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int p = 0; p < PT; ++p) b[u][p] = bl[(size_t)(c0 + 4 * u) * 16 + (size_t)p * Np * 16];
      };
      auto mm4 = [&](const double (&d)[4], const double (&b)[4][PT]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int p = 0; p < PT; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[u], b[u][p], acc[p], 0, 0, 0);
      };
      double a0[4], a1[4], q0[4][PT], q1[4][PT];
      lda4(0, a0);
      ldb4(0, q0);
      for (int c0 = 0; c0 < ncol; c0 += 32) {
        const bool two = c0 + 16 < ncol;
        if (two) { lda4(c0 + 16, a1); ldb4(c0 + 16, q1); }
        __builtin_amdgcn_sched_barrier(0);
        mm4(a0, q0);
        if (two) {
          if (c0 + 32 < ncol) { lda4(c0 + 32, a0); ldb4(c0 + 32, q0); }
          __builtin_amdgcn_sched_barrier(0);
          mm4(a1, q1);
        }
      }
      // C layout: lane (col = li = point, row = rt * 16 + lg + 4 q)
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        if (lc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) part[p] = fma(acc[p][q], acc[p][q], part[p]);     // sum(V .* V)          (:100)
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)                                                   // sum(Ks .* (L * Ks))  (:104)
            part[p] = fma(KsL[((size_t)p * Np + rt * 16 + lg + 4 * q) * 16 + li], acc[p][q], part[p]);
        }
      }
    }
    // ---- the waves' partial sums, added in wave order (the tiles' LDS is dead: it holds them)
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      double v = xor_sum16(part[p]);
      v = xor_sum32(v);
      if (lg == 0) KsL[(wave * PREDF_MAXPT + p) * 16 + li] = v;
    }
    __syncthreads();
    if (tid < npt * 16) {
      const int p = tid >> 4, i = tid & 15, jc = (pt0 + p) * 16 + i;
      double v = 0.0;
      for (int w = 0; w < NWV; ++w) v += KsL[(w * PREDF_MAXPT + p) * 16 + i];
      if (jc < a.Nstar) partV[((size_t)rs * a.S + s) * a.Nstar + jc] = v;
    }
    __syncthreads();      // the next unit writes the tiles
  }
}

// k_pred_slab: the large-N form of the prediction variance (N beyond what k_gp_pred keeps resident: a 16-row tile of the
// triangular inverse no longer fits the LDS).  One wave per CW test points: their sW-scaled cross-kernel columns form a slab
// in LDS, V = L' \ (sW .* Ks) by the blocked substitution of trsm_mfma.h (gplite_pred.m:99), sum(V.^2) per point (:100);
// low-noise samples (L = -inv(K + sn2 I)) accumulate Ks .* (L * Ks) (:103-104) with lanes along the rows.  Writes block
// partial 0 (k_pred_final is told there is one block per hyper-sample).
template <int CW>
__global__ void __launch_bounds__(64) k_pred_slab(PredArgs a, const double* __restrict__ KsW, double* __restrict__ partV) {
  extern __shared__ double lds[];
  const int cb = blockIdx.x, s = blockIdx.y, lane = threadIdx.x;
  const int N = a.N, Np = ((N + 15) >> 4) << 4;
  const int ntile = (a.Nstar + 15) >> 4;
  const int j0 = cb * CW;                       // first test point of the slab
  double* V = lds;
  double* P = V + (size_t)Np * (CW + 1);
  const double* ksw = KsW + (size_t)s * ntile * N * 16;   // [point tile][n][16]
  for (int c = 0; c < CW; ++c) {
    const int j = j0 + c;
    const double* col = ksw + (size_t)(j >> 4) * N * 16 + (j & 15);
    for (int i = lane; i < Np; i += 64) V[i * (CW + 1) + c] = (j < a.Nstar && i < N) ? col[(size_t)i * 16] : 0.0;
  }
  trsm_wsync();
  if (a.lchol[s]) {
    trsm_fwd_wave<CW>(N, a.L + (size_t)s * N * N, a.finv + (size_t)s * TRSM_NBLK(N) * 256, V, P, lane);
    for (int c = 0; c < CW; ++c) {
      double part = 0.0;
      for (int i = lane; i < N; i += 64) { const double v = V[i * (CW + 1) + c]; part = fma(v, v, part); }
      part = wave_sum(part);
      if (lane == 0 && j0 + c < a.Nstar) partV[(size_t)s * a.Nstar + j0 + c] = part;
    }
  } else {
    const double* Lm = a.L + (size_t)s * N * N;    // symmetric: element (i, j) at j * N + i
    for (int c = 0; c < CW; ++c) {
      double part = 0.0;
      for (int i = lane; i < N; i += 64) {
        double acc = 0.0;
        for (int j = 0; j < N; ++j) acc = fma(Lm[(size_t)j * N + i], V[j * (CW + 1) + c], acc);
        part = fma(V[i * (CW + 1) + c], acc, part);
      }
      part = wave_sum(part);
      if (lane == 0 && j0 + c < a.Nstar) partV[(size_t)s * a.Nstar + j0 + c] = part;
    }
  }
}

// fmu = m* + Ks' alpha (:83), fs2 = max(kss -/+ sum of the block partials, 0) (:100,:104,:120), ys2 (:121)
__global__ void __launch_bounds__(256) k_pred_final(PredArgs a, const int* __restrict__ grp, const double* __restrict__ partV,
                                                    const double* __restrict__ partF) {
  const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (i >= a.Nstar) return;
  const int D = a.D;
  const double* h = a.hyp + (size_t)s * a.Nhyp;
  const double sf2 = exp(2.0 * h[D]);
  const bool lc = a.lchol[s] != 0;
  double pv = 0.0;
  for (int g = 0; g < grp[s]; ++g) pv += partV[((size_t)g * a.S + s) * a.Nstar + i];
  const double mstar = gp_meanfun(a.meanfun, D, h + a.moff, a.Xs + i, (size_t)a.Nstar);
  const double fmu = mstar + partF[(size_t)s * a.Nstar + i];
  double fs2 = lc ? sf2 - pv : sf2 + pv;
  fs2 = fmax(fs2, 0.0);
  // noise at the test points (gplite_noisefun.m:176-207); the output-dependent term only with a non-empty ystar (:199)
  double sn2s = a.nf0 ? exp(2.0 * h[a.noff]) : 2.220446049250313e-16;
  if (a.nf1 == 1 && a.s2s) sn2s += a.s2s[i];
  else if (a.nf1 == 2 && a.s2s) sn2s += exp(h[a.noff + (a.nf0 ? 1 : 0)]) * a.s2s[i];
  if (a.nf2 == 1 && a.ys) {
    const int io = a.noff + (a.nf0 ? 1 : 0) + (a.nf1 == 2 ? 1 : 0);
    const double zz = fmax(0.0, h[io] - a.ys[i]);
    sn2s += exp(2.0 * h[io + 1]) * zz * zz;
  }
  a.fmu[i + (size_t)a.Nstar * s] = fmu;
  a.fs2[i + (size_t)a.Nstar * s] = fs2;
  a.ys2[i + (size_t)a.Nstar * s] = fs2 + sn2s * a.sn2_mult[s];
}

// gplite_pred.m:154-165 averaging over hyper-samples (in place into column 0 of the *_avg outputs)
__global__ void k_pred_avg(int Nstar, int S, const double* __restrict__ fmu, const double* __restrict__ fs2,
                           const double* __restrict__ ys2, double* __restrict__ out /* 4 x Nstar: ymu ys2 fmu fs2 */) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nstar) return;
  double fbar = 0.0, f2 = 0.0, y2 = 0.0;
  for (int s = 0; s < S; ++s) { fbar += fmu[i + (size_t)Nstar * s]; f2 += fs2[i + (size_t)Nstar * s]; y2 += ys2[i + (size_t)Nstar * s]; }
  fbar /= S;
  double vf = 0.0;
  for (int s = 0; s < S; ++s) { double d = fmu[i + (size_t)Nstar * s] - fbar; vf += d * d; }
  vf /= (S - 1);
  out[i] = fbar;                              // ymu = fmu without output warping
  out[i + (size_t)Nstar] = y2 / S + vf;       // ys2
  out[i + 2 * (size_t)Nstar] = fbar;          // fmu
  out[i + 3 * (size_t)Nstar] = f2 / S + vf;   // fs2
}

// Cross-covariance column for the rank-1 update (gplite_post.m:210-216): Ks[s][n] = k_s(X_n, x*),
// with sq_dist's two-argument centring for m = 1 test point.
__global__ void __launch_bounds__(256) k_gp_ks(int N, int D, int Nhyp, const double* __restrict__ X,
                                               const double* __restrict__ xstar, const double* __restrict__ hyp,
                                               const double* __restrict__ meanX, double* __restrict__ Ks) {
  const int s = blockIdx.y;
  const double* h = hyp + (size_t)s * Nhyp;
  __shared__ double ell[32], mu[32], xs[32];
  if (threadIdx.x < D) {
    const int d = threadIdx.x;
    ell[d] = exp(h[d]);
    const double n = (double)N, m = 1.0;
    mu[d] = (m / (n + m)) * (xstar[d] / ell[d]) + (n / (n + m)) * (meanX[d] / ell[d]);
    xs[d] = xstar[d] / ell[d] - mu[d];
  }
  __syncthreads();
  const double sf2 = exp(2.0 * h[D]);
  double bb = 0.0;
  for (int d = 0; d < D; ++d) bb = fma(xs[d], xs[d], bb);
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    double aa = 0.0, dot = 0.0;
    for (int d = 0; d < D; ++d) {
      double av = X[n + (size_t)N * d] / ell[d] - mu[d];
      aa = fma(av, av, aa);
      dot = fma(av, xs[d], dot);
    }
    Ks[(size_t)s * N + n] = sf2 * exp(-fmax(aa + (bb - 2.0 * dot), 0.0) / 2.0);
  }
}

// ------------------------------------------------------------------------------------------
// gplite_nlZ (gplite/private/gplite_core.m:128-275, covfun 1, no integrated mean / output warping)
// ------------------------------------------------------------------------------------------
// Partial sums of the Q-contractions over one 64 x 64 tile of (k, j):
//   Q = Kinv/sl - alpha*alpha'                                  (:240, Kinv = L\(L'\eye(N)))
//   part[d]  = sum Q .* K .* sq_dist(X(:,d)'/ell_d), d < D       (:244-247; the 1/2 is applied in k_nlz_final)
//   part[D]  = sum Q .* K                                       (:248)
//   part[D+1+i] = sum_j dsn2(j,i) Q_jj                          (:257-262)
// K is rebuilt exactly as k_gp_build forms it.  Fixed-order block reduction, no atomics.
#define NLZ_T 64
// One workgroup per 64 x 64 tile (k along the lanes, 16 rows j per wave) of the upper triangle of tiles; off-diagonal
// tiles count twice (Q, K and the distance matrices are symmetric), tiles below the diagonal write zeros.
template <int DT>
__global__ void __launch_bounds__(256) k_nlz_grad(int N, int D, int Nhyp, int Nnoise, const double* __restrict__ hyp,
                                                  const double* __restrict__ Xc, const double* __restrict__ aa,
                                                  const double* __restrict__ Kinv, const double* __restrict__ alpha,
                                                  const double* __restrict__ scal, const double* __restrict__ dsn2,  // B x Nnoise x N
                                                  double* __restrict__ part) {
  __shared__ double red[256];
  __shared__ double TAB[VB_EXP_TAB_N];
  __shared__ double XK[DT * NLZ_T], XJ[DT * NLZ_T];  // [d][point]
  __shared__ double AJ[NLZ_T], ALJ[NLZ_T];
  const int b = blockIdx.z, tid = threadIdx.x, tk = tid & 63, tj = tid >> 6;
  const int k0 = blockIdx.x * NLZ_T, j0 = blockIdx.y * NLZ_T;
  const int P = D + 1 + Nnoise;
  double* o = part + ((size_t)b * gridDim.x * gridDim.y + (size_t)blockIdx.y * gridDim.x + blockIdx.x) * P;
  if (k0 > j0) {
    for (int p = tid; p < P; p += 256) o[p] = 0.0;
    return;
  }
  const double* h = hyp + (size_t)b * Nhyp;
  const double sf2 = exp(2.0 * h[D]);
  const double isl = 1.0 / scal[b * 4 + 3];
  const double* xs = Xc + (size_t)b * N * D;
  const double* as = aa + (size_t)b * N;
  const double* Kb = Kinv + (size_t)b * N * N;
  const double* al = alpha + (size_t)b * N;
  TAB[tid] = c_exp2_tab[tid];
  for (int e = tid; e < NLZ_T * (DT - D); e += 256) { XK[D * NLZ_T + e] = 0.0; XJ[D * NLZ_T + e] = 0.0; }
  for (int e = tid; e < NLZ_T * D; e += 256) {
    const int pnt = e / D, d = e - pnt * D;
    XK[d * NLZ_T + pnt] = (k0 + pnt < N) ? xs[(size_t)k0 * D + e] : 0.0;
    XJ[d * NLZ_T + pnt] = (j0 + pnt < N) ? xs[(size_t)j0 * D + e] : 0.0;
  }
  if (tid < NLZ_T) {
    AJ[tid] = (j0 + tid < N) ? as[j0 + tid] : 0.0;
    ALJ[tid] = (j0 + tid < N) ? al[j0 + tid] : 0.0;
  }
  __syncthreads();
  double acc[DT], accK = 0.0, accN[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int p = 0; p < DT; ++p) acc[p] = 0.0;
  const int k = k0 + tk;
  if (k < N) {
    double xk[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) xk[d] = XK[d * NLZ_T + tk];
    const double ak = as[k], alk = al[k];
    const double sym = (k0 < j0) ? 2.0 : 1.0;
    for (int t = 0; t < 16; ++t) {
      const int jl = tj * 16 + t, j = j0 + jl;
      if (j >= N) break;
      double dot = 0.0;
#pragma unroll
      for (int d = 0; d < DT; ++d) dot = fma(xk[d], XJ[d * NLZ_T + jl], dot);
      const double c = fmax(AJ[jl] + (ak - 2.0 * dot), 0.0);
      const double kv = sf2 * vb_exp_tab<0>(-0.5 * c, TAB);
      const double q = Kb[(size_t)j * N + k] * isl - ALJ[jl] * alk;    // column j read along k (coalesced)
      const double qk = sym * (q * kv);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const double df = XJ[d * NLZ_T + jl] - xk[d];
        acc[d] = fma(qk, df * df, acc[d]);
      }
      accK += qk;
      if (k == j) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < Nnoise) accN[i] = dsn2[((size_t)b * Nnoise + i) * N + j] * q;
      }
    }
  }
  // fixed order: VALU butterfly inside a wave, then the four waves in turn -- one barrier for all D + 1 + Nnoise sums
  // (round 4: a block_sum -- twelve ds_bpermute and two barriers -- per sum)
  __shared__ double wr[4][DT + 5];
  {
    const int wv = tid >> 6, ln = tid & 63;
    double v;
#pragma unroll
    for (int d = 0; d < DT; ++d) { v = wave_sum_valu(acc[d]); if (ln == 0) wr[wv][d] = v; }
    v = wave_sum_valu(accK); if (ln == 0) wr[wv][DT] = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v = wave_sum_valu(accN[i]); if (ln == 0) wr[wv][DT + 1 + i] = v; }
  }
  __syncthreads();
  if (tid < DT + 5) {
    const double t = ((wr[0][tid] + wr[1][tid]) + wr[2][tid]) + wr[3][tid];
    if (tid < D) o[tid] = t;
    else if (tid == DT) o[D] = t;
    else if (tid > DT && tid - DT - 1 < Nnoise) o[D + 1 + (tid - DT - 1)] = t;
  }
  (void)red;
}

// D -> padded DT for k_gp_build / k_nlz_grad
#define DISPATCH_GPDT(D_, ...)                                   \
  do {                                                           \
    if ((D_) <= 4) { constexpr int DT = 4; __VA_ARGS__; }        \
    else if ((D_) <= 8) { constexpr int DT = 8; __VA_ARGS__; }   \
    else if ((D_) <= 12) { constexpr int DT = 12; __VA_ARGS__; } \
    else if ((D_) <= 16) { constexpr int DT = 16; __VA_ARGS__; } \
    else if ((D_) <= 24) { constexpr int DT = 24; __VA_ARGS__; } \
    else { constexpr int DT = 32; __VA_ARGS__; }                 \
  } while (0)

// The closing kernel of gplite_nlZ, one block per hyper-parameter vector:
//   nlZ  = (y-m)'*alpha/2 + sum(log(diag(L))) + N*log(2*pi*sl)/2                                   (gplite_core.m:205)
//   dnlZ = the tile partials of k_nlz_grad summed in tile order (grad != 0), and the mean-function block -dm'*alpha
//          (:274, gplite_meanfun.m:402,406,433-435)
// out = [nlZ B | failure index of the factorisation B | dnlZ B x Nhyp]: the one block that travels back to the host.
// (Round 5: the value, the tile sums and the 2 D + 1 mean-function dot products were two kernels, the second walking them one
// wave per hyper-parameter with dependent loads -- 7 + 15..24 us; now one pass over n per thread, every load of a thread in
// flight at once, wave sums and a fixed-order sum over the four waves.)
template <int DT>
__global__ void __launch_bounds__(256) k_nlz_final(int N, int D, int Nhyp, int Nnoise, int Nmean, int meanfun, int ntile, int grad,
                                                   const double* __restrict__ X, const double* __restrict__ y,
                                                   const double* __restrict__ hyp, const double* __restrict__ A,
                                                   const double* __restrict__ alpha, const double* __restrict__ scal,
                                                   const double* __restrict__ part, const double* __restrict__ pfd,
                                                   double* __restrict__ out) {
  __shared__ double wred[4][2 * DT + 3];
  __shared__ double sxm[DT], som[DT];
  const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int P = D + 1 + Nnoise, moff = D + 1 + Nnoise;
  const double* hm = hyp + (size_t)b * Nhyp + moff;
  const double* al = alpha + (size_t)b * N;
  const double* Ab = A + (size_t)b * N * N;
  double* g = out + 2 * (size_t)B + (size_t)b * Nhyp;
  if (tid < DT) {
    const bool on = meanfun == 4 && tid < D;
    sxm[tid] = on ? hm[1 + tid] : 0.0;
    som[tid] = on ? exp(hm[D + 1 + tid]) : 1.0;
  }
  if (pfd && tid == 0) out[B + b] = pfd[b];
  __shared__ double tq[4][64];
  if (grad && lane < P) {
    // the tile partials: wave q takes the q-th quarter of the tiles (in tile order, eight loads in flight), the quarters are
    // added in order below
    const int per = (ntile + 3) >> 2, j0 = wave * per, j1 = min(ntile, j0 + per);
    double t = 0.0;
    for (int jt = j0; jt < j1; jt += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = jt + u < j1 ? part[((size_t)b * ntile + jt + u) * P + lane] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    tq[wave][lane] = t;
  }
  __syncthreads();
  if (grad && tid < P) {
    const double t = ((tq[0][tid] + tq[1][tid]) + tq[2][tid]) + tq[3][tid];
    const double mult = scal[b * 4 + 1];
    if (tid < D) g[tid] = t / 2.0;                 // sum(sum(Q.*K_temp))/2
    else if (tid == D) g[D] = t;                   // sum(sum(Q.*(2*K_mat)))/2
    else g[tid] = 0.5 * mult * t;                  // 0.5*sn2_mult*sum(dsn2(:,i).*dgQ)
  }
  double t1[DT], t2[DT], t0 = 0.0, quad = 0.0, ld = 0.0;
#pragma unroll
  for (int d = 0; d < DT; ++d) { t1[d] = 0.0; t2[d] = 0.0; }
  for (int n = tid; n < N; n += 256) {
    const double a = al[n], yn = y[n], dg = Ab[(size_t)n * N + n];
    double x[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) x[d] = (meanfun == 4 && d < D) ? X[n + (size_t)N * d] : 0.0;
    double z2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const double z = (x[d] - sxm[d]) / som[d];
      z2 = fma(z, z, z2);
      t1[d] = fma(z / som[d], a, t1[d]);           // gplite_meanfun.m:433-435
      t2[d] = fma(z * z, a, t2[d]);
    }
    const double m = meanfun == 0 ? 0.0 : (meanfun == 1 ? hm[0] : hm[0] - 0.5 * z2);    // gplite_meanfun.m:425-431
    quad = fma(yn - m, a, quad);
    ld += log(dg);
    t0 += a;
  }
  // fixed order: butterfly inside a wave, then the four waves in turn
  {
    double v;
    v = wave_sum_valu(t0); if (lane == 0) wred[wave][0] = v;
    v = wave_sum_valu(quad); if (lane == 0) wred[wave][1] = v;
    v = wave_sum_valu(ld); if (lane == 0) wred[wave][2] = v;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      v = wave_sum_valu(t1[d]); if (lane == 0) wred[wave][3 + d] = v;
      v = wave_sum_valu(t2[d]); if (lane == 0) wred[wave][3 + DT + d] = v;
    }
  }
  __syncthreads();
  if (tid < 2 * DT + 3) {
    const double t = ((wred[0][tid] + wred[1][tid]) + wred[2][tid]) + wred[3][tid];
    wred[0][tid] = t;
  }
  __syncthreads();
  if (tid == 0) out[b] = wred[0][1] / 2.0 + wred[0][2] + N * log(2.0 * 3.14159265358979323846 * scal[b * 4 + 3]) / 2.0;
  if (grad && tid < Nmean) {
    const int i = tid;
    double t;
    if (i == 0) t = wred[0][0];
    else if (i <= D) t = wred[0][3 + (i - 1)];
    else t = wred[0][3 + DT + (i - 1 - D)];
    g[moff + i] = -t;
  }
}


// ------------------------------------------------------------------------------------------
// Acquisition sweep (acq/acqwrapper_vbmc.m:19-46) fused behind the prediction: per test point the
// hyper-sample statistics fbar / vtot (:21-29), the variational-posterior density p = max(vbmc_pdf(vp,Xs,0),
// realmin) (vbmc_pdf.m:57-63), the acquisition value (acqf_vbmc.m:9-10, acqflog_vbmc.m:17-18, acqus_vbmc.m:9,
// acqfsn2_vbmc.m:9-17), the variance regularisation (:35-45) and the -realmax clamp (:46).
// ------------------------------------------------------------------------------------------
struct AcqArgs {
  int Nstar, S, D, K, N, acq_id, reg;
  double ymax, TolVar;
  const double* Xs;     // Nstar x D col-major
  const double* fmu;    // Nstar x S
  const double* fs2;    // Nstar x S
  const double* mu;     // D x K
  const double* isl;    // K x D:  1 / (sigma_k lambda_d)
  const double* coef;   // K:      nf * w_k / sigma_k^D
  const double* sn2x;   // Nstar   gp.sn2new at the nearest row of gp.X_rescaled (k_nn_noise)   (acq_id 3)
  double* acq;          // Nstar
  double* fbar;         // Nstar
  double* vtot;         // Nstar
};

__global__ void __launch_bounds__(256) k_acq(AcqArgs a) {
  extern __shared__ double lds[];
  const int D = a.D, K = a.K, S = a.S, Nstar = a.Nstar;
  double* s_mu = lds;              // K x D (k-major)
  double* s_isl = s_mu + K * D;    // K x D
  double* s_coef = s_isl + K * D;  // K
  for (int idx = threadIdx.x; idx < K * D; idx += 256) {
    const int k = idx / D, d = idx % D;
    s_mu[idx] = a.mu[d + (size_t)D * k];
    s_isl[idx] = a.isl[idx];
  }
  for (int k = threadIdx.x; k < K; k += 256) s_coef[k] = a.coef[k];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nstar) return;
  double fbar = 0.0, vbar = 0.0;
  for (int s = 0; s < S; ++s) { fbar += a.fmu[i + (size_t)Nstar * s]; vbar += a.fs2[i + (size_t)Nstar * s]; }
  fbar /= S; vbar /= S;
  double vf = 0.0;
  if (S > 1) {
    for (int s = 0; s < S; ++s) { const double d = a.fmu[i + (size_t)Nstar * s] - fbar; vf += d * d; }
    vf /= (S - 1);
  }
  const double vtot = vf + vbar;
  double x[32];
#pragma unroll
  for (int d = 0; d < 32; ++d) x[d] = d < D ? a.Xs[i + (size_t)Nstar * d] : 0.0;
  double p = 0.0;
  for (int k = 0; k < K; ++k) {
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < 32; ++d)
      if (d < D) { const double t = (x[d] - s_mu[k * D + d]) * s_isl[k * D + d]; d2 = fma(t, t, d2); }
    p += s_coef[k] * exp(-0.5 * d2);
  }
  p = fmax(p, 2.2250738585072014e-308);
  double acq;
  if (a.acq_id == 0) acq = -vtot * exp(fbar - a.ymax) * p;
  else if (a.acq_id == 1) acq = -(log(vtot) + fbar - a.ymax + log(p));
  else if (a.acq_id == 2) acq = -vtot * p * p;
  else {
    const double sn2 = a.sn2x[i];   // observation noise at the nearest training input (k_nn_noise; acqfsn2_vbmc.m:11-13)
    acq = -vtot * (1.0 - sn2 / (vtot + sn2)) * exp(fbar - a.ymax) * p;
  }
  if (a.reg && vtot < a.TolVar) {
    if (a.acq_id == 1) acq = acq + a.TolVar / vtot - 1.0;
    else acq = acq * exp(-(a.TolVar / vtot - 1.0));
  }
  acq = fmax(acq, -1.7976931348623157e308);
  a.acq[i] = acq;
  if (a.fbar) a.fbar[i] = fbar;
  if (a.vtot) a.vtot[i] = vtot;
}

// ------------------------------------------------------------------------------------------
// Importance-sampled IQR acquisition functions (acq/acqviqr_vbmc.m:50-109, acq/acqimiqr_vbmc.m:40-95) and the
// Step-3 precomputation of private/activeimportancesampling_vbmc.m:248-276.
// ------------------------------------------------------------------------------------------
// Kax'[s] (N x Na, column a at a*N): k_s(X_n, Xa_a)  (:264-265)
__global__ void __launch_bounds__(256) k_cross_kernel(int N, int D, int Nhyp, int Na, int per_s, const double* __restrict__ X,
                                                      const double* __restrict__ Xa /* Na x D (x S) col-major */,
                                                      const double* __restrict__ hyp, double* __restrict__ Z) {
  const int s = blockIdx.y;
  const double* h = hyp + (size_t)s * Nhyp;
  __shared__ double iell[32];
  if (threadIdx.x < D) iell[threadIdx.x] = 1.0 / exp(h[threadIdx.x]);
  __syncthreads();
  const double sf2 = exp(2.0 * h[D]);
  const double* xa = Xa + (per_s ? (size_t)s * Na * D : 0);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)N * Na; idx += (size_t)gridDim.x * 256) {
    const int n = (int)(idx % N), a = (int)(idx / N);
    double c = 0.0;
    for (int d = 0; d < D; ++d) { const double t = (X[n + (size_t)N * d] - xa[a + (size_t)Na * d]) * iell[d]; c = fma(t, t, c); }
    Z[(size_t)s * N * Na + idx] = sf2 * exp(-c / 2.0);
  }
}

// CtmpT[s][n][a] (a fastest, padded to Nap with zeros) = U[s][a*N + n] / sn2_eff (Lchol, :271) or U (else, :273)
__global__ void __launch_bounds__(256) k_ctmp_pack(int N, int Na, int Nap, const double* __restrict__ U,
                                                   const double* __restrict__ sn2_eff, const unsigned char* __restrict__ lchol,
                                                   double* __restrict__ CT) {
  const int s = blockIdx.y;
  const double sc = lchol[s] ? 1.0 / sn2_eff[s] : 1.0;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)N * Nap; idx += (size_t)gridDim.x * 256) {
    const int a = (int)(idx % Nap), n = (int)(idx / Nap);
    CT[(size_t)s * N * Nap + idx] = a < Na ? U[(size_t)s * N * Na + (size_t)a * N + n] * sc : 0.0;
  }
}

// observation noise at the nearest training input in length-scale units (acqviqr_vbmc.m:42-43); first minimum wins.
// One wave per 16 test points; the inner products of |x - xr_n|^2 = |x|^2 + |xr_n|^2 - 2 x.xr_n for 16 x 16 blocks of
// (training point, test point) pairs are QS MFMAs, each lane then scans its four candidates in increasing n.
template <int QS>
__global__ void __launch_bounds__(64) k_nn_noise(int Nstar, int N, int D, const double* __restrict__ Xs, const double* __restrict__ gl,
                                                 const double* __restrict__ Xr, const double* __restrict__ sn2new,
                                                 double* __restrict__ sn2x) {
  const int lane = threadIdx.x, li = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x * 16 + li;
  const int gi = min(i, Nstar - 1);
  double xb[QS];                                   // B operand: test point li, dimensions 4q + lg
#pragma unroll
  for (int q = 0; q < QS; ++q) {
    const int d = 4 * q + lg;
    xb[q] = d < D ? Xs[gi + (size_t)Nstar * d] / gl[d] : 0.0;
  }
  double best = INFINITY;
  int pos = 0;
  for (int n0 = 0; n0 < N; n0 += 16) {
    const int na = min(n0 + li, N - 1);            // A operand: training point n0 + li, dimensions 4q + lg
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double rr = 0.0;
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int d = 4 * q + lg;
      const double av = d < D ? Xr[na + (size_t)N * d] : 0.0;
      rr = fma(av, av, rr);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, xb[q], acc, 0, 0, 0);
    }
    rr += __shfl_xor(rr, 16, 64);                  // |xr_n|^2 for n = n0 + li, on every lane group
    rr += __shfl_xor(rr, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nl = lg + 4 * r, n = n0 + nl;       // accumulator row
      const double rn = __shfl(rr, nl, 64);         // |xr_n|^2 lives on lane li == nl
      const double c = rn - 2.0 * acc[r];           // + |x|^2, the same for every n
      if (n < N && c < best) { best = c; pos = n; }
    }
  }
  // the four lane groups hold disjoint n: smallest distance, ties to the smaller index
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    const double ob = __shfl_xor(best, o, 64);
    const int op = __shfl_xor(pos, o, 64);
    if (ob < best || (ob == best && op < pos)) { best = ob; pos = op; }
  }
  if (lg == 0 && i < Nstar) sn2x[i] = sn2new[pos];
}

struct IqrArgs {
  int N, D, S, Nhyp, Nstar, Na, Nap, per_s;
  const double* Xs;      // Nstar x D col-major
  const double* Xa;      // Na x D (x S) col-major
  const double* hyp;     // Nhyp x S
  const double* Xc;      // S x N x D   ell-scaled, centred training inputs (k_pred_prep)
  const double* muv;     // S x 2D      centre, 1/ell
  const double* CT;      // S x N x Nap
  const double* fs2a;    // S x Nap
  const double* lnw;     // S x Nap (-inf in the padding) or null
  const double* fs2;     // Nstar x S   (k_gp_pred)
  const double* KsW;     // S x ceil(Nstar/16) x N x 16  sW-scaled cross-kernel matrix, tiled by 16 points (k_pred_ks)
  const double* sn2_eff; // S
  const double* sn2x;    // Nstar
  const unsigned char* lchol;
  double* acqs;          // Nstar x S
};

// One workgroup = 4 waves = 64 test points x one hyper-sample; wave w owns 16 of the points.
// C[i][a] = Ka[i][a] -/+ sum_n Ks[n][i] Ctmp[n][a] on the fp64 matrix cores: the A operand Ks[n][i] comes from the
// cross-kernel matrix k_pred_ks left in memory (one coalesced value per lane and k-step); the B operand -- rows of CtmpT,
// the same for all points -- is staged through LDS in chunks of 16 rows shared by its waves (double-buffered: the
// global loads of chunk c+1 are in flight while chunk c feeds the MFMAs), which divides the L2 traffic of the B stream by
// IQR_WPB; NT = Nap/16 accumulator tiles live at once.
// Epilogue per element: tau2 = C^2/ys2_i, s_pred = sqrt(max(fs2a_a - tau2, 0)), zz = lnw_a + u s + log1p(-exp(-2 u s)),
// then a log-sum-exp over a (16 lanes x NT tiles).
#define IQR_KC 16
#define IQR_WPB 8                       // waves (16-point tiles) per workgroup
#define IQR_PTS (16 * IQR_WPB)
#define IQR_THREADS (64 * IQR_WPB)
#define IQR_LDS_BYTES(NT) ((size_t)(2 * IQR_KC * (16 * (NT) + 8) + IQR_PTS * 33 + IQR_PTS) * sizeof(double))
template <int NT>
__global__ void __launch_bounds__(IQR_THREADS) k_acq_iqr(IqrArgs a) {
  constexpr int BS = 16 * NT + 8;          // padded row stride of a staged chunk
  constexpr int PER = (IQR_KC * 16 * NT + IQR_THREADS - 1) / IQR_THREADS;   // staged elements per thread
  extern __shared__ double iq_lds[];
  double* BL = iq_lds;                              // [2][IQR_KC][BS]
  double (*xs_s)[33] = (double (*)[33])(iq_lds + 2 * IQR_KC * BS);   // ell-scaled, centred test points of the workgroup's points
  double* ys2_s = iq_lds + 2 * IQR_KC * BS + IQR_PTS * 33;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  const int ib = blockIdx.x * IQR_PTS, i0 = ib + 16 * wv, s = blockIdx.y;
  const int N = a.N, D = a.D, Nap = a.Nap;
  const double* h = a.hyp + (size_t)s * a.Nhyp;
  const double sf2 = exp(2.0 * h[D]);
  const double* mu = a.muv + (size_t)s * 2 * D;
  const double* iell = mu + D;
  for (int idx = tid; idx < IQR_PTS * D; idx += IQR_THREADS) {
    const int i = idx / D, d = idx % D;
    const int gi = min(ib + i, a.Nstar - 1);
    xs_s[i][d] = a.Xs[gi + (size_t)a.Nstar * d] * iell[d] - mu[d];
  }
  if (tid < IQR_PTS) {
    const int gi = min(ib + tid, a.Nstar - 1);
    ys2_s[tid] = a.fs2[gi + (size_t)a.Nstar * s] + a.sn2x[gi];
  }
  d4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (d4_t){0.0, 0.0, 0.0, 0.0};
  const double* ct = a.CT + (size_t)s * N * Nap;
  const bool gvalid = i0 + li < a.Nstar;
  const double* kcol = a.KsW + ((size_t)s * ((a.Nstar + 15) >> 4) + (i0 >> 4)) * (size_t)N * 16 + li;   // [s][point tile][n][16]
  const double isw = a.lchol[s] ? sqrt(a.sn2_eff[s]) : 1.0;   // undo sW = 1/sqrt(sn2_eff)
  // chunk c = rows 16c .. 16c+15 of CtmpT (contiguous in memory: row-major N x Nap), zero beyond N
  double pre[PER];
  auto fetch = [&](int c) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + IQR_THREADS * u;
      pre[u] = (e < IQR_KC * Nap && 16 * c * Nap + e < N * Nap) ? ct[(size_t)16 * c * Nap + e] : 0.0;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + IQR_THREADS * u;
      if (e < IQR_KC * Nap) { const int row = e / Nap, col = e - row * Nap; BL[(buf * IQR_KC + row) * BS + col] = pre[u]; }
    }
  };
  const int nch = (N + IQR_KC - 1) / IQR_KC;
  // A operands of the four k-steps of a chunk: the cross-kernel values k_s(X_n, xs_i) (stored sW-scaled), loaded one
  // chunk ahead like the B rows
  double kv[4], kvn[4], kvn2[4];   // this chunk, the next one and the one after (the A stream comes from HBM)
  auto fetch_a = [&](int c, double* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 16 * c + 4 * q + lg;
      dst[q] = (n < N && gvalid) ? kcol[(size_t)n * 16] : 0.0;
    }
  };
  fetch(0);
  fetch_a(0, kv);
  fetch_a(1, kvn);
  stash(0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int cur = c & 1;
    if (c + 1 < nch) fetch(c + 1);
    fetch_a(c + 2, kvn2);
#pragma unroll
    for (int q = 0; q < 4; ++q) kv[q] *= isw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* brow = BL + (cur * IQR_KC + 4 * q + lg) * BS + li;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(kv[q], brow[16 * t], acc[t], 0, 0, 0);
    }
    if (c + 1 < nch) stash(cur ^ 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) { kv[q] = kvn[q]; kvn[q] = kvn2[q]; }
    __syncthreads();
  }
  const int xo = 16 * wv;   // this wave's rows of xs_s / ys2_s
  // the staged chunks are no longer needed: the exp table takes their place
  double* TAB = BL;
  for (int t = tid; t < VB_EXP_TAB_N; t += IQR_THREADS) TAB[t] = c_exp2_tab[t];
  // ... and the ell-scaled, centred importance points XA[d][a] (each is needed by 4 lanes of every wave)
  double* XA = BL + VB_EXP_TAB_N;
  const double* xa = a.Xa + (a.per_s ? (size_t)s * a.Na * D : 0);
  const bool xa_lds = (size_t)(VB_EXP_TAB_N + D * Nap) <= (size_t)2 * IQR_KC * BS;
  if (xa_lds)
    for (int e = tid; e < D * Nap; e += IQR_THREADS) {
      const int d = e / Nap, aa_ = e - d * Nap;
      XA[e] = aa_ < a.Na ? xa[aa_ + (size_t)a.Na * d] * iell[d] - mu[d] : 0.0;
    }
  __syncthreads();
  // epilogue: lane (li, lg) holds C'[i = lg + 4r][a = 16t + li]
  const double u = 0.6745;
  const double sgn = a.lchol[s] ? -1.0 : 1.0;
  double zz[NT][4];
  double mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int aa_ = 16 * t + li;
    const bool av = aa_ < a.Na;
    double c4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int d = 0; d < D; ++d) {
      const double xv = xa_lds ? XA[d * Nap + aa_] : (av ? xa[aa_ + (size_t)a.Na * d] * iell[d] - mu[d] : 0.0);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const double tt = xs_s[xo + lg + 4 * r][d] - xv; c4[r] = fma(tt, tt, c4[r]); }
    }
    const double fa = av ? a.fs2a[(size_t)s * Nap + aa_] : 0.0;
    const double lw = av ? (a.lnw ? a.lnw[(size_t)s * Nap + aa_] : 0.0) : -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = lg + 4 * r;
      const double ka = sf2 * vb_exp_tab<0>(-0.5 * c4[r], TAB);
      const double C = ka + sgn * acc[t][r];
      const double tau2 = C * C / ys2_s[xo + i];
      const double sp = sqrt(fmax(fa - tau2, 0.0));
      const double z = av ? lw + (u * sp + log1p(-vb_exp_tab<0>(-2.0 * u * sp, TAB))) : -INFINITY;
      zz[t][r] = z;
      mx[r] = fmax(mx[r], z);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double m = mx[r];
    m = fmax(m, __shfl_xor(m, 1, 64)); m = fmax(m, __shfl_xor(m, 2, 64));
    m = fmax(m, __shfl_xor(m, 4, 64)); m = fmax(m, __shfl_xor(m, 8, 64));
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) sum += vb_exp_tab<0>(zz[t][r] - m, TAB);
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 8, 64);
    const int gi = i0 + lg + 4 * r;
    // every term -inf: MATLAB's -inf - -inf = NaN propagates (the table exp itself swallows NaN)
    if (li == 0 && gi < a.Nstar) a.acqs[gi + (size_t)a.Nstar * s] = (m == -INFINITY) ? NAN : log(sum) + m;
  }
}

// acq = M + log(sum(exp(acq_s - M))/Ns) over hyper-samples (:104-107), then the log-flag variance regulariser and the
// clamp of acq/acqwrapper_vbmc.m:35-46; also fbar / vtot (:21-29)
__global__ void __launch_bounds__(256) k_iqr_final(int Nstar, int S, int reg, double TolVar, const double* __restrict__ acqs,
                                                   const double* __restrict__ fmu, const double* __restrict__ fs2,
                                                   double* __restrict__ acq, double* __restrict__ fbar_o, double* __restrict__ vtot_o) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nstar) return;
  double fbar = 0.0, vbar = 0.0;
  for (int s = 0; s < S; ++s) { fbar += fmu[i + (size_t)Nstar * s]; vbar += fs2[i + (size_t)Nstar * s]; }
  fbar /= S; vbar /= S;
  double vf = 0.0;
  if (S > 1) {
    for (int s = 0; s < S; ++s) { const double d = fmu[i + (size_t)Nstar * s] - fbar; vf += d * d; }
    vf /= (S - 1);
  }
  const double vtot = vf + vbar;
  double v;
  if (S > 1) {
    double M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmax(M, acqs[i + (size_t)Nstar * s]);
    double sum = 0.0;
    for (int s = 0; s < S; ++s) sum += exp(acqs[i + (size_t)Nstar * s] - M);
    v = M + log(sum / S);
  } else v = acqs[i];
  if (reg && vtot < TolVar) v = v + TolVar / vtot - 1.0;
  v = fmax(v, -1.7976931348623157e308);
  acq[i] = v;
  if (fbar_o) fbar_o[i] = fbar;
  if (vtot_o) vtot_o[i] = vtot;
}
