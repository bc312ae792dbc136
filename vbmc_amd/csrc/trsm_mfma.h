// Wave-level blocked triangular solves with 16 right-hand sides on v_mfma_f64_16x16x4_f64.
//
// R is the upper Cholesky factor of gplite (MATLAB chol convention, R'R = A), N x N column-major in
// global memory; the N x 16 slab of right-hand sides lives in LDS as V[row * TR_VS + col] (rows padded
// to a multiple of 16 with zeros).  Per 16-row block the trailing update
//     rhs_b -= R[0:b0, b]' * V[0:b0, :]      (forward, R' V = Z)
//     rhs_b -= R[b, e0:N] * X[e0:N, :]       (backward, R X = V)
// is a (16 x b0) x (b0 x 16) product accumulated by one MFMA per 4 rows of the inner dimension; the
// 16 x 16 diagonal block is then solved by substitution with the block's right-hand sides held in
// the MFMA accumulator layout (lane (li = column, lg): rows lg + 4 r), the freshly solved row being
// broadcast inside the 4 lanes of a column each step.  Used by gplite_pred's V = L' \ (sW .* Ks)
// (gplite/gplite_pred.m:99), the BQ variance (misc/gplogjoint.m:277,318), alpha = L \ (L' \ (y-m))
// (gplite/private/gplite_core.m:102) and the rank-1 update (gplite/gplite_post.m:227-229).
#pragma once
#include <hip/hip_runtime.h>

typedef double tmf4 __attribute__((ext_vector_type(4)));
#define TR_VS 17  // LDS row stride of the right-hand-side slab (16 columns + 1 pad: conflict-free column fills)

// Rd: 16 x 16 diagonal block (Rd[ii * 16 + jj] = R[b0+ii][b0+jj], identity beyond N), IDG: 1 / diag.
__device__ __forceinline__ void trsm_load_diag(int N, const double* __restrict__ Rm, int b0, int lane,
                                               double* __restrict__ Rd, double* __restrict__ IDG) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = lane + 64 * t, ii = e >> 4, jj = e & 15;
    double v = (ii == jj) ? 1.0 : 0.0;
    if (b0 + ii < N && b0 + jj < N && ii <= jj) v = Rm[(size_t)(b0 + jj) * N + b0 + ii];
    Rd[e] = v;
  }
  __syncthreads();
  if (lane < 16) IDG[lane] = 1.0 / Rd[lane * 17];
  __syncthreads();
}

// forward substitution R' V = Z for the slab in LDS (in place)
__device__ __forceinline__ void trsm_fwd_wave(int N, const double* __restrict__ Rm, double* __restrict__ V,
                                              double* __restrict__ Rd, double* __restrict__ IDG, int lane) {
  const int li = lane & 15, lg = lane >> 4;
  const int nblk = (N + 15) >> 4;
  for (int bi = 0; bi < nblk; ++bi) {
    const int b0 = bi << 4;
    tmf4 acc = {0.0, 0.0, 0.0, 0.0};
    const bool cv = b0 + li < N;
    const double* col = Rm + (size_t)(cv ? b0 + li : 0) * N;  // column b0+li of R (rows j contiguous)
    for (int j0 = 0; j0 < b0; j0 += 4) {
      const double a = cv ? col[j0 + lg] : 0.0;               // A[i = li][k = lg] = R[j0+lg][b0+li]
      const double b = V[(j0 + lg) * TR_VS + li];                // B[k = lg][c = li]
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    trsm_load_diag(N, Rm, b0, lane, Rd, IDG);
    double rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rhs[r] = V[(b0 + lg + 4 * r) * TR_VS + li] - acc[r];
    // v_ii = (rhs_ii - sum_{jj<ii} R[jj][ii] v_jj) / R[ii][ii], right-looking inside the block
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
      const double mine = rhs[ii >> 2] * IDG[ii];
      const double vi = __shfl(mine, li | ((ii & 3) << 4), 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lg + 4 * r;
        if (row > ii) rhs[r] = fma(-Rd[ii * 16 + row], vi, rhs[r]);
        else if (row == ii) rhs[r] = vi;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) V[(b0 + lg + 4 * r) * TR_VS + li] = rhs[r];
    __syncthreads();
  }
}

// backward substitution R X = V for the slab in LDS (in place)
__device__ __forceinline__ void trsm_bwd_wave(int N, const double* __restrict__ Rm, double* __restrict__ V,
                                              double* __restrict__ Rd, double* __restrict__ IDG, int lane) {
  const int li = lane & 15, lg = lane >> 4;
  const int nblk = (N + 15) >> 4;
  const int Np = nblk << 4;
  for (int bi = nblk - 1; bi >= 0; --bi) {
    const int b0 = bi << 4;
    tmf4 acc = {0.0, 0.0, 0.0, 0.0};
    const bool rv = b0 + li < N;
    for (int j0 = b0 + 16; j0 < Np; j0 += 4) {
      const int j = j0 + lg;
      const double a = (rv && j < N) ? Rm[(size_t)j * N + b0 + li] : 0.0;  // A[i = li][k = lg] = R[b0+li][j]
      const double b = V[j * TR_VS + li];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    trsm_load_diag(N, Rm, b0, lane, Rd, IDG);
    double rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rhs[r] = V[(b0 + lg + 4 * r) * TR_VS + li] - acc[r];
    // x_ii = (rhs_ii - sum_{jj>ii} R[ii][jj] x_jj) / R[ii][ii], from the bottom row up
#pragma unroll
    for (int ii = 15; ii >= 0; --ii) {
      const double mine = rhs[ii >> 2] * IDG[ii];
      const double xi = __shfl(mine, li | ((ii & 3) << 4), 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lg + 4 * r;
        if (row < ii) rhs[r] = fma(-Rd[row * 16 + ii], xi, rhs[r]);
        else if (row == ii) rhs[r] = xi;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) V[(b0 + lg + 4 * r) * TR_VS + li] = rhs[r];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Standalone kernels: Z is laid out [r][s][k][N] (column k of the right-hand sides contiguous);
// one wave per (16 columns, hyper-sample s, restart r).  lchol[s] == 0 samples are skipped.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void trsm_slab_load(int N, int K, int k0, const double* __restrict__ Zs, double* __restrict__ V, int lane) {
  const int Np = ((N + 15) >> 4) << 4;
  for (int c = 0; c < 16; ++c) {
    const bool cv = k0 + c < K;
    for (int i = lane; i < Np; i += 64) V[i * TR_VS + c] = (cv && i < N) ? Zs[(size_t)(k0 + c) * N + i] : 0.0;
  }
  __syncthreads();
}
__device__ __forceinline__ void trsm_slab_store(int N, int K, int k0, double* __restrict__ Zs, const double* __restrict__ V, int lane) {
  for (int c = 0; c < 16; ++c) {
    if (k0 + c >= K) break;
    for (int i = lane; i < N; i += 64) Zs[(size_t)(k0 + c) * N + i] = V[i * TR_VS + c];
  }
}
#define TRSM_LDS_BYTES(N) ((size_t)(((((N) + 15) >> 4) << 4) * TR_VS + 256 + 16) * sizeof(double))

__global__ void __launch_bounds__(64) k_trsm_fwd(int N, int K, int S, const double* __restrict__ Lall,
                                                 const unsigned char* __restrict__ lchol, double* __restrict__ Z) {
  extern __shared__ double lds[];
  const int cb = blockIdx.x, s = blockIdx.y, r = blockIdx.z, lane = threadIdx.x;
  if (!lchol[s]) return;
  const int Np = ((N + 15) >> 4) << 4;
  double* V = lds;
  double* Rd = V + (size_t)Np * TR_VS;
  double* IDG = Rd + 256;
  double* Zs = Z + ((size_t)r * S + s) * (size_t)K * N;
  trsm_slab_load(N, K, cb * 16, Zs, V, lane);
  trsm_fwd_wave(N, Lall + (size_t)s * N * N, V, Rd, IDG, lane);
  trsm_slab_store(N, K, cb * 16, Zs, V, lane);
}

__global__ void __launch_bounds__(64) k_trsm_bwd(int N, int K, int S, const double* __restrict__ Lall,
                                                 const unsigned char* __restrict__ lchol,
                                                 const double* __restrict__ Vin, double* __restrict__ Xo) {
  extern __shared__ double lds[];
  const int cb = blockIdx.x, s = blockIdx.y, r = blockIdx.z, lane = threadIdx.x;
  if (!lchol[s]) return;
  const int Np = ((N + 15) >> 4) << 4;
  double* V = lds;
  double* Rd = V + (size_t)Np * TR_VS;
  double* IDG = Rd + 256;
  trsm_slab_load(N, K, cb * 16, Vin + ((size_t)r * S + s) * (size_t)K * N, V, lane);
  trsm_bwd_wave(N, Lall + (size_t)s * N * N, V, Rd, IDG, lane);
  trsm_slab_store(N, K, cb * 16, Xo + ((size_t)r * S + s) * (size_t)K * N, V, lane);
}
