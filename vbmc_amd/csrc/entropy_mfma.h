// k_entropy_mfma: the Monte-Carlo entropy + reparameterisation gradient (ent/entmc_vbmc.m:49-104)
// organised around v_mfma_f64_16x16x4_f64, in the shape of a flash-attention tile:
//
//   S-step   E^T[k][i] = b_k . a_i - shift_i       (16 components x 16 samples, inner dim D+2)
//   exp      n_ik = exp(E_ik)                       (the only transcendental; 4 per lane per tile)
//   PV-step  Y[i][c] = sum_k n_ik V[k][c]           (16 samples x 16 columns: q', A', B'_1..D)
//
// with  a_i = [u'_i, |u'_i|^2, 1],  u'_i = eps_i * sigma_j  (coordinates centred on the sample's own
// component j and scaled by 1/lambda),  b_k = [m'_k/sigma_k^2, -1/(2 sigma_k^2), -D ln sigma_k -
// |m'_k|^2/(2 sigma_k^2)],  m'_k = (mu_k - mu_j)/lambda,  shift_i = exponent of the own component, and
// V[k] = w_k [1, 1/sigma_k^2, m'_k/sigma_k^2].  Computing the TRANSPOSED S product puts sample
// i = lane&15 and components 4r + (lane>>4) in accumulator register r -- exactly the A-operand
// layout of the PV MFMA, so n never leaves registers and the weight gradient
// W_l = sum_i n_il / q_i (entmc_vbmc.m:100) is one lane-local FMA per pair.
//
// One wave (= one 64-thread workgroup) per (sample chunk, source component j, restart r).  A tile is
// 16 base samples, processed twice (+eps, -eps: antithetic, entmc_vbmc.m:53-54).  All mixture-side
// MFMA operands are built once per wave and stay in registers; the only LDS traffic is the 16 x D eps
// tile (read in two layouts) and 32 scalars.  Partials have the same layout as k_entropy
// (sum log q | G[D] | SG | LG[D] | W[K]) and are reduced by k_finalize in a fixed order.
#pragma once
#include "elbo_kernels.h"

typedef double mf4 __attribute__((ext_vector_type(4)));

template <int QS, int KTM, bool GRAD>
__global__ void __launch_bounds__(WAVE) k_entropy_mfma(EntArgs a) {
  constexpr int DP = 4 * QS;               // padded eps row length
  constexpr int NPV = (4 * QS + 15) / 16;  // 16-column blocks of the PV output (D + 2 columns)
  __shared__ double Et[16 * DP];           // eps tile [i][d]
  __shared__ double E2[16];                // |eps_i|^2
  __shared__ double RQ[16];                // 1/q'_i
  const int lane = threadIdx.x;
  const int li = lane & 15, lg = lane >> 4;
  const int c = blockIdx.x, j = blockIdx.y, r = blockIdx.z;
  const int D = a.D, K = a.K;
  const int KT = (K + 15) >> 4;
  const int PSg = D + ENTP_EXTRA;
  const double* gp = a.entp + (size_t)r * K * PSg;  // [k][m_1..m_D, h, cK, w, wi]
  const double* pj = gp + (size_t)j * PSg;
  VpLayout L{D, K};
  const double sigj = a.vpd[(size_t)r * L.stride() + L.sigma() + j];
  const double cKj = pj[D + 1];

  // ---- mixture-side operand fragments (registers, built once)
  double SA[KTM][QS];          // S-step "A" operand: comp 16kt + li, inner c = 4q + lg
  double VB[KTM][4][NPV];      // PV "B" operand: comp 16kt + 4r + lg, column 16pv + li      (GRAD)
  double WF[KTM][4];           // w_k for comp 16kt + 4r + lg                                  (!GRAD)
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt) {
    const int k = 16 * kt + li;
    const bool kv = (kt < KT) && (k < K);
    const double* pk = gp + (size_t)(kv ? k : 0) * PSg;
    double h = pk[D];
    double m2 = 0.0;
    for (int d = 0; d < D; ++d) { double t = pk[d] - pj[d]; m2 = fma(t, t, m2); }
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int cc = 4 * q + lg;
      double v;
      if (!kv) v = (cc == D + 1) ? -1.0e300 : 0.0;          // padded component: exp -> 0
      else if (cc < D) v = -2.0 * h * (pk[cc] - pj[cc]);     // m'_ck / sigma_k^2   (h = -1/(2 sigma^2))
      else if (cc == D) v = h;
      else if (cc == D + 1) v = fma(h, m2, pk[D + 1]);       // -D ln sigma_k - |m'_k|^2/(2 sigma_k^2)
      else v = 0.0;
      SA[kt][q] = v;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int k2 = 16 * kt + 4 * rr + lg;
      const bool kv2 = (kt < KT) && (k2 < K);
      const double* p2 = gp + (size_t)(kv2 ? k2 : 0) * PSg;
      if (GRAD) {
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv) {
          const int col = 16 * pv + li;
          double v = 0.0;
          if (kv2) {
            if (col == 0) v = p2[D + 2];                                   // w_k            -> q'
            else if (col == 1) v = p2[D + 3];                              // w_k/sigma_k^2  -> A'
            else if (col < 2 + D) v = p2[D + 3] * (p2[col - 2] - pj[col - 2]);  // -> B'_d
          }
          VB[kt][rr][pv] = v;
        }
      } else {
        WF[kt][rr] = kv2 ? p2[D + 2] : 0.0;
      }
    }
  }

  double accH = 0.0, accG[NPV], accLG[NPV];
  double Wacc[KTM][4];
#pragma unroll
  for (int pv = 0; pv < NPV; ++pv) { accG[pv] = 0.0; accLG[pv] = 0.0; }
#pragma unroll
  for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Wacc[kt][rr] = 0.0;

  const int ntile = (a.Mh + 15) >> 4;
  const int t0 = c * a.tiles_per_chunk;
  const int t1 = min(t0 + a.tiles_per_chunk, ntile);
  const double* epsr = a.eps ? a.eps + (size_t)r * a.eps_stride_r + (size_t)j * a.Mh * D : nullptr;

  for (int tile = t0; tile < t1; ++tile) {
    const int b0 = tile * 16;
    // ---- stage the 16 x D eps tile in LDS (zeros for padded dims / samples beyond Mh)
    __syncthreads();
    if (epsr) {
      for (int idx = lane; idx < 16 * DP; idx += WAVE) {
        const int i = idx / DP, d = idx - i * DP;
        Et[idx] = (d < D && b0 + i < a.Mh) ? epsr[(size_t)(b0 + i) * D + d] : 0.0;
      }
    } else {
      const bool bv = b0 + li < a.Mh;
#pragma unroll
      for (int q = lg; q < QS; q += 4) {
        double z4[4] = {0.0, 0.0, 0.0, 0.0};
        if (q < (D + 3) / 4) vb_normal4(a.seed, (unsigned)(b0 + li), (unsigned)j, (unsigned)r, (unsigned)q, z4);
#pragma unroll
        for (int t = 0; t < 4; ++t) Et[li * DP + 4 * q + t] = (bv && 4 * q + t < D) ? z4[t] : 0.0;
      }
    }
    __syncthreads();
    // sample-side fragments for +eps: lane (li, lg) holds a_i[c = 4q + lg]
    double ev[QS];
    double e2 = 0.0;
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      ev[q] = Et[li * DP + 4 * q + lg];   // zero beyond D
      e2 = fma(ev[q], ev[q], e2);
    }
    e2 += __shfl_xor(e2, 16, 64);
    e2 += __shfl_xor(e2, 32, 64);
    if (lg == 0) E2[li] = e2;
    const double shift = cKj - 0.5 * e2;        // exponent of the sample's own component
    const double u2 = sigj * sigj * e2;         // |u'_i|^2
    __syncthreads();

#pragma unroll 1
    for (int sg = 0; sg < 2; ++sg) {
      const double sgn = sg ? -1.0 : 1.0;
      double sf[QS];
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        const int cc = 4 * q + lg;
        sf[q] = (cc < D) ? sgn * ev[q] * sigj : ((cc == D) ? u2 : ((cc == D + 1) ? 1.0 : 0.0));
      }
      // ---- S-step + exp
      double n[KTM][4];
#pragma unroll
      for (int kt = 0; kt < KTM; ++kt) {
        if (kt < KT) {
          mf4 acc = {-shift, -shift, -shift, -shift};
#pragma unroll
          for (int q = 0; q < QS; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SA[kt][q], sf[q], acc, 0, 0, 0);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) n[kt][rr] = (16 * kt + 4 * rr < K) ? vb_exp(acc[rr]) : 0.0;
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) n[kt][rr] = 0.0;
        }
      }
      if (GRAD) {
        // ---- PV-step: Y[i][col]; lane (col = li, lg) register rr <-> sample lg + 4 rr
        mf4 Y[NPV];
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv) Y[pv] = (mf4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt) {
          if (kt < KT) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              if (16 * kt + 4 * rr < K) {
#pragma unroll
                for (int pv = 0; pv < NPV; ++pv)
                  Y[pv] = __builtin_amdgcn_mfma_f64_16x16x4f64(n[kt][rr], VB[kt][rr][pv], Y[pv], 0, 0, 0);
              }
            }
          }
        }
        // ---- epilogue in the PV output layout
        const int base = lane & 48;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int i = lg + 4 * rr;
          const bool valid = b0 + i < a.Mh;
          const double qv = __shfl(Y[0][rr], base, 64);        // q'_i  (column 0)
          const double Av = __shfl(Y[0][rr], base | 1, 64);    // A'_i  (column 1)
          const double rq = valid ? 1.0 / qv : 0.0;
          if (li == 0) {
            RQ[i] = rq;
            if (valid) accH += (cKj - 0.5 * E2[i]) + log(qv);  // log q - log nf  (entmc_vbmc.m:67)
          }
#pragma unroll
          for (int pv = 0; pv < NPV; ++pv) {
            const int d = 16 * pv + li - 2;
            if (d >= 0 && d < D) {
              const double e = sgn * Et[i * DP + d];
              const double gd = (e * sigj * Av - Y[pv][rr]) * rq;  // lambda_d lsum_d / q  (:77-79)
              accG[pv] += gd;                                      // -> mu_grad (:82)
              accLG[pv] = fma(e, gd, accLG[pv]);                   // -> sigma/lambda grads (:87-93)
            }
          }
        }
        __syncthreads();
        const double rqs = RQ[li];
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) Wacc[kt][rr] = fma(n[kt][rr], rqs, Wacc[kt][rr]);  // (:100)
        __syncthreads();
      } else {
        double qp = 0.0;
#pragma unroll
        for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) qp = fma(WF[kt][rr], n[kt][rr], qp);
        qp += __shfl_xor(qp, 16, 64);
        qp += __shfl_xor(qp, 32, 64);
        if (lg == 0 && b0 + li < a.Mh) accH += shift + log(qp);
      }
    }
  }

  // ---- fixed-order reductions and the partial record
  double* o = a.part + (((size_t)r * K + j) * a.C + c) * a.ncol;
  accH = wave_sum(accH);
  if (lane == 0) o[0] = accH;
  if (GRAD) {
    double sgsum = 0.0;
#pragma unroll
    for (int pv = 0; pv < NPV; ++pv) {
      double g = accG[pv], lgd = accLG[pv];
      g += __shfl_xor(g, 16, 64); g += __shfl_xor(g, 32, 64);
      lgd += __shfl_xor(lgd, 16, 64); lgd += __shfl_xor(lgd, 32, 64);
      const int d = 16 * pv + li - 2;
      const bool dv = d >= 0 && d < D;
      if (dv && lg == 0) { o[1 + d] = g; o[2 + D + d] = lgd; }
      sgsum += (dv && lg == 0) ? lgd : 0.0;
    }
    sgsum = wave_sum(sgsum);            // SG = sum_d LG_d  (entmc_vbmc.m:87)
    if (lane == 0) o[1 + D] = sgsum;
#pragma unroll
    for (int kt = 0; kt < KTM; ++kt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        double wv = Wacc[kt][rr];
        wv += __shfl_xor(wv, 1, 64); wv += __shfl_xor(wv, 2, 64);
        wv += __shfl_xor(wv, 4, 64); wv += __shfl_xor(wv, 8, 64);
        const int k = 16 * kt + 4 * rr + lg;
        if (li == 0 && k < K) o[2 + 2 * D + k] = wv;
      }
  }
}
