// The body of the VALU expected-log-joint kernel (misc/gplogjoint.m:162-271), shared by k_logjoint (elbo_kernels.h: a workgroup of
// one or LJ_MAXW waves per group of four components and hyper-sample) and by the log-joint ROLE of the MFMA entropy kernel
// (entropy_mfma.h, CO = true: single-wave workgroups of the SAME launch, so that a single chain's two independent kernels run side
// by side instead of one after the other).
#pragma once
#include "device_math.h"
#include "elbo_types.h"

// One wave = four (component k, hyper-sample s) cells: lane (kq = lane>>4, ni = lane&15) strides over the training points
// n = ni + 16 (wv + NW i) for component k = 4 kgroup + kq.  The per-dimension constants tau_d, log tau_d are computed once by the
// lane with ni = d (and d+16) and broadcast inside the 16-lane row; the 2D+1 sums are reduced over the 16 lanes of a row only
// (4 butterfly steps, each shuffle serving four cells).
// record layout [2D+2] = I_k, w_k*dmu[D], w_k*dsigma (no Jacobian), w_k*dlambda[D]
// (round 6: on the VALU -- quad permutes and the row (half-)mirror pair the same groups the xor butterfly 1, 2, 4, 8 does, every lane of a
// group already holding the group's sum: the same bits as the __shfl_xor form without its eight ds_bpermute round trips per value)
__device__ __forceinline__ double row16_sum(double v) {
  v = dpp_pair_sum<0xB1>(v);
  v = dpp_pair_sum<0x4E>(v);
  v = dpp_pair_sum<0x141>(v);
  return dpp_pair_sum<0x140>(v);
}

// The sums of one wave over its share of the training set -> pp[kq][NC] (LDS rows of this wave, NC = 2 DT + 2: I, M[DT], S, L[DT]).
// `tab_ready()` is called once, after the set-up arithmetic and before the first exponential: the point where the exp table in
// LDS must be complete (a workgroup barrier in k_logjoint, a wave-level fence in the role).
template <int DT, class TabReady>
__device__ __forceinline__ void lj_wave_sums(const ElboDims& dm, const double* __restrict__ v, const double* __restrict__ X,
                                             const double* __restrict__ al, const double* __restrict__ g,
                                             const double* __restrict__ delta2, const double* TAB, int kgroup, int wv, int NW,
                                             int want_grad, double* pp_wave, TabReady tab_ready) {
  constexpr int NC = 2 * DT + 2;
  const int lane = threadIdx.x & 63;
  const int ni = lane & 15, kq = lane >> 4, rowbase = lane & 48;
  const int D = dm.D, K = dm.K, N = dm.N;
  const int kk = 4 * kgroup + kq;
  const int k = kk < K ? kk : K - 1;
  VpLayout L{D, K};
  const double sig = v[L.sigma() + k];
  // lane ni owns dimensions d = ni and ni + 16
  double my_lam[2] = {0.0, 0.0}, my_mu[2] = {0.0, 0.0}, my_itau[2] = {0.0, 0.0};
  double my_logtau = 0.0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int d = ni + 16 * h;
    if (d < D && (h == 0 || DT > 16)) {
      my_lam[h] = v[L.lambda() + d];
      my_mu[h] = v[L.mu() + d + D * k];
      double tau = sqrt(sig * sig * my_lam[h] * my_lam[h] + g[d] + delta2[d]);  // :164
      my_logtau += log(tau);
      my_itau[h] = 1.0 / tau;
    }
  }
  const double sumlogtau = row16_sum(my_logtau);
  double mu[DT], itau[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    const int src = rowbase | (d & 15), h = d >> 4;
    mu[d] = __shfl(my_mu[h], src, 64);
    itau[d] = __shfl(my_itau[h], src, 64);     // zero for padded dimensions: they vanish below
  }
  const double lnnf = g[3 * D] - sumlogtau;  // ln_sf2 + sum_lnell - sum(log(tau_k))  :165
  tab_ready();
  // The three gradient sums of :207-250 are  dz_dmu = -1/tau_d * A_d,  dz_dlambda = sigma^2 lambda_d / tau_d^2 * Q_d  and
  // dz_dsigma = sigma sum_d lambda_d^2 / tau_d^2 * Q_d  with only TWO sums over the training set per dimension,
  // A_d = sum_n delta_d z alpha  and  Q_d = sum_n (delta_d^2 - 1) z alpha: the factors do not depend on n and are applied once,
  // after the row sums (round 3: half the FMAs of the loop and 3 DT fewer live registers)
  double accI = 0.0;
  double accA[DT], accQ[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { accA[d] = 0.0; accQ[d] = 0.0; }
  // Slabs are loaded TWO ahead of their arithmetic (a single chain is bound by this kernel's memory latency, not its flops: a slab's ~100
  // instructions of one wave are shorter than a load's round trip).  Round 6: through buffer descriptors -- X and alpha as byte ranges, one
  // 32-bit offset per lane and a scalar offset per dimension: no address arithmetic, no bounds test and no branch per load (a dimension or
  // a point beyond the range reads 0; a point beyond N inside X's range reads a neighbour's value against alpha = 0) -- and the loop
  // unrolled by its three slab buffers, so that no register is moved: the loop body was 84 fp64 operations among 54 moves, 11 address
  // computations and 11 exec-masked branches.
  const __amdgpu_buffer_rsrc_t Xr = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((size_t)N * D * 8), 0x00020000);
  const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void*)al, 0, N * 8, 0x00020000);
  const int step = 16 * NW;
  int n = ni + 16 * wv;
  auto load = [&](const int nidx, double (&x)[DT], double& aa) {
    const int off = min(nidx, N) * 8;      // (a point beyond N: alpha's range ends at N -- it reads 0 -- and X yields some finite value)
#pragma unroll
    for (int d = 0; d < DT; ++d) x[d] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Xr, off, d * N * 8, 0));
    aa = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Ar, off, 0, 0));
  };
  auto trip = [&](const double (&x)[DT], const double aa) {
    double dl[DT];
    double a2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      dl[d] = (mu[d] - x[d]) * itau[d];  // delta_k :167  (the reference's form: X is not centred here, mu_d / tau_d - x_d / tau_d would cancel)
      a2 = fma(dl[d], dl[d], a2);
    }
    const double z = vb_exp_tab(lnnf - 0.5 * a2, TAB);  // z_k :168
    const double za = z * aa;
    accI += za;
    if (want_grad) {
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        accA[d] = fma(dl[d], za, accA[d]);                        // :207-208 without its factor
        accQ[d] = fma(fma(dl[d], dl[d], -1.0), za, accQ[d]);      // :228, :249-250 without theirs
      }
    }
  };
  double x0[DT], x1[DT], x2[DT], a0, a1, a2v;
  load(n, x0, a0);
  load(n + step, x1, a1);
  while (n < N) {
    load(n + 2 * step, x2, a2v);
    trip(x0, a0);
    n += step;
    if (n >= N) break;
    load(n + 2 * step, x0, a0);
    trip(x1, a1);
    n += step;
    if (n >= N) break;
    load(n + 2 * step, x1, a1);
    trip(x2, a2v);
    n += step;
  }
  accI = row16_sum(accI);
  if (want_grad) {
#pragma unroll
    for (int d = 0; d < DT; ++d) { accA[d] = row16_sum(accA[d]); accQ[d] = row16_sum(accQ[d]); }
  }
  double accS = 0.0;
  double* pp = pp_wave + kq * NC;
  if (want_grad) {
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const double lam_d = __shfl(my_lam[d >> 4], rowbase | (d & 15), 64);
      const double li = lam_d * itau[d], si = sig * itau[d];
      accS = fma(li * li, accQ[d], accS);                                  // dz_dsigma factor   (:228)
      if (ni == 0) {
        pp[1 + d] = -itau[d] * accA[d];                                    // dz_dmu factor      (:207)
        pp[2 + DT + d] = si * si * lam_d * accQ[d];                        // dz_dlambda factor  (:249)
      }
    }
  }
  if (ni == 0) {
    pp[0] = accI;
    if (want_grad) pp[1 + DT] = sig * accS;
  }
}

// Output record of component k = 4 kgroup + kq from `nrows` rows of wave sums (row stride `rstride` doubles, added in row order),
// one lane per output column.  with_const: the closed-form mean-function terms of :169-174 / :208-210 / :229-231 / :250-252 that do
// not depend on the training set are added (exactly one of the records that are later summed carries them).
template <int DT, int ROWS = 2 * DT + 2>
__device__ __forceinline__ void lj_write_record(const ElboDims& dm, const double* __restrict__ v, const double* __restrict__ g,
                                                const double* __restrict__ delta2, const double* pp0, int nrows, int rstride,
                                                int kgroup, int want_grad, bool with_const, double* __restrict__ o_s /* [K][2D+2] of (r, s) */) {
  constexpr int NC = 2 * DT + 2;
  const int lane = threadIdx.x & 63;
  const int ni = lane & 15, kq = lane >> 4;
  const int D = dm.D, K = dm.K;
  const int k = 4 * kgroup + kq;
  if (k >= K) return;
  VpLayout L{D, K};
  const double sig = v[L.sigma() + k];
  const double wk = v[L.w() + k];
  double* o = o_s + (size_t)k * (2 * D + 2);
  const int ncol = want_grad ? 2 * D + 2 : 1;
  for (int c = ni; c < ncol; c += 16) {
    // column c of the output record <-> slot of the row (padded to DT)
    const int d = (c >= 1 && c <= D) ? c - 1 : (c >= D + 2 ? c - D - 2 : 0);
    const int slot = c == 0 ? 0 : (c <= D ? c : (c == D + 1 ? 1 + DT : 2 + DT + d));
    double acc = pp0[kq * ROWS + slot];
    for (int w2 = 1; w2 < nrows; ++w2) acc += pp0[(size_t)w2 * rstride + kq * ROWS + slot];
    if (!with_const) {
      o[c] = c == 0 ? acc : wk * acc;
      continue;
    }
    // mean-function terms; iom2 = 0 and xm = 0 for meanfun 0/1 so they vanish  :169-174
    if (c == 0 || c == D + 1) {
      double nu = 0.0, sl2 = 0.0;
      for (int e = 0; e < D; ++e) {
        double xm = g[D + e], iom2 = g[2 * D + e];
        double lam_e = v[L.lambda() + e], mu_e = v[L.mu() + e + D * k];
        nu += iom2 * (mu_e * mu_e + sig * sig * lam_e * lam_e - 2.0 * mu_e * xm + xm * xm + delta2[e]);
        sl2 += iom2 * lam_e * lam_e;
      }
      o[c] = c == 0 ? acc + g[3 * D + 1] + (-0.5 * nu)          // I_k
                    : wk * acc - wk * sig * sl2;                 // :229-231
    } else {
      const double xm = g[D + d], iom2 = g[2 * D + d];
      const double lam_d = v[L.lambda() + d], mu_d = v[L.mu() + d + D * k];
      o[c] = c <= D ? wk * acc - wk * iom2 * (mu_d - xm)                   // :208-210
                    : wk * acc - wk * sig * sig * iom2 * lam_d;            // :250-252
    }
  }
}

// The log-joint role of a single-wave workgroup (entropy_mfma.h, CO kernels): workgroup `w` of the role handles the cell group
// kgroup = w mod G4, split sp = (w / G4) mod nsplit of the training set, hyper-sample s = w / (G4 nsplit), and writes ITS OWN record
// lj[r][s nsplit + sp][k][2D+2] -- the reduction over hyper-samples (k_reduce_both with S nsplit "samples") adds the splits, the
// record of split 0 carries the closed-form terms.  lds: >= 256 + 4 (2 DT + 2) doubles (exp table, then this wave's rows).
// lj_role_wave: the role for ONE wave -- cell-group index w of restart r, the 256-entry exp table TAB and the wave's own 4 (2 DT + 2)
// row doubles pp given by the caller.  share_tab: the table belongs to a workgroup of several role waves, each of which has filled its
// quarter of it -- the ordering point before the first exponential is then a workgroup barrier (every wave of the workgroup reaches it:
// callers do not return early), otherwise the wave-level fence.
template <int DT, bool SHARE_TAB = false>
__device__ __forceinline__ void lj_role_wave(const LjCo& a, const double* __restrict__ vpd, int w, int r, double* TAB, double* pp) {
  const ElboDims& dm = a.dm;
  const int D = dm.D, K = dm.K, G4 = (K + 3) / 4;
  const bool live = w < a.nwg;
  if (!live) w = 0;            // (SHARE_TAB: an idle wave still walks to the barrier; it writes nothing)
  const int kgroup = w % G4, t = w / G4, sp = t % a.nsplit, s = t / a.nsplit;
  VpLayout L{D, K};
  const double* v = vpd + (size_t)r * L.stride();
  const double* g = a.gpc + (size_t)s * GPC_STRIDE(D);
  auto fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto ready = [&] { if (SHARE_TAB) __syncthreads(); else fence(); };
  if (SHARE_TAB && !live) { __syncthreads(); return; }
  lj_wave_sums<DT>(dm, v, a.X, a.alpha + (size_t)s * dm.N, g, a.delta2, TAB, kgroup, sp, a.nsplit, a.want_grad, pp, ready);
  fence();
  double* o_s = a.lj + ((size_t)r * dm.S * a.nsplit + (size_t)s * a.nsplit + sp) * K * (2 * D + 2);
  lj_write_record<DT>(dm, v, g, a.delta2, pp, 1, 0, kgroup, a.want_grad, sp == 0, o_s);
}

template <int DT>
__device__ __forceinline__ void lj_co_role(const LjCo& a, const double* __restrict__ vpd, double* lds) {
  const int w = blockIdx.y * gridDim.x + blockIdx.x, r = blockIdx.z, lane = threadIdx.x & 63;
  if (w >= a.nwg) return;
  double* TAB = lds;
  double* pp = lds + VB_EXP_TAB_N;
  for (int i = lane; i < VB_EXP_TAB_N; i += 64) TAB[i] = c_exp2_tab[i];
  lj_role_wave<DT>(a, vpd, w, r, TAB, pp);
}

// The role inside k_entropy_lane (entropy_lane.h): everything the cell group reads is in LDS -- the restart's vp block VL, the
// per-hyper-sample constants GL (all S), delta^2, the training inputs XL ([D][NP], NP = N rounded up to 64, zero-padded) staged by the
// workgroup, and this hyper-sample's alpha in the wave's own block AL (zero-padded likewise) -- so the walk over the training set waits
// for LDS, not for memory.  One split (the wave walks the whole training set), records per hyper-sample.
// Lane layout: cell c = lane & 3 (component 4 kgroup + c), point lane pl = lane >> 2: the sums over a cell's sixteen point lanes are
// two row rotations (DPP) and the two permlane swaps, and the swaps carry TWO values each (cell4_sum_tree): 57 VALU instructions for
// the thirteen sums of D = 6 where the four-step row butterfly per value took 266.  The loop body is cut to what it needs (128 -> ~55
// VALU instructions per slab of 16 points): the standardised distance as ONE fused multiply-add per dimension (mu / tau carried
// beside 1 / tau), no select and no bound on the padded dimensions and points (their 1 / tau, their alpha are zero), four slabs per trip
// as independent chains, the exponential by the entropy kernel's 1024-entry table with the argument scaled in the multiply-add that
// forms it; the gradient factors (:207, :228, :249) are applied to the per-lane partial sums -- they are linear -- so the tree sums the
// record's columns directly.    scr: LJ_LANE_SCR(DT) doubles of the wave's own.
#define LJ_LANE_SCR(DT_) (4 * (3 * (DT_) + 2))
// sums over the lanes of equal (lane & 3) of NV values: on return register m of `out` holds, in the lanes of row q = lane >> 4, the total of
// value 4 m + {0, 2, 1, 3}[q] for cell lane & 3 (every point lane of the row holds it)
template <int NV>
__device__ __forceinline__ void cell4_sum_tree(const double (&v)[NV], double (&out)[(NV + 3) / 4]) {
  constexpr int N4 = (NV + 3) / 4;
  double h[2 * N4];
#pragma unroll
  for (int i = 0; i < 2 * N4; ++i) {
    const double x = 2 * i < NV ? v[2 * i < NV ? 2 * i : 0] : 0.0, y = 2 * i + 1 < NV ? v[2 * i + 1 < NV ? 2 * i + 1 : 0] : 0.0;
    h[i] = swap_sum32(x, y);
  }
#pragma unroll
  for (int m = 0; m < N4; ++m) {
    double t = swap_sum16(h[2 * m], h[2 * m + 1]);
    t = dpp_pair_sum<0x128>(t);        // row_ror:8
    out[m] = dpp_pair_sum<0x124>(t);   // row_ror:4
  }
}
template <int DT>
__device__ __forceinline__ void lj_lane_role(const LjCo& a, int w, int r, const double* VL, const double* GL, const double* D2L,
                                             const double* XL, int NP, const double* AL, const double* TAB1K, double* scr) {
  constexpr int NC = 2 * DT + 2;
  constexpr int SR = 3 * DT + 2;            // scratch row per cell: [mu / tau (DT) | 1 / tau (DT) | 2 free | lambda (DT)]; the NC record sums later take the head
  const ElboDims& dm = a.dm;
  const int D = dm.D, K = dm.K, G4 = (K + 3) / 4;
  const int kgroup = w % G4, s = w / G4;
  const double* g = GL + (size_t)s * GPC_STRIDE(D);
  const int lane = threadIdx.x & 63, pl = lane >> 2, cq = lane & 3;
  const int kk = 4 * kgroup + cq;
  const int k = kk < K ? kk : K - 1;
  VpLayout L{D, K};
  auto fence = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  const double sig = VL[L.sigma() + k];
  double* row = scr + cq * SR;
  // point lane pl owns dimension d = pl (DT <= 12)
  double my_logtau = 0.0;
  {
    double mi = 0.0, it = 0.0, lm = 0.0;
    if (pl < D) {
      lm = VL[L.lambda() + pl];
      const double tau = sqrt(sig * sig * lm * lm + g[pl] + D2L[pl]);  // :164
      my_logtau = log(tau);
      it = 1.0 / tau;
      mi = VL[L.mu() + pl + D * k] * it;
    }
    if (pl < DT) { row[pl] = mi; row[DT + pl] = it; row[NC + pl] = lm; }     // (lambda behind the NC slots the sums will take: it is read again at the end)
  }
  double sumlogtau = dpp_pair_sum<0x128>(my_logtau);      // over the sixteen point lanes of the cell: rotations inside the row, then the rows
  sumlogtau = dpp_pair_sum<0x124>(sumlogtau);
  sumlogtau = xor_sum16(sumlogtau);
  sumlogtau = xor_sum32(sumlogtau);
  fence();
  double mit[DT], itau[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { mit[d] = row[d]; itau[d] = row[DT + d]; }     // zero for padded dimensions: they vanish below
  const double lnnf = g[3 * D] - sumlogtau;  // ln_sf2 + sum_lnell - sum(log(tau_k))  :165
  const double ylnnf = VB_EXP_TAB1K_SCALE * lnnf;
  double accI = 0.0, accA[DT], accQ[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) { accA[d] = 0.0; accQ[d] = 0.0; }
  const double* xr[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) xr[d] = XL + NP * min(d, D - 1) + pl;
  const double* ar = AL + pl;
  // slabs of sixteen points per trip: four while their 4 DT distances fit the registers, two beyond (DT = 10, 12 with four spilled)
  constexpr int U = DT <= 8 ? 4 : 2;
  for (int n0 = 0; n0 < NP; n0 += 16 * U) {
    double dl[U][DT], a2[U], za[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a2[u] = 0.0;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        dl[u][d] = fma(-xr[d][n0 + 16 * u], itau[d], mit[d]);        // delta_k = (mu - x) / tau  :167
        a2[u] = fma(dl[u][d], dl[u][d], a2[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      za[u] = vb_exp_tab1k_m<false>(fma(a2[u], -0.5 * VB_EXP_TAB1K_SCALE, ylnnf), TAB1K) * ar[n0 + 16 * u];   // z_k alpha  :168
#pragma unroll
    for (int u = 0; u < U; ++u) {
      accI += za[u];
      if (a.want_grad) {
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const double t = dl[u][d] * za[u];
          accA[d] += t;                                      // A_d = sum delta_d z alpha          (:207-208 without its factor)
          accQ[d] = fma(dl[u][d], t, accQ[d]);               // sum delta_d^2 z alpha; Q_d = this - I   (:228, :249-250 without theirs)
        }
      }
    }
  }
  // the record's columns as per-lane partial sums (the factors are linear), summed over the cell's point lanes
  double vals[NC];
  vals[0] = accI;
  {
    double accS = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const double qd = accQ[d] - accI, lam_d = row[NC + d];
      const double li = lam_d * itau[d], si = sig * itau[d];
      accS = fma(li * li, qd, accS);                         // dz_dsigma factor   (:228)
      vals[1 + d] = -itau[d] * accA[d];                      // dz_dmu factor      (:207)
      vals[2 + DT + d] = si * si * lam_d * qd;               // dz_dlambda factor  (:249)
    }
    vals[1 + DT] = sig * accS;
  }
  constexpr int N4 = (NC + 3) / 4;
  double tot[N4];
  cell4_sum_tree<NC>(vals, tot);
  fence();           // the scratch rows become the rows of sums
  {
    const int q = lane >> 4, qv = ((q & 1) << 1) | (q >> 1);
#pragma unroll
    for (int m = 0; m < N4; ++m) {
      const int vi = 4 * m + qv;
      if ((lane & 12) == 0 && vi < NC) row[vi] = tot[m];      // the first point lane of each cell in the row
    }
  }
  fence();
  double* o_s = a.lj + ((size_t)r * dm.S + s) * K * (2 * D + 2);
  lj_write_record<DT, SR>(dm, VL, g, D2L, scr, 1, 0, kgroup, a.want_grad, true, o_s);
  fence();
}
