// k_entropy_lane: the Monte-Carlo entropy + reparameterisation gradient (ent/entmc_vbmc.m:49-104) for SMALL mixtures (K <= 16,
// D <= 12, less the corner that does not fit the registers) -- the class where the matrix-core kernel (entropy_mfma.h) pads ten components to a sixteen-wide k-tile, D + 2 columns to
// sixteen, spends a wave instruction on sixteen samples, and pays a 4 us set-up per wave for a 19 us life (BASELINE configs[1]).
//
//   lane  <-> one BASE sample; the lane evaluates both signs of the antithetic pair (entmc_vbmc.m:53-54), which share the even part of
//             every exponent:  E+-_k = C_k +- L_k,  L_k = b_k . u',  C_k = c0_k |u'|^2 + c1_k   (u' = eps sigma_j: coordinates centred on
//             the sample's own component j and scaled by 1 / lambda; b, c0, c1 as in entropy_mfma.h, with the exp table's 1024 / ln 2)
//   wave  <-> one (chunk of 64-sample tiles, source component j, restart r); the K (2 D + 4) coefficients of (r, j) are a table in LDS,
//             read with wave-uniform addresses (broadcast); the K densities of both signs stay in registers until q is known
//             (weight gradient W_l = sum_i n_il / q_i, entmc_vbmc.m:100: one lane-local FMA per pair)
//   workgroup = four waves of ONE restart: the restart's packed parameter block and the 8 KB exp table are staged once for the four
//
// The expected log joint rides in the same launch as a ROLE of the same waves (logjoint_body.h: lj_role_wave): the cell groups of restart
// r are dealt over the waves of r's workgroups, each wave runs its share between the set-up and its sample tiles -- the role's
// latency-bound walk over the training set of one wave overlaps the other wave's arithmetic on the SIMD, every wave carries the same
// mix of work (one round of resident waves, no second kind of workgroup competing for the slots), and a pass of this class is ONE
// chip-wide launch.
// Partial records as k_entropy_mfma's: sum log q' | G[D] | SG | LG[D] | W[K] per (r, j, chunk slot).
#pragma once
#include "device_math.h"
#include "elbo_types.h"
#include "exp2_tab1k.h"
#include "logjoint_body.h"

#define ENT_LANE_WAVES 4          // waves per workgroup
#define ENT_LANE_TILE 64          // base samples per tile (= lanes)
#define ENT_LANE_KMAX 16
#define ENT_LANE_DMAX 12

// dynamic LDS of a launch that carries the log-joint role: X | gpc of every hyper-sample | the restart's vp block | delta^2 | alpha per wave
#define ENT_LANE_ROLE_LDS_MAX (48 * 1024)
static inline size_t ent_lane_role_lds(const EntArgs& ea) {
  if (ea.lj.nwg <= 0) return 0;
  const int D = ea.D, K = ea.K, N = ea.lj.dm.N, S = ea.lj.dm.S;
  return ((size_t)((N + 63) & ~63) * (D + ENT_LANE_WAVES) + (size_t)S * GPC_STRIDE(D) + (size_t)VpLayout{D, K}.stride() + D) * sizeof(double);
}

// waves per SIMD the register budget is set for
#ifdef ENT_LANE_OCC_ALL          // A/B builds (tools/lane_build.py)
#define ENT_LANE_OCC(DT_, KP_) ENT_LANE_OCC_ALL
#endif
// Two.  No lane kernel may spill: with 87 spilled registers (DT = 12, KP = 14, four role slabs per trip) a launch returned run-to-run
// different, wrong sums -- the instantiations whose two signs' densities, weight-gradient and 4 DT gradient accumulators do not fit
// 256 registers are outside the class (abi_elbo.hip: lane_entropy_fits), and tests/test_lane_build.py compiles every instantiation
// and requires a zero spill count and no private segment.
#ifndef ENT_LANE_OCC
#define ENT_LANE_OCC(DT_, KP_) 2
#endif

#ifdef VBMC_INSTRUMENT   // per-wave timeline (tools/lane_timeline.py): [entry, staged, role begin, role end, tiles begin, tiles end, exit] on the 100 MHz counter + HW_ID | XCC_ID << 32
#define LANE_DBG_WAVES 16384
__device__ unsigned long long g_lane_dbg[8 * LANE_DBG_WAVES];
#define LANE_STAMP(i_) st_[i_] = wall_clock64()
#else
#define LANE_STAMP(i_) do { } while (0)
#endif

template <int DT, int KP, bool GRAD>
__global__ void __launch_bounds__(WAVE * ENT_LANE_WAVES, ENT_LANE_OCC(DT, KP)) k_entropy_lane(EntArgs a) {
  constexpr int NW = ENT_LANE_WAVES;
  constexpr int TS = 2 * DT + 4;            // table row of component k: b[DT] | c0 c1 | w a | v[DT]   (a = w / sigma^2, v = a m')
  constexpr int PSL = DT + ENTP_EXTRA;      // padded row of the staged parameter block
  constexpr int NQ = (DT + 3) / 4;          // dim-blocks of four draws
  __shared__ __attribute__((aligned(16))) double TAB[VB_EXP_TAB1K_N];
  __shared__ __attribute__((aligned(16))) double PB[KP * PSL];
  __shared__ __attribute__((aligned(16))) double TJ_all[NW][KP * TS];
  __shared__ double LJP[GRAD ? NW * LJ_LANE_SCR(DT) : 1];           // the role's scratch, per wave
  __shared__ double SM2[NW][KP];                                    // |m'_k|^2 of the wave's (r, j)
  extern __shared__ double RL[];      // the role's staged inputs (dynamic: sized by the launcher, absent without the role)
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = blockIdx.z;
  const int D = a.D, K = a.K;
#ifdef VBMC_INSTRUMENT
  unsigned long long st_[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
  LANE_STAMP(0);
  const bool role = GRAD && a.lj.nwg > 0;
  const int Nn = a.lj.dm.N, Sn = a.lj.dm.S;
  const VpLayout VLY{D, K};
  // role block: X [D][NP] | gpc [S (3 D + 2)] | vp of r [stride] | delta^2 [D] | alpha of the wave's hyper-sample [NW][NP]
  // (NP = N rounded up to 64 points, zero-padded: the role's slabs of four times sixteen points need no bounds)
  const int NP = (Nn + 63) & ~63;
  double* const XL = RL;
  double* const GL = XL + (role ? NP * D : 0);
  double* const VL = GL + (role ? Sn * GPC_STRIDE(D) : 0);
  double* const D2L = VL + (role ? VLY.stride() : 0);
  const int NA = NP;                      // alpha block per wave, zero-padded
  double* const AL = D2L + (role ? D : 0) + (role ? wv * NA : 0);

  // ---- stage the exp table and the restart's packed parameter block [k][m_1..m_D, h, cK, w, wi] (padded to DT, KP)
  {
    constexpr int NTB = VB_EXP_TAB1K_N / (WAVE * NW);
    double tt[NTB];
#pragma unroll
    for (int u = 0; u < NTB; ++u) tt[u] = c_exp2_tab1k[tid + u * WAVE * NW];
    const int PSg = D + ENTP_EXTRA;
    const double* gsrc = a.entp + (size_t)r * K * PSg;
    double pv = 0.0;
    const int k = tid / PSL, cc = tid - k * PSL;      // (KP PSL <= 256: one element per thread)
    if (k < K) {
      if (cc < DT) { if (cc < D) pv = gsrc[k * PSg + cc]; }
      else pv = gsrc[k * PSg + D + (cc - DT)];
    }
    if (role) {
      const double* vsrc = a.vpd + (size_t)r * VLY.stride();
      // (eight loads in flight per thread: the plain copy loop waits for every load in turn -- five round trips to memory for X)
      // X as D rows of NP (zero-padded), then gpc, the vp block and delta^2 -- the role block is contiguous in that order, so the four
      // are ONE index space and every load of a thread is in flight before its first store
      const int nX = NP * D, nG = Sn * GPC_STRIDE(D), nV = VLY.stride(), nAll = nX + nG + nV + D;
      for (int i0 = tid; i0 < nAll; i0 += 8 * WAVE * NW) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * WAVE * NW;
          // (ONE load per slot, its address chosen by selects: a chain of branches would issue -- and wait for -- each source in turn)
          const int d = i / NP, n = i - d * NP;
          const double* p = a.lj.X + ((size_t)d * Nn + n);
          bool ok = n < Nn;
          if (i >= nX) { p = a.lj.gpc + (i - nX); ok = true; }
          if (i >= nX + nG) p = vsrc + (i - nX - nG);
          if (i >= nX + nG + nV) p = a.lj.delta2 + (i - nX - nG - nV);
          if (i >= nAll) ok = false;
          t8[u] = ok ? *p : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u * WAVE * NW < nAll) RL[i0 + u * WAVE * NW] = t8[u];
      }
    }
    if (tid < KP * PSL) PB[tid] = pv;
#pragma unroll
    for (int u = 0; u < NTB; ++u) TAB[tid + u * WAVE * NW] = tt[u];
  }
  const int nc = a.nc_launch;
  const int item = (int)blockIdx.x * NW + wv;
  const bool live = item < K * nc;
  const int j = live ? item / nc : 0, c = live ? item - j * nc : 0;
  const double sigj = a.vpd[(size_t)r * VLY.stride() + VLY.sigma() + j];
  const int G4 = (K + 3) / 4;
  if (role && item < a.lj.nwg) {        // this wave's first cell group: its hyper-sample's alpha, in flight with the workgroup's loads
    const double* asrc = a.lj.alpha + (size_t)(item / G4) * Nn;
    for (int n = lane; n < NA; n += WAVE) AL[n] = n < Nn ? asrc[n] : 0.0;
  }
  __syncthreads();
  LANE_STAMP(1);

  // ---- the log-joint role: cell groups item, item + (waves of the restart), ... of restart r
  auto role_part = [&]() {
    LANE_STAMP(2);
    for (int w = item; w < a.lj.nwg; w += (int)gridDim.x * NW) {
      if (w != item) {
        const double* asrc = a.lj.alpha + (size_t)(w / G4) * Nn;
        for (int n = lane; n < NA; n += WAVE) AL[n] = n < Nn ? asrc[n] : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      lj_lane_role<DT>(a.lj, w, r, VL, GL, D2L, XL, NP, AL, TAB, LJP + wv * LJ_LANE_SCR(DT));
    }
    LANE_STAMP(3);
  };

  auto entropy_part = [&]() {
  // ---- the coefficient table of (r, j), by this wave for itself: |m'_k|^2 first (one lane per component), then every element
  // without a branch (a divergent chain of them was six exposed LDS round trips per pass)
  double* TJ = TJ_all[wv];
  const double* pj = PB + j * PSL;
  const double cKj = pj[DT + 1];
  const double hj_neg = 0.5 / (sigj * sigj);     // = -h_j bit for bit (k_prep computes h = -0.5 / (sigma * sigma))
  constexpr double ESC = VB_EXP_TAB1K_SCALE;
  if (lane < KP) {
    const double* pk = PB + (lane < K ? lane : 0) * PSL;
    double m2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) { const double t = pk[d] - pj[d]; m2 = fma(t, t, m2); }
    SM2[wv][lane] = m2;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int e = lane; e < KP * TS; e += WAVE) {
    const int k = e / TS, col = e - k * TS;
    const bool kv = k < K;
    const double* pk = PB + (kv ? k : 0) * PSL;
    const int dcol = col < DT ? col : (col >= DT + 4 ? col - DT - 4 : 0);
    const double h = pk[DT], cK = pk[DT + 1], wk = pk[DT + 2], wi = pk[DT + 3], dm_ = pk[dcol] - pj[dcol], m2 = SM2[wv][k];
    double v = ESC * (-2.0 * h * dm_);                             // col < DT: m'_ck / sigma_k^2   (zero beyond D: both sides padded with zeros)
    v = col == DT ? ESC * (h + hj_neg) : v;                        // coefficient of |u'|^2, the sample's own exponent folded in
    v = col == DT + 1 ? ESC * (fma(h, m2, cK) - cKj) : v;          // constant part
    v = col == DT + 2 ? wk : v;                                    // w_k             -> q'
    v = col == DT + 3 ? wi : v;                                    // w_k / sigma_k^2 -> A'
    v = col >= DT + 4 ? wi * dm_ : v;                              // -> B'_d
    if (!kv) v = col == DT + 1 ? ESC * -1.0e6 : 0.0;               // absent component: exp -> 0
    TJ[e] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  double accH = 0.0, pm = 1.0;
  int pe = 0, pcnt = 0;
  double accG[DT], accLG[DT], Wk[KP];
#pragma unroll
  for (int d = 0; d < DT; ++d) { accG[d] = 0.0; accLG[d] = 0.0; }
#pragma unroll
  for (int k = 0; k < KP; ++k) Wk[k] = 0.0;

  const int ntile = (a.Mh + ENT_LANE_TILE - 1) / ENT_LANE_TILE;
  const int t0 = (c + a.c0) * a.tiles_per_chunk;
  const int t1 = min(t0 + a.tiles_per_chunk, ntile);
  const double* epsr = a.eps ? a.eps + (size_t)r * a.eps_stride_r + (size_t)j * a.Mh * D : nullptr;
  const unsigned rkey = (unsigned)(a.r0 + r * a.rstride);

  LANE_STAMP(4);
  for (int tile = t0; tile < t1; ++tile) {
    const int b = tile * ENT_LANE_TILE + lane;
    const bool valid = b < a.Mh;
    double u[DT];
    if (epsr) {
#pragma unroll
      for (int d = 0; d < DT; ++d) u[d] = (valid && d < D) ? sigj * epsr[(size_t)b * D + d] : 0.0;
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const vb_d4 z = vb_normal4i(a.seed, (unsigned)b, (unsigned)j, rkey, (unsigned)q);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (4 * q + t < DT) u[4 * q + t] = (4 * q + t < D) ? sigj * z[t] : 0.0;
      }
    }
    double u2 = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) u2 = fma(u[d], u[d], u2);
    const double shift = fma(-u2, hj_neg, cKj);          // exponent of the sample's own component: cK_j - |eps|^2 / 2  (both signs)

    double nP[KP], nM[KP];
    double qP = 0.0, qM = 0.0, AP = 0.0, AM = 0.0;
    double BP[GRAD ? DT : 1], BM[GRAD ? DT : 1];
#pragma unroll
    for (int d = 0; d < (GRAD ? DT : 1); ++d) { BP[d] = 0.0; BM[d] = 0.0; }
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const double* row = TJ + k * TS;
      double L = row[0] * u[0];
#pragma unroll
      for (int d = 1; d < DT; ++d) L = fma(row[d], u[d], L);
      const double C = fma(row[DT], u2, row[DT + 1]);
      nP[k] = vb_exp_tab1k_m<false>(C + L, TAB);
      nM[k] = vb_exp_tab1k_m<false>(C - L, TAB);
      qP = fma(row[DT + 2], nP[k], qP);                  // q' += w_k n_k   (:64)
      qM = fma(row[DT + 2], nM[k], qM);
      if (GRAD) {
        AP = fma(row[DT + 3], nP[k], AP);
        AM = fma(row[DT + 3], nM[k], AM);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          BP[d] = fma(row[DT + 4 + d], nP[k], BP[d]);
          BM[d] = fma(row[DT + 4 + d], nM[k], BM[d]);
        }
      }
    }
    // ---- per-sample scalars and gradient pieces, sign by sign
    {
      const double qa = valid ? qP : 1.0, qb = valid ? qM : 1.0;
      pm *= __builtin_amdgcn_frexp_mant(qa) * __builtin_amdgcn_frexp_mant(qb);   // sum log q' = ln2 * sum exp + log(prod mant)
      pe += __builtin_amdgcn_frexp_exp(qa) + __builtin_amdgcn_frexp_exp(qb);
      accH += valid ? 2.0 * shift : 0.0;
      if (GRAD) {
        const double rqa = valid ? vb_rcp(qP) : 0.0, rqb = valid ? vb_rcp(qM) : 0.0;
        const double ara = AP * rqa, arb = AM * rqb;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const double ga = fma(u[d], ara, -(BP[d] * rqa));      // lambda_d lsum_d / q = (u'_d A' - B'_d) / q'   (:77-79)
          const double gb = fma(-u[d], arb, -(BM[d] * rqb));
          accG[d] += ga + gb;                                    // -> mu_grad (:82)
          accLG[d] = fma(u[d], ga - gb, accLG[d]);               // -> sigma / lambda grads (:87-93), times sigma_j (divided out at the end)
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) Wk[k] = fma(nP[k], rqa, fma(nM[k], rqb, Wk[k]));     // (:100)
      }
      if (++pcnt == 128) {       // renormalise the mantissa product before it can underflow
        pe += __builtin_amdgcn_frexp_exp(pm);
        pm = __builtin_amdgcn_frexp_mant(pm);
        pcnt = 0;
      }
    }
  }
  LANE_STAMP(5);
  accH += log(pm) + 0.693147180559945309417 * (double)pe;

  // ---- fixed-order reductions and the partial record: the per-lane values in record order (padded to DT, KP), summed as a tree
  double* o = a.part + (((size_t)r * K + j) * a.C + c) * a.ncol;
  if (!GRAD) {
    accH = wave_sum_valu(accH);
    if (lane == 0) o[0] = accH;
    return;
  }
  constexpr int NV = 2 + 2 * DT + KP;
  double vals[NV];
  vals[0] = accH;
  {
    const double rs = 1.0 / sigj;     // accLG carried u' = eps sigma_j in place of eps
    double sg = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      vals[1 + d] = accG[d];
      const double lgd = accLG[d] * rs;
      vals[2 + DT + d] = lgd;
      sg += lgd;                      // SG = sum_d LG_d  (entmc_vbmc.m:87; padded dimensions hold zeros)
    }
    vals[1 + DT] = sg;
#pragma unroll
    for (int k = 0; k < KP; ++k) vals[2 + 2 * DT + k] = Wk[k];
  }
  constexpr int N4 = (NV + 3) / 4;
  double tot[N4];
  wave_sum_tree<NV>(vals, tot);
  const int q = lane >> 4, qv = ((q & 1) << 1) | (q >> 1);       // row q holds value 4 m + {0, 2, 1, 3}[q]
#pragma unroll
  for (int m = 0; m < N4; ++m) {
    const int vi = 4 * m + qv;
    int col = -1;
    if (vi == 0) col = 0;
    else if (vi <= DT) { if (vi - 1 < D) col = vi; }
    else if (vi == 1 + DT) col = 1 + D;
    else if (vi < 2 + 2 * DT) { if (vi - 2 - DT < D) col = 2 + D + (vi - 2 - DT); }
    else if (vi < NV) { if (vi - 2 - 2 * DT < K) col = 2 + 2 * D + (vi - 2 - 2 * DT); }
    if ((lane & 15) == 0 && col >= 0) o[col] = tot[m];
  }
  };

  // (Tried: the two waves of a SIMD in opposite orders -- by wave slot parity -- so that one walks its cell group while the other issues
  // tiles.  960 of 1024 SIMDs did run that way and the launch was 17 % SLOWER: the role is issue-bound too, and two waves in different
  // code take longer for the same instructions than two waves in the same loop.  profiles/r06_small_class.md)
  if (role) role_part();
  if (live) entropy_part();
#ifdef VBMC_INSTRUMENT
  LANE_STAMP(6);
  if (lane == 0) {
    const size_t w = ((size_t)r * gridDim.x + blockIdx.x) * NW + wv;
    if (w < LANE_DBG_WAVES) {
      unsigned long long* g = g_lane_dbg + 8 * w;
      for (int i = 0; i < 7; ++i) g[i] = st_[i];
      g[7] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);    // HW_ID | XCC_ID
    }
  }
#endif
}
