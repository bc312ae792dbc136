// k_entropy_mfma: the Monte-Carlo entropy + reparameterisation gradient (ent/entmc_vbmc.m:49-104)
// organised around v_mfma_f64_16x16x4_f64, in the shape of a flash-attention tile:
//
//   S-step   E^T[k][i] = b_k . a_i - shift_i       (16 components x 16 samples, inner dim D+2)
//   exp      n_ik = exp(E_ik)                       (the only transcendental; 4 per lane per k-tile)
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
// 16 base samples, processed twice (+eps, -eps: antithetic, entmc_vbmc.m:53-54).  The pair shares the even part of the
// exponent: E+ = C + L, E- = 2C - E+ with ONE even product C and ONE linear product L per tile (see the S-step below).
// The S-step operands are built once per wave and stay in registers, the PV operands of larger mixtures in LDS; LDS also
// holds the 16 x D eps tile (read in two layouts), the 1024-entry exp table (in the space of the parameter block, which is
// dead once the operands are built; the table's factor 1024/ln2 rides in the S-step operands) and 16 scalars.  The number of
// k-tiles per wave KT <= 4 is a template parameter (K > 64: HV = 2 or 4 waves per workgroup share the components, see below);
// the last (components per wave) mod 16 <= 8 components can run as a lane-layout TAIL instead of a k-tile (TL, see below).
// The tile body is straight-line code: the sign loop, or -- where its registers allow -- both signs staggered (VBMC_STAG_FOR).
// Padded components carry the constant -1e6 and vanish in the exp.
// sum_i log q'_i is accumulated as a mantissa product + exponent sum (one log per 256 samples).
// Partials have the same layout as k_entropy (sum log q | G[D] | SG | LG[D] | W[K]) and are reduced
// by k_ent_reduce / k_finalize in a fixed order.
#pragma once
#include <type_traits>

#include "device_math.h"
#include "elbo_types.h"
#include "exp2_tab1k.h"
#include "logjoint_body.h"

typedef double mf4 __attribute__((ext_vector_type(4)));
// how the tile body learns whether its tile can hold samples beyond Mh: a run-time test, or compiled in (see VBMC_ENT_SPLIT)
struct EntTileAny { static constexpr bool rt = false, val = true; };     // one body for every tile: the selects are always there (a run-time test of
                                                                          // the tile gets if-converted into twice as many)
struct EntTileFull { static constexpr bool rt = false, val = false; };
struct EntTilePartial { static constexpr bool rt = false, val = true; };
#define VBMC_ENT_SPLIT(KT_, QS_, TL_, HV_) ((HV_) == 1)

#ifdef VBMC_INSTRUMENT   // timeline experiment (tools/archive/r4_timeline.py): per wave [entry, loop start, loop end, exit] on the 100 MHz counter + HW_ID + XCC_ID
#define VBMC_DBG_WAVES 32768
__device__ unsigned long long g_ent_dbg[6 * VBMC_DBG_WAVES];
#endif

// The staggered schedule: both signs in straight-line code, the second sign's exponentials issued inside the first sign's
// per-sample latency chain.  It pays where the registers are there: one and three k-tiles per wave (K <= 16, 33..48: no spills,
// 0-6 % faster, tools/tune_sweep.py small); with two k-tiles (168-VGPR budget of three waves per SIMD) and with four (256) it
// spills and loses 2-12 %, so those keep the sign loop.
// At three k-tiles with a component tail and D >= 19 (QS >= 6), and without a tail at D >= 31, the staggered code spills too
// (37-90 VGPRs) and the loop is 2-10 % faster again.
// Round 4 (the in-loop log's twelve constant registers are gone): four k-tiles (-0.6..-2.5 %) and two k-tiles from D = 19 on (-1.8..-3.7 %) too.
#define VBMC_STAG_FOR(KT_, QS_, TL_) ((KT_) == 1 || (KT_) == 4 || ((KT_) == 2 && (QS_) >= 6) || ((KT_) == 3 && (QS_) <= ((TL_) ? 5 : 8)))


// Round 5 (profiles/r05_experiments.md; the measured alternatives -- round 4's forms, the and-mask zeroing EVX, the fp32 S-step F32S -- are in
// the history, docs/history.md):
//  GP2  the gradient epilogue without shuffles and without branches: the lanes of column 1 hand A'_i over with q'_i (same exec-masked
//       store), the sample-layout lanes return 1/q'_i AND A'_i/q'_i, and the PV-layout lanes read the pair with one broadcast ds_read2 --
//       no ds_bpermute (16 per tile), and no `d < D` predicate around the four sample rows (lanes of the padded columns compute values
//       nobody stores): the compiler had made four basic blocks of them, each an exposed LDS round trip.
//  ETZ  dim-blocks that are all padding are zeroed once per wave, not once per tile (four v_mov_b64 per tile).
//  C2   the even part of the exponent rides in the linear product: D + 2 <= 4 QS always, so the inner slots D and D + 1 of the last
//       dim-block(s) are free -- they carry [|u'|^2, 1] against [h_k - h_j, const_k], and E+ = L + C comes out of QS MFMAs per k-tile
//       instead of QS + 1.  The second sign still needs C by itself (E- = 2C - E+): 2C_ik = 2 c0_k |u'_i|^2 + 2 c1_k, one FMA per
//       element from a pair table in LDS (one broadcast ds_read_b128 per element, immediate offsets, issued behind the MFMAs).
//       Per tile -KT MFMA (27 ns each), +4 KT VALU.

// Where C2 and GP2 are used: everywhere except the instantiations where the sweep of every (k-tiles, tail, waves per workgroup) class
// over D = 2..32 measured them slower than round 4's forms (tools/tune_sweep.py, variants noc2 / nogp2 of tools/tune_build.py against the
// tree and against round 4's HEAD; profiles/r05_shape_sweep.md).  The losers are the register-bound kernels: the pair reads and the lane
// constants of C2 tip them into spilling (D = 20, K = 112: 2.28 -> 2.90 ms; D = 18, K = 256: 10.6 -> 15.8; D = 28, K = 36: 0.74 -> 1.05),
// and at three waves per SIMD with one k-tile the shuffles GP2 removes were hidden anyway.
constexpr bool ent_c2_for(int KT, int QS, int TL, int HV) {
  if (HV == 1) {
    if (KT == 2) return !(TL == 1 && QS >= 6);
    if (KT == 3) return QS <= 5 || (TL == 0 && QS != 7) || (TL == 1 && QS == 7);
    if (KT == 4) return QS >= 5;
    return true;
  }
  if (HV == 2) {
    if (KT == 2) return TL != 2;
    if (KT == 3) return !((TL == 1 && (QS == 5 || QS == 6)) || (TL == 2 && QS == 6));
    return true;
  }
  if (KT == 2) return !((TL == 1 && QS >= 6) || (TL == 2 && QS >= 5));
  if (KT == 3) return !(TL == 1 && QS == 7);
  return QS != 5;
}
constexpr bool ent_gp2_for(int KT, int QS, int TL, int HV) { return !(HV == 1 && KT == 1 && TL == 0 && QS <= 3); }

// Ordering point for LDS words that only ONE wave touches (lanes of a wave exchanging values through LDS): the hardware
// executes a wave's LDS instructions in order, so no s_barrier and no full s_waitcnt drain is needed -- only the compiler must
// keep the accesses in program order (wave-level fence).  The waves of a two-wave workgroup meet only at the eps staging and
// at the PV exchange (__syncthreads there).
template <int HV>
__device__ __forceinline__ void ent_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the eps tile is shared by the waves of a workgroup: a real barrier when there are two, the wave-level fence when there is one
template <int HV>
__device__ __forceinline__ void ent_sync_wg() {
  if (HV == 1) ent_sync<1>();
  else __syncthreads();
}

// HV = 2 or 4 (K > 64): the components are split between the HV waves of a workgroup, each running the KT <= 4
// register-resident body on its share; per sign the waves exchange their partial PV outputs (16 x 16 NPV doubles per wave
// through LDS, one workgroup barrier) and all continue with the full q', A', B' -- the partials are added in wave order on
// every wave, so all hold identical bits.  Wave 0 owns the entropy accumulator, the column blocks of the gradient are dealt
// to the waves, each wave owns the weight gradient of its components.  The exchange buffers reuse the parameter block's LDS
// (needed only while the operand fragments are built).
//
// TL = 1, 2: the last (components of the wave) mod 16 <= 4 TL components do not get a k-tile of their own.  A 16-wide k-tile for two
// components (K = 50) costs four S-step MFMAs, 26 VGPRs of operands / exponents / accumulators and an exp register per sign; as a
// TAIL they live in the lane layout (sample li, tail component lg) -- one value per lane: the linear part of the exponent is D
// FMAs per tile from two LDS rows (the sample's draws, the component's coefficients), one exp per sign, and the lane's value IS
// the A operand of one PV MFMA (component index = inner index lg), its weight-gradient term one lane-local FMA.
// CO = true (single-wave workgroups, D <= 30): the launch carries the expected log joint as well -- its first a.lj.rows grid rows are
// single-wave log-joint workgroups (logjoint_body.h: lj_co_role), the entropy workgroups follow.  For a single chain (one restart, a
// few dozen samples per component) both kernels are bound by their own dependent chains, not by the chip: side by side an Adam
// iteration loses the shorter of the two (round 3).  These kernels are built for two waves per SIMD whatever the entropy body needs
// (ONE from D = 15 on): the grids they serve do not fill the chip anyway, and the log-joint body keeps 6 x 4 QS values per lane in
// registers (185 VGPRs at QS = 3, beyond 256 from QS = 6 on, where the accumulation registers take the overflow).
// EM (round 5): the instantiation reads its draws from memory (parity mode / eps_mode 2) -- it spends QS registers per lane on the tile
// loaded one tile ahead (EPF below).  EM = false: the device-RNG launch of the same shape; those registers hold PV operands instead
// (VBR).  Only the instantiations that prefetch exist twice (ent_mfma_inst.hip); everywhere else EM = true is the one kernel for both.
template <int QS, int KT, bool GRAD, bool SPARSE, int HV = 1, int TL = 0, bool CO = false, bool EM = true, bool WALK = false>
// Waves per SIMD the register budget is set for: three (168 VGPRs) for the small kernels, two (256) from three k-tiles on.
// Two k-tiles + a tail of ONE value per lane (K = 33..36) spills 14 VGPRs at 168 and is still 5-9 % faster than the spill-free
// two-wave build; with TWO tail values per lane (K = 37..40: 24 spilled) the two-wave build wins by 2-4 % (round 3,
// tools/ent_experiments.py x_w2 at D = 10), so that one instantiation moved (single-wave workgroups only: the multi-wave ones
// were not re-measured).
// Which workgroup shapes share the even part of the exponent between the antithetic pair (EO: 4 instead of 2 QS S-step MFMAs per k-tile and
// tile).  Single-wave workgroups since round 2; round 4: two- and four-wave workgroups (64 < K <= 256) too -- the second sign's exponents are
// parked in LDS (NML) and the PV exchange is single-buffered to pay for that LDS: up to -35 % at D >= 20 (tools/tune_sweep.py, r04_experiments.md
// section 10), BASELINE configs[4] 9.75 -> 8.5 ms.  (The plain per-sign S-step, !EO, remains as the general form of the sign loop.)
#define VBMC_ENT_EO(HV_) true
// ONE wave per SIMD (512 registers: nothing spills) where the two-wave build spills so much that losing the second wave's latency
// hiding is the smaller evil (round 3, tools/tune_build.py w1:-DVBMC_ENT_WAVES_ALL=1 against the policy over 112 shapes,
// profiles/r03_shape_sweep.md): four k-tiles on one wave from D = 15 on (K = 53..64: 11-23 % faster), three k-tiles from D = 27 on
// (7-23 %), four-wave workgroups with four k-tiles from D = 23 on (round 4: from D = 19 on -- K = 256, D = 20: 16.9 ms
// at two waves, 13.8 at one, with the shared even part in these kernels; D = 18: 12.3 against 13.1 the other way) and with three at D >= 31 (K = 193..256: 27-49 %).  Everywhere
// else one wave per SIMD costs 2-49 %.
#define VBMC_ENT_ONE_WAVE(KT_, QS_, TL_, HV_) \
  (((HV_) == 1 && (KT_) == 4 && (QS_) >= 5) || ((HV_) == 1 && (KT_) == 3 && (QS_) >= 8) || ((HV_) == 4 && (KT_) == 4 && (QS_) >= 6) || \
   ((HV_) == 4 && (KT_) == 3 && (QS_) >= 9))
#define VBMC_ENT_WAVES(KT_, QS_, TL_, HV_) \
  (VBMC_ENT_ONE_WAVE(KT_, QS_, TL_, HV_) ? 1 : ((((KT_) <= 2 && (QS_) <= 4) && !((KT_) == 2 && (TL_) == 2 && (HV_) == 1)) ? 3 : 2))
// ent_mfma_segment: the work of one wave (workgroup) on ONE (component j, restart r): tiles [t0, t1) of the component's samples, partial
// record into slot c.  The kernel below calls it once (the chunk grid) or once per segment of the wave's tile range (the walk, see there);
// pdone / ptot: the tiles the wave had behind it when it entered the segment / has in all (the progress its issue priority follows).
__device__ __forceinline__ void ent_mfma_segment(const EntArgs& a, const int c, const int j, const int r, const int t0, const int t1,
                                                 const int pdone, const int ptot) {
  static_assert(!CO || (HV == 1 && QS <= 8 && !SPARSE), "the log-joint role exists for single-wave dense kernels at D <= 30");
  static_assert(!WALK || (HV == 1 && !CO), "the walk exists for single-wave workgroups without the log-joint role");
  static_assert(KT <= 4 && (HV == 1 || HV == 2 || HV == 4 || HV == 8), "larger mixtures are split over the waves of a workgroup (HV = 2, 4; round 5: 8, K <= 512)");
  static_assert(TL == 0 || ((TL == 1 || TL == 2) && !SPARSE), "the component tail (one or two values per lane) exists for the dense kernels only");
  constexpr int TLN = TL > 0 ? TL : 1;     // tail values per lane: tail component 4u + lg, u < TL (the layout of a k-tile's register u)
  constexpr bool SP = SPARSE;  // block-sparse variant: uniform per-k-tile branches; the dense variant is branch-free
  constexpr int DP = 4 * QS;               // padded eps row length
  constexpr int NPV = (4 * QS + 15) / 16;  // 16-column blocks of the PV output (D + 2 columns)
  constexpr int QL = QS;                   // MFMAs of the linear part of the S-step (inner index c = 4q + lg < D, zero operands beyond D: for
                                           // D mod 4 in {3, 0} the last one multiplies zeros -- a compile-time count keeps the KT chains branch-free)
  __shared__ double Et[16 * DP];   // eps tile [i][d], staged by wave 0 and shared by the HV waves of the workgroup
  constexpr bool GP2 = GRAD && ent_gp2_for(KT, QS, TL, HV);
  __shared__ double RQ_all[HV][GP2 ? 32 : 16];   // q'_i then 1/q'_i  (GP2: and A'_i then A'_i/q'_i behind them)
  __shared__ double BND_all[HV][SPARSE ? KT * 16 * 3 : 1];  // per component: |m'_k|, cK_k - cK_j, h_k  (block-sparse bound)
  // partial PV outputs of the two halves, double-buffered by sign so that one workgroup barrier per sign is enough
  constexpr int YXN = NPV * 4 * WAVE;      // doubles per (sign, wave) slot of the PV exchange (in the dynamic LDS, see PB)
  // PV "B" operands of large mixtures live in LDS (lane-contiguous: conflict-free ds_read_b64 right before the MFMA that
  // consumes them) -- the 32 VGPRs they would occupy hold the second sign's exponents instead (see the S-step)
  // EO: the even / odd split of the S-step below (needs 32 more VGPRs for the second sign's exponents, paid for by VBL).
  // Multi-wave workgroups (K > 64) kept the plain per-sign S-step through round 3 (their LDS already holds the PV exchange buffers and a
  // larger parameter block, and 32 KB of PV operands more would halve the resident waves); since round 4 they share the even part as
  // well, with the second sign's exponents parked in LDS (NML) and the PV exchange single-buffered (YXSB) -- see VBMC_ENT_EO.
  constexpr bool EO = VBMC_ENT_EO(HV);
  constexpr bool YXSB = HV > 1 && EO;   // one PV exchange buffer for both signs (a barrier more per tile): the LDS it frees holds the parked exponents
  // US (round 4): the LDS tile holds u' = eps sigma_j, not eps -- every reader wanted the product (S-step operand, tail, gradient
  // epilogue: a multiply per use), the own exponent comes from |u'|^2 as well
  constexpr bool US = EO;
  constexpr bool VBL = GRAD && HV == 1 && (KT >= 3 || NPV >= 2);
  __shared__ double VBS_all[HV][VBL ? KT * 4 * NPV * WAVE : 1];
  constexpr bool C2 = EO && !SPARSE && ent_c2_for(KT, QS, TL, HV);
  // component (within the wave's share) held by accumulator register rr of this lane group in k-tile kt
#define ENT_CI(kt_, rr_) (16 * (kt_) + 4 * (rr_) + lg)
  __shared__ __attribute__((aligned(16))) double SCP_all[HV][C2 ? KT * 16 * 2 : 2];   // C2: [component 16 kt + c][2 c0, 2 c1] of the even part
  __shared__ double BTL_all[HV][TL ? 4 * TL * DP : 1];  // tail: linear S-step coefficients [t][d] (x 1024/ln2), zero beyond D and for absent components
  constexpr bool NML = HV > 1 && EO;   // multi-wave workgroups park the second sign's exponents in LDS: their registers are spoken for
  __shared__ double NMS_all[HV][NML ? KT * 4 * WAVE : 1];
#ifdef VBMC_INSTRUMENT
  const unsigned long long wckE = wall_clock64();
#endif
  int tid_ = threadIdx.x;
  // (the segment loop of the walk: nothing of a segment's set-up may be hoisted out of it and kept alive across the tile loops -- the exp table's
  // sixteen values per lane, for one.  Every lane-dependent value of the set-up derives from this.)
  if (WALK) asm volatile("" : "+v"(tid_));
  const int tid = tid_, wv = tid >> 6, hv = HV == 1 ? 0 : wv, lane = tid & 63;
  const int li = lane & 15, lg = lane >> 4;
  // (the uniform side of the same: the dimensions and the three base pointers the set-up and the epilogue start from)
  int D_ = a.D, K_ = a.K;
  const double *entp_ = a.entp, *vpd_ = a.vpd;
  double* part_ = a.part;
  if (WALK) asm volatile("" : "+s"(D_), "+s"(K_), "+s"(entp_), "+s"(vpd_), "+s"(part_));
  const int D = D_, K = K_;
  double* RQ = RQ_all[wv];
  double* NMS = NMS_all[wv];
  double* BND = BND_all[hv];
  double* VBS = VBS_all[hv];
  double* BTL = BTL_all[hv];
  double* SCP = SCP_all[hv];
  const int Kh = (K + HV - 1) / HV;                    // components per wave
  const int kbase = hv * Kh;
  const int Kw = min(K, kbase + Kh) - kbase;           // this wave's components: kbase .. kbase + Kw - 1
  const int PSg = D + ENTP_EXTRA;
  // stage this restart's packed parameter block [k][m_1..m_D, h, cK, w, wi] in LDS with coalesced loads;
  // the per-lane operand fragments below are gathered from LDS, not from global memory
  extern __shared__ double PB[];
  // The parameter block is needed only while the operand fragments are built; afterwards its LDS holds the exp table
  // 2^(j/1024) (8 KB) and, behind it, the PV exchange buffers of multi-wave workgroups [sign][wave][YXN] (the launcher sizes the
  // dynamic LDS for the larger of the two uses)
  double* const TAB = PB;
  double* const YX = PB + VB_EXP_TAB1K_N;
  // (round 5) the exp table's global loads and sigma_j leave at the kernel's entry, beside the parameter block's: three round trips to
  // memory one after the other (block, sigma_j behind the barrier, table behind the operand build) were a third of the 9.7 us set-up
  constexpr int NTH0 = WAVE * HV;
  constexpr int NTB0 = (VB_EXP_TAB1K_N + NTH0 - 1) / NTH0;
  double tt[NTB0];
#pragma unroll
  for (int u = 0; u < NTB0; ++u) tt[u] = c_exp2_tab1k[min(tid + u * NTH0, VB_EXP_TAB1K_N - 1)];
  const double sigj = vpd_[(size_t)r * VpLayout{D, K}.stride() + VpLayout{D, K}.sigma() + j];
  {
    // eight loads in flight per lane: the plain copy loop waits for every load in turn, and with few tiles per wave (a
    // single chain) this setup is a quarter of the kernel
    const double* gsrc = entp_ + (size_t)r * K * PSg;
    constexpr int NT = WAVE * HV;
    const int n = K * PSg;
    int idx = tid;
    for (; idx + 7 * NT < n; idx += 8 * NT) {
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t8[u] = gsrc[idx + u * NT];
#pragma unroll
      for (int u = 0; u < 8; ++u) PB[idx + u * NT] = t8[u];
    }
    {
      double t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t8[u] = (idx + u * NT < n) ? gsrc[idx + u * NT] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (idx + u * NT < n) PB[idx + u * NT] = t8[u];
    }
  }
  __syncthreads();
  const double* gp = PB;
  const double* pj = gp + (size_t)j * PSg;
  const double cKj = pj[D + 1];
  const double hj_neg = 0.5 / (sigj * sigj);   // = -h_j bit for bit (k_prep computes h = -0.5/(sigma*sigma))
  const int nr_last = TL ? 4 : max(1, min(4, (Kw - 16 * (KT - 1) + 3) >> 2));  // accumulator registers with a valid component in the last k-tile (1..4)
  constexpr unsigned FULL_MASK = (1u << KT) - 1u;
  const double logwj = SPARSE ? log(pj[D + 2]) : 0.0;

  // ---- mixture-side operand fragments (registers, built once).  The S-step operands carry the factor 1024/ln2 of the exp's
  // range reduction (device_math.h: vb_exp_tab1k): the MFMAs deliver E * 1024/ln2
  constexpr double ESC = VB_EXP_TAB1K_SCALE;
  double SA[KT][QL];  // S-step "A" operand, linear part: comp 16kt + li, inner c = 4q + lg < D
  double SC[KT];              // S-step "A" operand, even part: inner index lg = 0 (coefficient of |u'|^2), 1 (constant), 2, 3 (zero)
  double VB[VBL ? 1 : KT][4][NPV];   // PV "B" operand: comp 16kt + 4r + lg, column 16pv + li      (GRAD; in LDS when VBL)
  double WF[KT][4];           // w_k for comp 16kt + 4r + lg                                  (!GRAD)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = 16 * kt + li;
    const bool kv = k < Kw;
    const double* pk = gp + (size_t)(kv ? kbase + k : 0) * PSg;
    double h = pk[D];
    double m2 = 0.0;
    for (int d = 0; d < D; ++d) { double t = pk[d] - pj[d]; m2 = fma(t, t, m2); }
    if (SPARSE && lg == 0) {
      BND[(16 * kt + li) * 3 + 0] = sqrt(m2);
      BND[(16 * kt + li) * 3 + 1] = kv ? pk[D + 1] - cKj : -1.0e30;
      BND[(16 * kt + li) * 3 + 2] = h;
    }
    // E_ik = [linear in u'_i] + [even in u'_i]: the antithetic pair +-eps shares the even part and flips the linear one, so
    // per tile the S-step is ONE even product C (inner dimension 4: |u'|^2, 1, 0, 0) and ONE linear product L (inner
    // dimension D) for both signs: E+ = C + L accumulates L on top of C, E- = 2C - E+
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const int cc = 4 * q + lg;
      if (EO) {
        double sav = (kv && cc < D) ? ESC * (-2.0 * h * (pk[cc] - pj[cc])) : 0.0;   // m'_ck / sigma_k^2   (h = -1/(2 sigma^2))
        if (C2) {   // the even part in the free inner slots D (coefficient of |u'|^2) and D + 1 (constant; padded component: exp -> 0)
          if (cc == D) sav = kv ? ESC * (h + hj_neg) : 0.0;
          if (cc == D + 1) sav = ESC * (kv ? fma(h, m2, pk[D + 1]) - cKj : -1.0e6);
        }
        SA[kt][q] = sav;
      } else {   // plain S-step: linear and even columns in one (D + 2)-column operand, QS MFMAs per sign
        double v;
        if (!kv) v = (cc == D + 1) ? -1.0e6 : 0.0;
        else if (cc < D) v = -2.0 * h * (pk[cc] - pj[cc]);
        else if (cc == D) v = h + hj_neg;
        else if (cc == D + 1) v = fma(h, m2, pk[D + 1]) - cKj;
        else v = 0.0;
        SA[kt][q] = ESC * v;
      }
    }
    // the sample's own exponent -shift_i = -cK_j + |u'_i|^2/(2 sigma_j^2) is folded into the two even columns: accumulators start at 0
    SC[kt] = ESC * (!kv ? (lg == 1 ? -1.0e6 : 0.0)                          // padded component: exp -> 0
                        : (lg == 0 ? h + hj_neg : (lg == 1 ? fma(h, m2, pk[D + 1]) - cKj : 0.0)));
    if (C2 && lg < 2) SCP[(16 * kt + li) * 2 + lg] = 2.0 * SC[kt];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int k2 = ENT_CI(kt, rr);
      const bool kv2 = k2 < Kw;
      const double* p2 = gp + (size_t)(kv2 ? kbase + k2 : 0) * PSg;
      if (GRAD) {
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv) {
          const int col = 16 * pv + li;
          double v = 0.0;
          if (kv2) {
            if (col == 0) v = p2[D + 2];                                        // w_k            -> q'
            else if (col == 1) v = p2[D + 3];                                   // w_k/sigma_k^2  -> A'
            else if (col < 2 + D) v = p2[D + 3] * (p2[col - 2] - pj[col - 2]);  // -> B'_d
          }
          if (VBL) VBS[((kt * 4 + rr) * NPV + pv) * WAVE + lane] = v;
          else VB[VBL ? 0 : kt][rr][pv] = v;
        }
      } else {
        WF[kt][rr] = kv2 ? p2[D + 2] : 0.0;
      }
    }
  }

  // ---- tail components 16 KT + lg (TL): even-part coefficients, PV operand and weight in registers (one component per lane
  // group), the linear coefficients as a row of BTL
  double tC0[TLN], tC1[TLN], VBt[TLN][NPV], WFt[TLN], Wt[TLN];
#pragma unroll
  for (int u = 0; u < TLN; ++u) {
    tC0[u] = 0.0; tC1[u] = 0.0; WFt[u] = 0.0; Wt[u] = 0.0;
#pragma unroll
    for (int pv = 0; pv < NPV; ++pv) VBt[u][pv] = 0.0;
  }
  if (TL) {
#pragma unroll
    for (int u = 0; u < TLN; ++u) {
      const int tq = 4 * u + lg;               // tail component of this lane group
      const int kq = 16 * KT + tq;
      const bool kvq = kq < Kw;
      const double* pq = gp + (size_t)(kvq ? kbase + kq : 0) * PSg;
      const double h = pq[D];
      double m2 = 0.0;
      for (int d = 0; d < D; ++d) { double t = pq[d] - pj[d]; m2 = fma(t, t, m2); }
      tC0[u] = kvq ? ESC * (h + hj_neg) : 0.0;
      tC1[u] = ESC * (kvq ? fma(h, m2, pq[D + 1]) - cKj : -1.0e6);            // absent component: exp -> 0
      for (int d = li; d < DP; d += 16) BTL[tq * DP + d] = (kvq && d < D) ? ESC * (-2.0 * h * (pq[d] - pj[d])) : 0.0;
      if (GRAD) {
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv) {     // PV "B" operand: inner index lg <-> tail component 4u + lg, column 16 pv + li
          const int col = 16 * pv + li;
          double v = 0.0;
          if (kvq) {
            if (col == 0) v = pq[D + 2];
            else if (col == 1) v = pq[D + 3];
            else if (col < 2 + D) v = pq[D + 3] * (pq[col - 2] - pj[col - 2]);
          }
          VBt[u][pv] = v;
        }
      } else {
        WFt[u] = kvq ? pq[D + 2] : 0.0;
      }
    }
  }

  // ---- the parameter block is dead: its LDS becomes the exp table
  __syncthreads();
  {
#pragma unroll
    for (int u = 0; u < NTB0; ++u) if (VB_EXP_TAB1K_N % NTH0 == 0 || tid + u * NTH0 < VB_EXP_TAB1K_N) TAB[tid + u * NTH0] = tt[u];
  }
  __syncthreads();

#define SAV(kt_, q_) SA[kt_][q_]
  // VBR (round 5): with the registers the gradient epilogue gave back (GP2) and one PV accumulator set (TWO below), PV operands of the
  // three-k-tile kernels stay in registers again -- the LDS reads of the PV step were worth 2-3 % (profiles/r05_experiments.md: all
  // twelve in registers -3.4 %, ten -2.5 %, with the parity-mode prefetch holding its QS registers): all of them in the device-RNG
  // instantiation (EM = false), ten in the one that prefetches its draws.
  constexpr int VBR = (VBL && NPV == 1 && HV == 1 && KT == 3 && !CO) ? ((EM && QS <= 4) ? (TL == 2 ? 4 : 10) : 12) : 0;
  double VBreg[VBR > 0 ? VBR : 1];
#pragma unroll
  for (int u = 0; u < VBR; ++u) VBreg[u] = VBS[u * WAVE + lane];
#define VBV(kt_, rr_, pv_) (VBL ? (((kt_) * 4 + (rr_)) * NPV + (pv_) < VBR ? VBreg[(((kt_) * 4 + (rr_)) * NPV + (pv_)) < VBR ? (((kt_) * 4 + (rr_)) * NPV + (pv_)) : 0] : VBS[(((kt_) * 4 + (rr_)) * NPV + (pv_)) * WAVE + lane]) : VB[VBL ? 0 : (kt_)][rr_][pv_])
  double accH = 0.0, accG[NPV], accLG[NPV];
  double pm = 1.0;            // running product of mantissas of q'
  int pe = 0, pcnt = 0;       // running sum of exponents
  double Wacc[KT][4];
#pragma unroll
  for (int pv = 0; pv < NPV; ++pv) { accG[pv] = 0.0; accLG[pv] = 0.0; }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Wacc[kt][rr] = 0.0;

#ifdef VBMC_INSTRUMENT   // timing experiment: shader-clock ticks (s_memtime) against the constant 100 MHz counter over one wave's tile loop
  const unsigned long long wck0 = wall_clock64();
#endif
  const int ntile = (a.Mh + 15) >> 4;
  const double sfm0 = lg == 0 ? 1.0 : 0.0, sfm1 = lg == 1 ? 1.0 : 0.0;   // sample-side operand of the even S-step product (EO)
  const int emask = (4 * (QS - 1) + lg < D) ? -1 : 0;   // this lane's slot of the last dim-block: a dimension (all ones) or padding (zero)
  const int emask2 = (QS >= 2 && 4 * (QS - 2) + lg < D) ? -1 : 0;   // ... and of the one before
  const double* epsr = a.eps ? a.eps + (size_t)r * a.eps_stride_r + (size_t)j * a.Mh * D : nullptr;
  // Parity mode (draws from memory: ent/entmc_vbmc.m:53-55 with the caller's randn stream): the tile of draws is loaded ONE TILE AHEAD
  // into QS registers per lane (round 4) -- issued right after the current tile went to LDS, waited for at the next tile's start, a whole
  // tile of arithmetic later -- instead of a load-and-wait at the head of every tile.  Where the registers are there (EPF); the same
  // kernel serves the device-RNG mode, so the registers are spent in both.
  constexpr bool EPF = EM && HV == 1 && QS <= 4 && KT <= 3 && !SPARSE;
  double epre[EPF ? QS : 1];
  auto eps_fetch = [&](const int tile) {
#pragma unroll
    for (int u = 0; u < (EPF ? QS : 0); ++u) {
      const int idx = lane + u * WAVE, i = idx / DP, d = idx - i * DP;     // (16 DP = 64 QS slots: every lane has exactly QS)
      double v = 0.0;
      if (tile < t1 && d < D && tile * 16 + i < a.Mh) v = epsr[(size_t)(tile * 16 + i) * D + d];
      epre[u] = v;
    }
  };
  if (EPF && epsr) eps_fetch(t0);
  // (ETZ) the device-RNG path writes only the dim-blocks that hold a dimension: the others are zeroed here, once.  The first ordering point
  // of the tile body lies between this and the first read.
  if (!epsr && (HV == 1 || hv == 0))
    for (int idx = lane; idx < 16 * DP; idx += WAVE) Et[idx] = 0.0;
  const int evm1 = emask, evm2 = emask2;
  // (C2) the sample-side values of the even slots: slot D takes |u'_i|^2 (factor c2u), slot D + 1 the constant 1 (c2o); they sit in the last
  // dim-block, or -- D = 4 QS - 5 -- slot D in the last slot of the one before
  const double c2u1 = (4 * (QS - 1) + lg == D) ? 1.0 : 0.0, c2o1 = (4 * (QS - 1) + lg == D + 1) ? 1.0 : 0.0;
  const double c2u2 = (QS >= 2 && 4 * (QS - 2) + lg == D) ? 1.0 : 0.0;

  // The tile body, compiled twice where it pays (VBMC_ENT_SPLIT): once for the full tiles -- no sample-validity selects at all: a
  // v_cndmask_b32 costs four fp64 operations on this chip (tools/valu_rate.hip), and written as rare uniform branches inside one body
  // they cost registers the staggered schedule does not have -- and once for the single tile of a component that can hold samples
  // beyond Mh (its last).  Elsewhere one body with the run-time test.
  auto tile_body = [&](const int tile, auto pkind) __attribute__((always_inline)) {
    using PK = decltype(pkind);
    const int b0 = tile * 16;
    // ---- stage the 16 x D eps tile in LDS (zeros for padded dims / samples beyond Mh)
    ent_sync_wg<HV>();
    if (HV > 1 && hv != 0) {
      // wave 0 stages (and, in device-RNG mode, draws) the tile for both
    } else if (EPF && epsr) {
#pragma unroll
      for (int u = 0; u < (EPF ? QS : 0); ++u) Et[lane + u * WAVE] = US ? sigj * epre[u] : epre[u];
      eps_fetch(tile + 1);
    } else if (epsr) {
      for (int idx = lane; idx < 16 * DP; idx += WAVE) {
        const int i = idx / DP, d = idx - i * DP;
        Et[idx] = (d < D && b0 + i < a.Mh) ? (US ? sigj * epsr[(size_t)(b0 + i) * D + d] : epsr[(size_t)(b0 + i) * D + d]) : 0.0;
      }
    } else {
      // no select on the way into the tile (v_cndmask_b32 costs four fp64 operations on this chip: tools/valu_rate.hip): draws
      // land in the padded dimensions and in the samples beyond Mh too; the padded dimensions are masked where they could matter
      // (ev below: one multiply; every other reader has zero coefficients beyond D or checks d < D), the samples beyond Mh by
      // svalid in the per-sample scalars (their densities are those of ordinary draws: finite)
      // (the Philox key schedule -- fourteen uniform words -- is recomputed from the seed every tile, on the scalar unit beside the vector work:
      // kept across the tile loop it is fourteen scalar registers this kernel does not have, i.e. v_readlane_b32 reloads on the vector pipe)
      unsigned long long seed_t = a.seed;
      if (WALK) asm volatile("" : "+s"(seed_t));
#pragma unroll
      for (int q = lg; q < QS; q += 4) {
        double z4[4] = {0.0, 0.0, 0.0, 0.0};
        // inlined (round 4: -2.5 % at the headline shape -- no call, no wait for every outstanding memory operation at its entry, scalar key
        // schedule) except in the two instantiations where the inlined body costs registers the kernel does not have (tools/tune_sweep.py:
        // D = 18, K = 80: +6 %, D = 24, K = 96: +11 %)
#define VBMC_RNG_CALL_FOR(KT_, QS_, TL_, HV_) ((HV_) == 2 && (((KT_) == 2 && (TL_) == 2 && (QS_) == 5) || ((KT_) == 3 && (TL_) == 0 && (QS_) == 7)))
        if constexpr (!VBMC_RNG_CALL_FOR(KT, QS, TL, HV)) {
        if (q < (D + 3) / 4) { const vb_d4 zz = vb_normal4i(seed_t, (unsigned)(b0 + li), (unsigned)j, (unsigned)(a.r0 + r * a.rstride), (unsigned)q); z4[0] = zz[0]; z4[1] = zz[1]; z4[2] = zz[2]; z4[3] = zz[3]; }
        } else {
        if (q < (D + 3) / 4) vb_normal4(seed_t, (unsigned)(b0 + li), (unsigned)j, (unsigned)(a.r0 + r * a.rstride), (unsigned)q, z4);
        }
#undef VBMC_RNG_CALL_FOR
        if (q < (D + 3) / 4) {     // (ETZ: an all-padding dim-block has been zero since the start of the wave)
#pragma unroll
        for (int t = 0; t < 4; ++t) Et[li * DP + 4 * q + t] = US ? sigj * z4[t] : z4[t];
        }
      }
    }
    ent_sync_wg<HV>();
    // sample-side fragments: lane (li, lg) holds a_i[c = 4q + lg]
    double ev[QS];
    double e2 = 0.0;
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      ev[q] = Et[li * DP + 4 * q + lg];   // zero beyond D
      // (device-RNG tiles carry draws in the padded dimensions: D = 4 QS - 5 .. 4 QS - 2, so the padding is the tail of the last dim-block
      // and, for D = 4 QS - 5, the whole of it plus the last slot of the one before)
      if (q == QS - 1) ev[q] = __hiloint2double(__double2hiint(ev[q]) & evm1, __double2loint(ev[q]) & evm1);
      if (QS >= 2 && q == QS - 2) ev[q] = __hiloint2double(__double2hiint(ev[q]) & evm2, __double2loint(ev[q]) & evm2);
      e2 = fma(ev[q], ev[q], e2);
    }
    e2 = xor_sum16(e2);     // (PLS, round 5: on the VALU -- |u'|^2 heads the last MFMA group of the S-step and the second sign's exponents)
    e2 = xor_sum32(e2);
    // US: the sum above is |u'_i|^2 already
    double shift = US ? fma(-e2, hj_neg, cKj) : cKj - 0.5 * e2;              // exponent of the sample's own component: cK_j - |eps_i|^2 / 2
    const double u2 = US ? e2 : sigj * sigj * e2;                            // |u'_i|^2
    if (US && SPARSE) e2 *= 2.0 * hj_neg;                                    // |eps_i|^2 for the bound below
    constexpr bool partial = PK::val;   // can this tile hold samples beyond Mh?  (only the last tile of a component can)
    const bool svalid = b0 + li < a.Mh;
    if (partial) shift = svalid ? shift : 0.0;
    // ---- block-sparse mode: which k-tiles can contribute more than exp(-cutoff) * q to any sample of this tile?
    // n_ik / q'_i <= exp(cK_k - cK_j - (max(0, |m'_k| - |u'_i|))^2 / (2 sigma_k^2) + |u'_i|^2 / (2 sigma_j^2)) / w_j
    unsigned act = FULL_MASK;
    if (SPARSE) {
      double e2m = e2;
      e2m = fmax(e2m, __shfl_xor(e2m, 1, 64)); e2m = fmax(e2m, __shfl_xor(e2m, 2, 64));
      e2m = fmax(e2m, __shfl_xor(e2m, 4, 64)); e2m = fmax(e2m, __shfl_xor(e2m, 8, 64));
      const double umax = sigj * sqrt(e2m);
      const double own = 0.5 * e2m - logwj;
      act = 0u;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const double* bk = BND + (16 * kt + li) * 3;
        const double t = fmax(0.0, bk[0] - umax);
        const double bnd = fma(bk[2] * t, t, bk[1]) + own;
        if (__any(bnd > -a.cutoff)) act |= 1u << kt;
      }
    }

    // ---- S-step, once per tile for both signs
    mf4 n[KT], nm[(EO && !NML) ? KT : 1];
    if (EO) {
      const double sfc = fma(u2, sfm0, sfm1);   // [|u'|^2, 1, 0, 0] over the lane groups, exactly, without the two selects
      double sfl[QL];
#pragma unroll
      for (int q = 0; q < QL; ++q) sfl[q] = US ? ev[q] : ev[q] * sigj;       // u'_ic (zero beyond D)
      if (C2) {
        sfl[QL - 1] = fma(u2, c2u1, sfl[QL - 1] + c2o1);      // (the dimensions' lanes: + 0)
        if (QS >= 2) sfl[QL >= 2 ? QL - 2 : 0] = fma(u2, c2u2, sfl[QL >= 2 ? QL - 2 : 0]);   // (c2u2 = 0 unless D = 4 QS - 5; unconditional: the
                                                                                               //  compiler if-converts the test into two selects)
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        n[kt] = (mf4){0.0, 0.0, 0.0, 0.0};
        if (!NML) nm[(EO && !NML) ? kt : 0] = (mf4){0.0, 0.0, 0.0, 0.0};
        if (C2) {
#pragma unroll
          for (int q = 0; q < QL; ++q) n[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(SAV(kt, q), sfl[q], n[kt], 0, 0, 0);
          mf4 e2nd;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const double2 pr = *reinterpret_cast<const double2*>(SCP + ENT_CI(kt, rr) * 2);
            e2nd[rr] = fma(pr.x, u2, pr.y) - n[kt][rr];
          }
          if (NML) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) NMS[(kt * 4 + rr) * WAVE + lane] = e2nd[rr];
          } else {
            nm[(EO && !NML) ? kt : 0] = e2nd;
          }
        } else
        if (!SP || ((act >> kt) & 1u)) {
          const mf4 cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(SC[kt], sfc, n[kt], 0, 0, 0);
          n[kt] = cacc;
#pragma unroll
          for (int q = 0; q < QL; ++q)
            n[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(SAV(kt, q), sfl[q], n[kt], 0, 0, 0);
          if (NML) {
            const mf4 e2nd = 2.0 * cacc - n[kt];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) NMS[(kt * 4 + rr) * WAVE + lane] = e2nd[rr];
          } else {
            nm[(EO && !NML) ? kt : 0] = 2.0 * cacc - n[kt];
          }
        }
      }
    }

    // tail components: E+- = even +- linear, TL values per lane (sample li, tail component 4u + lg)
    double ntl[TLN], ntm[TLN];
#pragma unroll
    for (int u = 0; u < TLN; ++u) { ntl[u] = 0.0; ntm[u] = 0.0; }
    if (TL) {
#pragma unroll
      for (int u = 0; u < TLN; ++u) {
        double lt = 0.0;
#pragma unroll
        for (int d = 0; d < DP; ++d) lt = fma(BTL[(4 * u + lg) * DP + d], Et[li * DP + d], lt);    // zero beyond D on both sides
        if (!US) lt *= sigj;
        const double ct = fma(tC0[u], u2, tC1[u]);
        ntl[u] = ct + lt;
        ntm[u] = ct - lt;
      }
    }

    // ---- the phases of one sign as inlined pieces (x: the sign's exponents, overwritten by their exponentials)
    auto exps = [&](mf4 (&x)[KT], auto k0c, auto k1c) {   // k-tiles k0 .. k1-1: 4 straight-line exps each
      constexpr int k0 = decltype(k0c)::value, k1 = decltype(k1c)::value;
#pragma unroll
      for (int kt = k0; kt < k1; ++kt) {
        if (kt < KT - 1) {
          if (!SP || ((act >> kt) & 1u)) x[kt] = vb_exp_tab1k4<VB_EXP_TAB1K_QUAD>(x[kt], TAB);
          else x[kt] = (mf4){0.0, 0.0, 0.0, 0.0};
        } else if (SP && !((act >> (KT - 1)) & 1u)) {
          x[KT - 1] = (mf4){0.0, 0.0, 0.0, 0.0};
        } else if (nr_last == 4) {
          x[KT - 1] = vb_exp_tab1k4<VB_EXP_TAB1K_QUAD>(x[KT - 1], TAB);
        } else {  // registers whose four components are all padding stay exactly zero
          mf4 t = x[KT - 1];
          x[KT - 1] = (mf4){0.0, 0.0, 0.0, 0.0};
          x[KT - 1][0] = vb_exp_tab1k<VB_EXP_TAB1K_QUAD>(t[0], TAB);
          if (nr_last > 1) x[KT - 1][1] = vb_exp_tab1k<VB_EXP_TAB1K_QUAD>(t[1], TAB);
          if (nr_last > 2) x[KT - 1][2] = vb_exp_tab1k<VB_EXP_TAB1K_QUAD>(t[2], TAB);
        }
      }
    };
    auto texp = [&](double (&t)[TLN]) {
      if (TL) {
#pragma unroll
        for (int u = 0; u < TLN; ++u) t[u] = vb_exp_tab1k<VB_EXP_TAB1K_QUAD>(t[u], TAB);
      }
    };
    // PV-step: Y[i][col]; lane (col = li, lg) register rr <-> sample lg + 4 rr.  Two accumulator sets halve the dependent chain.
    // (NPV >= 2: the column blocks are independent chains already, one set is enough -- 16 VGPRs less.)
    auto pvstep = [&](mf4 (&x)[KT], mf4 (&Y)[NPV], int sg, const double (&tn)[TLN]) {
      constexpr bool TWO = NPV == 1 && KT <= 2;     // (round 5: at three k-tiles and more the second set's eight registers are worth more as PV
                                                    //  operands; the dependent MFMAs of one chain issue back to back anyway: 2.138 vs 2.140 ms)
      mf4 Y2s[TWO ? NPV : 1];
      mf4 (&Y2)[TWO ? NPV : 1] = Y2s;
#pragma unroll
      for (int pv = 0; pv < NPV; ++pv) { Y[pv] = (mf4){0.0, 0.0, 0.0, 0.0}; if (TWO) Y2[pv] = (mf4){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
      for (int kt = 0; kt < KT - 1; ++kt) {
        if (SP && !((act >> kt) & 1u)) continue;
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv) {
          mf4& Yb = TWO ? Y2[TWO ? pv : 0] : Y[pv];
          Y[pv] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[kt][0], VBV(kt, 0, pv), Y[pv], 0, 0, 0);
          Yb = __builtin_amdgcn_mfma_f64_16x16x4f64(x[kt][1], VBV(kt, 1, pv), Yb, 0, 0, 0);
          Y[pv] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[kt][2], VBV(kt, 2, pv), Y[pv], 0, 0, 0);
          Yb = __builtin_amdgcn_mfma_f64_16x16x4f64(x[kt][3], VBV(kt, 3, pv), Yb, 0, 0, 0);
        }
      }
#pragma unroll
      for (int pv = 0; pv < NPV; ++pv) {
        mf4& Yb = TWO ? Y2[TWO ? pv : 0] : Y[pv];
        if (SP && !((act >> (KT - 1)) & 1u)) { if (TWO) Y[pv] += Yb; continue; }
        Y[pv] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[KT - 1][0], VBV(KT - 1, 0, pv), Y[pv], 0, 0, 0);
        if (nr_last > 1) Yb = __builtin_amdgcn_mfma_f64_16x16x4f64(x[KT - 1][1], VBV(KT - 1, 1, pv), Yb, 0, 0, 0);
        if (nr_last > 2) Y[pv] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[KT - 1][2], VBV(KT - 1, 2, pv), Y[pv], 0, 0, 0);
        if (nr_last > 3) Yb = __builtin_amdgcn_mfma_f64_16x16x4f64(x[KT - 1][3], VBV(KT - 1, 3, pv), Yb, 0, 0, 0);
        if (TL) {   // the tail: inner index lg <-> tail component 4u + lg
#pragma unroll
          for (int u = 0; u < TLN; ++u) Y[pv] = __builtin_amdgcn_mfma_f64_16x16x4f64(tn[u], VBt[u][pv], Y[pv], 0, 0, 0);
        }
        if (TWO) Y[pv] += Yb;
      }
      if (HV > 1) {
        if (YXSB && sg == 1) __syncthreads();     // every wave has read the first sign's partials
        // all shares of the mixture: the partial q', A', B' of every wave, added in wave order
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) YX[((YXSB ? 0 : sg) * HV + hv) * YXN + (pv * 4 + rr) * WAVE + lane] = Y[pv][rr];
        __syncthreads();
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            double t = YX[((YXSB ? 0 : sg) * HV + 0) * YXN + (pv * 4 + rr) * WAVE + lane];
#pragma unroll
            for (int w = 1; w < HV; ++w) t += YX[((YXSB ? 0 : sg) * HV + w) * YXN + (pv * 4 + rr) * WAVE + lane];
            Y[pv][rr] = t;
          }
      }
    };
    // per-sample scalars in the sample layout (lane <-> sample li): q' from column 0, through LDS
    auto put_q = [&](mf4 (&Y)[NPV]) {
      if (GP2 ? li < 2 : li == 0) {      // column 0: q'_i; GP2: column 1 too, A'_i
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) RQ[(GP2 ? 16 * li : 0) + lg + 4 * rr] = Y[0][rr];
      }
      ent_sync<HV>();   // RQ is private to the wave
    };
    double arq = 0.0;   // GP2: A'_i / q'_i in the sample layout
    // TQ (round 5): the second sign's gradient epilogue reuses the four u'_id the first sign read (-0.2 %; eight registers across the second
    // sign's exponentials, so only in the headline class, where they are there).
    constexpr bool TQ = GP2 && US && NPV == 1 && HV == 1 && EO && KT == 3 && QS <= 3 && !CO && !EM && VBMC_STAG_FOR(KT, QS, TL);
    double tq[TQ ? 4 : 1];
    auto get_rq = [&]() -> double {
      double qs_ = RQ[li];
      double rqs = vb_rcp(qs_);
      if (partial) {
        qs_ = svalid ? qs_ : 1.0;
        rqs = svalid ? rqs : 0.0;
      }
      if (GP2) arq = RQ[GP2 ? 16 + li : 0] * rqs;
      pm *= __builtin_amdgcn_frexp_mant(qs_);   // sum log q' = ln2 * sum exp + log(prod mant)
      pe += __builtin_amdgcn_frexp_exp(qs_);
      accH += shift;
      return rqs;
    };
    auto wacc = [&](mf4 (&x)[KT], double rqs, const double (&tn)[TLN]) {
      if (TL) {
#pragma unroll
        for (int u = 0; u < TLN; ++u) Wt[u] = fma(tn[u], rqs, Wt[u]);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (SP && !((act >> kt) & 1u)) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Wacc[kt][rr] = fma(x[kt][rr], rqs, Wacc[kt][rr]);  // (:100)
      }
    };
    auto put_rq = [&](double rqs) {
      ent_sync<HV>();
      if (lg == 0) { RQ[li] = rqs; if (GP2) RQ[GP2 ? 16 + li : 0] = arq; }
      ent_sync<HV>();
    };
    // gradient pieces in the PV output layout
    auto gradpieces = [&](mf4 (&Y)[NPV], double ssig, auto sgc) {   // sgc: the sign as a compile-time +1 / -1 (US: no multiply), or 0: ssig at run time
      constexpr int SG = decltype(sgc)::value;
      if constexpr (GP2) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int i = lg + 4 * rr;
          const double rq = RQ[i], ar = RQ[GP2 ? 16 + i : 0];     // one broadcast read of the pair
#pragma unroll
          for (int pv = 0; pv < NPV; ++pv) {
            if (HV > 1 && (pv % HV) != hv) continue;      // the waves share the column blocks of the gradient
            const int d = min(max(16 * pv + li - 2, 0), DP - 1);   // columns that are no dimension read a valid slot and are never stored
            double tv;
            if (TQ) { if (SG >= 0) tq[TQ ? rr : 0] = Et[i * DP + d]; tv = tq[TQ ? rr : 0]; }    // (TQ: the second sign reuses the first sign's reads)
            else tv = Et[i * DP + d];
            const double t = (US && SG > 0) ? tv : ((US && SG < 0) ? -tv : ssig * tv);   // u'_id = +-eps_id sigma_j
            const double gd = fma(t, ar, -(Y[pv][rr] * rq));     // lambda_d lsum_d / q = (u'_id A'_i - B'_id) / q'_i  (:77-79)
            accG[pv] += gd;                                      // -> mu_grad (:82)
            accLG[pv] = fma(t, gd, accLG[pv]);                   // -> sigma/lambda grads (:87-93), times sigma_j (divided out at the end)
          }
        }
        ent_sync<HV>();
        return;
      }
      const int base = lane & 48;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = lg + 4 * rr;
        const double Av = __shfl(Y[0][rr], base | 1, 64);    // A'_i  (column 1)
        const double rq = RQ[i];
#pragma unroll
        for (int pv = 0; pv < NPV; ++pv) {
          if (HV > 1 && (pv % HV) != hv) continue;      // the waves share the column blocks of the gradient
          const int d = 16 * pv + li - 2;
          if (d >= 0 && d < D) {
            const double t = (US && SG > 0) ? Et[i * DP + d] : ((US && SG < 0) ? -Et[i * DP + d] : ssig * Et[i * DP + d]);   // u'_id = +-eps_id sigma_j
            const double gd = (t * Av - Y[pv][rr]) * rq;         // lambda_d lsum_d / q  (:77-79)
            accG[pv] += gd;                                      // -> mu_grad (:82)
            accLG[pv] = fma(t, gd, accLG[pv]);                   // -> sigma/lambda grads (:87-93), times sigma_j (divided out at the end)
          }
        }
      }
      ent_sync<HV>();
    };
    // renormalise the mantissa product before it can underflow (0.5^256 = 8.6e-78): its exponent joins the exponent sum.  (Round 4:
    // this used to take the logarithm here -- two instructions now instead of a library log in the tile loop, whose polynomial
    // constants sat in twelve VGPRs of a kernel that has none to spare.)
    auto fold = [&]() {
      if (++pcnt == 256) {
        pe += __builtin_amdgcn_frexp_exp(pm);
        pm = __builtin_amdgcn_frexp_mant(pm);
        pcnt = 0;
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, (KT + 1) / 2>;
    using IK = std::integral_constant<int, KT>;

    if constexpr (EO && GRAD && HV == 1 && VBMC_STAG_FOR(KT, QS, TL)) {
      // Both signs in straight-line code, staggered: the second sign's exponentials (independent of everything the first
      // sign's per-sample chain waits for -- the q' exchange through LDS, the reciprocal, the 1/q' exchange) are issued
      // inside that chain, so this wave keeps the pipe busy across its own latencies instead of leaving them to the one
      // other wave on the SIMD.  No copies on a loop back edge either: each sign works on its own registers.
      mf4 Y[NPV];
      exps(n, I0{}, IK{});
      texp(ntl);
      pvstep(n, Y, 0, ntl);
      put_q(Y);
      exps(nm, I0{}, IH{});
      const double rqs = get_rq();
      put_rq(rqs);
      exps(nm, IH{}, IK{});
      texp(ntm);
      wacc(n, rqs, ntl);
      gradpieces(Y, sigj, std::integral_constant<int, 1>{});
      fold();
      pvstep(nm, Y, 1, ntm);
      put_q(Y);
      const double rqs2 = get_rq();
      wacc(nm, rqs2, ntm);
      put_rq(rqs2);
      gradpieces(Y, -sigj, std::integral_constant<int, -1>{});
      fold();
    } else {
      // one loop body for both signs; the second sign's exponents are MOVED into n on the back edge -- written as a
      // conditional at the loop head the compiler turns them into 32 selects per sign
      int sg = 0;
      double ssig = US ? 1.0 : sigj;          // +-sigma_j (US: +-1, the tile holds the product)
#pragma unroll 1
      for (;;) {
        if (!EO) {   // plain S-step of this sign: KT independent accumulator chains
          double sf[QS];
#pragma unroll
          for (int q = 0; q < QS; ++q) {
            const int cc = 4 * q + lg;
            // the sample-side fragment: kept in registers across the sign loop, or re-read from the LDS tile (2 QS registers
            // less) where the kernel is short of them -- tools/tune_sweep.py: four k-tiles per wave or D >= 27
            constexpr bool EVREG = KT <= 3 && QS <= 7;
            sf[q] = (cc < D) ? ssig * (EVREG ? ev[q] : Et[li * DP + 4 * q + lg]) : ((cc == D) ? u2 : ((cc == D + 1) ? 1.0 : 0.0));
          }
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            n[kt] = (mf4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < QS; ++q) n[kt] = __builtin_amdgcn_mfma_f64_16x16x4f64(SA[kt][q], sf[q], n[kt], 0, 0, 0);
          }
        }
        exps(n, I0{}, IK{});
        texp(ntl);
        if (GRAD) {
          mf4 Y[NPV];
          pvstep(n, Y, sg, ntl);
          put_q(Y);
          const double rqs = get_rq();
          wacc(n, rqs, ntl);
          put_rq(rqs);
          gradpieces(Y, ssig, std::integral_constant<int, 0>{});
        } else {
          double qp = 0.0;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) qp = fma(WF[kt][rr], n[kt][rr], qp);
          if (TL) {
#pragma unroll
            for (int u = 0; u < TLN; ++u) qp = fma(WFt[u], ntl[u], qp);
          }
          qp += __shfl_xor(qp, 16, 64);
          qp += __shfl_xor(qp, 32, 64);
          if (HV > 1) {
            if (YXSB && sg == 1) __syncthreads();
            YX[((YXSB ? 0 : sg) * HV + hv) * YXN + lane] = qp;
            __syncthreads();
            qp = YX[((YXSB ? 0 : sg) * HV + 0) * YXN + lane];
#pragma unroll
            for (int w = 1; w < HV; ++w) qp += YX[((YXSB ? 0 : sg) * HV + w) * YXN + lane];
          }
          double qs_ = qp;
          if (partial) qs_ = svalid ? qp : 1.0;
          pm *= __builtin_amdgcn_frexp_mant(qs_);
          pe += __builtin_amdgcn_frexp_exp(qs_);
          accH += shift;
        }
        fold();
        if (sg) break;
        sg = 1;
        ssig = US ? -1.0 : -sigj;
        if (EO) {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            if (NML) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) n[kt][rr] = NMS[(kt * 4 + rr) * WAVE + lane];
            } else {
              n[kt] = nm[(EO && !NML) ? kt : 0];
            }
          }
        }
#pragma unroll
        for (int u = 0; u < TLN; ++u) ntl[u] = ntm[u];
      }
    }
  };
  if constexpr (VBMC_ENT_SPLIT(KT, QS, TL, HV)) {
    const int tf = min(t1, a.Mh >> 4);        // tiles [t0, tf) hold 16 samples each
    // A wave's issue priority falls as it progresses (s_setprio 3 -> 0 at the quarter points of its tiles): of two waves that share a
    // SIMD the one BEHIND is served first.  At equal priority the arbiter keeps serving the older wave: with a single round of waves
    // (8 restarts per device: 2000 waves on 2048 slots) one wave of each SIMD finished at 240 us, the other at 340, the last 100 us at
    // the single-wave issue rate (tools/archive/r4_timeline.py).  R = 8: 0.337 -> 0.327 ms per step (-3.1 %), R = 16: 0.644 -> 0.633, R = 32:
    // 1.246 -> 1.234, R = 64 (eleven rounds: there is always a younger wave to take over) within 0.2 % either way; the same bits.
    // Halves instead of quarters: a third of the gain.  EntArgs::prio = 0 (VBMC_ENT_PRIO=0): without (A/B).
    // Not in the instantiations built for ONE wave per SIMD (nothing to arbitrate; the thresholds only cost scalar registers there: +0.5-1.1 %
    // at D >= 20, K = 56..64 over the 112-shape sweep, against -3.2 % on average for the shapes of 8..64 components).
    constexpr bool PRIO = CO ? QS <= 4 : VBMC_ENT_WAVES(KT, QS, TL, HV) > 1;
    const bool pr = PRIO && a.prio != 0;
    // (the quarter points of the WAVE's tiles, as tile indices of this segment: one that lies in an earlier segment was passed there, one at
    // this segment's first tile fires there)
    const int q1 = pr ? t0 + (ptot + 3) / 4 - pdone : -1, q2 = pr ? t0 + (ptot + 1) / 2 - pdone : -1, q3 = pr ? t0 + (3 * ptot + 3) / 4 - pdone : -1;
    if (pr && pdone == 0) __builtin_amdgcn_s_setprio(3);
    for (int tile = t0; tile < tf; ++tile) {
      if (tile == q1) __builtin_amdgcn_s_setprio(2);
      if (tile == q2) __builtin_amdgcn_s_setprio(1);
      if (tile == q3) __builtin_amdgcn_s_setprio(0);
      tile_body(tile, EntTileFull{});
    }
    if (tf < t1) tile_body(tf, EntTilePartial{});
  } else {
    // (no priorities here: in the multi-wave kernels the three thresholds cost scalar registers these instantiations do not have --
    // BASELINE configs[4]: kernel alone 8.31 -> 8.64 ms with the code in place and switched off, 8.50 switched on)
    for (int tile = t0; tile < t1; ++tile) tile_body(tile, EntTileAny{});
  }
#ifdef VBMC_INSTRUMENT
  const unsigned long long wck1 = wall_clock64();
#endif
  if (WALK) accH += vb_log_pos(pm, &pe) + 0.693147180559945309417 * (double)pe;     // (no literal constants: device_math.h)
  else accH += log(pm) + 0.693147180559945309417 * (double)pe;
  if (lg != 0) accH = 0.0;   // the four lanes of a sample hold identical copies: count one

  // ---- fixed-order reductions and the partial record
  double* o = part_ + (((size_t)r * K + j) * a.C + c) * a.ncol;
  accH = wave_sum(accH);
  if (lane == 0 && hv == 0) o[0] = accH;
  if (GRAD) {
    double sgsum = 0.0;
#pragma unroll
    for (int pv = 0; pv < NPV; ++pv) {
      double g = accG[pv], lgd = accLG[pv] / sigj;   // accLG carried u' = eps sigma_j in place of eps
      g += __shfl_xor(g, 16, 64); g += __shfl_xor(g, 32, 64);
      lgd += __shfl_xor(lgd, 16, 64); lgd += __shfl_xor(lgd, 32, 64);
      const int d = 16 * pv + li - 2;
      const bool dv = d >= 0 && d < D;
      const bool mine = HV == 1 || (pv % HV) == hv;
      if (dv && lg == 0 && mine) { o[1 + d] = g; o[2 + D + d] = lgd; }
      sgsum += (dv && lg == 0 && mine) ? lgd : 0.0;
    }
    sgsum = wave_sum(sgsum);            // SG = sum_d LG_d  (entmc_vbmc.m:87)
    if (HV > 1) {
      __syncthreads();
      if (lane == 0) YX[hv * YXN] = sgsum;
      __syncthreads();
      sgsum = YX[0];
#pragma unroll
      for (int w = 1; w < HV; ++w) sgsum += YX[w * YXN];
    }
    if (lane == 0 && hv == 0) o[1 + D] = sgsum;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        double wv = Wacc[kt][rr];
        wv += __shfl_xor(wv, 1, 64); wv += __shfl_xor(wv, 2, 64);
        wv += __shfl_xor(wv, 4, 64); wv += __shfl_xor(wv, 8, 64);
        const int k = ENT_CI(kt, rr);
        if (li == 0 && k < Kw) o[2 + 2 * D + kbase + k] = wv;
      }
    if (TL) {
#pragma unroll
      for (int u = 0; u < TLN; ++u) {
        double wv = Wt[u];
        wv += __shfl_xor(wv, 1, 64); wv += __shfl_xor(wv, 2, 64);
        wv += __shfl_xor(wv, 4, 64); wv += __shfl_xor(wv, 8, 64);
        const int k = 16 * KT + 4 * u + lg;
        if (li == 0 && k < Kw) o[2 + 2 * D + kbase + k] = wv;
      }
    }
  }
#ifdef VBMC_INSTRUMENT
  if (lane == 0) {
    const size_t w = ((size_t)r * K + j) * a.C + c;
    if (w < VBMC_DBG_WAVES) {
      g_ent_dbg[6 * w + 0] = wckE; g_ent_dbg[6 * w + 1] = wck0; g_ent_dbg[6 * w + 2] = wck1; g_ent_dbg[6 * w + 3] = wall_clock64();
      g_ent_dbg[6 * w + 4] = __builtin_amdgcn_s_getreg(63492);    // HW_ID
      g_ent_dbg[6 * w + 5] = __builtin_amdgcn_s_getreg(63508);    // XCC_ID
    }
  }
#endif
#undef VBV
#undef SAV
#undef ENT_CI
}

// The kernel.  Chunk grid (WALK = false): workgroup (c, j, r) takes chunk c of component j of restart r -- a.tiles_per_chunk tiles, record
// slot c.  WALK = true (its own instantiations: gradient kernels of single-wave workgroups; grid (waves, 1, 1); a.walk_tpw tiles per wave): the tiles of all (restart, component) pairs form ONE sequence
// (pair p = r K + j holds tiles [p ntile, (p + 1) ntile)), wave w owns [w tpw, (w + 1) tpw) of it and walks its pairs one segment after the
// other; the record of a segment goes to slot (w - first wave of the pair) of the pair's a.C slots, which the reduction counts the same way
// (ent_walk_slots).  With R K C >> wave slots the chunk grid hands every slot a new wave -- a new set-up (staging the parameter block, building
// the operand fragments, the exp table: 8-11 us during which the SIMD's other wave issues alone, at 2/3 of the pair's rate) and an epilogue --
// every chunk; the walk pays them once per (wave, pair): 2.6 instead of 11 per slot at the headline shape, and 4x fewer partial records.
template <int QS, int KT, bool GRAD, bool SPARSE, int HV = 1, int TL = 0, bool CO = false, bool EM = true, bool WALK = false>
__global__ void __launch_bounds__(WAVE * HV, CO ? (QS <= 4 ? 2 : 1) : VBMC_ENT_WAVES(KT, QS, TL, HV)) k_entropy_mfma(EntArgs a) {
  const int ntile = (a.Mh + 15) >> 4;
  // ONE call site for all forms (the body is ~5000 instructions): the chunk grid is a walk of one segment
  constexpr bool walk = WALK;
  int c = (int)blockIdx.x, j = CO ? (int)blockIdx.y - a.lj.rows : (int)blockIdx.y, r = (int)blockIdx.z;
  int tlo = ((int)blockIdx.x + a.c0) * a.tiles_per_chunk, thi = min(tlo + a.tiles_per_chunk, ntile), pdone = 0, rem = 0;
  if (!WALK && a.co_c2 > 0 && c >= a.co_c1) {      // a chunk of the second class (launched without the role: the grid spans both classes)
    tlo = a.co_c1 * a.tiles_per_chunk + (c - a.co_c1) * a.co_tpc2;
    thi = min(tlo + a.co_tpc2, ntile);
  }
  if (CO) {   // workgroup-uniform: the first a.lj.rows grid rows are the log-joint role (>= 8 KB of dynamic LDS: table + rows fit)
    if ((int)blockIdx.y < a.lj.rows) {
      extern __shared__ double PB[];
      lj_co_role<4 * QS>(a.lj, a.vpd, PB);
      // Round 6: ... and then a SHORT chunk of the entropy.  A launch of one or two restarts at Ns = 1e4 is ~2000 entropy waves plus 500-1000 role
      // waves on 2048 wave slots: the role waves went first and a quarter of the entropy waves entered 18 us late, on a 45 us life
      // (tools/ent_timeline.py: 1525 = 6 per compute unit at once, the rest when the role was over).  Now every wave of the launch is
      // resident from the start: the role's workgroups carry the second chunk class (co_tpc2 tiles: a role's length fewer), all leave together.
      const int e = (int)(blockIdx.y * gridDim.x + blockIdx.x);
      if (a.co_c2 <= 0 || e >= a.K * a.co_c2) return;
      j = e / a.co_c2;
      c = a.co_c1 + (e - j * a.co_c2);
      tlo = a.co_c1 * a.tiles_per_chunk + (c - a.co_c1) * a.co_tpc2;
      thi = min(tlo + a.co_tpc2, ntile);
      __syncthreads();      // the role's table and rows: the entropy set-up overwrites them
    }
  }
  if (walk) {   // (the host keeps K R ntile below 2^31)
    const int g = (int)blockIdx.x * a.walk_tpw;
    rem = min(a.walk_tpw, a.K * a.walk_R * ntile - g);
    const int p = g / ntile;
    tlo = g - p * ntile;
    r = p / a.K;
    j = p - r * a.K;
    c = (int)blockIdx.x - (g - tlo) / a.walk_tpw;     // = w - ent_walk_first(p): a pair that begins inside this wave's range has it as its first
  }
  const int ptot = walk ? rem : thi - tlo;
  for (;;) {
    if (walk) thi = min(ntile, tlo + rem);
    ent_mfma_segment<QS, KT, GRAD, SPARSE, HV, TL, CO, EM, WALK>(a, c, j, r, tlo, thi, pdone, ptot);
    if (!walk) break;
    pdone += thi - tlo;
    rem -= thi - tlo;
    if (rem <= 0) break;
    if (++j == a.K) { j = 0; ++r; }
    tlo = 0;
    c = 0;
    __syncthreads();      // the next segment's parameter block overwrites the exp table
  }
}
