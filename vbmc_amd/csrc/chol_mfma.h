// Blocked right-looking Cholesky (upper factor R, R'R = A, in place, strict lower part zeroed), second generation:
// one 512-thread workgroup (8 waves, 256 VGPRs each) per matrix, every O(N^3) and O(N^2) part on v_mfma_f64_16x16x4_f64.
// Reference: gplite/private/gplite_core.m:77-100 ([L,p] = chol(K)), the p > 0 semantics of MATLAB's chol.
//
// Per 16-row block step (kb):
//   panel      R[kb.., j] = inv(Rkk') A[kb.., j] as an MFMA product with the explicit inverse of the 16 x 16 diagonal factor
//              (computed once per step by wave 0 next to the factorisation) instead of a 16-step substitution per column;
//              the tile is fetched straight into the B-operand layout (row = 4q + lane/16, column = lane%16), the product
//              leaves the MFMA as P[t = lane/16 + 4r][j = lane%16] and goes to the LDS panel as one 32-byte vector per lane;
//              one step of iterative refinement (P += W (A - Rkk' P)) restores the accuracy of a substitution.
//   update     A22 -= P'P on the upper triangle of 16 x 16 tiles, groups of CH2_G tiles per wave dealt statically (scalar tile
//              walk): CH2_G x 4 loads, one 32-byte LDS read per operand set, the accumulation chains back to back, stores
//              through SGPR-base + 32-bit-offset addressing.  This phase is bound by what ONE compute unit can stream through
//              the L2 (measured with tools/chol_bench.hip 9: ~65 GB/s for load + store of such tiles), so
//   two panels (TWO, N <= 592: both fit the LDS) the trailing matrix is read and written once per TWO steps: an even step
//              updates only the next row block, the following odd step applies both panels (rank 32) to everything else.
//   look-ahead wave 0 updates the next diagonal tile first, factors it in the accumulator layout (chol_diag_tile3: four
//              pivots at a time, the rank-4 updates inside the tile as one MFMA each) and inverts the factor, while the other
//              waves update; where that chain is the critical path wave 4 (same SIMD) stays idle.
// GP = true: the 16 x Np panel does not fit the LDS (N > 1200) and lives in a global scratch block.
// Measured (tools/chol_bench.hip, MI355X): N = 400: 0.29 ms for one matrix, 0.38 ms for 256 (first generation: 0.52 / 0.66).
#pragma once

#ifndef CH2_THREADS
#define CH2_THREADS 512
#endif
#define CH2_W (CH2_THREADS / 64)
#define CH2_G 4
#ifndef CH2_IDLE4
#define CH2_IDLE4 1
#endif
#ifndef CH2_FAST      // 0: the round-2 diagonal tile (pivot by pivot), stores on the look-ahead's own path
#define CH2_FAST 1
#endif
#ifndef CH2_PREF      // the next diagonal tile fetched before the panel phase: measured, wave 0's four extra loads in front of its panel
#define CH2_PREF 0    // tile cost the whole phase 0.5 us per step and saved 0.15 us of look-ahead (profiles/r05_chol.md) -- off
#endif
#ifndef CH2_X4        // the trailing update on 32 x 16 tile PAIRS with 16-byte accesses (0: round 2's 16 x 16 tiles, 8-byte accesses)
#define CH2_X4 1
#endif
#ifndef CH2_JOIN      // the look-ahead wave joins a large update from its third round of groups on
#define CH2_JOIN 1
#endif
#ifndef CH2_DEFER     // the diagonal tile / block inverse stored in the NEXT step's panel phase instead of on the look-ahead's path
#define CH2_DEFER CH2_FAST
#endif
// LDS doubles: 2 diagonal tiles + 2 inverses + the unit factor of the tile in flight (16 x 17 each) + 64 of gather scratch for the
// tile factorisation + the panel
#define CH2_LDS_FIXED (5 * 16 * 17 + 64)
#define CHOL2_LDS_BYTES(N) ((size_t)(CH2_LDS_FIXED + 16 * (size_t)((((N) + 15) >> 4) << 4)) * sizeof(double))

// sqrt(p) and 1/sqrt(p) from the v_rsq_f64 seed with two Goldschmidt steps and a final residual correction (the sequence of
// the compiler's own sqrt expansion, which also yields the reciprocal root): ~12 dependent operations instead of an IEEE sqrt
// followed by an IEEE division (~80) on the critical path of every pivot.  p is positive and finite here.
__device__ __forceinline__ void chol_sqrt_rsqrt(double p, double& rs, double& ri) {
  const double y = __builtin_amdgcn_rsq(p);
  double g = p * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double d = fma(-g, g, p);
  rs = fma(d, h, g);
  ri = h + h;
  if (__builtin_expect(p < 0x1p-767, 0)) {       // tiny pivot (never on the critical path of a sane matrix): redo it scaled by
    const double ps = __builtin_amdgcn_ldexp(p, 256);   // 2^256 as the library sqrt does -- p * y and the residual went subnormal
    const double y2 = __builtin_amdgcn_rsq(ps);
    g = ps * y2; h = 0.5 * y2;
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double d2 = fma(-g, g, ps);
    rs = __builtin_amdgcn_ldexp(fma(d2, h, g), -128);
    ri = __builtin_amdgcn_ldexp(h, 129);
  }
}

// Lower Cholesky of a 16 x 16 tile held by ONE wave in the MFMA accumulator layout: lane (li, lg), register r holds
// T[i = li][j = lg + 4 r], valid on and below the diagonal (i >= j; identity beyond nb).  Four blocks of four columns:
//   gather   the block's four columns live in the four lane groups (register b): through 512 bytes of LDS every lane gets the
//            four entries of ITS ROW (w[0..3]);
//   factor   four pivots on w: root / reciprocal root of the broadcast pivot (v_readlane, wave-uniform), scale, and the <= 3
//            updates inside the block with the broadcast multiplier -- 6 broadcast-FMA pairs per block instead of 54;
//   scatter  X[b] = w[lg]: every lane already holds all four entries of its row, no cross-lane traffic;
//   update   the rank-4 update of the columns to the right is ONE v_mfma_f64_16x16x4 with X[b] as both operands -- its output
//            layout is the tile's own layout.
// On exit X[r] = L[li][lg + 4 r] (i >= j), Dg[t * 17 + c] = R[t][c] = L[c][t] (upper factor, row stride 17), Di[t] = 1 / R[t][t].
// A non-positive or non-finite pivot records kb + t + 1 in *s_fail (first failure wins) and is replaced by 1.
__device__ __forceinline__ void chol_diag_tile3(double (&X)[4], double* __restrict__ Dg, double* __restrict__ Di, double* __restrict__ scr,
                                                int nb, int kb, int lane, int* s_fail) {
  typedef double d4w __attribute__((ext_vector_type(4)));
  const int li = lane & 15, lg = lane >> 4;
  int fail = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    scr[li * 4 + lg] = X[b];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const d4w wv = *reinterpret_cast<const d4w*>(scr + li * 4);
    double w[4] = {wv[0], wv[1], wv[2], wv[3]};
    __builtin_amdgcn_wave_barrier();             // the next block's writes must not overtake these reads
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const int t = 4 * b + sidx;
      double piv = chol_readlane(w[sidx], t);
      // the validity test runs beside the root chain, not in front of it: a bad pivot only poisons rs / ri, replaced below
      const bool bad = !(piv > 0.0 && piv < __builtin_inf());             // uniform (a broadcast value)
      fail = (bad && t < nb && fail == 0) ? kb + t + 1 : fail;
      double rs, ri;
      chol_sqrt_rsqrt(bad ? 1.0 : piv, rs, ri);
      w[sidx] = (li == t) ? rs : w[sidx] * ri;              // L[li][t] for li > t (rows above the diagonal are never read)
      if (lane == t) Di[t] = ri;
#pragma unroll
      for (int s2 = sidx + 1; s2 < 4; ++s2) {
        const double m = chol_readlane(w[sidx], 4 * b + s2); // L[t'][t]
        w[s2] = fma(-m, w[sidx], w[s2]);
      }
    }
    X[b] = lg == 0 ? w[0] : (lg == 1 ? w[1] : (lg == 2 ? w[2] : w[3]));
    if (b < 3) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[b], X[b], acc, 0, 0, 0);
#pragma unroll
      for (int r = b + 1; r < 4; ++r) X[r] -= acc[r];
    }
  }
  if (fail && lane == 0 && *s_fail == 0) *s_fail = fail;
#pragma unroll
  for (int r = 0; r < 4; ++r) Dg[(lg + 4 * r) * 17 + li] = X[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inverse of the upper factor held in Dg (row stride 17, identity beyond nb): Ri[c][t] = inv(R)[c][t] = inv(R')[t][c].
// Lane c (= lane & 15) solves R' x = e_c by forward substitution; Di[t] = 1 / R[t][t].
__device__ __forceinline__ void chol_tile_inverse(const double* __restrict__ Dg, const double* __restrict__ Di, double* __restrict__ Ri,
                                                  int lane) {
  const int c = lane & 15;
  double r[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    double v = (t == c) ? 1.0 : 0.0;
#pragma unroll
    for (int u = 0; u < t; ++u) v = fma(-Dg[u * 17 + t], r[u], v);
    r[t] = v * Di[t];     // r[u] = 0 for u < c falls out of the recursion
  }
  if (lane < 16) {
#pragma unroll
    for (int t = 0; t < 16; ++t) Ri[c * 17 + t] = r[t];
  }
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------
// Round 5: the same tile as a blocked LDL' -- factor and inverse in ~1.6 us instead of 3.7 (chol_diag_tile3 + chol_tile_inverse
// are a chain of 16 pivots, each a reciprocal-root refinement and two broadcasts behind the previous one, then a 16-step
// substitution that issues 120 LDS reads per lane).  Per block of four columns (register b of the accumulator layout):
//   uniform  the 4 x 4 diagonal block is read straight out of register b (entry (c0 + p, c0 + q) sits in lane c0 + p + 16 q) and
//            eliminated on wave-uniform values: pivots d_s, reciprocals by v_rcp_f64 + two Newton steps (five operations
//            where the reciprocal ROOT took twelve), multipliers -- no square root on the chain at all;
//   rows     every lane takes its row of the block (LDS gather, beside the uniform chain) through the same elimination:
//            u_is = d_s l_is and the unit-factor entry l_is;
//   update   ONE MFMA with A = l (lane group g holds column c0 + g), B = u: L~ D L~' of the block, subtracted from the columns
//            to the right;
//   roots    sqrt(d_s), 1 / sqrt(d_s) (chol_sqrt_rsqrt, four independent chains per block, beside everything else):
//            L[i][c0 + s] = u_is / sqrt(d_s).
// The inverse W = inv(L) = D^-1/2 inv(L~) on the unit factor, four rows at a time: the sums over the earlier blocks split over
// the four lane groups (u = 4 kk + g) and added by permlane swaps, the coupling inside the 4 x 4 diagonal block with uniform
// coefficients -- a chain of ~12 operations per FOUR rows.
// Anything unusual -- a pivot that is not a positive normal number of moderate size -- and the tile is redone by the careful
// routine above from the saved input (returns false; the caller does that): the fast path carries no failure bookkeeping.
#ifdef CHOL_TS   // harness only: shader-clock stamps inside the fast tile routine (look-ahead of step 12 of matrix 0)
__device__ long long g_chol_tt[16];
#define CH2_TSTAMP(slot) do { if (dbg) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if (lane == 0) g_chol_tt[slot] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CH2_TSTAMP(slot) do { } while (0)
#endif
__device__ __forceinline__ double chol_rcp2(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  return fma(y, e, y);
}
// sqrt and reciprocal root without the tiny-pivot branch (the caller guarantees 2^-700 < p < 2^700)
__device__ __forceinline__ void chol_sqrt_rsqrt_nb(double p, double& rs, double& ri) {
  const double y = __builtin_amdgcn_rsq(p);
  double g = p * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double d = fma(-g, g, p);
  rs = fma(d, h, g);
  ri = h + h;
}
__device__ __forceinline__ bool chol_diag_tile_fast(double (&X)[4], double* __restrict__ Dg, double* __restrict__ Di, double* __restrict__ Ri,
                                                    double* __restrict__ scr, double* __restrict__ Lt, int lane, bool dbg = false) {
  typedef double d4w __attribute__((ext_vector_type(4)));
  const int li = lane & 15, lg = lane >> 4;
  bool ok = true;
  (void)dbg;
  CH2_TSTAMP(0);
  double rg[4] = {0.0, 0.0, 0.0, 0.0};          // inverse: M[4 kk + lg][c = li] of the blocks done so far
  double m10 = 0.0, m20 = 0.0, m30 = 0.0, m21 = 0.0, m31 = 0.0, m32 = 0.0;   // block b - 1: its unit-factor entries inside the diagonal block
  // rows 4 k .. 4 k + 3 of the inverse: the sums over the earlier blocks split over the four lane groups (u = 4 kk + lg) and added by
  // permlane swaps, the coupling inside the 4 x 4 diagonal block with the uniform unit-factor entries (rows scaled by 1 / sqrt(d_t) at the end)
  auto inverse_block = [&](int k, double n10, double n20, double n30, double n21, double n31, double n32) {
    const int q = 4 * k;
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      if (kk < k) {
#pragma unroll
        for (int p = 0; p < 4; ++p) s4[p] = fma(Lt[(q + p) * 17 + 4 * kk + lg], rg[kk], s4[p]);
      }
    double bv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const double tot = k == 0 ? 0.0 : xor_sum32(xor_sum16(s4[p]));
      bv[p] = (li == q + p ? 1.0 : 0.0) - tot;
    }
    const double r0 = bv[0];
    const double r1 = fma(-n10, r0, bv[1]);
    const double r2 = fma(-n21, r1, fma(-n20, r0, bv[2]));
    const double r3 = fma(-n32, r2, fma(-n31, r1, fma(-n30, r0, bv[3])));
    const double rsel = lg == 0 ? r0 : (lg == 1 ? r1 : (lg == 2 ? r2 : r3));
    return rsel;
  };
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int c0 = 4 * b;
    // the diagonal block, wave-uniform
    const double a00 = chol_readlane(X[b], c0), a10 = chol_readlane(X[b], c0 + 1), a20 = chol_readlane(X[b], c0 + 2), a30 = chol_readlane(X[b], c0 + 3);
    const double a11 = chol_readlane(X[b], c0 + 1 + 16), a21 = chol_readlane(X[b], c0 + 2 + 16), a31 = chol_readlane(X[b], c0 + 3 + 16);
    const double a22 = chol_readlane(X[b], c0 + 2 + 32), a32 = chol_readlane(X[b], c0 + 3 + 32);
    const double a33 = chol_readlane(X[b], c0 + 3 + 48);
    // this lane's row of the block
    scr[li * 4 + lg] = X[b];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const d4w wv = *reinterpret_cast<const d4w*>(scr + li * 4);
    __builtin_amdgcn_wave_barrier();             // the next block's writes must not overtake these reads
    // uniform elimination of the 4 x 4 block
    const double d0 = a00, y0 = chol_rcp2(d0);
    const double l10 = a10 * y0, l20 = a20 * y0, l30 = a30 * y0;
    const double d1 = fma(-l10, a10, a11), y1 = chol_rcp2(d1);
    const double u21 = fma(-l20, a10, a21), u31 = fma(-l30, a10, a31);
    const double l21 = u21 * y1, l31 = u31 * y1;
    const double d2 = fma(-l21, u21, fma(-l20, a20, a22)), y2 = chol_rcp2(d2);
    const double u32 = fma(-l31, u21, fma(-l30, a20, a32)), l32 = u32 * y2;
    const double d3 = fma(-l32, u32, fma(-l31, u31, fma(-l30, a30, a33))), y3 = chol_rcp2(d3);
    // the rows (the rows of the block itself reproduce the uniform values: same operations)
    const double u0 = wv[0], l0 = u0 * y0;
    const double u1 = fma(-l0, a10, wv[1]), l1 = u1 * y1;
    const double u2 = fma(-l1, u21, fma(-l0, a20, wv[2])), l2 = u2 * y2;
    const double u3 = fma(-l2, u32, fma(-l1, u31, fma(-l0, a30, wv[3]))), l3 = u3 * y3;
    const double lsel = lg == 0 ? l0 : (lg == 1 ? l1 : (lg == 2 ? l2 : l3));
    const double usel = lg == 0 ? u0 : (lg == 1 ? u1 : (lg == 2 ? u2 : u3));
    if (b < 3) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lsel, usel, acc, 0, 0, 0);
#pragma unroll
      for (int r = b + 1; r < 4; ++r) X[r] -= acc[r];
    }
    // the previous block's rows of the inverse, beside this block's chain
    if (b > 0) rg[b - 1] = inverse_block(b - 1, m10, m20, m30, m21, m31, m32);
    Lt[li * 17 + c0 + lg] = lsel;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pivots to LDS (lane group g: d of column c0 + g); the roots of all sixteen are ONE chain after the loop
    if (li == 0) Di[c0 + lg] = lg == 0 ? d0 : (lg == 1 ? d1 : (lg == 2 ? d2 : d3));
    X[b] = usel;                                         // d_s L~[li][c0 + lg], scaled below
    m10 = l10; m20 = l20; m30 = l30; m21 = l21; m31 = l31; m32 = l32;
    CH2_TSTAMP(1 + b);
  }
  rg[3] = inverse_block(3, m10, m20, m30, m21, m31, m32);
  CH2_TSTAMP(5);
  // Roots: lane t < 16 takes pivot t -- sqrt and 1 / sqrt of all sixteen in one chain (the loop carried none), and the range test
  // with them: anything but a positive normal number of moderate size sends the tile to the careful routine.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    const double dt = Di[li];
    ok = dt > 0x1p-200 && dt < 0x1p200;
    double rs, ri;
    chol_sqrt_rsqrt_nb(ok ? dt : 1.0, rs, ri);
    __builtin_amdgcn_wave_barrier();
    if (lg == 0) { Di[li] = ri; scr[li] = rs; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double rik = Di[4 * k + lg], rsk = scr[4 * k + lg];
    X[k] = (li == 4 * k + lg) ? rsk : X[k] * rik;       // L[li][4k + lg] for li >= 4k + lg (rows above the diagonal are never read)
    Ri[li * 17 + 4 * k + lg] = rik * rg[k];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Dg[(lg + 4 * r) * 17 + li] = X[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  CH2_TSTAMP(6);
  return __builtin_amdgcn_ballot_w64(!ok) == 0;
}

#ifdef CHOL_TS   // harness only: shader-clock stamps inside the groups of wave 1 during the first full update (slot x group)
__device__ long long g_chol_gs[8 * 64];
__device__ int g_chol_gn;
#define CH2_GSTAMP(slot) do { if (blockIdx.x == 0 && wave == 1 && kb == (TWO ? 16 : 0) && gcount < 64) { __builtin_amdgcn_sched_barrier(0); \
    if ((slot) == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    if ((slot) == 3) asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory"); \
    g_chol_gs[8 * gcount + (slot)] = clock64(); __builtin_amdgcn_sched_barrier(0); if ((slot) == 4) { ++gcount; g_chol_gn = gcount; } } } while (0)
// shader-clock stamps inside the look-ahead of wave 0 at step 12
__device__ long long g_chol_la[8];
#define CH2_LSTAMP(slot) do { if (blockIdx.x == 0 && kb == 12 * 16) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if (lane == 0) g_chol_la[slot] = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CH2_GSTAMP(slot) do { } while (0)
#define CH2_LSTAMP(slot) do { } while (0)
#endif

// factor + invert one diagonal tile (wave-level): the fast path, the careful one when it declines
__device__ __forceinline__ void chol_tile(double (&X)[4], double* __restrict__ Dg, double* __restrict__ Di, double* __restrict__ Ri,
                                          double* __restrict__ scr, double* __restrict__ Lt, int nb, int kb, int lane, int* s_fail) {
  if (CH2_FAST) {
    double X0[4] = {X[0], X[1], X[2], X[3]};
#ifdef CHOL_TS
    const bool dbg = blockIdx.x == 0 && kb == 13 * 16;
#else
    const bool dbg = false;
#endif
    if (chol_diag_tile_fast(X, Dg, Di, Ri, scr, Lt, lane, dbg)) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = X0[r];
  }
  chol_diag_tile3(X, Dg, Di, scr, nb, kb, lane, s_fail);
  chol_tile_inverse(Dg, Di, Ri, lane);
}

template <bool GP, bool TWO>
__global__ void __launch_bounds__(CH2_THREADS) k_chol2(int N, double* __restrict__ Aall, int* __restrict__ pfail,
                                                       const unsigned char* __restrict__ active, double* __restrict__ Pg,
                                                       double* __restrict__ Finv, double* __restrict__ pfd,
                                                       const double* __restrict__ rin, double* __restrict__ zout) {
  extern __shared__ __attribute__((aligned(32))) double lds_c2[];   // 32-byte vectors of the panel
  double* lds = lds_c2;
  const int s = blockIdx.x;
  if (!active[s]) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform in an SGPR: tile walks and tile base addresses stay scalar
  const int li = lane & 15, lg = lane >> 4;
  const int Np = ((N + 15) >> 4) << 4;
  double* A = Aall + (size_t)s * N * N;
  double* DgB = lds;                      // [2][16 x 17] diagonal tiles: this step's and the next one's
  double* RiB = lds + 2 * 16 * 17;        // [2][16 x 17] their inverses
  double* scr = lds + 4 * 16 * 17;        // 64 doubles (chol_diag_tile3 / chol_diag_tile_fast)
  double* LtS = scr + 64;                 // 16 x 17: the unit factor of the tile in flight (chol_diag_tile_fast)
  double* Pl = lds + CH2_LDS_FIXED;       // 16 x Np panel rows (LDS variant); TWO: a second panel behind it
  double* Pgl = GP ? Pg + (size_t)s * 16 * Np : nullptr;
  // Right-hand side riding along (rin != null): z = R' \ r as a by-product -- r is one more column of the matrix: its row block
  // takes the panel operation (z_b = inv(Rkk') r_b, with the same refinement step), the rest of it the trailing update
  // (r -= P' z_b).  The vector lives behind the panel(s) in LDS; alpha = R \ z is then half of the two-sided solve.
  double* rv = Pl + (GP ? 0 : (TWO ? 32 : 16) * Np);    // Np doubles
  __shared__ double zb[16], zres[16];
  const bool RHS = rin != nullptr;
  if (RHS) for (int i = tid; i < Np; i += CH2_THREADS) rv[i] = i < N ? rin[(size_t)s * N + i] : 0.0;
  // z_b for the block at kb from the tile in buffer `c_`: one wave, lane (t = li, g = lg) takes the terms c = 4 j + g of row t's three
  // 16-term products, the four groups are added by permlane swaps
  auto rhs_block = [&](int kb_, int c_) {
    const double* Ri_ = RiB + c_ * 16 * 17;
    const double* Dg_ = DgB + c_ * 16 * 17;
    const int t = li;
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) a = fma(Ri_[(4 * j + lg) * 17 + t], rv[kb_ + 4 * j + lg], a);
    double z = xor_sum32(xor_sum16(a));
    if (lg == 0) zb[t] = z;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // refinement: res_t = r_t - sum_{u <= t} R[u][t] z_u,  z += inv(Rkk') res
    double p = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int u = 4 * j + lg;
      p = fma(u <= t ? Dg_[u * 17 + t] : 0.0, zb[u], p);
    }
    const double res = rv[kb_ + t] - xor_sum32(xor_sum16(p));
    if (lg == 0) zres[t] = res;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double c = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c = fma(Ri_[(4 * j + lg) * 17 + t], zres[4 * j + lg], c);
    z += xor_sum32(xor_sum16(c));
    if (lg == 0) {
      zb[t] = z;
      if (kb_ + t < N) zout[(size_t)s * N + kb_ + t] = z;
    }
  };
  static_assert(!(GP && TWO), "the two-panel scheme keeps both panels in LDS");
  __shared__ int s_fail;
  __shared__ double DiB[2][16];
  // Panel layout: row t = 4 q + g of the 16 x Np panel lives at ((g * Np + col) * 4 + q): the four q-slices an MFMA operand
  // needs for one column are 32 contiguous bytes (one vector read per operand set, one vector write per product column).
  typedef double d4v __attribute__((ext_vector_type(4)));
  auto ldP4 = [&](int buf, int g, int col) -> d4v {        // buf: 0 / 1 = first / second panel (TWO only)
    return GP ? *(const d4v*)(Pgl + (((size_t)g * Np + col) << 2)) : *(const d4v*)(Pl + buf * 16 * Np + ((g * Np + col) << 2));
  };
  auto stP4 = [&](int buf, int g, int col, d4v v) {
    if (GP) *(d4v*)(Pgl + (((size_t)g * Np + col) << 2)) = v; else *(d4v*)(Pl + buf * 16 * Np + ((g * Np + col) << 2)) = v;
  };
  if (tid == 0) s_fail = 0;
  __syncthreads();
  // the tile in the accumulator layout from the UPPER triangle of the symmetric matrix: T[i = li][j] = A[j][i] for i >= j
  auto load_diag = [&](const double* Ad, int nbt, double (&X)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = lg + 4 * r;
      const bool ok = li < nbt && j < nbt && li >= j;
      const double v = Ad[ok ? (size_t)li * N + j : 0];
      X[r] = ok ? v : (li == j ? 1.0 : 0.0);
    }
  };
  auto store_diag = [&](double* Ad, int nbt, const double* Dg) {
    for (int e = lane; e < 256; e += 64) {
      const int ii = e >> 4, jj = e & 15;
      if (ii < nbt && jj < nbt) Ad[(size_t)ii + (size_t)N * jj] = (ii <= jj) ? Dg[ii * 17 + jj] : 0.0;
    }
  };
  // Finv (optional): the inverse of every diagonal block goes out in the layout the triangular solves read (trsm_mfma.h
  // k_diag_inv: Finv[blk][ii][c] = inv(R_bb')[ii][c], row-major 16 x 16, identity beyond N) -- the factorisation has it anyway
  double* Fo = Finv ? Finv + (size_t)s * (Np >> 4) * 256 : nullptr;
  auto store_finv = [&](int blk, const double* Ri_) {
    if (Fo) {
#pragma unroll
      for (int e = lane; e < 256; e += 64) Fo[(size_t)blk * 256 + e] = Ri_[(e & 15) * 17 + (e >> 4)];
    }
  };
  if (wave == 0) {
    const int nb = min(16, N);
    double X[4];
    load_diag(A, nb, X);
    chol_tile(X, DgB, DiB[0], RiB, scr, LtS, nb, 0, lane, &s_fail);
    if (!CH2_DEFER) { store_diag(A, nb, DgB); store_finv(0, RiB); }
  }
  __syncthreads();
  int cur = 0;
  int gcount = 0; (void)gcount;
  for (int kb = 0; kb < N; kb += 16, cur ^= 1) {
    if (s_fail) break;            // uniform: written before the last barrier
    const int nb = min(16, N - kb);
    const int t0 = kb + nb;       // first trailing column
    const int ntr = N - t0;
    if (ntr <= 0) break;
    const int nt = (ntr + 15) >> 4;
    const double* Ri = RiB + cur * 16 * 17;
    // TWO: the trailing matrix is read and written once per TWO block steps.  Even step (A): panel A, the look-ahead, and the
    // update of the next row block only (what panel B is made of).  Odd step (B): panel B, then every remaining tile takes the
    // rank-32 update of both panels in one pass -- the update is bound by the L2 bandwidth of one CU, and this halves its traffic.
    const int pbuf = TWO ? ((kb >> 4) & 1) : 0;
    if (tid == 0) CHOL_STAMP(0, kb >> 4);
    // ---- panel: tile tj of the row block (16 x 16, rows kb.., columns t0 + 16 tj..) times inv(Rkk')
    // the next diagonal tile is complete up to this step's panel: wave 0 fetches it now, the L2 round trip hides behind the panel phase
    double Xn[4] = {0.0, 0.0, 0.0, 0.0};
    if (CH2_PREF && wave == 0) load_diag(A + (size_t)t0 + (size_t)N * t0, min(16, ntr), Xn);
    {
      double av[4], rv[4];
      const double* Dgc = DgB + cur * 16 * 17;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        av[q] = Ri[(4 * q + lg) * 17 + li];                                // A operand: inv(Rkk')[t = li][u = 4q + lg]
        rv[q] = (4 * q + lg <= li) ? Dgc[(4 * q + lg) * 17 + li] : 0.0;    // A operand: Rkk'[t = li][u = 4q + lg] (lower triangular)
      }
      constexpr int PT = CH2_W >= 16 ? 2 : 3;                              // tiles in flight per wave
      // the row block through a wave-uniform base + one 32-bit byte offset per element (N <= 9696: N^2 * 8 < 2^32): with 64-bit
      // addresses per element the compiler, short of registers in this kernel, loaded INTO the address registers and waited for
      // each load in turn (up to four L2 round trips per tile triple, +0.5 us on every panel phase; profiles/r05_chol.md)
      // (buffer accesses: base and extent in four SGPRs, ONE VGPR of offset per element, an offset past the extent reads 0 /
      // drops the store -- no masks on the data, no branches around the stores)
      const __amdgpu_buffer_rsrc_t Arow = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(A + (size_t)kb + (size_t)N * t0), 0, (int)(((size_t)N * N - ((size_t)kb + (size_t)N * t0)) * 8), 0x00020000);
      for (int tb = wave; tb < nt; tb += PT * CH2_W) {
        double bv[PT][4];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          const int tj = tb + p * CH2_W, j = (tj << 4) + li;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int u = 4 * q + lg;
            const bool ok = tj < nt && u < nb && j < ntr;
            const unsigned off = ok ? (unsigned)((u + N * j) * 8) : 0xfffffff0u;
            bv[p][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Arow, (int)off, 0, 0));
          }
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          const int tj = tb + p * CH2_W, j = (tj << 4) + li;
          if (tj < nt) {                                                   // wave-uniform
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[p][q], acc, 0, 0, 0);
            // One step of iterative refinement: the product with the explicit inverse is only accurate to cond(Rkk) eps (2e-8 in
            // alpha on a kernel matrix of condition 1e7, found by the random-shape sweep), the substitution it replaces was
            // backward stable.  res = A - Rkk' P (register q of the accumulator layout IS k-slice q of a B operand), P += W res.
            d4_t res = {bv[p][0], bv[p][1], bv[p][2], bv[p][3]};
#pragma unroll
            for (int q = 0; q < 4; ++q) res = __builtin_amdgcn_mfma_f64_16x16x4f64(-rv[q], acc[q], res, 0, 0, 0);
            {
              const d4_t rr = res;
#pragma unroll
              for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], rr[q], acc, 0, 0, 0);
            }
            stP4(pbuf, lg, j, acc);                                        // acc[r] = R[kb + lg + 4r][t0 + j]; zero for rows >= nb, j >= ntr
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int t = lg + 4 * r;
              const unsigned off = (t < nb && j < ntr) ? (unsigned)((t + N * j) * 8) : 0xfffffff0u;
              typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, (double)acc[r]), Arow, (int)off, 0, 0);
            }
          }
        }
      }
    }
    // behind this wave's panel tiles (their loads and stores are in flight): the right-hand side's block on the wave with the
    // fewest tiles; this block's diagonal tile and its inverse go out here, off the look-ahead's path (the tile buffers hold
    // them for a step)
    if (RHS && wave == CH2_W - 1) rhs_block(kb, cur);
    if (CH2_DEFER && wave == CH2_W - 2) { store_diag(A + (size_t)kb + (size_t)N * kb, nb, DgB + cur * 16 * 17); store_finv(kb >> 4, RiB + cur * 16 * 17); }
    // Nothing this phase wrote to GLOBAL memory is read before the barrier at the end of the step (the row block of R, the diagonal
    // tile, the block inverse and z are results; the panel the update reads is in LDS): an LDS-only barrier, so that the stores'
    // acknowledgements (1-2 us) are waited for behind the update instead of in front of it.  (GP: the panel itself is in global memory.)
    if (GP || !CH2_DEFER) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0) CHOL_STAMP(1, kb >> 4);
    // ---- trailing update.  Tile pair u = tj (tj + 1) / 2 + ti (ti <= tj) of the nt x nt tile triangle; transposed tiles so
    //      that lanes run along i (contiguous in the column-major matrix): lane (li, lg) register r holds
    //      A[t0 + 16 ti + li][t0 + 16 tj + lg + 4 r]  -=  sum_t P[t][16 tj + lg + 4 r] P[t][16 ti + li].
    const int npair = nt * (nt + 1) / 2;
    double* At = A + (size_t)t0 + (size_t)N * t0;     // trailing matrix; 32-bit offsets inside it (N <= 3872)
    if (wave == 0) {
      // look-ahead: next diagonal tile (pair 0)
      const int nb2 = min(16, ntr);
      double* Dn = DgB + (cur ^ 1) * 16 * 17;
      CH2_LSTAMP(0);
      double X[4];
      if (CH2_PREF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) X[r] = Xn[r];
      } else load_diag(At, nb2, X);
      CH2_LSTAMP(1);
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      const d4v pa = ldP4(pbuf, lg, li);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[q], pa[q], acc, 0, 0, 0);
      if (TWO && pbuf == 1) {                 // B step: the tile has not seen panel A yet (its columns sit 16 further in A's panel)
        const d4v pa0 = ldP4(0, lg, 16 + li);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa0[q], pa0[q], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = lg + 4 * r;
        if (li < nb2 && j < nb2 && li >= j) X[r] -= acc[r];     // the product is symmetric: either triangle of it will do
      }
      CH2_LSTAMP(2);
      chol_tile(X, Dn, DiB[cur ^ 1], RiB + (cur ^ 1) * 16 * 17, scr, LtS, nb2, t0, lane, &s_fail);
      CH2_LSTAMP(3);
      CH2_LSTAMP(4);
      if (!CH2_DEFER) { store_diag(At, nb2, Dn); store_finv(t0 >> 4, RiB + (cur ^ 1) * 16 * 17); }
      CH2_LSTAMP(5);
      if (lane == 0) CHOL_STAMP(2, kb >> 4);
    }
    // Wave 0 (look-ahead) is a long chain of dependent fp64 VALU operations, and an fp64 MFMA of another wave on the same SIMD
    // blocks its issue: where the look-ahead is the critical path (small updates, A steps) wave 4, which shares SIMD 0 with it,
    // stays out of the update; a large update hides the look-ahead anyway and wants all four matrix pipes.
    if (RHS && wave > 0) {
      // the rest of the vector takes this step's panel: r[t0 + col] -= sum_t P[t][col] z_b[t]
      for (int col = tid - 64; col < ntr; col += CH2_THREADS - 64) {
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const d4v pv = ldP4(pbuf, g, col);
#pragma unroll
          for (int q = 0; q < 4; ++q) a4[g] = fma(pv[q], zb[4 * q + g], a4[g]);
        }
        rv[t0 + col] -= (a4[0] + a4[1]) + (a4[2] + a4[3]);
      }
    }
    const bool idle4 = CH2_IDLE4 && !((!TWO || pbuf == 1) && npair > 48);
    const int UW = CH2_W - (idle4 ? 2 : 1);                              // waves that update
    const int uslot = (idle4 && wave > 4) ? wave - 2 : wave - 1;         // their index 0 .. UW - 1
    // Round 5: in a large update the look-ahead wave does not sit out the rest of the phase: from the third round of groups on
    // (its look-ahead takes about two) wave 0 is one more slot of the deal.
    const bool joins = CH2_X4 && CH2_JOIN && !idle4 && wave == 0;
    if ((wave > 0 && !(idle4 && wave == 4)) || joins) {
      // one tile with masks: ragged edges, diagonal tiles of the slow path, the row block of an A step
      auto tile_masked = [&](int a, int b, bool both) {
        const int i0 = a << 4, j0 = b << 4, i = i0 + li, jb = j0 + lg;
        double c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = jb + 4 * r;
          const bool ok = i < ntr && j < ntr;
          c[r] = At[ok ? j * N + i : 0];
        }
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        {
          const d4v pj = ldP4(pbuf, lg, j0 + li), pi = ldP4(pbuf, lg, i0 + li);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[q], pi[q], acc, 0, 0, 0);
        }
        if (TWO && both) {
          const d4v pj = ldP4(0, lg, 16 + j0 + li), pi = ldP4(0, lg, 16 + i0 + li);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[q], pi[q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = jb + 4 * r;
          if (i < ntr && j < ntr && i <= j) At[j * N + i] = c[r] - acc[r];
        }
      };
      if (TWO && pbuf == 0) {
        // A step: only the next row block, tiles (0, tj), tj = 1 .. nt - 1 (the diagonal tile is the look-ahead's)
        for (int tjj = 1 + uslot; tjj < nt; tjj += UW) tile_masked(0, tjj, false);
      } else {
      // Static deal: group n of update wave w' = pairs 1 + CH2_G (w' + UW n) .. + CH2_G - 1, walked incrementally (an LDS
      // work counter was measured: the atomic's round trip at the head of every group cost more than the imbalance it removes).
      // FAST groups lie entirely in full tile columns (16 (tj + 1) <= ntr): no masks -- the entries below the diagonal of a
      // diagonal tile are updated like the rest (they hold whatever the builder left there, nobody reads them, and the
      // strict lower triangle is zeroed at the end).
      const int ntf = ntr >> 4;
      if (CH2_X4) {
      // Round 5: ONE compute unit streams 16 x 16 tiles through the L2 at 65 GB/s with 8-byte accesses and at 125-135 GB/s with
      // 16-byte accesses (tools/chol_bench.hip 9: the memory pipe's rate is per instruction) -- and this phase is that stream.  A
      // lane now takes rows (2 li, 2 li + 1) of a column of a 32 x 16 tile PAIR (tiles ti = 2 p, 2 p + 1 of block column tj) in one
      // 16-byte access: the even rows are one MFMA tile, the odd rows the other -- the row permutation only changes which panel
      // column a lane feeds as the B operand (i0 + 2 li, i0 + 2 li + 1).  Block column tj has (tj + 1) / 2 pairs; with tj even its
      // diagonal tile is left over and goes tile by tile (as does a ragged last column).  Pairs v = 0, 1, ... run down the
      // columns from column 1 (column 0 is the look-ahead's tile); H(2 m) = m^2, H(2 m + 1) = m (m + 1) pairs lie before a column.
      typedef double d2x __attribute__((ext_vector_type(2), aligned(8)));
      const int vfast = (ntf & 1) ? (ntf >> 1) * ((ntf >> 1) + 1) : (ntf >> 1) * (ntf >> 1);     // H(ntf): pairs in the full columns
      unsigned lob2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) lob2[r] = (unsigned)(((lg + 4 * r) * N + 2 * li) * 8);
      const char* Ab = reinterpret_cast<const char*>(At);
      const __amdgpu_buffer_rsrc_t Atr = __builtin_amdgcn_make_buffer_rsrc(
          (void*)At, 0, (int)(((size_t)N * N - ((size_t)t0 + (size_t)N * t0)) * 8), 0x00020000);
      const int PSTR = 2 * UW;                                   // two pairs (four tiles' worth of registers) per wave and round
      const bool late = CH2_JOIN && !idle4;                     // rounds 0, 1: UW slots; from round 2 on UW + 1 (wave 0 = slot UW)
      int v0 = joins ? 2 * 3 * UW : 2 * uslot, pp, tj;
      {
        int m = (int)sqrtf((float)v0);
        m += ((m + 1) * (m + 1) <= v0) ? 1 : 0;
        m -= (m * m > v0) ? 1 : 0;
        if (v0 < m * (m + 1)) { tj = 2 * m; pp = v0 - m * m; } else { tj = 2 * m + 1; pp = v0 - m * (m + 1); }
      }
      auto padvance = [&](int by) { pp += by; while (pp >= ((tj + 1) >> 1)) { pp -= (tj + 1) >> 1; ++tj; } };
      auto pgroup = [&]() {
        d2x c[2][4];
        unsigned ob[2][4];
        int i0_[2], j0_[2];
        CH2_GSTAMP(0);
        {
          int a = pp, b = tj;
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            i0_[g] = a << 5; j0_[g] = b << 4;
            const unsigned tpb = (unsigned)((j0_[g] * N + i0_[g]) * 8);   // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ob[g][r] = tpb + lob2[r];
              c[g][r] = *reinterpret_cast<const d2x*>(Ab + ob[g][r]);
            }
            const bool wrap = a + 1 >= ((b + 1) >> 1);
            a = wrap ? 0 : a + 1;
            b += wrap ? 1 : 0;
          }
        }
        CH2_GSTAMP(1);
        d4_t ae[2], ao[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) { ae[g] = (d4_t){0.0, 0.0, 0.0, 0.0}; ao[g] = (d4_t){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
        for (int pn = 0; pn < (TWO ? 2 : 1); ++pn) {
          const int buf = TWO ? 1 - pn : 0, sh = (TWO && pn == 1) ? 16 : 0;   // panel B first (its columns start at this step's t0)
          d4v pj[2], pe[2], po[2];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            pj[g] = ldP4(buf, lg, sh + j0_[g] + li);
            pe[g] = ldP4(buf, lg, sh + i0_[g] + 2 * li);
            po[g] = ldP4(buf, lg, sh + i0_[g] + 2 * li + 1);
          }
          if (pn == 0) CH2_GSTAMP(2);
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ae[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[g][q], pe[g][q], ae[g], 0, 0, 0);
              ao[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[g][q], po[g][q], ao[g], 0, 0, 0);
            }
        }
        CH2_GSTAMP(3);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            d2x v = c[g][r];
            v[0] -= ae[g][r]; v[1] -= ao[g][r];
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, v), Atr, (int)ob[g][r], 0, 0);   // (SGPR base: no 64-bit address per store)
          }
        CH2_GSTAMP(4);
      };
      for (int n = joins ? 2 : 0; v0 + 2 <= vfast; ++n) {
        pgroup();
        const int by = (late && n >= 2) ? PSTR + 2 : PSTR;
        v0 += by; padvance(by);
      }
      // left over, tile by tile with masks, dealt round-robin: a last pair of the full columns (vfast odd), the diagonal tiles of the
      // even full columns (2, 4, ...; column 0's is the look-ahead's), every tile of a ragged last column
      {
        int k = 0;
        if (vfast & 1) {
          if ((k++ % UW) == uslot) {
            const int vl = vfast - 1;
            int m = (int)sqrtf((float)vl);
            m += ((m + 1) * (m + 1) <= vl) ? 1 : 0;
            m -= (m * m > vl) ? 1 : 0;
            int tjl, pl;
            if (vl < m * (m + 1)) { tjl = 2 * m; pl = vl - m * m; } else { tjl = 2 * m + 1; pl = vl - m * (m + 1); }
            tile_masked(2 * pl, tjl, true);
            tile_masked(2 * pl + 1, tjl, true);
          }
        }
        for (int c2 = 2; c2 < ntf; c2 += 2)
          if ((k++ % UW) == uslot) tile_masked(c2, c2, true);
        if (ntr & 15)
          for (int a = 0; a <= ntf; ++a)
            if ((k++ % UW) == uslot) tile_masked(a, ntf, true);
      }
      } else {
      const int STRIDE = CH2_G * UW;
      const int ufast = ntf * (ntf + 1) / 2;
      unsigned lob[4];                          // byte offset of this lane's element in register r of a tile
#pragma unroll
      for (int r = 0; r < 4; ++r) lob[r] = (unsigned)(((lg + 4 * r) * N + li) * 8);
      const char* Ab = reinterpret_cast<const char*>(At);
      int u0 = 1 + CH2_G * uslot;
      int ti, tj;                               // pair u0 = tj (tj + 1) / 2 + ti
      {
        int c = (int)((sqrtf(8.0f * (float)u0 + 1.0f) - 1.0f) * 0.5f);
        c += ((c + 1) * (c + 2) / 2 <= u0) ? 1 : 0;
        c -= (c * (c + 1) / 2 > u0) ? 1 : 0;
        tj = c; ti = u0 - c * (c + 1) / 2;
      }
      auto advance = [&](int by) { ti += by; while (ti > tj) { ti -= tj + 1; ++tj; } };
      // One fast group: CH2_G x 4 loads, the operand reads, the CH2_G accumulation chains back to back (per panel), CH2_G x 4
      // stores from the registers the loads filled through one 32-bit byte offset each (SGPR base + VGPR offset addressing).
      auto group = [&]() {
        double c[CH2_G][4];
        unsigned ob[CH2_G][4];
        int i0_[CH2_G], j0_[CH2_G];
        CH2_GSTAMP(0);
        {
          int a = ti, b = tj;
#pragma unroll
          for (int g = 0; g < CH2_G; ++g) {
            i0_[g] = a << 4; j0_[g] = b << 4;
            const unsigned tpb = (unsigned)((j0_[g] * N + i0_[g]) * 8);   // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ob[g][r] = tpb + lob[r];
              c[g][r] = *reinterpret_cast<const double*>(Ab + ob[g][r]);
            }
            const bool wrap = a == b;
            a = wrap ? 0 : a + 1;
            b += wrap ? 1 : 0;
          }
        }
        CH2_GSTAMP(1);
        d4_t acc[CH2_G];
#pragma unroll
        for (int g = 0; g < CH2_G; ++g) acc[g] = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pn = 0; pn < (TWO ? 2 : 1); ++pn) {
          const int buf = TWO ? 1 - pn : 0, sh = (TWO && pn == 1) ? 16 : 0;   // panel B first (its columns start at this step's t0)
          d4v pj[CH2_G], pi[CH2_G];
#pragma unroll
          for (int g = 0; g < CH2_G; ++g) { pj[g] = ldP4(buf, lg, sh + j0_[g] + li); pi[g] = ldP4(buf, lg, sh + i0_[g] + li); }
          if (pn == 0) CH2_GSTAMP(2);
#pragma unroll
          for (int g = 0; g < CH2_G; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(pj[g][q], pi[g][q], acc[g], 0, 0, 0);
        }
        CH2_GSTAMP(3);
#pragma unroll
        for (int g = 0; g < CH2_G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            c[g][r] -= acc[g][r];
            *reinterpret_cast<double*>(const_cast<char*>(Ab) + ob[g][r]) = c[g][r];
          }
        CH2_GSTAMP(4);
      };
      for (; u0 + CH2_G <= ufast; u0 += STRIDE, advance(STRIDE)) group();
      // the rest: groups that touch the ragged last tile column or run past the last pair, tile by tile with masks
      for (; u0 < npair; u0 += STRIDE, advance(STRIDE)) {
        int a = ti, b = tj;
        for (int g = 0; g < CH2_G && u0 + g < npair; ++g) {
          tile_masked(a, b, true);
          const bool wrap = a == b;
          a = wrap ? 0 : a + 1;
          b += wrap ? 1 : 0;
        }
      }
      }
      }
    }
    __syncthreads();
    if (tid == 0) CHOL_STAMP(3, kb >> 4);
  }
  if (RHS && wave == 0 && !s_fail) rhs_block(((N - 1) >> 4) << 4, cur);   // the last row block has no panel phase
  if (CH2_DEFER && wave == 1 && !s_fail) {
    const int kl = ((N - 1) >> 4) << 4;
    store_diag(A + (size_t)kl + (size_t)N * kl, N - kl, DgB + cur * 16 * 17);
    store_finv(kl >> 4, RiB + cur * 16 * 17);
  }
  if (tid == 0) { pfail[s] = s_fail; if (pfd) pfd[s] = (double)s_fail; }
  if (s_fail) return;
  // zero the strict lower triangle (MATLAB chol returns an upper-triangular matrix)
  for (int j = wave; j < N; j += CH2_W)
    for (int i = j + 1 + lane; i < N; i += 64) A[(size_t)i + (size_t)N * j] = 0.0;
}

// Launch on stream st: S matrices of order N in dA (N x N x S), pfail S ints, active S flags; Pg = S x 16 x Np doubles of
// scratch, needed only when chol2_needs_gpanel(N).  Two panels in LDS up to N = 592, one up to N = 1200, global panel beyond
// (with a right-hand side riding along -- Np more doubles of LDS -- up to N = 560 / 1104).
#define CHOL2_NP(N) ((size_t)((((N) + 15) >> 4) << 4))
#define CHOL2_LDS_BYTES1(N, rhs) ((size_t)(CH2_LDS_FIXED + (16 + ((rhs) ? 1 : 0)) * CHOL2_NP(N)) * sizeof(double))
#define CHOL2_LDS_BYTES2(N, rhs) ((size_t)(CH2_LDS_FIXED + (32 + ((rhs) ? 1 : 0)) * CHOL2_NP(N)) * sizeof(double))
static inline bool chol2_needs_gpanel(int N, bool rhs = false) { return CHOL2_LDS_BYTES1(N, rhs) > 159 * 1024; }   // + ~0.5 KB of static LDS
// dFinv (optional): S x nblk x 256 inverses of the diagonal blocks (what k_diag_inv would compute from the factor);
// dpfd (optional): the failure indices once more as doubles (so that they can ride in a block of results);
// drin / dzout (optional, both or neither): S x N right-hand sides r and the forward solves z = R' \ r
static inline hipError_t chol2_launch(int N, int S, double* dA, int* dpf, const unsigned char* dact, double* dPg, hipStream_t st,
                                      double* dFinv = nullptr, double* dpfd = nullptr, const double* drin = nullptr,
                                      double* dzout = nullptr) {
  const bool rhs = drin != nullptr;
  if (rhs != (dzout != nullptr)) return hipErrorInvalidValue;
  if (chol2_needs_gpanel(N, rhs)) {
    const size_t lds = (size_t)(CH2_LDS_FIXED + (rhs ? CHOL2_NP(N) : 0)) * sizeof(double);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)k_chol2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_chol2<true, false>), dim3(S), dim3(CH2_THREADS), lds, st, N, dA, dpf, dact, dPg, dFinv, dpfd, drin, dzout);
  } else if (CHOL2_LDS_BYTES2(N, rhs) > 159 * 1024) {
    const size_t lds = CHOL2_LDS_BYTES1(N, rhs);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)k_chol2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_chol2<false, false>), dim3(S), dim3(CH2_THREADS), lds, st, N, dA, dpf, dact, (double*)nullptr, dFinv, dpfd, drin, dzout);
  } else {
    const size_t lds = CHOL2_LDS_BYTES2(N, rhs);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)k_chol2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_chol2<false, true>), dim3(S), dim3(CH2_THREADS), lds, st, N, dA, dpf, dact, (double*)nullptr, dFinv, dpfd, drin, dzout);
  }
  return hipGetLastError();
}
