// One translation unit per QS (compiled with -DQS_VALUE=n, in parallel, see vbmc_amd/build.py):
// instantiates k_entropy_mfma<QS, KT, grad, sparse, HV, TL> for KT = 1..4 (HV = 1; KT = 1..3 also with a component tail), KT = 3, 4 with the components split over
// two waves (HV = 2) and KT = 2..4 over four waves (HV = 4), and exports a launcher.
#include "entropy_mfma.h"

#ifndef QS_VALUE
#error "compile with -DQS_VALUE=<1..9>"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

template <int KT, int HV, int TL = 0>
static void launch_kt(int grad, dim3 grid, hipStream_t st, const EntArgs& ea) {
  // dynamic LDS: the parameter block (<= 74 KB at K = 256, D = 32), reused by the exp table (8 KB) and the PV exchange of
  // multi-wave workgroups (2 signs x HV waves x NPV x 4 x 64 doubles)
  constexpr int NPV_ = (4 * QS_VALUE + 15) / 16;
  size_t lds = (size_t)ea.K * (ea.D + ENTP_EXTRA) * sizeof(double);
  size_t after = (size_t)VB_EXP_TAB1K_N * sizeof(double);          // the exp table takes the block's place once the operands are built
  if (HV > 1) after += (size_t)2 * HV * NPV_ * 4 * WAVE * sizeof(double);
  if (after > lds) lds = after;
  if (lds > 64 * 1024) {
    if (grad) (void)hipFuncSetAttribute((const void*)k_entropy_mfma<QS_VALUE, KT, true, false, HV, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)k_entropy_mfma<QS_VALUE, KT, false, false, HV, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if constexpr (HV == 1 && QS_VALUE <= 4) {
    if (ea.lj.rows > 0) {   // the launch carries the log-joint role (gradient kernels, dense): the caller checked vbmc_ent_mfma_has_co
      hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, false, 1, TL, true>), grid, dim3(WAVE), lds, st, ea);
      return;
    }
  }
  if (ea.cutoff > 0.0 && HV == 1 && !TL) {  // opt-in block-sparse variant (single-wave kernels only, no component tail)
    if (grad) hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, true, 1>), grid, dim3(WAVE), lds, st, ea);
    else hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, false, true, 1>), grid, dim3(WAVE), lds, st, ea);
  } else {
    if (grad) hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, false, HV, TL>), grid, dim3(WAVE * HV), lds, st, ea);
    else hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, false, false, HV, TL>), grid, dim3(WAVE * HV), lds, st, ea);
  }
}

// kt = k-tiles per wave (1..4), hv = waves per workgroup the components are split over (1: K <= 64, 2: K <= 128)
extern "C" int CAT(vbmc_launch_ent_mfma_qs, QS_VALUE)(int kt, int grad, int hv, unsigned gx, unsigned gy, unsigned gz, void* stream,
                                                      const EntArgs* ea) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gx, gy, gz);
  switch (kt + 16 * hv) {
    case 16 + 1: launch_kt<1, 1>(grad, grid, st, *ea); return 0;
    case 16 + 2: launch_kt<2, 1>(grad, grid, st, *ea); return 0;
    case 16 + 3: launch_kt<3, 1>(grad, grid, st, *ea); return 0;
    case 16 + 4: launch_kt<4, 1>(grad, grid, st, *ea); return 0;
    // hv + 16 TL: kt full k-tiles per wave + a tail of (components per wave) mod 16 <= 4 TL components, TL values per lane
    case 272 + 1: launch_kt<1, 1, 1>(grad, grid, st, *ea); return 0;   // one wave: K = 17..20
    case 272 + 2: launch_kt<2, 1, 1>(grad, grid, st, *ea); return 0;   //           33..36
    case 272 + 3: launch_kt<3, 1, 1>(grad, grid, st, *ea); return 0;   //           49..52
    case 288 + 2: launch_kt<2, 2, 1>(grad, grid, st, *ea); return 0;   // two waves: K = 66..72
    case 288 + 3: launch_kt<3, 2, 1>(grad, grid, st, *ea); return 0;   //            98..104
    case 320 + 2: launch_kt<2, 4, 1>(grad, grid, st, *ea); return 0;   // four waves: K = 130..144
    case 320 + 3: launch_kt<3, 4, 1>(grad, grid, st, *ea); return 0;   //             194..208
    case 528 + 1: launch_kt<1, 1, 2>(grad, grid, st, *ea); return 0;   // two values per lane: K = 21..24
    case 528 + 2: launch_kt<2, 1, 2>(grad, grid, st, *ea); return 0;   //                      37..40
    case 528 + 3: launch_kt<3, 1, 2>(grad, grid, st, *ea); return 0;   //                      53..56
    case 544 + 2: launch_kt<2, 2, 2>(grad, grid, st, *ea); return 0;   // two waves: K = 74..80
    case 544 + 3: launch_kt<3, 2, 2>(grad, grid, st, *ea); return 0;   //            106..112
    case 576 + 2: launch_kt<2, 4, 2>(grad, grid, st, *ea); return 0;   // four waves: K = 146..160
    case 576 + 3: launch_kt<3, 4, 2>(grad, grid, st, *ea); return 0;   //             210..224
    case 32 + 2: launch_kt<2, 2>(grad, grid, st, *ea); return 0;   // 32 < K <= 64 at D >= 17, two waves (round 3: the one-wave kernels spill there)
    case 32 + 3: launch_kt<3, 2>(grad, grid, st, *ea); return 0;   // 64 < K <= 96, two waves
    case 32 + 4: launch_kt<4, 2>(grad, grid, st, *ea); return 0;   // 96 < K <= 128
    case 64 + 2: launch_kt<2, 4>(grad, grid, st, *ea); return 0;   // 64 < K <= 128, four waves
    case 64 + 3: launch_kt<3, 4>(grad, grid, st, *ea); return 0;   // 128 < K <= 192
    case 64 + 4: launch_kt<4, 4>(grad, grid, st, *ea); return 0;   // 192 < K <= 256
    default: return 1;
  }
}
