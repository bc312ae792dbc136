// One translation unit per QS (compiled with -DQS_VALUE=n, in parallel, see vbmc_amd/build.py):
// instantiates k_entropy_mfma<QS, KT, grad, sparse, HV, TL> for KT = 1..4 (HV = 1; KT = 1..3 also with a component tail), KT = 3, 4 with the components split over
// two waves (HV = 2) and KT = 2..4 over four waves (HV = 4), and exports a launcher.
#include "entropy_mfma.h"

#ifndef QS_VALUE
#error "compile with -DQS_VALUE=<1..9>"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

// mode 0: launch.  mode 1: no launch -- returns the number of workgroups of this instantiation one compute unit holds at the launch's
// dynamic LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor: registers AND LDS), for the chunk model of elbo_plan.
template <int KT, int HV, int TL = 0>
static int launch_kt(int mode, int grad, dim3 grid, hipStream_t st, const EntArgs& ea) {
  // dynamic LDS: the parameter block (<= 74 KB at K = 256, D = 32), reused by the exp table (8 KB) and the PV exchange of
  // multi-wave workgroups (2 signs x HV waves x NPV x 4 x 64 doubles)
  constexpr int NPV_ = (4 * QS_VALUE + 15) / 16;
  size_t lds = (size_t)ea.K * (ea.D + ENTP_EXTRA) * sizeof(double);
  size_t after = (size_t)VB_EXP_TAB1K_N * sizeof(double);          // the exp table takes the block's place once the operands are built
  if (HV > 1) after += (size_t)(VBMC_ENT_EO(HV) ? 1 : 2) * HV * NPV_ * 4 * WAVE * sizeof(double);   // PV exchange: per sign, or one for both (entropy_mfma.h: YXSB)
  if (after > lds) lds = after;
  const void* fn = nullptr;
  if constexpr (HV == 1 && QS_VALUE <= 8) {
    // the launch carries the log-joint role (gradient kernels, dense): the caller checked the shape
    if (ea.lj.rows > 0) fn = (const void*)k_entropy_mfma<QS_VALUE, KT, true, false, 1, TL, true>;
  }
  // the walking launch (entropy_mfma.h: WALK; elbo_plan decides): the device-RNG gradient kernels of single-wave workgroups, without the role
  const bool walk = ea.walk_tpw > 0;
  if (walk && !(HV == 1 && QS_VALUE <= 4 && KT <= 3 && grad && !fn && !ea.eps && !(ea.cutoff > 0.0))) return mode != 0 ? -1 : 1;
  // the device-RNG launch of a kernel that otherwise spends registers on the parity mode's prefetch (entropy_mfma.h: EM, EPF)
  if constexpr (HV == 1 && QS_VALUE <= 4 && KT <= 3) {
    if (!fn && grad && !ea.eps && !(ea.cutoff > 0.0))
      fn = walk ? (const void*)k_entropy_mfma<QS_VALUE, KT, true, false, 1, TL, false, false, true> : (const void*)k_entropy_mfma<QS_VALUE, KT, true, false, 1, TL, false, false>;
  }
  if (!fn) {
    if (ea.cutoff > 0.0 && HV == 1 && !TL) {  // opt-in block-sparse variant (single-wave kernels only, no component tail)
      if constexpr (HV == 1 && TL == 0)
        fn = grad ? (const void*)k_entropy_mfma<QS_VALUE, KT, true, true, 1> : (const void*)k_entropy_mfma<QS_VALUE, KT, false, true, 1>;
    } else {
      fn = grad ? (const void*)k_entropy_mfma<QS_VALUE, KT, true, false, HV, TL> : (const void*)k_entropy_mfma<QS_VALUE, KT, false, false, HV, TL>;
    }
  }
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int threads = (ea.cutoff > 0.0 && HV == 1 && !TL) ? WAVE : WAVE * HV;
  if (mode == 1) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb;
  }
  EntArgs arg = ea;
  void* args[] = {(void*)&arg};
  (void)hipLaunchKernel(fn, grid, dim3(threads), args, lds, st);
  return 0;
}

// kt = k-tiles per wave (1..4), hv = waves per workgroup the components are split over (1: K <= 64, 2: K <= 128)
// mode 0: launch, returns 0 (1: no such instantiation).  mode 1: returns the workgroups per compute unit of the instantiation (-1: none).
static int dispatch(int mode, int kt, int grad, int hv, dim3 grid, hipStream_t st, const EntArgs* ea) {
  switch (kt + 16 * hv) {
    case 16 + 1: return launch_kt<1, 1>(mode, grad, grid, st, *ea);
    case 16 + 2: return launch_kt<2, 1>(mode, grad, grid, st, *ea);
    case 16 + 3: return launch_kt<3, 1>(mode, grad, grid, st, *ea);
    case 16 + 4: return launch_kt<4, 1>(mode, grad, grid, st, *ea);
    // hv + 16 TL: kt full k-tiles per wave + a tail of (components per wave) mod 16 <= 4 TL components, TL values per lane
    case 272 + 1: return launch_kt<1, 1, 1>(mode, grad, grid, st, *ea);   // one wave: K = 17..20
    case 272 + 2: return launch_kt<2, 1, 1>(mode, grad, grid, st, *ea);   //           33..36
    case 272 + 3: return launch_kt<3, 1, 1>(mode, grad, grid, st, *ea);   //           49..52
    case 288 + 2: return launch_kt<2, 2, 1>(mode, grad, grid, st, *ea);   // two waves: K = 66..72
    case 288 + 3: return launch_kt<3, 2, 1>(mode, grad, grid, st, *ea);   //            98..104
    case 320 + 2: return launch_kt<2, 4, 1>(mode, grad, grid, st, *ea);   // four waves: K = 130..144
    case 320 + 3: return launch_kt<3, 4, 1>(mode, grad, grid, st, *ea);   //             194..208
    case 528 + 1: return launch_kt<1, 1, 2>(mode, grad, grid, st, *ea);   // two values per lane: K = 21..24
    case 528 + 2: return launch_kt<2, 1, 2>(mode, grad, grid, st, *ea);   //                      37..40
    case 528 + 3: return launch_kt<3, 1, 2>(mode, grad, grid, st, *ea);   //                      53..56
    case 544 + 2: return launch_kt<2, 2, 2>(mode, grad, grid, st, *ea);   // two waves: K = 74..80
    case 544 + 3: return launch_kt<3, 2, 2>(mode, grad, grid, st, *ea);   //            106..112
    case 576 + 2: return launch_kt<2, 4, 2>(mode, grad, grid, st, *ea);   // four waves: K = 146..160
    case 576 + 3: return launch_kt<3, 4, 2>(mode, grad, grid, st, *ea);   //             210..224
    case 32 + 2: return launch_kt<2, 2>(mode, grad, grid, st, *ea);   // 32 < K <= 64 at D >= 17, two waves (round 3: the one-wave kernels spill there)
    case 32 + 3: return launch_kt<3, 2>(mode, grad, grid, st, *ea);   // 64 < K <= 96, two waves
    case 32 + 4: return launch_kt<4, 2>(mode, grad, grid, st, *ea);   // 96 < K <= 128
    case 64 + 2: return launch_kt<2, 4>(mode, grad, grid, st, *ea);   // 64 < K <= 128, four waves
    case 64 + 3: return launch_kt<3, 4>(mode, grad, grid, st, *ea);   // 128 < K <= 192
    case 64 + 4: return launch_kt<4, 4>(mode, grad, grid, st, *ea);   // 192 < K <= 256
    case 128 + 3: return launch_kt<3, 8>(mode, grad, grid, st, *ea);  // round 5: eight waves, 256 < K <= 384
    case 128 + 4: return launch_kt<4, 8>(mode, grad, grid, st, *ea);  //                       384 < K <= 512
    default: return mode != 0 ? -1 : 1;
  }
}

extern "C" int CAT(vbmc_launch_ent_mfma_qs, QS_VALUE)(int kt, int grad, int hv, unsigned gx, unsigned gy, unsigned gz, void* stream,
                                                      const EntArgs* ea) {
  return dispatch(0, kt, grad, hv, dim3(gx, gy, gz), (hipStream_t)stream, ea);
}

// workgroups of the instantiation (kt, grad, hv [+ 16 TL]; ea: D, K, cutoff, lj.rows) that one compute unit holds; -1: no such kernel
extern "C" int CAT(vbmc_occupancy_ent_mfma_qs, QS_VALUE)(int kt, int grad, int hv, const EntArgs* ea) {
  return dispatch(1, kt, grad, hv, dim3(1, 1, 1), nullptr, ea);
}

#ifdef VBMC_INSTRUMENT
extern "C" int CAT(vbmc_dbg_ent_read_qs, QS_VALUE)(unsigned long long* out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ent_dbg), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
