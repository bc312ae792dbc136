// One translation unit per QS (compiled with -DQS_VALUE=n, in parallel, see vbmc_amd/build.py):
// instantiates k_entropy_mfma<QS, KT, grad> for KT = 1..8 and exports a launcher.
#include "entropy_mfma.h"

#ifndef QS_VALUE
#error "compile with -DQS_VALUE=<1..9>"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

template <int KT>
static void launch_kt(int grad, dim3 grid, hipStream_t st, const EntArgs& ea) {
  const size_t lds = (size_t)ea.K * (ea.D + ENTP_EXTRA) * sizeof(double);  // parameter block (<= 39 KB)
  if (ea.cutoff > 0.0) {  // opt-in block-sparse variant
    if (grad) hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, true>), grid, dim3(WAVE), lds, st, ea);
    else hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, false, true>), grid, dim3(WAVE), lds, st, ea);
  } else {
    if (grad) hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, false>), grid, dim3(WAVE), lds, st, ea);
    else hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, false, false>), grid, dim3(WAVE), lds, st, ea);
  }
}

extern "C" int CAT(vbmc_launch_ent_mfma_qs, QS_VALUE)(int kt, int grad, unsigned gx, unsigned gy, unsigned gz, void* stream,
                                                      const EntArgs* ea) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gx, gy, gz);
  switch (kt) {
    case 1: launch_kt<1>(grad, grid, st, *ea); return 0;
    case 2: launch_kt<2>(grad, grid, st, *ea); return 0;
    case 3: launch_kt<3>(grad, grid, st, *ea); return 0;
    case 4: launch_kt<4>(grad, grid, st, *ea); return 0;
#if QS_VALUE <= 6
    case 5: launch_kt<5>(grad, grid, st, *ea); return 0;
    case 6: launch_kt<6>(grad, grid, st, *ea); return 0;
    case 7: launch_kt<7>(grad, grid, st, *ea); return 0;
    case 8: launch_kt<8>(grad, grid, st, *ea); return 0;
#endif
    default: return 1;
  }
}
