// One translation unit per QS (compiled with -DQS_VALUE=n, in parallel, see vbmc_amd/build.py):
// instantiates k_entropy_mfma<QS, KT, grad, sparse, HV> for KT = 1..4 (HV = 1) and KT = 3, 4 with the components split over
// two waves (HV = 2), and exports a launcher.
#include "entropy_mfma.h"

#ifndef QS_VALUE
#error "compile with -DQS_VALUE=<1..9>"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

template <int KT, int HV>
static void launch_kt(int grad, dim3 grid, hipStream_t st, const EntArgs& ea) {
  const size_t lds = (size_t)ea.K * (ea.D + ENTP_EXTRA) * sizeof(double);  // parameter block (<= 39 KB)
  if (ea.cutoff > 0.0 && HV == 1) {  // opt-in block-sparse variant (single-wave kernels only)
    if (grad) hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, true, 1>), grid, dim3(WAVE), lds, st, ea);
    else hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, false, true, 1>), grid, dim3(WAVE), lds, st, ea);
  } else {
    if (grad) hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, true, false, HV>), grid, dim3(WAVE * HV), lds, st, ea);
    else hipLaunchKernelGGL((k_entropy_mfma<QS_VALUE, KT, false, false, HV>), grid, dim3(WAVE * HV), lds, st, ea);
  }
}

// kt = k-tiles per wave (1..4), hv = waves per workgroup the components are split over (1: K <= 64, 2: K <= 128)
extern "C" int CAT(vbmc_launch_ent_mfma_qs, QS_VALUE)(int kt, int grad, int hv, unsigned gx, unsigned gy, unsigned gz, void* stream,
                                                      const EntArgs* ea) {
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(gx, gy, gz);
  switch (kt + 4 * (hv - 1)) {
    case 1: launch_kt<1, 1>(grad, grid, st, *ea); return 0;
    case 2: launch_kt<2, 1>(grad, grid, st, *ea); return 0;
    case 3: launch_kt<3, 1>(grad, grid, st, *ea); return 0;
    case 4: launch_kt<4, 1>(grad, grid, st, *ea); return 0;
    case 7: launch_kt<3, 2>(grad, grid, st, *ea); return 0;   // 64 < K <= 96
    case 8: launch_kt<4, 2>(grad, grid, st, *ea); return 0;   // 96 < K <= 128
    default: return 1;
  }
}
