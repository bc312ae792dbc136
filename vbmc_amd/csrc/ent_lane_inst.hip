// One translation unit per padded dimension DT (compiled with -DDT_VALUE=n, in parallel, see vbmc_amd/build.py): instantiates
// k_entropy_lane<DT, KP, grad> for KP = 2, 4, .., 16 and exports a launcher.
#include "entropy_lane.h"

#ifndef DT_VALUE
#error "compile with -DDT_VALUE=<2,4,..,12>"
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

// mode 0: launch.  mode 1: workgroups of the instantiation one compute unit holds (registers and LDS)
template <int KP>
static int launch_kp(int mode, int grad, dim3 grid, hipStream_t st, const EntArgs& ea) {
  const void* fn = grad ? (const void*)k_entropy_lane<DT_VALUE, KP, true> : (const void*)k_entropy_lane<DT_VALUE, KP, false>;
  if (mode == 1) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, WAVE * ENT_LANE_WAVES, ent_lane_role_lds(ea)) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nb;
  }
  EntArgs arg = ea;
  void* args[] = {(void*)&arg};
  (void)hipLaunchKernel(fn, grid, dim3(WAVE * ENT_LANE_WAVES), args, ent_lane_role_lds(ea), st);
  return 0;
}

// (DT = 12 with KP >= 10 and DT = 10 with KP >= 12 are outside the class: abi_elbo.hip, lane_entropy_fits)
static int dispatch(int mode, int kp, int grad, dim3 grid, hipStream_t st, const EntArgs* ea) {
  switch (kp) {
    case 2: return launch_kp<2>(mode, grad, grid, st, *ea);
    case 4: return launch_kp<4>(mode, grad, grid, st, *ea);
    case 6: return launch_kp<6>(mode, grad, grid, st, *ea);
    case 8: return launch_kp<8>(mode, grad, grid, st, *ea);
#if DT_VALUE <= 10
    case 10: return launch_kp<10>(mode, grad, grid, st, *ea);
#endif
#if DT_VALUE <= 8
    case 12: return launch_kp<12>(mode, grad, grid, st, *ea);
    case 14: return launch_kp<14>(mode, grad, grid, st, *ea);
    case 16: return launch_kp<16>(mode, grad, grid, st, *ea);
#endif
    default: return mode != 0 ? -1 : 1;
  }
}

extern "C" int CAT(vbmc_launch_ent_lane_dt, DT_VALUE)(int kp, int grad, unsigned gx, unsigned gy, unsigned gz, void* stream, const EntArgs* ea) {
  return dispatch(0, kp, grad, dim3(gx, gy, gz), (hipStream_t)stream, ea);
}
extern "C" int CAT(vbmc_occupancy_ent_lane_dt, DT_VALUE)(int kp, int grad, const EntArgs* ea) {
  return dispatch(1, kp, grad, dim3(1, 1, 1), nullptr, ea);
}

#ifdef VBMC_INSTRUMENT
extern "C" int CAT(vbmc_dbg_lane_read_dt, DT_VALUE)(unsigned long long* out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_dbg), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif
