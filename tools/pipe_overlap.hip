// Which VALU work overlaps with v_mfma_f64_16x16x4_f64 on gfx950?  One wave issues 4 independent MFMAs + NV VALU ops per iteration;
// the VALU ops are fp64 FMAs, fp32 FMAs, int32 mads or ds_reads.  time(mix) ~ time(mfma) + time(valu) => same pipe; ~ max => overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int NV, int KIND>
__global__ void k_mix(double* out, int iters, double a, double b) {
  __shared__ double sh[256];
  sh[threadIdx.x & 255] = threadIdx.x;
  __syncthreads();
  d4 acc[NACC > 0 ? NACC : 1];
  for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) acc[i] = (d4){0, 0, 0, 0};
  double va = threadIdx.x * 1e-3, vb = 1.0 + threadIdx.x * 1e-4;
  double x[NV > 0 ? NV : 1]; float xf[NV > 0 ? NV : 1]; unsigned xi[NV > 0 ? NV : 1];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) { x[i] = threadIdx.x * 1e-3 + i; xf[i] = x[i]; xi[i] = threadIdx.x + i; }
  float af = a, bf = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (KIND == 0) x[i] = fma(x[i], a, b);
      else if (KIND == 1) xf[i] = fmaf(xf[i], af, bf);
      else if (KIND == 2) xi[i] = xi[i] * 2654435761u + 12345u;
      else if (KIND == 3) xi[i] = (xi[i] & 0x3ff) + (xi[i] >> 3) + 77u;
      else if (KIND == 4) x[i] += sh[(xi[i] + it) & 255];
      else if (KIND == 5) x[i] = __builtin_rint(x[i] * a);
      else if (KIND == 6) xi[i] += (unsigned)__double2int_rz(x[i]);
      else if (KIND == 7) x[i] = ldexp(x[i], (int)(xi[i] & 1));
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i] + xf[i] + xi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
template <int KIND> static void run(const char* name, double* out, int blocks, int threads) {
  const int iters = 4096;
  double m = time_ms([&] { hipLaunchKernelGGL((k_mix<4, 0, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 3);
  double v = time_ms([&] { hipLaunchKernelGGL((k_mix<0, 16, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 3);
  double mv = time_ms([&] { hipLaunchKernelGGL((k_mix<4, 16, KIND>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 3);
  printf("%-28s mfma4 %.3f ms   valu16 %.3f ms   both %.3f ms   (sum %.3f, max %.3f)\n", name, m, v, mv, m + v, m > v ? m : v);
}
int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int blocks = prop.multiProcessorCount * 8, threads = 256;   // 8 waves per SIMD... 2 blocks x 4 waves per SIMD
  double* out; (void)hipMalloc(&out, sizeof(double) * blocks * threads);
  run<0>("fp64 fma", out, blocks, threads);
  run<1>("fp32 fma", out, blocks, threads);
  run<2>("int32 mul-add", out, blocks, threads);
  run<3>("int32 and/shift/add (3 ops)", out, blocks, threads);
  run<4>("ds_read_b64 + fp64 add", out, blocks, threads);
  run<5>("fp64 mul + rndne", out, blocks, threads);
  run<6>("cvt_i32_f64 + int add", out, blocks, threads);
  run<7>("ldexp f64 (+and)", out, blocks, threads);
  return 0;
}
