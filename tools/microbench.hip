// Peak-rate microbenchmarks for the roofline denominators used by bench.py / DESIGN.md:
//   fp64 FMA (VALU), the hot-loop exp (vb_exp), ocml exp, v_mfma_f64_16x16x4_f64, and a mixed
//   MFMA+VALU kernel (do the two pipes overlap for fp64?).  Prints one JSON object.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../vbmc_amd/csrc/device_math.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_fma(double* out, int iters, double a, double b) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP, int MODE>
__global__ void k_exp(double* out, int iters, double a) {
  __shared__ double tab[VB_EXP_TAB_N];
  for (int t = threadIdx.x; t < VB_EXP_TAB_N; t += blockDim.x) tab[t] = c_exp2_tab[t];
  __syncthreads();
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = -(threadIdx.x * 1e-2 + i);
  double s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      double e = MODE == 0 ? vb_exp(x[i]) : MODE == 1 ? exp(x[i]) : MODE == 2 ? vb_exp_tab<0>(x[i], tab) : vb_exp_tab<1>(x[i], tab);
      s += e;
      x[i] -= a;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU>
__global__ void k_mfma(double* out, int iters, double a, double b) {
  d4 acc[NACC > 0 ? NACC : 1];
#pragma unroll
  for (int i = 0; i < (NACC > 0 ? NACC : 1); ++i) acc[i] = (d4){0, 0, 0, 0};
  double va = threadIdx.x * 1e-3, vb = 1.0 + threadIdx.x * 1e-4;
  double x[NVALU > 0 ? NVALU : 1];
#pragma unroll
  for (int i = 0; i < (NVALU > 0 ? NVALU : 1); ++i) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NVALU; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < (NVALU > 0 ? NVALU : 1); ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8, threads = 256;
  double* out;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
  const int iters = 4096;
  const double lanes = (double)blocks * threads;
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.name, prop.gcnArchName, cus, prop.clockRate / 1000);
  {
    double ms = time_ms([&] { hipLaunchKernelGGL((k_fma<8>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 5);
    printf(", \"fma_f64_tflops\": %.2f", 2.0 * lanes * iters * 8 / (ms * 1e-3) / 1e12);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL((k_exp<4, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters / 4, 1e-3); }, 5);
    printf(", \"vb_exp_gexp_s\": %.1f", lanes * (iters / 4) * 4 / (ms * 1e-3) / 1e9);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL((k_exp<4, 1>), dim3(blocks), dim3(threads), 0, 0, out, iters / 4, 1e-3); }, 5);
    printf(", \"ocml_exp_gexp_s\": %.1f", lanes * (iters / 4) * 4 / (ms * 1e-3) / 1e9);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL((k_exp<4, 2>), dim3(blocks), dim3(threads), 0, 0, out, iters / 4, 1e-3); }, 5);
    printf(", \"vb_exp_tab0_gexp_s\": %.1f", lanes * (iters / 4) * 4 / (ms * 1e-3) / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL((k_exp<4, 3>), dim3(blocks), dim3(threads), 0, 0, out, iters / 4, 1e-3); }, 5);
    printf(", \"vb_exp_tab1_gexp_s\": %.1f", lanes * (iters / 4) * 4 / (ms * 1e-3) / 1e9);
  }
  {
    double waves = lanes / 64;
    double ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<4, 0>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 5);
    printf(", \"mfma_f64_16x16x4_tflops\": %.2f", 2.0 * 16 * 16 * 4 * waves * iters * 4 / (ms * 1e-3) / 1e12);
    double ms2 = time_ms([&] { hipLaunchKernelGGL((k_mfma<4, 16>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 5);
    double ms3 = time_ms([&] { hipLaunchKernelGGL((k_mfma<0, 16>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999, 1e-3); }, 5);
    printf(", \"mix_ms_mfma4\": %.4f, \"mix_ms_mfma4_valu16\": %.4f, \"mix_ms_valu16\": %.4f", ms, ms2, ms3);
  }
  printf("}\n");
  hipFree(out);
  return 0;
}
