// Throughput and accuracy of candidate formulations of the entropy kernel's table exp (device_math.h: vb_exp_tab):
//   A  current: clamp, magic-number rounding of x*256/ln2, 256-entry table, degree-4 expm1            (13 VALU + 1 LDS)
//   B  argument pre-scaled (y = x*N/ln2 comes out of the S-step MFMA), v_rndne + saturating v_cvt (no clamp), 256 / deg 4
//   C  as B with a 1024-entry table and a degree-3 polynomial
//   D  pre-scaled, magic-number rounding, exponent taken from mantissa bits 10..41 (v_alignbit), 1024 / deg 3, clamp
// Prints Gexp/s and the largest relative error against the host's exp over the sampled arguments.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double exp_A(double x, const double* tab) {
  const double INV = 369.3299304675746322841407, C_1 = 0.693147180559945309417232 / 256, MAGIC = 6755399441055744.0;
  asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "v"(-1.0e6));
  double t = fma(x, INV, MAGIC);
  int ni = __double2loint(t);
  double nf = t - MAGIC;
  double r = fma(nf, -C_1, x);
  double T = tab[ni & 255];
  double v = r * r;
  double u = fma(r, 1.6666666666666665741e-01, 0.5);
  u = fma(v, 4.1666666666666664354e-02, u);
  double p = fma(v, u, r);
  return ldexp(fma(T, p, T), ni >> 8);
}

// y = x * 256/ln2; r' = y - n in [-1/2, 1/2], exp = 2^(n>>8) T[n&255] (1 + expm1(c r')), c = ln2/256
__device__ __forceinline__ double exp_B(double y, const double* tab) {
  const double c = 0.693147180559945309417232 / 256;
  double nr = __builtin_rint(y);
  int ni = __double2int_rz(nr);          // saturates
  double r = y - nr;
  double T = tab[ni & 255];
  double v = r * r;
  double u = fma(r, c * c * c / 6, c * c / 2);
  u = fma(v, c * c * c * c / 24, u);
  double p = fma(v, u, c * r);
  return ldexp(fma(T, p, T), ni >> 8);
}

__device__ __forceinline__ double exp_C(double y, const double* tab) {   // y = x * 1024/ln2
  const double c = 0.693147180559945309417232 / 1024;
  double nr = __builtin_rint(y);
  int ni = __double2int_rz(nr);
  double r = y - nr;
  double T = tab[ni & 1023];
  double u = fma(r, c * c * c / 6, c * c / 2);
  double w = r * c;
  double p = fma(r * r, u, w);
  return ldexp(fma(T, p, T), ni >> 10);
}

// C with one multiply less: p = r (c + r (c^2/2 + r c^3/6)) in Horner form (3 dependent FMAs, no r*r)
__device__ __forceinline__ double exp_C2(double y, const double* tab) {
  const double c = 0.693147180559945309417232 / 1024;
  double nr = __builtin_rint(y);
  int ni = __double2int_rz(nr);
  double r = y - nr;
  double T = tab[ni & 1023];
  double u = fma(r, c * c * c / 6, c * c / 2);
  u = fma(r, u, c);
  double Tr = T * r;
  return ldexp(fma(Tr, u, T), ni >> 10);
}

__device__ __forceinline__ double exp_D(double y, const double* tab) {   // y = x * 1024/ln2
  const double c = 0.693147180559945309417232 / 1024, MAGIC = 6755399441055744.0;
  asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(y), "v"(-2.0e9));
  double t = y + MAGIC;
  int ni = __double2loint(t);
  double r = y - (t - MAGIC);
  double T = tab[ni & 1023];
  double u = fma(r, c * c * c / 6, c * c / 2);
  u = fma(r, u, c);
  double Tr = T * r;
  return ldexp(fma(Tr, u, T), ni >> 10);
}

template <int ILP, int MODE>
__global__ void k_exp(double* out, double* err, int iters, double a, const double* gtab, int ntab) {
  extern __shared__ double tab[];
  for (int t = threadIdx.x; t < ntab; t += blockDim.x) tab[t] = gtab[t];
  __syncthreads();
  const double SC = MODE == 0 ? 1.0 : (MODE == 1 ? 369.3299304675746322841407 : 1477.3197218702985291365628);
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = -(threadIdx.x * 1e-2 + i + blockIdx.x * 0.37) * SC;
  double s = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      double e = MODE == 0 ? exp_A(x[i], tab) : MODE == 1 ? exp_B(x[i], tab) : MODE == 2 ? exp_C(x[i], tab) : MODE == 3 ? exp_C2(x[i], tab) : exp_D(x[i], tab);
      s += e;
      x[i] -= a * SC;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (err) {   // accuracy probe: one value per thread against the caller's reference
    double xx = -(threadIdx.x * 0.173 + blockIdx.x * 0.0137);
    err[blockIdx.x * blockDim.x + threadIdx.x] =
        MODE == 0 ? exp_A(xx, tab) : MODE == 1 ? exp_B(xx * SC, tab) : MODE == 2 ? exp_C(xx * SC, tab) : MODE == 3 ? exp_C2(xx * SC, tab) : exp_D(xx * SC, tab);
  }
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
  return ms / reps;
}

template <int MODE>
static int run(const char* name, double* out, double* err, const double* gtab, int ntab, int blocks, int threads) {
  const int iters = 1024;
  const double lanes = (double)blocks * threads;
  double ms = time_ms([&] { hipLaunchKernelGGL((k_exp<4, MODE>), dim3(blocks), dim3(threads), ntab * 8, 0, out, err, iters, 1e-3, gtab, ntab); }, 5);
  std::vector<double> h((size_t)blocks * threads);
  CHECK(hipMemcpy(h.data(), err, h.size() * 8, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int b = 0; b < blocks; ++b)
    for (int t = 0; t < threads; ++t) {
      double xx = -(t * 0.173 + b * 0.0137);
      long double ref = expl((long double)xx);
      double e = fabs((double)((h[(size_t)b * threads + t] - ref) / ref));
      if (e > worst) worst = e;
    }
  printf("%-3s %8.1f Gexp/s   max rel err %.2e\n", name, lanes * iters * 4 / (ms * 1e-3) / 1e9, worst);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 8, threads = 256;
  double *out, *err, *t256, *t1024;
  CHECK(hipMalloc(&out, 8 * blocks * threads));
  CHECK(hipMalloc(&err, 8 * blocks * threads));
  CHECK(hipMalloc(&t256, 8 * 256));
  CHECK(hipMalloc(&t1024, 8 * 1024));
  std::vector<double> h(1024);
  for (int j = 0; j < 256; ++j) h[j] = (double)exp2l((long double)j / 256);
  CHECK(hipMemcpy(t256, h.data(), 8 * 256, hipMemcpyHostToDevice));
  for (int j = 0; j < 1024; ++j) h[j] = (double)exp2l((long double)j / 1024);
  CHECK(hipMemcpy(t1024, h.data(), 8 * 1024, hipMemcpyHostToDevice));
  run<0>("A", out, err, t256, 256, blocks, threads);
  run<1>("B", out, err, t256, 256, blocks, threads);
  run<2>("C", out, err, t1024, 1024, blocks, threads);
  run<3>("C2", out, err, t1024, 1024, blocks, threads);
  run<4>("D", out, err, t1024, 1024, blocks, threads);
  return 0;
}
