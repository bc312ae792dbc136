// Cycles per instruction of the fp64 / integer VALU instructions the entropy kernel is made of, and of v_mfma_f64_16x16x4_f64,
// measured INSIDE the kernel with s_memtime (shader clock) -- the cost model behind DESIGN.md section 4 "Where the time goes".
// The roofline arithmetic of round 2 priced every VALU instruction at 4 cycles per wave (16 lanes per clock); this measures it:
// per instruction type, NCH independent register chains per wave, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o vbmc_amd/lib/valu_rate && vbmc_amd/lib/valu_rate
// Output: one line per (instruction, waves per SIMD): cycles per instruction per wave-slot, i.e. SIMD cycles / instructions issued
// on that SIMD.  4.0 = full rate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

enum Op { FMA3, FMA_S, FMAC, ADD, MUL, LDEXP, RNDNE, CVT_I32, CVT_F64, AND32, LSHLADD, ASHR, RCP, FREXPM, CNDMASK, MOV64, MFMA, MFMA_FMA4, MFMA_FMA16, DSREAD, FMA3_SAMEBANK };

template <int OP>
__global__ void __launch_bounds__(1024) k_rate(unsigned long long* cyc, double* sink, int iters, double a, double b, int sh) {
  __shared__ double lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  constexpr int NCH = 8;
  double x[NCH], y[NCH], z[NCH];
  int n[NCH];
  d4 acc[4];
#pragma unroll
  for (int i = 0; i < NCH; ++i) { x[i] = 1.0 + threadIdx.x * 1e-3 + i; y[i] = 0.999 + i * 1e-5; z[i] = 1e-3 * (i + 1); n[i] = threadIdx.x + i; }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (OP == FMA3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
        if (OP == FMA3_SAMEBANK) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y[i]));
        if (OP == FMA_S) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(z[i]));
        if (OP == FMAC) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(z[i]));
        if (OP == ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (OP == MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (OP == LDEXP) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(n[i] & 1));
        if (OP == RNDNE) asm volatile("v_rndne_f64_e32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
        if (OP == CVT_I32) asm volatile("v_cvt_i32_f64_e32 %0, %1" : "=v"(n[i]) : "v"(x[i]));
        if (OP == CVT_F64) asm volatile("v_cvt_f64_i32_e32 %0, %1" : "=v"(y[i]) : "v"(n[i]));
        if (OP == AND32) asm volatile("v_and_b32_e32 %0, 0x3ff, %0" : "+v"(n[i]));
        if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[i]) : "v"(sh));
        if (OP == ASHR) asm volatile("v_ashrrev_i32_e32 %0, 10, %0" : "+v"(n[i]));
        if (OP == RCP) asm volatile("v_rcp_f64_e32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
        if (OP == FREXPM) asm volatile("v_frexp_mant_f64_e32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(sh));
        if (OP == MOV64) asm volatile("v_mov_b64_e32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
        if (OP == DSREAD) asm volatile("ds_read_b64 %0, %1" : "=v"(y[i]) : "v"((n[i] & 127) * 8));
      }
      if (OP == MFMA || OP == MFMA_FMA4 || OP == MFMA_FMA16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], y[i], acc[i], 0, 0, 0);
      }
      if (OP == MFMA_FMA4 || OP == MFMA_FMA16) {
#pragma unroll
        for (int rep = 0; rep < (OP == MFMA_FMA16 ? 8 : 2); ++rep)
#pragma unroll
          for (int i = 4; i < NCH; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(z[i]) : "v"(y[i]), "v"(x[i]));
      }
    }
    if (OP == DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += x[i] + y[i] + z[i] + n[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
static void run(const char* name, int cus, double insts_per_iter, double mfma_per_iter) {
  const int iters = 20000;
  for (int W : {1, 2, 4}) {
    // ONE workgroup of 4 W waves per compute unit: the waves of a workgroup are dealt round-robin to the four SIMDs, so every SIMD
    // holds exactly W of them
    const int blocks = cus, nw = 4 * W;
    unsigned long long* cyc;
    double* sink;
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * nw));
    CHECK(hipMalloc(&sink, sizeof(double) * blocks * nw * 64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(64 * nw), 0, 0, cyc, sink, iters, 0.999, 1e-3, 3);
    CHECK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(64 * nw), 0, 0, cyc, sink, iters, 0.999, 1e-3, 3);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * nw);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * nw, hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks * nw;
    // every SIMD holds W waves (if the dispatcher spreads them evenly); SIMD cycles per instruction issued on it:
    const double per_wave = mean / (iters * (insts_per_iter + mfma_per_iter));
    printf("%-14s W=%d  wave-cycles/inst %.2f  => SIMD cycles/inst %.2f   (kernel %.3f ms, s_memtime ticks/us %.1f)\n", name, W, per_wave, per_wave / W, ms,
           mean / (ms * 1e3));
    hipFree(cyc); hipFree(sink);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %d MHz; 8 independent chains per wave, 32 instructions per loop trip\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  run<FMA3>("v_fma_f64 vvv", cus, 32, 0);
  run<FMA3_SAMEBANK>("v_fma_f64 vv=", cus, 32, 0);
  run<FMA_S>("v_fma_f64 vss", cus, 32, 0);
  run<FMAC>("v_fmac_f64", cus, 32, 0);
  run<ADD>("v_add_f64", cus, 32, 0);
  run<MUL>("v_mul_f64", cus, 32, 0);
  run<LDEXP>("v_ldexp_f64", cus, 32, 0);
  run<RNDNE>("v_rndne_f64", cus, 32, 0);
  run<CVT_I32>("v_cvt_i32_f64", cus, 32, 0);
  run<CVT_F64>("v_cvt_f64_i32", cus, 32, 0);
  run<AND32>("v_and_b32", cus, 32, 0);
  run<LSHLADD>("v_lshl_add_u32", cus, 32, 0);
  run<ASHR>("v_ashrrev_i32", cus, 32, 0);
  run<RCP>("v_rcp_f64", cus, 32, 0);
  run<FREXPM>("v_frexp_mant", cus, 32, 0);
  run<CNDMASK>("v_cndmask_b32", cus, 32, 0);
  run<MOV64>("v_mov_b64", cus, 32, 0);
  run<DSREAD>("ds_read_b64", cus, 32, 0);
  run<MFMA>("mfma_f64", cus, 0, 16);
  run<MFMA_FMA4>("mfma+2fmac/mf", cus, 32, 16);
  run<MFMA_FMA16>("mfma+8fmac/mf", cus, 128, 16);
  return 0;
}
