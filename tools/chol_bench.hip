// Stand-alone check + timing of the blocked Cholesky kernel of vbmc_amd/csrc/chol_mfma.h (k_chol2): S copies of a
// random SPD matrix of order N, result against a host Cholesky, MATLAB's failure index on an indefinite matrix, kernel time by HIP
// events (median of reps) and -- built with -DCHOL_TS -- the per-step phase stamps of matrix 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCHOL_TS -Iinclude -o vbmc_amd/lib/chol_bench tools/chol_bench.hip
//   vbmc_amd/lib/chol_bench [variant=1] [N=400] [S=20] [reps=20] [stamps=0 | print every n-th step]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../vbmc_amd/csrc/gp_kernels.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int host_chol_upper(int N, std::vector<double>& A) {   // column-major, upper factor in place, strict lower zeroed; returns p
  for (int j = 0; j < N; ++j) {
    for (int i = 0; i <= j; ++i) {
      long double v = A[i + (size_t)N * j];
      for (int t = 0; t < i; ++t) v -= (long double)A[t + (size_t)N * i] * A[t + (size_t)N * j];
      if (i == j) {
        if (!(v > 0)) return j + 1;
        A[j + (size_t)N * j] = (double)sqrtl(v);
      } else {
        A[i + (size_t)N * j] = (double)(v / A[i + (size_t)N * i]);
      }
    }
    for (int i = j + 1; i < N; ++i) A[i + (size_t)N * j] = 0.0;
  }
  return 0;
}

// How fast can ONE compute unit stream 16 x 16 tiles of a column-major matrix through the L2?  (variant 9)
// mode 0: loads only; mode 1: load, +1, store back.  G tiles (4 loads of 512 B per wave each) in flight per wave.
template <int G>
__global__ void __launch_bounds__(512) k_tile_stream(int N, double* A, int mode, int passes, double* out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4, nt = N >> 4;
  unsigned lob[4];
  for (int r = 0; r < 4; ++r) lob[r] = (unsigned)(((lg + 4 * r) * N + li) * 8);
  char* Ab = reinterpret_cast<char*>(A);
  double sum = 0.0;
  for (int p = 0; p < passes; ++p)
    for (int t = wave * G; t + G <= nt * nt; t += 8 * G) {
      double c[G][4];
      unsigned ob[G][4];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int ti = (t + g) % nt, tj = (t + g) / nt;
        const unsigned tpb = (unsigned)(((tj << 4) * N + (ti << 4)) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ob[g][r] = tpb + lob[r]; c[g][r] = *reinterpret_cast<const double*>(Ab + ob[g][r]); }
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (mode) *reinterpret_cast<double*>(Ab + ob[g][r]) = c[g][r] + 1.0;
          else sum += c[g][r];
        }
    }
  if (sum == 1.2345e300) out[0] = sum;
}

// The same stream with 16-byte accesses: a lane takes rows (2 li, 2 li + 1) of a column of a 32 x 16 tile pair (variant 9, modes 2 / 3).
template <int G>
__global__ void __launch_bounds__(512) k_tile_stream_x4(int N, double* A, int mode, int passes, double* out) {
  typedef double d2a __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4, nti = N >> 5, ntj = N >> 4;        // 32-row pairs x 16-column tiles
  unsigned lob[4];
  for (int r = 0; r < 4; ++r) lob[r] = (unsigned)(((lg + 4 * r) * N + 2 * li) * 8);
  char* Ab = reinterpret_cast<char*>(A);
  double sum = 0.0;
  for (int p = 0; p < passes; ++p)
    for (int t = wave * G; t + G <= nti * ntj; t += 8 * G) {
      d2a c[G][4];
      unsigned ob[G][4];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int ti = (t + g) % nti, tj = (t + g) / nti;
        const unsigned tpb = (unsigned)(((tj << 4) * N + (ti << 5)) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) { ob[g][r] = tpb + lob[r]; c[g][r] = *reinterpret_cast<const d2a*>(Ab + ob[g][r]); }
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (mode) { d2a v = c[g][r]; v[0] += 1.0; v[1] += 1.0; *reinterpret_cast<d2a*>(Ab + ob[g][r]) = v; }
          else sum += c[g][r][0] + c[g][r][1];
        }
    }
  if (sum == 1.2345e300) out[0] = sum;
}

static double *g_dFinv = nullptr, *g_dpfd = nullptr, *g_drin = nullptr, *g_dzout = nullptr;   // variant 2: the by-products as well
static hipError_t launch(int variant, int N, int S, double* dA, int* dpf, unsigned char* dact, double* dPg, hipStream_t st) {
  switch (variant) {
    case 1: return chol2_launch(N, S, dA, dpf, dact, dPg, st);
    case 2: return chol2_launch(N, S, dA, dpf, dact, dPg, st, g_dFinv, g_dpfd, g_drin, g_dzout);
    default: return hipErrorInvalidValue;
  }
}

int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 1, N = argc > 2 ? atoi(argv[2]) : 400, S = argc > 3 ? atoi(argv[3]) : 20;
  const int reps = argc > 4 ? atoi(argv[4]) : 20, stamps = argc > 5 ? atoi(argv[5]) : 0;
  const size_t NN = (size_t)N * N;
  if (variant == 9) {
    double *dA, *dout;
    CHECK(hipMalloc(&dA, NN * 8)); CHECK(hipMalloc(&dout, 8));
    CHECK(hipMemset(dA, 0, NN * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int passes = 20;
    for (int G = 4; G <= 16; G *= 2)
      for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
          CHECK(hipEventRecord(e0, 0));
          if (G == 4) hipLaunchKernelGGL((k_tile_stream<4>), dim3(S), dim3(512), 0, 0, N, dA, mode, passes, dout);
          else if (G == 8) hipLaunchKernelGGL((k_tile_stream<8>), dim3(S), dim3(512), 0, 0, N, dA, mode, passes, dout);
          else hipLaunchKernelGGL((k_tile_stream<16>), dim3(S), dim3(512), 0, 0, N, dA, mode, passes, dout);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipDeviceSynchronize());
          float t;
          CHECK(hipEventElapsedTime(&t, e0, e1));
          best = std::min(best, t);
        }
        const double bytes = (double)passes * (N / 16) * (N / 16) * 2048.0 * (mode ? 2 : 1);
        printf("tile stream N=%d WGs=%d (same matrix) G=%d %s: %.3f ms  %.1f GB/s per WG\n", N, S, G, mode ? "load+store" : "load only", best,
               bytes / best * 1e-6);
      }
    for (int G = 2; G <= 8; G *= 2)
      for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
          CHECK(hipEventRecord(e0, 0));
          if (G == 2) hipLaunchKernelGGL((k_tile_stream_x4<2>), dim3(S), dim3(512), 0, 0, N, dA, mode, passes, dout);
          else if (G == 4) hipLaunchKernelGGL((k_tile_stream_x4<4>), dim3(S), dim3(512), 0, 0, N, dA, mode, passes, dout);
          else hipLaunchKernelGGL((k_tile_stream_x4<8>), dim3(S), dim3(512), 0, 0, N, dA, mode, passes, dout);
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipDeviceSynchronize());
          float t;
          CHECK(hipEventElapsedTime(&t, e0, e1));
          best = std::min(best, t);
        }
        const double bytes = (double)passes * (N / 32) * (N / 16) * 4096.0 * (mode ? 2 : 1);
        printf("tile-PAIR stream (16-byte accesses) N=%d WGs=%d G=%d pairs %s: %.3f ms  %.1f GB/s per WG\n", N, S, G, mode ? "load+store" : "load only",
               best, bytes / best * 1e-6);
      }
    return 0;
  }
  std::vector<double> G(NN), A(NN);
  unsigned long long sd = 88172645463325252ull;
  auto rnd = [&]() { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; return (double)(sd >> 11) / 9007199254740992.0 - 0.5; };
  for (auto& g : G) g = rnd();
  for (int j = 0; j < N; ++j)
    for (int i = 0; i <= j; ++i) {
      double v = 0;
      for (int t = 0; t < N; ++t) v += G[i + (size_t)N * t] * G[j + (size_t)N * t];
      A[i + (size_t)N * j] = A[j + (size_t)N * i] = v / N + (i == j ? 0.05 : 0.0);
    }
  std::vector<double> R = A;
  if (host_chol_upper(N, R)) { printf("host chol failed\n"); return 1; }
  const int Np = ((N + 15) >> 4) << 4;
  double *dA0, *dA, *dPg;
  int* dpf;
  unsigned char* dact;
  CHECK(hipMalloc(&dA0, S * NN * 8)); CHECK(hipMalloc(&dA, S * NN * 8)); CHECK(hipMalloc(&dPg, (size_t)S * 16 * Np * 8));
  CHECK(hipMalloc(&dpf, S * sizeof(int))); CHECK(hipMalloc(&dact, S));
  std::vector<unsigned char> ones(S, 1);
  CHECK(hipMemcpy(dact, ones.data(), S, hipMemcpyHostToDevice));
  for (int s = 0; s < S; ++s) CHECK(hipMemcpy(dA0 + s * NN, A.data(), NN * 8, hipMemcpyHostToDevice));
  const int nblk = Np >> 4;
  std::vector<double> rhs((size_t)S * N);
  for (auto& v : rhs) v = rnd();
  if (variant == 2) {
    CHECK(hipMalloc(&g_dFinv, (size_t)S * nblk * 256 * 8)); CHECK(hipMalloc(&g_dpfd, S * 8));
    CHECK(hipMalloc(&g_drin, (size_t)S * N * 8)); CHECK(hipMalloc(&g_dzout, (size_t)S * N * 8));
    CHECK(hipMemcpy(g_drin, rhs.data(), (size_t)S * N * 8, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps + 3; ++r) {
    CHECK(hipMemcpyAsync(dA, dA0, S * NN * 8, hipMemcpyDeviceToDevice, st));
    CHECK(hipEventRecord(e0, st));
    CHECK(launch(variant, N, S, dA, dpf, dact, dPg, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float t;
    CHECK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 3) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  // accuracy: every copy against the host factor
  std::vector<double> out(S * NN);
  std::vector<int> pf(S);
  CHECK(hipMemcpy(out.data(), dA, S * NN * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(pf.data(), dpf, S * sizeof(int), hipMemcpyDeviceToHost));
  double worst = 0, lowmax = 0;
  int anyfail = 0;
  for (int s = 0; s < S; ++s) {
    anyfail |= pf[s];
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < N; ++i) {
        const double v = out[s * NN + i + (size_t)N * j], r = R[i + (size_t)N * j];
        if (i <= j) worst = std::max(worst, std::fabs(v - r) / (std::fabs(r) + 1e-3));
        else lowmax = std::max(lowmax, std::fabs(v));
      }
  }
  double zworst = 0, fworst = 0, pfdbad = 0;
  if (variant == 2) {
    // by-products: z = R' \\ r against a long-double forward substitution with the host factor, the block inverses against
    // a long-double inverse of the host factor's diagonal blocks, the failure indices as doubles
    std::vector<double> z((size_t)S * N), fi((size_t)S * nblk * 256), pfd(S);
    CHECK(hipMemcpy(z.data(), g_dzout, z.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(fi.data(), g_dFinv, fi.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(pfd.data(), g_dpfd, S * 8, hipMemcpyDeviceToHost));
    for (int s = 0; s < S; ++s) {
      pfdbad += std::fabs(pfd[s]);
      std::vector<long double> zh(N);
      double zmax = 0;
      for (int i = 0; i < N; ++i) {
        long double v = rhs[(size_t)s * N + i];
        for (int t = 0; t < i; ++t) v -= (long double)R[t + (size_t)N * i] * zh[t];
        zh[i] = v / R[i + (size_t)N * i];
        zmax = std::max(zmax, (double)fabsl(zh[i]));
      }
      for (int i = 0; i < N; ++i) zworst = std::max(zworst, std::fabs(z[(size_t)s * N + i] - (double)zh[i]) / zmax);
      for (int b = 0; b < nblk; ++b) {
        const int b0 = b << 4;
        for (int c = 0; c < 16; ++c) {        // column c of inv(R_bb'): R_bb' x = e_c
          long double x[16];
          for (int ii = 0; ii < 16; ++ii) {
            long double t = ii == c ? 1.0L : 0.0L;
            for (int jj = 0; jj < ii; ++jj) {
              const double rji = (b0 + ii < N && b0 + jj < N) ? R[b0 + jj + (size_t)N * (b0 + ii)] : 0.0;
              t -= (long double)rji * x[jj];
            }
            x[ii] = t / (b0 + ii < N ? R[b0 + ii + (size_t)N * (b0 + ii)] : 1.0);
          }
          for (int ii = 0; ii < 16; ++ii)
            fworst = std::max(fworst, std::fabs(fi[((size_t)s * nblk + b) * 256 + ii * 16 + c] - (double)x[ii]) / (std::fabs((double)x[c]) + 1e-300));
        }
      }
    }
    printf("{\"by_products\": true, \"z_max_rel_err\": %.3e, \"finv_max_rel_err\": %.3e, \"pfd_sum\": %.1f, \"ok\": %s}\n", zworst, fworst, pfdbad,
           (zworst < 1e-11 && fworst < 1e-9 && pfdbad == 0) ? "true" : "false");
  }
  // failure index: make the leading minor of order jf+1 indefinite
  const int jf = std::min(N - 1, (2 * N) / 3);
  std::vector<double> B = A;
  B[jf + (size_t)N * jf] = -1.0;
  std::vector<double> RB = B;
  const int pref = host_chol_upper(N, RB);
  CHECK(hipMemcpy(dA, B.data(), NN * 8, hipMemcpyHostToDevice));
  CHECK(launch(variant, N, 1, dA, dpf, dact, dPg, st));
  CHECK(hipStreamSynchronize(st));
  int pgot = -1;
  CHECK(hipMemcpy(&pgot, dpf, sizeof(int), hipMemcpyDeviceToHost));
  printf("{\"variant\": %d, \"N\": %d, \"S\": %d, \"ms_median\": %.4f, \"ms_min\": %.4f, \"max_rel_err\": %.3e, \"lower_max\": %.1e, \"pfail\": %d, "
         "\"p_indefinite\": [%d, %d], \"ok\": %s}\n", variant, N, S, ms[ms.size() / 2], ms[0], worst, lowmax, anyfail, pgot, pref,
         (worst < 1e-11 && lowmax == 0 && !anyfail && pgot == pref) ? "true" : "false");
#ifdef CHOL_TS
  if (stamps) {
    CHECK(hipMemcpyAsync(dA, dA0, S * NN * 8, hipMemcpyDeviceToDevice, st));
    CHECK(launch(variant, N, S, dA, dpf, dact, dPg, st));
    CHECK(hipStreamSynchronize(st));
    std::vector<long long> ts(4 * 512);
    CHECK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_chol_ts), ts.size() * 8));
    const int nstep = (N + 15) / 16 - 1;
    printf("step: panel  lookahead-done(after panel)  update+barrier   (us; 100 MHz stamps)\n");
    double sp = 0, su = 0;
    for (int k = 0; k < nstep; ++k) {
      const double p = (ts[4 * k + 1] - ts[4 * k + 0]) * 0.01, la = (ts[4 * k + 2] - ts[4 * k + 1]) * 0.01, u = (ts[4 * k + 3] - ts[4 * k + 1]) * 0.01;
      sp += p; su += u;
      if (k % stamps == 0) printf("%3d: %6.2f %6.2f %6.2f\n", k, p, la, u);
    }
    printf("sum panel %.1f us, sum update %.1f us, first stamp to last %.1f us\n", sp, su, (ts[4 * (nstep - 1) + 3] - ts[0]) * 0.01);
    {
      std::vector<long long> gs(8 * 64);
      int gn = 0;
      CHECK(hipMemcpyFromSymbol(gs.data(), HIP_SYMBOL(g_chol_gs), gs.size() * 8));
      CHECK(hipMemcpyFromSymbol(&gn, HIP_SYMBOL(g_chol_gn), sizeof(int)));
      printf("wave 1, first full update, per group (shader cycles): address+issue loads | issue LDS reads+wait | MFMAs | sub+stores | to next group\n");
      for (int g = 0; g < gn && g < 64; ++g)
        printf("  g%02d: %6lld %6lld %6lld %6lld %6lld\n", g, gs[8 * g + 1] - gs[8 * g], gs[8 * g + 2] - gs[8 * g + 1], gs[8 * g + 3] - gs[8 * g + 2],
               gs[8 * g + 4] - gs[8 * g + 3], g + 1 < gn ? gs[8 * (g + 1)] - gs[8 * g + 4] : 0);
    }
    {
      long long tt[16];
      CHECK(hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_chol_tt), sizeof(tt)));
      printf("fast tile of the same look-ahead (shader cycles): block 0..3 (with the inverse of the block before) %lld %lld %lld %lld | inverse rows 12-15 %lld | roots, scaling, tile to LDS %lld\n",
             tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4], tt[6] - tt[5]);
      long long la[8];
      CHECK(hipMemcpyFromSymbol(la, HIP_SYMBOL(g_chol_la), sizeof(la)));
      printf("look-ahead of step 12 (shader cycles): load tile %lld | MFMA + tile to LDS %lld | factor %lld | invert %lld | store %lld\n", la[1] - la[0],
             la[2] - la[1], la[3] - la[2], la[4] - la[3], la[5] - la[4]);
    }
    std::vector<long long> ck(2 * 512);
    CHECK(hipMemcpyFromSymbol(ck.data(), HIP_SYMBOL(g_chol_clk), ck.size() * 8));
    printf("shader clock over the factorisation: %.0f MHz (clock64 ticks / wall_clock64 time)\n",
           (double)(ck[2 * (nstep - 1) + 1] - ck[0]) / ((ts[4 * (nstep - 1) + 3] - ts[0]) * 0.01));
  }
#endif
  return 0;
}
