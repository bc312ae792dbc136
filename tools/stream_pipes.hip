// Which streams of one process dispatch concurrently?  (GPU box)  hipcc --offload-arch=gfx950 -O2 tools/stream_pipes.hip -o /tmp/stream_pipes
// For every ordered pair (i, j) of NS streams created in order: a long kernel on stream i (ROUNDS rounds of workgroups that leave room on
// every compute unit: a large LDS block per workgroup, one wave each, spinning SPIN_US), then at once a one-wave kernel on stream j.
// Printed: when the small kernel finished, as a fraction of the long one's duration.  ~0: the two streams dispatch side by side;
// ~(ROUNDS-1)/ROUNDS: the small kernel waited until the long one's last workgroup was handed out (same dispatcher, in order);
// ~1: it waited for the long kernel to finish.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(long long ticks, int* sink) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (ticks < 0) sink[0] = lds[threadIdx.x];
}

int main(int argc, char** argv) {
  const int NS = argc > 1 ? atoi(argv[1]) : 8;      // normal-priority streams
  const int NLOW = argc > 2 ? atoi(argv[2]) : 1;    // + low-priority streams (created after the first normal one, like a context's second stream)
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  std::vector<hipStream_t> st;
  std::vector<int> low;
  for (int i = 0; i < NS; ++i) {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    st.push_back(s); low.push_back(0);
    if (i == 0)
      for (int l = 0; l < NLOW; ++l) { CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo)); st.push_back(s); low.push_back(1); }
  }
  const int n = (int)st.size();
  int* sink; CK(hipMalloc(&sink, 4));
  int wc_khz = 100000;
  CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
  const double SPIN_US = 30.0;
  const int ROUNDS = 4;
  const long long ticks = (long long)(SPIN_US * 1e-6 * wc_khz * 1e3);
  const int lds = 36 * 1024;   // 4 workgroups per compute unit (160 KB LDS)
  CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = ncu * 4 * ROUNDS;
  for (int i = 0; i < n; ++i) {   // touch every stream once: queues exist
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], 1, sink);
    CK(hipStreamSynchronize(st[i]));
  }
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  printf("CUs %d, wall clock %d kHz; long kernel: %d workgroups x %.0f us in %d rounds; streams in creation order (L = low priority)\n", ncu, wc_khz, grid, SPIN_US, ROUNDS);
  printf("row = stream of the long kernel, column = stream of the small kernel; entry = finish(small) / duration(long)\n      ");
  for (int j = 0; j < n; ++j) printf("  %2d%c ", j, low[j] ? 'L' : ' ');
  printf("\n");
  for (int i = 0; i < n; ++i) {
    printf("%2d%c   ", i, low[i] ? 'L' : ' ');
    for (int j = 0; j < n; ++j) {
      if (i == j) { printf("   -  "); continue; }
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[i]));
        hipLaunchKernelGGL(k_spin, dim3(grid), dim3(64), lds, st[i], ticks, sink);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[j], ticks / 30, sink);
        CK(hipEventRecord(e1, st[j]));
        CK(hipEventRecord(e2, st[i]));
        CK(hipDeviceSynchronize());
        float ts = 0, tb = 0;
        CK(hipEventElapsedTime(&ts, e0, e1));
        CK(hipEventElapsedTime(&tb, e0, e2));
        if (ts / tb < best) best = ts / tb;
      }
      printf(" %5.2f", best);
    }
    printf("\n");
  }
  return 0;
}
