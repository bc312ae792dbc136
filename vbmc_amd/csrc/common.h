// Shared host-side plumbing for libvbmc_hip.so (context, device buffers, error reporting).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vbmc_hip.h"

// the shapes the validation accepts (vbmc_get_limits reports them)
#define VBMC_LIM_D 32
#define VBMC_LIM_K 512
#define VBMC_LIM_NA 256
#define VBMC_LIM_MEANFUN_MASK ((1 << 0) | (1 << 1) | (1 << 4))

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// Pool of device blocks for the per-call temporaries of the GP entry points (abi_gp.hip): hipMalloc/hipFree
// cost ~0.1 ms each and synchronise the device, which dominated small calls.  Blocks are handed out only
// between a call's start and its final stream synchronisation, so reuse across calls is safe.
struct PoolBlk {
  void* p = nullptr;
  size_t cap = 0;
  bool busy = false;
};

struct vbmc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // the expected log joint runs beside the entropy kernel on a second, lower-priority stream (fork after k_prep, join
  // before the variance / finalize kernels): it fills the entropy kernel's tail, or, for a single chain, the idle CUs
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap = true;
  bool prof_alone = false;     // vbmc_ctx_set_profiling(ctx, 2): nothing forked beside the dominant kernel while it is timed
  void* bounce = nullptr;                       // (unused since round 2: see d2h_bounced)
  hipEvent_t bounce_ev[2] = {nullptr, nullptr};
  std::string err;
  int num_cu = 256;
  // grow-only scratch
  DevBuf theta, prep, entp, ljpart, entpart, out, eps, bnd, vpfix, misc, varbuf, zbuf;
  // pinned staging for small D2H/H2D
  void* pin = nullptr;
  size_t pin_cap = 0;
  // profiling
  bool profiling = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  double last_ent_ms = 0.0, last_lj_ms = 0.0;
  std::vector<PoolBlk> pool;
  size_t pool_bytes = 0;
  // pipelined evaluations (vbmc_elbo_submit / vbmc_elbo_collect, abi_elbo.hip): each of the two slots has its own pinned
  // staging block (swapped into `pin` for the duration of a submit), the plan of the pass in flight and its completion event
  void* slot_pin[2] = {nullptr, nullptr};
  size_t slot_pin_cap[2] = {0, 0};
  hipEvent_t slot_ev[2] = {nullptr, nullptr};
  void* slot_plan[2] = {nullptr, nullptr};     // ElboPlan*
  bool slot_busy[2] = {false, false};
  // Round 4: FOUR slots on TWO streams.  Slot s runs on child context s & 1 (own stream, own scratch, created on first use) as that
  // child's slot s >> 1 -- two passes queued per stream, two streams: the small kernels at the head and tail of one pass and the last round
  // of waves of another share the chip instead of queuing behind each other, and a stream never runs dry while the host collects and
  // re-submits (tools/archive/r4_two_ctx.py: the headline step 2.45 -> 2.41 ms, eight restarts 0.358 -> 0.336, four 0.221 -> 0.183).
  // slot_sub[]: the children; slot_where[s] / slot_inner[s]: the context and its slot the pass in flight was enqueued on (this context
  // itself, slots 0 and 1 only, for the variance forms and under VBMC_SLOT_STREAMS=0); slot_xev[s] orders the child's stream after
  // everything enqueued on this context's stream before the submit; slot_yev / slot_zev serve vbmc_elbo_multi_submit, whose exchange
  // stays on this context's stream (ordered after the pass; its end).
#define VBMC_SLOTS 4
  vbmc_ctx* slot_sub[2] = {nullptr, nullptr};
  vbmc_ctx* slot_where[VBMC_SLOTS] = {};
  int slot_inner[VBMC_SLOTS] = {};
  hipEvent_t slot_xev[VBMC_SLOTS] = {};
  hipEvent_t slot_yev[VBMC_SLOTS] = {};
  hipEvent_t slot_zev[VBMC_SLOTS] = {};
  bool is_sub = false;
  // entropy-only evaluations (vbmc_elbo_batch with gp == NULL): a one-point surrogate with alpha = 0 and a zero mean
  // function per dimension, whose expected log joint is exactly 0
  vbmc_gp* null_gp[33] = {};
};

// The context's second stream (low priority: what is forked onto it fills in around the kernels of the first).  A context that never
// forks -- the children behind the pipeline slots -- holds ONE stream (overlap = false).  The runtime maps streams onto a few hardware
// queues (four unless GPU_MAX_HW_QUEUES says otherwise), and two busy streams that share a queue run one after the other.
static inline bool ctx_aux(vbmc_ctx* ctx) {
  if (!ctx->overlap) return false;
  if (ctx->aux) return true;
  int lo = 0, hi = 0;   // numerically larger = lower priority
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, lo) != hipSuccess) {
    (void)hipGetLastError();
    ctx->aux = nullptr;
    ctx->overlap = false;
    return false;
  }
  return true;
}

// Passes submitted through the pipeline slots run on the child contexts' own streams, ordered after the parent's stream only at submit.
// Whatever then changes or releases something such a pass may still read -- a pooled surrogate given back (vbmc_gp_free), its noise
// model rewritten (vbmc_gp_set_noise), a rank-one update reusing pool blocks -- first waits for the child streams that hold a pass
// (round 5, ADVICE r4: the header's "collect first" was the only protection).  The slots stay collectable.
static inline void ctx_drain_slots(vbmc_ctx* ctx) {
  if (!ctx) return;
  for (int sl = 0; sl < VBMC_SLOTS; ++sl) {
    vbmc_ctx* w = ctx->slot_where[sl];
    if (w && w != ctx && w->slot_busy[ctx->slot_inner[sl]]) (void)hipStreamSynchronize(w->stream);
  }
  for (int sl = 0; sl < 2; ++sl)
    if (ctx->slot_busy[sl]) { (void)hipStreamSynchronize(ctx->stream); break; }
}

static inline hipError_t pool_get(vbmc_ctx* ctx, size_t bytes, void** out) {
  if (bytes < 256) bytes = 256;
  int best = -1;
  for (int i = 0; i < (int)ctx->pool.size(); ++i) {
    const PoolBlk& b = ctx->pool[i];
    if (b.busy || b.cap < bytes || b.cap > 4 * bytes + (1u << 20)) continue;
    if (best < 0 || b.cap < ctx->pool[best].cap) best = i;
  }
  if (best >= 0) { ctx->pool[best].busy = true; *out = ctx->pool[best].p; return hipSuccess; }
  // keep the cache bounded: drop idle blocks (largest first) once more than 16 GiB are held
  const size_t limit = (size_t)16 << 30;
  while (ctx->pool_bytes + bytes > limit) {
    int big = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i)
      if (!ctx->pool[i].busy && (big < 0 || ctx->pool[i].cap > ctx->pool[big].cap)) big = i;
    if (big < 0) break;
    (void)hipFree(ctx->pool[big].p);
    ctx->pool_bytes -= ctx->pool[big].cap;
    ctx->pool.erase(ctx->pool.begin() + big);
  }
  const size_t want = bytes + bytes / 8;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) return e;
  PoolBlk b;
  b.p = p; b.cap = want; b.busy = true;
  ctx->pool.push_back(b);
  ctx->pool_bytes += want;
  *out = p;
  return hipSuccess;
}

static inline void pool_put(vbmc_ctx* ctx, void* p) {
  for (auto& b : ctx->pool)
    if (b.p == p) { b.busy = false; return; }
}

struct vbmc_gp {
  int N = 0, D = 0, S = 0, Nhyp = 0, Ncov = 0, Nnoise = 0, meanfun = 0;
  bool hasL = false;
  bool pooled = false;      // device blocks below belong to the creating context's pool (pool_get / pool_put)
  double* X = nullptr;      // N x D col-major
  double* alpha = nullptr;  // N x S
  double* L = nullptr;      // N x N x S
  double* gpc = nullptr;    // S x GPC_STRIDE derived per-sample constants (see elbo.hip)
  double* hyp = nullptr;    // Nhyp x S
  double* d_sn2 = nullptr;  // S  sn2_eff
  unsigned char* d_lchol = nullptr;  // S
  double* d_mult = nullptr;   // S  sn2_mult (prediction)
  double* d_finv = nullptr;   // S x nblk x 256: inverses of the 16 x 16 diagonal blocks of L' (trsm_mfma.h)
  mutable double* d_tinv = nullptr;  // S x N x N: inv(L') per Lchol sample (built on the first prediction, abi_gp.hip)
  double* d_meanX = nullptr;  // D  column means of X (sq_dist centring in gplite_pred)
  void* blk_in = nullptr;     // in_views: the one pooled block X, hyp, gpc, d_sn2, d_lchol, d_mult and d_meanX are windows of
  bool in_views = false;
  int noisefun[3] = {1, 0, 0};
  bool has_noise = false;
  std::vector<double> sn2_eff;
  std::vector<uint8_t> Lchol;
  std::vector<double> hyp_host;
};

static inline vbmc_status set_err(vbmc_ctx* ctx, vbmc_status st, const char* fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  return st;
}

#define HIP_TRY(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      (void)hipGetLastError(); /* do not leave the error for the next call's launch checks */ \
      return set_err(ctx, VBMC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                    \
    }                                                                                        \
  } while (0)

static inline vbmc_status ensure(vbmc_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return VBMC_OK;
  if (b.p) HIP_TRY(ctx, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4 + 256;
  HIP_TRY(ctx, hipMalloc(&b.p, want));
  b.cap = want;
  return VBMC_OK;
}

// Large device -> pageable-host copy.  Round 1 staged it through two pinned bounce buffers with a CPU memcpy per chunk; measured
// on the box in round 2 (tools/d2h_probe.py, 25.6 MB = gp.post(1:20).L at N = 400): the runtime's own path into pageable memory
// runs at 55 GB/s into touched pages and 24 GB/s into a fresh array (first-touch faults), the bounce path at ~11 GB/s (bound by
// the single-threaded memcpy) -- so the plain copy it is.  Synchronises the stream before returning.
static inline vbmc_status d2h_bounced(vbmc_ctx* ctx, void* dst, const void* src, size_t bytes) {
  HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VBMC_OK;
}

// Wait for a stream at the end of a latency-bound call (one gplite_nlZ evaluation of a slice-sampling chain, one gplite_post): poll
// for up to 300 us before falling back to the blocking wait -- the blocking wait's wake-up costs 10-20 us of a 0.4 ms call whose
// caller is about to issue the next one.
//  The poll is bounded by the CLOCK -- 300 us -- not by a count of queries (4000 of them were a few ms of a host core when the
// work was long), and a caller that knows its work is long (N in the thousands: the factorisation alone is milliseconds) passes
// big = true and goes straight to the blocking wait.
static inline hipError_t stream_wait_latency(hipStream_t st, bool big = false) {
  if (!big) {
    const auto t0 = std::chrono::steady_clock::now();
    do {
      const hipError_t e = hipStreamQuery(st);
      if (e != hipErrorNotReady) return e;
    } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(300));
  }
  return hipStreamSynchronize(st);
}

static inline vbmc_status ensure_pin(vbmc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pin_cap) return VBMC_OK;
  if (ctx->pin) HIP_TRY(ctx, hipHostFree(ctx->pin));
  ctx->pin = nullptr;
  ctx->pin_cap = 0;
  size_t want = bytes * 2 + 4096;
  HIP_TRY(ctx, hipHostMalloc(&ctx->pin, want, hipHostMallocDefault));
  ctx->pin_cap = want;
  return VBMC_OK;
}
