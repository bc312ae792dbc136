// C-ABI entry points of libvbmc_hip.so for more than one GPU (include/vbmc_hip.h, "communicator"): RCCL reached from INSIDE
// the library, so that a host without torch.distributed -- the MATLAB process behind matlab/vbmc_hip_mex.cpp -- can drive all
// eight GPUs of a node, and so that the one exchange step of the path (the all-gather of the restarts' ELCBO values,
// misc/vpsieve_vbmc.m:74-83 sharded as SURVEY 8e prescribes) runs device to device over xGMI.
//
// Two ways to form a communicator, one data path:
//   vbmc_comm_create_all   ONE process, G devices: a context and an RCCL rank per device (ncclCommInitAll);
//   vbmc_comm_create_rank  one process PER device (torch.distributed.run, mpirun): every process brings its context and
//                          the 128-byte id rank 0 obtained from vbmc_comm_unique_id (ncclCommInitRank).
// librccl is opened with dlopen on first use: a single-GPU user needs no RCCL, and a process that already holds a copy
// (PyTorch loads its own) keeps exactly that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <string>
#include <vector>

namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// nullptr + message when RCCL cannot be opened
RcclApi* rccl_api(std::string& err) {
  static RcclApi api;
  static bool tried = false;
  if (api.handle) return &api;
  if (tried) { err = "librccl could not be loaded"; return nullptr; }
  tried = true;
  if (const char* f = getenv("VBMC_RCCL_DISABLE")) {   // tests: a host without librccl (every comm entry point must report, not crash)
    if (f[0] == '1') { err = "dlopen(librccl): disabled by VBMC_RCCL_DISABLE"; return nullptr; }
  }
  // a copy this process holds already (PyTorch's) first, then ROCm's
  const char* held[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : held)
    if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
  const char* fresh[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char* n : fresh)
    if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!api.handle) {
    const char* e = dlerror();      // once: the call returns the message AND clears it
    err = std::string("dlopen(librccl): ") + (e ? e : "not found");
    return nullptr;
  }
#define VB_SYM(name)                                                      \
  api.name = (decltype(api.name))dlsym(api.handle, "nccl" #name);         \
  if (!api.name) { err = "librccl lacks nccl" #name; api.handle = nullptr; return nullptr; }
  VB_SYM(GetUniqueId) VB_SYM(CommInitRank) VB_SYM(CommInitAll) VB_SYM(CommDestroy) VB_SYM(AllGather) VB_SYM(GroupStart)
  VB_SYM(GroupEnd) VB_SYM(GetErrorString)
#undef VB_SYM
  return &api;
}

}  // namespace

struct vbmc_comm {
  int n = 0;        // devices this process drives
  int world = 1;    // ranks of the communicator
  int rank0 = 0;    // rank of local device 0 (local device i is rank rank0 + i)
  bool own_ctx = false;
  std::vector<vbmc_ctx*> ctx;
  std::vector<ncclComm_t> comm;
  std::vector<double*> d_send, d_recv;   // exchange blocks, grown on demand (cap doubles per rank)
  size_t cap = 0;
  RcclApi* api = nullptr;
  std::string err;
  // Several ranks on ONE device (vbmc_comm_create_all with a device listed more than once: a multi-device run rehearsed on a one-device
  // box, tests/test_gpu_comm.py).  RCCL refuses that ("Duplicate GPU detected"), in one process and across processes alike, so this form
  // exchanges by device-to-device copies: rank j's block is ready at an event on its stream, every rank's stream waits for all of them
  // and copies the blocks into its own receive buffer in rank order.  Same blocks, same order, no arithmetic: an all-gather.
  bool local_copy = false;
  std::vector<hipEvent_t> lc_ev;
  // the pipelined form's exchange stream per local device: the collectives of consecutive batches in issue order on ONE stream that
  // holds nothing else (a pass on a slot stream is ordered after the context's stream, abi_elbo.hip: slot_ctx -- with the exchange on
  // the context's stream every pass would queue behind the previous batch's exchange)
  std::vector<hipStream_t> xs;
  // pipelined form (vbmc_elbo_multi_submit / _collect): per slot its own exchange blocks, a pinned landing block for the gathered
  // vectors and what collect needs to know about the submitted batch
  struct Slot {
    bool busy = false;
    int R = 0, T = 0, P = 0;
    std::vector<int> n;                       // restarts of local device i in the submitted batch
    std::vector<vbmc_status> st;              // ... and how its submit went
    std::vector<std::vector<double>> theta;   // its columns of theta, gathered (capacity kept between calls)
    std::vector<vbmc_elbo_args> sub;
    std::vector<double*> d_send, d_recv;
    std::vector<hipStream_t> xst;             // the stream local device i's exchange was enqueued on
    size_t cap = 0;
    double* h_gather = nullptr;
    size_t h_cap = 0;
  } slot[VBMC_SLOTS];
};

static vbmc_status comm_err(vbmc_comm* c, vbmc_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return st;
}
#define COMM_HIP(c, call)                                                                                       \
  do {                                                                                                          \
    hipError_t e_ = (call);                                                                                     \
    if (e_ != hipSuccess) { (void)hipGetLastError(); return comm_err(c, VBMC_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); } \
  } while (0)
#define COMM_NCCL(c, call)                                                                                      \
  do {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                   \
    if (r_ != ncclSuccess) return comm_err(c, VBMC_ERR_HIP, "%s: %s", #call, (c)->api->GetErrorString(r_));     \
  } while (0)

static vbmc_status comm_reserve(vbmc_comm* c, size_t count) {
  if (count <= c->cap) return VBMC_OK;
  const size_t cap = std::max<size_t>(count + count / 2, 256);
  for (int i = 0; i < c->n; ++i) {
    COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
    COMM_HIP(c, hipStreamSynchronize(c->ctx[i]->stream));
    if (c->d_send[i]) COMM_HIP(c, hipFree(c->d_send[i]));
    if (c->d_recv[i]) COMM_HIP(c, hipFree(c->d_recv[i]));
    c->d_send[i] = c->d_recv[i] = nullptr;
    COMM_HIP(c, hipMalloc((void**)&c->d_send[i], cap * sizeof(double)));
    COMM_HIP(c, hipMalloc((void**)&c->d_recv[i], cap * c->world * sizeof(double)));
  }
  c->cap = cap;
  return VBMC_OK;
}

extern "C" const char* vbmc_comm_last_error(const vbmc_comm* c) { return c ? c->err.c_str() : "null communicator"; }
extern "C" int vbmc_comm_size(const vbmc_comm* c) { return c ? c->world : 0; }
extern "C" int vbmc_comm_local(const vbmc_comm* c) { return c ? c->n : 0; }
extern "C" int vbmc_comm_rank(const vbmc_comm* c) { return c ? c->rank0 : -1; }
extern "C" vbmc_ctx* vbmc_comm_ctx(vbmc_comm* c, int i) { return (c && i >= 0 && i < c->n) ? c->ctx[i] : nullptr; }

extern "C" void vbmc_comm_destroy(vbmc_comm* c) {
  if (!c) return;
  for (int i = 0; i < c->n; ++i) {
    if (c->ctx[i]) { (void)hipSetDevice(c->ctx[i]->device); (void)hipStreamSynchronize(c->ctx[i]->stream); }
    if (i < (int)c->comm.size() && c->comm[i] && c->api) (void)c->api->CommDestroy(c->comm[i]);
    if (i < (int)c->lc_ev.size() && c->lc_ev[i]) (void)hipEventDestroy(c->lc_ev[i]);
    if (c->d_send[i]) (void)hipFree(c->d_send[i]);
    if (c->d_recv[i]) (void)hipFree(c->d_recv[i]);
    for (auto& sl : c->slot) {
      if (i < (int)sl.d_send.size() && sl.d_send[i]) (void)hipFree(sl.d_send[i]);
      if (i < (int)sl.d_recv.size() && sl.d_recv[i]) (void)hipFree(sl.d_recv[i]);
    }
    if (i < (int)c->xs.size() && c->xs[i]) { (void)hipStreamSynchronize(c->xs[i]); (void)hipStreamDestroy(c->xs[i]); }
    if (c->own_ctx && c->ctx[i]) vbmc_ctx_destroy(c->ctx[i]);
  }
  for (auto& sl : c->slot)
    if (sl.h_gather) (void)hipHostFree(sl.h_gather);
  delete c;
}

extern "C" vbmc_status vbmc_comm_create_all(int ndev, const int* devices, vbmc_comm** out) {
  if (!out || ndev < 1) return VBMC_ERR_INVALID;
  *out = nullptr;
  vbmc_comm* c = new vbmc_comm();
  c->n = c->world = ndev; c->rank0 = 0; c->own_ctx = true;
  c->ctx.assign(ndev, nullptr); c->comm.assign(ndev, nullptr); c->d_send.assign(ndev, nullptr); c->d_recv.assign(ndev, nullptr);
  std::vector<int> devs(ndev);
  for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
  for (int i = 0; i < ndev; ++i) {
    vbmc_status st = vbmc_ctx_create(devs[i], nullptr, &c->ctx[i]);
    if (st != VBMC_OK) { vbmc_comm_destroy(c); return st; }
  }
  bool dup = false;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j) dup = dup || devs[i] == devs[j];
  if (dup) {      // more than one rank on a device: the copy exchange (see vbmc_comm::local_copy)
    c->local_copy = true;
    c->lc_ev.assign(ndev, nullptr);
    for (int i = 0; i < ndev; ++i) {
      if (hipSetDevice(devs[i]) != hipSuccess || hipEventCreateWithFlags(&c->lc_ev[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); vbmc_comm_destroy(c); return VBMC_ERR_HIP;
      }
    }
    *out = c;
    return VBMC_OK;
  }
  c->api = rccl_api(c->err);
  if (!c->api) { vbmc_comm_destroy(c); return VBMC_ERR_HIP; }
  ncclResult_t r = c->api->CommInitAll(c->comm.data(), ndev, devs.data());
  if (r != ncclSuccess) { vbmc_comm_destroy(c); return VBMC_ERR_HIP; }
  *out = c;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_comm_unique_id(void* id128) {
  if (!id128) return VBMC_ERR_INVALID;
  std::string err;
  RcclApi* api = rccl_api(err);
  if (!api) return VBMC_ERR_HIP;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) return VBMC_ERR_HIP;
  memcpy(id128, &id, sizeof id);
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_comm_create_rank(vbmc_ctx* ctx, int rank, int world, const void* id128, vbmc_comm** out) {
  if (!out || !ctx || !id128 || world < 1 || rank < 0 || rank >= world) return VBMC_ERR_INVALID;
  *out = nullptr;
  vbmc_comm* c = new vbmc_comm();
  c->n = 1; c->world = world; c->rank0 = rank; c->own_ctx = false;
  c->ctx.assign(1, ctx); c->comm.assign(1, nullptr); c->d_send.assign(1, nullptr); c->d_recv.assign(1, nullptr);
  c->api = rccl_api(c->err);
  if (!c->api) { set_err(ctx, VBMC_ERR_HIP, "%s", c->err.c_str()); vbmc_comm_destroy(c); return VBMC_ERR_HIP; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  if (hipSetDevice(ctx->device) != hipSuccess) { vbmc_comm_destroy(c); return VBMC_ERR_HIP; }
  ncclResult_t r = c->api->CommInitRank(&c->comm[0], world, id, rank);
  if (r != ncclSuccess) {
    set_err(ctx, VBMC_ERR_HIP, "ncclCommInitRank: %s", c->api->GetErrorString(r));
    vbmc_comm_destroy(c);
    return VBMC_ERR_HIP;
  }
  *out = c;
  return VBMC_OK;
}

// the collective itself: every local device contributes `count` doubles, every device receives world * count in rank order
static vbmc_status comm_allgather_enqueue(vbmc_comm* c, const double* const* d_send, double* const* d_recv, size_t count,
                                          const hipStream_t* on = nullptr) {
  if (c->local_copy) {
    for (int j = 0; j < c->n; ++j) {
      COMM_HIP(c, hipSetDevice(c->ctx[j]->device));
      COMM_HIP(c, hipEventRecord(c->lc_ev[j], on ? on[j] : c->ctx[j]->stream));
    }
    for (int i = 0; i < c->n; ++i) {
      hipStream_t si = on ? on[i] : c->ctx[i]->stream;
      COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
      for (int j = 0; j < c->n; ++j) {
        if (j != i) COMM_HIP(c, hipStreamWaitEvent(si, c->lc_ev[j], 0));
        COMM_HIP(c, hipMemcpyAsync(d_recv[i] + (size_t)j * count, d_send[j], count * sizeof(double), hipMemcpyDeviceToDevice, si));
      }
    }
    return VBMC_OK;
  }
  COMM_NCCL(c, c->api->GroupStart());
  for (int i = 0; i < c->n; ++i) {
    ncclResult_t r = c->api->AllGather(d_send[i], d_recv[i], count, ncclDouble, c->comm[i], on ? on[i] : c->ctx[i]->stream);
    if (r != ncclSuccess) { (void)c->api->GroupEnd(); return comm_err(c, VBMC_ERR_HIP, "ncclAllGather: %s", c->api->GetErrorString(r)); }
  }
  COMM_NCCL(c, c->api->GroupEnd());
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_allgather_f64(vbmc_comm* c, const double* const* d_send, double* const* d_recv, size_t count) {
  if (!c) return VBMC_ERR_INVALID;
  if (!d_send || !d_recv || count == 0) return comm_err(c, VBMC_ERR_INVALID, "vbmc_allgather_f64: null blocks / zero count");
  { vbmc_status s_ = comm_allgather_enqueue(c, d_send, d_recv, count); if (s_) return s_; }
  for (int i = 0; i < c->n; ++i) {
    COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
    COMM_HIP(c, hipStreamSynchronize(c->ctx[i]->stream));
  }
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_allgather_host_f64(vbmc_comm* c, const double* send, double* recv, size_t count) {
  if (!c) return VBMC_ERR_INVALID;
  if (!send || !recv || count == 0) return comm_err(c, VBMC_ERR_INVALID, "vbmc_allgather_host_f64: null blocks / zero count");
  { vbmc_status s_ = comm_reserve(c, count); if (s_) return s_; }
  for (int i = 0; i < c->n; ++i) {
    COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
    COMM_HIP(c, hipMemcpyAsync(c->d_send[i], send + (size_t)i * count, count * sizeof(double), hipMemcpyHostToDevice, c->ctx[i]->stream));
  }
  { vbmc_status s_ = comm_allgather_enqueue(c, c->d_send.data(), c->d_recv.data(), count); if (s_) return s_; }
  COMM_HIP(c, hipSetDevice(c->ctx[0]->device));
  COMM_HIP(c, hipMemcpyAsync(recv, c->d_recv[0], count * c->world * sizeof(double), hipMemcpyDeviceToHost, c->ctx[0]->stream));
  for (int i = 0; i < c->n; ++i) {
    COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
    COMM_HIP(c, hipStreamSynchronize(c->ctx[i]->stream));
  }
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_gp_upload_all(vbmc_comm* c, int N, int D, int S, int Nhyp, int Ncov, int Nnoise, int meanfun,
                                          const double* X, const double* hyp, const double* alpha, const double* L, const double* sW1,
                                          const uint8_t* Lchol, vbmc_gp** gps) {
  if (!c || !gps) return VBMC_ERR_INVALID;
  for (int i = 0; i < c->n; ++i) gps[i] = nullptr;
  for (int i = 0; i < c->n; ++i) {
    vbmc_status st = vbmc_gp_upload(c->ctx[i], N, D, S, Nhyp, Ncov, Nnoise, meanfun, X, hyp, alpha, L, sW1, Lchol, &gps[i]);
    if (st != VBMC_OK) {
      comm_err(c, st, "device %d: %s", c->ctx[i]->device, vbmc_last_error(c->ctx[i]));
      for (int j = 0; j < i; ++j) { vbmc_gp_free(c->ctx[j], gps[j]); gps[j] = nullptr; }
      return st;
    }
  }
  return VBMC_OK;
}

extern "C" void vbmc_gp_free_all(vbmc_comm* c, vbmc_gp** gps) {
  if (!c || !gps) return;
  for (int i = 0; i < c->n; ++i)
    if (gps[i]) { vbmc_gp_free(c->ctx[i], gps[i]); gps[i] = nullptr; }
}

// F and varG of the n restarts of one device, out of its packed result records, into its exchange block [F (P) | varG (P)];
// slots beyond n carry NaN
__global__ void k_comm_pick(int n, int P, size_t OS, const double* __restrict__ out, double* __restrict__ send) {
  VB_SMALL_PRIO();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  send[i] = i < n ? out[(size_t)i * OS] : nan;
  send[P + i] = i < n ? out[(size_t)i * OS + 3] : nan;
}

// The R restarts of one batch dealt over the ranks of the communicator: rank g evaluates restarts g, g + G, g + 2G, ...
// (the independent iterations of misc/vpsieve_vbmc.m:74-78), F and varG of ALL restarts are all-gathered device to device, and
// every rank ends up with the identical vectors -- the identical stable sort, the identical sieve order, with no broadcast.
extern "C" vbmc_status vbmc_elbo_batch_multi(vbmc_comm* c, const vbmc_gp* const* gps, const vbmc_elbo_args* a) {
  if (!c) return VBMC_ERR_INVALID;
  if (!gps || !a) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_batch_multi: null surrogates / args");
  if (a->struct_size != sizeof(vbmc_elbo_args)) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_args.struct_size (ABI mismatch)");
  if (a->eps_mode != 0 && !(a->eps_mode == 1 && a->eps_shared))
    return comm_err(c, VBMC_ERR_UNSUPPORTED, "vbmc_elbo_batch_multi: device RNG (eps_mode 0) or one shared host block of draws only");
  if (a->restart_offset != 0 || a->restart_stride > 1) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_batch_multi deals the restarts itself");
  //  per-hyper-sample GRADIENTS are served by vbmc_elbo_batch alone: elbo_plan would accept the fields, this path neither
  // allocates nor reads them back -- refuse instead of returning OK with the caller's buffers untouched
  if (a->dG_s || a->dvarG_s) return comm_err(c, VBMC_ERR_UNSUPPORTED, "vbmc_elbo_batch_multi: per-hyper-sample gradients (dG_s, dvarG_s) only through vbmc_elbo_batch");
  const int G = c->world, R = a->R, K = a->K, D = a->D;
  if (R < 1 || K < 1 || D < 1) return comm_err(c, VBMC_ERR_INVALID, "D, K, R must be positive");
  int T = 0;
  { const int n[4] = {D * K, K, D, K}; for (int g = 0; g < 4; ++g) if (a->optimize[g]) T += n[g]; }
  if (T > 0 && !a->theta) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_batch_multi: theta is null");
  const int P = (R + G - 1) / G;                    // restarts per rank, padded
  { vbmc_status s_ = comm_reserve(c, 2 * (size_t)P); if (s_) return s_; }
  // A failure that is local to one rank (a resource error, a missing surrogate) must not leave the other ranks waiting in the
  // collective: the rank still enters it, contributing an all-NaN block, and reports its error after the exchange.
  vbmc_status local_fail = VBMC_OK;

  struct Local {
    int n = 0, S = 0;
    vbmc_elbo_args sub;
    ElboPlan plan;
    std::vector<double> theta, F, dF, G, H, dG, dH, varG, varGss, I, J, Gs, vGs;
  };
  std::vector<Local> loc(c->n);
  // ---- every local device: stage its restarts, enqueue the pass and the pick of (F, varG) into the exchange block
  for (int i = 0; i < c->n; ++i) {
    Local& L = loc[i];
    const int g = c->rank0 + i;
    L.n = g < R ? (R - g + G - 1) / G : 0;
    vbmc_ctx* ctx = c->ctx[i];
    COMM_HIP(c, hipSetDevice(ctx->device));
    if (L.n == 0) {   // more ranks than restarts: an all-NaN block
      hipLaunchKernelGGL(k_comm_pick, dim3((P + 63) / 64), dim3(64), 0, ctx->stream, 0, P, (size_t)1, (const double*)nullptr, c->d_send[i]);
      COMM_HIP(c, hipGetLastError());
      continue;
    }
    if (!gps[i]) {
      local_fail = comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_batch_multi: no surrogate for local device %d", i);
      L.n = 0;
      hipLaunchKernelGGL(k_comm_pick, dim3((P + 63) / 64), dim3(64), 0, ctx->stream, 0, P, (size_t)1, (const double*)nullptr, c->d_send[i]);
      continue;
    }
    L.S = gps[i]->S;
    L.sub = *a;
    L.sub.R = L.n;
    L.sub.restart_offset = g; L.sub.restart_stride = G;
    L.theta.resize((size_t)T * L.n);
    for (int q = 0; q < L.n; ++q) memcpy(&L.theta[(size_t)q * T], a->theta + (size_t)(g + (size_t)q * G) * T, T * sizeof(double));
    L.sub.theta = L.theta.data();
    auto want = [&](const double* p, std::vector<double>& v, size_t per) -> double* {
      if (!p) return nullptr;
      v.resize(per * L.n);
      return v.data();
    };
    L.sub.F = want(a->F, L.F, 1); L.sub.G = want(a->G, L.G, 1); L.sub.H = want(a->H, L.H, 1);
    L.sub.varG = want(a->varG, L.varG, 1); L.sub.varGss = want(a->varGss, L.varGss, 1);
    L.sub.dF = want(a->dF, L.dF, T); L.sub.dG = want(a->dG, L.dG, T); L.sub.dH = want(a->dH, L.dH, T);
    L.sub.I_sk = want(a->I_sk, L.I, (size_t)L.S * K); L.sub.J_sjk = want(a->J_sjk, L.J, (size_t)L.S * K * K);
    L.sub.G_s = want(a->G_s, L.Gs, L.S); L.sub.varG_s = want(a->varG_s, L.vGs, L.S);
    vbmc_status st = elbo_plan(ctx, gps[i], &L.sub, L.plan, 0, true);
    if (!st) st = elbo_enqueue(ctx, gps[i], L.plan, a->seed);
    if (st) {
      local_fail = comm_err(c, st, "device %d: %s", ctx->device, vbmc_last_error(ctx));
      L.n = 0;
      hipLaunchKernelGGL(k_comm_pick, dim3((P + 63) / 64), dim3(64), 0, ctx->stream, 0, P, (size_t)1, (const double*)nullptr, c->d_send[i]);
      continue;
    }
    const size_t OS = OUT_HDR + 3 * (size_t)T;
    hipLaunchKernelGGL(k_comm_pick, dim3((P + 63) / 64), dim3(64), 0, ctx->stream, L.n, P, OS, (const double*)L.plan.d_out, c->d_send[i]);
    COMM_HIP(c, hipGetLastError());
  }
  // ---- the exchange: [F | varG] of every rank to every rank, on the streams the passes run on
  { vbmc_status s_ = comm_allgather_enqueue(c, c->d_send.data(), c->d_recv.data(), 2 * (size_t)P); if (s_) return s_; }
  if (local_fail) {
    const std::string keep = c->err;
    for (int i = 0; i < c->n; ++i) { (void)hipSetDevice(c->ctx[i]->device); (void)hipStreamSynchronize(c->ctx[i]->stream); }
    c->err = keep;
    return local_fail;
  }
  // ---- results of the local restarts (synchronises each device), then the gathered vectors from local device 0
  for (int i = 0; i < c->n; ++i) {
    Local& L = loc[i];
    if (L.n == 0) continue;
    vbmc_ctx* ctx = c->ctx[i];
    COMM_HIP(c, hipSetDevice(ctx->device));
    vbmc_status st = elbo_read_results(ctx, L.plan, &L.sub);
    if (st) return comm_err(c, st, "device %d: %s", ctx->device, vbmc_last_error(ctx));
    const int g = c->rank0 + i;
    auto put = [&](double* dst, const std::vector<double>& v, size_t per) {
      if (!dst) return;
      for (int q = 0; q < L.n; ++q) memcpy(dst + (size_t)(g + (size_t)q * G) * per, &v[(size_t)q * per], per * sizeof(double));
    };
    put(a->G, L.G, 1); put(a->H, L.H, 1); put(a->varGss, L.varGss, 1);
    put(a->dF, L.dF, T); put(a->dG, L.dG, T); put(a->dH, L.dH, T);
    put(a->I_sk, L.I, (size_t)L.S * K); put(a->J_sjk, L.J, (size_t)L.S * K * K); put(a->G_s, L.Gs, L.S); put(a->varG_s, L.vGs, L.S);
  }
  std::vector<double> gathered(2 * (size_t)P * G);
  COMM_HIP(c, hipSetDevice(c->ctx[0]->device));
  COMM_HIP(c, hipMemcpyAsync(gathered.data(), c->d_recv[0], gathered.size() * sizeof(double), hipMemcpyDeviceToHost, c->ctx[0]->stream));
  for (int i = 0; i < c->n; ++i) {
    COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
    COMM_HIP(c, hipStreamSynchronize(c->ctx[i]->stream));
  }
  for (int r = 0; r < R; ++r) {
    const int g = r % G, q = r / G;
    if (a->F) a->F[r] = gathered[(size_t)g * 2 * P + q];
    if (a->varG) a->varG[r] = gathered[(size_t)g * 2 * P + P + q];
  }
  return VBMC_OK;
}


// ---- pipelined form: streams of INDEPENDENT batches dealt over the ranks (the sieve's candidates, misc/vpsieve_vbmc.m:74-78;
// bench.py's N > 1 step).  vbmc_elbo_multi_submit stages this process's restarts, enqueues the passes, the pick of (F, varG), ONE
// ncclAllGather and the copy of the gathered vectors into pinned memory -- and returns; vbmc_elbo_multi_collect waits for them.
// Two slots: the host stages batch i + 1 while the devices work on batch i.  Everything a call needs beyond the batch itself (the
// per-device argument structs, the staging vectors, the exchange blocks, the pinned landing block) lives in the communicator's slot
// and is sized once: a steady-state call allocates nothing.
static vbmc_status comm_slot_reserve(vbmc_comm* c, vbmc_comm::Slot& sl, size_t count) {
  if (sl.d_send.empty()) { sl.d_send.assign(c->n, nullptr); sl.d_recv.assign(c->n, nullptr); }
  if (count > sl.cap) {
    const size_t cap = std::max<size_t>(count + count / 2, 256);
    for (int i = 0; i < c->n; ++i) {
      COMM_HIP(c, hipSetDevice(c->ctx[i]->device));
      COMM_HIP(c, hipStreamSynchronize(c->ctx[i]->stream));
      if (i < (int)c->xs.size() && c->xs[i]) COMM_HIP(c, hipStreamSynchronize(c->xs[i]));
      if (sl.d_send[i]) COMM_HIP(c, hipFree(sl.d_send[i]));
      if (sl.d_recv[i]) COMM_HIP(c, hipFree(sl.d_recv[i]));
      sl.d_send[i] = sl.d_recv[i] = nullptr;
      COMM_HIP(c, hipMalloc((void**)&sl.d_send[i], cap * sizeof(double)));
      COMM_HIP(c, hipMalloc((void**)&sl.d_recv[i], cap * c->world * sizeof(double)));
    }
    sl.cap = cap;
  }
  const size_t hneed = count * c->world;
  if (hneed > sl.h_cap) {
    if (sl.h_gather) { COMM_HIP(c, hipHostFree(sl.h_gather)); sl.h_gather = nullptr; }
    COMM_HIP(c, hipHostMalloc((void**)&sl.h_gather, (hneed + hneed / 2) * sizeof(double), hipHostMallocDefault));
    sl.h_cap = hneed + hneed / 2;
  }
  return VBMC_OK;
}

static vbmc_status multi_submit_impl(vbmc_comm* c, const vbmc_gp* const* gps, const vbmc_elbo_args* a, int slot);
//  whatever way the enqueue fails after passes went onto the slot streams -- a HIP call, the all-gather's own enqueue, the end
// event -- nothing stays in flight behind a slot nobody will collect: the streams are drained and the slots given back before the error is
// returned.  (A rank whose own EVALUATION fails still enters the collective with an all-NaN block, below; a rank whose runtime fails
// cannot, and the others see the communicator's error.)
extern "C" vbmc_status vbmc_elbo_multi_submit(vbmc_comm* c, const vbmc_gp* const* gps, const vbmc_elbo_args* a, int slot) {
  const vbmc_status st = multi_submit_impl(c, gps, a, slot);
  if (st != VBMC_OK && c && slot >= 0 && slot < VBMC_SLOTS && !c->slot[slot].busy) {
    const std::string keep = c->err;
    for (int i = 0; i < c->n; ++i) {
      vbmc_ctx* ctx = c->ctx[i];
      (void)hipSetDevice(ctx->device);
      (void)hipStreamSynchronize(ctx->stream);
      if (i < (int)c->xs.size() && c->xs[i]) (void)hipStreamSynchronize(c->xs[i]);
      vbmc_ctx* w = ctx->slot_where[slot];
      if (w) { (void)hipStreamSynchronize(w->stream); w->slot_busy[ctx->slot_inner[slot]] = false; }
    }
    (void)hipGetLastError();
    c->err = keep;
  }
  return st;
}

static vbmc_status multi_submit_impl(vbmc_comm* c, const vbmc_gp* const* gps, const vbmc_elbo_args* a, int slot) {
  if (!c) return VBMC_ERR_INVALID;
  if (!gps || !a) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit: null surrogates / args");
  if (slot < 0 || slot >= VBMC_SLOTS) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit: slot must be 0 .. 3");
  if (a->struct_size != sizeof(vbmc_elbo_args)) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_args.struct_size (ABI mismatch)");
  if (a->eps_mode != 0) return comm_err(c, VBMC_ERR_UNSUPPORTED, "vbmc_elbo_multi_submit: device RNG (eps_mode 0) only");
  if (a->restart_offset != 0 || a->restart_stride > 1) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit deals the restarts itself");
  vbmc_comm::Slot& sl = c->slot[slot];
  if (sl.busy) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit: slot %d holds an uncollected batch", slot);
  const int G = c->world, R = a->R, K = a->K, D = a->D;
  if (R < 1 || K < 1 || D < 1) return comm_err(c, VBMC_ERR_INVALID, "D, K, R must be positive");
  int T = 0;
  { const int n[4] = {D * K, K, D, K}; for (int g = 0; g < 4; ++g) if (a->optimize[g]) T += n[g]; }
  if (T > 0 && !a->theta) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit: theta is null");
  const int P = (R + G - 1) / G;
  { vbmc_status s_ = comm_slot_reserve(c, sl, 2 * (size_t)P); if (s_) return s_; }
  sl.R = R; sl.T = T; sl.P = P;
  sl.n.assign(c->n, 0); sl.st.assign(c->n, VBMC_OK);
  if ((int)sl.theta.size() != c->n) { sl.theta.resize(c->n); sl.sub.resize(c->n); }
  if ((int)c->xs.size() != c->n) c->xs.assign(c->n, nullptr);
  sl.xst.assign(c->n, nullptr);
  vbmc_status local_fail = VBMC_OK;
  for (int i = 0; i < c->n; ++i) {
    const int g = c->rank0 + i;
    vbmc_ctx* ctx = c->ctx[i];
    COMM_HIP(c, hipSetDevice(ctx->device));
    int n = g < R ? (R - g + G - 1) / G : 0;
    vbmc_status st = VBMC_OK;
    if (n > 0 && !gps[i]) st = comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit: no surrogate for local device %d", i);
    if (n > 0 && !st) {
      vbmc_elbo_args& sub = sl.sub[i];
      sub = *a;
      sub.R = n;
      sub.restart_offset = g; sub.restart_stride = G;
      std::vector<double>& th = sl.theta[i];
      if (th.size() < (size_t)T * n) th.resize((size_t)T * n);
      for (int q = 0; q < n; ++q) memcpy(&th[(size_t)q * T], a->theta + (size_t)(g + (size_t)q * G) * T, T * sizeof(double));
      sub.theta = th.data();
      // the pass on the slot's own stream (abi_elbo.hip: slot_ctx), the pick after it on the same stream; the exchange on the
      // communicator's exchange stream, ordered after the pick by an event: the collectives of one communicator in issue order on one
      // stream, the passes of consecutive batches overlapping
      vbmc_ctx* sc = ctx;
      int inner = slot;
      if (slot_in_flight(ctx, slot)) st = set_err(ctx, VBMC_ERR_INVALID, "vbmc_elbo_multi_submit: slot %d holds an uncollected pass", slot);
      if (!st) st = slot_ctx(ctx, &sub, slot, &sc, &inner);
      if (!st) st = slot_err(ctx, sc, elbo_submit_core(sc, gps[i], &sub, inner, "vbmc_elbo_multi_submit"));
      if (!st) { ctx->slot_where[slot] = sc; ctx->slot_inner[slot] = inner; }
      if (st) comm_err(c, st, "device %d: %s", ctx->device, vbmc_last_error(ctx));
    }
    if (st) { local_fail = st; n = 0; }      // the rank still enters the collective: an all-NaN block (lock-step with the others)
    sl.n[i] = n; sl.st[i] = st;
    const size_t OS = OUT_HDR + 3 * (size_t)T;
    const double* dout = n > 0 ? (const double*)((const SlotPlan*)ctx->slot_where[slot]->slot_plan[ctx->slot_inner[slot]])->P.d_out : nullptr;
    // the pick runs on the stream the pass ran on (the result records are that stream's scratch: the next pass queued there overwrites
    // them)
    vbmc_ctx* ps = (n > 0 && ctx->slot_where[slot]) ? ctx->slot_where[slot] : ctx;
    // the communicator's own exchange stream (each exchange on the stream of its pass, and the exchange stream at high priority, were
    // measured equal in round 4)
    hipStream_t xst = ps->stream;
    if (ps != ctx) {
      if (!c->xs[i]) {     // beside both slot streams (abi_elbo.hip: stream_beside); they exist and, for the first batch, are idle but for this pass
        hipStream_t both[2];
        int nb = 0;
        for (vbmc_ctx* sub : ctx->slot_sub) if (sub) both[nb++] = sub->stream;
        c->xs[i] = stream_beside(ctx, both, nb, 0);     // (none to be had: the exchange stays on the pass's stream)
      }
      if (c->xs[i]) xst = c->xs[i];
    }
    if (n == 0 && c->xs[i]) xst = c->xs[i];     // a device without restarts: where this communicator's exchanges have been running
    sl.xst[i] = xst;
    hipLaunchKernelGGL(k_comm_pick, dim3((P + 63) / 64), dim3(64), 0, n > 0 ? ps->stream : xst, n, P, n > 0 ? OS : (size_t)1, dout, sl.d_send[i]);
    COMM_HIP(c, hipGetLastError());
    if (n > 0 && xst != ps->stream) {
      if (!ctx->slot_yev[slot]) COMM_HIP(c, hipEventCreateWithFlags(&ctx->slot_yev[slot], hipEventDisableTiming));
      COMM_HIP(c, hipEventRecord(ctx->slot_yev[slot], ps->stream));
      COMM_HIP(c, hipStreamWaitEvent(xst, ctx->slot_yev[slot], 0));
    }
  }
  const std::string keep = c->err;
  { vbmc_status s_ = comm_allgather_enqueue(c, sl.d_send.data(), sl.d_recv.data(), 2 * (size_t)P, sl.xst.data()); if (s_) return s_; }
  COMM_HIP(c, hipSetDevice(c->ctx[0]->device));
  COMM_HIP(c, hipMemcpyAsync(sl.h_gather, sl.d_recv[0], 2 * (size_t)P * G * sizeof(double), hipMemcpyDeviceToHost, sl.xst[0]));
  for (int i = 0; i < c->n; ++i) {
    vbmc_ctx* ctx = c->ctx[i];
    COMM_HIP(c, hipSetDevice(ctx->device));
    if (sl.n[i] > 0) {
      vbmc_ctx* sc = ctx->slot_where[slot];
      vbmc_status s_ = elbo_submit_mark(sc, ctx->slot_inner[slot], "vbmc_elbo_multi_submit");
      if (s_) return comm_err(c, s_, "device %d: %s", ctx->device, vbmc_last_error(sc));
      if (sl.xst[i] != sc->stream) {     // the exchange (and, on local device 0, the copy of the gathered vectors) ends on the exchange stream
        if (!ctx->slot_zev[slot]) COMM_HIP(c, hipEventCreateWithFlags(&ctx->slot_zev[slot], hipEventDisableTiming));
        COMM_HIP(c, hipEventRecord(ctx->slot_zev[slot], sl.xst[i]));
      }
    }
  }
  if (local_fail) {     // the exchange is enqueued (the other ranks are not left waiting); drain and report
    for (int i = 0; i < c->n; ++i) {
      (void)hipSetDevice(c->ctx[i]->device);
      (void)hipStreamSynchronize(c->ctx[i]->stream);
      if (sl.xst[i]) (void)hipStreamSynchronize(sl.xst[i]);
      if (c->ctx[i]->slot_where[slot]) {
        (void)hipStreamSynchronize(c->ctx[i]->slot_where[slot]->stream);
        c->ctx[i]->slot_where[slot]->slot_busy[c->ctx[i]->slot_inner[slot]] = false;
      }
    }
    c->err = keep;
    return local_fail;
  }
  sl.busy = true;
  return VBMC_OK;
}

extern "C" vbmc_status vbmc_elbo_multi_collect(vbmc_comm* c, const vbmc_elbo_args* a, int slot) {
  if (!c) return VBMC_ERR_INVALID;
  if (!a || slot < 0 || slot >= VBMC_SLOTS) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_collect: null args / slot not 0 .. 3");
  vbmc_comm::Slot& sl = c->slot[slot];
  if (!sl.busy) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_collect: nothing submitted in slot %d", slot);
  int T = 0;
  { const int n[4] = {a->D * a->K, a->K, a->D, a->K}; for (int g = 0; g < 4; ++g) if (a->optimize[g]) T += n[g]; }
  if (a->R != sl.R || T != sl.T) return comm_err(c, VBMC_ERR_INVALID, "vbmc_elbo_multi_collect: args differ from the submitted ones (R, optimize flags)");
  sl.busy = false;
  const int G = c->world, R = sl.R, P = sl.P;
  vbmc_status fail = VBMC_OK;
  for (int i = 0; i < c->n; ++i) {
    vbmc_ctx* ctx = c->ctx[i];
    COMM_HIP(c, hipSetDevice(ctx->device));
    if (sl.n[i] > 0) {
      const SlotPlan* sp = nullptr;
      vbmc_ctx* sc = ctx->slot_where[slot] ? ctx->slot_where[slot] : ctx;
      vbmc_status s_ = elbo_collect_core(sc, &sl.sub[i], ctx->slot_where[slot] ? ctx->slot_inner[slot] : slot, &sp, "vbmc_elbo_multi_collect");
      if (s_) { fail = comm_err(c, s_, "device %d: %s", ctx->device, vbmc_last_error(sc)); continue; }
      if (sl.xst[i] != sc->stream) {
        hipError_t e_ = hipEventSynchronize(ctx->slot_zev[slot]);
        if (e_ != hipSuccess) { (void)hipGetLastError(); fail = comm_err(c, VBMC_ERR_HIP, "device %d: %s", ctx->device, hipGetErrorString(e_)); continue; }
      }
      vbmc_elbo_args view = *a;            // the caller's arrays; F and varG come from the gathered vectors below
      view.F = nullptr; view.varG = nullptr;
      elbo_unpack(sp->P, &view, sp->hout, c->rank0 + i, G);
    } else {
      COMM_HIP(c, hipStreamSynchronize(sl.xst[i]));     // a device without restarts still took part in the exchange
    }
  }
  if (c->n > 0 && sl.n[0] == 0) COMM_HIP(c, hipStreamSynchronize(sl.xst[0]));
  if (fail) return fail;
  for (int r = 0; r < R; ++r) {
    const int g = r % G, q = r / G;
    if (a->F) a->F[r] = sl.h_gather[(size_t)g * 2 * P + q];
    if (a->varG) a->varG[r] = sl.h_gather[(size_t)g * 2 * P + P + q];
  }
  return VBMC_OK;
}
