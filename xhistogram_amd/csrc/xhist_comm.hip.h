// xhist_comm.hip.h — host side: the exchange step of sharded inputs (RCCL over xGMI) behind the C ABI
// Part of the single translation unit xhist_capi.hip (included there, in order).
//
// One process per GPU.  The only data that ever crosses GPUs on this path are the partial
// histograms (800 B … 8 MiB) and two doubles for a global min/max, so the exchange is a single
// in-place all-reduce (shards cut along a reduced axis: the reference's `bin_counts.sum(drop_axes)`,
// core.py:439) or an all-gather of rows (shards cut along a kept axis).  RCCL is opened with
// dlopen on the first call — a process that never exchanges anything (single GPU, or a host that
// brings its own collective, like the Python package with torch.distributed) never loads it, and
// one that already has an RCCL mapped (torch's) shares that copy instead of loading a second.
//
// Deadlines (round 4).  A peer that never joins, or dies inside a collective, must not hang a long-lived worker: the header
// promises status codes.  What can wait for a peer:
//   the rendezvous (xhist_comm_create) — ncclCommInitRankConfig runs on a helper thread while the caller waits for it with
//     a deadline; on expiry the caller calls ncclCommAbort on the half-built communicator (RCCL publishes the handle before
//     it starts to wait for its peers), which makes the rendezvous return with an error within a second.  RCCL's own
//     non-blocking mode does not do this job here: with config.blocking = 0 the init call of RCCL 2.27.7 (ROCm 7.2) itself
//     never returns while a peer is missing (tools/ubench/rccl_lonely.cpp, profiles/r04_c_rccl_lonely_rank.txt);
//   the completion of a collective (xhist_comm_wait) — polls the stream and ncclCommGetAsyncError against the deadline.
// The deadline is $XHIST_AMD_COMM_TIMEOUT_S seconds (default 300; <= 0 = wait for ever; XHIST_AMD_COMM_CREATE_TIMEOUT_S overrides it for the rendezvous alone).  On expiry, or on an asynchronous
// RCCL error, the communicator is torn down with ncclCommAbort (kernels of a collective in flight return), the call reports
// XHIST_ERR_COMM with what it was waiting for, and every later call on that communicator fails at once with the same code.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRankConfig) comm_init_rank_config = nullptr;
  decltype(&ncclCommGetAsyncError) get_async_error = nullptr;
  decltype(&ncclCommAbort) comm_abort = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclGetVersion) get_version = nullptr;
  std::string path;
};

static std::mutex g_rccl_mu;
static RcclApi g_rccl;
static bool g_rccl_tried = false;

static int rccl_api(const RcclApi** out) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.handle) {
    if (g_rccl_tried) return fail(XHIST_ERR_COMM, "RCCL could not be loaded earlier in this process (set XHIST_AMD_RCCL=/path/to/librccl.so)");
    g_rccl_tried = true;
    const char* env = getenv("XHIST_AMD_RCCL");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string tried;
    // first an RCCL this process has mapped already (same soname), then a fresh one, private to us
    for (int pass = 0; pass < 2 && !g_rccl.handle; ++pass)
      for (const char* nm : names) {
        if (!nm || !*nm) continue;
        void* h = dlopen(nm, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_LOCAL));
        if (h) { g_rccl.handle = h; g_rccl.path = nm; break; }
        if (pass == 1) { const char* e = dlerror(); tried += std::string(" [") + nm + ": " + (e ? e : "?") + "]"; }
      }
    if (!g_rccl.handle) return fail(XHIST_ERR_COMM, "librccl not found:%s", tried.c_str());
    bool ok = true;
    auto sym = [&](const char* name) { void* s = dlsym(g_rccl.handle, name); if (!s) ok = false; return s; };
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))sym("ncclGetUniqueId");
    g_rccl.comm_init_rank_config = (decltype(g_rccl.comm_init_rank_config))sym("ncclCommInitRankConfig");
    g_rccl.get_async_error = (decltype(g_rccl.get_async_error))sym("ncclCommGetAsyncError");
    g_rccl.comm_abort = (decltype(g_rccl.comm_abort))sym("ncclCommAbort");
    g_rccl.all_reduce = (decltype(g_rccl.all_reduce))sym("ncclAllReduce");
    g_rccl.all_gather = (decltype(g_rccl.all_gather))sym("ncclAllGather");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))sym("ncclCommDestroy");
    g_rccl.error_string = (decltype(g_rccl.error_string))sym("ncclGetErrorString");
    g_rccl.get_version = (decltype(g_rccl.get_version))sym("ncclGetVersion");
    if (!ok) {
      dlclose(g_rccl.handle);
      g_rccl.handle = nullptr;
      return fail(XHIST_ERR_COMM, "%s lacks an nccl* entry point this library needs", g_rccl.path.c_str());
    }
  }
  *out = &g_rccl;
  return XHIST_OK;
}

#define RCCLC(api, expr)                                                                          \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess) return fail(XHIST_ERR_COMM, "%s failed: %s", #expr, (api)->error_string(r_)); \
  } while (0)

struct xhist_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
  bool aborted = false;     // torn down after a deadline or an asynchronous error: every later call reports XHIST_ERR_COMM
  std::string why;          // what the communicator was waiting for when it was aborted
  std::mutex mu;            // one call at a time per communicator (RCCL's own rule for one communicator)
};

// Default 300 s (torch.distributed waits 600): long-lived workers whose peers start tens of seconds apart must not
// lose their communicator to the deadline; a host that wants a failure sooner sets the variable.
// XHIST_AMD_COMM_CREATE_TIMEOUT_S, when set, applies to the rendezvous of xhist_comm_create alone.
static double comm_timeout_s(bool create = false) {
  const char* e = create ? getenv("XHIST_AMD_COMM_CREATE_TIMEOUT_S") : nullptr;  // read at every call: a test, or a host that knows better, may change it
  if (!e || !*e) e = getenv("XHIST_AMD_COMM_TIMEOUT_S");
  if (e && *e) return atof(e);
  return 300.0;
}

typedef std::chrono::steady_clock comm_clock;

static int comm_dead(const xhist_comm* c) {
  return fail(XHIST_ERR_COMM, "communicator (rank %d of %d, device %d) was aborted earlier: %s", c->rank, c->world, c->device, c->why.c_str());
}

// tear the communicator down and report; `what` says what was being waited for
static int comm_abort_with(const RcclApi* api, xhist_comm* c, const char* what, const char* detail) {
  if (c->comm) (void)api->comm_abort(c->comm);  // frees the communicator; kernels of a collective in flight return
  c->comm = nullptr;
  c->aborted = true;
  char buf[512];
  snprintf(buf, sizeof buf, "%s (%s)", what, detail);
  c->why = buf;
  return fail(XHIST_ERR_COMM, "RCCL rank %d of %d on device %d: %s — %s; the communicator was aborted (XHIST_AMD_COMM_TIMEOUT_S = %g s)", c->rank,
              c->world, c->device, what, detail, comm_timeout_s());
}

// Poll the communicator until RCCL reports it settled (ncclSuccess), failed, or the deadline passes.
// A non-blocking communicator answers ncclInProgress while a rendezvous or a connection setup is still going on.
static int comm_settle(const RcclApi* api, xhist_comm* c, const char* what, comm_clock::time_point t0) {
  const double limit = comm_timeout_s();
  for (int spin = 0;; ++spin) {
    ncclResult_t state = ncclSuccess;
    const ncclResult_t r = api->get_async_error(c->comm, &state);
    if (r != ncclSuccess) return comm_abort_with(api, c, what, api->error_string(r));
    if (state == ncclSuccess) return XHIST_OK;
    if (state != ncclInProgress) return comm_abort_with(api, c, what, api->error_string(state));
    if (limit > 0 && std::chrono::duration<double>(comm_clock::now() - t0).count() > limit)
      return comm_abort_with(api, c, what, "deadline passed while RCCL was still waiting for a peer");
    if (spin < 64) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spin < 1024 ? 50 : 1000));
  }
}

static int nccl_dtype(int dtype, ncclDataType_t* out) {
  switch (dtype) {
    case XHIST_I64: *out = ncclInt64; return XHIST_OK;    // counts: exact and order-independent
    case XHIST_F64: *out = ncclFloat64; return XHIST_OK;  // weighted sums, min/max of the data
    case XHIST_F32: *out = ncclFloat32; return XHIST_OK;
    default: return fail(XHIST_ERR_INVALID, "exchange of dtype tag %d: histograms are int64, float64 or float32", dtype);
  }
}

extern "C" int xhist_comm_unique_id(void* id, size_t cap) {
  if (!id || cap < XHIST_COMM_ID_BYTES) return fail(XHIST_ERR_INVALID, "id buffer must hold XHIST_COMM_ID_BYTES (%d) bytes", XHIST_COMM_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == XHIST_COMM_ID_BYTES, "XHIST_COMM_ID_BYTES must be RCCL's id size");
  if (n_devices() <= 0) return fail(XHIST_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
  const RcclApi* api;
  if (int rc = rccl_api(&api)) return rc;
  ncclUniqueId u;
  RCCLC(api, api->get_unique_id(&u));
  memcpy(id, &u, sizeof u);
  return XHIST_OK;
}

extern "C" int xhist_comm_create(int device, int rank, int world_size, const void* id, size_t id_bytes, xhist_comm** out) {
  Range range_("xhist_comm_create[RCCL init]");
  if (!id || !out) return fail(XHIST_ERR_INVALID, "id / out is NULL");
  if (id_bytes != XHIST_COMM_ID_BYTES) return fail(XHIST_ERR_INVALID, "id must be the XHIST_COMM_ID_BYTES bytes rank 0 got from xhist_comm_unique_id");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(XHIST_ERR_INVALID, "rank %d of %d", rank, world_size);
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available; this library has no CPU path", device);
  const RcclApi* api;
  if (int rc = rccl_api(&api)) return rc;
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  xhist_comm* c = new (std::nothrow) xhist_comm;
  if (!c) return fail(XHIST_ERR_NOMEM, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world_size;
  // the rendezvous on a helper thread, watched from here (see the header comment)
  struct InitJob {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t result = ncclSuccess;
    ncclComm_t comm = nullptr;  // written by RCCL (on the helper thread) as soon as the communicator object exists
  };
  auto job = std::make_shared<InitJob>();
  const int phys = physical_device(device);
  const auto init_fn = api->comm_init_rank_config;
  std::thread([job, init_fn, u, world_size, rank, phys] {
    ncclResult_t r = ncclInternalError;
    if (hipSetDevice(phys) == hipSuccess) {
      ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
      cfg.blocking = 1;
      r = init_fn(&job->comm, world_size, u, rank, &cfg);
    }
    std::lock_guard<std::mutex> lk(job->mu);
    job->result = r;
    job->done = true;
    job->cv.notify_all();
  }).detach();
  const double limit = comm_timeout_s(true);
  bool expired = false;
  {
    std::unique_lock<std::mutex> lk(job->mu);
    if (limit > 0) expired = !job->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return job->done; });
    else job->cv.wait(lk, [&] { return job->done; });
  }
  if (expired) {
    // nobody else joined in time: abort the half-built communicator so that the helper thread's rendezvous returns
    ncclComm_t half = *(ncclComm_t volatile*)&job->comm;
    if (half) (void)api->comm_abort(half);
    bool back = false;
    {
      std::unique_lock<std::mutex> lk(job->mu);
      back = job->cv.wait_for(lk, std::chrono::seconds(10), [&] { return job->done; });
    }
    delete c;
    return fail(XHIST_ERR_COMM,
                "RCCL rank %d of %d on device %d: rendezvous of %d ranks (xhist_comm_create) — deadline passed while RCCL was still waiting for a "
                "peer; the communicator was aborted%s (XHIST_AMD_COMM_TIMEOUT_S = %g s)",
                rank, world_size, device, world_size, back ? "" : " (its helper thread is still inside RCCL and was left behind)", limit);
  }
  if (job->result != ncclSuccess) {
    delete c;
    return fail(XHIST_ERR_COMM, "ncclCommInitRankConfig(rank %d of %d, device %d) failed: %s", rank, world_size, device, api->error_string(job->result));
  }
  c->comm = job->comm;
  *out = c;
  return XHIST_OK;
}

extern "C" int xhist_comm_info(const xhist_comm* comm, int* rank, int* world_size, int* device, int* rccl_version) {
  if (!comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  if (rank) *rank = comm->rank;
  if (world_size) *world_size = comm->world;
  if (device) *device = comm->device;
  if (rccl_version) {
    const RcclApi* api;
    if (int rc = rccl_api(&api)) return rc;
    RCCLC(api, api->get_version(rccl_version));
  }
  return XHIST_OK;
}

extern "C" int xhist_comm_allreduce(xhist_comm* comm, void* buf, int64_t count, int dtype, int op, void* stream) {
  Range range_("xhist_comm_allreduce[partials]");
  if (!comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  if (comm->aborted) return comm_dead(comm);
  if (!comm->comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  if (count < 0 || (count > 0 && !buf)) return fail(XHIST_ERR_INVALID, "buffer is NULL / count < 0");
  ncclDataType_t dt;
  if (int rc = nccl_dtype(dtype, &dt)) return rc;
  ncclRedOp_t rop;
  switch (op) {
    case XHIST_REDUCE_SUM: rop = ncclSum; break;
    case XHIST_REDUCE_MIN: rop = ncclMin; break;
    case XHIST_REDUCE_MAX: rop = ncclMax; break;
    default: return fail(XHIST_ERR_INVALID, "unknown reduction %d", op);
  }
  if (count == 0) return XHIST_OK;  // every rank sees the same count: nobody enters the collective
  const RcclApi* api;
  if (int rc = rccl_api(&api)) return rc;
  std::lock_guard<std::mutex> lk(comm->mu);
  if (comm->aborted) return comm_dead(comm);
  DeviceGuard g;
  if (int rc = g.set(comm->device)) return rc;
  const auto t0 = comm_clock::now();
  const ncclResult_t r = api->all_reduce(buf, buf, (size_t)count, dt, rop, comm->comm, static_cast<hipStream_t>(stream));
  if (r != ncclSuccess && r != ncclInProgress) return comm_abort_with(api, comm, "ncclAllReduce", api->error_string(r));
  return comm_settle(api, comm, "enqueue of an all-reduce (connection setup)", t0);  // (the first poll answers in steady state)
}

extern "C" int xhist_comm_allgather(xhist_comm* comm, const void* send, void* recv, int64_t count, int dtype, void* stream) {
  Range range_("xhist_comm_allgather[rows]");
  if (!comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  if (comm->aborted) return comm_dead(comm);
  if (!comm->comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  if (count < 0 || (count > 0 && (!send || !recv))) return fail(XHIST_ERR_INVALID, "buffer is NULL / count < 0");
  ncclDataType_t dt;
  if (int rc = nccl_dtype(dtype, &dt)) return rc;
  if (count == 0) return XHIST_OK;
  const RcclApi* api;
  if (int rc = rccl_api(&api)) return rc;
  std::lock_guard<std::mutex> lk(comm->mu);
  if (comm->aborted) return comm_dead(comm);
  DeviceGuard g;
  if (int rc = g.set(comm->device)) return rc;
  const auto t0 = comm_clock::now();
  const ncclResult_t r = api->all_gather(send, recv, (size_t)count, dt, comm->comm, static_cast<hipStream_t>(stream));
  if (r != ncclSuccess && r != ncclInProgress) return comm_abort_with(api, comm, "ncclAllGather", api->error_string(r));
  return comm_settle(api, comm, "enqueue of an all-gather (connection setup)", t0);
}

// Wait until everything enqueued on `stream` — the collectives above — has completed, watching the communicator: a peer
// that died inside a collective shows either as an asynchronous RCCL error or as a stream that never drains.  Returns
// XHIST_OK, or XHIST_ERR_COMM after aborting the communicator (deadline: XHIST_AMD_COMM_TIMEOUT_S).
extern "C" int xhist_comm_wait(xhist_comm* comm, void* stream) {
  Range range_("xhist_comm_wait[collective completion]");
  if (!comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  const RcclApi* api;
  if (int rc = rccl_api(&api)) return rc;
  std::lock_guard<std::mutex> lk(comm->mu);
  if (comm->aborted) return comm_dead(comm);
  if (!comm->comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  DeviceGuard g;
  if (int rc = g.set(comm->device)) return rc;
  const double limit = comm_timeout_s();
  const auto t0 = comm_clock::now();
  for (int spin = 0;; ++spin) {
    const hipError_t q = hipStreamQuery(static_cast<hipStream_t>(stream));
    if (q == hipSuccess) return XHIST_OK;
    if (q != hipErrorNotReady) {
      (void)hipGetLastError();
      return comm_abort_with(api, comm, "completion of a collective (xhist_comm_wait)", hipGetErrorString(q));
    }
    (void)hipGetLastError();
    ncclResult_t state = ncclSuccess;
    const ncclResult_t r = api->get_async_error(comm->comm, &state);
    if (r != ncclSuccess) return comm_abort_with(api, comm, "completion of a collective (xhist_comm_wait)", api->error_string(r));
    if (state != ncclSuccess && state != ncclInProgress)
      return comm_abort_with(api, comm, "completion of a collective (xhist_comm_wait)", api->error_string(state));
    if (limit > 0 && std::chrono::duration<double>(comm_clock::now() - t0).count() > limit)
      return comm_abort_with(api, comm, "completion of a collective (xhist_comm_wait)", "deadline passed with the collective still in flight: a peer never entered it, or died in it");
    if (spin < 256) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spin < 4096 ? 20 : 500));
  }
}

extern "C" int xhist_comm_destroy(xhist_comm* comm) {
  if (!comm) return XHIST_OK;
  int rc = XHIST_OK;
  if (comm->comm) {
    const RcclApi* api;
    rc = rccl_api(&api);
    if (rc == XHIST_OK) {
      DeviceGuard g;
      rc = g.set(comm->device);
      if (rc == XHIST_OK) {
        ncclResult_t r = api->comm_destroy(comm->comm);
        if (r != ncclSuccess) rc = fail(XHIST_ERR_COMM, "ncclCommDestroy failed: %s", api->error_string(r));
      }
    }
  }
  delete comm;
  return rc;
}
