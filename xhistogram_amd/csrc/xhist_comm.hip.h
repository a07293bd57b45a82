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
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
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
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))sym("ncclCommInitRank");
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
};

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
  ncclResult_t r = api->comm_init_rank(&c->comm, world_size, u, rank);  // collective: returns once every rank has joined
  if (r != ncclSuccess) {
    delete c;
    return fail(XHIST_ERR_COMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world_size, device, api->error_string(r));
  }
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
  if (!comm || !comm->comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
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
  DeviceGuard g;
  if (int rc = g.set(comm->device)) return rc;
  RCCLC(api, api->all_reduce(buf, buf, (size_t)count, dt, rop, comm->comm, static_cast<hipStream_t>(stream)));
  return XHIST_OK;
}

extern "C" int xhist_comm_allgather(xhist_comm* comm, const void* send, void* recv, int64_t count, int dtype, void* stream) {
  Range range_("xhist_comm_allgather[rows]");
  if (!comm || !comm->comm) return fail(XHIST_ERR_INVALID, "comm is NULL");
  if (count < 0 || (count > 0 && (!send || !recv))) return fail(XHIST_ERR_INVALID, "buffer is NULL / count < 0");
  ncclDataType_t dt;
  if (int rc = nccl_dtype(dtype, &dt)) return rc;
  if (count == 0) return XHIST_OK;
  const RcclApi* api;
  if (int rc = rccl_api(&api)) return rc;
  DeviceGuard g;
  if (int rc = g.set(comm->device)) return rc;
  RCCLC(api, api->all_gather(send, recv, (size_t)count, dt, comm->comm, static_cast<hipStream_t>(stream)));
  return XHIST_OK;
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
