// xhist_host_common.hip.h — host side: error reporting, device guard, dtype helpers, the plan object, device queries
// Part of the single translation unit xhist_capi.hip (included there, in order).
#pragma once

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string tl_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  tl_err = buf;
  return code;
}

#define HIPC(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) return fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct DeviceGuard {  // set the plan's device for this call, restore the caller's on exit
  int prev = -1;
  bool changed = false;
  int set(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) return fail(XHIST_ERR_NO_DEVICE, "no HIP device is usable in this process");
    if (prev != dev) {
      if (hipSetDevice(dev) != hipSuccess) return fail(XHIST_ERR_NO_DEVICE, "hipSetDevice(%d) failed", dev);
      changed = true;
    }
    return XHIST_OK;
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

// ------------------------------------------------------------------------------------------
// roctx ranges (plan build / execute / exchange show up as named ranges in rocprofv3
// --marker-trace timelines).  The marker library is not linked: a copy the process has mapped already
// (the profiler preloads it) is used, or one is loaded when XHIST_AMD_ROCTX=1; otherwise ranges cost
// one predictable branch.
// ------------------------------------------------------------------------------------------
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    const char* want = getenv("XHIST_AMD_ROCTX");
    void* h = nullptr;
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h && want && want[0] == '1')
      for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
static const Roctx& roctx() {
  static Roctx r;
  return r;
}
struct Range {  // RAII: a named range on the calling thread
  bool on;
  explicit Range(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~Range() {
    if (on) roctx().pop();
  }
};

static int n_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// Scratch (staging of host inputs, the partitioned mode's record streams, transposes) comes from the device's
// stream-ordered pool, so concurrent callers never share it.  By default that pool gives everything back to
// the driver at the next synchronisation and the following call pays for fresh allocations: 100 us of a
// 465 us numpy call of 10^6 samples — and 1.1 to 1.6 SECONDS per call for the 56 GB of record streams of
// C5 at 4*10^9 samples, against 40 ms of kernels (profiles/r02_c_alloc_probe.jsonl).  So the pool keeps what
// it was given, up to half of the device's memory ($XHIST_AMD_POOL_KEEP_GB overrides; xhist_shutdown trims).
static std::mutex g_pool_mu;
static bool g_pool_done[64] = {false};

static void keep_pool_warm(int device) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (device < 0 || device >= 64 || g_pool_done[device]) return;
  g_pool_done[device] = true;
  hipMemPool_t pool;
  if (hipDeviceGetDefaultMemPool(&pool, device) != hipSuccess) return;
  size_t free_b = 0, total_b = 0;
  uint64_t want = (uint64_t)2 << 30;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) want = std::max<uint64_t>(want, (uint64_t)total_b / 2);
  if (const char* env = getenv("XHIST_AMD_POOL_KEEP_GB")) {
    const double gb = atof(env);
    if (gb >= 0) want = (uint64_t)(gb * 1073741824.0);
  }
  uint64_t cur = 0;
  if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &cur) == hipSuccess && cur >= want) return;
  (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &want);
}

static void trim_pools() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) return;
  for (int d = 0; d < 64; ++d) {
    if (!g_pool_done[d]) continue;
    hipMemPool_t pool;
    if (hipSetDevice(d) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, d) == hipSuccess) {
      (void)hipDeviceSynchronize();
      (void)hipMemPoolTrimTo(pool, 0);
    }
  }
  (void)hipSetDevice(prev);
}

static int dtype_size(int dt) {
  switch (dt) {
    case XHIST_F64: case XHIST_I64: case XHIST_U64: return 8;
    case XHIST_F32: case XHIST_I32: case XHIST_U32: return 4;
    case XHIST_F16: case XHIST_I16: case XHIST_U16: return 2;
    case XHIST_I8: case XHIST_U8: case XHIST_BOOL: return 1;
    default: return 0;
  }
}

static bool dtype_is_int(int dt) { return dt >= XHIST_I64 && dt <= XHIST_BOOL; }

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct TableSet {
  DimTable dim[kMaxDims];
  uint64_t* blob = nullptr;  // device: [edges + 4 sentinels per dimension][bucket tables]
  int32_t words = 0;         // blob size in 8-byte words
  int max_cnt = 0;           // most edges sharing one bucket
};

struct xhist_plan {
  int device = 0;
  int n_dims = 0;
  int cmp = 0;
  // table sets: [compare domain: 0 native (float64 / int64), 1 float32 thresholds (float64 plans)]
  //             [0: (start | cnt << 16) uint32 buckets, 1: uint16 start-only buckets on a 2x finer
  //                 grid for the linear-scan kernels (float domains only)]
  TableSet ts[2][2];
  bool uns = false;    // the int64-domain inputs hold unsigned 64-bit values (XHIST_CMP_UNSIGNED)
  bool huge = false;   // some dimension has more than 65535 edges: no bucket tables (lut_k = 0)
  bool arith = false;  // every dimension has arithmetic (numpy.linspace) edges: table-free digitize available
  int64_t n_bins = 0;
  int cus = 256;
  size_t lds_max = 64 * 1024;
  // tuning / diagnostics
  int block_threads = 0;
  int grid_blocks = 0;
  int force_global = 0;
  int force_generic = 0;
  int partition = 0;  // 0 auto, 1 prefer the partitioned mode whenever it is legal, -1 never
  int fused_pref = 0;  // partitioned mode: 0 one routing pass where it applies, -1 always count + prefix + scatter
  int slices_pref = 0;  // 0 auto, 1 prefer bin slices for histograms beyond LDS, -1 never
  int arith_pref = 0;  // 0 auto, 1 table-free digitize whenever the edges are arithmetic, -1 never
  int lanes = 0;      // 0 auto, 1 prefer the row-per-lane kernels whenever they are legal, -1 never
  int lds_copies = 0;
  int profile = 0;
  int profile_stride = 1;  // record the event pair of every stride-th execute only (microsecond kernels: two event records cost as much as the launch)
  int64_t n_seen = 0;
  std::mutex mu;  // guards events + desc
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ring;  // profile > 0: event pairs around the main kernel
  int64_t n_recorded = 0;                                // executes recorded since the last read
  std::string desc;
};

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

extern "C" int xhist_abi_version(void) { return XHIST_ABI_VERSION; }

extern "C" const char* xhist_last_error(void) { return tl_err.c_str(); }

extern "C" int xhist_device_count(int* count) {
  if (!count) return fail(XHIST_ERR_INVALID, "count is NULL");
  *count = n_devices();
  return XHIST_OK;
}

extern "C" int xhist_device_info(int device, char* name, size_t name_cap, int* compute_units, size_t* total_mem_bytes) {
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "device %d not available", device);
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, device));
  if (name && name_cap) {
    strncpy(name, prop.gcnArchName, name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
  return XHIST_OK;
}
