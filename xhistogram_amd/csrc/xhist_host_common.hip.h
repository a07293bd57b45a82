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

// Census of the dispatch surface (development): with XHIST_AMD_KERNEL_LOG=<file> the name of every distinct kernel
// instantiation this process launches through a dispatch table is appended to <file> once (tools/kernel_census.py compares
// the union over the tests, the cliff scanners and the soak with what the library holds; VERDICT r4 "next" #7).
static void log_picked_kernel(const void* fn) {
  static const char* const path = getenv("XHIST_AMD_KERNEL_LOG");
  if (!path || !*path) return;
  static const bool every_pick = [] { const char* e = getenv("XHIST_AMD_KERNEL_LOG_ALL"); return e && *e == '1'; }();  // (tests/test_gpu_census.py's discovery mode: which CASE picks what)
  static std::mutex mu;
  static std::set<const void*> seen;
  std::lock_guard<std::mutex> lk(mu);
  if (!seen.insert(fn).second && !every_pick) return;
  // the host stub's own symbol (`__device_stub__<kernel>`), from the dynamic symbol table: no call into the HIP runtime
  // (hipKernelNameRefByPtr hung a process that had loaded this library before torch's bundled runtime)
  Dl_info info;
  const char* name = dladdr(fn, &info) && info.dli_sname ? info.dli_sname : "?";
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "%s\n", name);
    fclose(f);
  }
}
#define XH_LAUNCH_PICKED(fn, ...)                                \
  do {                                                           \
    log_picked_kernel(reinterpret_cast<const void*>(fn));        \
    hipLaunchKernelGGL(fn, __VA_ARGS__);                         \
  } while (0)

// Logical -> physical devices (tests only): XHIST_AMD_DEVICE_ALIAS="0,0" makes the library show TWO devices that are both
// HIP device 0, so that everything keyed by device — plan caches, per-GPU host threads, per-device streams and buffers, the
// block -> GPU assignment — runs its N > 1 code with real kernels on a box with one GPU (VERDICT r2 "next" #2b).  Every
// `device` argument of the C ABI is a LOGICAL index; only the calls into HIP translate.  RCCL refuses two ranks on one
// GPU, so the exchange itself stays out of reach of this trick.  Unset (production): the identity.
static const std::vector<int>& device_alias() {
  static const std::vector<int> v = [] {
    std::vector<int> a;
    const char* e = getenv("XHIST_AMD_DEVICE_ALIAS");
    if (!e || !*e) return a;
    int real = 0;
    if (hipGetDeviceCount(&real) != hipSuccess) { (void)hipGetLastError(); return a; }
    for (const char* q = e; *q;) {
      char* end = nullptr;
      const long d = strtol(q, &end, 10);
      if (end == q || d < 0 || d >= real) { a.clear(); return a; }  // malformed / names a GPU that is not there: ignored
      a.push_back((int)d);
      q = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') { a.clear(); return a; }
    }
    return a;
  }();
  return v;
}
static int physical_device(int dev) {
  const std::vector<int>& a = device_alias();
  return (!a.empty() && dev >= 0 && dev < (int)a.size()) ? a[(size_t)dev] : dev;
}
static int logical_device(int phys) {  // the first logical device on that GPU
  const std::vector<int>& a = device_alias();
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i] == phys) return (int)i;
  return phys;
}

struct DeviceGuard {  // set the plan's device for this call, restore the caller's on exit
  int prev = -1;
  bool changed = false;
  int set(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) return fail(XHIST_ERR_NO_DEVICE, "no HIP device is usable in this process");
    const int phys = physical_device(dev);
    if (prev != phys) {
      if (hipSetDevice(phys) != hipSuccess) return fail(XHIST_ERR_NO_DEVICE, "hipSetDevice(%d) failed", phys);
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
  return device_alias().empty() ? n : (int)device_alias().size();
}

// ------------------------------------------------------------------------------------------
// Scratch memory (staging of host inputs, the partitioned mode's record streams, transposes, gathers): a caching
// allocator of the library's own on top of hipMalloc, with stream-ordered reuse.
//
// Why not hipMallocAsync: (1) the device's stream-ordered pool returns everything to the driver at the next
// synchronisation unless its release threshold is raised — 100 us of a 465 us numpy call of 10^6 samples, and 1.1-1.6 s
// per call for the 56 GB of record streams of C5 at 4*10^9 samples against 40 ms of kernels
// (profiles/r02_c_alloc_probe.jsonl); (2) with ROCm 7.0 it is not safe next to plain hipMalloc / hipFree calls of OTHER
// threads: host-route calls (hipMallocAsync'ed staging) running while another thread creates a plan, or merely
// allocates and frees device memory, came back with a sample in the wrong bin about once in 10^4 calls
// (tools/race_probe.py, profiles/r02_n_race_probe.txt; 0 in 7*10^4 without the other thread, 0 in 7*10^4 for
// device-resident calls, which allocate nothing) — found as a flaky dask test: the threaded scheduler creates the plan
// of a fresh graph while other blocks are already running.
//
// A block freed on a stream is reusable on that stream at once, on other streams once an event recorded at the free
// has completed.  Freed blocks are kept up to what the largest recent call held at once (at least 64 MiB, at most half of
// the device's memory; $XHIST_AMD_POOL_KEEP_GB overrides) and handed back to the driver beyond that, or by xhist_shutdown.
// ------------------------------------------------------------------------------------------
struct ScratchBlock {
  void* ptr = nullptr;
  size_t size = 0;
  int device = 0;
  hipStream_t stream = nullptr;  // last used on
  uint64_t owner = 0;            // (hipStreamPerThread names a different stream in every thread)
  hipEvent_t ev = nullptr;
  bool pending = false;          // `ev` marks the point on `stream` after which the block is free
  uint64_t freed_at = 0;         // tick of the free that put it into the cache (eviction: oldest first)
};
static uint64_t g_sc_tick = 0;
static std::mutex g_sc_mu;
static std::vector<ScratchBlock> g_sc_free;
static std::map<void*, ScratchBlock> g_sc_live;
static uint64_t g_sc_cached[64] = {0}, g_sc_limit[64] = {0};
// what callers hold right now, and its peak over the current and the previous window of 256 frees: the cache keeps no more
// than the largest RECENT call used (VERDICT r2 "weak" #8: a fixed half of the device was right for C5's 56 GB of record
// streams and hostile to a torch process sharing the GPU with a histogram of 10^6 samples)
static uint64_t g_sc_live_b[64] = {0}, g_sc_win_peak[64] = {0}, g_sc_prev_peak[64] = {0};
static uint32_t g_sc_win_n[64] = {0};
constexpr uint32_t kScratchWindow = 256;
constexpr uint64_t kScratchFloor = (uint64_t)64 << 20;

static void scratch_note_alloc(int device, size_t size) {  // g_sc_mu held
  if (device < 0 || device >= 64) return;
  g_sc_live_b[device] += size;
  g_sc_win_peak[device] = std::max(g_sc_win_peak[device], g_sc_live_b[device]);
}
static void scratch_note_free(int device, size_t size) {  // g_sc_mu held
  if (device < 0 || device >= 64) return;
  g_sc_live_b[device] -= std::min<uint64_t>(g_sc_live_b[device], size);
  if (++g_sc_win_n[device] >= kScratchWindow) {
    g_sc_prev_peak[device] = g_sc_win_peak[device];
    g_sc_win_peak[device] = g_sc_live_b[device];
    g_sc_win_n[device] = 0;
  }
}

static uint64_t scratch_owner(hipStream_t s) {
  static thread_local char key;
  return s == hipStreamPerThread ? (uint64_t)(uintptr_t)&key : 0;
}

static size_t scratch_round(size_t b) {  // size classes 1/8 apart: a cached block serves requests close to its size
  if (b < 4096) return 4096;
  size_t step = (size_t)1 << 9;
  while ((step << 4) <= b) step <<= 1;
  return (b + step - 1) / step * step;
}

// bytes the cache of `device` may keep: twice what the largest call of the last 256-512 frees held at once (at least 64 MiB,
// at most half of the device); $XHIST_AMD_POOL_KEEP_GB fixes it instead.  Call with g_sc_mu held and `device` current.
static uint64_t scratch_limit(int device) {
  if (device < 0 || device >= 64) return (uint64_t)2 << 30;
  static const double env_gb = [] { const char* e = getenv("XHIST_AMD_POOL_KEEP_GB"); return e && *e ? atof(e) : -1.0; }();
  if (env_gb >= 0) return (uint64_t)(env_gb * 1073741824.0) + 1;
  if (!g_sc_limit[device]) {  // the cap: half of the device's memory
    size_t free_b = 0, total_b = 0;
    uint64_t cap = (uint64_t)2 << 30;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) cap = std::max<uint64_t>(cap, (uint64_t)total_b / 2);
    g_sc_limit[device] = cap;
  }
  // twice the recent peak: the blocks ONE call cycles through are more than it ever holds at once (three routing passes over
  // 11 + 11 + 10 rows ask for three sets of sizes, 16.8 GB of blocks for 12.3 GB held) — a cache of exactly the peak evicted,
  // oldest first, precisely what the next call asks for first: 8 -> 700 ms per call
  const uint64_t recent = std::max(g_sc_win_peak[device], g_sc_prev_peak[device]);
  return std::min(g_sc_limit[device], std::max(kScratchFloor, 2 * recent));
}

// give cached blocks of `device` back to the driver until at most `keep` bytes stay (blocks still in flight are skipped
// unless `sync`, which waits for the device first); g_sc_mu held, `device` current
static void scratch_evict(int device, uint64_t keep, bool sync) {
  if (sync) (void)hipDeviceSynchronize();
  while (device >= 0 && device < 64 && g_sc_cached[device] > keep) {
    int best = -1;
    for (int i = 0; i < (int)g_sc_free.size(); ++i) {
      ScratchBlock& b = g_sc_free[(size_t)i];
      if (b.device != device) continue;
      if (b.pending && !sync && hipEventQuery(b.ev) != hipSuccess) continue;
      // the block that has sat in the cache longest goes first: what the current calls reuse is the newest (evicting by size
      // threw out the record streams a call had just freed, in favour of stale blocks of an earlier shape: 7 -> 418 ms per call)
      if (best < 0 || b.freed_at < g_sc_free[(size_t)best].freed_at) best = i;
    }
    if (best < 0) break;
    ScratchBlock b = g_sc_free[(size_t)best];
    g_sc_free.erase(g_sc_free.begin() + best);
    g_sc_cached[device] -= b.size;
    if (b.ev) (void)hipEventDestroy(b.ev);
    (void)hipFree(b.ptr);
  }
}

// same shape as hipMallocAsync / hipFreeAsync; the device the caller made current is the block's device
static bool scratch_use_hip_pool() {  // A/B switch for measurements only (the HIP pool is the unsafe one, see above)
  static const bool on = [] { const char* e = getenv("XHIST_AMD_SCRATCH"); return e && !strcmp(e, "hip-pool"); }();
  return on;
}

static hipError_t scratch_malloc(void** out, size_t bytes, hipStream_t stream) {
  if (scratch_use_hip_pool()) {
    static thread_local int warmed = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (warmed != dev) {  // (the A/B needs the pool to keep its memory, as the round-2 code before the switch did)
      hipMemPool_t pool;
      uint64_t keep = ~(uint64_t)0;
      if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      warmed = dev;
    }
    return hipMallocAsync(out, bytes, stream);
  }
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return hipErrorInvalidDevice;
  const size_t want = scratch_round(bytes);
  const uint64_t owner = scratch_owner(stream);
  {
    std::lock_guard<std::mutex> lk(g_sc_mu);
    const size_t slack = std::max<size_t>(want / 4, (size_t)1 << 20);
    int best = -1;
    for (int i = 0; i < (int)g_sc_free.size(); ++i) {
      ScratchBlock& b = g_sc_free[(size_t)i];
      if (b.device != device || b.size < want || b.size > want + slack) continue;
      bool ok = !b.pending || (b.stream == stream && b.owner == owner);
      if (!ok && hipEventQuery(b.ev) == hipSuccess) {
        b.pending = false;
        ok = true;
      }
      if (ok && (best < 0 || b.size < g_sc_free[(size_t)best].size)) best = i;
    }
    if (best >= 0) {
      ScratchBlock b = g_sc_free[(size_t)best];
      g_sc_free.erase(g_sc_free.begin() + best);
      if (device < 64) g_sc_cached[device] -= b.size;
      g_sc_live[b.ptr] = b;
      scratch_note_alloc(device, b.size);
      *out = b.ptr;
      return hipSuccess;
    }
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {  // out of memory: everything cached goes back first
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lk(g_sc_mu);
      scratch_evict(device, 0, true);
    }
    e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
  }
  ScratchBlock b;
  b.ptr = p;
  b.size = want;
  b.device = device;
  std::lock_guard<std::mutex> lk(g_sc_mu);
  g_sc_live[p] = b;
  scratch_note_alloc(device, b.size);
  *out = p;
  return hipSuccess;
}

// synced: the caller has synchronised `stream` since the block's last use (no event needed)
static hipError_t scratch_free(void* p, hipStream_t stream, bool synced = false) {
  if (!p) return hipSuccess;
  if (scratch_use_hip_pool()) return hipFreeAsync(p, stream);
  std::lock_guard<std::mutex> lk(g_sc_mu);
  auto it = g_sc_live.find(p);
  if (it == g_sc_live.end()) return hipErrorInvalidValue;
  ScratchBlock b = it->second;
  g_sc_live.erase(it);
  scratch_note_free(b.device, b.size);
  b.stream = stream;
  b.owner = scratch_owner(stream);
  b.pending = false;
  b.freed_at = ++g_sc_tick;
  if (!synced) {
    if (!b.ev && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) b.ev = nullptr;
    if (b.ev && hipEventRecord(b.ev, stream) == hipSuccess) {
      b.pending = true;
    } else {  // no event to wait on: wait now
      (void)hipStreamSynchronize(stream);
    }
  }
  g_sc_free.push_back(b);
  if (b.device >= 0 && b.device < 64) {
    g_sc_cached[b.device] += b.size;
    if (g_sc_cached[b.device] > scratch_limit(b.device)) scratch_evict(b.device, scratch_limit(b.device), false);
  }
  return hipSuccess;
}

extern "C" int xhist_scratch_stats(int device, uint64_t* stats, int n) {
  if (!stats || n < 4) return fail(XHIST_ERR_INVALID, "stats is NULL / shorter than 4");
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "device %d not available", device);
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  const int phys = physical_device(device);
  std::lock_guard<std::mutex> lk(g_sc_mu);
  stats[0] = phys < 64 ? g_sc_cached[phys] : 0;
  stats[1] = phys < 64 ? g_sc_live_b[phys] : 0;
  stats[2] = scratch_limit(phys);
  stats[3] = phys < 64 ? std::max(g_sc_win_peak[phys], g_sc_prev_peak[phys]) : 0;
  return XHIST_OK;
}

// test support: somebody else's kernel holding compute units (see include/xhist_amd.h)
__global__ void __launch_bounds__(64) debug_hold_kernel(long long ticks) {
  extern __shared__ unsigned char hold_smem[];
  if (threadIdx.x == 0) hold_smem[0] = 1;  // (the LDS is this kernel's whole point: keep the allocation alive)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

extern "C" int xhist_debug_hold_cus(int device, int workgroups, int lds_bytes, int64_t microseconds, void* stream) {
  if (workgroups < 1 || workgroups > 4096 || lds_bytes < 0 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 10 * 1000 * 1000)
    return fail(XHIST_ERR_INVALID, "hold_cus: workgroups in [1, 4096], lds_bytes in [0, 163840], microseconds in [0, 10^7]");
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "device %d not available", device);
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  if (lds_bytes > 48 * 1024 && hipFuncSetAttribute((const void*)debug_hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
    return fail(XHIST_ERR_HIP, "hold_cus: %s", hipGetErrorString(hipGetLastError()));
  hipLaunchKernelGGL(debug_hold_kernel, dim3((unsigned)workgroups), dim3(64), (size_t)lds_bytes, static_cast<hipStream_t>(stream), (long long)microseconds * 100);
  if (hipError_t e = hipGetLastError()) return fail(XHIST_ERR_HIP, "hold_cus launch: %s", hipGetErrorString(e));
  return XHIST_OK;
}

static void trim_pools() {  // xhist_shutdown: every cached block of every device goes back to the driver
  std::lock_guard<std::mutex> lk(g_sc_mu);
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) return;
  for (int d = 0; d < 64; ++d) {
    if (!g_sc_cached[d]) continue;
    if (hipSetDevice(d) == hipSuccess) scratch_evict(d, 0, true);
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
  // packed 16-byte bucket entries for float64 samples on NON-uniform edges (count_le_pack): float64 edges + entries;
  // pk_np = 0 (not offered: some bucket would hold more than three edges, or the edges leave float32's range), 2 or 3
  TableSet ts_pk;
  int pk_np = 0;
  TableSet ts_pk32;    // the same for float32 SAMPLES: entries only (exact float32 thresholds, no redo path, no edges in LDS)
  int pk32_np = 0;
  int pack_pref = 0;   // 0 auto, 1 packed entries whenever the plan has them, -1 never
  bool uns = false;    // the int64-domain inputs hold unsigned 64-bit values (XHIST_CMP_UNSIGNED)
  bool huge = false;   // some dimension has more than 65535 edges: no bucket tables (lut_k = 0)
  bool arith = false;  // every dimension has arithmetic (numpy.linspace) edges: table-free digitize available
  bool arith32 = false;  // ... and float32 samples can be decided in float32 arithmetic (DimTable::a32_h > 0 in every dimension)
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
  int records48_pref = 0;  // routing pass, float64 weights: 0 packed 8-byte records while the weights have one sign, -1 never
  uint32_t* mixed_hint = nullptr;  // pinned host words the GPU sets: [0] a call met weights of both signs, [1] a chunk pool ran dry (see execute_partitioned_fused)
  int route_spl = 0;       // routing pass: samples per lane and tile (0 auto; 4; 8 = auto: the long tile exists only where auto picks it)
  int flat_rows = 0;       // dense short rows streamed flat (hist_flat_rows): -1 off, 0 auto, 1 for any row length below 65536
  int min_parts = 0;       // partitioned mode: bins are cut finer until a pass has this many partitions (0 auto = 16; 1 = never)
  int route_grid = 0, acc_grid = 0;  // workgroups of the routing / adding-up pass of execute_partitioned_fused (0 auto) — scaling runs
  int exchange_pref = 0;   // exchange mode of the partitioned path (xhist_exchange.hip.h): -1 never, 0 where eligible and the probe's window holds enough samples, 1 whenever the kernel can run (tests)
  int exchange_skip = 0;       // an exchange kernel gave up in flight (deadline, placement): eligible calls that still stay on the classic passes
  int exchange_backoff = 16;   // ... and how many that will be after the next abort (doubles per abort, back to 16 after a clean call)
  bool exchange_ran_last = false;  // the last eligible call launched the exchange kernel
  int exchange_min_pct = 0;    // window coverage (per cent of the probe's samples) from which the mode takes a call; 0 = kExchMinPpm
  int exchange_arrive_us = 0;  // how long its workgroups wait for one another to start; 0 = 200 us
  uint32_t exchange_aborts_seen = 0;
  size_t exchange_occ_lds = 0;  // the LDS size the occupancy question below was asked for, and its answer
  bool exchange_occ_ok = false;
  int exchange_budget_ms = 0;  // deadline of a workgroup's waits in that mode: 0 = 500 ms; -1: every wait gives up at once (tests of the fallback)
  int route_pool_pct = 0;  // routing pass: chunk pool cut to this percentage of its worst-case size (tests of the pool-dry path; 0 = full)
  int slices_pref = 0;  // 0 auto, 1 prefer bin slices for histograms beyond LDS, -1 never
  int arith_pref = 0;  // 0 auto, 1 table-free digitize whenever the edges are arithmetic, -1 never
  int arith32_pref = 0;  // 0 auto, 1 float32 arithmetic digitize for float32 samples wherever the plan offers it, -1 never
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
  HIPC(hipGetDeviceProperties(&prop, physical_device(device)));
  if (name && name_cap) {
    strncpy(name, prop.gcnArchName, name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
  return XHIST_OK;
}
