// xhist_capi.hip — host side of libxhist_amd.so: the C ABI declared in include/xhist_amd.h.
// Plans (device-resident edge tables), kernel-family selection, launch geometry, host staging.
// No CPU compute path exists here on purpose: without a HIP device every compute entry point
// fails with XHIST_ERR_NO_DEVICE.
#include "xhist_kernels.hip.h"
#include "xhist_partition.hip.h"
#include "xhist_lanes.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/xhist_amd.h"

using namespace xhist;

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

static int n_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
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
  int arith_pref = 0;  // 0 auto, 1 table-free digitize whenever the edges are arithmetic, -1 never
  int lanes = 0;      // 0 auto, 1 prefer the row-per-lane kernels whenever they are legal, -1 never
  int lds_copies = 0;
  int profile = 0;
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

// Build the device table blob of one compare domain:
//   [per-dimension edge arrays, 8-byte aligned] [per-dimension bucket tables (uint32 x K)]
// dom: 0 float64, 1 int64, 2 float32 thresholds.  `words[d]` holds dimension d's edge array
// already converted to the domain's element type; `edges` are the caller's original arrays.
static int build_domain(xhist_plan* p, int dom, bool lut16, int n_inputs, const int64_t* n_edges,
                        const std::vector<std::vector<uint64_t>>& words, const void* const* edges, TableSet* ts) {
  DimTable* dims = ts->dim;
  uint64_t** d_blob_out = &ts->blob;
  int32_t* table_words_out = &ts->words;
  int* max_cnt_out = &ts->max_cnt;
  int32_t edge_off = 0;
  int64_t max_e = 0;
  for (int d = 0; d < n_inputs; ++d) {
    DimTable& t = dims[d];
    memset(&t, 0, sizeof t);
    const int E = (int)n_edges[d];
    max_e = std::max<int64_t>(max_e, E);
    t.n_edges = E;
    t.nb = E - 1;
    t.edge_off = edge_off;
    edge_off += (int32_t)words[d].size();
    double range;
    if (dom == 0) {
      const double* e = static_cast<const double*>(edges[d]);
      t.e0_f = e[0];
      t.eL_f = e[E - 1];
      range = t.eL_f - t.e0_f;
    } else if (dom == 1) {
      const int64_t* e = static_cast<const int64_t*>(edges[d]);
      t.e0_i = e[0];
      t.eL_i = e[E - 1];
      range = (double)((uint64_t)t.eL_i - (uint64_t)t.e0_i);
    } else {
      const double* e = static_cast<const double*>(edges[d]);
      const float* thr = reinterpret_cast<const float*>(words[d].data());
      float last = (float)e[E - 1];  // largest float32 <= e_last
      if ((double)last > e[E - 1]) last = std::nextafterf(last, -INFINITY);
      t.e0_f = (double)thr[0];
      t.eL_f = (double)last;
      range = (double)((float)t.eL_f - (float)t.e0_f);
    }
    int K = std::min(4096, std::max(8, next_pow2((int)std::min<int64_t>(4 * (int64_t)E, 1 << 20))));
    if (lut16) K *= 2;  // 2-byte entries: twice the buckets for the same LDS bytes
    // more than 65535 edges: `start` no longer fits the 16-bit table fields — no bucket table at
    // all (lut_k = 0): digitize is a plain binary search over the edge array (generic family), or
    // table-free when the edges are arithmetic
    const bool no_lut = E > 65535;
    double scale = (double)K / range;
    if (dom == 2) scale = (double)(float)scale;
    if (!(range > 0.0) || !std::isfinite(range) || !std::isfinite(scale) || !(scale > 0.0)) {
      K = 1;  // degenerate span: one bucket holding every edge, pure binary search
      scale = 0.0;
    }
    if (no_lut) { K = 0; scale = 0.0; }
    t.lut_k = K;
    t.scale = scale;
    if (dom == 2) t.bias = (double)(-(float)t.e0_f * (float)scale);
    else if (dom == 0) t.bias = -t.e0_f * scale;
    if (!std::isfinite(t.bias)) {  // e.g. e_0 = -inf with scale 0: keep the map defined (bucket 0)
      t.bias = 0.0;
      if (K > 1) { K = 1; t.lut_k = 1; t.scale = 0.0; }
    }
  }
  int64_t stride = 1;
  for (int d = n_inputs - 1; d >= 0; --d) {
    dims[d].out_stride = stride;
    stride *= dims[d].nb;
  }
  // bucket tables follow the edges; lut_off counts table ENTRIES (4-byte, or 2-byte for lut16)
  int32_t off = (lut16 ? 4 : 2) * edge_off;
  for (int d = 0; d < n_inputs; ++d) {
    dims[d].lut_off = off;
    off += dims[d].lut_k;
  }
  const int32_t per_word = lut16 ? 4 : 2;
  const int32_t table_words = (off + per_word - 1) / per_word;
  std::vector<uint64_t> blob((size_t)table_words, 0);
  for (int d = 0; d < n_inputs; ++d) memcpy(blob.data() + dims[d].edge_off, words[d].data(), words[d].size() * 8);

  uint64_t* d_blob = nullptr;
  int32_t* d_scratch = nullptr;
  auto cleanup = [&](int rc) {
    if (d_scratch) (void)hipFree(d_scratch);
    if (rc != XHIST_OK && d_blob) (void)hipFree(d_blob);
    return rc;
  };
#define HIPP(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return cleanup(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  HIPP(hipMalloc(&d_blob, blob.size() * 8));
  HIPP(hipMalloc(&d_scratch, (size_t)max_e * 4));
  HIPP(hipMemcpy(d_blob, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
  for (int d = 0; d < n_inputs; ++d) {
    if (dims[d].lut_k == 0) continue;
    if (dom == 0 && !lut16) hipLaunchKernelGGL((build_tables<0, false>), dim3(1), dim3(256), 0, 0, dims[d], d_blob, d_scratch);
    else if (dom == 0) hipLaunchKernelGGL((build_tables<0, true>), dim3(1), dim3(256), 0, 0, dims[d], d_blob, d_scratch);
    else if (dom == 1) hipLaunchKernelGGL((build_tables<1, false>), dim3(1), dim3(256), 0, 0, dims[d], d_blob, d_scratch);
    else if (!lut16) hipLaunchKernelGGL((build_tables<2, false>), dim3(1), dim3(256), 0, 0, dims[d], d_blob, d_scratch);
    else hipLaunchKernelGGL((build_tables<2, true>), dim3(1), dim3(256), 0, 0, dims[d], d_blob, d_scratch);
    HIPP(hipGetLastError());
    HIPP(hipDeviceSynchronize());
  }
  HIPP(hipMemcpy(blob.data(), d_blob, blob.size() * 8, hipMemcpyDeviceToHost));
#undef HIPP
  const uint32_t* lut4 = reinterpret_cast<const uint32_t*>(blob.data());
  const uint16_t* lut2 = reinterpret_cast<const uint16_t*>(blob.data());
  for (int d = 0; d < n_inputs; ++d) {
    DimTable& t = dims[d];
    uint32_t maxcnt = 0;
    uint64_t total = 0;
    if (t.lut_k == 0) { maxcnt = (uint32_t)t.n_edges; total = (uint64_t)t.n_edges; }
    for (int b = 0; b < t.lut_k; ++b) {
      uint32_t cnt;
      if (lut16) {
        const uint32_t next = b + 1 < t.lut_k ? lut2[t.lut_off + b + 1] : (uint32_t)t.n_edges;
        cnt = next - lut2[t.lut_off + b];
      } else {
        cnt = lut4[t.lut_off + b] >> 16;
      }
      maxcnt = std::max(maxcnt, cnt);
      total += cnt;
    }
    if (total != (uint64_t)t.n_edges) return cleanup(fail(XHIST_ERR_HIP, "bucket table of dim %d is inconsistent", d));
    *max_cnt_out = std::max<int>(*max_cnt_out, (int)maxcnt);
    int steps = 0;
    while ((1u << steps) <= maxcnt) ++steps;
    t.steps = steps;
  }
  (void)p;
  *d_blob_out = d_blob;
  *table_words_out = table_words;
  return cleanup(XHIST_OK);
}

extern "C" int xhist_plan_create(int device, int n_inputs, const void* const* edges, const int64_t* n_edges,
                                 int cmp_domain, xhist_plan** out_plan) {
  if (!out_plan) return fail(XHIST_ERR_INVALID, "plan out-pointer is NULL");
  *out_plan = nullptr;
  if (n_inputs < 1 || n_inputs > XHIST_MAX_DIMS)
    return fail(XHIST_ERR_INVALID, "n_inputs must be in [1, %d], got %d", XHIST_MAX_DIMS, n_inputs);
  if (!edges || !n_edges) return fail(XHIST_ERR_INVALID, "edges / n_edges is NULL");
  if (cmp_domain != XHIST_CMP_F64 && cmp_domain != XHIST_CMP_I64)
    return fail(XHIST_ERR_INVALID, "unknown compare domain %d", cmp_domain);
  int64_t max_e = 0;
  for (int d = 0; d < n_inputs; ++d) {
    if (!edges[d]) return fail(XHIST_ERR_INVALID, "edges[%d] is NULL", d);
    if (n_edges[d] < 1) return fail(XHIST_ERR_INVALID, "edges[%d] needs at least one edge", d);
    if (n_edges[d] > ((int64_t)1 << 30))
      return fail(XHIST_ERR_UNSUPPORTED, "edges[%d] has %lld edges; this build supports at most 2^30 per dimension", d,
                  (long long)n_edges[d]);
    max_e = std::max(max_e, n_edges[d]);
    if (cmp_domain == XHIST_CMP_F64) {
      const double* e = static_cast<const double*>(edges[d]);
      for (int64_t j = 0; j < n_edges[d]; ++j) {
        if (e[j] != e[j]) return fail(XHIST_ERR_EDGES, "edges[%d] contains NaN", d);
        if (j && e[j] < e[j - 1]) return fail(XHIST_ERR_EDGES, "bins must increase monotonically (edges[%d])", d);
      }
    } else {
      const int64_t* e = static_cast<const int64_t*>(edges[d]);
      for (int64_t j = 1; j < n_edges[d]; ++j)
        if (e[j] < e[j - 1]) return fail(XHIST_ERR_EDGES, "bins must increase monotonically (edges[%d])", d);
    }
  }
  if (device < 0 || device >= n_devices())
    return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available (%d visible); this library has no CPU path", device,
                n_devices());
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;

  xhist_plan* p = new (std::nothrow) xhist_plan();
  if (!p) return fail(XHIST_ERR_NOMEM, "out of host memory");
  p->device = device;
  p->n_dims = n_inputs;
  p->cmp = cmp_domain;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    p->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p->lds_max = prop.sharedMemPerBlock;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess && optin > 0)
      p->lds_max = std::max(p->lds_max, (size_t)optin);
  }

  // ---- table sets: one per compare domain this plan can be asked for -----------------------
  //   native (float64 or int64): every kernel family;  float32 thresholds: float32 fast family
  int64_t n_bins = 1;
  for (int d = 0; d < n_inputs; ++d) {
    const int64_t nb = n_edges[d] - 1;
    if (nb > 0 && n_bins > (int64_t)1 << 40) {
      delete p;
      return fail(XHIST_ERR_UNSUPPORTED, "histogram has more than 2^40 bins");
    }
    n_bins *= nb;
  }
  p->n_bins = n_bins;
  std::vector<std::vector<uint64_t>> words(n_inputs);
  std::vector<double> lo(n_inputs), hi(n_inputs);
  for (int d = 0; d < n_inputs; ++d) {
    const int E = (int)n_edges[d];
    words[d].assign((size_t)E, 0);
    memcpy(words[d].data(), edges[d], (size_t)E * 8);
  }
  // every edge array is followed by 4 sentinels that compare false against any sample (NaN), so
  // the linear in-bucket count may read up to 4 entries past a bucket's start unconditionally
  const uint64_t kNaN64 = 0x7ff8000000000000ull;
  for (int d = 0; d < n_inputs; ++d)
    for (int k = 0; k < 4; ++k) words[d].push_back(cmp_domain == XHIST_CMP_F64 ? kNaN64 : 0x7fffffffffffffffull);
  // more than 65535 edges in some dimension: only the native set, without bucket tables (the vector
  // family then runs table-free on arithmetic edges, everything else takes the generic family)
  p->huge = max_e > 65535;
  const bool vector_sets = cmp_domain == XHIST_CMP_F64 && !p->huge;
  int rc = build_domain(p, cmp_domain == XHIST_CMP_F64 ? 0 : 1, false, n_inputs, n_edges, words, edges, &p->ts[0][0]);
  if (rc == XHIST_OK && vector_sets) rc = build_domain(p, 0, true, n_inputs, n_edges, words, edges, &p->ts[0][1]);
  if (rc == XHIST_OK && vector_sets) {
    // float32 thresholds: thr_j = smallest float32 >= e_j (then (double)x >= e_j <=> x >= thr_j)
    for (int d = 0; d < n_inputs; ++d) {
      const int E = (int)n_edges[d];
      const double* e = static_cast<const double*>(edges[d]);
      std::vector<float> thr((size_t)E + 6, std::nanf(""));  // >= 4 NaN sentinels after the thresholds
      for (int j = 0; j < E; ++j) {
        float f = (float)e[j];
        if ((double)f < e[j]) f = std::nextafterf(f, INFINITY);
        thr[(size_t)j] = f;
      }
      words[d].assign(((size_t)E + 5) / 2, 0);
      memcpy(words[d].data(), thr.data(), words[d].size() * 8);
    }
    rc = build_domain(p, 2, false, n_inputs, n_edges, words, edges, &p->ts[1][0]);
    if (rc == XHIST_OK) rc = build_domain(p, 2, true, n_inputs, n_edges, words, edges, &p->ts[1][1]);
  }
  if (rc != XHIST_OK) {
    for (auto& dom : p->ts)
      for (auto& t : dom)
        if (t.blob) (void)hipFree(t.blob);
    delete p;
    return rc;
  }
  // ---- arithmetic edges: e_j == fl(fl(j * step) + e_0) for every j < nb, step = (e_nb - e_0) / nb ----
  // (what numpy.linspace / histogram_bin_edges produce for `bins=int`).  Checked edge by edge with the
  // two roundings kept apart (volatile product: no fma contraction), and only when bins are well
  // resolved (step >= 4 ulp of the largest magnitude) — the bound count_le_arith's guess relies on.
  if (cmp_domain == XHIST_CMP_F64) {
    bool all = true;
    for (int d = 0; d < n_inputs && all; ++d) {
      const double* e = static_cast<const double*>(edges[d]);
      const int nb = (int)n_edges[d] - 1;
      bool ok = nb >= 1 && std::isfinite(e[0]) && std::isfinite(e[nb]);
      double step = 0.0;
      if (ok) {
        step = (e[nb] - e[0]) / (double)nb;
        const double mag = std::max(std::max(std::fabs(e[0]), std::fabs(e[nb])), e[nb] - e[0]);
        const double ulp = std::nextafter(mag, INFINITY) - mag;
        ok = std::isfinite(step) && step > 0.0 && step >= 4.0 * ulp && std::isfinite(1.0 / step);
      }
      for (int j = 0; j < nb && ok; ++j) {
        volatile double m = (double)j * step;
        ok = (m + e[0]) == e[j];
      }
      if (ok) ok = e[nb] >= e[nb - 1];
      all = ok;
      if (ok)
        for (auto& dom : p->ts[0]) {
          dom.dim[d].step = step;
          dom.dim[d].inv_step = 1.0 / step;
          dom.dim[d].arith = 1;
        }
    }
    p->arith = all;
  }
  *out_plan = p;
  return XHIST_OK;
}


extern "C" int xhist_plan_destroy(xhist_plan* p) {
  if (!p) return XHIST_OK;
  DeviceGuard g;
  if (g.set(p->device) == XHIST_OK) {
    for (auto& dom : p->ts)
      for (auto& t : dom)
        if (t.blob) (void)hipFree(t.blob);
    for (auto& e : p->ring) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
  delete p;
  return XHIST_OK;
}

extern "C" int xhist_plan_set_param(xhist_plan* p, const char* key, int64_t value) {
  if (!p || !key) return fail(XHIST_ERR_INVALID, "plan / key is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  if (!strcmp(key, "block_threads")) {
    if (value != 0 && (value < 64 || value > 1024 || value % 64)) return fail(XHIST_ERR_INVALID, "block_threads must be a multiple of 64 in [64, 1024]");
    p->block_threads = (int)value;
  } else if (!strcmp(key, "grid_blocks")) {
    if (value < 0) return fail(XHIST_ERR_INVALID, "grid_blocks must be >= 0");
    p->grid_blocks = (int)std::min<int64_t>(value, 1 << 30);
  } else if (!strcmp(key, "force_global")) {
    p->force_global = value != 0;
  } else if (!strcmp(key, "force_generic")) {
    p->force_generic = value != 0;
  } else if (!strcmp(key, "partition")) {
    p->partition = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "lanes")) {
    p->lanes = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "arith")) {
    p->arith_pref = value > 0 ? 1 : (value < 0 ? -1 : 0);
  } else if (!strcmp(key, "lds_copies")) {
    if (value != 0 && (value < 1 || value > 32 || (value & (value - 1)))) return fail(XHIST_ERR_INVALID, "lds_copies must be a power of two in [1, 32]");
    p->lds_copies = (int)value;
  } else if (!strcmp(key, "profile")) {
    // value = number of most recent executes whose main-kernel duration is kept (0 = off)
    if (value < 0 || value > 4096) return fail(XHIST_ERR_INVALID, "profile must be in [0, 4096]");
    DeviceGuard g;
    if (int rc = g.set(p->device)) return rc;
    while ((int64_t)p->ring.size() < value) {
      hipEvent_t a = nullptr, b = nullptr;
      HIPC(hipEventCreate(&a));
      HIPC(hipEventCreate(&b));
      p->ring.emplace_back(a, b);
    }
    p->profile = (int)value;
    p->n_recorded = 0;
  } else {
    return fail(XHIST_ERR_INVALID, "unknown parameter '%s'", key);
  }
  return XHIST_OK;
}

extern "C" int xhist_plan_describe(xhist_plan* p, char* buf, size_t cap) {
  if (!p || !buf || !cap) return fail(XHIST_ERR_INVALID, "plan / buf is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  strncpy(buf, p->desc.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return XHIST_OK;
}

extern "C" int xhist_plan_profile_read(xhist_plan* p, float* ms, int cap, int* n_out) {
  if (!p || !ms || !n_out || cap < 0) return fail(XHIST_ERR_INVALID, "plan / ms / n_out is NULL");
  std::lock_guard<std::mutex> lk(p->mu);
  *n_out = 0;
  if (!p->profile || p->n_recorded == 0) return XHIST_OK;
  DeviceGuard g;
  if (int rc = g.set(p->device)) return rc;
  const int64_t kept = std::min<int64_t>(p->n_recorded, p->profile);
  for (int64_t k = p->n_recorded - kept; k < p->n_recorded && *n_out < cap; ++k) {
    auto& e = p->ring[(size_t)(k % p->profile)];
    HIPC(hipEventSynchronize(e.second));
    HIPC(hipEventElapsedTime(&ms[*n_out], e.first, e.second));
    ++*n_out;
  }
  p->n_recorded = 0;
  return XHIST_OK;
}

// ------------------------------------------------------------------------------------------
// kernel selection
// ------------------------------------------------------------------------------------------
// Tables for the vector / lanes / partition kernels: the finer uint16 set whenever none of its
// buckets holds more than 4 edges (linear scan, *scan = that maximum), else the uint32
// (start, cnt) set with the branch-free binary search (*scan = 0).
static const TableSet& pick_tables(const xhist_plan* p, bool use_f32, int* scan) {
  const TableSet& fine = p->ts[use_f32 ? 1 : 0][1];
  if (fine.blob && fine.max_cnt >= 1 && fine.max_cnt <= 4) {
    *scan = fine.max_cnt;
    return fine;
  }
  *scan = 0;
  return p->ts[use_f32 ? 1 : 0][0];
}

// HIP-event pair around the kernels of one execute ("profile" plan parameter) + the launch
// description kept for xhist_plan_describe.  begin() before the first launch, end() after the last.
struct LaunchRecord {
  xhist_plan* p;
  hipStream_t stream;
  int slot = -1;
  LaunchRecord(xhist_plan* plan, hipStream_t s) : p(plan), stream(s) {}
  int begin(int profile) {
    if (!profile) return XHIST_OK;
    std::lock_guard<std::mutex> lk(p->mu);
    slot = (int)(p->n_recorded % profile);
    HIPC(hipEventRecord(p->ring[(size_t)slot].first, stream));
    return XHIST_OK;
  }
  int end(const char* desc) {
    std::lock_guard<std::mutex> lk(p->mu);
    p->desc = desc;
    if (slot >= 0) {
      HIPC(hipEventRecord(p->ring[(size_t)slot].second, stream));
      ++p->n_recorded;
    }
    return XHIST_OK;
  }
};

typedef void (*kernel_fn)(const Params);
typedef void (*kernel_fn_acc)(const uint16_t*, const double*, const uint64_t*, void*, int64_t, int, int);
typedef void (*kernel_fn_count)(const Params, uint32_t*);
typedef void (*kernel_fn_lanes)(const Params, int32_t, int64_t);
typedef void (*kernel_fn_scatter)(const uint32_t*, const void*, int64_t, const uint64_t*, uint16_t*, double*, int, int);

// Samples a lane bins as one branch-free batch = VEC x UNROLL, capped by register pressure: per
// sample and dimension the batch keeps the value, its running count and (linear scan) up to
// SCAN edge values in VGPRs, and 1024-thread workgroups leave 128 VGPRs per lane.
constexpr int unroll_for(int D, int vec, int scan) {
  int cap = D == 1 ? 16 : (D == 2 ? 8 : 4);
  if (D >= 2 && scan >= 3) cap /= 2;
  if (D == 1 && scan >= 3 && vec == 4) cap = 8;
  const int u = cap / vec < 1 ? 1 : cap / vec;
  return u > 4 ? 4 : u;
}

// partitioned mode: pseudo "hist" codes selecting the two part_pass kernels, and their geometry
constexpr int kHistPartCount = 4, kHistLanes = 6, kHistLanes16 = 7;
constexpr int kPartMaxParts = 256;

template <typename ST, typename WT, int D, int SCAN>
static kernel_fn fast_pick(int hist) {
  constexpr bool unweighted = std::is_same<WT, NoWeight>::value;
  constexpr int wsz = unweighted ? 0 : (int)sizeof(typename std::conditional<unweighted, float, WT>::type);
  constexpr int VEC = 16 / (((int)sizeof(ST) > wsz) ? (int)sizeof(ST) : wsz);
  constexpr int U = unroll_for(D, VEC, SCAN);
  if (hist == kHistPartCount) return (kernel_fn)part_count<ST, D, VEC, SCAN>;
  if (hist == kHistLanes) return (kernel_fn)hist_lanes<ST, WT, D, SCAN, (D == 1 ? 8 : 4), false>;
  if (hist == kHistLanes16) {
    if constexpr (unweighted) return (kernel_fn)hist_lanes<ST, WT, D, SCAN, (D == 1 ? 8 : 4), true>;
    else return nullptr;
  }
  if (hist == kHistLds) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistLds, SCAN>;
  if (hist == kHistPacked) {
    if constexpr (unweighted) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistPacked, SCAN>;
    else return nullptr;
  }
  return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistGlobal, SCAN>;
}

// table-free digitize (arithmetic edges): only the kernels that mode is selected for
template <typename ST, typename WT, int D>
static kernel_fn fast_pick_arith(int hist) {
  constexpr bool unweighted = std::is_same<WT, NoWeight>::value;
  constexpr int wsz = unweighted ? 0 : (int)sizeof(typename std::conditional<unweighted, float, WT>::type);
  constexpr int VEC = 16 / (((int)sizeof(ST) > wsz) ? (int)sizeof(ST) : wsz);
  constexpr int U = unroll_for(D, VEC, kScanArith);
  if (hist == kHistPartCount) return (kernel_fn)part_count<ST, D, VEC, kScanArith>;
  if (hist == kHistLds) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistLds, kScanArith>;
  if (hist == kHistPacked) {
    if constexpr (unweighted) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistPacked, kScanArith>;
    else return nullptr;
  }
  if (hist == kHistGlobal) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistGlobal, kScanArith>;
  return nullptr;
}

template <typename ST, typename WT, int D>
static kernel_fn fast_pick_s(int scan, int hist) {
  switch (scan) {
    case kScanArith: return fast_pick_arith<ST, WT, D>(hist);
    case 1: return fast_pick<ST, WT, D, 1>(hist);
    case 2: return fast_pick<ST, WT, D, 2>(hist);
    case 3: return fast_pick<ST, WT, D, 3>(hist);
    case 4: return fast_pick<ST, WT, D, 4>(hist);
    default: return fast_pick<ST, WT, D, 0>(hist);
  }
}

template <typename ST, typename WT>
static kernel_fn fast_pick_d(int D, int scan, int hist) {
  switch (D) {
    case 1: return fast_pick_s<ST, WT, 1>(scan, hist);
    case 2: return fast_pick_s<ST, WT, 2>(scan, hist);
    case 3: return fast_pick_s<ST, WT, 3>(scan, hist);
    default: return nullptr;
  }
}

template <typename ST>
static kernel_fn fast_pick_w(int wdt, int D, int scan, int hist) {
  switch (wdt) {
    case -1: return fast_pick_d<ST, NoWeight>(D, scan, hist);
    case XHIST_F64: return fast_pick_d<ST, double>(D, scan, hist);
    case XHIST_F32: return fast_pick_d<ST, float>(D, scan, hist);
    default: return nullptr;
  }
}

// Integer and half-precision samples (category ids, sensor counts, packed fields): the same vector
// kernel with an in-register conversion to double — numpy compares them in float64 against
// float64 edges too.  Kept to the shapes that matter so the instantiation count stays small:
// one input, unweighted or float64 weights, LDS or global histogram, uniform-style tables
// (SCAN 1) or binary search (SCAN 0); everything else takes the generic family.
template <typename ST>
static kernel_fn small_pick(int wdt, int D, int scan, int hist) {
  if (D != 1 || (scan != 0 && scan != 1) || (hist != kHistLds && hist != kHistGlobal)) return nullptr;
  if (wdt == -1) {
    constexpr int VEC = 16 / (int)sizeof(ST);
    constexpr int U0 = unroll_for(1, VEC, 0), U1 = unroll_for(1, VEC, 1);
    if (hist == kHistLds) return scan ? (kernel_fn)hist_fast<ST, NoWeight, 1, VEC, U1, kHistLds, 1> : (kernel_fn)hist_fast<ST, NoWeight, 1, VEC, U0, kHistLds, 0>;
    return scan ? (kernel_fn)hist_fast<ST, NoWeight, 1, VEC, U1, kHistGlobal, 1> : (kernel_fn)hist_fast<ST, NoWeight, 1, VEC, U0, kHistGlobal, 0>;
  }
  if (wdt == XHIST_F64) {
    constexpr int VEC = 16 / (sizeof(ST) > 8 ? (int)sizeof(ST) : 8);
    constexpr int U0 = unroll_for(1, VEC, 0), U1 = unroll_for(1, VEC, 1);
    if (hist == kHistLds) return scan ? (kernel_fn)hist_fast<ST, double, 1, VEC, U1, kHistLds, 1> : (kernel_fn)hist_fast<ST, double, 1, VEC, U0, kHistLds, 0>;
    return scan ? (kernel_fn)hist_fast<ST, double, 1, VEC, U1, kHistGlobal, 1> : (kernel_fn)hist_fast<ST, double, 1, VEC, U0, kHistGlobal, 0>;
  }
  return nullptr;
}

static kernel_fn fast_kernel(int sdt, int wdt, int D, int scan, int hist, int* vec) {
  const int ssz = dtype_size(sdt), wsz = wdt < 0 ? 0 : dtype_size(wdt);
  *vec = 16 / std::max(ssz, wsz);
  switch (sdt) {
    case XHIST_F64: return fast_pick_w<double>(wdt, D, scan, hist);
    case XHIST_F32: return fast_pick_w<float>(wdt, D, scan, hist);
    case XHIST_I32: return small_pick<int32_t>(wdt, D, scan, hist);
    case XHIST_I64: return small_pick<int64_t>(wdt, D, scan, hist);
    case XHIST_I16: return small_pick<int16_t>(wdt, D, scan, hist);
    case XHIST_U8: return small_pick<uint8_t>(wdt, D, scan, hist);
    case XHIST_F16: return small_pick<_Float16>(wdt, D, scan, hist);
    default: return nullptr;
  }
}

typedef void (*kernel_fn_rows1)(const Params, int32_t);

template <typename ST>
static kernel_fn_rows1 rows1_pick(int scan) {
  switch (scan) {
    case 1: return (kernel_fn_rows1)hist_lanes_rows1<ST, 1>;
    case 2: return (kernel_fn_rows1)hist_lanes_rows1<ST, 2>;
    case 3: return (kernel_fn_rows1)hist_lanes_rows1<ST, 3>;
    case 4: return (kernel_fn_rows1)hist_lanes_rows1<ST, 4>;
    default: return (kernel_fn_rows1)hist_lanes_rows1<ST, 0>;
  }
}

static kernel_fn_rows1 rows1_kernel(int sdt, int scan) {
  if (sdt == XHIST_F64) return rows1_pick<double>(scan);
  if (sdt == XHIST_F32) return rows1_pick<float>(scan);
  return nullptr;
}

static kernel_fn generic_kernel(int cmp, bool weighted, bool lds) {
  if (cmp == XHIST_CMP_F64) {
    if (weighted) return lds ? (kernel_fn)hist_generic<0, true, true> : (kernel_fn)hist_generic<0, true, false>;
    return lds ? (kernel_fn)hist_generic<0, false, true> : (kernel_fn)hist_generic<0, false, false>;
  }
  if (weighted) return lds ? (kernel_fn)hist_generic<1, true, true> : (kernel_fn)hist_generic<1, true, false>;
  return lds ? (kernel_fn)hist_generic<1, false, true> : (kernel_fn)hist_generic<1, false, false>;
}

// ------------------------------------------------------------------------------------------
// device-resident execute
// ------------------------------------------------------------------------------------------
static int validate_arrays(const xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows,
                           int64_t n_cols, const void* out, int out_dtype) {
  if (!p) return fail(XHIST_ERR_INVALID, "plan is NULL");
  if (!samples) return fail(XHIST_ERR_INVALID, "samples is NULL");
  if (n_rows < 0 || n_cols < 0) return fail(XHIST_ERR_INVALID, "negative shape");
  const bool empty = n_rows == 0 || n_cols == 0;
  for (int d = 0; d < p->n_dims; ++d) {
    if (!empty && !samples[d].data) return fail(XHIST_ERR_INVALID, "samples[%d].data is NULL", d);
    if (!dtype_size(samples[d].dtype)) return fail(XHIST_ERR_INVALID, "samples[%d] has unknown dtype tag %d", d, samples[d].dtype);
    if (samples[d].row_stride < 0 || samples[d].col_stride < 0 || samples[d].inner_rows < 0 || samples[d].outer_stride < 0)
      return fail(XHIST_ERR_UNSUPPORTED, "negative strides are not supported; pass a contiguous copy");
    if (p->cmp == XHIST_CMP_I64 && (!dtype_is_int(samples[d].dtype) || samples[d].dtype == XHIST_U64))
      return fail(XHIST_ERR_UNSUPPORTED, "int64 compare domain needs signed/small integer samples (got dtype tag %d)", samples[d].dtype);
  }
  if (weights) {
    if (!empty && !weights->data) return fail(XHIST_ERR_INVALID, "weights.data is NULL");
    if (!dtype_size(weights->dtype)) return fail(XHIST_ERR_INVALID, "weights has unknown dtype tag %d", weights->dtype);
    if (weights->row_stride < 0 || weights->col_stride < 0 || weights->inner_rows < 0 || weights->outer_stride < 0)
      return fail(XHIST_ERR_UNSUPPORTED, "negative strides are not supported; pass a contiguous copy");
    if (out_dtype != XHIST_F64) return fail(XHIST_ERR_INVALID, "weighted histograms are float64 (out_dtype XHIST_F64)");
  } else if (out_dtype != XHIST_I64) {
    return fail(XHIST_ERR_INVALID, "unweighted histograms are int64 (out_dtype XHIST_I64)");
  }
  if (!out && n_rows * p->n_bins > 0) return fail(XHIST_ERR_INVALID, "out is NULL");
  return XHIST_OK;
}

// zero n 8-byte output words on `stream` (see zero_words)
static int zero_output(void* out, int64_t n_words, hipStream_t stream) {
  if (n_words <= 0) return XHIST_OK;
  const int grid = (int)std::min<int64_t>(2048, (n_words + 255) / 256);
  hipLaunchKernelGGL(zero_words, dim3(grid), dim3(256), 0, stream, static_cast<unsigned long long*>(out), n_words);
  HIPC(hipGetLastError());
  return XHIST_OK;
}

static const void* advance(const void* base, int dt, int64_t elems) {
  return static_cast<const char*>(base) + elems * dtype_size(dt);
}

// Partitioned mode (xhist_partition.hip.h): count -> prefix -> scatter -> accumulate, all on
// `stream`, scratch from the stream-ordered allocator (so concurrent callers never share it).
static int execute_partitioned(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_cols, void* out,
                               hipStream_t stream, int sdt, int wdt, int scan, bool use_f32, const TableSet& tset, int shift,
                               int n_parts, int profile) {
  const int D = p->n_dims;
  const bool weighted = weights != nullptr;
  if (n_cols >= ((int64_t)1 << 40)) return XHIST_ERR_UNSUPPORTED;
  int vec = 1;
  kernel_fn_count k_count = (kernel_fn_count)fast_kernel(sdt, wdt, D, scan, kHistPartCount, &vec);
  if (!k_count) return XHIST_ERR_UNSUPPORTED;
  // records leave part_scatter in aligned groups: 8 (one 16-byte code store) while the carried
  // records of all partitions fit LDS next to the tile, else 4
  const int grp = n_parts <= 128 ? 8 : 4;
  kernel_fn_scatter k_scatter;
  if (grp == 8)
    k_scatter = wdt < 0 ? (kernel_fn_scatter)part_scatter<NoWeight, 8>
                        : (wdt == XHIST_F64 ? (kernel_fn_scatter)part_scatter<double, 8> : (kernel_fn_scatter)part_scatter<float, 8>);
  else
    k_scatter = wdt < 0 ? (kernel_fn_scatter)part_scatter<NoWeight, 4>
                        : (wdt == XHIST_F64 ? (kernel_fn_scatter)part_scatter<double, 4> : (kernel_fn_scatter)part_scatter<float, 4>);
  const int32_t table_words = scan == kScanArith ? 0 : tset.words;  // arithmetic edges: no tables
  const size_t table_bytes = (size_t)table_words * 8;
  const size_t lds_count = table_bytes + (size_t)(n_parts + 1) * 32 * 4;
  const size_t lds_scatter = part_scatter_lds(n_parts, grp, weighted);
  const size_t lds_acc = (size_t)((1u << shift) + 1) * (weighted ? 8 : 4);  // + the trash slot of padding records
  if (lds_count > p->lds_max || lds_scatter > p->lds_max || lds_acc > p->lds_max) return XHIST_ERR_UNSUPPORTED;
  const int per_cu = std::max<int>(1, std::min<int>(4, (int)(160 * 1024 / std::max(lds_count, lds_scatter))));
  const int64_t n_tiles = (n_cols + kPartTile - 1) / kPartTile;
  const int G = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)p->cus * per_cu, n_tiles));
  const int Gb = (int)std::max<int64_t>(1, std::min<int64_t>(p->cus, (n_cols + 65535) / 65536));

  uint32_t* d_counts = nullptr;
  uint64_t *d_base = nullptr, *d_offsets = nullptr;
  uint16_t* d_codes = nullptr;
  double* d_w = nullptr;
  uint32_t* d_flat = nullptr;
  auto release = [&](int rc) {
    if (d_flat) (void)hipFreeAsync(d_flat, stream);
    if (d_counts) (void)hipFreeAsync(d_counts, stream);
    if (d_base) (void)hipFreeAsync(d_base, stream);
    if (d_offsets) (void)hipFreeAsync(d_offsets, stream);
    if (d_codes) (void)hipFreeAsync(d_codes, stream);
    if (d_w) (void)hipFreeAsync(d_w, stream);
    return rc;
  };
#define HIPR(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return release(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  HIPR(hipMallocAsync((void**)&d_counts, (size_t)G * n_parts * 4, stream));
  HIPR(hipMallocAsync((void**)&d_base, (size_t)G * n_parts * 8, stream));
  HIPR(hipMallocAsync((void**)&d_offsets, (size_t)(n_parts + 1) * 8, stream));
  HIPR(hipMallocAsync((void**)&d_flat, (size_t)n_tiles * kPartTile * 4, stream));
  const size_t n_rec = (size_t)n_cols + (size_t)G * n_parts * grp;  // every slice rounded up to whole groups
  HIPR(hipMallocAsync((void**)&d_codes, n_rec * 2 + 16, stream));
  if (weighted) HIPR(hipMallocAsync((void**)&d_w, n_rec * 8 + 16, stream));

  Params kp;
  memset(&kp, 0, sizeof kp);
  const DimTable* dims = tset.dim;
  for (int d = 0; d < D; ++d) {
    kp.s_ptr[d] = samples[d].data;
    kp.s_rs[d] = samples[d].row_stride;
    kp.s_cs[d] = 1;
    kp.s_dt[d] = samples[d].dtype;
    kp.dim[d] = dims[d];
  }
  if (weighted) {
    kp.w_ptr = weights->data;
    kp.w_rs = weights->row_stride;
    kp.w_cs = 1;
    kp.w_dt = weights->dtype;
  }
  kp.n_dims = D;
  kp.tables = tset.blob;
  kp.table_words = table_words;
  kp.tables_in_lds = 1;
  kp.n_rows = 1;
  kp.n_cols = n_cols;
  kp.n_bins = p->n_bins;
  kp.out = out;
  kp.segs = G;
  kp.part_counts = d_counts;
  kp.part_base = d_base;
  kp.part_codes = d_codes;
  kp.part_w = d_w;
  kp.part_shift = shift;
  kp.n_parts = n_parts;

  if (lds_count > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count));
  if (lds_scatter > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scatter));
  kernel_fn_acc k_acc = weighted ? (kernel_fn_acc)part_accumulate<true> : (kernel_fn_acc)part_accumulate<false>;
  if (lds_acc > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc));

  LaunchRecord rec(p, stream);
  if (int rrc = rec.begin(profile)) return release(rrc);
  hipLaunchKernelGGL(k_count, dim3(G), dim3(kPartBlock), lds_count, stream, kp, d_flat);
  HIPR(hipGetLastError());
  hipLaunchKernelGGL(part_prefix, dim3(1), dim3(1024), 0, stream, (const uint32_t*)d_counts, G, n_parts, grp, d_offsets, d_base);
  HIPR(hipGetLastError());
  hipLaunchKernelGGL(k_scatter, dim3(G), dim3(kPartBlock), lds_scatter, stream, (const uint32_t*)d_flat,
                     weighted ? weights->data : nullptr, n_cols, (const uint64_t*)d_base, d_codes, d_w, shift, n_parts);
  HIPR(hipGetLastError());
  hipLaunchKernelGGL(k_acc, dim3(Gb), dim3(1024), lds_acc, stream, (const uint16_t*)d_codes, (const double*)d_w,
                     (const uint64_t*)d_offsets, out, p->n_bins, shift, n_parts);
  HIPR(hipGetLastError());
  {
    char desc[384];
    snprintf(desc, sizeof desc,
             "family=fast hist=partitioned parts=%d bins_per_part=%d group=%d vec=%d tile=%d block=%d grid=%d acc_grid=%d "
             "lds_count=%zu lds_scatter=%zu lds_acc=%zu scan=%d weighted=%d D=%d cmp=%s",
             n_parts, 1 << shift, grp, vec, kPartTile, kPartBlock, G, Gb, lds_count, lds_scatter, lds_acc, scan, (int)weighted, D,
             use_f32 ? "f32thr" : "f64");
    if (int rrc = rec.end(desc)) return release(rrc);
  }
#undef HIPR
  return release(XHIST_OK);
}

// Row-per-lane mode (xhist_lanes.hip.h).  Takes (a) views whose ROWS are the contiguous direction
// (row stride 1: reductions over leading axes) as they are, and (b) many short contiguous rows
// after transposing them into a [cols, rows] scratch.  Returns XHIST_ERR_UNSUPPORTED when the
// shape is better served by the row-streaming kernels.
static int execute_lanes(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows, int64_t n_cols,
                         void* out, int accumulate, hipStream_t stream, bool prefer, int profile) {
  const int D = p->n_dims;
  const bool weighted = weights != nullptr;
  if (p->cmp != XHIST_CMP_F64 || D > 3 || n_cols >= ((int64_t)1 << 31) || p->n_bins >= (1 << 16) || p->huge) return XHIST_ERR_UNSUPPORTED;
  const int sdt = samples[0].dtype, wdt = weighted ? weights->dtype : -1;
  if ((sdt != XHIST_F64 && sdt != XHIST_F32) || (wdt != -1 && wdt != XHIST_F64 && wdt != XHIST_F32)) return XHIST_ERR_UNSUPPORTED;
  // shape class of every array: natural (row stride 0/1, any column stride) or needs a transpose
  // (unit column stride, dense-ish rows)
  bool all_natural = true, all_rowmajor = true, grouped_any = false;
  for (int d = 0; d <= D; ++d) {
    if (d == D && !weighted) break;
    const xhist_array& a = d < D ? samples[d] : *weights;
    if (d < D && a.dtype != sdt) return XHIST_ERR_UNSUPPORTED;
    const bool bcast = a.row_stride == 0 || a.col_stride == 0;
    const bool natural = bcast || (a.row_stride == 1 && (a.inner_rows ? a.col_stride >= 1 : a.col_stride >= n_rows));
    const bool rowmajor = bcast || (a.col_stride == 1 && a.row_stride >= n_cols);
    grouped_any |= a.inner_rows != 0 && !bcast;
    all_natural &= natural;
    all_rowmajor &= rowmajor;
  }
  const bool use_f32 = sdt == XHIST_F32 && p->ts[1][0].blob != nullptr;
  int scan = 0;
  const TableSet& tset = pick_tables(p, use_f32, &scan);
  const size_t table_bytes = (size_t)tset.words * 8;
  const size_t lds_bytes = table_bytes + (size_t)p->n_bins * kLanePitch * (weighted ? 8 : 4);
  if (lds_bytes > p->lds_max) return XHIST_ERR_UNSUPPORTED;
  bool transpose = false;
  if (all_natural && (samples[0].row_stride == 1 || prefer)) {
    // rows are the contiguous direction: the row-streaming kernels cannot coalesce this at all
  } else if (all_rowmajor && (prefer || (n_rows >= 4096 && n_cols <= ((D == 1 && !weighted) ? 896 : 384)))) {
    // many short rows.  Measured crossovers with the row-streaming kernels at 64-thread workgroups
    // (profiles/r01_f_shapes.jsonl): ~900 columns for the fused kernel (one unweighted input),
    // ~400 for scratch-transpose + lanes
    transpose = true;
  } else {
    return XHIST_ERR_UNSUPPORTED;
  }

  int vec = 1;
  kernel_fn_lanes fn = (kernel_fn_lanes)fast_kernel(sdt, wdt, D, scan, kHistLanes, &vec);
  if (!fn) return XHIST_ERR_UNSUPPORTED;

  const bool fused_ok = D == 1 && !weighted && n_cols < 65536 && samples[0].col_stride == 1 && samples[0].row_stride != 0;
  if (transpose && grouped_any && !fused_ok) return XHIST_ERR_UNSUPPORTED;  // transpose_2d takes plain row strides only
  // one contiguous-row input, unweighted, < 65536 columns: fused load-transpose-count kernel
  if (transpose && D == 1 && !weighted && n_cols < 65536 && samples[0].col_stride == 1 && samples[0].row_stride != 0) {
    const int es = dtype_size(sdt);
    const size_t hist_bytes = (size_t)p->n_bins * (kLaneBlock / 2 + 1) * 4;
    const size_t lds_f = ((table_bytes + hist_bytes + 15) & ~(size_t)15) + (size_t)kLaneBlock * (128 / es + 1) * es;
    kernel_fn_rows1 f1 = rows1_kernel(sdt, scan);
    if (f1 && lds_f <= p->lds_max) {
      Params kp;
      memset(&kp, 0, sizeof kp);
      kp.s_ptr[0] = samples[0].data;
      kp.s_rs[0] = samples[0].row_stride;
      kp.s_cs[0] = 1;
      kp.s_ir[0] = samples[0].inner_rows;
      kp.s_os[0] = samples[0].outer_stride;
      kp.s_dt[0] = sdt;
      kp.dim[0] = tset.dim[0];
      kp.n_dims = 1;
      kp.tables = tset.blob;
      kp.table_words = tset.words;
      kp.tables_in_lds = 1;
      kp.n_rows = n_rows;
      kp.n_cols = n_cols;
      kp.n_bins = p->n_bins;
      kp.out = out;
      const int64_t row_blocks = (n_rows + kLaneBlock - 1) / kLaneBlock;
      if (row_blocks > 2147483647LL) return XHIST_ERR_UNSUPPORTED;
      const int direct = accumulate ? 0 : 1;
      LaunchRecord rec(p, stream);
      if (int rrc = rec.begin(profile)) return rrc;
      if (lds_f > 48 * 1024) HIPC(hipFuncSetAttribute((const void*)f1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      hipLaunchKernelGGL(f1, dim3((unsigned)row_blocks), dim3(kLaneBlock), lds_f, stream, kp, (int32_t)direct);
      HIPC(hipGetLastError());
      char desc[384];
      snprintf(desc, sizeof desc,
               "family=lanes hist=lds16 transpose=fused direct_store=%d block=%d grid=%lld lds_bytes=%zu scan=%d weighted=0 D=1 cmp=%s",
               direct, kLaneBlock, (long long)row_blocks, lds_f, scan, use_f32 ? "f32thr" : "f64");
      return rec.end(desc);
    }
  }

  void* scratch[kMaxDims + 1] = {nullptr};
  auto release = [&](int rc) {
    for (auto s : scratch)
      if (s) (void)hipFreeAsync(s, stream);
    return rc;
  };
#define HIPL(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return release(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)

  Params kp;
  memset(&kp, 0, sizeof kp);
  const DimTable* dims = tset.dim;
  LaunchRecord rec(p, stream);
  if (int rrc = rec.begin(profile)) return release(rrc);
  for (int d = 0; d <= D; ++d) {
    if (d == D && !weighted) break;
    const xhist_array& a = d < D ? samples[d] : *weights;
    const void* ptr = a.data;
    int64_t rs = a.row_stride, cs = a.col_stride, ir = a.inner_rows, os = a.outer_stride;
    if (transpose && rs != 0 && cs != 0) {
      ir = os = 0;
      const int es = dtype_size(a.dtype);
      HIPL(hipMallocAsync(&scratch[d], (size_t)n_rows * n_cols * es, stream));
      const dim3 grid((unsigned)((n_rows + 63) / 64), (unsigned)((n_cols + 63) / 64));
      if (es == 8)
        hipLaunchKernelGGL(transpose_2d<double>, grid, dim3(256), 0, stream, (const double*)a.data, rs, n_rows, n_cols, (double*)scratch[d]);
      else
        hipLaunchKernelGGL(transpose_2d<float>, grid, dim3(256), 0, stream, (const float*)a.data, rs, n_rows, n_cols, (float*)scratch[d]);
      HIPL(hipGetLastError());
      ptr = scratch[d];
      rs = 1;
      cs = n_rows;
    }
    if (d < D) {
      kp.s_ptr[d] = ptr;
      kp.s_rs[d] = rs;
      kp.s_cs[d] = cs;
      kp.s_ir[d] = ir;
      kp.s_os[d] = os;
      kp.s_dt[d] = a.dtype;
      kp.dim[d] = dims[d];
    } else {
      kp.w_ptr = ptr;
      kp.w_rs = rs;
      kp.w_cs = cs;
      kp.w_ir = ir;
      kp.w_os = os;
      kp.w_dt = a.dtype;
    }
  }
  kp.n_dims = D;
  kp.tables = tset.blob;
  kp.table_words = tset.words;
  kp.tables_in_lds = 1;
  kp.n_rows = n_rows;
  kp.n_cols = n_cols;
  kp.n_bins = p->n_bins;
  kp.out = out;

  const int64_t row_blocks = (n_rows + kLaneBlock - 1) / kLaneBlock;
  // unweighted and few enough columns per workgroup: uint16 counters, half the LDS
  size_t lds_use = lds_bytes;
  bool packed16 = false;
  if (!weighted) {
    const size_t lds16 = table_bytes + (size_t)p->n_bins * (kLaneBlock / 2 + 1) * 4;
    const int bpc16 = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)160 * 1024 / lds16));
    int64_t segs16 = std::max<int64_t>(1, ((int64_t)p->cus * bpc16 * 2 + row_blocks - 1) / row_blocks);
    segs16 = std::min<int64_t>(std::min<int64_t>(segs16, std::max<int64_t>(1, n_cols / 64)), 65535);
    if ((n_cols + segs16 - 1) / segs16 <= 65535) {
      kernel_fn_lanes f16 = (kernel_fn_lanes)fast_kernel(sdt, wdt, D, scan, kHistLanes16, &vec);
      if (f16) {
        fn = f16;
        packed16 = true;
        lds_use = lds16;
      }
    }
  }
  const int bpc = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)160 * 1024 / lds_use));
  int64_t col_segs = std::max<int64_t>(1, ((int64_t)p->cus * bpc * 2 + row_blocks - 1) / row_blocks);
  col_segs = std::min<int64_t>(col_segs, std::max<int64_t>(1, n_cols / 64));
  col_segs = std::min<int64_t>(col_segs, 65535);
  const int64_t cols_per_seg = (n_cols + col_segs - 1) / col_segs;
  col_segs = (n_cols + cols_per_seg - 1) / cols_per_seg;
  const int direct = (col_segs == 1 && !accumulate) ? 1 : 0;
  if (!direct && !accumulate)
    if (int zrc = zero_output(out, n_rows * p->n_bins, stream)) return release(zrc);
  if (row_blocks > 2147483647LL) return release(XHIST_ERR_UNSUPPORTED);
  if (lds_use > 48 * 1024) HIPL(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_use));
  hipLaunchKernelGGL(fn, dim3((unsigned)row_blocks, (unsigned)col_segs), dim3(kLaneBlock), lds_use, stream, kp, (int32_t)direct,
                     cols_per_seg);
  HIPL(hipGetLastError());
  {
    char desc[384];
    snprintf(desc, sizeof desc,
             "family=lanes hist=%s transpose=%d direct_store=%d block=%d grid=%lldx%lld lds_bytes=%zu scan=%d weighted=%d D=%d cmp=%s",
             packed16 ? "lds16" : "lds", (int)transpose, direct, kLaneBlock, (long long)row_blocks, (long long)col_segs, lds_use, scan,
             (int)weighted, D,
             use_f32 ? "f32thr" : "f64");
    if (int rrc = rec.end(desc)) return release(rrc);
  }
#undef HIPL
  return release(XHIST_OK);
}

static int execute_device(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows,
                          int64_t n_cols, void* out, int accumulate, hipStream_t stream) {
  const int D = p->n_dims;
  const bool weighted = weights != nullptr;
  const int64_t out_elems = n_rows * p->n_bins;
  if (out_elems == 0) return XHIST_OK;
  if (n_cols == 0) {
    if (!accumulate)
      if (int zrc = zero_output(out, out_elems, stream)) return zrc;
    return XHIST_OK;
  }

  int block_threads, grid_blocks, force_global, force_generic, lds_copies, profile, partition, lanes, arith_pref;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    block_threads = p->block_threads; grid_blocks = p->grid_blocks; force_global = p->force_global;
    force_generic = p->force_generic; lds_copies = p->lds_copies; profile = p->profile; partition = p->partition;
    lanes = p->lanes; arith_pref = p->arith_pref;
  }

  // ---- many short rows / leading-axis reductions: one row per lane (xhist_lanes.hip.h) --------
  if (lanes >= 0 && !force_generic && !force_global) {
    const int rc = execute_lanes(p, samples, weights, n_rows, n_cols, out, accumulate, stream, lanes > 0, profile);
    if (rc != XHIST_ERR_UNSUPPORTED) return rc;  // UNSUPPORTED = not this shape, fall through
  }
  if (!accumulate)
    if (int zrc = zero_output(out, out_elems, stream)) return zrc;

  // ---- family: fast (vector loads, homogeneous f64/f32) or generic --------------------------
  const size_t lds_cap = p->lds_max;
  const int sdt = samples[0].dtype;
  const int wdt = weighted ? weights->dtype : -1;
  int vec = 1;
  const bool float_samples = sdt == XHIST_F64 || sdt == XHIST_F32;
  const bool small_samples = sdt == XHIST_I32 || sdt == XHIST_I64 || sdt == XHIST_I16 || sdt == XHIST_U8 || sdt == XHIST_F16;
  bool fast_ok = !force_generic && p->cmp == XHIST_CMP_F64 && p->n_bins < ((int64_t)1 << 31) &&
                 ((float_samples && D <= 3 && (wdt == -1 || wdt == XHIST_F64 || wdt == XHIST_F32)) ||
                  (small_samples && D == 1 && (wdt == -1 || wdt == XHIST_F64)));
  if (fast_ok) {
    // unit column stride is all the vector family needs: gfx950 vector loads take any
    // element-aligned address (rows of 365 or 3650 samples stay on 16-byte loads)
    for (int d = 0; d < D && fast_ok; ++d) {
      const xhist_array& a = samples[d];
      fast_ok = a.dtype == sdt && a.col_stride == 1 && ((uintptr_t)a.data % (size_t)dtype_size(sdt) == 0);
    }
    if (fast_ok && weighted) fast_ok = weights->col_stride == 1 && ((uintptr_t)weights->data % (size_t)dtype_size(wdt) == 0);
  }

  // Two attempts: the vector family with its tables, then (if it has no kernel for this
  // combination, or its tables do not fit LDS) the generic family with the native tables.
  bool fast = false, use_f32 = false, tables_fit = false, lds_hist = false, tables_in_lds = false;
  int scan = 0, hist = kHistGlobal, cl2 = 0;
  const TableSet* tset = nullptr;
  size_t table_bytes = 0, hist_bytes = 0, lds_bytes = 0;
  kernel_fn fn = nullptr;
  const int acc_size = weighted ? 8 : 4;
  const int max_cl2 = weighted ? 4 : 5;
  // histogram placement for a given table footprint:
  //   lds:    replicated sub-histograms in LDS (one copy per lane bank), uint32 / float64
  //   packed: unweighted vector family only, uint16 counters packed two per word (exact, see kernel)
  //   global: device-scope atomics straight into the output
  auto place = [&](size_t tbytes, bool vector_family) {
    hist = kHistGlobal;
    cl2 = 0;
    hist_bytes = 0;
    if (!force_global && tbytes + 1024 <= lds_cap && p->n_bins < ((int64_t)1 << 24)) {
      const size_t soft = 24 * 1024;  // replication is only worth LDS that small workgroups can share
      cl2 = max_cl2;
      if (lds_copies) { cl2 = 0; while ((1 << cl2) < lds_copies) ++cl2; cl2 = std::min(cl2, max_cl2); }
      auto bytes_at = [&](int c) { return ((size_t)p->n_bins + 1) * ((size_t)acc_size << c); };
      if (!lds_copies) while (cl2 > 0 && bytes_at(cl2) > soft) --cl2;
      while (cl2 > 0 && tbytes + bytes_at(cl2) > lds_cap) --cl2;
      if (tbytes + bytes_at(cl2) <= lds_cap) {
        hist = kHistLds;
        hist_bytes = bytes_at(cl2);
      } else if (vector_family && float_samples && !weighted && tbytes + ((size_t)p->n_bins + 1) / 2 * 4 <= lds_cap) {
        hist = kHistPacked;
        cl2 = 0;
        hist_bytes = ((size_t)p->n_bins + 1) / 2 * 4;
      }
    }
    if (hist == kHistGlobal) { cl2 = 0; hist_bytes = 0; }
  };
  for (int attempt = fast_ok ? 0 : 1; attempt < 2 && !fn; ++attempt) {
    fast = attempt == 0;
    // float32 samples are digitized against the float32-threshold tables (exact, see Dom<2>)
    use_f32 = fast && sdt == XHIST_F32 && p->ts[1][0].blob != nullptr;
    scan = 0;
    tset = &p->ts[0][0];  // generic family: native domain, (start, cnt) tables
    if (fast) tset = &pick_tables(p, use_f32, &scan);
    table_bytes = (size_t)tset->words * 8;
    tables_fit = table_bytes + 1024 <= lds_cap && !(fast && p->huge);  // no bucket tables: not for the vector family
    if (tables_fit || !fast) place(table_bytes, fast);
    // Arithmetic edges (bins=int, np.linspace): when the edge tables are what keeps the histogram
    // out of LDS — or do not fit LDS at all — digitize without tables (count_le_arith): 30000
    // uniform bins stay on the streaming kernels instead of 43 ms/10^9 samples of global atomics.
    // Also when the tables fit but only with 3-4 edges per bucket (float32, 20000 bins: 1.17 against 1.39 ms);
    // with 1-2 edges per bucket the tables win (C2: 2.28 against 2.40 ms, float32 50 bins: 0.69 against 1.12).
    if (fast && float_samples && p->arith && arith_pref >= 0 &&
        (!tables_fit || hist == kHistGlobal || scan == 0 || scan >= 3 || arith_pref > 0)) {
      const int h0 = hist, c0 = cl2;
      const size_t b0 = hist_bytes;
      place(0, true);
      if (hist != kHistGlobal || !tables_fit || arith_pref > 0) {
        scan = kScanArith;
        use_f32 = false;
        tset = &p->ts[0][0];  // float64-domain DimTable (e_0, e_last, step); its tables are not read
        table_bytes = 0;
        tables_fit = true;
      } else {
        hist = h0; cl2 = c0; hist_bytes = b0;
      }
    }
    if (fast && !tables_fit) continue;  // the vector family keeps its tables in LDS
    lds_hist = hist == kHistLds;
    tables_in_lds = tables_fit;
    lds_bytes = (tables_in_lds ? table_bytes : 0) + hist_bytes;
    // (scan 1..4: linear in-bucket count, no bucket holds more than 4 edges — always for uniform bins)
    fn = fast ? fast_kernel(sdt, wdt, D, scan, hist, &vec) : generic_kernel(p->cmp, weighted, lds_hist);
  }
  if (!fn) return fail(XHIST_ERR_HIP, "internal: no kernel for this combination");
  if (!fast) vec = 1;
  const DimTable* dims = tset->dim;

  // ---- histograms beyond LDS: partitioned multi-pass instead of memory-side atomics ----------
  if (fast && float_samples && hist == kHistGlobal && !force_global && partition >= 0 && n_rows == 1) {
    const int shift = weighted ? 14 : 15;  // 2^14 float64 or 2^15 uint32 bins = 128 KiB of LDS
    const int64_t n_parts = (p->n_bins + ((int64_t)1 << shift) - 1) >> shift;
    const bool big_enough = n_cols >= ((int64_t)1 << 22) || (partition > 0 && n_cols >= 4);  // part_scatter reads whole weight quads
    if (n_parts <= kPartMaxParts && big_enough && (size_t)(1u << shift) * (weighted ? 8 : 4) + 1024 <= lds_cap) {
      const int rc = execute_partitioned(p, samples, weights, n_cols, out, stream, sdt, wdt, scan, use_f32, *tset, shift,
                                         (int)n_parts, profile);
      if (rc != XHIST_ERR_UNSUPPORTED) return rc;  // UNSUPPORTED = fall through to global atomics
    }
  }
  const int kUnroll = fast ? unroll_for(D, vec, scan) : 1;

  // ---- geometry -----------------------------------------------------------------------------
  // Workgroups per CU are sized by bytes in flight, not by occupancy: measured on MI355X
  // (profiles/r01_a_sweep.jsonl) the streaming rate peaks at ~64 KiB of outstanding loads per CU
  // (f64+weights: 2 x 256 threads x 128 B; f64: 4 x 256 x 64 B) and falls by 5-10% with more.
  // big LDS footprints leave room for one or two workgroups per CU; within-box sweeps: 512 threads
  // for the replicated/plain LDS histograms (2.43-2.47 ms against 2.50-2.55 at 1024 for 10^9 x 2
  // f64), 768 = three wavefronts per SIMD for the packed-uint16 one (C3: 2.45 against 2.54 at 1024,
  // 2.86 at 512)
  int block = block_threads ? block_threads : (lds_bytes > 40 * 1024 ? (hist == kHistPacked ? 768 : 512) : 256);
  if (!block_threads && n_rows > 1 && lds_bytes <= 40 * 1024) {
    // many rows, one workgroup each: a tile should be ~1/4 of the row or most of the workgroup
    // idles in the ragged tile (100k rows x 3650: 0.46 -> 0.39 ms; 356k x 1024: 1.3 -> 0.58 ms)
    const int64_t per_lane = fast ? (int64_t)vec * unroll_for(D, vec, scan) : 4;
    int64_t want = n_cols / (4 * per_lane);
    block = 64;
    while (block < 256 && block * 2 <= want) block *= 2;
  }
  int64_t lane_bytes = 0;
  for (int d = 0; d < D; ++d) lane_bytes += dtype_size(samples[d].dtype);
  if (weighted) lane_bytes += dtype_size(weights->dtype);
  lane_bytes *= fast ? (int64_t)vec * kUnroll : 4;
  int bpc = (int)std::max<int64_t>(1, std::min<int64_t>(8, (64 * 1024 + block * lane_bytes / 2) / (block * lane_bytes)));
  bpc = std::min<int>(bpc, 2048 / block);
  if (lds_bytes) bpc = std::max<int>(1, std::min<int64_t>(bpc, (int64_t)(160 * 1024 / lds_bytes)));
  // one row: the segs workgroups share the row's tiles round-robin, so exactly one resident wave
  // of workgroups is balanced by construction.  Many rows: a workgroup is tied to one row, so the
  // tail is balanced by making 8x more, smaller workgroups (C4 shape: 5.6 -> 6.5 TB/s)
  int64_t target = grid_blocks ? grid_blocks : (int64_t)p->cus * bpc * (n_rows > 1 ? 8 : 1);
  if (!grid_blocks && n_rows == 1) {
    // small inputs: every workgroup ends with one global atomic per non-empty bin, and atomics on
    // one address serialise at ~12 ns; streaming gains ~25 GB/s per workgroup.  The sum of the
    // two is minimal at sqrt(bytes / (25 GB/s * 12 ns)) workgroups (10^6 f64 samples: 18 -> 9 us)
    const double bytes = (double)n_cols * (double)(lane_bytes / (fast ? (int64_t)vec * kUnroll : 4));
    target = std::max<int64_t>(1, std::min<int64_t>(target, (int64_t)std::sqrt(bytes / 300.0)));
  }
  const int64_t tile = fast ? (int64_t)block * vec * kUnroll : (int64_t)block * 4;
  const int64_t tiles_per_row = (n_cols + tile - 1) / tile;
  if (lds_bytes > 48 * 1024) HIPC(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));

  // rows per launch bounded by the grid limit; columns per launch bounded so that no workgroup
  // can overflow a uint32 LDS counter (< 2^31 samples per workgroup per launch)
  const int64_t kMaxGrid = ((int64_t)1 << 31) - 1;
  int64_t col_chunk = n_cols;
  {
    int64_t segs_full = std::max<int64_t>(1, std::min<int64_t>(tiles_per_row, (target + n_rows - 1) / n_rows));
    const int64_t per_wg = ((tiles_per_row + segs_full - 1) / segs_full) * tile;
    if (per_wg >= ((int64_t)1 << 31)) col_chunk = segs_full * (((int64_t)1 << 30) / tile) * tile;
  }
  bool first_launch = true;
  LaunchRecord rec(p, stream);
  char desc[384];
  for (int64_t c0 = 0; c0 < n_cols; c0 += col_chunk) {
    const int64_t nc = std::min(col_chunk, n_cols - c0);
    const int64_t tpr = (nc + tile - 1) / tile;
    for (int64_t r0 = 0; r0 < n_rows;) {
      int64_t segs = std::max<int64_t>(1, std::min<int64_t>(tpr, (target + (n_rows - r0) - 1) / (n_rows - r0)));
      const int64_t nr = std::min<int64_t>(n_rows - r0, kMaxGrid / segs);
      Params kp;
      memset(&kp, 0, sizeof kp);
      for (int d = 0; d < D; ++d) {
        const xhist_array& a = samples[d];
        kp.s_ptr[d] = advance(a.data, a.dtype, c0 * a.col_stride);
        kp.s_rs[d] = a.row_stride;
        kp.s_cs[d] = a.col_stride;
        kp.s_ir[d] = a.inner_rows;
        kp.s_os[d] = a.outer_stride;
        kp.s_dt[d] = a.dtype;
        kp.dim[d] = dims[d];
      }
      if (weighted) {
        kp.w_ptr = advance(weights->data, weights->dtype, c0 * weights->col_stride);
        kp.w_rs = weights->row_stride;
        kp.w_cs = weights->col_stride;
        kp.w_ir = weights->inner_rows;
        kp.w_os = weights->outer_stride;
        kp.w_dt = weights->dtype;
      }
      kp.row0 = r0;
      kp.n_dims = D;
      kp.tables = tset->blob;
      kp.table_words = scan == kScanArith ? 0 : tset->words;  // arithmetic edges: nothing to stage
      kp.tables_in_lds = tables_in_lds ? 1 : 0;
      kp.n_rows = nr;
      kp.n_cols = nc;
      kp.n_bins = p->n_bins;
      kp.out = static_cast<char*>(out) + (size_t)r0 * p->n_bins * 8;
      kp.copies_log2 = cl2;
      kp.segs = (int32_t)segs;
      const dim3 grid((unsigned)(nr * segs));
      if (first_launch)
        if (int rrc = rec.begin(profile)) return rrc;
      hipLaunchKernelGGL(fn, grid, dim3(block), lds_bytes, stream, kp);
      HIPC(hipGetLastError());
      if (first_launch) {
        snprintf(desc, sizeof desc,
                 "family=%s hist=%s vec=%d unroll=%d block=%d grid=%lld segs=%lld lds_bytes=%zu copies=%d table_bytes=%zu "
                 "lut_k0=%d steps0=%d scan=%d weighted=%d D=%d cmp=%s lds_cap=%zu",
                 fast ? "fast" : "generic", hist == kHistLds ? "lds" : (hist == kHistPacked ? "packed16" : "global"),
                 fast ? vec : 1, fast ? kUnroll : 1, block, (long long)(nr * segs), (long long)segs, lds_bytes, 1 << cl2,
                 table_bytes, dims[0].lut_k, dims[0].steps, scan, (int)weighted, D,
                 use_f32 ? "f32thr" : (p->cmp == XHIST_CMP_I64 ? "i64" : "f64"), lds_cap);
      }
      first_launch = false;
      r0 += nr;
    }
  }
  return rec.end(desc);
}

// ------------------------------------------------------------------------------------------
// host-resident execute: stage chunks through device memory (PCIe-bound by construction)
// ------------------------------------------------------------------------------------------
struct Staged {
  void* dptr = nullptr;
  size_t cap = 0;
};

static int stage_chunk(const xhist_array& a, int64_t r0, int64_t nr, int64_t c0, int64_t nc, Staged& st, xhist_array* view,
                       hipStream_t stream) {
  const int es = dtype_size(a.dtype);
  if (a.row_stride == 1 && a.col_stride >= 1 && a.col_stride != 1) {
    // rows are the contiguous direction (a reduction over leading axes of a C-ordered array):
    // copy the [nc, nr] rectangle as it lies and hand the device the same transposed view — the
    // row-per-lane kernels take it; no host-side transposition
    const size_t need = (size_t)nr * nc * es;
    if (need > st.cap) {
      if (st.dptr) (void)hipFree(st.dptr);
      st.dptr = nullptr;
      st.cap = 0;
      HIPC(hipMalloc(&st.dptr, need));
      st.cap = need;
    }
    const char* src = static_cast<const char*>(a.data) + (r0 + c0 * a.col_stride) * es;
    if (a.col_stride == nr) {
      HIPC(hipMemcpyAsync(st.dptr, src, need, hipMemcpyHostToDevice, stream));
    } else {
      HIPC(hipMemcpy2DAsync(st.dptr, (size_t)nr * es, src, (size_t)a.col_stride * es, (size_t)nr * es, (size_t)nc,
                            hipMemcpyHostToDevice, stream));
    }
    view->data = st.dptr;
    view->dtype = a.dtype;
    view->reserved = 0;
    view->row_stride = 1;
    view->col_stride = nr;
    view->inner_rows = 0;
    view->outer_stride = 0;
    return XHIST_OK;
  }
  if (a.col_stride != 0 && a.col_stride != 1)
    return fail(XHIST_ERR_UNSUPPORTED, "host arrays need a unit row or column stride (got %lld, %lld); pass a contiguous copy",
                (long long)a.row_stride, (long long)a.col_stride);
  const int64_t rows = a.row_stride == 0 ? 1 : nr;
  const int64_t cols = a.col_stride == 0 ? 1 : nc;
  const size_t need = (size_t)rows * cols * es;
  if (need > st.cap) {
    if (st.dptr) (void)hipFree(st.dptr);
    st.dptr = nullptr;
    st.cap = 0;
    HIPC(hipMalloc(&st.dptr, need));
    st.cap = need;
  }
  const char* src = static_cast<const char*>(a.data) + ((a.row_stride ? r0 * a.row_stride : 0) + (a.col_stride ? c0 : 0)) * es;
  if (rows == 1 || a.row_stride == cols) {
    HIPC(hipMemcpyAsync(st.dptr, src, need, hipMemcpyHostToDevice, stream));
  } else {
    HIPC(hipMemcpy2DAsync(st.dptr, (size_t)cols * es, src, (size_t)a.row_stride * es, (size_t)cols * es, (size_t)rows,
                          hipMemcpyHostToDevice, stream));
  }
  view->data = st.dptr;
  view->dtype = a.dtype;
  view->reserved = 0;
  view->row_stride = a.row_stride == 0 ? 0 : cols;
  view->col_stride = a.col_stride == 0 ? 0 : 1;
  view->inner_rows = 0;
  view->outer_stride = 0;
  return XHIST_OK;
}

static int execute_host(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows, int64_t n_cols,
                        void* out, int accumulate, hipStream_t stream) {
  const int D = p->n_dims;
  const int64_t out_elems = n_rows * p->n_bins;
  if (out_elems == 0) return XHIST_OK;
  if (n_cols == 0) {
    if (!accumulate) memset(out, 0, (size_t)out_elems * 8);
    return XHIST_OK;
  }
  void* d_out = nullptr;
  Staged st[kMaxDims + 1];
  int rc = XHIST_OK;
  auto done = [&](int code) {
    for (auto& s : st)
      if (s.dptr) (void)hipFree(s.dptr);
    if (d_out) (void)hipFree(d_out);
    return code;
  };
  if (hipMalloc(&d_out, (size_t)out_elems * 8) != hipSuccess) return done(fail(XHIST_ERR_NOMEM, "hipMalloc of %lld output bytes failed", (long long)out_elems * 8));
  if (int zrc = zero_output(d_out, out_elems, stream)) return done(zrc);

  // views with grouped rows (reduced axes between kept axes) are staged whole, strides intact:
  // the bytes between the first and the last element are copied as they lie
  bool grouped = false;
  for (int d = 0; d < D; ++d) grouped |= samples[d].inner_rows != 0;
  if (weights) grouped |= weights->inner_rows != 0;
  if (grouped) {
    xhist_array views[kMaxDims];
    xhist_array wview;
    for (int d = 0; d <= D && rc == XHIST_OK; ++d) {
      if (d == D && !weights) break;
      const xhist_array& a = d < D ? samples[d] : *weights;
      const int es = dtype_size(a.dtype);
      const int64_t last_row = n_rows - 1;
      const int64_t roff = a.inner_rows ? (last_row / a.inner_rows) * a.outer_stride + (last_row % a.inner_rows) * a.row_stride
                                        : last_row * a.row_stride;
      const int64_t extent = roff + (n_cols - 1) * a.col_stride + 1;
      if (extent > ((int64_t)1 << 32)) { rc = fail(XHIST_ERR_UNSUPPORTED, "grouped host view spans more than 2^32 elements"); break; }
      Staged& s = st[d < D ? d : kMaxDims];
      if (hipMalloc(&s.dptr, (size_t)extent * es) != hipSuccess) { rc = fail(XHIST_ERR_NOMEM, "hipMalloc of a staging buffer failed"); break; }
      s.cap = (size_t)extent * es;
      if (hipMemcpyAsync(s.dptr, a.data, (size_t)extent * es, hipMemcpyHostToDevice, stream) != hipSuccess) {
        rc = fail(XHIST_ERR_HIP, "host to device copy failed");
        break;
      }
      xhist_array v = a;
      v.data = s.dptr;
      if (d < D) views[d] = v; else wview = v;
    }
    if (rc == XHIST_OK) rc = execute_device(p, views, weights ? &wview : nullptr, n_rows, n_cols, d_out, 1, stream);
  }
  // chunks of <= 2^27 elements per array: whole rows when a row fits, else column spans of one row
  const int64_t kChunk = (int64_t)1 << 27;
  const int64_t rows_per = n_cols <= kChunk ? std::max<int64_t>(1, kChunk / n_cols) : 1;
  const int64_t cols_per = n_cols <= kChunk ? n_cols : kChunk;
  for (int64_t r0 = 0; !grouped && r0 < n_rows && rc == XHIST_OK; r0 += rows_per) {
    const int64_t nr = std::min(rows_per, n_rows - r0);
    for (int64_t c0 = 0; c0 < n_cols && rc == XHIST_OK; c0 += cols_per) {
      const int64_t nc = std::min(cols_per, n_cols - c0);
      xhist_array views[kMaxDims];
      xhist_array wview;
      for (int d = 0; d < D && rc == XHIST_OK; ++d) rc = stage_chunk(samples[d], r0, nr, c0, nc, st[d], &views[d], stream);
      if (rc == XHIST_OK && weights) rc = stage_chunk(*weights, r0, nr, c0, nc, st[kMaxDims], &wview, stream);
      if (rc == XHIST_OK)
        rc = execute_device(p, views, weights ? &wview : nullptr, nr, nc, static_cast<char*>(d_out) + (size_t)r0 * p->n_bins * 8, 1, stream);
      // the staging buffers are reused by the next chunk: same-stream ordering makes that safe
    }
  }
  if (rc != XHIST_OK) { (void)hipStreamSynchronize(stream); return done(rc); }
  if (!accumulate) {
    if (hipMemcpyAsync(out, d_out, (size_t)out_elems * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return done(fail(XHIST_ERR_HIP, "copy of the result to the host failed: %s", hipGetErrorString(hipGetLastError())));
  } else {
    std::vector<uint64_t> tmp((size_t)out_elems);
    if (hipMemcpyAsync(tmp.data(), d_out, (size_t)out_elems * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return done(fail(XHIST_ERR_HIP, "copy of the result to the host failed: %s", hipGetErrorString(hipGetLastError())));
    if (weights) {
      double* o = static_cast<double*>(out);
      const double* t = reinterpret_cast<const double*>(tmp.data());
      for (int64_t i = 0; i < out_elems; ++i) o[i] += t[i];
    } else {
      int64_t* o = static_cast<int64_t*>(out);
      for (int64_t i = 0; i < out_elems; ++i) o[i] += (int64_t)tmp[(size_t)i];
    }
  }
  return done(XHIST_OK);
}

extern "C" int xhist_plan_execute(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows,
                                  int64_t n_cols, void* out, int out_dtype, int mem_kind, int accumulate, void* stream) {
  if (int rc = validate_arrays(p, samples, weights, n_rows, n_cols, out, out_dtype)) return rc;
  if (mem_kind != XHIST_MEM_HOST && mem_kind != XHIST_MEM_DEVICE) return fail(XHIST_ERR_INVALID, "unknown mem_kind %d", mem_kind);
  DeviceGuard g;
  if (int rc = g.set(p->device)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mem_kind == XHIST_MEM_DEVICE) return execute_device(p, samples, weights, n_rows, n_cols, out, accumulate, s);
  return execute_host(p, samples, weights, n_rows, n_cols, out, accumulate, s);
}

// ------------------------------------------------------------------------------------------
// one-shot form with a plan cache
// ------------------------------------------------------------------------------------------
static std::mutex g_cache_mu;
static std::map<std::string, xhist_plan*> g_cache;

static std::string cache_key(int device, int n_inputs, const void* const* edges, const int64_t* n_edges, int cmp) {
  std::string k;
  k.append(reinterpret_cast<const char*>(&device), sizeof device);
  k.append(reinterpret_cast<const char*>(&cmp), sizeof cmp);
  for (int d = 0; d < n_inputs; ++d) {
    k.append(reinterpret_cast<const char*>(&n_edges[d]), sizeof(int64_t));
    k.append(static_cast<const char*>(edges[d]), (size_t)n_edges[d] * 8);
  }
  return k;
}

extern "C" int xhist_bincount_rows(int device, int n_inputs, const xhist_array* samples, const xhist_array* weights,
                                   int64_t n_rows, int64_t n_cols, const void* const* edges, const int64_t* n_edges,
                                   int cmp_domain, void* out, int out_dtype, int mem_kind, int accumulate, void* stream) {
  if (n_inputs < 1 || n_inputs > XHIST_MAX_DIMS || !edges || !n_edges) return fail(XHIST_ERR_INVALID, "bad n_inputs / edges");
  for (int d = 0; d < n_inputs; ++d)
    if (!edges[d] || n_edges[d] < 1) return fail(XHIST_ERR_INVALID, "edges[%d] is NULL or empty", d);
  xhist_plan* plan = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    const std::string key = cache_key(device, n_inputs, edges, n_edges, cmp_domain);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) {
      plan = it->second;
    } else {
      if (int rc = xhist_plan_create(device, n_inputs, edges, n_edges, cmp_domain, &plan)) return rc;
      if (g_cache.size() >= 64) {  // bounded: drop everything rather than track recency
        for (auto& kv : g_cache) xhist_plan_destroy(kv.second);
        g_cache.clear();
      }
      g_cache[key] = plan;
    }
  }
  return xhist_plan_execute(plan, samples, weights, n_rows, n_cols, out, out_dtype, mem_kind, accumulate, stream);
}

extern "C" int xhist_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (auto& kv : g_cache) xhist_plan_destroy(kv.second);
  g_cache.clear();
  return XHIST_OK;
}

// ------------------------------------------------------------------------------------------
// min / max
// ------------------------------------------------------------------------------------------
extern "C" int xhist_minmax(int device, const xhist_array* a, int64_t n_rows, int64_t n_cols, double* result, int mem_kind,
                            void* stream) {
  if (!a || !result) return fail(XHIST_ERR_INVALID, "array / result is NULL");
  if (!dtype_size(a->dtype)) return fail(XHIST_ERR_INVALID, "unknown dtype tag %d", a->dtype);
  if (n_rows <= 0 || n_cols <= 0) return fail(XHIST_ERR_INVALID, "min/max of an empty array");
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available; this library has no CPU path", device);
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Staged st;
  xhist_array view = *a;
  if (mem_kind == XHIST_MEM_HOST) {
    if (n_rows * n_cols > ((int64_t)1 << 31)) return fail(XHIST_ERR_UNSUPPORTED, "host min/max above 2^31 elements: reduce on the host");
    if (int rc = stage_chunk(*a, 0, n_rows, 0, n_cols, st, &view, s)) { if (st.dptr) (void)hipFree(st.dptr); return rc; }
  }
  const int grid = 1024;
  double* d_part = nullptr;
  auto done = [&](int code) {
    if (d_part) (void)hipFree(d_part);
    if (st.dptr) (void)hipFree(st.dptr);
    return code;
  };
  if (hipMalloc(&d_part, sizeof(double) * 3 * grid) != hipSuccess) return done(fail(XHIST_ERR_NOMEM, "hipMalloc failed"));
  hipLaunchKernelGGL(minmax_kernel, dim3(grid), dim3(256), 0, s, view.data, view.dtype, view.row_stride, view.col_stride, view.inner_rows,
                     view.outer_stride, n_rows, n_cols, d_part);
  std::vector<double> part(3 * grid);
  if (hipGetLastError() != hipSuccess ||
      hipMemcpyAsync(part.data(), d_part, sizeof(double) * 3 * grid, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return done(fail(XHIST_ERR_HIP, "min/max kernel failed: %s", hipGetErrorString(hipGetLastError())));
  double mn = HUGE_VAL, mx = -HUGE_VAL;
  bool nan = false;
  for (int b = 0; b < grid; ++b) {
    mn = std::fmin(mn, part[3 * b]);
    mx = std::fmax(mx, part[3 * b + 1]);
    nan |= part[3 * b + 2] != 0.0;
  }
  result[0] = nan ? NAN : mn;
  result[1] = nan ? NAN : mx;
  return done(XHIST_OK);
}
