// xhist_capi.hip — host side of libxhist_amd.so: the C ABI declared in include/xhist_amd.h.
// Plans (device-resident edge tables), kernel-family selection, launch geometry, host staging.
// No CPU compute path exists here on purpose: without a HIP device every compute entry point
// fails with XHIST_ERR_NO_DEVICE.
#include "xhist_pick.hip.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>


#include "xhist_host_common.hip.h"
#include "xhist_plan.hip.h"
#include "xhist_select.hip.h"
#include "xhist_exec_device.hip.h"
#include "xhist_comm.hip.h"

// ------------------------------------------------------------------------------------------
// host-resident execute: stage chunks through device memory (PCIe-bound by construction)
// ------------------------------------------------------------------------------------------
struct Staged {
  void* dptr = nullptr;
  size_t cap = 0;
};

static void stage_free(Staged& st, hipStream_t stream, bool synced = false) {
  if (st.dptr) (void)scratch_free(st.dptr, stream, synced);
  st.dptr = nullptr;
  st.cap = 0;
}

static int stage_chunk(const xhist_array& a, int64_t r0, int64_t nr, int64_t c0, int64_t nc, Staged& st, xhist_array* view,
                       hipStream_t stream) {
  const int es = dtype_size(a.dtype);
  if (a.row_stride == 1 && a.col_stride >= 1 && a.col_stride != 1) {
    // rows are the contiguous direction (a reduction over leading axes of a C-ordered array):
    // copy the [nc, nr] rectangle as it lies and hand the device the same transposed view — the
    // row-per-lane kernels take it; no host-side transposition
    const size_t need = (size_t)nr * nc * es;
    if (need > st.cap) {
      stage_free(st, stream);
      HIPC(scratch_malloc(&st.dptr, need, stream));
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
    stage_free(st, stream);
    HIPC(scratch_malloc(&st.dptr, need, stream));
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

// device_out: `out` is a DEVICE buffer of the plan's GPU (XHIST_MEM_HOST_TO_DEVICE): the partial histogram stays where it
// was computed — for the sum over blocks / GPUs that follows (xhist_buffer_add, xhist_comm_allreduce) — and only the
// inputs cross PCIe
static int execute_host(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows, int64_t n_cols,
                        void* out, int accumulate, hipStream_t stream, bool device_out = false) {
  const int D = p->n_dims;
  const int64_t out_elems = n_rows * p->n_bins;
  if (out_elems == 0) return XHIST_OK;
  if (n_cols == 0) {
    if (!accumulate) {
      if (!device_out) memset(out, 0, (size_t)out_elems * 8);
      else if (int zrc = zero_output(out, out_elems, stream)) return zrc;
    }
    if (device_out) HIPC(hipStreamSynchronize(stream));
    return XHIST_OK;
  }
  void* d_out = nullptr;
  Staged st[kMaxDims + 1];
  int rc = XHIST_OK;
  bool synced = false;  // the stream has been waited for since the scratch was last used: no event needed to recycle it
  auto done = [&](int code) {
    for (auto& s : st) stage_free(s, stream, synced);
    if (d_out) (void)scratch_free(d_out, stream, synced);
    return code;
  };
  void* const user_out = out;
  if (device_out) {  // accumulate straight into the caller's device buffer
    if (!accumulate)
      if (int zrc = zero_output(out, out_elems, stream)) return done(zrc);
  } else {
    if (scratch_malloc(&d_out, (size_t)out_elems * 8, stream) != hipSuccess) return done(fail(XHIST_ERR_NOMEM, "hipMalloc of %lld output bytes failed", (long long)out_elems * 8));
    if (int zrc = zero_output(d_out, out_elems, stream)) return done(zrc);
  }
  void* const acc = device_out ? user_out : d_out;

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
      if (scratch_malloc(&s.dptr, (size_t)extent * es, stream) != hipSuccess) { rc = fail(XHIST_ERR_NOMEM, "allocation of a staging buffer failed"); break; }
      s.cap = (size_t)extent * es;
      if (hipMemcpyAsync(s.dptr, a.data, (size_t)extent * es, hipMemcpyHostToDevice, stream) != hipSuccess) {
        rc = fail(XHIST_ERR_HIP, "host to device copy failed");
        break;
      }
      xhist_array v = a;
      v.data = s.dptr;
      if (d < D) views[d] = v; else wview = v;
    }
    if (rc == XHIST_OK) rc = execute_device(p, views, weights ? &wview : nullptr, n_rows, n_cols, acc, 1, stream);
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
        rc = execute_device(p, views, weights ? &wview : nullptr, nr, nc, static_cast<char*>(acc) + (size_t)r0 * p->n_bins * 8, 1, stream);
      // the staging buffers are reused by the next chunk: same-stream ordering makes that safe
    }
  }
  if (rc != XHIST_OK) { synced = hipStreamSynchronize(stream) == hipSuccess; return done(rc); }
  if (device_out) {  // the result stays on the GPU; the staging copies of the caller's (pageable) inputs must be done
    if (hipStreamSynchronize(stream) != hipSuccess) return done(fail(XHIST_ERR_HIP, "stream synchronisation failed: %s", hipGetErrorString(hipGetLastError())));
    synced = true;
    return done(XHIST_OK);
  }
  if (!accumulate) {
    if (hipMemcpyAsync(out, d_out, (size_t)out_elems * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return done(fail(XHIST_ERR_HIP, "copy of the result to the host failed: %s", hipGetErrorString(hipGetLastError())));
    synced = true;
  } else {
    std::vector<uint64_t> tmp((size_t)out_elems);
    if (hipMemcpyAsync(tmp.data(), d_out, (size_t)out_elems * 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return done(fail(XHIST_ERR_HIP, "copy of the result to the host failed: %s", hipGetErrorString(hipGetLastError())));
    synced = true;
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
  if (mem_kind != XHIST_MEM_HOST && mem_kind != XHIST_MEM_DEVICE && mem_kind != XHIST_MEM_HOST_TO_DEVICE)
    return fail(XHIST_ERR_INVALID, "unknown mem_kind %d", mem_kind);
  DeviceGuard g;
  if (int rc = g.set(p->device)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mem_kind == XHIST_MEM_DEVICE) {
    Range r("xhist_plan_execute[device]");
    return execute_device(p, samples, weights, n_rows, n_cols, out, accumulate, s);
  }
  // Host calls are synchronous and self-contained.  With no stream given each calling thread gets its
  // own (hipStreamPerThread) instead of the device's one NULL stream: concurrent callers — dask's
  // threaded scheduler runs many blocks at once — then overlap one block's staging copy with another
  // block's kernel instead of queueing behind each other.
  Range r("xhist_plan_execute[host: stage + bin]");
  static const bool null_stream = [] { const char* e = getenv("XHIST_AMD_HOST_STREAM"); return e && !strcmp(e, "null"); }();
  return execute_host(p, samples, weights, n_rows, n_cols, out, accumulate, s ? s : (null_stream ? nullptr : hipStreamPerThread),
                      mem_kind == XHIST_MEM_HOST_TO_DEVICE);
}

extern "C" int xhist_plan_execute_two_weights(xhist_plan* p, const xhist_array* samples, const xhist_array* weights_a,
                                              const xhist_array* weights_b, int64_t n_rows, int64_t n_cols, void* out_a,
                                              void* out_b, int mem_kind, int accumulate, void* stream) {
  if (!weights_a || !weights_b) return fail(XHIST_ERR_INVALID, "two weight arrays are required");
  if (int rc = validate_arrays(p, samples, weights_a, n_rows, n_cols, out_a, XHIST_F64)) return rc;
  if (int rc = validate_arrays(p, samples, weights_b, n_rows, n_cols, out_b, XHIST_F64)) return rc;
  if (mem_kind != XHIST_MEM_HOST && mem_kind != XHIST_MEM_DEVICE) return fail(XHIST_ERR_INVALID, "unknown mem_kind %d", mem_kind);
  if (mem_kind == XHIST_MEM_DEVICE) {
    DeviceGuard g;
    if (int rc = g.set(p->device)) return rc;
    const int rc = execute_device(p, samples, weights_a, n_rows, n_cols, out_a, accumulate, static_cast<hipStream_t>(stream),
                                  weights_b, out_b);
    if (rc != XHIST_ERR_UNSUPPORTED) return rc;  // UNSUPPORTED: no fused kernel for this case, nothing written yet
  }
  if (int rc = xhist_plan_execute(p, samples, weights_a, n_rows, n_cols, out_a, XHIST_F64, mem_kind, accumulate, stream)) return rc;
  return xhist_plan_execute(p, samples, weights_b, n_rows, n_cols, out_b, XHIST_F64, mem_kind, accumulate, stream);
}

// ------------------------------------------------------------------------------------------
// one-shot form with a plan cache
// ------------------------------------------------------------------------------------------
// Cached plans are shared: the cache holds one reference, every call in flight holds another, so
// evicting (or xhist_shutdown) while another thread is still executing never frees a plan in use.
static std::mutex g_cache_mu;
static std::map<std::string, std::shared_ptr<xhist_plan>> g_cache;

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
  std::shared_ptr<xhist_plan> plan;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    const std::string key = cache_key(device, n_inputs, edges, n_edges, cmp_domain);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) {
      plan = it->second;
    } else {
      xhist_plan* raw = nullptr;
      if (int rc = xhist_plan_create(device, n_inputs, edges, n_edges, cmp_domain, &raw)) return rc;
      plan.reset(raw, [](xhist_plan* q) { (void)xhist_plan_destroy(q); });
      if (g_cache.size() >= 64) g_cache.clear();  // bounded: drop the cache's references rather than track recency
      g_cache[key] = plan;
    }
  }
  return xhist_plan_execute(plan.get(), samples, weights, n_rows, n_cols, out, out_dtype, mem_kind, accumulate, stream);
}

extern "C" int xhist_shutdown(void) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.clear();
  }
  trim_pools();  // cached scratch goes back to the driver
  return XHIST_OK;
}

// ------------------------------------------------------------------------------------------
// device buffers of the library: partial histograms that stay on their GPU between the kernel and the
// exchange (xhist_comm_*), for hosts that have no device allocator of their own
// ------------------------------------------------------------------------------------------
extern "C" int xhist_buffer_alloc(int device, size_t bytes, void** dptr) {
  if (!dptr) return fail(XHIST_ERR_INVALID, "dptr is NULL");
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available", device);
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  // from the library's caching allocator (stream-ordered on the NULL stream): a destination per strided copy, a partial
  // histogram per dask block — hipMalloc + hipFree cost 0.25 ms per 256 MB buffer, more than the copy kernel itself
  void* d = nullptr;
  if (scratch_malloc(&d, bytes ? bytes : 8, nullptr) != hipSuccess) {
    (void)hipGetLastError();
    return fail(XHIST_ERR_NOMEM, "device allocation of %zu bytes failed", bytes);
  }
  *dptr = d;
  return XHIST_OK;
}

extern "C" int xhist_buffer_free(int device, void* dptr) {
  if (!dptr) return XHIST_OK;
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  // (reused by another stream only once everything queued up to here — on the NULL stream, which waits for the blocking
  // streams before it — has completed)
  HIPC(scratch_free(dptr, nullptr));
  return XHIST_OK;
}

extern "C" int xhist_buffer_copy(int device, void* dst, const void* src, size_t bytes, int direction, void* stream) {
  if ((!dst || !src) && bytes) return fail(XHIST_ERR_INVALID, "dst / src is NULL");
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const hipMemcpyKind kind = direction == 0 ? hipMemcpyHostToDevice : direction == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (direction < 0 || direction > 2) return fail(XHIST_ERR_INVALID, "direction must be 0 (to device), 1 (to host) or 2 (on device)");
  if (bytes) HIPC(hipMemcpyAsync(dst, src, bytes, kind, s));
  if (direction != 2) HIPC(hipStreamSynchronize(s));  // host memory is involved: final when the call returns
  return XHIST_OK;
}

__global__ void __launch_bounds__(256) buffer_add_kernel(uint64_t* dst, const uint64_t* src, int64_t n, int is_f64) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (is_f64) reinterpret_cast<double*>(dst)[i] += reinterpret_cast<const double*>(src)[i];
    else dst[i] += src[i];
  }
}

extern "C" int xhist_buffer_add(int device, void* dst, const void* src, int64_t count, int dtype, void* stream) {
  if (dtype != XHIST_I64 && dtype != XHIST_F64) return fail(XHIST_ERR_INVALID, "partial histograms are int64 or float64");
  if (count < 0 || ((!dst || !src) && count)) return fail(XHIST_ERR_INVALID, "bad dst / src / count");
  if (!count) return XHIST_OK;
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  const int grid = (int)std::min<int64_t>((count + 255) / 256, 4096);
  hipLaunchKernelGGL(buffer_add_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<uint64_t*>(dst),
                     static_cast<const uint64_t*>(src), count, dtype == XHIST_F64 ? 1 : 0);
  HIPC(hipGetLastError());
  return XHIST_OK;
}

// ------------------------------------------------------------------------------------------
// strided N-D copies between device buffers (blocks of a device-resident dask array: slices, concatenations of
// unaligned chunks, the reference's moveaxis + reshape copy of core.py:218-226 where no three strides describe the
// block), optionally converting to float64 on the way (numpy's promotion inside searchsorted, core.py:170)
// ------------------------------------------------------------------------------------------
struct NdCopy {
  int32_t ndim;        // >= 1 after normalisation
  int64_t shape[8];
  int64_t ss[8];       // source strides, bytes (0 and negative allowed)
  int64_t ds[8];       // destination strides, bytes
};

template <int ITEM>  // bytes per element of a raw copy; 0: convert src_dt -> float64
__global__ void __launch_bounds__(256) copy_nd_kernel(const char* __restrict__ src, char* __restrict__ dst, NdCopy nd, int wlog2,
                                                      int64_t rows, int64_t col_tiles, int32_t src_dt) {
  // a workgroup covers (256 >> wlog2) rows x (16 << wlog2) elements of the innermost dimension per step
  const int last = nd.ndim - 1;
  const int64_t inner = nd.shape[last], ss_in = nd.ss[last], ds_in = nd.ds[last];
  const int rows_per_wg = 256 >> wlog2;
  const int64_t row_groups = (rows + rows_per_wg - 1) / rows_per_wg;
  for (int64_t t = blockIdx.x; t < row_groups * col_tiles; t += gridDim.x) {
    const int64_t rg = t / col_tiles, ct = t - rg * col_tiles;
    const int64_t row = rg * rows_per_wg + (threadIdx.x >> wlog2);
    if (row >= rows) continue;
    int64_t rem = row, so = 0, dof = 0;
    for (int k = last - 1; k >= 0; --k) {
      const int64_t q = rem / nd.shape[k], idx = rem - q * nd.shape[k];
      rem = q;
      so += idx * nd.ss[k];
      dof += idx * nd.ds[k];
    }
    const int64_t c0 = (ct << (wlog2 + 4)) + (threadIdx.x & ((1 << wlog2) - 1));
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int64_t c = c0 + ((int64_t)j << wlog2);
      if (c >= inner) break;
      const char* s = src + so + c * ss_in;
      char* d = dst + dof + c * ds_in;
      if constexpr (ITEM == 0) *reinterpret_cast<double*>(d) = load_as<double>(s, src_dt, 0);
      else if constexpr (ITEM == 16) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
      else if constexpr (ITEM == 8) *reinterpret_cast<uint64_t*>(d) = *reinterpret_cast<const uint64_t*>(s);
      else if constexpr (ITEM == 4) *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(s);
      else if constexpr (ITEM == 2) *reinterpret_cast<uint16_t*>(d) = *reinterpret_cast<const uint16_t*>(s);
      else *reinterpret_cast<uint8_t*>(d) = *reinterpret_cast<const uint8_t*>(s);
    }
  }
}

// The transposing case: the destination's fastest dimension (the last) is strided in the source, and another dimension `kd` is
// the source's fastest (stride of one to eight elements).  64 x 64 tiles of the (kd, last) plane go through LDS — reads coalesced along kd, writes along the last
// dimension; every other dimension is a batch index.  (The general kernel reads such layouts one element per cache line.)
template <typename T>
__global__ void __launch_bounds__(256) copy_nd_transpose(const char* __restrict__ src, char* __restrict__ dst, NdCopy nd, int kd,
                                                         int64_t tiles_k, int64_t tiles_l, int64_t n_tiles) {
  __shared__ T tile[64][65];
  const int last = nd.ndim - 1;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    int64_t rem = t;
    const int64_t tl = rem % tiles_l; rem /= tiles_l;
    const int64_t tk = rem % tiles_k; rem /= tiles_k;
    int64_t so = 0, dof = 0;
    for (int d = last - 1; d >= 0; --d) {
      if (d == kd) continue;
      const int64_t q = rem / nd.shape[d], idx = rem - q * nd.shape[d];
      rem = q;
      so += idx * nd.ss[d];
      dof += idx * nd.ds[d];
    }
    const int64_t k0 = tk * 64, l0 = tl * 64;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t ik = k0 + tx, il = l0 + ty + 4 * j;
      if (ik < nd.shape[kd] && il < nd.shape[last])
        tile[ty + 4 * j][tx] = *reinterpret_cast<const T*>(src + so + ik * nd.ss[kd] + il * nd.ss[last]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t ik = k0 + ty + 4 * j, il = l0 + tx;
      if (ik < nd.shape[kd] && il < nd.shape[last])
        *reinterpret_cast<T*>(dst + dof + ik * nd.ds[kd] + il * nd.ds[last]) = tile[tx][ty + 4 * j];
    }
    __syncthreads();
  }
}

extern "C" int xhist_buffer_copy_nd(int device, int ndim, const int64_t* shape, const void* src, int src_dtype,
                                    const int64_t* src_strides, void* dst, int dst_dtype, const int64_t* dst_strides, void* stream) {
  if (ndim < 0 || ndim > 8) return fail(XHIST_ERR_INVALID, "copy_nd takes 0..8 dimensions, got %d", ndim);
  if (ndim && (!shape || !src_strides || !dst_strides)) return fail(XHIST_ERR_INVALID, "shape / strides is NULL");
  const int item = dtype_size(src_dtype);
  if (!item || !dtype_size(dst_dtype)) return fail(XHIST_ERR_INVALID, "unknown dtype tag %d / %d", src_dtype, dst_dtype);
  const bool convert = dst_dtype != src_dtype;
  if (convert && dst_dtype != XHIST_F64) return fail(XHIST_ERR_UNSUPPORTED, "copy_nd converts to float64 only");
  int64_t n = 1;
  for (int k = 0; k < ndim; ++k) {
    if (shape[k] < 0) return fail(XHIST_ERR_INVALID, "negative extent");
    n *= shape[k];
  }
  if (n == 0) return XHIST_OK;
  if (!src || !dst) return fail(XHIST_ERR_INVALID, "src / dst is NULL");
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available", device);
  // drop extent-1 dimensions, merge neighbours that walk both sides like one dimension
  NdCopy nd{};
  for (int k = 0; k < ndim; ++k) {
    if (shape[k] == 1) continue;
    const int m = nd.ndim;
    if (m && nd.ss[m - 1] == src_strides[k] * shape[k] && nd.ds[m - 1] == dst_strides[k] * shape[k]) {
      nd.shape[m - 1] *= shape[k];
      nd.ss[m - 1] = src_strides[k];
      nd.ds[m - 1] = dst_strides[k];
    } else {
      nd.shape[m] = shape[k];
      nd.ss[m] = src_strides[k];
      nd.ds[m] = dst_strides[k];
      nd.ndim = m + 1;
    }
  }
  if (nd.ndim == 0) { nd.ndim = 1; nd.shape[0] = 1; nd.ss[0] = 0; nd.ds[0] = 0; }
  // a raw copy whose innermost dimension is contiguous on both sides moves wider elements where everything is aligned for it
  int item_eff = item;
  if (!convert && nd.ss[nd.ndim - 1] == item && nd.ds[nd.ndim - 1] == item) {
    for (int wide = 16; wide > item; wide >>= 1) {
      bool ok = (nd.shape[nd.ndim - 1] * item) % wide == 0 && (reinterpret_cast<uintptr_t>(src) % wide) == 0 && (reinterpret_cast<uintptr_t>(dst) % wide) == 0;
      for (int k = 0; ok && k < nd.ndim - 1; ++k) ok = nd.ss[k] % wide == 0 && nd.ds[k] % wide == 0;
      if (!ok) continue;
      nd.shape[nd.ndim - 1] = nd.shape[nd.ndim - 1] * item / wide;
      nd.ss[nd.ndim - 1] = nd.ds[nd.ndim - 1] = wide;
      item_eff = wide;
      n = n * item / wide;
      break;
    }
  }
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const char* sp = static_cast<const char*>(src);
  char* dp = static_cast<char*>(dst);
  if (!convert && nd.ndim >= 2 && nd.ds[nd.ndim - 1] == item && nd.ss[nd.ndim - 1] != item && nd.shape[nd.ndim - 1] >= 16) {
    int kd = -1;  // the source's fastest dimension: smallest positive stride, below the last dimension's (x.T[::2]: 2 elements)
    for (int k = 0; k < nd.ndim - 1; ++k)
      if (nd.ss[k] > 0 && nd.shape[k] >= 16 && (kd < 0 || nd.ss[k] < nd.ss[kd])) kd = k;
    const int64_t ss_last = nd.ss[nd.ndim - 1] < 0 ? -nd.ss[nd.ndim - 1] : nd.ss[nd.ndim - 1];
    if (kd >= 0 && !(nd.ss[kd] < ss_last && nd.ss[kd] <= 8 * (int64_t)item)) kd = -1;
    if (kd >= 0) {
      const int64_t tiles_k = (nd.shape[kd] + 63) / 64, tiles_l = (nd.shape[nd.ndim - 1] + 63) / 64;
      int64_t n_tiles = tiles_k * tiles_l;
      for (int k = 0; k < nd.ndim - 1; ++k)
        if (k != kd) n_tiles *= nd.shape[k];
      const int tgrid = (int)std::min<int64_t>(n_tiles, 256 * 16);
#define XHIST_COPY_T(T) hipLaunchKernelGGL((copy_nd_transpose<T>), dim3(tgrid), dim3(256), 0, s, sp, dp, nd, kd, tiles_k, tiles_l, n_tiles)
      if (item == 8) XHIST_COPY_T(uint64_t);
      else if (item == 4) XHIST_COPY_T(uint32_t);
      else if (item == 2) XHIST_COPY_T(uint16_t);
      else XHIST_COPY_T(uint8_t);
#undef XHIST_COPY_T
      HIPC(hipGetLastError());
      return XHIST_OK;
    }
  }
  const int64_t inner = nd.shape[nd.ndim - 1], rows = n / inner;
  int wlog2 = 0;
  while (wlog2 < 8 && ((int64_t)1 << wlog2) < inner) ++wlog2;
  const int64_t col_tiles = (inner + ((int64_t)16 << wlog2) - 1) / ((int64_t)16 << wlog2);
  const int rows_per_wg = 256 >> wlog2;
  const int64_t tiles = ((rows + rows_per_wg - 1) / rows_per_wg) * col_tiles;
  const int grid = (int)std::min<int64_t>(tiles, 256 * 32);
#define XHIST_COPY_ND(ITEM) hipLaunchKernelGGL((copy_nd_kernel<ITEM>), dim3(grid), dim3(256), 0, s, sp, dp, nd, wlog2, rows, col_tiles, (int32_t)src_dtype)
  if (convert) XHIST_COPY_ND(0);
  else if (item_eff == 16) XHIST_COPY_ND(16);
  else if (item_eff == 8) XHIST_COPY_ND(8);
  else if (item_eff == 4) XHIST_COPY_ND(4);
  else if (item_eff == 2) XHIST_COPY_ND(2);
  else XHIST_COPY_ND(1);
#undef XHIST_COPY_ND
  HIPC(hipGetLastError());
  return XHIST_OK;
}

extern "C" int xhist_pointer_device(const void* ptr, int* device) {
  if (!ptr || !device) return fail(XHIST_ERR_INVALID, "ptr / device is NULL");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, ptr) != hipSuccess || attr.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    return fail(XHIST_ERR_INVALID, "%p is not device memory of this process", ptr);
  }
  *device = logical_device(attr.device);
  return XHIST_OK;
}

// ------------------------------------------------------------------------------------------
// moments (bin-width estimators)
// ------------------------------------------------------------------------------------------
extern "C" int xhist_moments(int device, const xhist_array* a, int64_t n_rows, int64_t n_cols, int use_range, double lo, double hi,
                             int want_m2, double* result, int mem_kind, void* stream) {
  if (!a || !result) return fail(XHIST_ERR_INVALID, "array / result is NULL");
  if (!dtype_size(a->dtype)) return fail(XHIST_ERR_INVALID, "unknown dtype tag %d", a->dtype);
  if (n_rows <= 0 || n_cols <= 0) return fail(XHIST_ERR_INVALID, "moments of an empty array");
  if (mem_kind != XHIST_MEM_DEVICE) return fail(XHIST_ERR_UNSUPPORTED, "xhist_moments takes device-resident arrays (host arrays: numpy has them already)");
  if (device < 0 || device >= n_devices()) return fail(XHIST_ERR_NO_DEVICE, "HIP device %d not available; this library has no CPU path", device);
  DeviceGuard g;
  if (int rc = g.set(device)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = 1024;
  double* d_part = nullptr;
  auto done = [&](int code) {
    if (d_part) (void)scratch_free(d_part, s);
    return code;
  };
  if (scratch_malloc((void**)&d_part, sizeof(double) * 5 * grid, s) != hipSuccess) return done(fail(XHIST_ERR_NOMEM, "device allocation failed"));
  std::vector<double> part(5 * grid);
  auto pass = [&](int which, double mean) -> int {
    hipLaunchKernelGGL(moments_kernel, dim3(grid), dim3(256), 0, s, a->data, a->dtype, a->row_stride, a->col_stride, a->inner_rows, a->outer_stride,
                       n_rows, n_cols, use_range, lo, hi, which, mean, d_part);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(part.data(), d_part, sizeof(double) * 5 * grid, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return fail(XHIST_ERR_HIP, "moments kernel failed: %s", hipGetErrorString(hipGetLastError()));
    return XHIST_OK;
  };
  if (int rc = pass(0, 0.0)) return done(rc);
  double cnt = 0.0, mn = HUGE_VAL, mx = -HUGE_VAL, sum = 0.0;
  bool nan = false;
  for (int b = 0; b < grid; ++b) {
    cnt += part[5 * b];
    mn = std::fmin(mn, part[5 * b + 1]);
    mx = std::fmax(mx, part[5 * b + 2]);
    sum += part[5 * b + 3];
    nan |= part[5 * b + 4] != 0.0;
  }
  result[0] = cnt;
  result[1] = nan ? NAN : mn;
  result[2] = nan ? NAN : mx;
  result[3] = cnt > 0.0 ? sum / cnt : NAN;
  result[4] = NAN;
  if (want_m2 && cnt > 0.0 && !nan) {
    if (int rc = pass(1, result[3])) return done(rc);
    double m2 = 0.0;
    for (int b = 0; b < grid; ++b) m2 += part[5 * b];
    result[4] = m2;
  }
  return done(XHIST_OK);
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
    if (int rc = stage_chunk(*a, 0, n_rows, 0, n_cols, st, &view, s)) { stage_free(st, s); return rc; }
  }
  const int grid = 1024;
  double* d_part = nullptr;
  auto done = [&](int code) {
    if (d_part) (void)scratch_free(d_part, s);
    stage_free(st, s);
    return code;
  };
  if (scratch_malloc((void**)&d_part, sizeof(double) * 3 * grid, s) != hipSuccess) return done(fail(XHIST_ERR_NOMEM, "device allocation failed"));
  // contiguous float data (the usual `bins=int` on a whole array): the vectorised kernel
  const bool flat = (view.dtype == XHIST_F64 || view.dtype == XHIST_F32) && view.inner_rows == 0 && (n_cols == 1 || view.col_stride == 1) &&
                    (n_rows == 1 || view.row_stride == n_cols) && ((uintptr_t)view.data % (size_t)dtype_size(view.dtype)) == 0;
  if (flat)  // (`bins=int` on a whole float array is a first call's shape too: these two live in the hot code object)
    (void)xhist_hot_minmax_flat(view.dtype == XHIST_F64, view.data, n_rows * n_cols, d_part, grid, s);
  else
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
