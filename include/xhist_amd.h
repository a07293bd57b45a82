/*
 * xhist_amd.h — C ABI of the MI355X-native xhistogram hot path (libxhist_amd.so).
 *
 * The reference (xgcm/xhistogram) is pure Python and has NO FFI; its hot path sits behind plain
 * Python calls.  This header is the boundary a maintainer would bind (ctypes) in place of the body
 * of `_bincount_2d_vectorized` — see INTEGRATION.md for the stub.  Every entry point cites the
 * reference lines whose work it replaces (paths relative to /root/reference).
 *
 * Contract of the path (xhistogram/core.py:137-194):
 *   D sample arrays of identical logical shape [M rows, C cols], D edge arrays (sorted, length
 *   E_d >= 1), optional weights [M, C]  ->  out[M, nb_0, ..., nb_{D-1}],  nb_d = E_d - 1.
 *   bin k of a dimension holds  edges[k] <= x < edges[k+1];  the last bin also holds
 *   x == edges[E-1];  NaN, x < edges[0], x > edges[E-1] drop the sample (in ANY dimension).
 *   Unweighted -> exact int64 counts.  Weighted -> float64 sums of weights (cast to f64 first).
 *   The first input is the slowest-varying bin axis (C order), as `ravel_multi_index` at
 *   core.py:178-181.  Comparisons are made in float64 (XHIST_CMP_F64: every sample is converted
 *   to double first, which is what numpy's searchsorted does for f32/int data against f64 edges)
 *   or exactly in int64 (XHIST_CMP_I64: integer / datetime64 data against integer edges).
 *
 * Ownership: the caller owns every input and output buffer; the library owns plans and scratch.
 * Threading: all entry points are thread-safe; one plan may be executed from many threads.
 * Errors: functions return XHIST_OK (0) or a negative xhist_status; xhist_last_error() returns a
 * thread-local description.  The library never aborts and never falls back to a CPU path: without
 * a usable HIP device every compute call fails with XHIST_ERR_NO_DEVICE.
 */
#ifndef XHIST_AMD_H
#define XHIST_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XHIST_ABI_VERSION 9
#define XHIST_MAX_DIMS 8 /* max number of sample arrays (histogram dimensionality) */

typedef enum {
  XHIST_OK = 0,
  XHIST_ERR_INVALID = -1,     /* bad argument: null pointer, bad dtype tag, sizes < 0, D out of range */
  XHIST_ERR_UNSUPPORTED = -2, /* legal for the reference but not for this build (documented)        */
  XHIST_ERR_NO_DEVICE = -3,   /* no HIP device visible / device index out of range                   */
  XHIST_ERR_HIP = -4,         /* HIP runtime error (message in xhist_last_error)                      */
  XHIST_ERR_NOMEM = -5,       /* host or device allocation failed                                     */
  XHIST_ERR_EDGES = -6,       /* edges decrease somewhere or contain NaN (numpy: ValueError)          */
  XHIST_ERR_COMM = -7         /* RCCL not loadable / a collective failed (message in xhist_last_error) */
} xhist_status;

/* element type tags (numpy dtypes the reference accepts for samples / weights) */
typedef enum {
  XHIST_F64 = 0, XHIST_F32 = 1, XHIST_F16 = 2,
  XHIST_I64 = 3, XHIST_I32 = 4, XHIST_I16 = 5, XHIST_I8 = 6,
  XHIST_U64 = 7, XHIST_U32 = 8, XHIST_U16 = 9, XHIST_U8 = 10,
  XHIST_BOOL = 11
} xhist_dtype;

typedef enum { XHIST_CMP_F64 = 0, XHIST_CMP_I64 = 1 } xhist_cmp_domain;
/* Per-input domains (a datetime64 axis next to a float axis): cmp_domain = XHIST_CMP_PER_DIM | mask,
 * bit d of mask set <=> input d compares in int64 (its edge array is int64), else in float64.
 * numpy digitizes every argument on its own (core.py:163-174), so such mixtures are legal there. */
#define XHIST_CMP_PER_DIM 0x100
/* OR-ed onto XHIST_CMP_I64 / XHIST_CMP_PER_DIM: the int64-domain inputs are UNSIGNED 64-bit — their
 * edge arrays hold uint64 values and their samples (uint8..uint64, bool) compare as unsigned
 * (numpy: uint64 data against uint64 edges). */
#define XHIST_CMP_UNSIGNED 0x200
/* XHIST_MEM_HOST_TO_DEVICE: samples / weights are HOST pointers (staged by the library), `out` is a DEVICE buffer on the plan's GPU:
 * the partial histogram of a host-resident block stays on the GPU that computed it, for the sum over blocks and GPUs
 * (xhist_buffer_add, xhist_comm_allreduce) that replaces dask's `.sum(drop_axes)` (core.py:439) */
typedef enum { XHIST_MEM_HOST = 0, XHIST_MEM_DEVICE = 1, XHIST_MEM_HOST_TO_DEVICE = 2 } xhist_mem_kind;

/* A logical [M, C] array addressed as data[row_offset(r) + c * col_stride] (strides in ELEMENTS):
 *   row_offset(r) = r * row_stride                                                  if inner_rows == 0
 *                 = (r / inner_rows) * outer_stride + (r % inner_rows) * row_stride  otherwise.
 * row_stride == 0 or col_stride == 0 express numpy broadcasting without materialising it
 * (core.py:366 broadcast_arrays).  The grouped form expresses an N-D array whose reduced axes
 * sit BETWEEN kept axes — (time, LAT, lon) histogrammed over lat: rows = (time, lon) pairs,
 * inner_rows = n_lon, outer_stride = n_lat * n_lon — so that the reference's moveaxis + reshape
 * copy (core.py:211-229) is never made. */
typedef struct {
  const void* data;
  int32_t dtype; /* xhist_dtype */
  int32_t reserved;
  int64_t row_stride;
  int64_t col_stride;
  int64_t inner_rows;   /* rows per group; 0 = ungrouped */
  int64_t outer_stride; /* stride between groups of rows */
} xhist_array;

typedef struct xhist_plan xhist_plan; /* opaque: device-resident edge tables + launch geometry */

/* ---- library / device queries ------------------------------------------------------------ */
int xhist_abi_version(void);
const char* xhist_last_error(void);
int xhist_device_count(int* count);
/* name (may be NULL) receives the gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
int xhist_device_info(int device, char* name, size_t name_cap, int* compute_units, size_t* total_mem_bytes);

/* ---- plans ------------------------------------------------------------------------------- */
/* Upload D edge arrays (HOST pointers; float64 for XHIST_CMP_F64, int64 for XHIST_CMP_I64, per input
 * for XHIST_CMP_PER_DIM) and
 * build the per-dimension bucket->edge-range tables used by the branch-free digitize.
 * Replaces the per-call edge handling of core.py:154-155, 163-174. */
int xhist_plan_create(int device, int n_inputs, const void* const* edges, const int64_t* n_edges,
                      int cmp_domain, xhist_plan** plan);
int xhist_plan_destroy(xhist_plan* plan);

/* The fused hot path — replaces core.py:137-194 (_bincount_2d_vectorized: searchsorted 170-173,
 * ravel_multi_index 178-181, _dispatch_bincount/_bincount_2d 73-134, trim 189-192) in ONE kernel.
 *   samples[n_inputs], weights (NULL = unweighted): xhist_array views, all HOST or all DEVICE (mem_kind).
 *   out: contiguous [n_rows, prod(nb_d)];  out_dtype XHIST_I64 (unweighted) or XHIST_F64 (weighted).
 *   accumulate != 0: add into `out` instead of overwriting it — this is the reference's
 *     "sum over blocks" (dask `.sum(drop_axes)`, core.py:439) fused into the kernel's flush.
 *   stream: hipStream_t (NULL = default stream).  XHIST_MEM_DEVICE calls are asynchronous on
 *     `stream`; XHIST_MEM_HOST calls stage through device memory and return when `out` is final
 *     (with stream NULL they run on the calling thread's own stream, so concurrent host callers —
 *     dask's threaded scheduler — overlap one block's staging copy with another block's kernel).
 * Named roctx ranges wrap plan creation, execute and the exchange calls when a roctx library is
 * mapped in the process (rocprofv3 --marker-trace) or XHIST_AMD_ROCTX=1.
 */
int xhist_plan_execute(xhist_plan* plan, const xhist_array* samples, const xhist_array* weights,
                       int64_t n_rows, int64_t n_cols, void* out, int out_dtype, int mem_kind,
                       int accumulate, void* stream);

/* Two weight arrays binned in ONE pass over the samples: out_a = histogram weighted by weights_a,
 * out_b by weights_b (both float64 [n_rows, prod(nb_d)]).  The idiom behind it is the ratio of two
 * histograms, "mean of A in the bins of x" = hist(x, weights=A*w) / hist(x, weights=w): the
 * reference runs the whole path twice (its own TODO, xarray.py:106); here the samples are read and
 * digitized once when the arrays are device-resident and the vector kernels apply, and the call
 * degrades to two xhist_plan_execute passes otherwise — same results either way. */
int xhist_plan_execute_two_weights(xhist_plan* plan, const xhist_array* samples, const xhist_array* weights_a,
                                   const xhist_array* weights_b, int64_t n_rows, int64_t n_cols, void* out_a,
                                   void* out_b, int mem_kind, int accumulate, void* stream);

/* One-shot form of the two calls above with an internal plan cache keyed on (device, edges). */
int xhist_bincount_rows(int device, int n_inputs, const xhist_array* samples,
                        const xhist_array* weights, int64_t n_rows, int64_t n_cols,
                        const void* const* edges, const int64_t* n_edges, int cmp_domain, void* out,
                        int out_dtype, int mem_kind, int accumulate, void* stream);

/* min / max over a logical [M, C] array, NaN-propagating like numpy's a.min()/a.max(): feeds
 * np.histogram_bin_edges' (first_edge, last_edge) when bins is an int and range is None
 * (core.py:383-388).  result[0] = min, result[1] = max, as float64 (HOST pointer). */
int xhist_minmax(int device, const xhist_array* a, int64_t n_rows, int64_t n_cols, double* result,
                 int mem_kind, void* stream);

/* count / min / max / mean / sum of squared deviations of the elements of a DEVICE-resident [M, C] array that lie in
 * [lo, hi] (use_range != 0; NaN never does) or of all of them: what numpy's bin-width estimators need of the data —
 * np.histogram_bin_edges with bins = "sqrt" | "sturges" | "rice" | "scott" (core.py:383-388 hands it the whole array,
 * which for a GPU-resident array would mean a copy to the host).  result (HOST, 5 doubles) = {count, min, max, mean, M2};
 * min / max are NaN when a counted element is NaN (numpy then rejects the range); M2 only when want_m2 != 0 (a second
 * pass over the data with the mean of the first).  Synchronises `stream`. */
int xhist_moments(int device, const xhist_array* a, int64_t n_rows, int64_t n_cols, int use_range, double lo, double hi,
                  int want_m2, double* result, int mem_kind, void* stream);

/* ---- exchange between GPUs (one process per GPU; RCCL over xGMI) ---------------------------- */
/* Sharded inputs produce one partial histogram per GPU; what the reference does with dask's
 * `bin_counts.sum(drop_axes)` (core.py:439) is ONE in-place all-reduce of that small buffer
 * (shards cut along a reduced axis) or an all-gather of rows (shards cut along a kept axis:
 * disjoint output rows).  These calls wrap RCCL, which is dlopen-ed on first use (an RCCL the
 * process has mapped already is shared; XHIST_AMD_RCCL overrides the path), so single-GPU users and
 * hosts that bring their own collective never load it.
 *   rank 0: xhist_comm_unique_id -> hand the 128 bytes to every rank by any out-of-band means
 *   (the host's launcher, a file, MPI) -> every rank: xhist_comm_create (collective).
 * Buffers are DEVICE pointers on the comm's device; calls are asynchronous on `stream` and must be
 * issued in the same order on every rank.  int64 sums are exact and order-independent. */
#define XHIST_COMM_ID_BYTES 128
typedef enum { XHIST_REDUCE_SUM = 0, XHIST_REDUCE_MIN = 1, XHIST_REDUCE_MAX = 2 } xhist_reduce_op;
typedef struct xhist_comm xhist_comm; /* opaque: one RCCL communicator bound to one device */
int xhist_comm_unique_id(void* id, size_t cap);
int xhist_comm_create(int device, int rank, int world_size, const void* id, size_t id_bytes, xhist_comm** out);
/* any of rank / world_size / device / rccl_version may be NULL */
int xhist_comm_info(const xhist_comm* comm, int* rank, int* world_size, int* device, int* rccl_version);
/* in place; dtype XHIST_I64 (counts), XHIST_F64 or XHIST_F32 (weighted sums; min / max of the data
 * for bins=int: the reference's np.histogram_bin_edges sees the whole array, core.py:383-388) */
int xhist_comm_allreduce(xhist_comm* comm, void* buf, int64_t count, int dtype, int op, void* stream);
/* recv holds world_size * count elements, rank r's block at offset r * count */
int xhist_comm_allgather(xhist_comm* comm, const void* send, void* recv, int64_t count, int dtype, void* stream);
/* Deadlines: xhist_comm_create (the rendezvous) and xhist_comm_wait (the completion of collectives) do not wait for a
 * peer for ever: XHIST_AMD_COMM_TIMEOUT_S seconds (environment, default 300; <= 0: no deadline;
 * XHIST_AMD_COMM_CREATE_TIMEOUT_S, when set, overrides it for the rendezvous alone).  On expiry, or on an
 * asynchronous RCCL error, the communicator is aborted (ncclCommAbort: kernels of a collective in flight return), the call
 * returns XHIST_ERR_COMM with a message naming rank, world size and what was waited for, and every later call on the
 * communicator returns XHIST_ERR_COMM at once; xhist_comm_destroy is still due.
 * xhist_comm_wait: block until everything enqueued on `stream` has completed — the host-side synchronisation point of
 * an exchange, in place of a bare hipStreamSynchronize that a dead peer would hang. */
int xhist_comm_wait(xhist_comm* comm, void* stream);
int xhist_comm_destroy(xhist_comm* comm);

/* ---- device buffers ------------------------------------------------------------------------ */
/* Partial histograms that stay on their GPU between the kernel and the exchange, for hosts without a
 * device allocator of their own (the Python shim with numpy inputs; a C host).  What they carry is the
 * per-block result of `_bincount` (core.py:197-247) on its way into the sum over blocks (core.py:439).
 *   xhist_buffer_copy direction: 0 host -> device, 1 device -> host (both final when the call returns),
 *   2 device -> device on `device` (asynchronous on `stream`).
 *   xhist_buffer_add: dst[i] += src[i] on `device`, int64 or float64, asynchronous on `stream` — the sum
 *   of the partials of the blocks one GPU processed, before the all-reduce adds up the GPUs. */
int xhist_buffer_alloc(int device, size_t bytes, void** dptr);
int xhist_buffer_free(int device, void* dptr);
int xhist_buffer_copy(int device, void* dst, const void* src, size_t bytes, int direction, void* stream);
int xhist_buffer_add(int device, void* dst, const void* src, int64_t count, int dtype, void* stream);

/* Strided N-D copy between device buffers of `device`, asynchronous on `stream`: element (i_0 … i_{ndim-1}) of the source
 * (byte strides `src_strides`, 0 and negative allowed) goes to the same index of the destination (`dst_strides`).
 * ndim <= 8.  dst_dtype == src_dtype copies raw elements; dst_dtype == XHIST_F64 converts every supported dtype to float64
 * (numpy's promotion inside searchsorted, core.py:170).  This is what blocks of a device-resident dask array need around
 * the path: the moveaxis + reshape copy of core.py:218-226 for layouts no three strides describe, slices, and the
 * concatenation of unaligned chunks (test_chunking.py:104-146) — without a host round trip and without torch. */
int xhist_buffer_copy_nd(int device, int ndim, const int64_t* shape, const void* src, int src_dtype, const int64_t* src_strides,
                         void* dst, int dst_dtype, const int64_t* dst_strides, void* stream);
/* GPU a device pointer belongs to (for arrays that arrive through __cuda_array_interface__, which does not say) */
int xhist_pointer_device(const void* ptr, int* device);

/* ---- diagnostics / tuning (not part of the reference contract) ----------------------------- */
/* keys: "block_threads", "grid_blocks" (0 = auto), "force_global" (0/1), "force_generic" (0/1),
 *       "partition" (0 auto / 1 prefer / -1 never: multi-pass mode for histograms beyond LDS),
 *       "fused" (0 auto / 1 always / -1 never: that mode in one routing pass instead of count + prefix + scatter),
 *       "records48" (0 auto / -1 never: that pass moves float64 weights as 8-byte records — 36 mantissa bits next to the bin code,
 *       2^-37 relative per weight — while a call's weights have one sign, decided on the GPU; both signs fall back to full
 *       float64 records in the same call; XHIST_AMD_EXACT_RECORDS=1 is the process-wide "never"; full float64 records of a joint
 *       histogram travel through the rings of the "exchange" mode too, as 12-byte records in two tagged words — the weight bit
 *       for bit, any sign mixture),
 *       "exchange" (0 auto / 1 whenever the kernel can run / -1 never: such packed records never written to HBM — one persistent
 *       workgroup per compute unit keeps rows of a window of the histogram in LDS and records travel through rings inside each
 *       XCD; auto = float64 samples + float64 weights on numpy.linspace-style edges, one row of >= 2^25 samples, a chip of
 *       8 x 32 compute units, and a window that a probe on the GPU finds to hold 88 % of the call's samples; setting the key
 *       also re-admits a plan at once that an exchange aborted IN FLIGHT had taken off the mode — otherwise such a plan stays
 *       on the classic passes for its next 16 eligible calls, twice as many after every further abort, and is admitted again),
 *       "exchange_arrive_us" (0 = 200: how long the mode's 256 workgroups wait for one another to START — a compute unit held
 *       by another stream's or process's kernel keeps one out; after that nothing has been produced and the classic passes queued
 *       behind take the call; such calls are counted in the description, "exchange_arrival_misses", and do not take the plan off
 *       the mode), "exchange_budget_ms" (0 = 500: how long a workgroup waits for its peers ONCE RECORDS TRAVEL before the mode is
 *       switched off for the call and the classic passes take it; -1: not at all — tests), "exchange_min_pct" (0 = 88: the window
 *       coverage, per cent of the probe's samples, from which the mode takes a call),
 *       "lanes" (0 auto / 1 prefer / -1 never: one-row-per-lane kernels for many short rows),
 *       "arith" (0 auto / 1 prefer / -1 never: table-free digitize for numpy.linspace-style edges),
 *       "arith32" (0 auto / 1 prefer / -1 never: float32 samples on such edges digitized in float32 arithmetic),
 *       "pack" (0 auto / 1 whenever the plan has them / -1 never: packed 16-byte bucket entries — one LDS read per sample and
 *       dimension — for float64 / float32 samples on non-uniform edges, on a linear or a float-bit-pattern (logarithmic) grid),
 *       "route_grid", "acc_grid" (workgroups of the routing / adding-up pass of the multi-pass mode; 0 auto),
 *       "slices" (0 auto / 1 prefer / -1 never: histograms of a few times the LDS capacity in bin slices),
 *       "route_spl" (0 auto / 4: never the long 8-samples-per-lane tile) and "min_parts"
 *       (0 auto = 16 / 1 = as few as the histogram's size asks for / up to 128: partitions per row) shape the routing pass of
 *       the multi-pass mode; "route_pool_pct" (tests: its chunk pool cut to this percentage),
 *       "flat_rows" (0 auto / 1 for any row length below 65536 / -1 never: many dense short rows streamed as one array),
 *       "lds_copies" (0 = auto), "profile" (0 = off, R = keep the last R kernel timings),
 *       "profile_stride" (S >= 1: time every S-th execute only — for microsecond kernels, where the two event records of
 *       the profile mode cost as much as the launch itself) */
int xhist_plan_set_param(xhist_plan* plan, const char* key, int64_t value);
/* human-readable description of the last launch (kernel family, LDS bytes, copies, grid...) */
int xhist_plan_describe(xhist_plan* plan, char* buf, size_t cap);
/* "profile" = R > 0 keeps HIP-event pairs (recorded on the caller's stream, tightly around the
 * histogram kernel launch(es), after the output memset) for the R most recent device executes.
 * This call synchronises them, writes up to `cap` durations (ms, oldest first) and resets. */
int xhist_plan_profile_read(xhist_plan* plan, float* ms, int cap, int* n_out);

/* the library's scratch cache on `device` (staging buffers, record streams of the partitioned mode, transposes): stats[0] = bytes
 * cached (free, kept for reuse), [1] = bytes callers hold right now, [2] = bytes the cache may keep — twice what the largest
 * recent call held at once, at least 64 MiB, at most half of the device; $XHIST_AMD_POOL_KEEP_GB fixes it — [3] = that recent peak. */
int xhist_scratch_stats(int device, uint64_t* stats, int n);

/* TEST SUPPORT (ABI v9) — stands in for "somebody else's kernel holds compute units" (an RCCL kernel of a collective on another
 * stream, another process on the GPU): launches `workgroups` workgroups of 64 threads with `lds_bytes` of LDS each on `stream`,
 * which do nothing but wait until `microseconds` have passed.  With lds_bytes >= 64 KiB no 158 KB workgroup of the exchange mode
 * (the persistent kernel behind BASELINE C5's path, /root/reference/xhistogram/core.py:73-83 beyond LDS) can share their compute
 * units, which is what tests/test_gpu_exchange.py needs to show that such a call fails fast instead of waiting for a deadline.
 * Asynchronous; replaces nothing of the reference. */
int xhist_debug_hold_cus(int device, int workgroups, int lds_bytes, int64_t microseconds, void* stream);

/* free cached plans and scratch on every device */
int xhist_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* XHIST_AMD_H */
