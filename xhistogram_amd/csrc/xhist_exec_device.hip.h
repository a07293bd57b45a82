// xhist_exec_device.hip.h — host side: execution on device-resident arrays — partitioned mode, row-per-lane mode, streaming kernels
// Part of the single translation unit xhist_capi.hip (included there, in order).
#pragma once

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
    const bool int_dim = p->cmp == XHIST_CMP_I64 || ((p->cmp & ~0xff) == XHIST_CMP_PER_DIM && ((p->cmp >> d) & 1));
    const int sdt_d = samples[d].dtype;
    const bool unsigned_dt = sdt_d == XHIST_U64 || sdt_d == XHIST_U32 || sdt_d == XHIST_U16 || sdt_d == XHIST_U8 || sdt_d == XHIST_BOOL;
    if (int_dim && p->uns && !unsigned_dt)
      return fail(XHIST_ERR_UNSUPPORTED, "unsigned int64 compare domain needs unsigned integer samples (got dtype tag %d)", sdt_d);
    if (int_dim && !p->uns && (!dtype_is_int(sdt_d) || sdt_d == XHIST_U64))
      return fail(XHIST_ERR_UNSUPPORTED, "int64 compare domain needs signed/small integer samples (got dtype tag %d)", sdt_d);
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
  HIPC((hipError_t)xhist_hot_zero_words(static_cast<unsigned long long*>(out), n_words, grid, stream));  // (the kernel lives in the small hot code object)
  return XHIST_OK;
}

static const void* advance(const void* base, int dt, int64_t elems) {
  return static_cast<const char*>(base) + elems * dtype_size(dt);
}

// Partitioned mode (xhist_partition.hip.h): count -> prefix -> scatter -> accumulate, all on
// `stream`, scratch from the stream-ordered allocator (so concurrent callers never share it).
static int execute_partitioned(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_cols, void* out,
                               hipStream_t stream, int sdt, int wdt, int scan, bool use_f32, const TableSet& tset, int shift,
                               int n_parts, int profile, LaunchRecord& rec, bool first, bool last) {
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
  void* d_w = nullptr;  // record weights: float64, or float32 when the caller's weights are
  uint32_t* d_flat = nullptr;
  auto release = [&](int rc) {
    if (d_flat) (void)scratch_free(d_flat, stream);
    if (d_counts) (void)scratch_free(d_counts, stream);
    if (d_base) (void)scratch_free(d_base, stream);
    if (d_offsets) (void)scratch_free(d_offsets, stream);
    if (d_codes) (void)scratch_free(d_codes, stream);
    if (d_w) (void)scratch_free(d_w, stream);
    return rc;
  };
#define HIPR(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return release(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  HIPR(scratch_malloc((void**)&d_counts, (size_t)G * n_parts * 4, stream));
  HIPR(scratch_malloc((void**)&d_base, (size_t)G * n_parts * 8, stream));
  HIPR(scratch_malloc((void**)&d_offsets, (size_t)(n_parts + 1) * 8, stream));
  HIPR(scratch_malloc((void**)&d_flat, (size_t)n_tiles * kPartTile * 4, stream));
  const size_t n_rec = (size_t)n_cols + (size_t)G * n_parts * grp;  // every slice rounded up to whole groups
  HIPR(scratch_malloc((void**)&d_codes, n_rec * 2 + 16, stream));
  const bool rec_f32 = wdt == XHIST_F32;
  if (weighted) HIPR(scratch_malloc(&d_w, n_rec * (rec_f32 ? 4 : 8) + 16, stream));

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
  kp.part_w = static_cast<double*>(d_w);
  kp.part_shift = shift;
  kp.n_parts = n_parts;

  if (lds_count > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_count));
  if (lds_scatter > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scatter));
  kernel_fn_acc k_acc = weighted ? (rec_f32 ? (kernel_fn_acc)part_accumulate<true, float> : (kernel_fn_acc)part_accumulate<true, double>)
                                 : (kernel_fn_acc)part_accumulate<false, double>;
  if (lds_acc > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc));

  if (first)  // one timing record per execute: opened before the first row's kernels, closed after the last row's
    if (int rrc = rec.begin(profile)) return release(rrc);
  XH_LAUNCH_PICKED(k_count, dim3(G), dim3(kPartBlock), lds_count, stream, kp, d_flat);
  HIPR(hipGetLastError());
  hipLaunchKernelGGL(part_prefix, dim3(1), dim3(1024), 0, stream, (const uint32_t*)d_counts, G, n_parts, grp, d_offsets, d_base);
  HIPR(hipGetLastError());
  XH_LAUNCH_PICKED(k_scatter, dim3(G), dim3(kPartBlock), lds_scatter, stream, (const uint32_t*)d_flat,
                     weighted ? weights->data : nullptr, n_cols, (const uint64_t*)d_base, d_codes, d_w, shift, n_parts);
  HIPR(hipGetLastError());
  XH_LAUNCH_PICKED(k_acc, dim3(Gb), dim3(1024), lds_acc, stream, (const uint16_t*)d_codes, (const void*)d_w,
                     (const uint64_t*)d_offsets, out, p->n_bins, shift, n_parts);
  HIPR(hipGetLastError());
  {
    char desc[384];
    snprintf(desc, sizeof desc,
             "family=fast hist=partitioned parts=%d bins_per_part=%d group=%d vec=%d tile=%d block=%d grid=%d acc_grid=%d "
             "lds_count=%zu lds_scatter=%zu lds_acc=%zu scan=%d weighted=%d D=%d cmp=%s",
             n_parts, 1 << shift, grp, vec, kPartTile, kPartBlock, G, Gb, lds_count, lds_scatter, lds_acc, scan, (int)weighted, D,
             use_f32 ? "f32thr" : "f64");
    if (last)
      if (int rrc = rec.end(desc)) return release(rrc);
  }
#undef HIPR
  return release(XHIST_OK);
}

// Partitioned mode in one routing pass (xhist_route.hip.h): zero the chunk counters -> part_route ->
// part_accumulate_chunks, all on `stream`.  Returns XHIST_ERR_UNSUPPORTED (nothing launched) for what only
// the multi-pass form takes: more than 128 partitions, tables that do not fit LDS next to the sort buffers,
// in-bucket scans of 3 or 4 edges, integer samples.
// The exchange mode's kernel wants every compute unit for itself (256 persistent workgroups that wait for one another): two
// of them running on two streams of one GPU would each hold part of the chip and wait for the rest until their deadlines.
// So a call whose stream is not the one the last exchange kernel of this device went to looks at that kernel's event first
// and takes the classic passes when it has not finished.  (Other people's kernels on other streams can still hold compute
// units back; that costs the deadline once and takes the plan off the mode — execute_partitioned_fused.)
struct ExchInFlight {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  hipStream_t stream = nullptr;
  bool armed = false;
};
static ExchInFlight& exch_in_flight(int device) {
  static ExchInFlight slots[64];
  return slots[device & 63];
}

static bool exact_records_env() {
  static const bool v = [] { const char* e = getenv("XHIST_AMD_EXACT_RECORDS"); return e && *e && *e != '0'; }();
  return v;
}

static kernel_fn_route route_kernel(int sdt, int wdt, int D, int scan, bool multi, int block, int spl = 4) {
  if (block != 1024) return nullptr;
  if (spl == 8) return sdt == XHIST_F64 ? xhist_pick_route_f64_b1024s8(wdt, D, scan, multi) : sdt == XHIST_F32 ? xhist_pick_route_f32_b1024s8(wdt, D, scan, multi) : nullptr;
  if (spl != 4) return nullptr;
  return sdt == XHIST_F64 ? xhist_pick_route_f64_b1024(wdt, D, scan, multi) : sdt == XHIST_F32 ? xhist_pick_route_f32_b1024(wdt, D, scan, multi) : nullptr;
}

// Geometry of the routing pass: 1024-thread workgroups, 4 or 8 samples per lane and tile ("route_spl" override).
// Long tiles — 1024 threads x 8 samples, 64 KB read per input and tile — are what the mixed read/write traffic of the pass
// wants (tools/ubench/mixbw `mb`: 24 B read : 8 B written per sample runs at 4.8-5.6 TB/s with 32 KB bursts per workgroup and
// at 5.7-6.0 with 64 KB and more), as long as the tile's samples and weights fit the 128 registers a lane of a 1024-thread
// workgroup has: 2 x (bytes of one sample of every input + bytes of its weight) registers hold the tile, and up to 32 of
// them leave room for the rest (24 with table lookups, which take more registers than the arithmetic digitize).
// Measured over 5*10^8 samples, ms per call, 1024 x 4 -> 1024 x 8 | 2 x 512 x 4 (profiles/r03_s8_route_geometry.txt):
//   counts:  f64 pair 2.58 -> 2.40;  f64, 10^6 bins 1.82 -> 1.58;  f32 pair 1.98 -> 1.60 | 1.96;  f32 triple (3*10^8) 1.39 -> 1.20
//            | 1.57;  f64 triple (48 registers) 2.13 -> 2.19 (stays at 4)
//   f32 weights:  f32 pair 2.92 -> 2.64 | 3.62;  f32, 10^6 bins 2.47 -> 2.26 | 3.21;  f64, 10^6 bins 2.78 -> 2.55 | 3.33;
//            f64 pair (40 registers) 3.51 -> 3.49 | 4.12;  32 rows of 3*10^7 f32 pairs, 400 x 400 bins 5.42 -> 4.44 | 5.88
//   f64 weights, packed records:  f32 pair 3.20 -> 3.03 | 3.32;  f32, 10^6 bins 3.02 -> 2.77 | 3.00;
//            f32 triple (40 registers) 2.16 -> 2.47 | 2.26
//   f64 samples AND f64 weights are the exception to the register count: the long tile loses (C5 4.13 -> 4.52, it spills;
//   10^6 bins 3.36 -> 3.57) and they keep 1024 x 4.
// Two 512-thread workgroups per CU (VERDICT r2 "next" #1b: one's loads and stores under the other's arithmetic) beat one of
// 1024 by 2-3 % on C5 while the pass loaded its weights with the samples (3.43-3.48 -> 3.35-3.37 ms,
// profiles/r03_c5_blocks.txt); with the weights loaded a tile later and the arithmetic digitize they lose everywhere
// (C5 3.94 | 4.11, uniform 3.93 | 4.12, 10^6 bins 3.28 | 3.32, 8 rows x 512 x 512 bins 3.82 | 3.89; exact float64 records
// 4.21 | 4.32): the 512-thread instantiations were removed in round 5.
struct RouteGeom {
  int block, spl;
};
// Table lookups (edges that are not arithmetic) take more registers than the arithmetic digitize: 24 there.
static RouteGeom route_geom_for(const xhist_plan* p, int sdt, int wdt /* -1: counts */, int D, int scan) {
  RouteGeom g;
  g.block = 1024;
  // (route_long_tile_ok, xhist_pick.hip.h: the same rule decides which long-tile kernels exist; a forced 8 without one falls to 4)
  const bool long_ok = route_long_tile_ok(sdt == XHIST_F64 ? 8 : 4, wdt == XHIST_F64 ? 8 : wdt == XHIST_F32 ? 4 : 0, D, scan == kScanArith);
  g.spl = p->route_spl == 4 ? 4 : (long_ok ? 8 : 4);
  return g;
}

//
// Packed records (float64 weights).  A record is normally a 16-bit bin code plus the float64 weight: 10 bytes in two streams.
// Rounding the weight to 36 mantissa bits makes room for the code in its low 16 bits: ONE 8-byte record, 40 instead of 44
// bytes of traffic per C5 sample.  The rounding is 2^-37 = 7.3e-12 relative per weight, so a bin's sum is off by at most
// 7.3e-12 x sum|w| — five orders inside the 1e-6 contract as long as sum|w| is comparable to |sum w|, i.e. as long as the
// weights of a bin do not cancel.  That is guaranteed when all weights have one sign, which the routing pass learns for free
// while it reads them.  So: route with packed records and note the signs seen (on the GPU); the packed adding-up pass runs only
// if one sign was seen; if both were, an exact routing + adding-up pass — queued behind, and returning at once otherwise —
// redoes the work with full float64 records.  Every decision is taken on the GPU (no host synchronisation: the call stays
// asynchronous); the plan remembers mixed signs through a pinned host word the exact pass sets, and later calls skip the
// packed attempt.  "records48" = -1 or XHIST_AMD_EXACT_RECORDS=1 turn packing off.
//
// `rows` > 1: that many rows (uniform row strides) in ONE pass — the routing pass treats (row, partition) as the partition, so
// a few time steps of a big joint histogram cost one launch pair instead of one per row; rows * parts_per_row <= 128.
static int execute_partitioned_fused(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_cols, void* out,
                                     hipStream_t stream, int sdt, int wdt, int scan, bool use_f32, const TableSet& tset, int shift,
                                     int parts_per_row, int profile, LaunchRecord& rec, bool first, bool last, RouteGeom geom, int rows = 1) {
  const int D = p->n_dims;
  const bool weighted = weights != nullptr;
  const int n_parts = parts_per_row * rows;
  const int64_t n_total = n_cols * rows;
  if (n_total >= ((int64_t)1 << 40) || n_cols < 4 || n_parts > 128) return XHIST_ERR_UNSUPPORTED;
  const int block = geom.block, spl = geom.spl;
  kernel_fn_route k_route = route_kernel(sdt, wdt, D, scan, rows > 1, block, spl);
  if (!k_route) return XHIST_ERR_UNSUPPORTED;
  bool pack = weighted && wdt == XHIST_F64 && p->records48_pref >= 0 && !exact_records_env();
  if (!p->mixed_hint) {  // pinned host words the GPU writes: [0] a call met weights of both signs, [1] a chunk pool ran dry
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->mixed_hint) {
      uint32_t* h = nullptr;
      if (hipHostMalloc((void**)&h, 64, hipHostMallocDefault) == hipSuccess) { memset(h, 0, 64); p->mixed_hint = h; }
      else (void)hipGetLastError();
    }
  }
  if (p->mixed_hint) {
    // An exchange kernel of an earlier call gave up IN FLIGHT (a wait that outlived its deadline, a placement other than 32
    // workgroups per XCD) and the classic passes took that call: the plan stays away from the mode for its next 16 eligible
    // calls, twice as many after every further abort (a second hang would cost its deadline again), and is admitted again after
    // that; a clean call resets the count.  (Workgroups that were not all resident at the START cost arrive_ticks, not the
    // deadline, and do not count: note[5].)  "exchange" set again forgets all of it.
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->mixed_hint[2] != p->exchange_aborts_seen) {
      p->exchange_aborts_seen = p->mixed_hint[2];
      p->exchange_skip = p->exchange_backoff;
      p->exchange_backoff = std::min(p->exchange_backoff * 2, 4096);
    } else if (p->exchange_skip > 0) {
      p->exchange_skip -= 1;
    } else if (p->exchange_ran_last) {
      p->exchange_backoff = 16;
    }
    p->exchange_ran_last = false;
  }
  if (pack && (!p->mixed_hint || *p->mixed_hint != 0u)) pack = false;  // (earlier calls met both signs: straight to exact records)
  kernel_fn_route k_route48 = pack ? route_kernel(sdt, kWdtPacked48, D, scan, rows > 1, block, spl) : nullptr;
  if (pack && !k_route48) pack = false;
  // float64 weights that travel as FULL float64 records ("records48" = -1, XHIST_AMD_EXACT_RECORDS, or a plan that has met weights of
  // both signs): the exchange mode takes these too, with 12-byte records through two rings (part_exchange<D, true>, round 6)
  // (by default for joint histograms only: measured over the shape matrix of tools/exchange_cliffs.py with exact records,
  //  3*10^8 samples, 2-3 inputs read 0.93-1.03 of the classic exact passes — whose own time moves by 10 % from process to process
  //  and size to size with the placement of their five streams, while this form does not: C5 4.14-4.18 ms in every process against
  //  4.09 | 4.50-4.53 — and ONE input reads 1.08-1.11: profiles/r06_e_*)
  const bool xexact = weighted && wdt == XHIST_F64 && !pack && rows == 1 && (D >= 2 || p->exchange_pref > 0);
  // an exchange kernel of ANOTHER stream may still hold the GPU: looked at before the scratch allocations below, and the lock is
  // kept until this call's own exchange kernel has been launched and its event recorded (two host threads that both saw "idle"
  // would otherwise both launch one)
  bool other_stream_busy = false;
  std::unique_lock<std::mutex> xfl_lock;
  if ((pack || xexact) && p->exchange_pref >= 0) {
    ExchInFlight& fl = exch_in_flight(p->device);
    xfl_lock = std::unique_lock<std::mutex>(fl.mu);
    if (fl.armed && fl.stream != stream && hipEventQuery(fl.ev) == hipErrorNotReady) other_stream_busy = true;
    (void)hipGetLastError();
    if (other_stream_busy) xfl_lock.unlock();
  }
  const int32_t table_words = scan == kScanArith ? 0 : tset.words;  // arithmetic edges: no tables
  const int tile = route_tile(block, spl);
  const size_t lds_route = part_route_lds((size_t)table_words * 8, n_parts, weighted, tile, block);
  const bool rec_f32 = wdt == XHIST_F32;
  const size_t hist_bytes = (size_t)((1u << shift) + 1) * (weighted ? 8 : 4);
  const size_t lds_acc = ((hist_bytes + 15) & ~(size_t)15) + (size_t)kAccBatch * 8 + (size_t)(n_parts + 1) * 4 + 16;
  if (lds_route > p->lds_max || lds_acc > p->lds_max) return XHIST_ERR_UNSUPPORTED;
  // (Round 4 also ran the adding-up pass of sample sub-batch k on a second stream under the routing pass of sub-batch k + 1:
  // 25-60 % slower, both passes scale with the compute units they get — DESIGN_HISTORY 4.2b, profiles/r04_b_*; removed in
  // round 5.)
  int route_grid = 0, acc_grid = 0;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    route_grid = p->route_grid;
    acc_grid = p->acc_grid;
  }
  const int64_t n_piece = n_total;  // samples one routing pass sees
  const int64_t n_tiles = ((n_cols + tile - 1) / tile) * rows;
  // workgroups resident per CU: by LDS — and by registers: the routing pass is held to 128 per lane (waves_per_eu 4), so a
  // CU holds 1024 of its threads.  (Sized by LDS alone, a 77 KB workgroup of 1024 threads got a grid of 2 per CU, ran it
  // in two rounds, and 10^7 float64 pairs + weights into 512 x 512 bins took 0.233 ms against 0.170 for the three-pass route.)
  const int per_cu = std::max<int>(1, std::min<int>(std::min<int>(4, 1024 / block), (int)((size_t)160 * 1024 / lds_route)));
  int G = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)p->cus * per_cu, n_tiles));
  int Gb = (int)std::max<int64_t>(1, std::min<int64_t>(p->cus, (n_piece + 65535) / 65536));
  if (route_grid) G = (int)std::min<int64_t>(route_grid, n_tiles);
  if (acc_grid) Gb = acc_grid;
  // chunk size: a workgroup files at most kRouteListCap chunks (its list lives in LDS), 2^10 .. 2^14 records each
  int lg = 10;
  while (lg < 14 && (((n_piece / G) + tile) >> lg) + n_parts + 16 > route_list_cap(block)) ++lg;
  static const int lg_env = [] { const char* e = getenv("XHIST_AMD_ROUTE_CHUNK_LOG2"); return e && *e ? atoi(e) : 0; }();  // A/B runs only
  if (lg_env >= 10 && lg_env <= 14 && lg_env > lg) lg = lg_env;
  const int64_t GP = (int64_t)G * n_parts;
  // Chunks that hold records: every chunk but the one in use by its (workgroup, partition) owner is full, and at most 7
  // padding records are added per owner.  Ids taken from the pool but never used: a workgroup's stock drops fewer ids than
  // its largest request whenever a range of `batch` ids runs out (batch >= 8 x that request: < 1/7 of the ids it served),
  // and ends with at most two ranges in hand.
  const int64_t used = ((n_piece + 7 * GP) >> lg) + GP;
  int64_t pool_chunks = used + used / 6 + (int64_t)G * (2 * route_batch(n_parts, lg, tile) + 2 * route_max_need(lg, tile)) + 64;
  // "route_pool_pct" < 100 (tests): a pool too small on purpose — what finds no chunk goes straight into the output
  if (p->route_pool_pct > 0 && p->route_pool_pct < 100) pool_chunks = std::max<int64_t>(1, pool_chunks * p->route_pool_pct / 100);
  if (pool_chunks >= ((int64_t)1 << 31) || (pool_chunks << lg) >= ((int64_t)1 << 44)) return XHIST_ERR_UNSUPPORTED;

  uint32_t *d_ctr = nullptr, *d_plist = nullptr, *d_cmeta = nullptr;
  uint16_t* d_codes = nullptr;
  void* d_w = nullptr;
  void *x_ctl = nullptr, *x_part = nullptr;  // the exchange mode's scratch: what is zeroed per call (one block), and the XCD partials
  int64_t x_zero_words = 0;
  auto release = [&](int rc) {
    for (void* q : {x_ctl, x_part})
      if (q) (void)scratch_free(q, stream);
    if (d_ctr) (void)scratch_free(d_ctr, stream);
    if (d_plist) (void)scratch_free(d_plist, stream);
    if (d_cmeta) (void)scratch_free(d_cmeta, stream);
    if (d_codes) (void)scratch_free(d_codes, stream);
    if (d_w) (void)scratch_free(d_w, stream);
    return rc;
  };
#define HIPR(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return release(fail(XHIST_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  const int64_t ctr_words = (2 + n_parts + 1) / 2 + 1;  // [0] pool, [1] signs seen, [2 .. 2 + P) chunks filed per partition; two sets (packed + exact)
  const size_t rec_bytes = rec_f32 ? 4 : 8;
  const size_t plist_elems = (size_t)n_parts * pool_chunks, pool_recs = ((size_t)pool_chunks << lg);
  HIPR(scratch_malloc((void**)&d_ctr, (size_t)ctr_words * 8 * 2, stream));
  HIPR(scratch_malloc((void**)&d_plist, plist_elems * 4, stream));
  HIPR(scratch_malloc((void**)&d_cmeta, (size_t)pool_chunks * 4, stream));
  HIPR(scratch_malloc((void**)&d_codes, pool_recs * 2, stream));
  if (weighted) HIPR(scratch_malloc(&d_w, pool_recs * rec_bytes, stream));

  // The exchange mode (xhist_exchange.hip.h): packed records that never go through HBM — one persistent workgroup per compute
  // unit keeps rows of a WINDOW of the histogram in LDS, records travel through rings inside each XCD.  Offered for float64
  // samples + float64 weights on arithmetic edges, one row, on the 8 x 32-CU chip; whether a call takes it is decided on the
  // GPU by the probe (the window must hold >= 88 % of the samples) — the classic packed kernels below are queued all the same
  // and return at once when it does.  "exchange" = -1 / 1: never / whenever it can run (any size, any coverage: tests).
  ExchArgs xa;
  memset(&xa, 0, sizeof xa);
  bool xch = false, xch_probe = false;
  kernel_fn_exch k_xch = nullptr, k_xprobe = nullptr;
  size_t lds_xch = 0;
  if ((pack || xexact) && rows == 1 && p->exchange_pref >= 0 && (p->exchange_skip == 0 || p->exchange_pref > 0) && sdt == XHIST_F64 && p->arith && p->arith_pref >= 0 && D <= 3 && p->cus == kExchXcds * kExchRings &&
      p->n_bins <= ((int64_t)1 << 23) /* (the side copy is zeroed per call) */ && (p->exchange_pref > 0 || n_cols >= ((int64_t)1 << 25))) {
    const int64_t L = D >= 2 ? (int64_t)p->ts[0][0].dim[D - 1].nb : 256;
    const int64_t hist_rows = D >= 2 ? p->n_bins / L : (p->n_bins + 255) / 256;
    const int64_t units = (hist_rows + kExchUnitRows - 1) / kExchUnitRows;
    const int64_t max_local = xexact ? kExchMaxLocalExact : kExchMaxLocal;
    const int64_t rows_per = L <= max_local ? std::min<int64_t>(max_local / L, units) : 0;
    k_xch = xhist_pick_exchange(D, xexact);
    k_xprobe = xhist_pick_exchange_probe(D);
    lds_xch = rows_per >= 1 ? exchange_lds((int)(rows_per * L), xexact) : 0;
    if (rows_per >= 1 && units <= 4096 && k_xch && k_xprobe && lds_xch <= p->lds_max && !other_stream_busy) {
      xch = true;
      xch_probe = true;  // (always: a histogram that fits the window has nothing outside it, but its owners' loads must be even too)
      xa.row_len = L;
      xa.n_hist_rows = hist_rows;
      xa.rows_per = (int32_t)rows_per;
      xa.local_bins = (int32_t)(rows_per * L);
      xa.n_units = (int32_t)units;
      xa.force = p->exchange_pref > 0 ? 1 : 0;
      { int64_t p2 = 1; while (p2 * 2 <= L) p2 *= 2; xa.side_rot_mask = (int32_t)(p2 - 1); }
      xa.min_ppm = p->exchange_min_pct > 0 ? p->exchange_min_pct * 10000 : kExchMinPpm;
      xa.max_uneven_ppm = 1250000;
      xa.budget_ticks = p->exchange_budget_ms < 0 ? 0 : (long long)(p->exchange_budget_ms ? p->exchange_budget_ms : 500) * 100000;
      xa.arrive_ticks = (long long)(p->exchange_arrive_us > 0 ? p->exchange_arrive_us : 200) * 100;  // 200 us: the launch of 256 workgroups on a free chip takes a few
      const size_t words_bytes = ((size_t)(units + kExchRings + 8 + 32) * 4 + 7) & ~(size_t)7;  // win[8], the cold arguments (32 words), the probe's counts (units, then owners)
      // one block for everything that is zeroed per call (one launch): control words | window, cold arguments, probe counts | side copy | rings
      const size_t ctl_bytes = sizeof(ExchCtl) * kExchXcds, side_bytes = (size_t)(hist_rows * L) * 8;  // (whole rows: exch_side_index rotates inside a row)
      const size_t rings_bytes = (size_t)kExchXcds * kExchRings * kExchRings * kExchCap * (xexact ? 12 : 8);  // (exact records: the ring of low bits behind the ring of packed words)
      static_assert(sizeof(ExchCtl) % 8 == 0, "the block's parts stay 8-byte aligned");
      x_zero_words = (int64_t)((ctl_bytes + words_bytes + side_bytes + rings_bytes) / 8);
      HIPR(scratch_malloc(&x_ctl, ctl_bytes + words_bytes + side_bytes + rings_bytes, stream));
      HIPR(scratch_malloc(&x_part, (size_t)kExchXcds * kExchRings * (size_t)xa.local_bins * 8, stream));
      xa.ctl = static_cast<ExchCtl*>(x_ctl);
      xa.win = reinterpret_cast<uint32_t*>(static_cast<char*>(x_ctl) + ctl_bytes);
      xa.side = reinterpret_cast<double*>(static_cast<char*>(x_ctl) + ctl_bytes + words_bytes);
      xa.rings = reinterpret_cast<uint64_t*>(static_cast<char*>(x_ctl) + ctl_bytes + words_bytes + side_bytes);
      xa.rings_lo = xexact ? reinterpret_cast<uint32_t*>(xa.rings + (size_t)kExchXcds * kExchRings * kExchRings * kExchCap) : nullptr;
      xa.part = static_cast<double*>(x_part);
      xa.cold = reinterpret_cast<ExchCold*>(xa.win + 8);
      static_assert(sizeof(ExchCold) <= 32 * 4, "the cold arguments fit their 32 words");
      xa.counts = xa.win + 8 + 32;
      xa.note = p->mixed_hint;
      if (lds_xch > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_xch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_xch));
      // every workgroup of the kernel must be resident (they wait for one another): one per compute unit is what its LDS and
      // registers must allow; asked once per plan and LDS size (a kernel of the host's that holds compute units at run time is
      // the deadline's business)
      if (p->exchange_occ_lds != lds_xch) {
        int per_cu_x = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_x, (const void*)k_xch, kExchBlock, lds_xch) != hipSuccess) { per_cu_x = 0; (void)hipGetLastError(); }
        std::lock_guard<std::mutex> lk(p->mu);
        p->exchange_occ_lds = lds_xch;
        p->exchange_occ_ok = per_cu_x >= 1;
      }
      if (!p->exchange_occ_ok) xch = false;
    }
  }

  if (lds_route > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_route, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_route));
  kernel_fn_acc_chunks k_acc = weighted ? (rec_f32 ? (kernel_fn_acc_chunks)part_accumulate_chunks<true, float>
                                                   : (kernel_fn_acc_chunks)part_accumulate_chunks<true, double>)
                                        : (kernel_fn_acc_chunks)part_accumulate_chunks<false, double>;
  if (lds_acc > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_acc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc));
  kernel_fn_acc_chunks k_acc48 = (kernel_fn_acc_chunks)part_accumulate_chunks<true, double, true>;
  if (pack) {
    if (lds_route > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_route48, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_route));
    if (lds_acc > 48 * 1024) HIPR(hipFuncSetAttribute((const void*)k_acc48, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_acc));
  }

  if (first)  // one timing record per execute: opened before the first row's kernels, closed after the last row's
    if (int rrc = rec.begin(profile)) return release(rrc);
  const DimTable* dims = tset.dim;
  {
    Params kp;
    memset(&kp, 0, sizeof kp);
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
    kp.n_rows = rows;
    kp.n_cols = n_cols;
    kp.n_bins = p->n_bins;
    kp.out = out;
    kp.part_shift = shift;
    kp.n_parts = n_parts;
    kp.parts_per_row = parts_per_row;
    const int Gk = G;
    kp.segs = Gk;
    uint32_t* ctr = d_ctr;  // (two counter sets of ctr_words 8-byte words each)
    RouteArgs ra;
    ra.pool = ctr;
    ra.pcount = ctr + 2;
    ra.plist = d_plist;
    ra.cmeta = d_cmeta;
    ra.codes = d_codes;
    ra.wrec = weighted ? d_w : nullptr;
    ra.list_cap = (uint32_t)pool_chunks;
    ra.chunk_log2 = lg;
    ra.flags = ctr + 1;
    ra.gate = nullptr;
    ra.hint = nullptr;
    ra.gate_mode = 0;
    ra.dry = p->mixed_hint ? p->mixed_hint + 1 : nullptr;

    if (int zrc = zero_output(ctr, ctr_words * 2, stream)) return release(zrc);
    ra.xgate = nullptr;
    if (xch) {
      xa.flags = ctr + 1;
      if (int zrc = zero_output(x_ctl, x_zero_words, stream)) return release(zrc);
      for (int d = 0; d < D; ++d) {  // (the arithmetic-edge constants of the float64 table set, whatever the classic routing pass digitizes with)
        const DimTable& t = p->ts[0][0].dim[d];
        xa.s_ptr[d] = static_cast<const double*>(samples[d].data);
        xa.dim[d].e0 = t.e0_f;
        xa.dim[d].eL = t.eL_f;
        xa.dim[d].step = t.step;
        xa.dim[d].inv_step = t.inv_step;
        xa.dim[d].arith_h = t.arith_h;
        xa.dim[d].nb = t.nb;
      }
      xa.w_ptr = static_cast<const double*>(weights->data);
      xa.n = n_cols;
      if (xch_probe) {
        XH_LAUNCH_PICKED(k_xprobe, dim3(256), dim3(256), 0, stream, xa);
        HIPR(hipGetLastError());
      }
      XH_LAUNCH_PICKED(xhist_pick_exchange_pick(), dim3(1), dim3(64), 0, stream, xa);
      HIPR(hipGetLastError());
      XH_LAUNCH_PICKED(k_xch, dim3(kExchXcds * kExchRings), dim3(kExchBlock), lds_xch, stream, xa);
      HIPR(hipGetLastError());
      {
        ExchInFlight& fl = exch_in_flight(p->device);  // (xfl_lock holds fl.mu since the look at the last kernel's event)
        if (!fl.ev && hipEventCreateWithFlags(&fl.ev, hipEventDisableTiming) != hipSuccess) { fl.ev = nullptr; (void)hipGetLastError(); }
        if (fl.ev && hipEventRecord(fl.ev, stream) == hipSuccess) { fl.stream = stream; fl.armed = true; }
        else { fl.armed = false; (void)hipGetLastError(); }
      }
      {
        std::lock_guard<std::mutex> lk(p->mu);
        p->exchange_ran_last = true;
      }
      ra.xgate = xa.win + 1;  // the classic packed kernels below return at once when the mode took the call
    }
    if (xfl_lock.owns_lock()) xfl_lock.unlock();
    RouteArgs ra48 = ra;
    if (pack) {
      XH_LAUNCH_PICKED(k_route48, dim3(Gk), dim3(block), lds_route, stream, kp, ra);  // packed records, notes the signs
      HIPR(hipGetLastError());
      ra48.gate = ctr + 1;
      ra48.gate_mode = 1;  // one sign: add the packed records up
      // both signs: the exact pass below runs (its own counters: the second set), otherwise its kernels return at once
      ra.pool = ctr + 2 * ctr_words;
      ra.pcount = ra.pool + 2;
      ra.flags = ra.pool + 1;
      ra.gate = ctr + 1;
      ra.gate_mode = 2;
      ra.hint = p->mixed_hint;
      ra.xgate = nullptr;  // (the exact passes redo the call whenever the sign word says so, exchange mode or not)
    }
    XH_LAUNCH_PICKED(k_route, dim3(Gk), dim3(block), lds_route, stream, kp, ra);
    HIPR(hipGetLastError());
    if (pack) {
      XH_LAUNCH_PICKED(k_acc48, dim3(Gb), dim3(1024), lds_acc, stream, ra48, out, p->n_bins, shift, n_parts, rows > 1 ? parts_per_row : 0);
      HIPR(hipGetLastError());
    }
    XH_LAUNCH_PICKED(k_acc, dim3(Gb), dim3(1024), lds_acc, stream, ra, out, p->n_bins, shift, n_parts, rows > 1 ? parts_per_row : 0);
    HIPR(hipGetLastError());
    if (xch) {  // (runs when the mode was on and the weights had one sign: exactly when none of the four kernels above did)
      XH_LAUNCH_PICKED(xhist_pick_exchange_merge(), dim3((unsigned)((p->n_bins + 255) / 256)), dim3(256), 0, stream, xa, static_cast<double*>(out), p->n_bins);
      HIPR(hipGetLastError());
    }
  }
  {
    char desc[768];
    snprintf(desc, sizeof desc,
             "family=fast hist=partitioned route=fused rows_per_pass=%d parts=%d bins_per_part=%d group=%d chunk=%d chunks<=%lld tile=%d block=%d grid=%d "
             "acc_grid=%d lds_route=%zu lds_acc=%zu scan=%d weighted=%d D=%d cmp=%s records=%s exchange=%s exchange_window_ppm_before=%u exchange_aborts=%u exchange_busiest_owner_ppm_before=%u exchange_arrival_misses=%u exchange_records=%s",
             rows, n_parts, 1 << shift, kRouteGrp, 1 << lg, (long long)pool_chunks, tile, block, G, Gb, lds_route, lds_acc, scan,
             (int)weighted, D, use_f32 ? "f32thr" : "f64",
             !weighted ? "u16" : pack ? "packed48(+exact if both signs)" : wdt == XHIST_F32 ? "u16+f32" : "u16+f64",
             !xch ? "no" : xa.force ? "forced" : xa.n_units > xa.rows_per ? "if the probe's window holds enough of the samples" : "whole histogram in the window",
             p->mixed_hint ? p->mixed_hint[3] : 0u, p->mixed_hint ? p->mixed_hint[2] : 0u, p->mixed_hint ? p->mixed_hint[4] : 0u, p->mixed_hint ? p->mixed_hint[5] : 0u,  // (what the GPU has reported so far: the calls before this one)
             !xch ? "-" : xexact ? "exact12(two_rings)" : "packed8");
    if (last)
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
  // integer / half-precision samples: one input, unweighted or float64 weights, and only as views whose rows
  // are the contiguous direction (10^6 x 100 int32 over the leading axis: generic family 1.70 ms)
  const bool small = sdt == XHIST_I32 || sdt == XHIST_I64 || sdt == XHIST_I16 || sdt == XHIST_U8 || sdt == XHIST_F16;
  if (small ? !(D == 1 && (wdt == -1 || wdt == XHIST_F64))
            : ((sdt != XHIST_F64 && sdt != XHIST_F32) || (wdt != -1 && wdt != XHIST_F64 && wdt != XHIST_F32)))
    return XHIST_ERR_UNSUPPORTED;
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
  const TableSet* picked = &pick_tables(p, use_f32, &scan);
  if (small && scan > 1) {  // their kernels exist for the one-compare scan and the binary search only
    picked = &p->ts[0][0];
    scan = 0;
  }
  // crowded buckets (geometric edges: a binary search in the two-level tables): the packed entries, general variant — the
  // row-per-lane and flat-rows kernels digitize sample by sample and feel every table read ((1825, 360, 720) float32 over
  // time, 50 geometric bins: 0.97 -> see profiles/r04_j_*)
  {
    int pack_pref;
    { std::lock_guard<std::mutex> lk(p->mu); pack_pref = p->pack_pref; }
    const int pk_np = small ? 0 : (sdt == XHIST_F64 ? p->pk_np : (use_f32 ? p->pk32_np : 0));
    // (only against the binary search: with 3-4 edges per bucket the two-level tables are AHEAD in these LDS-hungry kernels —
    //  400 random edges, float32: row-per-lane 0.62 against 0.89 ms, flat rows 0.95 against 1.60)
    if (pk_np && pack_pref >= 0 && (scan == 0 || pack_pref > 0)) {
      picked = sdt == XHIST_F64 ? &p->ts_pk : &p->ts_pk32;
      scan = kScanPackG;
    }
  }
  const TableSet& tset = *picked;
  const size_t table_bytes = (size_t)tset.words * 8;
  const size_t lds_bytes = table_bytes + (size_t)p->n_bins * kLanePitch * (weighted ? 8 : 4);
  bool transpose = false;
  if (small && !(all_natural && samples[0].row_stride == 1)) return XHIST_ERR_UNSUPPORTED;
  // dense short rows of one or two float inputs, with or without float weights: streamed flat (hist_flat_rows)
  // (3.65 x 10^8 float32 samples, 50 bins, ms per call, fused turn-around kernel / row streaming | flat: rows of 20 2.25 | 1.83;
  // 64 1.00 | 0.72; 100 1.07 | 0.66; 200 0.84 | 0.46; 365 0.79 | 0.41; 512 0.59 | 0.42; 720 0.46 | 0.43; 1024 0.33 | 0.38;
  // 2048 0.28 | 0.37: profiles/r03_f_flat_rows.txt)
  constexpr int64_t kFlatMaxCols = 800;
  const uint32_t flat_nbp = weighted ? (uint32_t)p->n_bins : (((uint32_t)p->n_bins + 1u) & ~1u);
  // (copies of a row's counters for the lanes of a wavefront to spread over — they read neighbouring samples, one or two
  // rows at a time — were measured and bought nothing: 10^6 rows of 365 float32, 1 / 2 / 4 / 8 copies 0.403 / 0.411 / 0.422 /
  // 0.497 ms; rows of 200: 0.463 / 0.468 / 0.520 / 0.684; the kernel keeps the parameter, XHIST_AMD_FLAT_K_LOG2 sets it)
  static const int flat_k_env = [] { const char* e = getenv("XHIST_AMD_FLAT_K_LOG2"); return e && *e ? atoi(e) : 0; }();
  const int flat_k_log2 = std::max(0, std::min(3, flat_k_env));
  const size_t flat_row_bytes = ((size_t)flat_nbp << flat_k_log2) * (weighted ? 8 : 2);  // LDS per row of a workgroup
  bool flat_ok = (D == 1 || D == 2) && !small && n_cols >= 1 && n_cols < 65536 && (sdt != XHIST_F32 || use_f32) && flat_row_bytes <= 4096 &&
                 (!weighted || wdt == XHIST_F32 || wdt == XHIST_F64);
  for (int d = 0; d <= D && flat_ok; ++d) {
    if (d == D && !weighted) break;
    const xhist_array& a = d < D ? samples[d] : *weights;
    const size_t vec_bytes = (size_t)(16 / dtype_size(sdt)) * (size_t)dtype_size(a.dtype);  // one load: as many elements as a 16-byte sample vector
    flat_ok = a.col_stride == 1 && a.row_stride == n_cols && a.inner_rows == 0 && ((uintptr_t)a.data & (vec_bytes - 1)) == 0;
  }
  const bool use_flat = flat_ok && !prefer && p->flat_rows >= 0 && n_rows >= 4096 && n_cols <= (p->flat_rows > 0 ? 65535 : kFlatMaxCols) &&
                        !(all_natural && samples[0].row_stride == 1);
  // (hist_lanes: a counter column per lane, or — shared by a row's lane groups — per row, for as few as 16 rows per workgroup)
  const size_t lds_least = table_bytes + (size_t)p->n_bins * (weighted ? 17 * 8 : 9 * 4);
  if (lds_least > p->lds_max && !use_flat) return XHIST_ERR_UNSUPPORTED;
  if (all_natural && (samples[0].row_stride == 1 || prefer)) {
    // rows are the contiguous direction: the row-streaming kernels cannot coalesce this at all
  } else if (use_flat) {
    transpose = true;  // (dense short rows: hist_flat_rows below)
  } else if (all_rowmajor && (prefer || (n_rows >= 4096 && n_cols <= ((D == 1 && !weighted) ? 400 : 80)))) {
    // many short rows.  Measured crossovers with the row-streaming kernels (64-thread workgroups,
    // few LDS copies, plain-store flush; 3.65 x 10^8 f32 samples): ~400 columns for the fused kernel
    // (one unweighted input: 384 columns 0.64 against 0.67 ms, 512 columns 0.62 against 0.50), below
    // ~90 for scratch-transpose + lanes (weighted, 96 columns: 3.6 against 3.3 ms)
    transpose = true;
  } else {
    return XHIST_ERR_UNSUPPORTED;
  }

  int vec = 1;
  // (counts: uint16 counter columns, two per word — the only unweighted form: a column segment longer than 65535 samples
  //  would need 2^24 rows to be asked for, see the packed16 rule below; sums: float64 columns)
  kernel_fn_lanes fn = (kernel_fn_lanes)fast_kernel(sdt, wdt, D, scan, weighted ? kHistLanes : kHistLanes16, &vec);
  if (!fn) return XHIST_ERR_UNSUPPORTED;

  const bool fused_ok = D == 1 && !weighted && n_cols < 65536 && samples[0].col_stride == 1 && samples[0].row_stride != 0;
  if (transpose && grouped_any && !fused_ok) return XHIST_ERR_UNSUPPORTED;  // transpose_2d takes plain row strides only
  if (use_flat) {
    kernel_fn_flat ff = xhist_pick_flat_rows(sdt, weighted ? wdt : -1, D, scan);
    // R rows per workgroup: ~16 Ki samples, at most 24 KiB of counters, and enough workgroups for every CU
    int64_t R = std::min<int64_t>(16384 / n_cols, (int64_t)(24576 / flat_row_bytes));
    R = std::min<int64_t>(R, std::max<int64_t>((2048 + n_cols - 1) / n_cols, n_rows / ((int64_t)p->cus * 8)));  // (2048+ samples a workgroup)
    R = std::max<int64_t>(1, std::min<int64_t>(R, 1024));
    const size_t lds_flat = ((table_bytes + 15) & ~(size_t)15) + (size_t)R * flat_row_bytes + 32 * (weighted ? 8 : 4);
    const int64_t row_blocks = (n_rows + R - 1) / R;
    if (ff && lds_flat <= p->lds_max && row_blocks <= 2147483647LL) {
      Params kp;
      memset(&kp, 0, sizeof kp);
      for (int d = 0; d < D; ++d) {
        kp.s_ptr[d] = samples[d].data;
        kp.s_rs[d] = samples[d].row_stride;
        kp.s_cs[d] = 1;
        kp.s_dt[d] = sdt;
        kp.dim[d] = tset.dim[d];
      }
      if (weighted) {
        kp.w_ptr = weights->data;
        kp.w_rs = weights->row_stride;
        kp.w_cs = 1;
        kp.w_dt = weights->dtype;
      }
      kp.n_dims = D;
      kp.tables = tset.blob;
      kp.table_words = tset.words;
      kp.tables_in_lds = 1;
      kp.n_rows = n_rows;
      kp.n_cols = n_cols;
      kp.n_bins = p->n_bins;
      kp.out = out;
      const int direct = accumulate ? 0 : 1;
      const uint64_t magic = (((uint64_t)1 << 40) / (uint64_t)n_cols) + 1u;
      LaunchRecord rec(p, stream);
      if (int rrc = rec.begin(profile)) return rrc;
      if (lds_flat > 48 * 1024) HIPC(hipFuncSetAttribute((const void*)ff, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_flat));
      XH_LAUNCH_PICKED(ff, dim3((unsigned)row_blocks), dim3(kLaneBlock), lds_flat, stream, kp, (int32_t)direct, (int32_t)R, (int32_t)flat_k_log2,
                         magic, (int64_t)(n_rows * n_cols));
      HIPC(hipGetLastError());
      char desc[320];
      snprintf(desc, sizeof desc,
               "family=flat_rows hist=%s rows_per_wg=%lld copies=%d direct_store=%d block=%d grid=%lld lds_bytes=%zu scan=%d weighted=%d D=%d cmp=%s",
               weighted ? "lds" : "lds16", (long long)R, 1 << flat_k_log2, direct, kLaneBlock, (long long)row_blocks, lds_flat, scan, (int)weighted, D,
               use_f32 ? "f32thr" : "f64");
      return rec.end(desc);
    }
  }
  // the flat kernel was the only reason to be here and was not launched (no such variant, or its LDS does not fit): beyond
  // the column limits of the scratch-transpose route the row-streaming kernels are the faster answer
  if (use_flat && !(all_rowmajor && n_cols <= ((D == 1 && !weighted) ? 400 : 80))) return XHIST_ERR_UNSUPPORTED;
  if (lds_least > p->lds_max) return XHIST_ERR_UNSUPPORTED;
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
      XH_LAUNCH_PICKED(f1, dim3((unsigned)row_blocks), dim3(kLaneBlock), lds_f, stream, kp, (int32_t)direct);
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
      if (s) (void)scratch_free(s, stream);
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
      HIPL(scratch_malloc(&scratch[d], (size_t)n_rows * n_cols * es, stream));
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

  int lane_rows = kLaneBlock;  // the smallest power of two that holds the rows when there are fewer than 256
  while (lane_rows > 1 && lane_rows / 2 >= n_rows) lane_rows /= 2;
  // more rows than that, long reductions: 64-row workgroups with four column streams keep 4x the loads in
  // flight per row (10^5 x 1000 f32 over the leading axis: 0.25 -> 0.14 ms); short reductions pay for the
  // smaller blocks' zeroing and write-out instead (10^6 rows of 100: 0.17 -> 0.19 ms) and keep 256
  if (n_rows > 128 && n_cols >= 512) lane_rows = 64;
  // Counter columns per lane need n_bins x 257 words.  Where that does not fit (joint histograms over a leading axis:
  // 20 x 20 bins, (1825, 360, 720) float32 pairs over `time`: the generic family took 7.8 ms, now 1.0) or leaves one
  // workgroup per CU (float64 sums of 50 bins: 103 KB; the same array with weights 1.49 -> 0.78 ms), the lane groups of a
  // row share one column per row and the workgroup shrinks to as many rows as fit 40 KiB, 16 at least (joint counts: 64
  // rows / 54 KB 1.36 ms, 32 rows / 28 KB 1.02, 16 rows 1.80; weighted 50 bins: 128 rows / 53 KB 0.96-1.05, 64 rows 0.78-1.09).
  int lane_pitch = 0;
  auto lds_shared = [&](int rows, bool u16) { return table_bytes + (size_t)p->n_bins * (u16 ? (size_t)(rows / 2 + 1) * 4 : (size_t)(rows + 1) * (weighted ? 8 : 4)); };
  const bool u16_ok = !weighted && n_cols <= 65535;  // (one workgroup per row block then sees every column: see segs16 below)
  static const size_t share_above = [] { const char* e = getenv("XHIST_AMD_LANES_SHARE_ABOVE_KB"); return (size_t)(e && *e ? atoi(e) : 64) * 1024; }();  // A/B runs
  static const size_t share_target = [] { const char* e = getenv("XHIST_AMD_LANES_SHARE_TARGET_KB"); return (size_t)(e && *e ? atoi(e) : 40) * 1024; }();
  if ((u16_ok ? table_bytes + (size_t)p->n_bins * (kLaneBlock / 2 + 1) * 4 : lds_bytes) > std::min(p->lds_max, share_above)) {
    lane_rows = std::min(lane_rows, 128);
    while (lane_rows > 16 && lds_shared(lane_rows, u16_ok) > share_target) lane_rows /= 2;
    if (lds_shared(lane_rows, u16_ok) > p->lds_max) return release(XHIST_ERR_UNSUPPORTED);
    lane_pitch = 1;  // (the number of words is set below, once the counter width is known)
  }
  kp.lane_rows = lane_rows;
  const int64_t row_blocks = (n_rows + lane_rows - 1) / lane_rows;
  // unweighted and few enough columns per workgroup: uint16 counters, half the LDS
  size_t lds_use = lane_pitch ? lds_shared(lane_rows, false) : lds_bytes;
  bool packed16 = false;
  if (!weighted) {
    const size_t lds16 = lane_pitch ? lds_shared(lane_rows, true) : table_bytes + (size_t)p->n_bins * (kLaneBlock / 2 + 1) * 4;
    const int bpc16 = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)160 * 1024 / lds16));
    int64_t segs16 = std::max<int64_t>(1, ((int64_t)p->cus * bpc16 * 2 + row_blocks - 1) / row_blocks);
    segs16 = std::min<int64_t>(std::min<int64_t>(segs16, std::max<int64_t>(1, n_cols / 64)), 65535);
    // (a segment of more than 65535 columns: only with so many row blocks that there is one segment each — beyond 10^10
    //  samples; the row-streaming kernels take that call.  The uint32-column kernels this used to fall back to were never
    //  selected by anything — census of round 6 — and are gone)
    if ((n_cols + segs16 - 1) / segs16 > 65535) return release(XHIST_ERR_UNSUPPORTED);
    packed16 = true;
    lds_use = lds16;
  }
  if (lane_pitch) lane_pitch = packed16 ? lane_rows / 2 + 1 : lane_rows + 1;
  kp.lane_pitch = lane_pitch;
  if (lds_use > p->lds_max) return release(XHIST_ERR_UNSUPPORTED);
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
  XH_LAUNCH_PICKED(fn, dim3((unsigned)row_blocks, (unsigned)col_segs), dim3(kLaneBlock), lds_use, stream, kp, (int32_t)direct,
                     cols_per_seg);
  HIPL(hipGetLastError());
  {
    char desc[384];
    snprintf(desc, sizeof desc,
             "family=lanes hist=%s%s rows_per_wg=%d transpose=%d direct_store=%d block=%d grid=%lldx%lld lds_bytes=%zu scan=%d weighted=%d D=%d cmp=%s",
             packed16 ? "lds16" : "lds", lane_pitch ? "(shared)" : "", lane_rows, (int)transpose, direct, kLaneBlock, (long long)row_blocks, (long long)col_segs, lds_use, scan,
             (int)weighted, D,
             use_f32 ? "f32thr" : "f64");
    if (int rrc = rec.end(desc)) return release(rrc);
  }
#undef HIPL
  return release(XHIST_OK);
}

// weights2 / out2 (optional): a second weight array binned in the same pass (hist_fast<..., W2>).
// Only the vector family with LDS histograms does that; anything else returns
// XHIST_ERR_UNSUPPORTED before touching the outputs, and the caller runs two passes.
static int execute_device(xhist_plan* p, const xhist_array* samples, const xhist_array* weights, int64_t n_rows,
                          int64_t n_cols, void* out, int accumulate, hipStream_t stream,
                          const xhist_array* weights2 = nullptr, void* out2 = nullptr) {
  const int D = p->n_dims;
  const bool weighted = weights != nullptr;
  const bool two = weights2 != nullptr;
  const int64_t out_elems = n_rows * p->n_bins;
  if (out_elems == 0) return XHIST_OK;
  if (n_cols == 0) {
    if (!accumulate) {
      if (int zrc = zero_output(out, out_elems, stream)) return zrc;
      if (two)
        if (int zrc = zero_output(out2, out_elems, stream)) return zrc;
    }
    return XHIST_OK;
  }

  int block_threads, grid_blocks, force_global, force_generic, lds_copies, profile, partition, lanes, arith_pref, slices_pref, fused_pref, pack_pref, arith32_pref;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    block_threads = p->block_threads; grid_blocks = p->grid_blocks; force_global = p->force_global;
    force_generic = p->force_generic; lds_copies = p->lds_copies; profile = p->profile; partition = p->partition;
    lanes = p->lanes; arith_pref = p->arith_pref; slices_pref = p->slices_pref; fused_pref = p->fused_pref;
    pack_pref = p->pack_pref;
    arith32_pref = p->arith32_pref;
  }

  // ---- many short rows / leading-axis reductions: one row per lane (xhist_lanes.hip.h) --------
  if (lanes >= 0 && !force_generic && !force_global && !two) {
    const int rc = execute_lanes(p, samples, weights, n_rows, n_cols, out, accumulate, stream, lanes > 0, profile);
    if (rc != XHIST_ERR_UNSUPPORTED) return rc;  // UNSUPPORTED = not this shape, fall through
  }
  // (the output is zeroed further down, once the launch geometry says whether it has to be at all)

  // ---- a few rows that are the contiguous direction, too many bins for the row-per-lane kernels ----------------
  // (the columns of an (n, K) table histogrammed over its leading axis with thousands of bins): every other kernel
  // would walk them with a column stride of K elements — the generic family at ~1 TB/s, memory-side atomics at 0.1
  // beyond LDS.  Gather the rows into dense scratch first (one pass, read + write) and run the vector kernels on that:
  // 25 x 10^6 x 4 float32, 5000 bins 0.44 -> see profiles/r02_j_table_columns*.jsonl.
  if (!force_generic && !two && n_rows >= 1 && n_rows <= 64 && n_cols >= 4096) {
    bool table = true;
    for (int d = 0; d <= D && table; ++d) {
      if (d == D && !weighted) break;
      const xhist_array& a = d < D ? samples[d] : *weights;
      table = a.inner_rows == 0 && a.row_stride == 1 && a.col_stride >= n_rows && a.col_stride > 1 &&
              ((uintptr_t)a.data % (size_t)dtype_size(a.dtype)) == 0;
    }
    // (up to four 8-byte columns with a histogram that fits LDS: the strided walk of the generic family is as fast — 25 x 10^6 x 4
    //  float64 + weights, 5000 bins: 0.50 ms against 0.87 with the gather; every other measured shape gains 1.3-9 x)
    if (table && n_rows <= 4 && dtype_size(samples[0].dtype) == 8 && p->n_bins * (weighted ? 8 : 4) <= (int64_t)96 * 1024) table = false;
    if (table) {
      void* scratch[kMaxDims + 1] = {nullptr};
      xhist_array dense[kMaxDims], dense_w;
      int rc = XHIST_OK;
      for (int d = 0; d <= D && rc == XHIST_OK; ++d) {
        if (d == D && !weighted) break;
        const xhist_array& a = d < D ? samples[d] : *weights;
        const int es = dtype_size(a.dtype);
        if (scratch_malloc(&scratch[d], (size_t)n_rows * n_cols * es, stream) != hipSuccess) {
          (void)hipGetLastError();
          rc = fail(XHIST_ERR_NOMEM, "allocation of %lld bytes of gather scratch failed", (long long)n_rows * n_cols * es);
          break;
        }
        const int grid = (int)std::min<int64_t>((n_cols + 255) / 256, (int64_t)p->cus * 16);
        // whole tables with rows of whole 16-byte units: tiled through LDS (256 rows x (k + 1) elements)
        const size_t tile_lds = (size_t)256 * (n_rows + 1) * es;
        if (a.col_stride == n_rows && ((int64_t)n_rows * es) % 16 == 0 && ((uintptr_t)a.data % 16) == 0 && es >= 4 && tile_lds <= 64 * 1024) {
          const int tgrid = (int)std::min<int64_t>((n_cols + 255) / 256, (int64_t)p->cus * (int64_t)std::max<size_t>(1, (160 * 1024) / tile_lds));
          if (es == 8) hipLaunchKernelGGL(gather_rows_tiled<uint64_t>, dim3(tgrid), dim3(256), tile_lds, stream, (const uint64_t*)a.data, (int)n_rows, n_cols, (uint64_t*)scratch[d]);
          else hipLaunchKernelGGL(gather_rows_tiled<uint32_t>, dim3(tgrid), dim3(256), tile_lds, stream, (const uint32_t*)a.data, (int)n_rows, n_cols, (uint32_t*)scratch[d]);
        } else
        switch (es) {
          case 8: hipLaunchKernelGGL(gather_rows<uint64_t>, dim3(grid), dim3(256), 0, stream, (const uint64_t*)a.data, a.col_stride, (int)n_rows, n_cols, (uint64_t*)scratch[d]); break;
          case 4: hipLaunchKernelGGL(gather_rows<uint32_t>, dim3(grid), dim3(256), 0, stream, (const uint32_t*)a.data, a.col_stride, (int)n_rows, n_cols, (uint32_t*)scratch[d]); break;
          case 2: hipLaunchKernelGGL(gather_rows<uint16_t>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)a.data, a.col_stride, (int)n_rows, n_cols, (uint16_t*)scratch[d]); break;
          default: hipLaunchKernelGGL(gather_rows<uint8_t>, dim3(grid), dim3(256), 0, stream, (const uint8_t*)a.data, a.col_stride, (int)n_rows, n_cols, (uint8_t*)scratch[d]); break;
        }
        if (hipGetLastError() != hipSuccess) rc = fail(XHIST_ERR_HIP, "gather_rows launch failed");
        xhist_array v = a;
        v.data = scratch[d];
        v.row_stride = n_cols;
        v.col_stride = 1;
        if (d < D) dense[d] = v; else dense_w = v;
      }
      if (rc == XHIST_OK) rc = execute_device(p, dense, weighted ? &dense_w : nullptr, n_rows, n_cols, out, accumulate, stream);
      for (auto* sc : scratch)
        if (sc) (void)scratch_free(sc, stream);
      return rc;
    }
  }

  // ---- family: fast (vector loads, homogeneous f64/f32) or generic --------------------------
  const size_t lds_cap = p->lds_max;
  const int sdt = samples[0].dtype;
  const int wdt = weighted ? weights->dtype : -1;
  int vec = 1;
  const bool float_samples = sdt == XHIST_F64 || sdt == XHIST_F32;
  const bool small_samples = sdt == XHIST_I32 || sdt == XHIST_I64 || sdt == XHIST_I16 || sdt == XHIST_U8 || sdt == XHIST_F16;
  // the exact int64 domain has vector kernels for one int64 / datetime64 input
  const bool i64dom = p->cmp == XHIST_CMP_I64 && !p->uns && sdt == XHIST_I64 && D == 1 && (wdt == -1 || wdt == XHIST_F64);
  bool fast_ok = !force_generic && (p->cmp == XHIST_CMP_F64 || i64dom) && p->n_bins < ((int64_t)1 << 31) &&
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

  if (fast_ok && !i64dom) {  // development switch: send homogeneous float inputs through the mixed-dtype kernels (A/B of the load path)
    static const bool force_mixed = [] { const char* e = getenv("XHIST_AMD_FORCE_MIXED"); return e && e[0] == '1'; }();
    if (force_mixed) fast_ok = false;
  }
  // Mixtures the homogeneous vector kernels do not take — float32 next to float64, integers in a joint histogram,
  // integer / bool / half weights — with unit column strides and the float64 compare domain: the MIXED variant of the
  // vector kernels (four elements per load in each array's own dtype, consumed as float64).  LDS histograms only;
  // beyond LDS the Python layer converts such inputs to float64 first (core._promote_for_big_histograms).
  bool mixed_ok = false;
  if (!fast_ok && !force_generic && !two && p->cmp == XHIST_CMP_F64 && D <= 3 && p->n_bins < ((int64_t)1 << 24) && !p->huge) {
    // element-aligned, and for 1- / 2-byte elements every row dword-aligned (their four-element loads are one dword / two)
    auto vector_loadable = [&](const xhist_array& a) {
      const size_t es = (size_t)dtype_size(a.dtype), al = es < 4 ? 4 : es;
      if (!(a.col_stride == 1 || n_cols == 1) || (uintptr_t)a.data % al != 0) return false;
      if (es >= 4 || n_rows == 1) return true;
      return (a.row_stride * es) % 4 == 0 && (a.outer_stride * es) % 4 == 0;
    };
    mixed_ok = true;
    for (int d = 0; d < D && mixed_ok; ++d) mixed_ok = vector_loadable(samples[d]);
    if (mixed_ok && weighted) mixed_ok = vector_loadable(*weights);
    // measured (tools/mixtures.py, 2 x 10^8 samples, against the generic family): float32 + int32 weights 1.06 -> 0.30 ms,
    // int32 x int32 1.17 -> 0.40, uint8 x float32 1.15 -> 0.37, float32 x float64 0.91 -> 0.46
  }

  // Two attempts: the vector family with its tables, then (if it has no kernel for this
  // combination, or its tables do not fit LDS) the generic family with the native tables.
  bool fast = false, use_f32 = false, tables_fit = false, lds_hist = false, tables_in_lds = false;
  int scan = 0, hist = kHistGlobal, cl2 = 0;
  const TableSet* tset = nullptr;
  size_t table_bytes = 0, hist_bytes = 0, lds_bytes = 0;
  kernel_fn fn = nullptr;
  const int acc_size = (weighted ? 8 : 4) * (two ? 2 : 1);  // two weights: two replicated histograms side by side
  const int max_cl2 = weighted ? 4 : 5;
  // histogram placement for a given table footprint:
  //   lds:    replicated sub-histograms in LDS (one copy per lane bank), uint32 / float64
  //   packed: unweighted vector family only, uint16 counters packed two per word (exact, see kernel)
  //   global: device-scope atomics straight into the output
  bool padded_bins = false;  // (hist_fast's PADDED layout; set before place() is asked about it)
  auto place = [&](size_t tbytes, bool vector_family) {
    hist = kHistGlobal;
    cl2 = 0;
    hist_bytes = 0;
    if (!force_global && tbytes + 1024 <= lds_cap && p->n_bins < ((int64_t)1 << 24)) {
      // replication is only worth LDS that small workgroups can share; unweighted 4-byte samples
      // issue twice the LDS atomics per byte streamed and gain 5 % from more copies (1-D, 500-2000 bins)
      const size_t soft = (!weighted && D == 1 && dtype_size(samples[0].dtype) <= 4 ? 64 : 24) * 1024;
      cl2 = max_cl2;
      if (lds_copies) { cl2 = 0; while ((1 << cl2) < lds_copies) ++cl2; cl2 = std::min(cl2, max_cl2); }
      // + 32 trash slots; the padded layout of the float32 arithmetic digitize (one input): a bin in front and one behind instead
      // (2 x max(1, 32 >> c) bins of 2^c copies: 64 slots for every c <= 5)
      auto bytes_at = [&](int c) { return (((size_t)p->n_bins << c) + (padded_bins ? (size_t)std::max(64, 2 << c) : 32)) * (size_t)acc_size; };
      if (!lds_copies) while (cl2 > 0 && bytes_at(cl2) > soft) --cl2;
      // short rows: a workgroup zeroes and reads back every copy, which must stay small next to the
      // samples it bins (10^5 rows x 1000 f32, 50 bins: 32 copies 0.162 ms, 4 copies 0.096 ms)
      if (!lds_copies) {
        int64_t sb = 0;
        for (int d = 0; d < D; ++d) sb += dtype_size(samples[d].dtype);
        if (weighted) sb += dtype_size(weights->dtype);
        const double wgs = (double)std::max<int64_t>(n_rows, (int64_t)p->cus * 8);
        const double per_wg_bytes = (double)n_rows * (double)n_cols * (double)sb / wgs;
        while (cl2 > 0 && (double)bytes_at(cl2) > per_wg_bytes / 4) --cl2;
      }
      while (cl2 > 0 && tbytes + bytes_at(cl2) > lds_cap) --cl2;
      if (tbytes + bytes_at(cl2) <= lds_cap) {
        hist = kHistLds;
        hist_bytes = bytes_at(cl2);
      } else if (vector_family && float_samples && !weighted && tbytes + (((size_t)p->n_bins + 1) / 2 + 32) * 4 <= lds_cap) {
        hist = kHistPacked;
        cl2 = 0;
        hist_bytes = (((size_t)p->n_bins + 1) / 2 + 32) * 4;
      }
    }
    if (hist == kHistGlobal) { cl2 = 0; hist_bytes = 0; }
  };
  bool mixed = false;
  for (int attempt = (fast_ok || mixed_ok) ? 0 : 1; attempt < 2 && !fn; ++attempt) {
    fast = attempt == 0;
    mixed = fast && !fast_ok;
    padded_bins = false;
    // float32 samples are digitized against the float32-threshold tables (exact, see Dom<2>)
    use_f32 = fast && !mixed && sdt == XHIST_F32 && p->ts[1][0].blob != nullptr;  // (mixed dtypes are consumed as float64)
    scan = 0;
    tset = &p->ts[0][0];  // generic family: native domain, (start, cnt) tables
    if (fast) tset = &pick_tables(p, use_f32, &scan);
    if (mixed && scan >= 3) {  // the mixed variant has no 3- / 4-edge scans: binary search on the (start, cnt) tables
      scan = 0;
      tset = &p->ts[0][0];
    }
    table_bytes = (size_t)tset->words * 8;
    tables_fit = table_bytes + 1024 <= lds_cap && !(fast && p->huge);  // no bucket tables: not for the vector family
    if (tables_fit || !fast) place(table_bytes, fast);
    // Arithmetic edges (bins=int, np.linspace): when the edge tables are what keeps the histogram
    // out of LDS — or do not fit LDS at all — digitize without tables (count_le_arith): 30000
    // uniform bins stay on the streaming kernels instead of 43 ms/10^9 samples of global atomics.
    // Also when the tables fit but only with 3-4 edges per bucket (float32, 20000 bins: 1.17 against 1.39 ms);
    // with 1-2 edges per bucket the tables win (C2: 2.28 against 2.40 ms, float32 50 bins: 0.69 against 1.12).
    // (Round 3: the arithmetic digitize needs no edge any more unless the sample is next to one — bin_arith_fast, 4-19 % off
    // every table-free launch — and on one box it beat the one-edge-per-bucket tables for float64 samples too, 1.33 -> 1.22 ms
    // for the unweighted headline; on another the two were level (1.19 / 1.21), 10^6 samples went 8.8 -> 12.7 us and a 20^3
    // float64 histogram 1.97 -> 2.06 ms: the rule stays as it was.  profiles/r03_ar_arith_ab.txt)
    if (fast && (float_samples || mixed) && p->arith && arith_pref >= 0 &&
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
    // float32 samples on arithmetic edges with an LDS histogram: the bin from float32 arithmetic, no table at all
    // (bin_arith32_fast: 7-9 vector instructions and one LDS operation per sample against 12 and three for the threshold
    // tables — BASELINE C4, whose dask-chunk-sized calls run while the clock of a just-woken GPU dips)
    // ("arith" = 1 asks for the float64 arithmetic by name and keeps it; "arith" = -1 rules out both)
    if (fast && !mixed && !two && !i64dom && sdt == XHIST_F32 && p->arith32 && arith32_pref >= 0 && arith_pref >= 0 &&
        (arith32_pref > 0 || (arith_pref == 0 && (scan == 1 || scan == 2 || scan == kScanArith)))) {
      const int h0 = hist, c0 = cl2;
      const size_t b0 = hist_bytes;
      padded_bins = D == 1;
      place(0, true);
      // (no second try without the padding: the kernel's PADDED layout is a compile-time property of <kScanArith32, D == 1, LDS>,
      //  so a one-input histogram that fits LDS only unpadded — n_bins within 64 of the cap — keeps the digitize it had)
      if (hist == kHistLds) {
        scan = kScanArith32;
        use_f32 = false;
        tset = &p->ts[0][0];  // float64-domain DimTable (e_0, e_last, step, a32_*); its tables are not read
        table_bytes = 0;
        tables_fit = true;
      } else {
        padded_bins = false;
        hist = h0; cl2 = c0; hist_bytes = b0;
      }
    }
    // Non-uniform edges, float64 samples: packed 16-byte bucket entries (count_le_pack) — one LDS read per sample and
    // dimension instead of two dependent ones — where the histogram stays in LDS with them.  Measured for joint histograms
    // (C3: see DESIGN 4); 1-D histograms keep the two-level tables unless "pack" = 1.
    const int pk_np = sdt == XHIST_F64 ? p->pk_np : (sdt == XHIST_F32 && use_f32 ? p->pk32_np : 0);
    if (fast && !mixed && !two && !i64dom && pk_np && pack_pref >= 0 && scan != kScanArith && scan != kScanArith32 && tables_fit &&
        (scan == 0 || scan >= 2 || pack_pref > 0) && (D >= 2 || scan == 0 || (sdt == XHIST_F32 && !weighted) || pack_pref > 0)) {
      // (scan 0: the alternative is a binary search; 1-D: measured level for float64 samples and for weighted float32 ones,
      //  0.74 -> 0.85 of 8 TB/s for float32 counts — twice the table reads per byte streamed: profiles/r04_h_*)
      const TableSet& pk = sdt == XHIST_F64 ? p->ts_pk : p->ts_pk32;
      const int h0 = hist, c0 = cl2;
      const size_t b0 = hist_bytes, tb = (size_t)pk.words * 8;
      if (tb + 1024 <= lds_cap) {
        place(tb, true);
        if ((hist == kHistLds || hist == kHistPacked) && (hist == h0 || pack_pref > 0)) {
          scan = pk_np == 2 ? kScanPack2 : (pk_np == 3 ? kScanPack3 : kScanPackG);
          tset = &pk;
          table_bytes = tb;
        } else {
          hist = h0; cl2 = c0; hist_bytes = b0;
        }
      }
    }
    if (fast && !tables_fit) continue;  // the vector family keeps its tables in LDS
    if (mixed && hist != kHistLds) continue;  // (packed / memory-side histograms: the generic family)
    lds_hist = hist == kHistLds;
    tables_in_lds = tables_fit;
    lds_bytes = (tables_in_lds ? table_bytes : 0) + hist_bytes;
    // (scan 1..4: linear in-bucket count, no bucket holds more than 4 edges — always for uniform bins)
    if (two) {  // one attempt only: the vector family, LDS histograms, table digitize with <= 2 edges per bucket
      if (!fast || hist != kHistLds || (scan != 1 && scan != 2) || !float_samples || weights2->dtype != wdt ||
          weights2->col_stride != 1 || ((uintptr_t)weights2->data % (size_t)dtype_size(wdt)) != 0)
        return XHIST_ERR_UNSUPPORTED;
      fn = fast_kernel_two_weights(sdt, wdt, D, scan, &vec);
      if (!fn) return XHIST_ERR_UNSUPPORTED;
      break;
    }
    if (mixed) {
      fn = xhist_pick_mixed(weighted, D, scan);
      vec = 4;
      continue;  // (no kernel: the loop goes on to the generic family)
    }
    fn = fast ? (i64dom ? int64_domain_kernel(wdt, D, scan, hist, &vec) : fast_kernel(sdt, wdt, D, scan, hist, &vec))
              : generic_kernel(p->cmp, weighted, lds_hist, tables_in_lds);
  }
  // One input without weights, long rows: twice the samples per lane and tile — the 128 bytes per lane in flight that the
  // weighted kernel has with its two streams.  float64, one row of 10^9 samples, 100 bins (the headline's 8 B/sample variant):
  // 1.20-1.24 -> 1.14-1.16 ms (0.83 -> 0.86-0.88 of 8 TB/s); float32, BASELINE C4 (456 / 3650 rows of 1 036 800): 0.2872 ->
  // 0.2838 ms (0.823 -> 0.833) and 2.228 -> 2.208 ms (0.849 -> 0.857), profiles/r03_u8_unroll8.txt.  Short rows keep their tiles
  // (the geometry rules further down were fitted to them); float64 with many rows was not measured and stays as it was.
  bool long_tiles = false;
  if (fn && fast && !mixed && !two && !i64dom && !weighted && D == 1 && hist == kHistLds && (scan == 1 || scan == 2 || scan == kScanArith32) && !block_threads && !grid_blocks) {
    kernel_fn lf = nullptr;
    // (exactly the two shape classes that were measured: one float32 row of 3*10^7 ... 10^9 samples came out 4-10 % SLOWER
    // with the long tiles in tools/size_ramp.py — its one-workgroup-per-CU geometry already has 64 KiB per CU in flight)
    if (sdt == XHIST_F64 && n_rows == 1 && n_cols >= ((int64_t)1 << 29)) lf = xhist_pick_f64_long(scan);
    else if (sdt == XHIST_F32 && n_rows >= 64 && n_cols >= ((int64_t)1 << 19)) lf = xhist_pick_f32_long(scan);
    if (lf) {
      fn = lf;
      long_tiles = true;
    }
  }
  if (two && !accumulate) {
    if (int zrc = zero_output(out, out_elems, stream)) return zrc;
    if (int zrc = zero_output(out2, out_elems, stream)) return zrc;
  }
  if (!fn) return fail(XHIST_ERR_HIP, "internal: no kernel for this combination");
  if (!fast) vec = 1;
  const DimTable* dims = tset->dim;

  // ---- partitions of the partitioned mode (below; the slice rule prices it) ---------------------------
  // 2^14 float64 or 2^15 uint32 bins = 128 KiB of LDS per partition, unless that leaves only a handful:
  // A handful of partitions is slower than a few dozen: the routing pass ranks a tile's records with ONE returning LDS
  // counter per partition, and a tile's 8192 adds on two or three addresses serialise (a tile lies in one row, so rows
  // that share a pass do not help).  5*10^8 samples, ms: float32 pairs + weights 160 x 160 bins, 2 partitions -> 13:
  // 3.04 -> 2.32; 200 x 200, 3 -> 20: 2.82 -> 2.35; float32 pair counts 300 x 300, 3 -> 22: 2.75 -> 1.58; 400 x 400,
  // 5 -> 20: 2.33 -> 1.60; float64 pair counts 300 x 300 2.78 -> 2.36; 10^5 bins 1-D, 4 -> 25: 2.65 -> 1.53; 12 rows x
  // 200 x 200 float32 weighted, 3 -> 10 per row: 2.89 -> 2.43 (profiles/r03_t_few_partitions.txt).  So the bins are cut
  // finer — down to 2^11 per partition — until a row has 16 partitions ("min_parts" overrides).  float64 samples with
  // float64 weights, whose tiles are half as long, gain nothing (4.10 -> 4.06, 12 rows 4.09 -> 4.34) and keep the big ones.
  auto part_geometry = [&](int& shift, int64_t& n_parts) {
    shift = weighted ? 14 : 15;
    n_parts = (p->n_bins + ((int64_t)1 << shift) - 1) >> shift;
    if (weighted && sdt == XHIST_F64 && wdt == XHIST_F64 && !p->min_parts) return;
    const bool rows_share = n_rows > 1 && n_cols < ((int64_t)1 << 27);  // (uniform_rows below: several rows per pass)
    const int64_t want_parts = p->min_parts ? p->min_parts : 16;
    while (shift > 11 && n_parts < want_parts) {
      const int64_t finer = (p->n_bins + ((int64_t)1 << (shift - 1)) - 1) >> (shift - 1);
      // rows that fit ONE pass with 8+ partitions each stay in one pass (a pass costs ~0.1 ms beyond its traffic)
      if (rows_share && n_parts >= 8 && n_rows * n_parts <= 128 && n_rows * finer > 128) break;
      --shift;
      n_parts = finer;
    }
  };

  // ---- a few times the LDS capacity: bin slices ---------------------------------------------------
  // S launches, each streaming all samples and keeping 1/S of the bins in LDS (hist_fast<SLICED>),
  // cost S x the streaming time; the partitioned mode below moves B + 2 x record bytes per sample once and needs a few
  // long rows.  Round 3, 5*10^8 samples, slices | partitioned ms (profiles/r03_t_few_partitions.txt): float32 pairs +
  // float32 weights S = 2: 1.80 | 2.32, S = 3: 2.67 | 2.35, S = 4: 3.53 | 2.21; float64 pairs + float64 weights S = 2:
  // 3.47 | 4.03, S = 3: 5.21 | 4.06; float32 pair counts (packed uint16 slices) S = 2: 1.73 | 1.58, S = 3: 2.62 | 1.60;
  // float64 pair counts S = 2: 2.43 | 2.36, S = 3: 3.69 | 2.38.  (Round 1's rule — written when the partitioned mode took
  // three passes, 9.3 ms for 200 x 200 weighted bins — kept slices up to S = 3-4.)
  int n_slices = 1;
  int64_t slice_bins = p->n_bins;
  if (fast && float_samples && hist == kHistGlobal && !force_global && !two && slices_pref >= 0 && p->n_bins < ((int64_t)1 << 24) &&
      (scan == 1 || scan == 2 || scan == kScanArith) && table_bytes + 4096 < lds_cap) {
    const size_t budget = lds_cap - table_bytes - 2048;
    const int64_t cap = weighted ? (int64_t)(budget / 8) - 32 : (((int64_t)(budget / 2) - 66) & ~(int64_t)1);
    const int64_t S = (p->n_bins + cap - 1) / cap;
    int64_t B = 0;
    for (int d = 0; d < D; ++d) B += dtype_size(samples[d].dtype);
    if (weighted) B += dtype_size(weights->dtype);
    const int64_t rec = 2 + (weighted ? (wdt == XHIST_F32 ? 4 : 8) : 0);  // a record: uint16 code + the weight in its own precision
    int part_shift = 0;
    int64_t parts_per_row = 0;
    part_geometry(part_shift, parts_per_row);
    const bool part_ok = partition >= 0 && n_rows <= 64 && n_cols >= ((int64_t)1 << 22) && parts_per_row <= kPartMaxParts;
    // per sample: a slice pass streams B bytes at ~6.5 TB/s (packed uint16 counters: no faster than 5.7 x 10^11 samples
    // a second); the partitioned mode moves B + 2 x record bytes at ~4.7 TB/s and no faster than 3.3 x 10^11 samples a second
    const double per_slice = std::max((double)B / 6.5e12, weighted ? 0.0 : 1.0 / 5.7e11);
    const double per_part = std::max(((double)B + 2.0 * (double)(weighted ? std::min<int64_t>(rec, 8) : rec)) / 4.7e12, 1.0 / 3.3e11);
    // (+ ~0.1 ms per routing / adding-up pass: rows share a pass while rows x partitions <= 128)
    const double n_all = (double)n_rows * (double)n_cols;
    const int64_t rows_per_pass = (n_rows > 1 && n_cols < ((int64_t)1 << 27) && parts_per_row * 2 <= 128) ? std::max<int64_t>(1, 128 / parts_per_row) : 1;
    const double t_part = n_all * per_part + (double)((n_rows + rows_per_pass - 1) / rows_per_pass) * 1.0e-4;
    // every slice pass ends with one global atomic per non-empty bin and workgroup (~2 x 10^11 a second for the chip) and costs
    // ~13 us to launch: 10^7 float64 pairs into 512 x 512 counts, 5 slices 0.377 ms against 0.112 partitioned
    const double slice_wgs = (double)std::max<int64_t>(p->cus, n_rows);
    const double slice_bins_each = (double)((p->n_bins + S - 1) / S);
    const double t_slice_fixed = 13e-6 + slice_wgs * std::min(slice_bins_each, n_all / slice_wgs) / 2.0e11;
    bool choose = slices_pref > 0 ? S <= 64 : (partition > 0 ? false : (part_ok ? (double)S * (n_all * per_slice + t_slice_fixed) <= t_part : S <= 16));
    if (choose && slices_pref == 0) {
      // few samples: S launches cost S x ~13 us before they stream anything, memory-side atomics 2.4-2.7 x 10^10
      // per second (10^5 samples, 256 x 256 weighted bins: 4 slices 54 us, global atomics 13 us)
      const double n_tot = (double)n_rows * (double)n_cols;
      const double t_global = 8e-6 + n_tot / (weighted ? 2.4e10 : 2.7e10);
      const double t_slices = (double)S * (13e-6 + n_tot * (double)B / 4.0e12);
      choose = t_slices <= t_global;
    }
    if (choose && S >= 1) {
      const int shist = weighted ? kHistLds : kHistPacked;
      kernel_fn sfn = fast_kernel_sliced(sdt, wdt, D, scan, shist, &vec);
      if (sfn) {
        fn = sfn;
        hist = shist;
        cl2 = 0;
        n_slices = (int)S;
        slice_bins = ((p->n_bins + S - 1) / S + 1) & ~(int64_t)1;  // even: packed counters pair up inside a slice
        hist_bytes = weighted ? ((size_t)slice_bins + 32) * 8 : (((size_t)slice_bins + 1) / 2 + 32) * 4;
        lds_hist = hist == kHistLds;
        tables_in_lds = true;
        lds_bytes = table_bytes + hist_bytes;
      }
    }
  }

  // ---- histograms beyond LDS: partitioned multi-pass instead of memory-side atomics ----------
  // (a few long rows — e.g. one joint histogram per time step — run it row by row)
  if (fast && float_samples && hist == kHistGlobal && !force_global && partition >= 0 && n_rows <= 64 && !two && n_slices == 1) {
    int shift = 0;
    int64_t n_parts = 0;
    part_geometry(shift, n_parts);
    const bool big_enough = n_cols >= ((int64_t)1 << 22) || (partition > 0 && n_cols >= 4);  // part_scatter reads whole weight quads
    if (n_parts <= kPartMaxParts && big_enough && (size_t)(1u << shift) * (weighted ? 8 : 4) + 1024 <= lds_cap) {
      if (!accumulate)
        if (int zrc = zero_output(out, out_elems, stream)) return zrc;
      LaunchRecord rec(p, stream);
      int rc = XHIST_OK;
      // the routing pass keeps the digitize tables in LDS next to its sort buffers; where they do not fit (or cost a
      // workgroup per CU) and the edges are arithmetic (C5: 2 x 1025 np.linspace edges, 32 KiB of tables) it
      // digitizes table-free
      int r_scan = scan;
      const TableSet* r_tset = tset;
      bool r_f32 = use_f32;
      const bool can_arith = p->arith && arith_pref >= 0 && n_parts <= 128;
      const int w_tag = weighted ? wdt : -1;
      // geometry first: the long tile with the arithmetic digitize where the edges allow it, else with the tables if both
      // fit the LDS, else the short tile
      RouteGeom geom = route_geom_for(p, sdt, w_tag, D, can_arith ? kScanArith : scan);
      if (geom.spl == 8 && (can_arith || scan == kScanArith)) {
        if (scan != kScanArith) {
          r_scan = kScanArith;
          r_tset = &p->ts[0][0];
          r_f32 = false;
        }
      } else {
        if (geom.spl == 8 && !p->route_spl &&
            part_route_lds((size_t)tset->words * 8, (int)n_parts, weighted, route_tile(geom.block, 8), geom.block) > p->lds_max)
          geom.spl = 4;
        if (can_arith && scan != kScanArith) {
          const int t4 = route_tile(geom.block, geom.spl);
          const size_t lds_tab = part_route_lds((size_t)tset->words * 8, (int)n_parts, weighted, t4, geom.block),
                       lds_notab = part_route_lds(0, (int)n_parts, weighted, t4, geom.block);
          if (lds_tab > p->lds_max || (size_t)160 * 1024 / lds_notab > (size_t)160 * 1024 / lds_tab) {
            r_scan = kScanArith;
            r_tset = &p->ts[0][0];
            r_f32 = false;
          }
        }
      }
      // non-uniform edges: the packed entries (general variant) instead of a binary search or a 2-4-edge scan per sample and
      // dimension — the routing pass feels its digitize (5*10^8 float64 pairs into 512 x 512 counts: linspace 2.07 ms, random
      // edges 2.48, geometric 3.42; profiles/r04_l_*) — where entries and sort buffers fit the LDS together
      {
        const int pk_np = sdt == XHIST_F64 ? p->pk_np : (sdt == XHIST_F32 && use_f32 ? p->pk32_np : 0);
        if (pk_np && pack_pref >= 0 && r_scan != kScanArith && (r_scan == 0 || r_scan >= 2 || pack_pref > 0) && n_parts <= 128) {
          const TableSet& pk = sdt == XHIST_F64 ? p->ts_pk : p->ts_pk32;
          int spl = geom.spl;
          if (spl == 8 && part_route_lds((size_t)pk.words * 8, (int)n_parts, weighted, route_tile(geom.block, 8), geom.block) > p->lds_max) spl = 4;
          if (part_route_lds((size_t)pk.words * 8, (int)n_parts, weighted, route_tile(geom.block, spl), geom.block) <= p->lds_max &&
              route_kernel(sdt, w_tag, D, kScanPackG, false, geom.block, spl)) {
            r_scan = kScanPackG;
            r_tset = &pk;
            geom.spl = spl;
          }
        }
      }
      const int r_block = geom.block, r_tile = route_tile(r_block, geom.spl);
      // rows that follow each other at one stride go through the routing pass several at a time
      // (32 x 3*10^7 float32 pairs + weights, 300 x 300 bins: 8.05 -> 6.02 ms; 8 x 6*10^7 float64, 512 x 512: 5.17 -> 4.18;
      // rows of 5*10^8 samples gain nothing — 8.63 -> 8.95 ms with 128 partitions in flight — and stay one per pass)
      // a single row of float64 samples with float64 weights below ~1.5 x 10^7 samples: the routing pass's fixed costs (chunk
      // pool, lists, one partial chunk per workgroup and partition) show, and count + prefix + scatter is 9-15 % faster
      // (5*10^6: 0.155 | 0.132 ms, 10^7: 0.191 | 0.174, 2*10^7: 0.273 | 0.269: profiles/r03_h_fused_vs_three_pass_mid_sizes.txt);
      // every other dtype combination is ahead with one pass from 5*10^6 samples on
      const bool three_pass = fused_pref == 0 && partition == 0 && n_rows == 1 && weighted && sdt == XHIST_F64 && wdt == XHIST_F64 && n_cols < 15000000;
      const int fused_use = three_pass ? -1 : fused_pref;
      bool uniform_rows = fused_use >= 0 && n_rows > 1 && n_parts * 2 <= 128 && n_cols < ((int64_t)1 << 27) && (!weighted || weights->inner_rows == 0);
      for (int d = 0; d < D; ++d) uniform_rows &= samples[d].inner_rows == 0;
      static const bool batch_off = [] { const char* e = getenv("XHIST_AMD_ROW_BATCH"); return e && *e == '0'; }();
      if (batch_off) uniform_rows = false;
      int64_t r = 0;
      while (uniform_rows && r < n_rows && rc == XHIST_OK) {
        int rows = (int)std::min<int64_t>(n_rows - r, 128 / n_parts);
        const size_t tab_bytes = r_scan == kScanArith ? 0 : (size_t)r_tset->words * 8;
        while (rows > 1 && part_route_lds(tab_bytes, rows * (int)n_parts, weighted, r_tile, r_block) > p->lds_max) --rows;
        if (rows < 2) break;  // (one row at a time below)
        xhist_array row_s[XHIST_MAX_DIMS], row_w;
        for (int d = 0; d < D; ++d) {
          row_s[d] = samples[d];
          row_s[d].data = const_cast<void*>(advance(samples[d].data, samples[d].dtype, r * samples[d].row_stride));
        }
        if (weighted) {
          row_w = *weights;
          row_w.data = const_cast<void*>(advance(weights->data, weights->dtype, r * weights->row_stride));
        }
        void* row_out = static_cast<char*>(out) + (size_t)r * p->n_bins * 8;
        rc = execute_partitioned_fused(p, row_s, weighted ? &row_w : nullptr, n_cols, row_out, stream, sdt, wdt, r_scan, r_f32, *r_tset, shift,
                                       (int)n_parts, profile, rec, r == 0, r + rows == n_rows, geom, rows);
        if (rc == XHIST_ERR_UNSUPPORTED) {
          if (r > 0) rc = fail(XHIST_ERR_HIP, "internal: partitioned mode refused rows from %lld after accepting row 0", (long long)r);
          else rc = XHIST_OK;  // nothing launched: row by row below
          break;
        }
        r += rows;
      }
      for (; r < n_rows && rc == XHIST_OK; ++r) {
        xhist_array row_s[XHIST_MAX_DIMS], row_w;
        for (int d = 0; d < D; ++d) {
          row_s[d] = samples[d];
          row_s[d].data = const_cast<void*>(advance(samples[d].data, samples[d].dtype,
                                                    row_offset(r, samples[d].row_stride, samples[d].inner_rows, samples[d].outer_stride)));
        }
        if (weighted) {
          row_w = *weights;
          row_w.data = const_cast<void*>(advance(weights->data, weights->dtype,
                                                 row_offset(r, weights->row_stride, weights->inner_rows, weights->outer_stride)));
        }
        void* row_out = static_cast<char*>(out) + (size_t)r * p->n_bins * 8;
        // one routing pass (44 B per C5 sample) where it applies, else count + prefix + scatter (52.5 B); the
        // choice depends on the plan and the dtypes only, so every row of a call takes the same route
        rc = fused_use >= 0 ? execute_partitioned_fused(p, row_s, weighted ? &row_w : nullptr, n_cols, row_out, stream, sdt, wdt, r_scan,
                                                         r_f32, *r_tset, shift, (int)n_parts, profile, rec, r == 0, r == n_rows - 1, geom)
                             : XHIST_ERR_UNSUPPORTED;
        if (rc == XHIST_ERR_UNSUPPORTED)
          rc = execute_partitioned(p, row_s, weighted ? &row_w : nullptr, n_cols, row_out, stream,
                                   sdt, wdt, scan, use_f32, *tset, shift, (int)n_parts, profile, rec, r == 0, r == n_rows - 1);
        if (rc == XHIST_ERR_UNSUPPORTED && r > 0) rc = fail(XHIST_ERR_HIP, "internal: partitioned mode refused row %lld after accepting row 0", (long long)r);
      }
      if (rc != XHIST_ERR_UNSUPPORTED) return rc;  // UNSUPPORTED (from row 0, nothing launched) = fall through to global atomics
      accumulate = 1;  // the output has just been zeroed
    }
  }
  const int kUnroll = long_tiles ? 8 : (mixed ? mixed_unroll(D) : (fast ? unroll_for(D, vec, scan) : 1));

  // ---- geometry -----------------------------------------------------------------------------
  // Workgroups per CU are sized by bytes in flight, not by occupancy: measured on MI355X
  // (profiles/r01_a_sweep.jsonl) the streaming rate peaks at ~64 KiB of outstanding loads per CU
  // (f64+weights: 2 x 256 threads x 128 B; f64: 4 x 256 x 64 B) and falls by 5-10% with more.
  int64_t lane_bytes = 0;
  for (int d = 0; d < D; ++d) lane_bytes += dtype_size(samples[d].dtype);
  if (weighted) lane_bytes += dtype_size(weights->dtype);
  if (two) lane_bytes += dtype_size(weights->dtype);
  lane_bytes *= fast ? (int64_t)vec * kUnroll : 4;
  int block = block_threads ? block_threads : 256;
  if (!block_threads && lds_bytes > 40 * 1024) {
    // one to three workgroups fit a CU: make them carry the ~64 KiB of loads in flight together
    // (2 x f64 with 128 B per lane: 512 threads; 2 x f32 with 64 B per lane and 88 KB of LDS:
    // 1024 — 4.85 TB/s at 512).  The packed-uint16 histogram waits on returning atomics and wants
    // half as much again (C3: 768).
    const int fit = (int)std::max<int64_t>(1, std::min<int64_t>(3, (int64_t)(160 * 1024 / lds_bytes)));
    int64_t want = (64 * 1024) / (fit * std::max<int64_t>(lane_bytes, 1));
    if (hist == kHistPacked) want += want / 2;
    block = (int)std::min<int64_t>(1024, std::max<int64_t>(256, (want + 255) / 256 * 256));
  }
  if (!block_threads && n_rows > 1 && lds_bytes <= 40 * 1024) {
    // many rows, one workgroup each: a tile should be ~1/4 of the row or most of the workgroup
    // idles in the ragged tile (100k rows x 3650: 0.46 -> 0.39 ms; 356k x 1024: 1.3 -> 0.58 ms)
    const int64_t per_lane = fast ? (int64_t)vec * unroll_for(D, vec, scan) : 4;
    int64_t want = n_cols / (4 * per_lane);
    block = 64;
    while (block < 256 && block * 2 <= want) block *= 2;
  }
  // (generic family, long rows: 512-thread workgroups — 2 x 10^8 f32 samples 0.74 -> 0.61 ms)
  if (!block_threads && !fast && n_cols >= 65536 && lds_bytes <= 40 * 1024) block = 512;
  int bpc = (int)std::max<int64_t>(1, std::min<int64_t>(8, (64 * 1024 + block * lane_bytes / 2) / (block * lane_bytes)));
  // Many rows: a workgroup is tied to one row, so the tail is balanced by OVERSUBSCRIBING the chip 8x
  // with small workgroups (C4 shape: 5.6 -> 6.5 TB/s) — as long as their flushes are cheap: every
  // workgroup ends with one global atomic per non-empty bin (runs of consecutive bins: ~2*10^11 per
  // second for the chip), and the workgroups of one row meet on the same addresses (~25 ns each).  The
  // factor is halved until the flushes would take under 20 % of the streaming time; a few rows are one
  // row in a few parts and never oversubscribe.
  // (2 x 5*10^7 f32, 50 bins: 8192 workgroups 149 us, 2 x 128 of 1024 threads 65 us;
  //  32 x 312500 f64, 1000 bins: 4896 workgroups 45 us, 32 x 8 of 512 threads 19 us.)
  const int64_t sample_bytes = lane_bytes / (fast ? (int64_t)vec * kUnroll : 4);
  const double total_samples = (double)n_rows * (double)n_cols;
  // (C4 shard, 456 rows: 18 workgroups per row 0.2875 ms, 36: 0.2839 — and an ODD number per row is 2.5-7 % slower than its
  // even neighbours: 17 / 19 / 21 / 23 / 27 / 31 per row 0.2946 / 0.2942 / 0.2980 / 0.2965 / 0.3067 / 0.3013 ms against
  // 0.2854-0.2876 for 16 ... 28, profiles/r03_n_c4_segs_per_row.txt: the workgroups of a row walk its 16 KiB tiles interleaved,
  // and an odd stride between a workgroup's tiles lands on the memory channels worse.  Hence 16 x, and an even count below.)
  static const bool segs_legacy = [] { const char* e = getenv("XHIST_AMD_SEGS_LEGACY"); return e && *e == '1'; }();  // A/B switch
  int over = n_rows <= 16 ? 1 : (segs_legacy ? 8 : 16);
  if (!grid_blocks && over > 1) {
    const double t_stream = total_samples * (double)sample_bytes / 6.0e12;
    while (over > 1) {
      const double wgs = std::max<double>((double)n_rows, (double)p->cus * bpc * over);
      const double dense = wgs * std::min<double>((double)p->n_bins, total_samples / wgs) / 2.0e11;
      const double serial = wgs / (double)n_rows * 25e-9;
      if (std::max(dense, serial) <= 0.2 * t_stream) break;
      over /= 2;
    }
  }
  const bool exact = over == 1;  // exactly the workgroups that are resident at once, a whole number per row
  if (!block_threads && !grid_blocks && exact && n_cols >= 65536 && fast && hist == kHistLds && lds_bytes <= 40 * 1024) {
    // Small histogram, long rows: ONE workgroup per CU, as wide as the loads in flight ask for.  Measured
    // (tools/size_ramp.py, profiles/r01_u_*): ~32 KiB per CU for 8-byte samples with the one-compare
    // digitize (f64, uniform-style edges: 512 threads), ~64 KiB for 4-byte samples, which do twice the LDS
    // atomics per byte, and for every heavier digitize (f32, non-uniform edges, int64 domain: 1024; 8-byte
    // geometric edges at 512 threads: 300 us per 10^8, at 1024: 234).  8-byte samples with 128 B per lane
    // (f64 + f64 weights, two f64 inputs) only below 256 MB — above, two
    // 256-thread workgroups per CU are 15-20 % ahead.  The block shrinks until there is a tile for half
    // the CUs, and to whatever size lets a whole number of workgroups per row fill the chip best.
    int64_t ssz = 0;
    for (int d = 0; d < D; ++d) ssz = std::max<int64_t>(ssz, dtype_size(samples[d].dtype));
    int64_t threads_per_cu = (int64_t)bpc * 256;  // (block is 64..256 here, bpc was sized for 256)
    block = 256;
    if (mixed || !(lane_bytes >= 128 && ssz >= 8 && total_samples * (double)sample_bytes > (double)((int64_t)256 << 20))) {
      int64_t per_cu = (ssz >= 8 && scan == 1 && float_samples ? 32 : 64) * 1024 / std::max<int64_t>(lane_bytes, 1);
      // the mixed-dtype kernels convert after loading and want the whole CU: float32 x float64, 2 x 10^8 samples:
      // 256 threads per CU 0.88 ms, 512 0.52, 1024 0.44 (tools/mixtures.py)
      if (mixed) per_cu = 1024;
      threads_per_cu = std::min<int64_t>(1024, std::max<int64_t>(256, per_cu / 256 * 256));
      block = (int)threads_per_cu;
      while (block > 256 && n_rows * (n_cols / ((int64_t)block * vec * kUnroll)) < p->cus / 2) block -= 256;
      if (block == 768) block = 512;
    }
    const int64_t slots = (int64_t)p->cus * threads_per_cu;  // threads the launch should keep resident
    int best = 256;
    int64_t best_used = -1;
    for (int b = block; b >= 256; b /= 2) {
      const int64_t used = n_rows * (slots / (n_rows * b)) * b;
      if (used > best_used + best_used / 16) { best_used = used; best = b; }  // a smaller block must fill 6 % more
    }
    if (best_used > 0) {
      block = best;
      bpc = (int)std::max<int64_t>(1, threads_per_cu / block);
    } else {
      block = 256;  // more rows than the chip holds workgroups: one 256-thread workgroup each
    }
  }
  bpc = std::min<int>(bpc, 2048 / block);
  if (lds_bytes) bpc = std::max<int>(1, std::min<int64_t>(bpc, (int64_t)(160 * 1024 / lds_bytes)));
  int64_t target = grid_blocks ? grid_blocks : (int64_t)p->cus * bpc * over;
  if (!grid_blocks && exact) {
    // small inputs: streaming gains ~25 GB/s per 256 threads, every workgroup of a row costs ~12 ns at the
    // end.  The sum of the two is minimal at sqrt(bytes / (25 GB/s * 12 ns)) workgroups per row
    // (10^6 f64 samples: 18 -> 9 us)
    // A big histogram raises the second term: a workgroup flushes min(bins, its samples) counters at
    // ~2*10^11 per second (256 x 256 packed counters, 10^7 samples: 256 workgroups 117 us, 96: 94 us).
    const double bytes = (double)n_cols * (double)sample_bytes;
    const double rate = 25e9 * block / 256;
    const int64_t cap = std::max<int64_t>(1, target / n_rows);
    int64_t per_row = std::max<int64_t>(1, std::min<int64_t>(cap, (int64_t)std::sqrt(bytes / (rate * 12e-9))));
    const double flushed = std::min<double>((double)(n_slices > 1 ? slice_bins : p->n_bins), (double)n_cols / (double)per_row);
    const double per_wg = std::max(12e-9, flushed / 2.0e11);
    per_row = std::max<int64_t>(1, std::min<int64_t>(cap, (int64_t)std::sqrt(bytes / (rate * per_wg))));
    target = per_row * n_rows;  // a whole number per row, never more than are resident at once
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
  // One workgroup per row (many short rows): its LDS histogram IS the row's result, so it is stored
  // with plain writes — no zeroing pass over the output, no global atomics (10^5 rows x 1000 f32,
  // 50 bins: the 5 x 10^6 flush atomics alone took the whole 0.18 ms).
  int64_t segs0 = std::max<int64_t>(1, std::min<int64_t>((std::min(col_chunk, n_cols) + tile - 1) / tile, (target + n_rows - 1) / n_rows));
  if (!segs_legacy && !exact && !grid_blocks && segs0 > 1 && (segs0 & 1) && segs0 + 1 <= (std::min(col_chunk, n_cols) + tile - 1) / tile) ++segs0;
  const bool direct = !accumulate && !two && fast && hist == kHistLds && n_slices == 1 && col_chunk >= n_cols && segs0 == 1 &&
                      n_rows <= kMaxGrid;
  if (!accumulate && !direct && !two)
    if (int zrc = zero_output(out, out_elems, stream)) return zrc;
  bool first_launch = true;
  LaunchRecord rec(p, stream);
  char desc[384];
  for (int64_t c0 = 0; c0 < n_cols; c0 += col_chunk) {
    const int64_t nc = std::min(col_chunk, n_cols - c0);
    const int64_t tpr = (nc + tile - 1) / tile;
    for (int64_t r0 = 0; r0 < n_rows;) {
      int64_t segs = std::max<int64_t>(1, std::min<int64_t>(tpr, (target + (n_rows - r0) - 1) / (n_rows - r0)));
      if (!segs_legacy && !exact && !grid_blocks && segs > 1 && (segs & 1) && segs + 1 <= tpr) ++segs;  // oversubscribed rows: an even number each
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
      if (two) {
        kp.w2_ptr = advance(weights2->data, weights2->dtype, c0);
        kp.w2_rs = weights2->row_stride;
        kp.w2_ir = weights2->inner_rows;
        kp.w2_os = weights2->outer_stride;
        kp.out2 = static_cast<char*>(out2) + (size_t)r0 * p->n_bins * 8;
      }
      kp.row0 = r0;
      kp.n_dims = D;
      kp.tables = tset->blob;
      kp.table_words = (scan == kScanArith || scan == kScanArith32) ? 0 : tset->words;  // arithmetic edges: nothing to stage
      kp.tables_in_lds = tables_in_lds ? 1 : 0;
      kp.n_rows = nr;
      kp.n_cols = nc;
      kp.n_bins = p->n_bins;
      kp.out = static_cast<char*>(out) + (size_t)r0 * p->n_bins * 8;
      kp.copies_log2 = cl2;
      kp.direct_store = direct ? 1 : 0;
      kp.segs = (int32_t)segs;
      const dim3 grid((unsigned)(nr * segs));
      if (first_launch)
        if (int rrc = rec.begin(profile)) return rrc;
      for (int sl = 0; sl < n_slices; ++sl) {  // (one launch unless the histogram is built in bin slices)
        kp.slice_lo = (int64_t)sl * slice_bins;
        kp.slice_n = (int32_t)std::min<int64_t>(slice_bins, p->n_bins - kp.slice_lo);
        XH_LAUNCH_PICKED(fn, grid, dim3(block), lds_bytes, stream, kp);
        HIPC(hipGetLastError());
      }
      if (first_launch) {
        snprintf(desc, sizeof desc,
                 "family=%s hist=%s vec=%d unroll=%d block=%d grid=%lld segs=%lld lds_bytes=%zu copies=%d table_bytes=%zu "
                 "lut_k0=%d steps0=%d scan=%d weighted=%d D=%d cmp=%s lds_cap=%zu%s slices=%d direct_store=%d",
                 mixed ? "fast mixed-dtypes" : (fast ? "fast" : "generic"), hist == kHistLds ? "lds" : (hist == kHistPacked ? "packed16" : "global"),
                 fast ? vec : 1, fast ? kUnroll : 1, block, (long long)(nr * segs), (long long)segs, lds_bytes, 1 << cl2,
                 table_bytes, dims[0].lut_k, dims[0].steps, scan, (int)weighted, D,
                 use_f32 ? "f32thr" : (p->cmp == XHIST_CMP_I64 ? "i64" : (p->cmp == XHIST_CMP_F64 ? "f64" : "per-input")), lds_cap,
                 two ? " weights=2" : "", n_slices, (int)direct);
      }
      first_launch = false;
      r0 += nr;
    }
  }
  return rec.end(desc);
}
