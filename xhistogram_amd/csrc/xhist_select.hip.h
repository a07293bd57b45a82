// xhist_select.hip.h — host side: table-set choice, launch bookkeeping, kernel pickers (template instantiation points)
// Part of the single translation unit xhist_capi.hip (included there, in order).
#pragma once

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
    if (p->profile_stride > 1 && (p->n_seen++ % p->profile_stride) != 0) return XHIST_OK;  // this execute is not sampled
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

// Integer and half-precision samples (category ids, sensor counts, packed fields): the same vector
// kernel with an in-register conversion to double — numpy compares them in float64 against
// float64 edges too.  Kept to the shapes that matter so the instantiation count stays small:
// one input, unweighted or float64 weights, LDS or global histogram, uniform-style tables
// (SCAN 1) or binary search (SCAN 0); everything else takes the generic family.
template <typename ST>
static kernel_fn small_pick(int wdt, int D, int scan, int hist) {
  if (D != 1 || (scan != 0 && scan != 1)) return nullptr;
  if (hist == kHistLanes || hist == kHistLanes16) {  // row-per-lane kernels: leading-axis reductions of such arrays
    if (wdt == -1) {  // (counts: uint16 counter columns only, as for float samples)
      if (hist == kHistLanes16) return scan ? (kernel_fn)hist_lanes<ST, NoWeight, 1, 1, 8, true> : (kernel_fn)hist_lanes<ST, NoWeight, 1, 0, 8, true>;
      return nullptr;
    }
    if (wdt == XHIST_F64 && hist == kHistLanes) return scan ? (kernel_fn)hist_lanes<ST, double, 1, 1, 8, false> : (kernel_fn)hist_lanes<ST, double, 1, 0, 8, false>;
    return nullptr;
  }
  if (hist != kHistLds && hist != kHistGlobal) return nullptr;
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

// int64 / datetime64 samples against integer edges, compared exactly in int64: one input, unweighted or
// float64 weights, LDS or global histogram, (start, cnt) tables with the branch-free binary search
static kernel_fn int64_domain_kernel(int wdt, int D, int scan, int hist, int* vec) {
  if (D != 1 || scan != 0 || (hist != kHistLds && hist != kHistGlobal)) return nullptr;
  *vec = 2;
  constexpr int U = unroll_for(1, 2, 0);
  if (wdt == -1)
    return hist == kHistLds ? (kernel_fn)hist_fast<int64_t, NoWeight, 1, 2, U, kHistLds, 0, false, false, true>
                            : (kernel_fn)hist_fast<int64_t, NoWeight, 1, 2, U, kHistGlobal, 0, false, false, true>;
  if (wdt == XHIST_F64)
    return hist == kHistLds ? (kernel_fn)hist_fast<int64_t, double, 1, 2, U, kHistLds, 0, false, false, true>
                            : (kernel_fn)hist_fast<int64_t, double, 1, 2, U, kHistGlobal, 0, false, false, true>;
  return nullptr;
}

static kernel_fn fast_kernel(int sdt, int wdt, int D, int scan, int hist, int* vec) {
  const int ssz = dtype_size(sdt), wsz = wdt < 0 ? 0 : dtype_size(wdt);
  *vec = 16 / std::max(ssz, wsz);
  switch (sdt) {
    // (the kernels of a typical first call come from the small hot unit, xhist_hot.hip; the rest are instantiated in
    //  xhist_pick_f64.hip / _f32.hip, whose code objects are loaded when one of them is first needed)
    case XHIST_F64: if (kernel_fn h = xhist_pick_hot(sdt, wdt, D, scan, hist)) return h; return xhist_pick_f64(wdt, D, scan, hist);
    case XHIST_F32: if (kernel_fn h = xhist_pick_hot(sdt, wdt, D, scan, hist)) return h; return xhist_pick_f32(wdt, D, scan, hist);
    case XHIST_I32: return small_pick<int32_t>(wdt, D, scan, hist);
    case XHIST_I64: return small_pick<int64_t>(wdt, D, scan, hist);
    case XHIST_I16: return small_pick<int16_t>(wdt, D, scan, hist);
    case XHIST_U8: return small_pick<uint8_t>(wdt, D, scan, hist);
    case XHIST_F16: return small_pick<_Float16>(wdt, D, scan, hist);
    default: return nullptr;
  }
}

// bin slices (hist_fast<..., SLICED = true>): float samples, LDS or packed-uint16 histograms, table
// digitize with <= 2 edges per bucket or arithmetic edges
static kernel_fn fast_kernel_sliced(int sdt, int wdt, int D, int scan, int hist, int* vec) {
  const int ssz = dtype_size(sdt), wsz = wdt < 0 ? 0 : dtype_size(wdt);
  *vec = 16 / std::max(ssz, wsz);
  if (sdt == XHIST_F64) return xhist_pick_sliced_f64(wdt, D, scan, hist);
  if (sdt == XHIST_F32) return xhist_pick_sliced_f32(wdt, D, scan, hist);
  return nullptr;
}

// two weight arrays in one pass (hist_fast<..., W2 = true>): float samples, one weight dtype for
// both arrays, LDS histograms, bucket tables with 1 or 2 edges per bucket
template <typename ST, typename WT, int D>
static kernel_fn two_weights_pick(int scan) {
  constexpr int wsz = (int)sizeof(WT);
  constexpr int VEC = 16 / (((int)sizeof(ST) > wsz) ? (int)sizeof(ST) : wsz);
  if (scan == 1) return (kernel_fn)hist_fast<ST, WT, D, VEC, unroll_for(D, VEC, 1), kHistLds, 1, true>;
  if (scan == 2) return (kernel_fn)hist_fast<ST, WT, D, VEC, unroll_for(D, VEC, 2), kHistLds, 2, true>;
  return nullptr;
}

template <typename ST, typename WT>
static kernel_fn two_weights_pick_d(int D, int scan) {
  switch (D) {
    case 1: return two_weights_pick<ST, WT, 1>(scan);
    case 2: return two_weights_pick<ST, WT, 2>(scan);
    case 3: return two_weights_pick<ST, WT, 3>(scan);
    default: return nullptr;
  }
}

static kernel_fn fast_kernel_two_weights(int sdt, int wdt, int D, int scan, int* vec) {
  const int ssz = dtype_size(sdt), wsz = dtype_size(wdt);
  *vec = 16 / std::max(ssz, wsz);
  if (sdt == XHIST_F64 && wdt == XHIST_F64) return two_weights_pick_d<double, double>(D, scan);
  if (sdt == XHIST_F64 && wdt == XHIST_F32) return two_weights_pick_d<double, float>(D, scan);
  if (sdt == XHIST_F32 && wdt == XHIST_F64) return two_weights_pick_d<float, double>(D, scan);
  if (sdt == XHIST_F32 && wdt == XHIST_F32) return two_weights_pick_d<float, float>(D, scan);
  return nullptr;
}

typedef void (*kernel_fn_rows1)(const Params, int32_t);

template <typename ST>
static kernel_fn_rows1 rows1_pick(int scan) {
  switch (scan) {
    case 1: return (kernel_fn_rows1)hist_lanes_rows1<ST, 1>;
    case 2: return (kernel_fn_rows1)hist_lanes_rows1<ST, 2>;
    case 3: return (kernel_fn_rows1)hist_lanes_rows1<ST, 3>;
    case 4: return (kernel_fn_rows1)hist_lanes_rows1<ST, 4>;
    case kScanPackG: return (kernel_fn_rows1)hist_lanes_rows1<ST, kScanPackG>;
    default: return (kernel_fn_rows1)hist_lanes_rows1<ST, 0>;
  }
}

static kernel_fn_rows1 rows1_kernel(int sdt, int scan) {
  if (sdt == XHIST_F64) return rows1_pick<double>(scan);
  if (sdt == XHIST_F32) return rows1_pick<float>(scan);
  return nullptr;
}

template <bool TLDS>
static kernel_fn generic_kernel_t(int cmp, bool weighted, bool lds) {
  if (cmp == XHIST_CMP_F64) {
    if (weighted) return lds ? (kernel_fn)hist_generic<0, true, true, TLDS> : (kernel_fn)hist_generic<0, true, false, TLDS>;
    return lds ? (kernel_fn)hist_generic<0, false, true, TLDS> : (kernel_fn)hist_generic<0, false, false, TLDS>;
  }
  if (cmp == XHIST_CMP_I64) {
    if (weighted) return lds ? (kernel_fn)hist_generic<1, true, true, TLDS> : (kernel_fn)hist_generic<1, true, false, TLDS>;
    return lds ? (kernel_fn)hist_generic<1, false, true, TLDS> : (kernel_fn)hist_generic<1, false, false, TLDS>;
  }
  // per-input domains (XHIST_CMP_PER_DIM | mask)
  if (weighted) return lds ? (kernel_fn)hist_generic<3, true, true, TLDS> : (kernel_fn)hist_generic<3, true, false, TLDS>;
  return lds ? (kernel_fn)hist_generic<3, false, true, TLDS> : (kernel_fn)hist_generic<3, false, false, TLDS>;
}

// tables outside LDS: the histogram is never in LDS then (place() in execute_device puts it there only next to its tables), so
// those three-times-two kernels are not instantiated (census of round 6)
static kernel_fn generic_kernel_no_lds(int cmp, bool weighted) {
  if (cmp == XHIST_CMP_F64) return weighted ? (kernel_fn)hist_generic<0, true, false, false> : (kernel_fn)hist_generic<0, false, false, false>;
  if (cmp == XHIST_CMP_I64) return weighted ? (kernel_fn)hist_generic<1, true, false, false> : (kernel_fn)hist_generic<1, false, false, false>;
  return weighted ? (kernel_fn)hist_generic<3, true, false, false> : (kernel_fn)hist_generic<3, false, false, false>;
}

static kernel_fn generic_kernel(int cmp, bool weighted, bool lds, bool tables_in_lds) {
  if (tables_in_lds) return generic_kernel_t<true>(cmp, weighted, lds);
  return lds ? nullptr : generic_kernel_no_lds(cmp, weighted);
}
