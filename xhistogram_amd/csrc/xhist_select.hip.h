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
typedef void (*kernel_fn_acc)(const uint16_t*, const void*, const uint64_t*, void*, int64_t, int, int);
typedef void (*kernel_fn_count)(const Params, uint32_t*);
typedef void (*kernel_fn_lanes)(const Params, int32_t, int64_t);
typedef void (*kernel_fn_scatter)(const uint32_t*, const void*, int64_t, const uint64_t*, uint16_t*, void*, int, int);

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

// bin slices (hist_fast<..., SLICED = true>): float samples, LDS or packed-uint16 histograms, table
// digitize with <= 2 edges per bucket or arithmetic edges
template <typename ST, typename WT, int D, int SCAN>
static kernel_fn sliced_pick(int hist) {
  constexpr bool unweighted = std::is_same<WT, NoWeight>::value;
  constexpr int wsz = unweighted ? 0 : (int)sizeof(typename std::conditional<unweighted, float, WT>::type);
  constexpr int VEC = 16 / (((int)sizeof(ST) > wsz) ? (int)sizeof(ST) : wsz);
  constexpr int U = unroll_for(D, VEC, SCAN);
  if (hist == kHistLds) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistLds, SCAN, false, true>;
  if (hist == kHistPacked) {
    if constexpr (unweighted) return (kernel_fn)hist_fast<ST, WT, D, VEC, U, kHistPacked, SCAN, false, true>;
  }
  return nullptr;
}

template <typename ST, typename WT>
static kernel_fn sliced_pick_ds(int D, int scan, int hist) {
#define XH_SLICED_CASE(DD)                                            \
  case DD:                                                            \
    if (scan == 1) return sliced_pick<ST, WT, DD, 1>(hist);           \
    if (scan == 2) return sliced_pick<ST, WT, DD, 2>(hist);           \
    if (scan == kScanArith) return sliced_pick<ST, WT, DD, kScanArith>(hist); \
    return nullptr;
  switch (D) {
    XH_SLICED_CASE(1)
    XH_SLICED_CASE(2)
    XH_SLICED_CASE(3)
    default: return nullptr;
  }
#undef XH_SLICED_CASE
}

static kernel_fn fast_kernel_sliced(int sdt, int wdt, int D, int scan, int hist, int* vec) {
  const int ssz = dtype_size(sdt), wsz = wdt < 0 ? 0 : dtype_size(wdt);
  *vec = 16 / std::max(ssz, wsz);
  if (sdt == XHIST_F64) {
    if (wdt == -1) return sliced_pick_ds<double, NoWeight>(D, scan, hist);
    if (wdt == XHIST_F64) return sliced_pick_ds<double, double>(D, scan, hist);
    if (wdt == XHIST_F32) return sliced_pick_ds<double, float>(D, scan, hist);
  } else if (sdt == XHIST_F32) {
    if (wdt == -1) return sliced_pick_ds<float, NoWeight>(D, scan, hist);
    if (wdt == XHIST_F64) return sliced_pick_ds<float, double>(D, scan, hist);
    if (wdt == XHIST_F32) return sliced_pick_ds<float, float>(D, scan, hist);
  }
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
  if (cmp == XHIST_CMP_I64) {
    if (weighted) return lds ? (kernel_fn)hist_generic<1, true, true> : (kernel_fn)hist_generic<1, true, false>;
    return lds ? (kernel_fn)hist_generic<1, false, true> : (kernel_fn)hist_generic<1, false, false>;
  }
  // per-input domains (XHIST_CMP_PER_DIM | mask)
  if (weighted) return lds ? (kernel_fn)hist_generic<3, true, true> : (kernel_fn)hist_generic<3, true, false>;
  return lds ? (kernel_fn)hist_generic<3, false, true> : (kernel_fn)hist_generic<3, false, false>;
}
