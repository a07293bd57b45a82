// xhist_pick_flat.hip — instantiates hist_flat_rows (dense short rows streamed flat, xhist_lanes.hip.h): float32 / float64
// samples, one or two inputs, counts or float32 / float64 weights
#include "xhist_pick.hip.h"

template <typename ST, typename WT, int D>
static kernel_fn_flat flat_pick_scan(int scan) {
  switch (scan) {
    case 0: return (kernel_fn_flat)hist_flat_rows<ST, WT, D, 0>;
    case 1: return (kernel_fn_flat)hist_flat_rows<ST, WT, D, 1>;
    case 2: return (kernel_fn_flat)hist_flat_rows<ST, WT, D, 2>;
    case 3: if constexpr (D == 1) return (kernel_fn_flat)hist_flat_rows<ST, WT, D, 3>; else return nullptr;
    case 4: if constexpr (D == 1) return (kernel_fn_flat)hist_flat_rows<ST, WT, D, 4>; else return nullptr;
    case kScanPackG: return (kernel_fn_flat)hist_flat_rows<ST, WT, D, kScanPackG>;  // packed bucket entries, map per dimension
    default: return nullptr;
  }
}

template <typename ST, typename WT>
static kernel_fn_flat flat_pick_d(int D, int scan) {
  return D == 1 ? flat_pick_scan<ST, WT, 1>(scan) : (D == 2 ? flat_pick_scan<ST, WT, 2>(scan) : nullptr);
}

template <typename ST>
static kernel_fn_flat flat_pick_w(int wdt, int D, int scan) {
  if (wdt == -1) return flat_pick_d<ST, NoWeight>(D, scan);
  if (wdt == XHIST_F32) return flat_pick_d<ST, float>(D, scan);
  if (wdt == XHIST_F64) return flat_pick_d<ST, double>(D, scan);
  return nullptr;
}

kernel_fn_flat xhist_pick_flat_rows(int sdt, int wdt, int D, int scan) {
  if (sdt == XHIST_F64) return flat_pick_w<double>(wdt, D, scan);
  if (sdt == XHIST_F32) return flat_pick_w<float>(wdt, D, scan);
  return nullptr;
}
