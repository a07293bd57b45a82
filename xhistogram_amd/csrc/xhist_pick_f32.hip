// xhist_pick_f32.hip — instantiates the float-sample kernels of the vector family (see xhist_pick.hip.h)
#include "xhist_pick.hip.h"

kernel_fn xhist_pick_f32(int wdt, int D, int scan, int hist) { return fast_pick_w<float>(wdt, D, scan, hist); }

kernel_fn xhist_pick_sliced_f32(int wdt, int D, int scan, int hist) {
  if (wdt == -1) return sliced_pick_ds<float, NoWeight>(D, scan, hist);
  if (wdt == XHIST_F64) return sliced_pick_ds<float, double>(D, scan, hist);
  if (wdt == XHIST_F32) return sliced_pick_ds<float, float>(D, scan, hist);
  return nullptr;
}

// one float32 input, unweighted, LDS histogram: 32 samples per lane and tile instead of 16 (128 bytes per lane in flight) for
// long rows — BASELINE C4's shape (xhist_exec_device.hip.h)
kernel_fn xhist_pick_f32_long(int scan) {
  if (scan == 1 || scan == kScanArith32) return xhist_pick_hot_long(XHIST_F32, scan);  // (BASELINE C4's kernels: in the hot unit)
  if (scan == 2) return (kernel_fn)hist_fast<float, NoWeight, 1, 4, 8, kHistLds, 2>;
  return nullptr;
}
