// xhist_pick_f64.hip — instantiates the double-sample kernels of the vector family (see xhist_pick.hip.h)
#include "xhist_pick.hip.h"

kernel_fn xhist_pick_f64(int wdt, int D, int scan, int hist) { return fast_pick_w<double>(wdt, D, scan, hist); }

kernel_fn xhist_pick_sliced_f64(int wdt, int D, int scan, int hist) {
  if (wdt == -1) return sliced_pick_ds<double, NoWeight>(D, scan, hist);
  if (wdt == XHIST_F64) return sliced_pick_ds<double, double>(D, scan, hist);
  if (wdt == XHIST_F32) return sliced_pick_ds<double, float>(D, scan, hist);
  return nullptr;
}

// one float64 input, unweighted, LDS histogram: 16 samples per lane and tile instead of 8 — 128 bytes per lane in flight, what
// the weighted kernel has with its two streams (xhist_exec_device.hip.h picks it for one long row)
kernel_fn xhist_pick_f64_long(int scan) {
  if (scan == 1) return xhist_pick_hot_long(XHIST_F64, 1);  // (the headline's 8 B/sample variant: in the hot unit)
  if (scan == 2) return (kernel_fn)hist_fast<double, NoWeight, 1, 2, 8, kHistLds, 2>;
  return nullptr;
}
