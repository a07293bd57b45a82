// xhist_pick_f64.hip — instantiates the double-sample kernels of the vector family (see xhist_pick.hip.h)
#include "xhist_pick.hip.h"

kernel_fn xhist_pick_f64(int wdt, int D, int scan, int hist) { return fast_pick_w<double>(wdt, D, scan, hist); }

kernel_fn xhist_pick_sliced_f64(int wdt, int D, int scan, int hist) {
  if (wdt == -1) return sliced_pick_ds<double, NoWeight>(D, scan, hist);
  if (wdt == XHIST_F64) return sliced_pick_ds<double, double>(D, scan, hist);
  if (wdt == XHIST_F32) return sliced_pick_ds<double, float>(D, scan, hist);
  return nullptr;
}
