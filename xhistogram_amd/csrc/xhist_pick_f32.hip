// xhist_pick_f32.hip — instantiates the float-sample kernels of the vector family (see xhist_pick.hip.h)
#include "xhist_pick.hip.h"

kernel_fn xhist_pick_f32(int wdt, int D, int scan, int hist) { return fast_pick_w<float>(wdt, D, scan, hist); }

kernel_fn xhist_pick_sliced_f32(int wdt, int D, int scan, int hist) {
  if (wdt == -1) return sliced_pick_ds<float, NoWeight>(D, scan, hist);
  if (wdt == XHIST_F64) return sliced_pick_ds<float, double>(D, scan, hist);
  if (wdt == XHIST_F32) return sliced_pick_ds<float, float>(D, scan, hist);
  return nullptr;
}
