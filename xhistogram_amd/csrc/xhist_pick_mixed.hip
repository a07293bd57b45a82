// xhist_pick_mixed.hip — instantiates the mixed-dtype variant of the vector kernels (see xhist_pick.hip.h)
#include "xhist_pick.hip.h"

kernel_fn xhist_pick_mixed(bool weighted, int D, int scan) {
  return weighted ? mixed_pick_ds<double>(D, scan) : mixed_pick_ds<NoWeight>(D, scan);
}
