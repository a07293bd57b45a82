// xhist_exchange.hip — instantiates the exchange mode of the partitioned path (see xhist_exchange.hip.h, xhist_pick.hip.h)
#include "xhist_pick.hip.h"

kernel_fn_exch xhist_pick_exchange(int D, bool exact) {
  if (exact) return D == 1 ? (kernel_fn_exch)part_exchange<1, true> : D == 2 ? (kernel_fn_exch)part_exchange<2, true> : D == 3 ? (kernel_fn_exch)part_exchange<3, true> : nullptr;
  return D == 1 ? (kernel_fn_exch)part_exchange<1> : D == 2 ? (kernel_fn_exch)part_exchange<2> : D == 3 ? (kernel_fn_exch)part_exchange<3> : nullptr;
}
kernel_fn_exch xhist_pick_exchange_probe(int D) {
  return D == 1 ? (kernel_fn_exch)exchange_probe<1> : D == 2 ? (kernel_fn_exch)exchange_probe<2> : D == 3 ? (kernel_fn_exch)exchange_probe<3> : nullptr;
}
kernel_fn_exch_pick xhist_pick_exchange_pick() { return (kernel_fn_exch_pick)exchange_pick<0>; }
kernel_fn_exch_merge xhist_pick_exchange_merge() { return (kernel_fn_exch_merge)exchange_merge<0>; }
