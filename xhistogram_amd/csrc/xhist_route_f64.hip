// xhist_route_f64.hip — instantiates part_route for double samples (see xhist_pick.hip.h, xhist_route.hip.h)
#include "xhist_pick.hip.h"

kernel_fn_route xhist_pick_route_f64(int wdt, int D, int scan, bool multi) { return route_pick<double>(wdt, D, scan, multi); }
