// xhist_route_f64_b1024.hip — instantiates part_route for double samples, 1024-thread workgroups (see xhist_pick.hip.h, xhist_route.hip.h)
#include "xhist_pick.hip.h"

kernel_fn_route xhist_pick_route_f64_b1024(int wdt, int D, int scan, bool multi) { return route_pick<double, 1024>(wdt, D, scan, multi); }
