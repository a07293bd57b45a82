// xhist_route_f32.hip — instantiates part_route for float samples (see xhist_pick.hip.h, xhist_route.hip.h)
#include "xhist_pick.hip.h"

kernel_fn_route xhist_pick_route_f32(int wdt, int D, int scan, bool multi) { return route_pick<float>(wdt, D, scan, multi); }
