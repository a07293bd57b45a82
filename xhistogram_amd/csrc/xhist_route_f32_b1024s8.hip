// xhist_route_f32_b1024s8.hip — instantiates part_route for float samples, 1024-thread workgroups, 8 samples per lane and tile
#include "xhist_pick.hip.h"

kernel_fn_route xhist_pick_route_f32_b1024s8(int wdt, int D, int scan, bool multi) { return route_pick<float, 1024, 8>(wdt, D, scan, multi); }
