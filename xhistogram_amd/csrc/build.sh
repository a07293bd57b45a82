#!/bin/bash
# Build libxhist_amd.so in-tree for gfx950 (cross-compiles without a GPU).  Eleven translation units (xhist_hot: the small code object a first call loads),
# compiled in parallel: nine that instantiate the float64 / float32 / mixed-dtype vector, routing and exchange kernels, and the rest.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${XHIST_BUILD_OUT:-$here/../libxhist_amd.so}"  # (development: XHIST_BUILD_OUT / XHIST_BUILD_FLAGS build an A/B variant next to the library)
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
obj="$(mktemp -d)"
trap 'rm -rf "$obj"' EXIT
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${XHIST_BUILD_FLAGS:-})
pids=()
tus=(xhist_hot xhist_capi xhist_pick_f64 xhist_pick_f32 xhist_pick_mixed xhist_pick_flat xhist_route_f64_b1024 xhist_route_f64_b1024s8 xhist_route_f32_b1024 xhist_route_f32_b1024s8 xhist_exchange)
for tu in "${tus[@]}"; do
  "$HIPCC" "${flags[@]}" -c -o "$obj/$tu.o" "$here/$tu.hip" &
  pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid"; done
objs=()
for tu in "${tus[@]}"; do objs+=("$obj/$tu.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out.tmp" "${objs[@]}"
mv -f "$out.tmp" "$out"
echo "built $out"
