#!/bin/bash
# Build libxhist_amd.so in-tree for gfx950 (cross-compiles without a GPU).  Three translation units,
# compiled in parallel: the two that instantiate the float64 / float32 vector kernels, and the rest.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libxhist_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
obj="$(mktemp -d)"
trap 'rm -rf "$obj"' EXIT
flags=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function)
pids=()
for tu in xhist_capi xhist_pick_f64 xhist_pick_f32; do
  "$HIPCC" "${flags[@]}" -c -o "$obj/$tu.o" "$here/$tu.hip" &
  pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out.tmp" "$obj"/xhist_capi.o "$obj"/xhist_pick_f64.o "$obj"/xhist_pick_f32.o
mv -f "$out.tmp" "$out"
echo "built $out"
