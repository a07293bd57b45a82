#!/bin/bash
# Build libxhist_amd.so in-tree for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libxhist_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function \
  -o "$out.tmp" "$here/xhist_capi.hip"
mv -f "$out.tmp" "$out"
echo "built $out"
