#!/bin/bash
# development A/B: record quads per lane in flight in the adding-up pass of the partitioned mode (libraries built with -DXHIST_ACC_GROUPS=g)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; export TMPDIR=/tmp; cd /tmp
for lib in "" build_ab/libxhist_acc1.so build_ab/libxhist_acc4.so; do
  rm -rf /tmp/prof_a
  [ -n "$lib" ] && export XHIST_AMD_LIB="$R/$lib" || unset XHIST_AMD_LIB
  rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o p -- python "$R/tools/c5_ab.py" --steps 10 --variants default > /dev/null 2>&1
  echo "== ${lib:-shipped (2 groups)}"
  python "$R/tools/rocpd_summary.py" "$(find /tmp/prof_a -name '*.db' | head -1)" | grep -E "part_accumulate_chunks<true, double, true>|part_route<double, xhist::Packed48" | grep -v "| 1024 |" | cut -c1-70,112-180
done
