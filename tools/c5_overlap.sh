#!/bin/bash
# C5 shard: serial passes against sub-batches on two streams (development tool; run on the GPU box).
#   (1) whole-call times of the variants in ONE process   (2) per-kernel times of the two passes alone at reduced grids
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$(realpath -m "${1:-$R/gpurun_out/c5_overlap.txt}")"
export TMPDIR=/tmp
cd /tmp
{
echo "# whole call (HIP events around it), one process"
python "$R/tools/c5_ab.py" --steps 12 --variants "default;overlap=4,overlap_cus=32;overlap=4,overlap_cus=48;overlap=4,overlap_cus=64;overlap=8,overlap_cus=32;overlap=8,overlap_cus=48;overlap=8,overlap_cus=64;overlap=16,overlap_cus=48;overlap=2,overlap_cus=48;default" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print('%-28s median %.3f  min %.3f  max %.3f  frac %.3f  ok=%s' % (d['variant'], d['ms_median'], d['ms_min'], d['ms_max'], d['frac_of_8TBs'], d['matches_first']))
"
echo "# the passes alone at reduced grids (rocprofv3 --kernel-trace --stats, one run per variant): name | calls | total | avg | min | max"
for v in default route_grid=240 route_grid=224 route_grid=208 route_grid=192 acc_grid=128 acc_grid=64 acc_grid=48 acc_grid=32; do
  rm -rf /tmp/prof_o
  rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o p -- python "$R/tools/c5_ab.py" --steps 8 --variants "$v" > /dev/null 2>&1
  echo "## $v"
  python "$R/tools/rocpd_summary.py" "$(find /tmp/prof_o -name '*.db' | head -1)" | grep -E "xhist::part" | grep -v "^void.*|.*|.*|.*|.*|.*|.*|.*|" | cut -c1-60,110-190
done
} > "$out" 2>&1
cat "$out"
