#!/bin/bash
# Routing-pass geometry matrix (partitioned mode): for every dtype combination, 5*10^8 samples through the automatic
# choice, 1024 threads x 4 samples, 1024 x 8 and 2 x 512 x 4 (tools/c5_ab.py; HIP-event time of the whole call — zeroing,
# routing pass, adding-up pass — median of 10 after 3).  Source of the table in route_geom_for (xhist_exec_device.hip.h).
#   usage (GPU box): bash tools/route_geometry_matrix.sh > gpurun_out/<tag>/route_geometry.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
V="default;route_block=1024,route_spl=4;route_block=1024,route_spl=8;route_block=512;default"
run() {
  echo "== $*"
  timeout 300 python "$R/tools/c5_ab.py" --variants "$V" "$@" | python -c "
import sys, json, re
for l in sys.stdin:
    r = json.loads(l); d = r['desc']
    print('  %-34s %7.3f ms (min %7.3f)  %s %s %s %s' % (r['variant'], r['ms_median'], r['ms_min'], 'ok' if r['matches_first'] else 'MISMATCH',
          ' '.join(re.findall(r'tile=\d+ block=\d+', d)), ' '.join(re.findall(r'scan=\d', d)), ' '.join(re.findall(r'records=\S+', d))))"
}
for st in f64 f32; do
  run --dtype $st --dims 2 --bins 1024 --unweighted
  run --dtype $st --dims 1 --bins 1000000 --unweighted
  run --dtype $st --dims 3 --bins 100 --n 300000000 --unweighted
  run --dtype $st --dims 2 --bins 1024 --unweighted --edges jitter
  for wt in f32 f64; do
    run --dtype $st --wdtype $wt --dims 2 --bins 1024
    run --dtype $st --wdtype $wt --dims 1 --bins 1000000
    run --dtype $st --wdtype $wt --dims 3 --bins 100 --n 300000000
    run --dtype $st --wdtype $wt --dims 2 --bins 1024 --edges jitter
  done
  run --dtype $st --wdtype f64 --dims 2 --bins 1024 --signs both
done
run --dtype f64 --dims 2 --bins 512 --rows 8 --n 480000000 --unweighted
run --dtype f64 --wdtype f64 --dims 2 --bins 512 --rows 8 --n 480000000
run --dtype f32 --wdtype f32 --dims 2 --bins 400 --rows 32 --n 960000000
