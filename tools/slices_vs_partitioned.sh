#!/bin/bash
# Histograms of 2-5 x the LDS capacity: bin slices (S passes over the samples) against the partitioned mode, and the
# partitioned mode with the bins cut into a handful of partitions ("min_parts=1": 2^14 / 2^15 bins each, round 2's form)
# against 16+ finer ones.  5*10^8 samples per call, HIP-event median of 10 calls after 3 (tools/c5_ab.py).  Source of the
# cost model in execute_device (xhist_exec_device.hip.h: "bin slices") and of its few-partitions rule.
#   usage (GPU box): bash tools/slices_vs_partitioned.sh > gpurun_out/<tag>/few_partitions.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
V="default;slices=1;partition=1;partition=1,min_parts=1;default"
run() {
  echo "== $*"
  timeout 300 python "$R/tools/c5_ab.py" --variants "$V" "$@" 2>/dev/null | python -c "
import sys, json, re
for l in sys.stdin:
    r = json.loads(l); d = r['desc']
    print('  %-26s %7.3f ms (min %7.3f)  %s %s' % (r['variant'], r['ms_median'], r['ms_min'], 'ok' if r['matches_first'] else 'MISMATCH',
          ' '.join(re.findall(r'hist=\S+|rows_per_pass=\d+|parts=\d+|bins_per_part=\d+|tile=\d+|slices=\d+|records=\S+', d))))"
}
for rows in 1 12; do
  for b in 160 200 256; do
    run --dtype f32 --wdtype f32 --dims 2 --bins $b --rows $rows
    run --dtype f64 --wdtype f64 --dims 2 --bins $b --rows $rows
  done
  for b in 300 400 512; do
    run --dtype f32 --dims 2 --bins $b --unweighted --rows $rows
    run --dtype f64 --dims 2 --bins $b --unweighted --rows $rows
  done
done
run --dtype f64 --dims 1 --bins 100000 --unweighted
run --dtype f32 --dims 1 --bins 100000 --unweighted
run --dtype f64 --wdtype f64 --dims 1 --bins 40000
run --dtype f32 --wdtype f32 --dims 3 --bins 40 --n 300000000
