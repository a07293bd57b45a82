#!/bin/bash
# development helper: FETCH_SIZE / WRITE_SIZE passes over one xchg variant; arguments: output dir, variant, dist, mode
out=$1; v=$2; dist=$3; mode=$4
mkdir -p $out
R=$(cd $(dirname $0)/../.. && pwd)
cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_x; timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/prof_x -o p -- $R/tools/ubench/xchg_$v $dist $mode 3 > $out/pmc_${v}_${dist}_${mode}_$ctr.log 2>&1
  python $R/tools/rocpd_summary.py "$(find /tmp/prof_x -name '*.db' | head -1)" > $out/pmc_${v}_${dist}_${mode}_$ctr.txt 2>&1
  grep -A12 "PMC counters" $out/pmc_${v}_${dist}_${mode}_$ctr.txt
done
