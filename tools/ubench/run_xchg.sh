#!/bin/bash
# development helper: run the xchg variants found next to this script; arguments: output dir, then "variant:dist:mode:reps" items
out=$1; shift
mkdir -p $out
cd $(dirname $0)
for item in "$@"; do
  IFS=: read v dist mode reps <<< "$item"
  echo "== $v dist=$dist mode=$mode" >> $out/runs.txt
  timeout 120 ./xchg_$v $dist $mode $reps 2>&1 | cat >> $out/runs.txt
done
cat $out/runs.txt
