#!/bin/bash
# SQ counters of ONE bench.py config's main kernel, two rocprofv3 --pmc passes of 8 counters (never with other trace domains).
#   usage: bash tools/sq_counters.sh <out.txt> <bench.py arguments...>      e.g.  bash tools/sq_counters.sh gpurun_out/c3_sq.txt --config c3
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$(realpath -m "$1")"; shift
export TMPDIR=/tmp
cd /tmp
: > "$out"
for pass in "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" \
            "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/prof_q
  timeout 600 rocprofv3 --pmc $pass --kernel-trace -d /tmp/prof_q -o p -- python "$R/bench.py" "$@" --profiler-pass --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
  python "$R/tools/rocpd_summary.py" "$(find /tmp/prof_q -name '*.db' | head -1)" 2>&1 | grep -v "zero_words\|^#" >> "$out"
done
