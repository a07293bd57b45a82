#!/bin/bash
# Evidence run on the GPU box (gpurun): for each BASELINE config, (1) the plain bench line, (2) rocprofv3
# --kernel-trace --stats of the same command, (3) / (4) --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own
# (counters never together with tracing of other domains).  Everything lands in gpurun_out/<tag>/, reduced to text
# by tools/rocpd_summary.py; tools/pmc_traffic.py turns the summaries into profiles/traffic.json.
#   usage: bash tools/profile_configs.sh <tag> [configs...]      e.g.  bash tools/profile_configs.sh r02_f c2 c3 c4 c5
set -u
tag="$1"; shift
configs=("$@"); [ ${#configs[@]} -eq 0 ] && configs=(c1 c2 c2u c3 c4 c4full c5 c5full)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$R/gpurun_out/$tag"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for c in "${configs[@]}"; do
  extra=""; name=$c
  [ "$c" = c4full ] && { c=c4; extra="--full"; name=c4_full; }
  [ "$c" = c5full ] && { c=c5; extra="--full --steps 5 --warmup 1"; name=c5_full; }
  [ "$c" = c2u ] && { c=c2; extra="--unweighted"; name=c2u; }
  prof_extra="$extra --profiler-pass"   # (the driver's line carries a second kernel for c2 and a cold burst for c4; a counter pass wants one kernel, steps + warmup launches)
  timeout 600 python "$R/bench.py" --config $c $extra > "$out/${name}_bench.json" 2> "$out/${name}_bench.err"
  rm -rf /tmp/prof_s; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o p -- python "$R/bench.py" --config $c $prof_extra --no-cpu-baseline > "$out/${name}_bench_under_rocprof.json" 2> /dev/null
  python "$R/tools/rocpd_summary.py" "$(find /tmp/prof_s -name '*.db' | head -1)" > "$out/${name}_kernel_stats.txt" 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_c; timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/prof_c -o p -- python "$R/bench.py" --config $c $prof_extra --no-cpu-baseline --steps 5 --warmup 1 > "$out/${name}_bench_under_pmc_${ctr}.json" 2> /dev/null
    python "$R/tools/rocpd_summary.py" "$(find /tmp/prof_c -name '*.db' | head -1)" > "$out/${name}_pmc_${ctr}.txt" 2>&1
  done
  echo "$name done: $(python -c "import json; d=json.load(open('$out/${name}_bench.json')); print('%.4g samples/s, kernel %.3f ms, frac %.3f' % (d['value'], d['roofline']['kernel_ms_mean'], d['roofline']['frac']))" 2>&1)"
done
# the code state every file of this set was measured on: the commit when the box has .git (it does not under gpurun: the
# caller passes XHIST_CODE_STATE=$(git rev-parse --short HEAD)), so that *_bench.json, traffic.json and this file agree
echo "${XHIST_CODE_STATE:-$(cd "$R" && git rev-parse --short HEAD 2>/dev/null || echo snapshot)}" > "$out/code_state.txt"
# regression gate over the shape scanners (tools/check_cliffs.py): every evidence set carries its own cells
if [ "${XHIST_SKIP_CLIFFS:-0}" != 1 ]; then
  python "$R/tools/check_cliffs.py" run "$out/cliffs" > "$out/cliffs_run.log" 2>&1 \
    && python "$R/tools/check_cliffs.py" compare "$R/profiles/cliffs_baseline" "$out/cliffs" > "$out/cliffs_gate.txt" 2>&1
  tail -3 "$out/cliffs_gate.txt" 2>/dev/null
fi
