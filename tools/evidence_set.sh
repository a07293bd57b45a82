#!/bin/bash
# One evidence set on the GPU box: tools/profile_configs.sh (bench line, rocprofv3 stats, FETCH / WRITE passes, cliff gate per
# config), then profiles/traffic.json from those passes, then the bench lines ONCE MORE so that every *_bench.json cites the
# traffic figures and the code state of this very set (VERDICT r3 "weak" #8: labels that disagree).
#   usage (from the build container):  gpurun -- "XHIST_CODE_STATE=$(git rev-parse --short HEAD) bash tools/evidence_set.sh r04_z"
#   then locally:                       python tools/pmc_traffic.py gpurun_out/r04_z <commit> --copy-to profiles/r04_z
set -u
tag="$1"; shift
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
state="${XHIST_CODE_STATE:-snapshot}"
export XHIST_CODE_STATE="$state"
out="$R/gpurun_out/$tag"; mkdir -p "$out"
# what a kernel that only READS reaches on this box, in this call (VERDICT r4 "next" #4): one / two 8 GB streams, 16-byte nt loads
[ -x "$R/tools/ubench/readbw" ] && (cd /tmp && timeout 300 "$R/tools/ubench/readbw" quick > "$out/bare_read_ceiling.txt" 2>&1)
bash "$R/tools/profile_configs.sh" "$tag" "$@"
[ -x "$R/tools/ubench/readbw" ] && (cd /tmp && timeout 300 "$R/tools/ubench/readbw" quick >> "$out/bare_read_ceiling.txt" 2>&1)
python "$R/tools/pmc_traffic.py" "$out" "$state" > "$out/traffic_summary.txt" 2>&1
cp "$R/profiles/traffic.json" "$out/traffic.json"
cd /tmp
for name in c1 c2 c2u c3 c4 c4_full c5 c5_full; do
  [ -f "$out/${name}_bench.json" ] || continue
  c=${name%%_*}; extra=""
  [ "$name" = c2u ] && { c=c2; extra="--unweighted"; }
  [ "$name" = c4_full ] && extra="--full"
  [ "$name" = c5_full ] && extra="--full --steps 5 --warmup 1"
  timeout 600 python "$R/bench.py" --config $c $extra > "$out/${name}_bench.json" 2> "$out/${name}_bench.err"
  python -c "import json; d=json.load(open('$out/${name}_bench.json')); r=d['roofline']; print('%-8s kernel %.4f ms  frac %.4f  traffic %s  (%s)' % ('$name', r['kernel_ms_mean'], r['frac'], r['traffic'], r.get('traffic_source')))"
done | tee "$out/final_lines.txt"
cat "$out/traffic_summary.txt"
