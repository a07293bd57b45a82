export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o c5 -- python $R/bench.py --config c5 --no-cpu-baseline "$@" > /tmp/prof.log 2>&1
cd $R; python tools/rocpd_summary.py $(find /tmp/prof -name "*.db" | head -1) | grep -E "xhist::part|xhist dispatch" | cut -c1-175
