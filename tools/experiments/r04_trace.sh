#!/bin/bash
# kernel trace of the default bench (+ the list statistics of a second run): usage tools/experiments/r04_trace.sh <outdir> [ENV...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
env "$@" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph-pass ${BENCH_ARGS:-} > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%-60s calls %5s avg %9.1f us  min %9.1f  max %9.1f  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
env "$@" NW_LANE_STATS=1 NW_VERBOSE=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/stats.log 2> $OUT/stats.err || { echo "stats run failed"; exit 1; }
grep "list queries" $OUT/stats.err | sed -n '1p;3p;5p;7p;9p'
