#!/bin/bash
# A/B of builds / knobs on one box: usage tools/experiments/r05_ab.sh <outdir> "<ENV...>" "<ENV...>" ...
# every variant: the driver-shaped bench, no CPU baseline; one line per variant: ms_per_step, stage times.  BENCH_ARGS adds arguments.
set -u
OUT=$1; shift
mkdir -p $OUT
n=0
for v in "$@"; do
  n=$((n+1))
  env $v timeout -k 10 240 python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/v$n.log 2> $OUT/v$n.err || { echo "variant $n failed"; tail -5 $OUT/v$n.err; exit 1; }
  python3 - "$OUT/v$n.log" "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s=d.get('stage_ms_per_iter',{})
print('%-52s step %.4f graph %.4f | nn %.4f grid %.4f attract %.4f prior %.4f as %.4f update %.4f total %.4f' % (sys.argv[2], d['ms_per_step'], d.get('ms_per_step_graph_replay') or 0, s.get('nn',0), s.get('grid',0), s.get('attract',0), s.get('prior',0), s.get('as',0), s.get('update',0), s.get('total',0)))
PY
done
