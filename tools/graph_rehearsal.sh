#!/bin/bash
# One-GPU rehearsal of recorded blocks with RCCL collectives inside (ONE rank: NW_BENCH_FORCE_DIST=1 takes bench.py's N > 1 path).
# NW_GRAPH_COLLECTIVES: 1 = record every block, 0 = launch by launch, auto = record where the host is the bound (the default).
# usage: tools/graph_rehearsal.sh [extra bench.py arguments, e.g. --scale 0.1]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export NW_BENCH_FORCE_DIST=1
for mode in tiles halo; do
  for g in 1 0 auto; do
    echo "== mode $mode NW_GRAPH_COLLECTIVES=$g $*"
    NW_GRAPH_COLLECTIVES=$g timeout -k 10 300 python bench.py --gpus 1 --steps 40 --warmup 10 --mode $mode --no-cpu-baseline "$@" > gpurun_out/reh_${mode}_$g.json 2> gpurun_out/reh_${mode}_$g.err || { tail -20 gpurun_out/reh_${mode}_$g.err; exit 1; }
    python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/reh_${mode}_$g.json') if l.startswith('{')][0])
print('ms_per_step %.4f  device %.4f  replayed %d  collectives/iter %.4f ms' % (j['ms_per_step'], j['stage_ms_per_iter'].get('total'), j['collectives']['blocks_replayed_with_their_collectives'], j['collectives']['ms_per_iter']), j.get('halo', {}).get('repartitions_in_timed_region'))
PY
  done
done
