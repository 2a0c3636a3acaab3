#!/bin/bash
# C1 / C2: localizations per wave of the query against iteration time (NW_ITEM_PTS), stage times at level 2
cd "$(dirname "$0")/../.."
for cfg in c1 c2; do
  for pts in 32 16 8; do
    NW_ITEM_PTS=$pts python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/items_${cfg}_$pts.json 2> gpurun_out/items_${cfg}_$pts.err || exit 1
    python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/items_${cfg}_$pts.json') if l.startswith('{')][0])
print('$cfg pts/wave $pts: %.4f ms/step, graph %.4f, stages' % (j['ms_per_step'], j['ms_per_step_graph_replay']), {k: round(v*1e3,1) for k,v in j['stage_ms_per_iter'].items()})
PY
  done
done
