#!/bin/bash
# one bench line per knob setting: usage r05_knob.sh <config> "<ENV...>" ...
cfg=$1; shift
for v in "$@"; do
  env $v timeout -k 10 300 python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > /tmp/o.log 2>/dev/null || { echo "$cfg $v failed"; continue; }
  python3 - "$cfg" "$v" <<'PY'
import json,sys
d=json.loads(open('/tmp/o.log').read().strip().splitlines()[-1])
print('%s %-28s: step %.4f graph %.4f fused launch %.4f ms | apart: nn %.4f total %.4f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d.get('ms_per_step_graph_replay') or 0, d['roofline']['avg_launch_ms'], d['stage_ms_per_iter']['nn'], d['stage_ms_per_iter']['total']))
PY
done
