#!/bin/bash
# One-GPU rehearsal of the multi-rank block (ONE rank over RCCL: NW_BENCH_FORCE_DIST=1 takes bench.py's N > 1 path with the library's own
# communicator) beside the plain single-GPU run of the same workload.  usage (GPU box, repo root): tools/graph_rehearsal.sh [bench args]
set -u
mkdir -p gpurun_out
run() {
    tag=$1; shift
    env "$@" timeout -k 10 300 python3 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline $EXTRA > gpurun_out/reh_$tag.json 2> gpurun_out/reh_$tag.err || { tail -20 gpurun_out/reh_$tag.err; exit 1; }
    python3 - gpurun_out/reh_$tag.json $tag <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-14s ms_per_step %.4f  device %.4f  host per block %s' % (sys.argv[2], j['ms_per_step'], j['stage_ms_per_iter'].get('total'), j.get('halo', {}).get('host_ms_per_block')))
PY
}
EXTRA="$*"
run single NW_X=0
EXTRA="$* --mode tiles"; run rccl1_tiles NW_BENCH_FORCE_DIST=1
EXTRA="$* --mode halo"; run rccl1_halo NW_BENCH_FORCE_DIST=1
