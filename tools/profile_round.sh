#!/bin/bash
# Evidence run for profiles/: rocprofv3 kernel-trace stats of the default bench + PMC traffic passes (separate runs).
# A failed or timed-out step ends the script: no further GPU step is started after it.
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/trace.log 2>&1 || { echo "trace run failed"; exit 1; }
# counter passes, each twice: as the timed region runs (the attraction step inside the query launch: the dominant launch's counters) and
# with every stage a launch of its own (NW_ATTRACT_IN_NN=0: the per-kernel numbers)
for mode in fused apart; do
  if [ $mode = apart ]; then export NW_ATTRACT_IN_NN=0; else unset NW_ATTRACT_IN_NN; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$mode -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/fetch_$mode.log 2>&1 || { echo "fetch run ($mode) failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$mode -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/write_$mode.log 2>&1 || { echo "write run ($mode) failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$mode -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/sq_$mode.log 2>&1 || { echo "sq run ($mode) failed"; exit 1; }
done
unset NW_ATTRACT_IN_NN
timeout -k 10 400 python3 bench.py --steps 50 --warmup 10 > $OUT/bench.log 2> $OUT/bench.err || { echo "bench run failed"; exit 1; }
tail -1 $OUT/bench.log | cut -c1-300
