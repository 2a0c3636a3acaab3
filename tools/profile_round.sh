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
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/fetch.log 2>&1 || { echo "fetch run failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/write.log 2>&1 || { echo "write run failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-graph-pass > $OUT/sq.log 2>&1 || { echo "sq run failed"; exit 1; }
timeout -k 10 400 python3 bench.py --steps 50 --warmup 10 > $OUT/bench.log 2> $OUT/bench.err || { echo "bench run failed"; exit 1; }
tail -1 $OUT/bench.log | cut -c1-300
