#!/bin/bash
# SQ counter passes for the NN query on the bench workload (each group its own rocprofv3 run, kernel-trace only).
# usage: tools/pmc_nn.sh <outdir-under-gpurun_out> ; run ON the GPU box from the repo root.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-graph-pass > $OUT/$name.log 2>&1 || { echo "$name failed"; exit 1; }; echo "$name ok"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH
python3 tools/pmc_summary.py $OUT/sq1 $OUT/sq2 > $OUT/summary.json
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k in d:
    if 'nn_wave' in k or 'attract' in k: print(k, json.dumps(d[k]))
PY
