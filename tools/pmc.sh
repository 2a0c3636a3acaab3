#!/bin/bash
# PMC passes for the bench workload (each counter group in its own rocprofv3 run, kernel-trace only).
# usage: tools/pmc.sh <outdir-under-gpurun_out> ; run ON the GPU box from the repo root.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/$name.log 2>&1 || { echo "$name failed"; exit 1; }; echo "$name ok"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $OUT/*/* | head -20
