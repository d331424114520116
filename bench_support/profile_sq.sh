#!/bin/bash
# extra SQ counter passes for the match kernel: bench_support/profile_sq.sh <tag> [bench args]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -o tcc -- python3 $R/bench.py $ARGS > $OUT/tcc.log 2>&1
