#!/bin/bash
# memory-pipeline counters of the match kernel: bench_support/profile_mem.sh <tag> [bench args]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $OUT/ta -o ta -- python3 $R/bench.py $ARGS > $OUT/ta.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -o tcp -- python3 $R/bench.py $ARGS > $OUT/tcp.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $OUT/ea -o ea -- python3 $R/bench.py $ARGS > $OUT/ea.log 2>&1
rocprofv3 --pmc TCC_BUSY_avr TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $OUT/tcc2 -o tcc2 -- python3 $R/bench.py $ARGS > $OUT/tcc2.log 2>&1
