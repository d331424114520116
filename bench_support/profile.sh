#!/bin/bash
# rocprofv3 recipe for bench.py (run on the GPU box through gpurun):
#   bench_support/profile.sh <tag> [bench args...]      e.g.  profile.sh r03c2 ; profile.sh r03c3 --mode all --totalk 2 ;
#                                                             profile.sh r03c5 --patl 150 --seedl 64 --totalk 5
# kernel trace + stats in one run, PMC counters in runs of their own (never combined with traces);
# writes gpurun_out/prof_<tag>/ ; bench_support/parse_prof.py summarises into profiles/.
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --extras off $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -o tcc -- python3 $R/bench.py $ARGS > $OUT/tcc.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -o lds -- python3 $R/bench.py $ARGS > $OUT/lds.log 2>&1
grep -h '^{' $OUT/trace.log > $OUT/bench_under_trace.json || true
# the same step on the reads in random order: L2 misses (lines per read) beside the sorted figure (C2 only: no bench args)
if [ $# -eq 0 ]; then rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_shuffled -o tcc -- python3 $R/bench.py $ARGS --shuffle-reads > $OUT/tcc_shuffled.log 2>&1; fi
