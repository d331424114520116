#!/bin/bash
# round 3, GPU call 2: the new tests, then the index build with / without the helper-thread allocation of the rows
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_c_tests.log 2>&1; tail -3 gpurun_out/r3_c_tests.log
for t in 0 1 6 1 0; do
  REAL_HIP_PREALLOC_THREADS=$t timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras off 2>gpurun_out/r3_prealloc_$t.err | tail -1 > gpurun_out/r3_prealloc_$t.json
  python -c "
import json;d=json.load(open('gpurun_out/r3_prealloc_$t.json'));print('prealloc threads $t', d['ms_per_step'], json.dumps({k:(round(v,2) if isinstance(v,float) else v) for k,v in d['config']['index_build'].items() if k!='note'}))"
done
