#!/bin/bash
# round 3, final measurements 1: the suite, the bench line, rocprof passes of C2 / C3 / C5 on the final kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests.log 2>&1; tail -3 gpurun_out/r3_final_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_r03.err | tail -1 > gpurun_out/bench_r03.json; tail -3 gpurun_out/bench_r03.err; cut -c1-400 gpurun_out/bench_r03.json
bash bench_support/profile.sh r03c2 && echo c2 profiled
bash bench_support/profile.sh r03c3 --mode all --totalk 2 && echo c3 profiled
bash bench_support/profile.sh r03c5 --patl 150 --seedl 64 --totalk 5 && echo c5 profiled
