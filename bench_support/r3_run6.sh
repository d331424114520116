#!/bin/bash
# round 3, GPU call 6: the second pass of the lane matcher (reads on many copies)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_g_tests.log 2>&1; tail -3 gpurun_out/r3_g_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
$RG --share 0.1 --copies 5 --check 2000000 2>gpurun_out/r3_g_c5.err | tail -1 | tee gpurun_out/r3_g_c5.json
$RG --share 0.1 --copies 8 --check 2000000 2>gpurun_out/r3_g_c8.err | tail -1 | tee gpurun_out/r3_g_c8.json
$RG --share 0.05 --copies 16 2>gpurun_out/r3_g_c16.err | tail -1 | tee gpurun_out/r3_g_c16.json
bash bench_support/run_quick.sh r3q6 2>&1 | tail -3
timeout -k 10 400 python bench_support/fuzz_parity.py --seconds 200 --seed 8 --copy-prob 0.6 > gpurun_out/r3_fuzz8.log 2>&1; tail -2 gpurun_out/r3_fuzz8.log
