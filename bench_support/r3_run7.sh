#!/bin/bash
# round 3, GPU call 7: two-phase decode, early hand-over, second pass with qualities in LDS
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_h_tests.log 2>&1; tail -3 gpurun_out/r3_h_tests.log
bash bench_support/run_quick.sh r3q7 2>&1 | tail -3
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
$RG --share 0.1 --copies 5 --check 2000000 2>gpurun_out/r3_h_c5.err | tail -1 | tee gpurun_out/r3_h_c5.json
$RG --share 0.1 --copies 8 2>gpurun_out/r3_h_c8.err | tail -1 | tee gpurun_out/r3_h_c8.json
$RG --share 0 --at 0.6 --check 1000000 2>gpurun_out/r3_h_at6.err | tail -1 | tee gpurun_out/r3_h_at6.json
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_phase.so $RG --share 0 --at 0.6 2>gpurun_out/r3_h_at6p.err | tail -1 | tee gpurun_out/r3_h_at6p.json
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_phase.so $RG --share 0 2>gpurun_out/r3_h_at5p.err | tail -1 | tee gpurun_out/r3_h_at5p.json
timeout -k 10 400 python bench_support/fuzz_parity.py --seconds 150 --seed 9 --copy-prob 0.5 > gpurun_out/r3_fuzz9.log 2>&1; tail -2 gpurun_out/r3_fuzz9.log
