#!/bin/bash
# round 3, GPU call 4: wave matcher with the one-round path; where the 60 % A+T step spends its time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_e_tests.log 2>&1; tail -3 gpurun_out/r3_e_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
$RG --share 0.1 --copies 5 --check 2000000 2>gpurun_out/r3_e_c5.err | tail -1 | tee gpurun_out/r3_e_c5.json
$RG --share 0.1 --copies 8 2>gpurun_out/r3_e_c8.err | tail -1 | tee gpurun_out/r3_e_c8.json
$RG --share 0 --at 0.6 2>gpurun_out/r3_e_at6.err | tail -1 | tee gpurun_out/r3_e_at6.json
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_phase.so $RG --share 0 --at 0.6 2>gpurun_out/r3_e_at6p.err | tail -1 | tee gpurun_out/r3_e_at6p.json
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_phase.so $RG --share 0 2>gpurun_out/r3_e_at5p.err | tail -1 | tee gpurun_out/r3_e_at5p.json
$RG --share 0 --at 0.7 2>gpurun_out/r3_e_at7.err | tail -1 | tee gpurun_out/r3_e_at7.json
