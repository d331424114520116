#!/bin/bash
# round 3, GPU call 8: overflow entries four at a time; 60/70 % A+T; the suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q > gpurun_out/r3_i_tests.log 2>&1; tail -3 gpurun_out/r3_i_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
$RG --share 0 --at 0.6 --check 1000000 2>gpurun_out/r3_i_at6.err | tail -1 | tee gpurun_out/r3_i_at6.json
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_phase.so $RG --share 0 --at 0.6 2>gpurun_out/r3_i_at6p.err | tail -1 | tee gpurun_out/r3_i_at6p.json
$RG --share 0 --at 0.7 2>gpurun_out/r3_i_at7.err | tail -1 | tee gpurun_out/r3_i_at7.json
$RG --share 0 --at 0.4 2>gpurun_out/r3_i_at4.err | tail -1 | tee gpurun_out/r3_i_at4.json
bash bench_support/run_quick.sh r3q8 2>&1 | tail -3
