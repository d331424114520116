#!/bin/bash
# round 3, GPU call 9: next list's rows requested behind the overflow prefetch -- A/B on one box (product vs the "early" variant)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_j_tests.log 2>&1; tail -3 gpurun_out/r3_j_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
for rep in 1 2; do
$RG --share 0 --at 0.6 2>gpurun_out/r3_j_at6.err | tail -1 | cut -c1-20,330-470
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_early.so $RG --share 0 --at 0.6 2>gpurun_out/r3_j_at6e.err | tail -1 | cut -c1-20,330-470
done
REAL_HIP_LIB=$R/real_amd/variants/libreal_hip_phase.so $RG --share 0 --at 0.6 2>gpurun_out/r3_j_at6p.err | tail -1 | tee gpurun_out/r3_j_at6p.json | cut -c1-300
timeout -k 10 500 python bench_support/ab_match.py --libs real_amd/libreal_hip.so real_amd/variants/libreal_hip_early.so --rounds 2 --steps 8 2>&1 | tail -6
