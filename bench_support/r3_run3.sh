#!/bin/bash
# round 3, GPU call 3: rows addressed by the mixed signature -- the suite, a fuzz campaign, skewed composition, C2/C3/C5 quick
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_d_tests.log 2>&1; tail -3 gpurun_out/r3_d_tests.log
timeout -k 10 400 python bench_support/fuzz_parity.py --seconds 240 --seed 7 > gpurun_out/r3_fuzz7.log 2>&1; tail -2 gpurun_out/r3_fuzz7.log
bash bench_support/run_quick.sh r3q3 2>&1 | tail -3
for at in 0.6 0.7; do
  timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000 --share 0 --at $at --check 1000000 2>gpurun_out/r3_at$at.err | tail -1 > gpurun_out/r3_at$at.json; cat gpurun_out/r3_at$at.json
done
timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000 --share 0.1 --copies 5 2>gpurun_out/r3_c5copy.err | tail -1 > gpurun_out/r3_c5copy.json; cat gpurun_out/r3_c5copy.json
