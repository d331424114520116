#!/bin/bash
# round 3, GPU call 13: wave kernel -- range descriptions cached across rounds, score memo across rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_n_tests.log 2>&1; tail -3 gpurun_out/r3_n_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
( $RG --share 0 --at 0.7 ; $RG --share 0.05 --copies 16 --check 1000000 ; $RG --share 0.02 --copies 64 ; $RG --share 0.1 --copies 8 ) 2>gpurun_out/r3_n_rg.err | grep '^{' > gpurun_out/r3_n_rg.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r3_n_rg.jsonl'):
    j=json.loads(l)
    print("share %.2f copies %d at %.1f : %.2f ms  lane %.2f  2nd+wave %.2f  handed %.4f  parity %s" % (j['share_in_repeats'], j['copies'], j['at'], j['ms_per_step'], j['lane_kernel_ms'], j['wave_kernel_ms'], j['handed_over_frac'], j['parity_with_cpu_port']))
PY
timeout -k 10 300 python bench_support/fuzz_parity.py --seconds 150 --seed 11 --copy-prob 0.6 > gpurun_out/r3_fuzz11.log 2>&1; tail -1 gpurun_out/r3_fuzz11.log
