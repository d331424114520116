#!/bin/bash
# round 3, GPU call 11: wave kernel with flattened rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r3_l_tests.log 2>&1; tail -3 gpurun_out/r3_l_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
( $RG --share 0.05 --copies 16 --check 2000000 ; $RG --share 0.1 --copies 8 ; $RG --share 0.1 --copies 5 ; $RG --share 0 --at 0.7 --check 500000; $RG --share 0.02 --copies 64 ) 2>gpurun_out/r3_l_rg.err | grep '^{' > gpurun_out/r3_l_rg.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r3_l_rg.jsonl'):
    j=json.loads(l)
    print("share %.2f copies %d at %.1f : %.2f ms  lane %.2f  2nd+wave %.2f  handed %.4f  parity %s" % (j['share_in_repeats'], j['copies'], j['at'], j['ms_per_step'], j['lane_kernel_ms'], j['wave_kernel_ms'], j['handed_over_frac'], j['parity_with_cpu_port']))
PY
timeout -k 10 400 python bench_support/fuzz_parity.py --seconds 200 --seed 10 --copy-prob 0.6 > gpurun_out/r3_fuzz10.log 2>&1; tail -2 gpurun_out/r3_fuzz10.log
