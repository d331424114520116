#!/bin/bash
# round 3, GPU call 12: new tests; a genome that is skewed AND repetitive; 70 % A+T with the final wave kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_cli_gpu.py -m gpu -x -q > gpurun_out/r3_m_tests.log 2>&1; tail -3 gpurun_out/r3_m_tests.log
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
( $RG --share 0.1 --copies 5 --at 0.6 --check 1000000 ; $RG --share 0 --at 0.7 ; $RG --share 0.2 --copies 3 --at 0.6 ) 2>gpurun_out/r3_m_rg.err | grep '^{' > gpurun_out/r3_m_rg.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r3_m_rg.jsonl'):
    j=json.loads(l)
    print("share %.2f copies %d at %.1f : %.2f ms  lane %.2f  2nd+wave %.2f  handed %.4f  parity %s" % (j['share_in_repeats'], j['copies'], j['at'], j['ms_per_step'], j['lane_kernel_ms'], j['wave_kernel_ms'], j['handed_over_frac'], j['parity_with_cpu_port']))
PY
