#!/bin/bash
# round 3, GPU call 10: changed tests, smoke, the bench line with this round's traffic figures
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_k_tests.log 2>&1; tail -3 gpurun_out/r3_k_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_r03b.err | tail -1 > gpurun_out/bench_r03b.json; python -c "
import json;d=json.load(open('gpurun_out/bench_r03b.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['extra']['c3_match_all']['roofline']['traffic'], d['extra']['c5_150bp_l64']['roofline']['traffic'])"
