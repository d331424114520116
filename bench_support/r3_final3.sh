#!/bin/bash
# round 3, final measurements 3: the bench line with the final kernel's traffic figures, then final2 (full parity, non-i.i.d. genomes, CLI)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/bench_r03c.err | tail -1 > gpurun_out/bench_r03c.json; python -c "
import json;d=json.load(open('gpurun_out/bench_r03c.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['extra']['c3_match_all']['roofline']['traffic'], d['extra']['c5_150bp_l64']['roofline']['traffic'])"
bash bench_support/r3_final2.sh
