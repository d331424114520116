#!/bin/bash
# round 3, final measurements 2: all 50 M reads of a step against the CPU port; the non-i.i.d. genomes; the CLI at C2 size
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python bench.py --steps 5 --warmup 2 --cpu-seconds 60 --extras off 2>gpurun_out/bench_r03_full_parity.err | tail -1 > gpurun_out/bench_r03_full_parity.json; tail -2 gpurun_out/bench_r03_full_parity.err
RG="timeout -k 10 300 python bench_support/repeat_genome.py --genome-mbp 3000 --reads 50000000"
( $RG --share 0 ; $RG --share 0.1 --copies 2 ; $RG --share 0.1 --copies 4 ; $RG --share 0.1 --copies 5 --check 2000000 ; $RG --share 0.1 --copies 8 ; $RG --share 0.05 --copies 16 ; $RG --share 0 --at 0.6 --check 1000000 ; $RG --share 0 --at 0.4 ; $RG --share 0 --at 0.7 ) 2>gpurun_out/r3_final_rg.err | grep '^{' > gpurun_out/r3_final_rg.jsonl; wc -l gpurun_out/r3_final_rg.jsonl
timeout -k 10 400 python bench_support/cli_midsize.py --reads 50000000 --genome-mbp 3000 --out gpurun_out/r3_cli_c2size.json > gpurun_out/r3_cli_c2size.log 2>&1; tail -c 600 gpurun_out/r3_cli_c2size.json
