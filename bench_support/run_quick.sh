#!/bin/bash
# quick A/B on the GPU box: bench_support/run_quick.sh [tag]   (C2, C3, C5 without the CPU baseline)
T=${1:-q}
for cfg in "" "--mode all --totalk 2" "--patl 150 --seedl 64 --totalk 5"; do
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $cfg 2>gpurun_out/$T.err | tail -1 > gpurun_out/$T.json && python -c "
import json;d=json.load(open('gpurun_out/$T.json'))
print('$cfg', d.get('ms_per_step'), d.get('roofline',{}).get('avg_launch_ms'), {k:v for k,v in d.items() if k.endswith('_ms') or k.endswith('per_s')})"
done
