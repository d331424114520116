#!/bin/bash
# round 3, GPU call 1: host CPU facts, the GPU suite, rocprof passes of C2 / C3 / C5
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import bench, json, os; print(json.dumps(bench.host_cpu_info())); print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max'); print(len(os.sched_getaffinity(0)))" > gpurun_out/r3_cpuinfo.txt 2>&1
nproc >> gpurun_out/r3_cpuinfo.txt; free -g >> gpurun_out/r3_cpuinfo.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r3_b_tests.log 2>&1; tail -3 gpurun_out/r3_b_tests.log
bash bench_support/profile.sh r03c2 && echo c2 profiled
bash bench_support/profile.sh r03c3 --mode all --totalk 2 && echo c3 profiled
bash bench_support/profile.sh r03c5 --patl 150 --seedl 64 --totalk 5 && echo c5 profiled
