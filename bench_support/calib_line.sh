#!/bin/bash
# FETCH_SIZE calibration on random line reads: how many bytes does the memory side move per random 128-byte line
# when a lane reads 16, 64 or all 128 bytes of it?   bench_support/calib_line.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/calib_line
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in 1 4 8; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$q -o f -- $R/bench_support/randline 100 8 $q 3 > $OUT/f$q.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/t$q -o t -- $R/bench_support/randline 100 8 $q 3 > $OUT/t$q.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/calib_line"
lines = 256 * 3 * 8 * 256 * 8 * 2   # blocks x threads x rounds x 2 lines per round
for q in (1, 4, 8):
    vals = {}
    for f in glob.glob(out + "/[ft]%d/*counter_collection.csv" % q):
        for r in csv.DictReader(open(f)):
            if "randline" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("nq=%d (%3d bytes of each line read): " % (q, 16 * q) + ", ".join("%s=%.3f per line" % (k, (sum(v) / len(v)) * (1024 if k == "FETCH_SIZE" else 1) / lines) for k, v in sorted(vals.items())))
PY
