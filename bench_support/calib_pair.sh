#!/bin/bash
# does the dependent second read of a random line (bench_support/randpair) miss the L2 again?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/calib_pair
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 0 1 3; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $OUT/t$m -o t -- $R/bench_support/randpair 100 8 $m 3 > $OUT/t$m.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/calib_pair"
lookups = 256 * 3 * 8 * 256 * 8 * 6
for m in (0, 1, 3):
    vals = {}
    for f in glob.glob(out + "/t%d/*counter_collection.csv" % m):
        for r in csv.DictReader(open(f)):
            if "randpair" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("mode=%d: " % m + ", ".join("%s=%.3f per lookup" % (k, (sum(v) / len(v)) / lookups) for k, v in sorted(vals.items())))
PY
