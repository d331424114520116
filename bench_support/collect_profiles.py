#!/usr/bin/env python3
"""Copy the judged summaries of one bench_support/profile.sh run into profiles/.

    python bench_support/collect_profiles.py r01 gpurun_out/prof_r01g gpurun_out/bench_r01g.json

Writes profiles/<round>_{bench.json, bench_under_rocprof_trace.json, kernel_stats.csv, counters.txt} and
profiles/traffic.json (HBM bytes per launch of the dominant kernel, read by bench.py for roofline.traffic).
"""
import collections, csv, glob, json, os, shutil, subprocess, sys

rnd, pdir, bench_json = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
W = (bench["config"]["read_len"] + 31) // 32                # match_kernel<W, scores, all, table kind>: the lane-per-read matcher
kname = None
stats = os.path.join(pdir, "trace", "trace_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
for r in rows:
    if r["Name"].startswith("void match_kernel<%d, true, false," % W):
        kname = r["Name"]
        avg_ms = float(r["AverageNs"]) / 1e6
        calls = int(r["Calls"])
assert kname, "match kernel not in " + stats
sel = kname[len("void "):]

shutil.copy(bench_json, os.path.join(out, rnd + "_bench.json"))
shutil.copy(os.path.join(pdir, "bench_under_trace.json"), os.path.join(out, rnd + "_bench_under_rocprof_trace.json"))
shutil.copy(stats, os.path.join(out, rnd + "_kernel_stats.csv"))
txt = subprocess.run([sys.executable, os.path.join(ROOT, "bench_support", "parse_prof.py"), pdir, sel],
                     stdout=subprocess.PIPE, check=True).stdout.decode()
open(os.path.join(out, rnd + "_counters.txt"), "w").write("# per-launch averages of `%s`\n" % sel + txt)

pmc, shuf = collections.defaultdict(list), collections.defaultdict(list)
for f in glob.glob(os.path.join(pdir, "*", "*_counter_collection.csv")):
    into = shuf if os.path.basename(os.path.dirname(f)).endswith("_shuffled") else pmc        # (the pass on shuffled reads is a figure of its own)
    for r in csv.DictReader(open(f)):
        if sel in r["Kernel_Name"]:
            into[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in pmc.items()}
avg_shuf = {k: sum(v) / len(v) for k, v in shuf.items()}
hbm = 2 * avg["FETCH_SIZE"] * 1024 + avg["WRITE_SIZE"] * 1024
sys.path.insert(0, ROOT)
import bench as bench_py
fmt = "packed" if bench["config"].get("input_format", "").startswith("2-bit") else "bytes"
key = "match_unique_%dMbp_%dreads_%s" % (bench["config"]["genome_bp"] // 1_000_000, bench["config"]["reads_per_gpu_per_step"], fmt)
tj = {key: {
    "hbm_bytes_per_launch": hbm, "kernel": sel, "round": int(rnd.lstrip("r")), "kernel_source_sha": bench_py.kernel_source_hash(),
    "FETCH_SIZE_KiB_per_launch": avg["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": avg["WRITE_SIZE"],
    "correction": "reads = 2 x FETCH_SIZE x 1024 (gfx950: requests tallied at 64 B, L2 lines are 128 B; MI355X_MICROARCH.md "
                  "section HBM), writes = WRITE_SIZE x 1024; separate --pmc passes (bench_support/profile.sh)",
    "TCC_MISS_per_launch": avg.get("TCC_MISS_sum"), "TCC_REQ_per_launch": avg.get("TCC_REQ_sum"),
    "TCC_MISS_per_launch_shuffled_reads": avg_shuf.get("TCC_MISS_sum"),
    "avg_launch_ms_under_trace": avg_ms, "launches_under_trace": calls}}
json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(tj[key], indent=1))
