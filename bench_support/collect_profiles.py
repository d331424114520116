#!/usr/bin/env python3
"""Copy the judged summaries of one bench_support/profile.sh run into profiles/.

    python bench_support/collect_profiles.py r03 c2 gpurun_out/prof_r03c2 [gpurun_out/bench_r03.json]
    python bench_support/collect_profiles.py r03 c3 gpurun_out/prof_r03c3
    python bench_support/collect_profiles.py r03 c5 gpurun_out/prof_r03c5

tag = the BASELINE configuration the passes ran (c2: bench.py's default; c3: --mode all --totalk 2; c5: --patl 150
--seedl 64 --totalk 5).  Writes profiles/<round>_<tag>_{bench_under_rocprof_trace.json, kernel_stats.csv, counters.txt}
(+ <round>_bench.json for c2 when a bench line is given) and adds the configuration's entry to profiles/traffic.json
(HBM bytes per launch of its match kernel, read by bench.py for roofline.traffic).
"""
import collections, csv, glob, json, os, shutil, subprocess, sys

rnd, tag, pdir = sys.argv[1], sys.argv[2], sys.argv[3]
bench_json = sys.argv[4] if len(sys.argv) > 4 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
CFG = {"c2": ("unique", 100, 32, 3, "false"), "c3": ("all", 100, 32, 2, "true"), "c5": ("unique", 150, 64, 5, "false")}
mode, patl, seedl, totalk, all_flag = CFG[tag]
under = json.loads([l for l in open(os.path.join(pdir, "bench_under_trace.json")) if l.startswith("{")][-1])
W = (patl + 31) // 32                                      # match_kernel<W, scores, all, table kind>: the lane-per-read matcher
stats = os.path.join(pdir, "trace", "trace_kernel_stats.csv")
kname = None
for r in csv.DictReader(open(stats)):
    # (the first pass: `..., false>`; the same template with `true` is the second pass over the few reads that outgrow a lane)
    if r["Name"].startswith("void match_kernel<%d, true, %s," % (W, all_flag)) and r["Name"].rstrip().endswith("false>(MatchArgs)"):
        kname, avg_ms, calls = r["Name"], float(r["AverageNs"]) / 1e6, int(r["Calls"])
assert kname, "match kernel not in " + stats
sel = kname[len("void "):]

if bench_json:
    shutil.copy(bench_json, os.path.join(out, rnd + "_bench.json"))
shutil.copy(os.path.join(pdir, "bench_under_trace.json"), os.path.join(out, "%s_%s_bench_under_rocprof_trace.json" % (rnd, tag)))
shutil.copy(stats, os.path.join(out, "%s_%s_kernel_stats.csv" % (rnd, tag)))
txt = subprocess.run([sys.executable, os.path.join(ROOT, "bench_support", "parse_prof.py"), pdir, sel],
                     stdout=subprocess.PIPE, check=True).stdout.decode()
open(os.path.join(out, "%s_%s_counters.txt" % (rnd, tag)), "w").write("# per-launch averages of `%s`\n" % sel + txt)

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
SHA = bench_py.kernel_source_hash(os.environ.get("KERNEL_REV") or None)      # KERNEL_REV=<commit>: the passes ran on that commit's kernel sources
cfg = under["config"]
n = cfg.get("reads_per_gpu_per_step") or int(cfg["workload"].split("M synthetic")[0].split()[-1]) * 1_000_000
gmbp = cfg.get("genome_bp", 0) // 1_000_000 or int(float(cfg["workload"].split(" vs ")[1].split(" Mbp")[0]))
fmt = "bytes" if cfg.get("input_format", "2-bit").startswith("one") else "packed"
key = bench_py.traffic_key(mode, patl, seedl, totalk, gmbp, n, fmt)
tfile = os.path.join(out, "traffic.json")
tj = json.load(open(tfile)) if os.path.exists(tfile) else {}
tj = {k: v for k, v in tj.items() if v.get("kernel_source_sha") == SHA}      # entries of another kernel source are of no use
tj[key] = {
    "hbm_bytes_per_launch": hbm, "kernel": sel, "round": int(rnd.lstrip("r")), "kernel_source_sha": SHA,
    "FETCH_SIZE_KiB_per_launch": avg["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": avg["WRITE_SIZE"],
    "correction": "reads = 2 x FETCH_SIZE x 1024 (gfx950: requests tallied at 64 B, L2 lines are 128 B; MI355X_MICROARCH.md "
                  "section HBM), writes = WRITE_SIZE x 1024; separate --pmc passes (bench_support/profile.sh)",
    "TCC_MISS_per_launch": avg.get("TCC_MISS_sum"), "TCC_REQ_per_launch": avg.get("TCC_REQ_sum"),
    "TCC_MISS_per_launch_shuffled_reads": avg_shuf.get("TCC_MISS_sum"),
    "avg_launch_ms_under_trace": avg_ms, "launches_under_trace": calls}
json.dump(tj, open(tfile, "w"), indent=1)
print(json.dumps(tj[key], indent=1))
