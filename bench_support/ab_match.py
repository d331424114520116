#!/usr/bin/env python3
"""A/B timing of builds of the matcher on ONE box (boxes differ by several per cent on this instruction-bound kernel,
so variants are only compared inside one gpurun call): every library given is run in a fresh process on the C2
workload, round-robin, and the HIP-event time of the match kernel is printed.

    make -C real_amd/csrc variant NAME=x DEFS=-DRH_ABLATE=1
    python bench_support/ab_match.py --libs real_amd/libreal_hip.so real_amd/variants/libreal_hip_x.so --rounds 2
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ONE = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
import bench
from real_amd import lib as rlib
from real_amd.matcher import HipMatcher, RealOptions
dev = torch.device("cuda", 0)
G, n, patl, seedl, totalk, mode = %(G)d, %(n)d, %(patl)d, %(seedl)d, %(totalk)d, %(mode)r
sym = bench.gen_genome(torch, G, 3, dev)
m = HipMatcher(RealOptions(seedl=seedl, seedkmax=2, totalkmax=totalk, scores=True).normalise(), device=0)
m.set_text_symbols(0, sym, np.array([0, G], dtype=np.uint64))
m.build_index_block()
bases, qual, _, _ = bench.gen_reads(torch, sym, n, patl, 0.02, 4, dev)
del sym
packed = bool(%(packed)d)
if packed:
    bases = bench.pack_bases(torch, bases, n, patl)
if mode == "unique":
    dt, ctr, (ms, ln), (rms, rn), _ = bench.timed_unique(torch, None, m, rlib, bases, qual, patl, n, %(steps)d, 2, 1, 0, dev, dev, packed=packed)
else:
    dt, ctr, kt, nh, _, _ = bench.timed_all(torch, None, m, rlib, bases, qual, patl, n, %(steps)d, 2, 1, 0, dev, dev)
    ms, ln = kt["match"]; rms, rn = kt["repeat"]
print(json.dumps({"kernel_ms": ms / max(ln, 1), "wave_pass_ms": rms / max(rn, 1), "ms_per_step": dt / %(steps)d * 1e3, "hits_per_read": ctr["hits"] / max(ctr["reads"], 1),
                  "per_wave": {k: v / max(ctr["reads"] / 64.0, 1) for k, v in ctr.items()}}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--genome-mbp", type=float, default=3000)
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--patl", type=int, default=100)
    ap.add_argument("--seedl", type=int, default=32)
    ap.add_argument("--totalk", type=int, default=3)
    ap.add_argument("--mode", default="unique")
    ap.add_argument("--packed", type=int, default=1, help="resident bases as 2 bits per base (the bench's headline format)")
    args = ap.parse_args()
    code = ONE % {"root": ROOT, "G": int(args.genome_mbp * 1e6), "n": args.reads, "patl": args.patl, "seedl": args.seedl, "totalk": args.totalk,
                  "mode": args.mode, "steps": args.steps, "packed": args.packed if args.mode == "unique" else 0}
    res = {l: [] for l in args.libs}
    for r in range(args.rounds):
        for l in args.libs:
            env = dict(os.environ, REAL_HIP_LIB=os.path.abspath(l))
            p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print(l, "FAILED", p.stderr[-1500:], flush=True)
                continue
            j = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
            res[l].append(j)
            print("round %d %-50s kernel %.3f ms  step %.3f ms  wave pass %.3f ms" % (r, os.path.basename(l), j["kernel_ms"], j["ms_per_step"], j["wave_pass_ms"]), flush=True)
            if "phase" in os.path.basename(l):   # a -DRH_PHASE_TIMING=1 build: the counters are ticks of 10 ns per wave
                pw = j["per_wave"]
                print("        per wave, us: front %.2f  waiting for rows %.2f  decoding %.2f  draining %.2f  staging qualities %.2f  scoring %.2f  whole wave %.2f" % tuple(
                    pw[k] / 100.0 for k in ("lookups", "probes", "candidates", "seedpass", "hits", "verified", "handed_over")), flush=True)
    print(json.dumps({os.path.basename(l): {"kernel_ms": [x["kernel_ms"] for x in v], "mean": sum(x["kernel_ms"] for x in v) / max(len(v), 1)} for l, v in res.items()}))


if __name__ == "__main__":
    main()
