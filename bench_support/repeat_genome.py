#!/usr/bin/env python3
"""How the matcher takes a genome with repeats (the BASELINE genome is i.i.d. random and has none): a synthetic genome in
which a given share of the positions lies in families of exact copies of 1 kbp segments, reads sampled uniformly.  Prints
the step time, the time of the lane-per-read and of the wave-per-read kernel, and the share of reads handed over.

    python bench_support/repeat_genome.py [--genome-mbp 1000] [--reads 20000000] [--share 0.1] [--copies 5]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=1000)
    ap.add_argument("--reads", type=int, default=20_000_000)
    ap.add_argument("--share", type=float, default=0.1, help="share of the genome that lies in repeat copies")
    ap.add_argument("--copies", type=int, default=5)
    ap.add_argument("--seg", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--errprob", type=float, default=0.02)
    ap.add_argument("--check", type=int, default=0, help="compare this many reads of the step (strided) with the CPU port of the oracle on the same genome and index")
    ap.add_argument("--at", type=float, default=0.5, help="share of A+T in the (i.i.d.) genome: 0.5 = uniform; 0.8 = an AT-rich genome with its uneven buckets")
    ap.add_argument("--fragments", type=int, default=1, help="the genome as this many fragments (chromosomes, scaffolds)")
    ap.add_argument("--n-runs", type=int, default=0, help="runs of 1000 N each")
    ap.add_argument("--table-kind", type=int, default=0)
    ap.add_argument("--prefix-bits", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from real_amd import lib as rlib
    from real_amd.matcher import HipMatcher, RealOptions
    dev = torch.device("cuda", 0)
    G, n, patl = int(args.genome_mbp * 1e6), args.reads, 100
    sym = bench.gen_genome(torch, G, 3, dev)
    if args.at != 0.5:           # A, T with probability at/2 each, C, G with (1 - at)/2 each
        for lo in range(0, G, 1 << 28):
            hi = min(G, lo + (1 << 28))
            u = torch.rand(hi - lo, device=dev)
            at = u < args.at
            sym[lo:hi] = torch.where(at, torch.where(u < args.at / 2, 0, 3), torch.where(u < args.at + (1 - args.at) / 2, 1, 2)).to(torch.uint8)
            del u, at
    fam = int(G * args.share / (args.copies * args.seg))
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    ar = torch.arange(args.seg, device=dev)
    src = (torch.randint(0, G - args.seg, (fam,), generator=g)).to(dev)
    for c in range(args.copies - 1):           # every family: the source segment and copies - 1 exact copies elsewhere
        dst = (torch.randint(0, G - args.seg, (fam,), generator=g)).to(dev)
        sym[(dst[:, None] + ar[None, :]).reshape(-1)] = sym[(src[:, None] + ar[None, :]).reshape(-1)]
    torch.cuda.synchronize(); torch.cuda.empty_cache()   # (the index wants the memory torch's allocator would keep)
    m = HipMatcher(RealOptions(seedl=32, seedkmax=2, totalkmax=3, scores=True).normalise(), device=0, table_kind=args.table_kind, prefix_bits=args.prefix_bits)
    if args.n_runs:
        ns = torch.randint(0, G - 1000, (args.n_runs,), generator=g).to(dev)
        sym[(ns[:, None] + ar[None, :1000]).reshape(-1)] = 4
    cuts = np.unique(np.random.default_rng(5).integers(1000, G - 1000, size=max(args.fragments - 1, 0))) if args.fragments > 1 else np.zeros(0, dtype=np.int64)
    frag = np.concatenate([[0], cuts, [G]]).astype(np.uint64)
    m.set_text_symbols(0, sym, frag)
    m.build_index_block()
    bases, qual, _, _ = bench.gen_reads(torch, sym, n, patl, args.errprob, 4, dev)
    sym_host = sym.cpu().numpy() if args.check else None
    del sym
    packed = not args.n_runs   # (reads that hold an N cannot be packed without their flags: bytes then)
    pk = bench.pack_bases(torch, bases, n, patl) if packed else bases
    dt, ctr, (ms, ln), (rms, rn), (info, score) = bench.timed_unique(torch, None, m, rlib, pk, qual, patl, n, args.steps, 1, 1, 0, dev, dev, packed=packed)
    st = (info >> 61) & 7
    parity = None
    if args.check:          # the oracle's CPU port on the same genome and the same index (the six lists downloaded from the device)
        opts = RealOptions(seedl=32, seedkmax=2, totalkmax=3, scores=True).normalise()
        cpu = bench.CpuSide(m, sym_host, frag, opts, 16)
        b, q, off, stride, k1 = bench.strided_sample(torch, bases, qual, n, patl, args.check)
        oinfo, oscore, _ = cpu.ora.match_unique(cpu.og, cpu.ix, cpu.params(opts), b, q, off)
        gi = info[::stride][:k1].contiguous().cpu().numpy().view(np.uint64)
        gs = score[::stride][:k1].contiguous().cpu().numpy()
        parity = bool(np.array_equal(gi, oinfo) and np.array_equal(gs.view(np.uint32), oscore.view(np.uint32)))
    extra = {"work_per_read": {k: ctr[k] / max(ctr["reads"], 1) for k in ("lookups", "probes", "candidates", "seedpass", "hits", "verified")}}
    if "phase" in os.path.basename(os.environ.get("REAL_HIP_LIB", "")):   # a -DRH_PHASE_TIMING=1 build: the counters are ticks of 10 ns per wave
        extra = {"per_wave_us": dict(zip(("front", "waiting_for_rows", "decoding", "draining", "staging_qualities", "scoring", "whole_tile"),
                                         (ctr[k] / max(ctr["reads"] / 64.0, 1) / 100.0 for k in ("lookups", "probes", "candidates", "seedpass", "hits", "verified", "handed_over"))))}
    print(json.dumps({**extra, "genome_mbp": args.genome_mbp, "reads": n, "share_in_repeats": args.share, "copies": args.copies,
                      "ms_per_step": dt / args.steps * 1e3, "reads_per_s": n * args.steps / dt,
                      "lane_kernel_ms": ms / max(ln, 1), "wave_kernel_ms": rms / max(rn, 1),
                      "table_kind": m.table_kind, "prefix_bits": m.prefix_bits, "at": args.at, "fragments": args.fragments, "n_runs": args.n_runs, "parity_with_cpu_port": parity, "checked_reads": args.check, "handed_over_frac": ctr["handed_over"] / max(ctr["reads"], 1),
                      "nonunique_frac": float((st == 4).float().mean().item()), "unique_frac": float(((st == 1) | (st == 2)).float().mean().item())}))


if __name__ == "__main__":
    main()
