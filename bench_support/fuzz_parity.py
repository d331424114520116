#!/usr/bin/env python3
"""Randomised parity campaign: many small random configurations (seed length, read length, mismatch bounds, scores, table
kind, genomes with repeats / N runs / fragments / skewed composition, reads with planted near-copies that differ from them in
different seed segments (scores on: the order of the update() calls decides the record), ragged and uniform batches, packed and byte bases,
matchUnique and matchAll), the HIP path through the C ABI against the oracle.  Not part of the test suite (run time is
open-ended); prints every configuration that differs and exits 1 if any did.

    python bench_support/fuzz_parity.py [--seconds 300] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--copy-prob", type=float, default=0.3, help="share of the configurations whose genome holds exact copies of a segment")
    ap.add_argument("--skew-prob", type=float, default=0.2, help="share of the configurations whose genome has 70 .. 95 %% A+T (equal ranges of tens to hundreds of entries)")
    ap.add_argument("--diverged-prob", type=float, default=0.3, help="share of the uniform-length configurations whose reads have planted near-copies "
                    "(1-2 substitutions in chosen seed segments, low qualities there): the class on which the order of the update() calls decides")
    ap.add_argument("--only", type=int, nargs="*", default=[], help="run only these iterations of the seed (the others are generated and skipped) and say what differs")
    args = ap.parse_args()
    import torch
    torch.cuda.init()       # (before the library opens its own contexts: afterwards torch finds no device in this process)
    import oracle_lib as ora
    from real_amd import synth
    from real_amd.matcher import AllMatcher, RealOptions, UniqueMatcher
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    it = bad = 0
    while time.time() - t0 < args.seconds and (not args.only or it < max(args.only)):
        it += 1
        seedl = int(rng.choice([8, 12, 16, 20, 24, 28, 32, 36, 48, 64]))
        patl = int(rng.integers(seedl, min(seedl + 200, 330)))
        k = int(rng.integers(0, 9))
        seedk = int(min(rng.integers(0, 3), k))
        scores = int(rng.integers(0, 2))
        n = int(rng.choice([500, 3000, 20000, 200000])) if seedl >= 16 else int(rng.choice([300, 2000]))
        if seedl <= 12:
            n = min(n, 3000)
        repeats = int(rng.choice([0, 5, 40]))
        g = synth.random_genome(n, seed=int(rng.integers(1 << 30)), n_frag=int(rng.choice([1, 2, 7])), n_runs=int(rng.choice([0, 3, 20])),
                                repeats=repeats, repeat_len=int(rng.choice([150, 400, 1200])))
        if rng.random() < args.skew_prob:  # a skewed base composition: long equal ranges whose entries nearly all fail the partner filter
            at = float(rng.choice([0.7, 0.85, 0.95]))
            keep_n = g.sym > 3
            g.sym[:] = rng.choice(np.array([0, 1, 2, 3], dtype=np.uint8), size=g.n, p=[at / 2, (1 - at) / 2, (1 - at) / 2, at / 2])
            g.sym[keep_n] = 4
        if rng.random() < args.copy_prob:  # exact copies of a segment: reads on them have several locations
            L = int(min(rng.choice([300, 1000]), n // 8))
            src = int(rng.integers(0, n - L))
            # (1..6 copies: the first pass' and, from five on, the second pass' reads; now and then a dozen or thirty: the second
            # pass at its limit and the wave-per-read kernel)
            for _ in range(int(rng.choice([1, 2, 3, 4, 5, 6, 6, 8, 13, 26, 31]))):
                d = int(rng.integers(0, n - L))
                g.sym[d:d + L] = g.sym[src:src + L]
        nreads = int(rng.choice([64, 257, 1500]))
        ragged = rng.random() < 0.3
        diverged = False
        if ragged:
            parts = [synth.sample_reads(g, max(1, nreads // 3), int(rng.integers(max(4, seedl - 3), patl + 1)), 0.02, seed=int(rng.integers(1 << 30)),
                                        n_read_prob=0.001) for _ in range(3)]
            b = synth.concat_batches(parts)
        elif rng.random() < args.diverged_prob and g.n >= 4 * nreads * (patl + 8):
            b = synth.diverged_copy_reads(g, nreads, patl, seedl, seed=int(rng.integers(1 << 30)), q_max=int(rng.choice([40, 40, 63])))
            diverged = True
        else:
            b = synth.sample_reads(g, nreads, patl, float(rng.choice([0.0, 0.02, 0.05])), seed=int(rng.integers(1 << 30)), n_read_prob=0.001)
        kind = int(rng.choice([0, 0, 2, 3, 3]))
        lg = max(int(np.log2(max(g.n, 2))), 2)
        pb = 0
        if kind == 3:
            # rows: 32-bit signatures: the row number is all signature bits but 1..4; wider ones: at most seedl - 32 bits
            pb = min(seedl - 32, max(2, lg - int(rng.integers(1, 5)))) if seedl > 32 else min(30, max(1, seedl - int(rng.integers(1, 5))))
        if args.only and it not in args.only:
            rng.random()  # (the mode draw below)
            continue
        desc = dict(it=it, seedl=seedl, patl=patl, k=k, seedk=seedk, scores=scores, n=g.n, reads=b.n_reads, ragged=bool(ragged), kind=kind, pb=pb, repeats=repeats, diverged=diverged)
        try:
            opts = RealOptions(seedl=seedl, seedkmax=seedk, totalkmax=min(k, 15), scores=bool(scores), filter_level=2).normalise()
            p = ora.make_params(seedl=seedl, seedkmax=seedk, totalkmax=min(k, 15), scores=scores)
            og = ora.Genome(g.sym, g.frag_start)
            ix = ora.Index(og, seedl)
            if rng.random() < 0.7:
                oinfo, oscore, octr = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets)
                m = UniqueMatcher(opts, prefix_bits=pb, table_kind=kind)
                m.set_text_symbols(0, g.sym, g.frag_start)
                m.build_index_block()
                variant = "host"
                if ragged:
                    info, score = m.match_unique(b.bases, b.qual, b.offsets)
                elif rng.random() < 0.4:
                    # the same batch resident on the device: arrays at any byte address, bases packed where the batch allows
                    # it (reads that hold an N are flagged), records written from scratch
                    import torch
                    variant = "device"
                    nr = b.n_reads
                    sh_b, sh_q = int(rng.integers(0, 16)), int(rng.integers(0, 16))
                    pack = (nr * patl) % 4 == 0 and rng.random() < 0.6
                    big_q = torch.zeros(nr * patl + 64, dtype=torch.uint8, device="cuda")
                    dq = big_q[sh_q:sh_q + nr * patl]; dq.copy_(torch.from_numpy(b.qual))
                    nfl = None
                    if pack:
                        variant = "device, packed"
                        q4 = np.minimum(b.bases, 3).reshape(-1, 4)
                        pk = ((q4[:, 0] << 6) | (q4[:, 1] << 4) | (q4[:, 2] << 2) | q4[:, 3]).astype(np.uint8)
                        flags = np.zeros((nr + 7) // 8, dtype=np.uint8)
                        badr = np.nonzero((b.bases.reshape(nr, patl) > 3).any(axis=1))[0]
                        np.bitwise_or.at(flags, badr // 8, (1 << (badr % 8)).astype(np.uint8))
                        big_b = torch.zeros(pk.shape[0] + 64, dtype=torch.uint8, device="cuda")
                        db = big_b[sh_b:sh_b + pk.shape[0]]; db.copy_(torch.from_numpy(pk))
                        nfl = torch.from_numpy(flags).cuda()
                    else:
                        big_b = torch.zeros(nr * patl + 64, dtype=torch.uint8, device="cuda")
                        db = big_b[sh_b:sh_b + nr * patl]; db.copy_(torch.from_numpy(b.bases))
                    di = torch.full((nr,), 0x7123456789abcdef, dtype=torch.int64, device="cuda")
                    ds = torch.full((nr,), 1e30, dtype=torch.float32, device="cuda")
                    m.match_unique(db, dq, patl=patl, info=di, score=ds, n_reads=nr, packed=pack, nflags=nfl, fresh=True)
                    info, score = di.cpu().numpy().view(np.uint64), ds.cpu().numpy()
                else:
                    info, score = m.match_unique(b.bases, b.qual, patl=patl)
                desc["variant"] = variant
                c = m.counters()
                ok = np.array_equal(info, oinfo) and (not scores or np.array_equal(score.view(np.uint32), oscore.view(np.uint32)))
                ok = ok and all(c[kk] == octr[kk] for kk in ("reads", "lookups", "candidates", "seedpass", "hits"))
                if args.only and not ok:
                    d = np.nonzero(info != oinfo)[0]
                    print("  records differ at", d[:10], "of", info.shape[0], "; counters", {kk: (c[kk], octr[kk]) for kk in octr if kk in c})
                    for i in d[:5]:
                        print("   read %d gpu %s score %r | oracle %s score %r" % (i, ora.unpack_record(info[i:i + 1]), score[i], ora.unpack_record(oinfo[i:i + 1]), oscore[i]))
                    ds = np.nonzero(score.view(np.uint32) != oscore.view(np.uint32))[0] if scores else []
                    print("  scores differ at", ds[:10])
                desc["mode"] = "unique"; desc["handed_over"] = c["handed_over"]
            else:
                oh, ooff, octr = ora.match_all(og, ix, p, b.bases, b.qual, b.offsets)
                m = AllMatcher(opts, prefix_bits=pb, table_kind=kind)
                m.set_text_symbols(0, g.sym, g.frag_start)
                m.build_index_block()
                hits, hoff = (m.match_all(b.bases, b.qual, b.offsets) if ragged else m.match_all(b.bases, b.qual, patl=patl))
                ok = (np.array_equal(hoff, ooff) and np.array_equal(hits["pos"], oh["pos"]) and np.array_equal(hits["k"], oh["k"]) and
                      np.array_equal(hits["inverted"], oh["inverted"]) and np.array_equal(hits["score"].view(np.uint32), oh["score"].view(np.uint32)))
                desc["mode"] = "all"; desc["hits"] = int(hits.shape[0])
            m.close()
        except Exception as e:      # a configuration the ABI refuses is reported, not fatal
            print("EXC", desc, repr(e)[:200], flush=True)
            continue
        if not ok:
            bad += 1
            print("DIFF", desc, flush=True)
        elif it % 25 == 0:
            print("ok  ", desc, "%.0f s" % (time.time() - t0), flush=True)
    print("fuzz: %d configurations, %d differ" % (it, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
