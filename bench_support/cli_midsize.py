#!/usr/bin/env python3
"""End-to-end run of the `real` command line at a size where the host side matters:
FASTA genome + FASTQ reads on disk -> 11-column TSV on disk, with the wall time of every stage
(the "timing:" line the driver prints) and the whole TSV checked against the oracle's records.

    python bench_support/cli_midsize.py [--reads 10000000] [--genome-mbp 100] [--out gpurun_out/cli_midsize.json]

The oracle (tests/oracle_lib.py) is the checker here, as in tests/: it matches the same reads against the same
genome (its index = the six sorted lists exported from the device) and every column of every output line is
compared with what printMatchUnlocked (matchUniqueImplementation.cpp:252-321) would print for its records.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REAL = os.path.join(ROOT, "real_amd", "host", "real")


def write_genome(path, sym, name=" random", cols=60):
    n = sym.shape[0]
    rows = n // cols
    body = np.empty((rows, cols + 1), dtype=np.uint8)
    body[:, :cols] = np.frombuffer(b"ACGT", dtype=np.uint8)[sym[:rows * cols].reshape(rows, cols)]
    body[:, cols] = 10
    with open(path, "wb") as f:
        f.write((">" + name + "\n").encode())
        f.write(body.tobytes())
        if n > rows * cols:
            f.write(np.frombuffer(b"ACGT", dtype=np.uint8)[sym[rows * cols:]].tobytes() + b"\n")


def make_reads(sym, n_reads, patl, errprob, seed):
    """genpat's distribution (genpat.cpp:96-157), vectorised: sorted uniform starts, strand flip p=0.5, substitutions"""
    rng = np.random.default_rng(seed)
    pos = np.sort(rng.integers(0, sym.shape[0] - patl + 1, size=n_reads))
    b = np.empty((n_reads, patl), dtype=np.uint8)
    q = np.empty((n_reads, patl), dtype=np.uint8)
    for lo in range(0, n_reads, 1_000_000):                       # (in slices: the index arrays are 8 bytes per base)
        hi = min(n_reads, lo + 1_000_000)
        x = sym[pos[lo:hi, None] + np.arange(patl)[None, :]]
        inv = rng.integers(0, 2, size=hi - lo).astype(bool)
        x[inv] = (3 - x[inv])[:, ::-1]
        mut = rng.random((hi - lo, patl)) < errprob
        b[lo:hi] = np.where(mut, (x + rng.integers(1, 4, size=(hi - lo, patl), dtype=np.uint8)) & 3, x)
        q[lo:hi] = np.where(mut, 9, 35)
    return b, q


def write_fastq(path, b, q, step=2_000_000):
    n, patl = b.shape
    w = 1 + 10 + 1 + patl + 3 + patl + 1
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for lo in range(0, n, step):                                  # (in slices: a record is 216 bytes)
            hi = min(n, lo + step)
            rec = np.empty((hi - lo, w), dtype=np.uint8)
            rec[:, 0] = ord("@")
            i = np.arange(lo, hi, dtype=np.int64)
            for d in range(10):
                rec[:, 1 + d] = 48 + (i // 10 ** (9 - d)) % 10
            rec[:, 11] = 10
            rec[:, 12:12 + patl] = lut[b[lo:hi]]
            rec[:, 12 + patl] = 10; rec[:, 13 + patl] = ord("+"); rec[:, 14 + patl] = 10
            rec[:, 15 + patl:15 + 2 * patl] = q[lo:hi] + 33
            rec[:, 15 + 2 * patl] = 10
            f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--genome-mbp", type=float, default=100.0)
    ap.add_argument("--patl", type=int, default=100)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cli_midsize.json"))
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    import oracle_lib as ora
    from real_amd.matcher import RealOptions, UniqueMatcher

    d = tempfile.mkdtemp(prefix="real_mid_", dir=os.environ.get("TMPDIR", "/tmp"))
    G = int(args.genome_mbp * 1e6)
    t0 = time.time()
    sym = np.random.default_rng(5).integers(0, 4, size=G, dtype=np.uint8)
    b, q = make_reads(sym, args.reads, args.patl, 0.02, 6)
    fa, fq, out = os.path.join(d, "g.fa"), os.path.join(d, "r.fq"), os.path.join(d, "out.tsv")
    write_genome(fa, sym)
    write_fastq(fq, b, q)
    print("inputs: %.2f GB of FASTQ, %.2f GB of FASTA, made in %.0f s" % (os.path.getsize(fq) / 1e9, os.path.getsize(fa) / 1e9, time.time() - t0), flush=True)

    env = dict(os.environ, OMP_NUM_THREADS=str(args.threads))
    runs = []
    for rep in range(2):                                            # (second run: files in the page cache)
        t = time.time()
        r = subprocess.run([REAL, "-t", fa, "-p", fq, "-o", out, "-e", "3", "-s", "2", "-l", "32", "-q", "1"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, env=env)
        wall = time.time() - t
        err = r.stderr.decode()
        assert r.returncode == 0, err[-2000:]
        tl = [l for l in err.split("\n") if l.startswith("timing: ")][-1]
        tm = {k: float(v) for k, v in (kv.split("=") for kv in tl[len("timing: "):].split())}
        tm["process_wall_s"] = wall
        runs.append(tm)
        print("run %d: %.1f s wall; %s" % (rep, wall, tl), flush=True)
    res = {"reads": args.reads, "read_len": args.patl, "genome_bp": G, "fastq_bytes": os.path.getsize(fq), "tsv_bytes": os.path.getsize(out),
           "host_threads": args.threads, "runs": runs,
           "reads_per_s_end_to_end": args.reads / runs[-1]["total_s"],
           "command": "real -t g.fa -p r.fq -o out.tsv -e 3 -s 2 -l 32 -q 1"}

    if not args.no_check:
        t0 = time.time()
        frag = np.array([0, G], dtype=np.uint64)
        opts = RealOptions(seedl=32, seedkmax=2, totalkmax=3, scores=True).normalise()
        m = UniqueMatcher(opts)
        m.set_text_symbols(0, sym, frag)
        m.build_index_block()
        og = ora.Genome(sym, frag)
        signs, poss = [], []
        for k in range(6):
            sg, ps = m.index_export(k)
            signs.append(sg); poss.append(ps)
            print("  list %d exported" % k, flush=True)
        ix = ora.CompactIndex(og, 32, list(signs), list(poss))
        m.close()
        p = ora.make_params(seedl=32, seedkmax=2, totalkmax=3, scores=True, threads=args.threads)
        off = np.arange(args.reads + 1, dtype=np.uint64) * np.uint64(args.patl)
        info, score, _ = ora.match_unique(og, ix, p, b.reshape(-1), q.reshape(-1), off)
        st, fr, er, fi, po = ora.unpack_record(info)
        sel = np.nonzero((st == 1) | (st == 2))[0]
        print("oracle: %d of %d reads uniquely matched, %.0f s" % (sel.shape[0], args.reads, time.time() - t0), flush=True)
        import pandas as pd
        names = ("lines", "id", "sequence", "score", "constant_columns", "strand", "fragment", "position", "errors")
        checks = {k: True for k in names}
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        at = 0
        for df in pd.read_csv(out, sep="\t", header=None, dtype=str, keep_default_na=False, quoting=3, engine="c", chunksize=4_000_000):
            m_ = df.shape[0]
            s_ = sel[at:at + m_]
            if s_.shape[0] != m_ or df.shape[1] != 11:
                checks["lines"] = False
                break
            checks["id"] &= bool(np.array_equal(df[0].astype(np.int64).values, s_))
            inv = st[s_] == 2
            eb = b[s_]
            eb[inv] = (3 - eb[inv])[:, ::-1]
            got_seq = np.array(df[1].values, dtype="S%d" % args.patl).view(np.uint8).reshape(-1, args.patl)
            checks["sequence"] &= bool(np.array_equal(got_seq, lut[eb]))
            checks["score"] &= bool(np.array_equal(np.array(df[2].values, dtype=str), np.char.mod("%g", score[s_].astype(np.float64))))
            checks["constant_columns"] &= bool((df[3] == "1").all() and (df[4] == "a").all() and (df[5] == str(args.patl)).all() and (df[9] == "").all())
            checks["strand"] &= bool(np.array_equal(df[6].values == "-", inv))
            checks["fragment"] &= bool((df[7] == " random").all())
            checks["position"] &= bool(np.array_equal(df[8].astype(np.int64).values, po[s_] + 1))
            checks["errors"] &= bool(np.array_equal(df[10].astype(np.int64).values, er[s_]))
            at += m_
            print("  compared %d lines" % at, flush=True)
        checks["lines"] &= at == sel.shape[0]
        res["tsv_equals_oracle"] = bool(all(checks.values()))
        res["checks"] = checks
        print("TSV vs oracle: %s (%.0f s)" % (checks, time.time() - t0), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))
    for f in (fa, fq, out):
        os.remove(f)
    os.rmdir(d)
    assert args.no_check or res["tsv_equals_oracle"]


if __name__ == "__main__":
    main()
