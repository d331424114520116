#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ (run in the build
container only, where /root/reference exists).

For every case below the seeded synthetic inputs are written to a scratch
directory, the reference's own header-only hot path (oracle/_ref/ref_harness,
built by `make -C oracle ref` from the sources where they lie) is run on them,
and inputs + reference outputs are stored as one small .npz:

  * LL          the reference's 4x4x64 log-odds table (Scoring.cpp), bit patterns
  * records     UniqueMatchInfo bit packing known answers
  * sigs        per read m[4], im[4], straight[6], reverse[6], fullrestwords, fracrestsyms
  * events      per genome block, the ordered updater::update stream of the 12
                ::match calls per read: (read, call, inverted, pos, totalk, frag, score bits)
  * digests     sha256 of every sorted list's sign / ptr / pos arrays and of the
                sparse 22-bit lookup tables, per block
  * text        packed 2-bit text words (AutoTextArray)

The fixtures are data only.  tests/test_oracle_golden.py checks
oracle/real_oracle.c against them without needing /root/reference.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from real_amd import synth  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

CASES = {
    # name: genome kwargs, list of (n_reads, patl, errprob, seed), params
    "c1_l32_k0_36bp": dict(genome=dict(n=40000, seed=1), reads=[(300, 36, 0.0, 2)],
                           seedl=32, seedkmax=0, totalkmax=0, scores=0, n_list=0),
    "c2_l32_k3_100bp_q": dict(genome=dict(n=60000, seed=3, n_frag=3, n_runs=4, repeats=8),
                              reads=[(300, 100, 0.02, 4)], seedl=32, seedkmax=2, totalkmax=3, scores=1, n_list=0),
    "c2_noscores": dict(genome=dict(n=60000, seed=3, n_frag=3, n_runs=4, repeats=8),
                        reads=[(300, 100, 0.02, 4)], seedl=32, seedkmax=2, totalkmax=3, scores=0, n_list=0),
    "c2_blocks": dict(genome=dict(n=60000, seed=3, n_frag=3, n_runs=4, repeats=8),
                      reads=[(300, 100, 0.02, 4)], seedl=32, seedkmax=2, totalkmax=3, scores=1, n_list=25000),
    "c3_l32_k2_all": dict(genome=dict(n=50000, seed=5, n_frag=2, repeats=12, repeat_len=200),
                          reads=[(300, 100, 0.01, 6)], seedl=32, seedkmax=2, totalkmax=2, scores=1, n_list=0),
    "c5_l64_k5_150bp_q": dict(genome=dict(n=60000, seed=7, n_frag=2, n_runs=2, repeats=8, repeat_len=400),
                              reads=[(200, 150, 0.02, 12)], seedl=64, seedkmax=2, totalkmax=5, scores=1, n_list=0),
    "l12_dense": dict(genome=dict(n=3000, seed=9, n_frag=2, n_runs=2), reads=[(150, 40, 0.03, 10)],
                      seedl=12, seedkmax=2, totalkmax=4, scores=1, n_list=0),
    "l48_ragged": dict(genome=dict(n=30000, seed=11, n_frag=4, n_runs=3, repeats=5),
                       reads=[(60, 48, 0.0, 13), (60, 75, 0.02, 14), (60, 131, 0.03, 15), (20, 40, 0.0, 16)],
                       seedl=48, seedkmax=1, totalkmax=4, scores=1, n_list=0, n_read_prob=0.002),
    # other seed lengths (RealOptions.cpp:439-443 makes every seed length a multiple of four, so the four segments are
    # always equally long): l = 20 and 28 with 32-bit signatures, l = 36 -- the shortest seed with 64-bit ones; reads
    # exactly as long as the seed; seedkmax 1
    "l20_k4": dict(genome=dict(n=20000, seed=21, n_frag=2, n_runs=2, repeats=6), reads=[(200, 60, 0.03, 22)],
                   seedl=20, seedkmax=2, totalkmax=4, scores=1, n_list=0),
    "l28_s1_noscores": dict(genome=dict(n=50000, seed=23, n_frag=3, n_runs=3, repeats=8), reads=[(200, 100, 0.02, 24), (50, 28, 0.0, 25)],
                            seedl=28, seedkmax=1, totalkmax=3, scores=0, n_list=0),
    "l36_wide": dict(genome=dict(n=40000, seed=27, n_frag=2, n_runs=2, repeats=6, repeat_len=300), reads=[(200, 125, 0.02, 28)],
                     seedl=36, seedkmax=2, totalkmax=5, scores=1, n_list=0),
    # 250 bp reads (eight words per oriented read), seed 32, many mismatches allowed
    "l32_250bp_k8": dict(genome=dict(n=60000, seed=29, n_frag=2, n_runs=3, repeats=6, repeat_len=500), reads=[(150, 250, 0.02, 30)],
                         seedl=32, seedkmax=2, totalkmax=8, scores=1, n_list=0),
    # scores on, near-copies of a read that differ from it in different seed segments and score an epsilon-step apart: the
    # order of the update() calls decides the record (X, A | X, C | X, A, C ... ends NonUnique; grouped by window it
    # would end Straight).  The compiled reference pins the event stream: list by list, ascending position inside a list.
    "near_copies_xac": dict(make="near_copy", n=60000, seedl=32, seedkmax=2, totalkmax=3, scores=1, n_list=0),
    "diverged_copies": dict(make="diverged", genome=dict(n=120000, seed=41, n_frag=2, n_runs=2), n_reads=200, patl=100, seed=42,
                            seedl=32, seedkmax=2, totalkmax=3, scores=1, n_list=0),
    "diverged_copies_l64": dict(make="diverged", genome=dict(n=160000, seed=43, n_frag=2), n_reads=150, patl=150, seed=44,
                                seedl=64, seedkmax=2, totalkmax=5, scores=1, n_list=0),
}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_inputs(case):
    if case.get("make") == "near_copy":
        g, b, _ = synth.near_copy_case(n=case["n"], seedl=case["seedl"])
        return g, b
    if case.get("make") == "diverged":
        g = synth.random_genome(**case["genome"])
        b = synth.diverged_copy_reads(g, case["n_reads"], case["patl"], case["seedl"], seed=case["seed"])     # (plants the copies into g)
        return g, b
    g = synth.random_genome(**case["genome"])
    batches = [synth.sample_reads(g, n, patl, err, seed, n_read_prob=case.get("n_read_prob", 0.0))
               for (n, patl, err, seed) in case["reads"]]
    b = synth.concat_batches(batches)
    return g, b


def run_harness(g, b, case, d):
    g.sym.tofile(os.path.join(d, "genome.u8"))
    g.frag_start.astype(np.uint64).tofile(os.path.join(d, "frag.u64"))
    b.offsets.astype(np.uint64).tofile(os.path.join(d, "reads_off.u64"))
    b.bases.tofile(os.path.join(d, "reads_bases.u8"))
    b.qual.tofile(os.path.join(d, "reads_qual.u8"))
    subprocess.check_call([HARNESS, d, str(case["seedl"]), str(case["seedkmax"]), str(case["totalkmax"]),
                           str(case["scores"]), str(case["n_list"])], stderr=subprocess.DEVNULL)
    out = {}
    out["LL"] = np.fromfile(os.path.join(d, "ref_LL.f64"), dtype=np.float64)
    out["text"] = np.fromfile(os.path.join(d, "ref_text.u64"), dtype=np.uint64)
    sym = np.fromfile(os.path.join(d, "ref_sym.u8"), dtype=np.uint8)
    assert np.array_equal(sym, g.sym), "AutoTextArray[] disagrees with the input symbols"
    out["records"] = np.fromfile(os.path.join(d, "ref_records.u64"), dtype=np.uint64)
    sig_rows = [list(map(int, ln.split())) for ln in open(os.path.join(d, "ref_sigs.txt")) if ln.strip()]
    out["sigs"] = np.array(sig_rows, dtype=np.uint64).reshape(-1, 23)
    nblocks = int(open(os.path.join(d, "ref_meta.txt")).read().split()[1])
    out["nblocks"] = np.int64(nblocks)
    digests = []
    for blk in range(nblocks):
        ev = np.loadtxt(os.path.join(d, "ref_b%d_events.txt" % blk), dtype=np.uint64, ndmin=2)
        out["events_b%d" % blk] = ev.reshape(-1, 7)
        for k in range(6):
            pre = os.path.join(d, "ref_b%d_l%d_" % (blk, k))
            sg = np.fromfile(pre + "sign.u64", dtype=np.uint64)
            pt = np.fromfile(pre + "ptr.u32", dtype=np.uint32)
            ps = np.fromfile(pre + "pos.u32", dtype=np.uint32)
            lk = np.fromfile(pre + "lookup.u64", dtype=np.uint64)
            digests.append("%d %d %d %s %s %s %s" % (blk, k, sg.shape[0], sha(sg), sha(pt), sha(ps), sha(lk)))
    out["digests"] = np.array(digests)
    return out


def main():
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    only = set(sys.argv[1:])           # (names on the command line: only those cases are written)
    for name, case in CASES.items():
        if only and name not in only:
            continue
        g, b = make_inputs(case)
        with tempfile.TemporaryDirectory() as d:
            out = run_harness(g, b, case, d)
        nev = sum(out["events_b%d" % k].shape[0] for k in range(int(out["nblocks"])))
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            genome=g.sym, frag_start=g.frag_start, bases=b.bases, qual=b.qual, offsets=b.offsets,
            params=np.array([case["seedl"], case["seedkmax"], case["totalkmax"], case["scores"], case["n_list"]],
                            dtype=np.int64), **out)
        print("%-22s genome %6d reads %4d blocks %d events %5d" % (name, g.n, b.n_reads, int(out["nblocks"]), nev))


if __name__ == "__main__":
    main()
