"""End-to-end: the `real` command line (real_amd/host/real, C++ over the C ABI) against the
11-column output the reference's printMatchUnlocked would write for the oracle's records
(matchUniqueImplementation.cpp:252-321; matchAll: matchAllImplementation.cpp:481-518)."""
import os
import subprocess

import numpy as np
import pytest

from real_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "real_amd", "host", "real")


def fmt_float(x):
    return "%g" % float(np.float32(x))          # ostream << float: 6 significant digits


def seq(bases, inverted):
    b = synth.revcomp(bases) if inverted else bases
    return "".join("ACGTN"[c] for c in b)


def expected_unique(ora, g, b, info, score, scores):
    st, fr, er, fi, po = ora.unpack_record(info)
    lines = []
    for i in range(b.n_reads):
        if st[i] not in (1, 2):
            continue
        lo, hi = int(b.offsets[i]), int(b.offsets[i + 1])
        lines.append("\t".join([b.ids[i], seq(b.bases[lo:hi], st[i] == 2), fmt_float(score[i]) if scores else "", "1", "a",
                                str(hi - lo), "-" if st[i] == 2 else "+", g.frag_names[fr[i]],
                                str(po[i] - int(g.frag_start[fr[i]]) + 1), "", str(er[i])]))
    return lines


def write_inputs(tmp_path, g, b, fastq=True):
    fa = str(tmp_path / "genome.fa")
    synth.genome_to_fasta(g, fa)
    rd = str(tmp_path / ("reads.fq" if fastq else "reads.fa"))
    (synth.reads_to_fastq if fastq else synth.reads_to_fasta)(b, rd)
    return fa, rd


def oracle_unique(ora, g, b, seedl, seedk, totalk, scores, n_list=0, fasta=False):
    og = ora.Genome(g.sym, g.frag_start)
    p = ora.make_params(seedl=seedl, seedkmax=seedk, totalkmax=totalk, scores=scores)
    qual = np.full_like(b.qual, 30) if fasta else b.qual
    info = np.zeros(b.n_reads, np.uint64)
    score = np.full(b.n_reads, ora.NOSCORE_INIT, np.float32)
    first = 0
    while True:
        ix = ora.Index(og, seedl, first, n_list if n_list else 1 << 62)
        info, score, _ = ora.match_unique(og, ix, p, b.bases, qual, b.offsets, info=info, score=score)
        first += ix.n
        if not ix.have_next:
            break
    return info, score


@pytest.mark.parametrize("scores,extra,n_list,fastq", [
    (1, [], 0, True),
    (0, [], 0, True),
    (1, ["-index", "host", "-T", "3"], 0, True),
    (1, ["-block", "30000", "-batch", "500"], 30000, True),      # three index blocks, several read batches
    (1, [], 0, False),                                             # FASTA reads: constant quality 30
    (1, ["-gpuparse", "0"], 0, True),                              # read file parsed by the host reader
    (1, ["-table_kind", "3", "-l", "16", "-prefix_bits", "13"], 0, True),   # bucket rows (seed length 16 so that they are small)
    (1, ["-wrap"], 0, False),                                      # wrapped FASTA: the device parser refuses, host reader takes over
])
def test_real_cli_match_unique(ora, tmp_path, scores, extra, n_list, fastq):
    g = synth.random_genome(80_000, seed=41, n_frag=3, n_runs=6, repeats=15)
    b = synth.concat_batches([synth.sample_reads(g, 1500, 100, 0.02, seed=42, n_read_prob=0.0005),
                              synth.sample_reads(g, 300, 60, 0.02, seed=43)])
    fa, rd = write_inputs(tmp_path, g, b, fastq)
    if "-wrap" in extra:                                           # every sequence line cut in two (FastAReader.hpp:107-138 joins them)
        extra = []
        lines = open(rd).read().split("\n")
        open(rd, "w").write("\n".join(l if l.startswith(">") or not l else l[:37] + "\n" + l[37:] for l in lines))
    out = str(tmp_path / "out.tsv")
    cmd = [REAL, "-t", fa, "-p", rd, "-o", out, "-e", "3", "-s", "2", "-l", "32", "-q", str(scores)] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    seedl = int(extra[extra.index("-l") + 1]) if "-l" in extra else 32       # (a later -l overrides the one above)
    info, score = oracle_unique(ora, g, b, seedl, 2, 3, scores, n_list, fasta=not fastq)
    want = expected_unique(ora, g, b, info, score, scores)
    got = open(out).read().split("\n")[:-1]
    assert len(got) == len(want)
    assert got == want
    assert ("unique: %d" % len(want)) in r.stderr.decode()


def test_real_cli_match_all(ora, tmp_path):
    g = synth.random_genome(60_000, seed=51, n_frag=2, repeats=25, repeat_len=200)
    b = synth.sample_reads(g, 800, 100, 0.01, seed=52)
    fa, rd = write_inputs(tmp_path, g, b, True)
    out = str(tmp_path / "all.tsv")
    r = subprocess.run([REAL, "-t", fa, "-p", rd, "-o", out, "-u", "0", "-e", "2", "-s", "2", "-l", "32", "-q", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 32)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=2, scores=1)
    hits, hoff, _ = ora.match_all(og, ix, p, b.bases, b.qual, b.offsets)
    want = []
    for h in hits:
        i = int(h["read"])
        lo, hi = int(b.offsets[i]), int(b.offsets[i + 1])
        want.append("\t".join([b.ids[i], seq(b.bases[lo:hi], bool(h["inverted"])), fmt_float(h["score"]), "1", "a", str(hi - lo),
                               "-" if h["inverted"] else "+", g.frag_names[int(h["frag"])],
                               str(int(h["pos"]) - int(g.frag_start[int(h["frag"])]) + 1), "", str(int(h["k"]))]))
    got = open(out).read().split("\n")[:-1]
    assert got == want


def test_real_cli_errors_are_loud(tmp_path):
    r = subprocess.run([REAL, "-t", "/nonexistent.fa", "-p", "/nonexistent.fq", "-o", str(tmp_path / "o")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode != 0
