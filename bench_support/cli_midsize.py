#!/usr/bin/env python3
"""Mid-size end-to-end check of the `real` driver on the GPU box: a FASTQ file of several 256 MiB text chunks matched with
the device parser and with the host reader must give byte-identical output.   python bench_support/cli_midsize.py"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from real_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL = os.path.join(ROOT, "real_amd", "host", "real")
d = tempfile.mkdtemp(prefix="real_mid_")
g = synth.random_genome(20_000_000, seed=5, n_frag=3)
b = synth.sample_reads(g, 2_600_000, 100, 0.02, seed=6)
fa, fq = os.path.join(d, "g.fa"), os.path.join(d, "r.fq")
synth.genome_to_fasta(g, fa)
synth.reads_to_fastq(b, fq)
print("inputs: %.0f MB of FASTQ" % (os.path.getsize(fq) / 1e6), flush=True)
outs = []
for gp in ("1", "0"):
    out = os.path.join(d, "out%s.tsv" % gp)
    t = time.time()
    r = subprocess.run([REAL, "-t", fa, "-p", fq, "-o", out, "-e", "3", "-s", "2", "-l", "32", "-q", "1", "-gpuparse", gp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    print("gpuparse=%s: %.1f s, %d lines" % (gp, time.time() - t, sum(1 for _ in open(out))), flush=True)
    outs.append(open(out, "rb").read())
assert outs[0] == outs[1] and len(outs[0]) > 0
print("identical output")
