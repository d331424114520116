"""CPU tests of the C++ host side (real_amd/host): genome loader, FASTA/FASTQ readers,
host index builder and the RealOptions parser, through the host_selftest binary."""
import os
import subprocess

import numpy as np
import pytest

from real_amd import host_index, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "real_amd", "host", "host_selftest")


@pytest.fixture(scope="module")
def selftest():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "real_amd", "host"), "host_selftest"], stdout=subprocess.DEVNULL)
    return SELFTEST


def test_genome_loader(selftest, tmp_path):
    g = synth.random_genome(5000, seed=3, n_frag=3, n_runs=5)
    fa = tmp_path / "g.fa"
    synth.genome_to_fasta(g, str(fa))
    # quirk 4: anything but ACGTN outside headers is dropped, lowercase included
    txt = open(fa).read().replace("ACG", "ACacgtG", 3).replace("T", "T-", 2)
    open(fa, "w").write(txt)
    subprocess.check_call([selftest, "genome", str(fa), str(tmp_path)])
    sym = np.fromfile(tmp_path / "sym.u8", dtype=np.uint8)
    frag = np.fromfile(tmp_path / "frag.u64", dtype=np.uint64)
    assert np.array_equal(sym, g.sym) and np.array_equal(frag, g.frag_start)
    names = open(tmp_path / "names.txt").read().split("\n")[:-1]
    assert names == g.frag_names                      # header after '>' incl. the leading space
    text, wild = host_index.pack_text(g.sym)
    assert np.array_equal(np.fromfile(tmp_path / "text.u64", dtype=np.uint64), text)
    assert np.array_equal(np.fromfile(tmp_path / "wild.u64", dtype=np.uint64), wild)


@pytest.mark.parametrize("text,sym", [(">a\nAC\n", [0, 1]), (">\nG", [2]), (">x\n", []), ("", [])])
def test_genome_loader_file_shorter_than_the_thread_count(selftest, tmp_path, text, sym):
    """the parallel sweeps cut the file into one piece per thread at line starts; with fewer bytes than threads several
    cuts fall on offset 0, where there is no byte in front to look at (data[-1])"""
    fa = tmp_path / "tiny.fa"
    open(fa, "w").write(text)
    env = dict(os.environ, OMP_NUM_THREADS="64")
    r = subprocess.run([selftest, "genome", str(fa), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if not sym and r.returncode != 0:
        return                                        # an empty genome may be refused, loudly; it must not crash
    assert r.returncode == 0, r.stderr.decode()
    assert np.fromfile(tmp_path / "sym.u8", dtype=np.uint8).tolist() == sym


def test_fastq_and_fasta_readers(selftest, tmp_path):
    g = synth.random_genome(20000, seed=4)
    b = synth.concat_batches([synth.sample_reads(g, 20, 36, 0.05, seed=5, n_read_prob=0.02),
                              synth.sample_reads(g, 13, 100, 0.05, seed=6)])
    fq = tmp_path / "r.fq"
    synth.reads_to_fastq(b, str(fq), offset=33)
    subprocess.check_call([selftest, "reads", str(fq), "1", "0", str(tmp_path)])
    assert np.array_equal(np.fromfile(tmp_path / "bases.u8", dtype=np.uint8), b.bases)
    assert np.array_equal(np.fromfile(tmp_path / "qual.u8", dtype=np.uint8), b.qual)   # '*'=42 <= 54 -> offset 33 detected
    assert np.array_equal(np.fromfile(tmp_path / "off.u64", dtype=np.uint64), b.offsets)
    assert open(tmp_path / "ids.txt").read().split("\n")[:-1] == b.ids
    assert open(tmp_path / "meta.txt").read().split() == [str(b.n_reads), "33"]
    # Illumina-1.3 style qualities -> 64
    synth.reads_to_fastq(b, str(fq), offset=64)
    subprocess.check_call([selftest, "reads", str(fq), "1", "0", str(tmp_path)])
    assert np.array_equal(np.fromfile(tmp_path / "qual.u8", dtype=np.uint8), b.qual)
    assert open(tmp_path / "meta.txt").read().split()[1] == "64"
    # FASTA: multi-line sequences, lowercase -> 4, constant quality 30
    fa = tmp_path / "r.fa"
    with open(fa, "w") as f:
        for i in range(b.n_reads):
            s = "".join("ACGTN"[c] for c in b.bases[int(b.offsets[i]):int(b.offsets[i + 1])])
            f.write(">" + b.ids[i] + "\n" + s[:10] + "\n" + s[10:] + "\n")
        f.write(">lower\nacgtACGT\n")
    subprocess.check_call([selftest, "reads", str(fa), "0", "0", str(tmp_path)])
    bases = np.fromfile(tmp_path / "bases.u8", dtype=np.uint8)
    assert np.array_equal(bases[:-8], b.bases) and list(bases[-8:]) == [4, 4, 4, 4, 0, 1, 2, 3]
    assert np.all(np.fromfile(tmp_path / "qual.u8", dtype=np.uint8) == 30)
    assert open(tmp_path / "meta.txt").read().split()[0] == str(b.n_reads + 1)


@pytest.mark.parametrize("seedl,threads", [(32, 1), (32, 3), (64, 2), (12, 2)])
def test_host_index_builder(selftest, tmp_path, seedl, threads):
    g = synth.random_genome(30000, seed=7 + seedl, n_frag=2, n_runs=8, repeats=6)
    fa = tmp_path / "g.fa"
    synth.genome_to_fasta(g, str(fa))
    first, mx = 5000, 12000
    subprocess.check_call([selftest, "index", str(fa), str(seedl), str(first), str(mx), str(threads), str(tmp_path)])
    sign, pos, n, nxt = host_index.build_lists(g.sym, seedl, first, mx)
    meta = open(tmp_path / "meta.txt").read().split()
    assert int(meta[0]) == n and bool(int(meta[1])) == nxt and int(meta[2]) == (4 if seedl <= 32 else 8)
    for k in range(6):
        sg = np.fromfile(tmp_path / ("l%d_sign.bin" % k), dtype=np.uint32 if seedl <= 32 else np.uint64)
        ps = np.fromfile(tmp_path / ("l%d_pos.u32" % k), dtype=np.uint32)
        assert np.array_equal(sg, sign[k]) and np.array_equal(ps, pos[k])


def test_realoptions_cpp(selftest, tmp_path):
    fq = tmp_path / "r.fq"
    open(fq, "w").write("@a\nACGT\n+\nIIII\n")
    out = subprocess.check_output([selftest, "options", "-t", "g.fa", "-p", str(fq), "-o", "out", "-e", "30", "-s", "5", "-l", "70",
                                   "-q", "0", "-u", "0", "-filter_level", "3", "--bogus", "-gpus", "2", "-index", "host"],
                                  stderr=subprocess.DEVNULL).decode().split()
    assert out[0] == "g.fa" and out[2] == "out"
    assert out[3:10] == ["2", "15", "64", "0", "0", "0", "3"]
    assert abs(float(out[10]) - 2 * 15 / 70.0) < 1e-6 and out[11:] == ["1", "2", "1"]
    # missing mandatory argument / missing value: error exit, like the reference's exceptions
    assert subprocess.call([selftest, "options", "-t", "g.fa"], stderr=subprocess.DEVNULL) != 0
    assert subprocess.call([selftest, "options", "-t", "g.fa", "-p", str(fq), "-o"], stderr=subprocess.DEVNULL) != 0


def test_realoptions_patterns_from_stdin(selftest, tmp_path):
    """-p - (RealOptions.cpp:418-426, real.cpp:240-257): the patterns come from standard input; they are spooled into a
    temporary file (the driver reads them once per genome block and once more for the output, as the reference does through
    its rewritten pattern file), the format is taken from the first character, and a FASTQ without -Q is taken to be
    Illumina GA (offset 64) with the reference's warning.  The spool file is gone when the options are."""
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run([selftest, "options", "-t", "g.fa", "-p", "-", "-o", "out"], input=b"@a\nACGT\n+\nhhhh\n", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    out = r.stdout.decode().split()
    assert out[1].startswith(str(tmp_path)) and "real_stdin_" in out[1] and out[8] == "64" and out[11] == "1"
    assert b"Assuming input" in r.stderr and not os.path.exists(out[1])
    r = subprocess.run([selftest, "options", "-t", "g.fa", "-p", "-", "-o", "out", "-Q", "33"], input=b">a\nACGT\n", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = r.stdout.decode().split()
    assert r.returncode == 0 and out[8] == "33" and out[11] == "0" and b"Assuming input" not in r.stderr
    r = subprocess.run([selftest, "options", "-t", "g.fa", "-p", "-", "-o", "out"], input=b"", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and not list(tmp_path.glob("real_stdin_*"))       # empty input: the reference's error, and no file left behind


def test_fast_score_formatter_equals_printf(selftest):
    """the score column: fastformat::fmt_g6 (integer arithmetic) against printf's %g -- which is what the reference's
    operator<<(float) prints -- on random bit patterns, score-like values, integers and both sides of every power of ten"""
    r = subprocess.run([selftest, "fmtcheck", "400000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stderr[-2000:]
