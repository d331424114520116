"""The host-side readers (real_amd/host/ReadReader.cpp, GenomeText.cpp) against what the REFERENCE's own readers make
of the same texts: tests/golden/readers.npz holds tricky FASTQ / FASTA / genome inputs and the outputs of FastQReader,
FastAReader, Pattern::computeMapped and countLength / readFile compiled from the reference (oracle/ref_readers.cpp,
tests/golden/make_reader_golden.py).  Pins ids (a '\\r' in front of the newline belongs to the id), lowercase and IUPAC
letters -> 4, wrapped records, '@' at the start of a quality line, the quality-offset autodetection, the read count,
fragment names (everything behind the LAST '>' of a header line) and starts, dropped characters."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "real_amd", "host", "host_selftest")
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "readers.npz"))
CASES = sorted({k.split("/")[0] for k in Z.files})


@pytest.fixture(scope="module")
def selftest():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "real_amd", "host"), "host_selftest"], stdout=subprocess.DEVNULL)
    return SELFTEST


@pytest.mark.parametrize("case", [c for c in CASES if not c.startswith("g_")])
def test_read_readers_equal_the_reference(selftest, tmp_path, case):
    fn = tmp_path / "in.txt"
    fn.write_bytes(Z[case + "/input"].tobytes())
    cnt, det, n = [int(x) for x in Z[case + "/meta"]]
    fastq = case.startswith("fq_")
    subprocess.check_call([selftest, "reads", str(fn), "1" if fastq else "0", "0", str(tmp_path)])
    got_cnt, got_det = [int(x) for x in open(tmp_path / "meta.txt").read().split()]
    assert (got_cnt, got_det) == (cnt, det)
    assert (tmp_path / "ids.bin").read_bytes() == Z[case + "/ids"].tobytes()
    assert np.array_equal(np.fromfile(tmp_path / "off.u64", dtype=np.uint64), Z[case + "/off"])
    assert np.array_equal(np.fromfile(tmp_path / "bases.u8", dtype=np.uint8), Z[case + "/bases"])
    assert np.array_equal(np.fromfile(tmp_path / "qual.u8", dtype=np.uint8), Z[case + "/qual"])
    assert n == Z[case + "/off"].shape[0] - 1


@pytest.mark.parametrize("case", [c for c in CASES if c.startswith("g_")])
@pytest.mark.parametrize("threads", ["1", "3", "8"])
def test_genome_loader_equals_the_reference(selftest, tmp_path, case, threads):
    fn = tmp_path / "g.fa"
    fn.write_bytes(Z[case + "/input"].tobytes())
    subprocess.check_call([selftest, "genome", str(fn), str(tmp_path)], env=dict(os.environ, OMP_NUM_THREADS=threads))
    assert (tmp_path / "names.bin").read_bytes() == Z[case + "/names"].tobytes()
    assert np.array_equal(np.fromfile(tmp_path / "frag.u64", dtype=np.uint64), Z[case + "/frag"])
    assert np.array_equal(np.fromfile(tmp_path / "sym.u8", dtype=np.uint8), Z[case + "/sym"])
