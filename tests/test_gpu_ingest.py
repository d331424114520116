"""Read ingestion on the device: real_hip_parse_reads against the generator's own arrays and the
host reader's semantics (FastQReader.hpp:130-180, FastAReader.hpp:107-138, Pattern.hpp:105-128)."""
import numpy as np
import pytest

from real_amd import synth
from real_amd.lib import RealHipError
from real_amd.matcher import RealOptions, UniqueMatcher

pytestmark = pytest.mark.gpu

BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _opts(scores=1):
    return RealOptions(seedl=32, seedkmax=2, totalkmax=3, scores=scores)


def _fastq(b, qoff=33, crlf=False, trailing_newline=True, lower_at=None):
    nl = b"\r\n" if crlf else b"\n"
    parts = []
    for i in range(len(b.offsets) - 1):
        lo, hi = int(b.offsets[i]), int(b.offsets[i + 1])
        seq = BASES[b.bases[lo:hi]].tobytes()
        if lower_at is not None and i == lower_at:
            seq = seq.lower()                               # lowercase maps to 4 like N (acgtnMap.hpp:39-50)
        parts += [b"@" + b.ids[i].encode(), seq, b"+", (b.qual[lo:hi] + qoff).astype(np.uint8).tobytes()]
    text = nl.join(parts) + (nl if trailing_newline else b"")
    return text


class _Batch:
    pass


def _ragged(b, patl, rng, lo=40):
    """truncate the reads of a uniform batch to random lengths lo..patl"""
    n = len(b.offsets) - 1
    lens = rng.integers(lo, patl + 1, size=n)
    keep = (np.arange(patl)[None, :] < lens[:, None]).reshape(-1)
    r = _Batch()
    r.bases, r.qual, r.ids = b.bases[keep], b.qual[keep], b.ids
    r.offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return r


def _check(m, p, b, want_bases=None):
    n = len(b.offsets) - 1
    assert p.n_reads == n and p.n_symbols == int(b.offsets[-1])
    off = m.download(p.offsets, n + 1, np.uint64)
    assert np.array_equal(off, b.offsets.astype(np.uint64))
    bases = m.download(p.bases, int(p.n_symbols), np.uint8)
    assert np.array_equal(bases, b.bases if want_bases is None else want_bases)
    assert p.max_patl == int(np.diff(b.offsets).max())
    return off


@pytest.mark.parametrize("crlf,trail", [(False, True), (False, False), (True, True)])
def test_parse_fastq_equals_generator(crlf, trail):
    g = synth.random_genome(50_000, seed=3, n_runs=4)
    b = _ragged(synth.sample_reads(g, 3000, 100, 0.02, seed=4, n_read_prob=0.01), 100, np.random.default_rng(11))   # ragged, some N
    text = _fastq(b, crlf=crlf, trailing_newline=trail)
    m = UniqueMatcher(_opts())
    p = m.parse_reads(text, fastq=True, quality_offset=33)
    _check(m, p, b)
    qual = m.download(p.qual, int(p.n_symbols), np.uint8)
    assert np.array_equal(qual, b.qual)
    ids0 = m.download(p.id_start, p.n_reads, np.uint32)
    idl = m.download(p.id_len, p.n_reads, np.uint32)
    for i in (0, 1, p.n_reads // 2, p.n_reads - 1):
        assert text[int(ids0[i]):int(ids0[i]) + int(idl[i])].decode() == b.ids[i]
    m.close()


def test_parse_text_end_at_every_alignment():
    # the last characters of the text are read with care (no load may run past its end): every alignment of the end
    g = synth.random_genome(20_000, seed=15)
    b0 = synth.sample_reads(g, 40, 100, 0.02, seed=16)
    m = UniqueMatcher(_opts())
    for cut in range(9):
        b = _Batch()
        n = len(b0.offsets) - 1
        keep = np.ones(int(b0.offsets[-1]), dtype=bool)
        keep[int(b0.offsets[-1]) - cut:] = False                 # the last read loses `cut` bases
        b.bases, b.qual, b.ids = b0.bases[keep], b0.qual[keep], b0.ids
        b.offsets = b0.offsets.copy().astype(np.uint64)
        b.offsets[-1] -= np.uint64(cut)
        for trail in (True, False):
            p = m.parse_reads(_fastq(b, trailing_newline=trail), fastq=True, quality_offset=33)
            _check(m, p, b)
            assert np.array_equal(m.download(p.qual, int(p.n_symbols), np.uint8), b.qual)
    m.close()


def test_parse_device_text_at_odd_address():
    import torch
    g = synth.random_genome(50_000, seed=13)
    b = synth.sample_reads(g, 2000, 100, 0.02, seed=14)
    text = _fastq(b)
    t = torch.frombuffer(bytearray(b"#" + text), dtype=torch.uint8).cuda()[1:]      # device pointer = base + 1
    m = UniqueMatcher(_opts())
    p = m.parse_reads(t, fastq=True, quality_offset=33)
    _check(m, p, b)
    assert np.array_equal(m.download(p.qual, int(p.n_symbols), np.uint8), b.qual)
    m.close()


def test_parse_fasta_and_lowercase():
    g = synth.random_genome(20_000, seed=5)
    b = synth.sample_reads(g, 500, 36, 0.0, seed=6)
    parts = []
    for i in range(500):
        lo, hi = int(b.offsets[i]), int(b.offsets[i + 1])
        seq = BASES[b.bases[lo:hi]].tobytes()
        parts += [b">" + b.ids[i].encode(), seq.lower() if i == 7 else seq]
    text = b"\n".join(parts) + b"\n"
    want = b.bases.copy()
    want[int(b.offsets[7]):int(b.offsets[8])] = 4
    m = UniqueMatcher(_opts(0))
    p = m.parse_reads(text, fastq=False)
    _check(m, p, b, want_bases=want)
    assert not p.qual                      # FASTA: no qualities (the matcher then uses 30, Pattern.hpp:42-45)
    m.close()


def test_parse_refuses_non_canonical_text():
    g = synth.random_genome(20_000, seed=7)
    b = synth.sample_reads(g, 50, 100, 0.02, seed=8)
    text = _fastq(b)
    m = UniqueMatcher(_opts())
    lines = text.split(b"\n")
    wrapped = b"\n".join(lines[:1] + [lines[1][:50], lines[1][50:]] + lines[2:])           # a wrapped sequence
    for bad in (wrapped, text[:-30], text.replace(b"\n+\n", b"\n-\n", 1), b"x" + text, lines[0] + b"\n" + lines[1][:10] + b" " + lines[1][11:] + b"\n" + b"\n".join(lines[2:])):
        with pytest.raises(RealHipError):
            m.parse_reads(bad, fastq=True)
    p = m.parse_reads(b"", fastq=True)
    assert p.n_reads == 0
    m.close()


def test_match_from_parsed_text_equals_match_from_arrays(ora):
    g = synth.random_genome(300_000, seed=9, n_frag=3, repeats=10)
    b = synth.sample_reads(g, 5000, 100, 0.02, seed=10, n_read_prob=0.001)
    m = UniqueMatcher(_opts())
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    info0, score0 = m.match_unique(b.bases, b.qual, b.offsets)
    p = m.parse_reads(_fastq(b), fastq=True, quality_offset=33)
    info1, score1 = m.match_unique_parsed(p)
    assert np.array_equal(info0, info1) and np.array_equal(score0.view(np.uint32), score1.view(np.uint32))
    m.close()


def _golden_reader_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "readers.npz"))
    return z, sorted({k.split("/")[0] for k in z.files if not k.startswith("g_")})


@pytest.mark.parametrize("case", _golden_reader_cases()[1])
def test_device_parser_equals_the_reference_readers_or_refuses(case):
    """tests/golden/readers.npz: what the REFERENCE's FastQReader / FastAReader (compiled, oracle/ref_readers.cpp) make of
    tricky texts.  Text in one-line-per-field form must come out of real_hip_parse_reads exactly like that -- mapped
    symbols, qualities minus the detected offset, lengths, ids (a '\\r' in front of the newline belongs to the id) --
    and everything else must be refused with REAL_HIP_E_UNSUPPORTED (the host reader, pinned by the same fixture in
    tests/test_reader_golden.py, takes it)."""
    from real_amd import lib as rlib
    z, _ = _golden_reader_cases()
    text = z[case + "/input"].tobytes()
    cnt, det, n = [int(x) for x in z[case + "/meta"]]
    fastq = case.startswith("fq_")
    canonical = case in ("fq_canonical_q33", "fq_q64", "fq_crlf", "fq_lowercase_iupac_n", "fq_at_in_quality", "fq_no_final_newline",
                         "fq_ragged_lengths", "fa_canonical", "fa_no_final_newline")   # (fa_crlf: its second record is wrapped)
    m = UniqueMatcher(_opts().normalise())
    if not canonical:
        with pytest.raises(RealHipError) as e:
            m.parse_reads(text, fastq=fastq, quality_offset=det or 33)
        assert e.value.status == rlib.REAL_HIP_E_UNSUPPORTED
        m.close()
        return
    p = m.parse_reads(text, fastq=fastq, quality_offset=det)
    assert p.n_reads == n == cnt
    assert np.array_equal(m.download(p.offsets, n + 1, np.uint64), z[case + "/off"])
    assert np.array_equal(m.download(p.bases, int(p.n_symbols), np.uint8), z[case + "/bases"])
    if fastq:
        assert np.array_equal(m.download(p.qual, int(p.n_symbols), np.uint8), z[case + "/qual"])
    ids0, idl = m.download(p.id_start, n, np.uint32), m.download(p.id_len, n, np.uint32)
    got = b""
    for s, l in zip(ids0.tolist(), idl.tolist()):
        if text[s + l:s + l + 1] == b"\r":               # (the rule of the output formatter, real_amd/host/real.cpp)
            l += 1
        got += text[s:s + l] + b"\0"
    assert got == z[case + "/ids"].tobytes()
    m.close()
