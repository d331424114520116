"""CPU-only tests: host logic, the oracle's own invariants, and that the C-ABI
library loads and exports every symbol include/real_hip.h declares (no compute
calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from real_amd import host_index, synth
from real_amd import lib as rlib
from real_amd.matcher import RealOptions, new_unique_info, unpack_info

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "real_hip.h")).read()
    declared = set(re.findall(r"\b(real_hip_[a-z_]+)\s*\(", hdr))
    assert declared == set(rlib.ABI_SYMBOLS), declared ^ set(rlib.ABI_SYMBOLS)
    L = rlib.load()
    for s in declared:
        assert hasattr(L, s), "libreal_hip.so does not export %s" % s
    assert L.real_hip_abi_version() == 2
    assert C.sizeof(rlib.RealHipParams) == 8 * 4 + 8 + 1024 * 8
    assert C.sizeof(rlib.RealHipBatch) == 64          # version 1 ended behind max_patl (48 bytes, still accepted)


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from real_amd.matcher import UniqueMatcher
    with pytest.raises(rlib.RealHipError):
        UniqueMatcher(RealOptions().normalise())


def test_product_package_never_touches_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "real_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in src and "real_oracle" not in src and "libreal_oracle" not in src, f


def test_scoring_table_known_answers():
    # printScores of the compiled reference, default flags
    LL = rlib.scoring_table()
    def ll(a, b, q): return LL[(a << 8) | (b << 6) | q]
    assert ll(0, 0, 0) == 0.0
    assert abs(ll(0, 0, 9) - 1.53673) < 5e-6
    assert abs(ll(0, 0, 30) - 1.7563) < 5e-5
    assert abs(ll(0, 0, 35) - 1.7575) < 5e-5
    assert abs(ll(0, 2, 30) - -7.03873) < 5e-6
    assert abs(ll(1, 1, 30) - 2.27091) < 5e-6
    assert abs(ll(2, 3, 9) - -6.82892) < 5e-6


def test_scoring_table_matches_oracle_bits(ora):
    for args in [(0.995, 0.41, 0.71, 0.0, 2.0), (0.98, 0.5, 1 / 3, 0.01, 1.0), (0.9, 0.35, 0.6, 0.05, 3.0)]:
        a = rlib.scoring_table(*args)
        b, _ = ora.scoring_table(*args)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_realoptions_parse_and_clamps():
    o = RealOptions.parse("-t g.fa -p r.fq -o out -e 30 -s 5 -l 70 -q 0 -u 0 -filter_level 3 --bogus x".split())
    assert (o.totalkmax, o.seedkmax, o.seedl, o.scores, o.match_unique) == (15, 2, 64, False, False)
    assert abs(o.filter_mult - 2 * 15 / 70.0) < 1e-15
    o = RealOptions.parse(["-l", "30"])
    assert o.seedl == 28
    d = RealOptions()
    assert (d.seedkmax, d.totalkmax, d.seedl, d.scores, d.filter_level) == (2, 5, 32, True, 2)      # RealOptions.hpp:27-36
    assert abs(RealOptions(totalkmax=3).getFilterValue(100) - 3 / 70 * 100) < 1e-12
    with pytest.raises(ValueError):
        RealOptions.parse(["-l", "3"])
    with pytest.raises(ValueError):
        RealOptions.parse(["-t"])


def test_record_known_answers(ora):
    # Straight, pos 207, file 0, frag 0, 0 errors
    assert ora.lib().ora_record_pack(1, 0, 0, 0, 207) == 0x20000000000000CF
    assert ora.lib().ora_record_pack(2, 0, 0, 0, 207) == 0x40000000000000CF
    assert ora.lib().ora_record_pack(4, 0, 0, 0, 207) == 0x80000000000000CF
    assert ora.lib().ora_record_pack(1, 0, 3, 0, 207) == 0x20000000000000CF + (3 << 41)
    st, fr, er, fi, po = unpack_info(np.array([0x20000000000000CF + (3 << 41)], dtype=np.uint64))
    assert (st[0], fr[0], er[0], fi[0], po[0]) == (1, 0, 3, 0, 207)
    info, score = new_unique_info(3, True)
    assert np.all(info == 0) and np.all(score == np.float32(-3.4028234663852886e38))


def test_diffcountpair_and_signature_known_answers(ora):
    L = ora.lib()
    assert L.ora_diffcountpair32(0b0001, 0b0010) == 1
    assert L.ora_diffcountpair32(0xFFFFFFFF, 0) == 16
    assert L.ora_diffcountpair64(0x123456789abcdef0, 0x123456789abcdef0) == 0
    rd = np.array([0, 1, 2, 3] * 8, dtype=np.uint8)          # ACGT x 8
    m = np.zeros(4, np.uint32); im = np.zeros(4, np.uint32); s = np.zeros(6, np.uint64); rs = np.zeros(6, np.uint64)
    assert L.ora_signature_mapped(32, rd.ctypes.data, m.ctypes.data)
    assert L.ora_reverse_mapped_signature(32, rd.ctypes.data, im.ctypes.data)
    L.ora_signatures(32, m.ctypes.data, s.ctypes.data)
    L.ora_signatures(32, im.ctypes.data, rs.ctypes.data)
    assert all(int(x) == 0x1B1B for x in m) and all(int(x) == 0x1B1B1B1B for x in s)
    assert np.array_equal(s, rs)                               # the prefix is its own reverse complement
    assert int(s[0]) >> 10 == 0x06C6C6


@pytest.mark.parametrize("seedl", [12, 32, 48, 64])
def test_host_index_equals_oracle_lists(ora, seedl):
    g = synth.random_genome(20_000, seed=31 + seedl, n_frag=3, n_runs=6, repeats=6)
    text, wild = host_index.pack_text(g.sym)
    og = ora.Genome(g.sym, g.frag_start)
    assert np.array_equal(text, og.text) and np.array_equal(wild, og.wild)
    first, blk = 0, 7000
    while True:
        sign, pos, n, nxt = host_index.build_lists(g.sym, seedl, first, blk)
        oix = ora.Index(og, seedl, first_window=first, max_entries=blk)
        assert n == oix.n and nxt == oix.have_next
        for k in range(6):
            assert np.array_equal(sign[k].astype(np.uint64), oix.sign(k))
            assert np.array_equal(pos[k], oix.pos(k))
        first += n
        if not nxt:
            break


def test_compact_index_is_the_same_matcher(ora):
    """the CPU-baseline form of the oracle (lists as {sign,pos} pairs, partner signature
    re-read from the text) gives the same records, scores and counters."""
    g = synth.random_genome(80_000, seed=5, n_frag=2, n_runs=4, repeats=20)
    b = synth.sample_reads(g, 1500, 100, 0.02, seed=6)
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 32)
    cix = ora.CompactIndex(og, 32, [ix.sign(k).astype(np.uint32) for k in range(6)], [ix.pos(k) for k in range(6)])
    for scores in (0, 1):
        p = ora.make_params(seedl=32, seedkmax=2, totalkmax=3, scores=scores)
        i1, s1, c1 = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets)
        i2, s2, c2 = ora.match_unique(og, cix, p, b.bases, b.qual, b.offsets)
        assert np.array_equal(i1, i2) and np.array_equal(s1.view(np.uint32), s2.view(np.uint32)) and c1 == c2


def test_oracle_finds_planted_reads(ora):
    """end-to-end sanity of the oracle against generator ground truth (genpat ids)."""
    g = synth.random_genome(300_000, seed=8)
    b = synth.sample_reads(g, 3000, 100, 0.0, seed=9)
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 32)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=3, scores=1)
    info, score, ctr = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets)
    st, fr, er, fi, po = ora.unpack_record(info)
    ok = (st == 1) | (st == 2)
    assert ok.mean() > 0.99
    assert np.array_equal(po[ok], b.true_pos[ok].astype(np.int64))
    assert np.array_equal(st[ok] == 2, b.true_inv[ok])
    assert np.all(er[ok] == 0)
    assert ctr["lookups"] == 12 * ctr["reads"]        # scores on: no uni0 early-out
