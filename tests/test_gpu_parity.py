"""Parity tests proper: the HIP path, called through the C ABI (libreal_hip.so),
against the oracle on the same seeded inputs.  Bit-exact: records (state,
fragment, errors, file, position), float score bits, matchAll hit lists, and
the logical work counters.
"""
import glob
import os

import numpy as np
import pytest

from real_amd import host_index, synth
from real_amd.matcher import AllMatcher, RealOptions, UniqueMatcher, new_unique_info, unpack_info

pytestmark = pytest.mark.gpu

GOLDEN = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if os.path.basename(f) != "readers.npz")   # (readers.npz: tests/test_reader_golden.py)


def _opts(seedl, seedkmax, totalkmax, scores, filter_level=2):
    return RealOptions(seedl=seedl, seedkmax=seedkmax, totalkmax=totalkmax, scores=bool(scores),
                       filter_level=filter_level).normalise()


def _oracle_unique(ora, g, sym, frag, seedl, n_list, p, bases, qual, offsets):
    og = ora.Genome(sym, frag)
    info = np.zeros(offsets.shape[0] - 1, dtype=np.uint64)
    score = np.full(offsets.shape[0] - 1, ora.NOSCORE_INIT, dtype=np.float32)
    first = 0
    tot = None
    while True:
        ix = ora.Index(og, seedl, first_window=first, max_entries=(n_list if n_list else 1 << 62))
        info, score, ctr = ora.match_unique(og, ix, p, bases, qual, offsets, info=info, score=score)
        tot = ctr if tot is None else {k: tot[k] + ctr[k] for k in tot}
        first += ix.n
        if not ix.have_next:
            break
    return info, score, tot


def _compare_unique(info, score, oinfo, oscore, scores):
    st, fr, er, fi, po = unpack_info(info)
    ost, ofr, oer, ofi, opo = unpack_info(oinfo)
    assert np.array_equal(st, ost), "match state differs"
    assert np.array_equal(er, oer), "mismatch count differs"
    # raw records are bit-identical too (the pos/frag bits of a NonUnique record are those of
    # the first hit seen at that level in canonical order, which the device fold preserves)
    assert np.array_equal(info, oinfo), "records differ"
    if scores:
        assert np.array_equal(np.asarray(score).view(np.uint32), np.asarray(oscore).view(np.uint32)), "score bits differ"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("builder", ["device", "host"])
def test_match_unique_golden_inputs(ora, path, builder):
    z = np.load(path)
    seedl, seedkmax, totalkmax, scores, n_list = [int(x) for x in z["params"]]
    sym, frag = z["genome"], z["frag_start"]
    bases, qual, offsets = z["bases"], z["qual"], z["offsets"]
    opts = _opts(seedl, seedkmax, totalkmax, scores)
    p = ora.make_params(seedl=seedl, seedkmax=seedkmax, totalkmax=totalkmax, scores=scores)
    oinfo, oscore, octr = _oracle_unique(ora, None, sym, frag, seedl, n_list, p, bases, qual, offsets)

    m = UniqueMatcher(opts)
    assert np.array_equal(m.LL.view(np.uint64), np.array(list(p.LL)).view(np.uint64)), "LL table differs from the oracle's"
    if builder == "device":
        m.set_text_symbols(0, sym, frag)
    else:
        text, wild = host_index.pack_text(sym)
        m.set_text(0, text, wild, sym.shape[0], frag)
    info, score = new_unique_info(offsets.shape[0] - 1, scores)
    first = 0
    m.counters(reset=True)
    while True:
        if builder == "device":
            n, nxt = m.build_index_block(first, n_list if n_list else 1 << 62)
        else:
            sign, pos, n, nxt = host_index.build_lists(sym, seedl, first, n_list if n_list else 1 << 62)
            m.set_index_block(sign, pos)
        m.match_unique(bases, qual, offsets, info=info, score=score)
        first += n
        if not nxt:
            break
    _compare_unique(info, score, oinfo, oscore, scores)
    c = m.counters()
    for k in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[k] == octr[k], "work counter %s: %d != oracle %d" % (k, c[k], octr[k])
    m.close()


def _hits_tuple(h):
    return (h["read"].astype(np.int64), h["pos"].astype(np.int64), h["frag"].astype(np.int64),
            h["k"].astype(np.int64), h["inverted"].astype(np.int64), h["score"].view(np.uint32).astype(np.int64))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_match_all_golden_inputs(ora, path):
    z = np.load(path)
    seedl, seedkmax, totalkmax, scores, n_list = [int(x) for x in z["params"]]
    if n_list:
        pytest.skip("matchAll is per block; covered by the one-block cases")
    sym, frag = z["genome"], z["frag_start"]
    bases, qual, offsets = z["bases"], z["qual"], z["offsets"]
    p = ora.make_params(seedl=seedl, seedkmax=seedkmax, totalkmax=totalkmax, scores=scores)
    og = ora.Genome(sym, frag)
    ix = ora.Index(og, seedl)
    ohits, ooff, octr = ora.match_all(og, ix, p, bases, qual, offsets)
    m = AllMatcher(_opts(seedl, seedkmax, totalkmax, scores))
    m.set_text_symbols(0, sym, frag)
    m.build_index_block()
    hits, hoff = m.match_all(bases, qual, offsets, cap=8)      # tiny cap: exercises the overflow retry
    m.counters(reset=True)
    hits2, hoff2 = m.match_all(bases, qual, offsets)
    assert np.array_equal(hits, hits2) and np.array_equal(hoff, hoff2)
    assert np.array_equal(hoff, ooff), "per-read hit counts differ"
    for a, b in zip(_hits_tuple(hits), _hits_tuple(ohits)):
        assert np.array_equal(a, b)
    c = m.counters()
    for k in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[k] == octr[k], "work counter %s: %d != oracle %d" % (k, c[k], octr[k])
    m.close()


@pytest.mark.parametrize("seedl,patl,k,scores,pb", [
    (32, 100, 3, 1, 0), (32, 100, 3, 0, 0), (32, 36, 0, 0, 0), (64, 150, 5, 1, 0),
    (32, 100, 3, 1, 4),          # 16 buckets: every bucket is large -> in-bucket binary search path
    (64, 150, 5, 1, 12),         # signature wider than prefix+32: fingerprint + text confirmation path
    (16, 50, 4, 1, 0), (20, 255, 15, 1, 0), (4, 20, 2, 0, 0), (32, 32, 2, 1, 0),
    # fine bucket tables (prefix = all signature bits but three / two / one): equal ranges straight from the table
    (16, 50, 4, 1, 13), (16, 50, 4, 1, 14), (16, 50, 4, 0, 13), (12, 40, 4, 1, 11), (8, 30, 3, 1, 5), (8, 30, 3, 1, 6),
    (32, 100, 3, 1, 29),
    # fingerprint bucket tables (64-bit signatures): mean bucket 6 (some overflow eight slots), 0.4 and 50 entries
    (64, 150, 5, 1, -15), (64, 150, 5, 0, -19), (48, 100, 4, 1, -12), (36, 80, 3, 1, -4),
    # bucket rows (pb + 100): 8 / 16 signature values per row, rows of 6..100 entries (most of them complex at pb 12 / 5)
    (16, 50, 4, 1, 113), (16, 50, 4, 0, 112), (16, 50, 4, 1, 115), (12, 40, 4, 1, 109), (8, 30, 3, 1, 105), (20, 255, 15, 1, 116),
    # bucket rows of wide signatures (key group = four key bits, 16-bit key fingerprints): rows of 6, 0.4, 50 entries; 32 key bits exactly
    (64, 150, 5, 1, 115), (64, 150, 5, 0, 119), (48, 100, 4, 1, 112), (36, 80, 3, 1, 104),
])
def test_match_unique_random(ora, seedl, patl, k, scores, pb):
    # (short seeds on a 3 kbp genome: equal ranges of hundreds of entries -> queue refills, saturated groups)
    g = synth.random_genome(200_000 if seedl >= 16 else 3000, seed=100 + seedl + patl, n_frag=5, n_runs=20, repeats=30)
    b = synth.sample_reads(g, 4000 if seedl >= 16 else 300, patl, 0.02, seed=200 + patl, n_read_prob=0.0005)
    seedk = min(2, k)
    p = ora.make_params(seedl=seedl, seedkmax=seedk, totalkmax=k, scores=scores)
    oinfo, oscore, octr = _oracle_unique(ora, None, g.sym, g.frag_start, seedl, 0, p, b.bases, b.qual, b.offsets)
    m = UniqueMatcher(_opts(seedl, seedk, k, scores), prefix_bits=abs(pb) % 100, table_kind=3 if pb >= 100 else (2 if pb < 0 else 0))   # pb < 0: directory tables, pb >= 100: bucket rows
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    info, score = m.match_unique(b.bases, b.qual, patl=patl)        # uniform-length batch form
    _compare_unique(info, score, oinfo, oscore, scores)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    m.close()


def test_match_all_fine_tables(ora):
    g = synth.random_genome(100_000, seed=91, n_frag=3, n_runs=5, repeats=40)
    b = synth.sample_reads(g, 3000, 60, 0.02, seed=92)
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 16)
    p = ora.make_params(seedl=16, seedkmax=2, totalkmax=3, scores=1)
    ohits, ooff, octr = ora.match_all(og, ix, p, b.bases, b.qual, b.offsets)
    m = AllMatcher(_opts(16, 2, 3, 1), prefix_bits=13)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    hits, hoff = m.match_all(b.bases, b.qual, b.offsets)
    assert np.array_equal(hoff, ooff)
    for x, y in zip(_hits_tuple(hits), _hits_tuple(ohits)):
        assert np.array_equal(x, y)
    c = m.counters()
    for k in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[k] == octr[k]
    m.close()


@pytest.mark.parametrize("scores,rows", [(1, 0), (0, 0), (1, 1)])
def test_match_all_many_hits_per_read(ora, scores, rows):
    # 8-base seeds on a 3 kbp tandem repeat (period 40, 2 % diverged copies): tens of hits per read -- the per-read
    # ordering pass takes its workgroup-per-read path (> 32 hits), the matcher its repeat pass (scores on)
    g = synth.random_genome(3000, seed=31, n_frag=2)
    rng = np.random.default_rng(33)
    unit = rng.integers(0, 4, size=40, dtype=np.uint8)
    g.sym[:] = np.where(rng.random(3000) < 0.02, rng.integers(0, 4, size=3000, dtype=np.uint8), np.tile(unit, 75))
    b = synth.sample_reads(g, 400, 30, 0.03, seed=32)
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 8)
    p = ora.make_params(seedl=8, seedkmax=2, totalkmax=4, scores=scores)
    ohits, ooff, octr = ora.match_all(og, ix, p, b.bases, b.qual, b.offsets)
    assert int(np.diff(ooff).max()) > 32, "test data must hold a read with more than 32 hits"
    m = AllMatcher(_opts(8, 2, 4, scores), prefix_bits=5 if rows else 0, table_kind=3 if rows else 0)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    hits, hoff = m.match_all(b.bases, b.qual, b.offsets, cap=len(ohits) + 8)   # (no overflow retry: work counted once)
    assert np.array_equal(hoff, ooff)
    for x, y in zip(_hits_tuple(hits), _hits_tuple(ohits)):
        assert np.array_equal(x, y)
    c = m.counters()
    for k in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[k] == octr[k]
    m.close()


@pytest.mark.parametrize("kind,pb", [(3, 4), (3, 7), (0, 7), (0, 0)])
def test_saturated_key_groups(ora, kind, pb):
    # a 3 kbp tandem repeat of period 10 and 8-base seeds: a handful of signatures with 300 entries each -- the 8-bit /
    # 4-bit group counts of the bucket rows / directory tables saturate and the bounds come from the binary search
    g = synth.random_genome(3000, seed=61)
    rng = np.random.default_rng(62)
    unit = rng.integers(0, 4, size=10, dtype=np.uint8)
    g.sym[:] = np.where(rng.random(3000) < 0.01, rng.integers(0, 4, size=3000, dtype=np.uint8), np.tile(unit, 300))
    b = synth.sample_reads(g, 200, 24, 0.03, seed=63)
    p = ora.make_params(seedl=8, seedkmax=2, totalkmax=3, scores=1)
    oinfo, oscore, octr = _oracle_unique(ora, None, g.sym, g.frag_start, 8, 0, p, b.bases, b.qual, b.offsets)
    m = UniqueMatcher(_opts(8, 2, 3, 1), prefix_bits=pb, table_kind=kind)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    info, score = m.match_unique(b.bases, b.qual, b.offsets)
    _compare_unique(info, score, oinfo, oscore, 1)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    m.close()


@pytest.mark.parametrize("kind,pb,device_build", [(3, 13, True), (3, 13, False), (2, 13, True), (0, 0, True)])
def test_index_blocks_compose(ora, kind, pb, device_build):
    # the genome in three index blocks (ListSetBlockReader.hpp:24-52): the records fold across the blocks exactly as the
    # reference's uniqueinfo[] does; every block rebuilds the resident tables (rows, directory, starts) in place
    g = synth.random_genome(90_000, seed=71, n_frag=2, n_runs=4, repeats=12)
    b = synth.sample_reads(g, 3000, 60, 0.02, seed=72)
    n_list = 32_000
    p = ora.make_params(seedl=16, seedkmax=2, totalkmax=3, scores=1)
    oinfo, oscore, _ = _oracle_unique(ora, None, g.sym, g.frag_start, 16, n_list, p, b.bases, b.qual, b.offsets)
    m = UniqueMatcher(_opts(16, 2, 3, 1), prefix_bits=pb, table_kind=kind)
    if device_build:
        m.set_text_symbols(0, g.sym, g.frag_start)
    else:
        text, wild = host_index.pack_text(g.sym)
        m.set_text(0, text, wild, g.n, g.frag_start)
    info = score = None
    first, nxt, blocks = 0, True, 0
    while nxt:
        if device_build:
            n, nxt = m.build_index_block(first, n_list)
        else:
            sign, pos, n, nxt = host_index.build_lists(g.sym, 16, first, n_list)
            m.set_index_block(sign, pos)
        first += n
        blocks += 1
        info, score = m.match_unique(b.bases, b.qual, b.offsets, info=info, score=score)
    assert blocks == 3
    _compare_unique(info, score, oinfo, oscore, 1)
    m.close()


def test_index_layout_device_equals_host(ora):
    g = synth.random_genome(50_000, seed=77, n_frag=3, n_runs=10, repeats=10)
    for seedl, kw in ((32, {}), (64, {}), (12, {}), (16, dict(table_kind=3, prefix_bits=12)), (16, dict(table_kind=3)),
                      (64, dict(table_kind=3, prefix_bits=13))):
        a = UniqueMatcher(_opts(seedl, 2, 3, 1), **kw)
        a.set_text_symbols(0, g.sym, g.frag_start)
        a.build_index_block()
        h = UniqueMatcher(_opts(seedl, 2, 3, 1), **kw)
        text, wild = host_index.pack_text(g.sym)
        h.set_text(0, text, wild, g.n, g.frag_start)
        sign, pos, n, nxt = host_index.build_lists(g.sym, seedl)
        h.set_index_block(sign, pos)
        assert a.n_entries == h.n_entries == n and a.prefix_bits == h.prefix_bits
        og = ora.Genome(g.sym, g.frag_start)
        oix = ora.Index(og, seedl)
        for k in range(6):
            ea, ba = a.index_download(k)
            eh, bh = h.index_download(k)
            assert np.array_equal(ea, eh) and np.array_equal(ba, bh)
            sg, ps = a.index_export(k)
            assert np.array_equal(sg.astype(np.uint64), oix.sign(k)) and np.array_equal(ps, oix.pos(k)), "exported list != reference list"
            osign, opos = oix.sign(k), oix.pos(k)
            if a.table_kind == 3:
                # bucket rows are addressed by the mixed signature (real_hip_internal.h: rh_mix32 / rh_mix64: sign * odd constant
                # mod 2^seedl, a bijection): the device list is the reference's list stably re-sorted by it -- equal
                # signatures stay together and in ascending position, which is all the matcher's candidate order needs
                mult, mask = (0x9E3779B1, (1 << seedl) - 1) if seedl <= 32 else (0x9E3779B97F4A7C15, (1 << seedl) - 1)
                mixed = np.array([(int(x) * mult) & mask for x in osign], dtype=np.uint64)
                order = np.argsort(mixed, kind="stable")
                osign, opos = mixed[order], opos[order]
            assert np.array_equal(ea[:, 1], opos), "device list order != reference list order"
            # bucket table: starts are the lower bounds of the prefixes
            pref = (osign >> np.uint64(seedl - a.prefix_bits)).astype(np.int64)
            want = np.searchsorted(pref, np.arange((1 << a.prefix_bits) + 1), side="left")
            assert np.array_equal(ba.astype(np.int64), want)
        a.close(); h.close()


def test_errors_are_loud():
    from real_amd.lib import RealHipError
    m = UniqueMatcher(_opts(32, 2, 3, 1))
    with pytest.raises(RealHipError):           # no text yet
        m.build_index_block()
    g = synth.random_genome(5000, seed=5)
    m.set_text_symbols(0, g.sym, g.frag_start)
    with pytest.raises(RealHipError):           # no index yet
        m.match_unique(np.zeros(100, np.uint8), np.zeros(100, np.uint8), patl=100)
    m.build_index_block()
    with pytest.raises(RealHipError):           # longer than REAL_HIP_MAX_PATL_LONG: an error, not a silent skip
        m.match_unique(np.zeros(17000, np.uint8), np.zeros(17000, np.uint8), patl=17000)
    info, score = m.match_unique(np.zeros(300, np.uint8), np.zeros(300, np.uint8), patl=300)     # (longer than the registers hold: a wave's job)
    # device-resident offsets with a declared max_patl are never scanned by the host: a span of 2^32 + 100 bases must not
    # alias to a 100-base read inside the kernel -- the kernel raises its flag, the call fails
    import torch
    off = torch.tensor([0, 100, (1 << 32) + 200], dtype=torch.int64, device="cuda")
    zb, zq = torch.zeros(4096, dtype=torch.uint8, device="cuda"), torch.zeros(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(RealHipError):
        m.match_unique(zb, zq, off, max_patl=100, info=torch.zeros(2, dtype=torch.int64, device="cuda"),
                       score=torch.zeros(2, dtype=torch.float32, device="cuda"))
    assert info.shape[0] == 1
    # empty batch and all-skipped batch are fine
    info, score = m.match_unique(np.zeros(0, np.uint8), np.zeros(0, np.uint8), patl=100, n_reads=0)
    assert info.shape[0] == 0
    info, score = m.match_unique(np.full(40, 4, np.uint8), np.zeros(40, np.uint8), patl=20)
    assert np.all(info == 0)
    m.close()


def _repeat_cliff_case(seedl, n_poly, n_copies):
    """a random genome with (a) a poly-A run of n_poly bases -- n_poly windows with signature 0 in all six lists, every
    one of them a verified hit of a poly-A read -- and (b) n_copies planted copies of one seed-length/2 word followed by
    random bases: an equal range of n_copies entries in list 0 whose partner signatures differ"""
    rng = np.random.default_rng(7 + seedl)
    half = seedl // 2
    parts = [rng.integers(0, 4, size=400_000, dtype=np.uint8), np.zeros(n_poly + seedl, dtype=np.uint8),
             rng.integers(0, 4, size=100_000, dtype=np.uint8)]
    word = rng.integers(0, 4, size=half, dtype=np.uint8)
    tail = rng.integers(0, 4, size=(n_copies, 40 - half), dtype=np.uint8)
    parts.append(np.concatenate([np.broadcast_to(word, (n_copies, half)), tail], axis=1).reshape(-1))
    parts.append(rng.integers(0, 4, size=100_000, dtype=np.uint8))
    sym = np.concatenate(parts)
    frag = np.array([0, 250_000, sym.shape[0]], dtype=np.uint64)
    g = synth.Genome(sym=sym, frag_start=frag, frag_names=[" a", " b"])
    patl = 100
    reads = []
    planted0 = 400_000 + n_poly + seedl + 100_000
    for i in range(12):
        reads.append(np.zeros(patl, dtype=np.uint8))                                  # poly-A
        reads.append(np.full(patl, 3, dtype=np.uint8))                                # poly-T: its reverse complement
    for i in range(100):
        p = planted0 + 40 * int(rng.integers(0, n_copies - 4))
        r = sym[p:p + patl].copy()
        if i % 2:
            r = synth.revcomp(r)
        if i % 3 == 0:
            r[50 + i % 40] = (r[50 + i % 40] + 1) & 3
        reads.append(r)
    b2 = synth.sample_reads(g, 2000, patl, 0.02, seed=5)
    bases = np.concatenate([np.concatenate(reads), b2.bases])
    n = len(reads) + b2.n_reads
    qual = np.concatenate([np.full(len(reads) * patl, 30, dtype=np.uint8), b2.qual])
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(patl)
    return g, bases, qual, offsets, patl


@pytest.mark.parametrize("seedl,kind,pb,scores", [(32, 0, 0, 1), (16, 3, 13, 1), (16, 2, 13, 1), (32, 0, 0, 0)])
def test_repeat_cliff_is_bit_exact_and_bounded_in_time(ora, seedl, kind, pb, scores):
    """Reads on signatures with 10^5 entries: bit-exact against the oracle, and matched in bounded wall time.  One lane
    walking such an equal range alone (as the reference's thread does, match.hpp:383-413) takes 0.1-1 s per lookup; a
    wave walks it 64 entries at a time (match_wave.hip)."""
    import time
    g, bases, qual, offsets, patl = _repeat_cliff_case(seedl, 100_000, 60_000)
    p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=3, scores=scores)
    oinfo, oscore, octr = _oracle_unique(ora, None, g.sym, g.frag_start, seedl, 0, p, bases, qual, offsets)
    m = UniqueMatcher(_opts(seedl, 2, 3, scores), prefix_bits=pb, table_kind=kind)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    m.match_unique(bases, qual, offsets)                    # warm-up (allocations)
    m.counters(reset=True)
    for k in (0, 4):
        m.kernel_time(k, reset=True)
    t0 = time.perf_counter()
    info, score = m.match_unique(bases, qual, offsets)
    dt = time.perf_counter() - t0
    _compare_unique(info, score, oinfo, oscore, scores)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    assert c["handed_over"] >= 24, c                         # the poly-A / poly-T reads at least
    assert c["candidates"] > 24 * 100_000                    # the case is what it claims to be
    assert dt < 0.5, "repeat-rich reads took %.3f s (lane-per-read kernel %.1f ms, wave-per-read kernel %.1f ms)" % (
        dt, m.kernel_time(0)[0], m.kernel_time(4)[0])
    m.close()


@pytest.mark.parametrize("copies,kind", [(2, 3), (3, 3), (4, 3), (6, 3), (3, 2), (4, 0), (10, 3), (14, 3), (22, 3), (30, 3), (6, 0)])
def test_reads_on_few_copy_repeats_stay_with_the_lane_matcher(ora, copies, kind):
    """Exact copies of a 1 kbp segment (what a real genome is full of): every read on them has `copies` locations per
    strand, each reached through up to six lists.  Records, scores and counters are the oracle's; and up to four copies
    the reads are matched by the first pass of the lane-per-read kernel -- a window is queued once, a parked location is
    recognised when it comes again (match_kernel.hip: queue_push, process_loaded).  Five to twenty-four copies are the second
    pass' (bucket rows: the same kernel with twenty-four parked locations per lane), more the wave-per-read kernel's."""
    rng = np.random.default_rng(11 + copies)
    G = 300_000
    sym = rng.integers(0, 4, size=G, dtype=np.uint8)
    src = 10_000
    for c in range(copies - 1):
        d = 40_000 + min(18_000, 250_000 // copies) * c
        sym[d:d + 1000] = sym[src:src + 1000]
    frag = np.array([0, G], dtype=np.uint64)
    reads = []
    for i in range(120):
        r = sym[src + 50 + 7 * i: src + 150 + 7 * i].copy()
        if i % 2:
            r = synth.revcomp(r)
        if i % 3 == 0:
            r[(11 * i) % 100] = (r[(11 * i) % 100] + 1) & 3          # an error, in the seed or behind it
        reads.append(r)
    bases = np.concatenate(reads).astype(np.uint8)
    qual = (rng.integers(5, 40, size=bases.shape[0])).astype(np.uint8)
    offsets = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(100)
    # (bucket rows of 32-bit signatures are 2^28 rows per list whatever the genome: the row cases run with 16-base seeds, 2^13 rows)
    seedl, pb = (16, 13) if kind == 3 else (32, 29 if kind == 2 else 0)
    p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=3, scores=1)
    oinfo, oscore, octr = _oracle_unique(ora, None, sym, frag, seedl, 0, p, bases, qual, offsets)
    m = UniqueMatcher(_opts(seedl, 2, 3, 1), table_kind=kind, prefix_bits=pb)
    m.set_text_symbols(0, sym, frag)
    m.build_index_block()
    assert m.table_kind == (3 if kind == 3 else (1 if kind == 2 else 0)), m.table_kind
    info, score = m.match_unique(bases, qual, patl=100)
    _compare_unique(info, score, oinfo, oscore, 1)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    assert (unpack_info(info)[0] == 4).sum() >= 100           # NonUnique: the case is what it claims to be
    if copies <= 4:
        assert c["handed_over"] == 0, c
    else:
        assert c["handed_over"] > 0, c
    m.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 1000])
def test_batch_sizes_around_the_tile_of_64_reads(ora, n):
    """The waves of the resident grid take tiles of 64 reads from a counter: batches of less than a tile, of whole tiles
    and with a partial last tile, packed bases and bytes."""
    g = synth.random_genome(80_000, seed=400, n_frag=2, n_runs=2, repeats=6)
    b = synth.sample_reads(g, n, 100, 0.02, seed=401 + n)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=3, scores=1)
    oinfo, oscore, octr = _oracle_unique(ora, None, g.sym, g.frag_start, 32, 0, p, b.bases, b.qual, b.offsets)
    m = UniqueMatcher(_opts(32, 2, 3, 1), table_kind=3, prefix_bits=13)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    info, score = m.match_unique(b.bases, b.qual, patl=100)
    _compare_unique(info, score, oinfo, oscore, 1)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    if n % 4 == 0:                                           # the same batch with the bases 2-bit packed
        q4 = b.bases.reshape(-1, 4)
        pk = ((q4[:, 0] << 6) | (q4[:, 1] << 4) | (q4[:, 2] << 2) | q4[:, 3]).astype(np.uint8)
        nfl = np.zeros((n + 7) // 8, dtype=np.uint8)
        bad = np.nonzero((b.bases.reshape(n, 100) > 3).any(axis=1))[0]
        np.bitwise_or.at(nfl, bad // 8, (1 << (bad % 8)).astype(np.uint8))
        pk = ((np.minimum(q4[:, 0], 3) << 6) | (np.minimum(q4[:, 1], 3) << 4) | (np.minimum(q4[:, 2], 3) << 2) | np.minimum(q4[:, 3], 3)).astype(np.uint8)
        info2, score2 = m.match_unique(pk, b.qual, patl=100, n_reads=n, packed=True, nflags=nfl)
        _compare_unique(info2, score2, oinfo, oscore, 1)
    m.close()


@pytest.mark.parametrize("patl,shift", [(100, 0), (100, 4), (100, 7), (150, 0), (150, 5), (36, 3)])
def test_device_batch_at_any_address(ora, patl, shift):
    """Resident batches whose arrays start at any byte address (a slice of a larger device buffer): the qualities of a wave
    travel by LDS-DMA only when their first byte lies at a multiple of 16 (match_kernel.hip: quals_ahead), in one piece up to
    8 KiB per wave and in two above (150 bp reads); everything else is staged by the wave itself.  Same records either way."""
    import torch
    seedl = 32 if patl < 150 else 64
    k = 3 if patl < 150 else 5
    g = synth.random_genome(150_000, seed=300 + patl, n_frag=2, n_runs=4, repeats=10)
    b = synth.sample_reads(g, 3000, patl, 0.02, seed=301 + shift)
    p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=k, scores=1)
    oinfo, oscore, _ = _oracle_unique(ora, None, g.sym, g.frag_start, seedl, 0, p, b.bases, b.qual, b.offsets)
    m = UniqueMatcher(_opts(seedl, 2, k, 1), table_kind=3, prefix_bits=14)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    n = b.n_reads
    big_b = torch.zeros(n * patl + 64, dtype=torch.uint8, device="cuda")
    big_q = torch.zeros(n * patl + 64, dtype=torch.uint8, device="cuda")
    db, dq = big_b[shift:shift + n * patl], big_q[shift:shift + n * patl]
    db.copy_(torch.from_numpy(b.bases)); dq.copy_(torch.from_numpy(b.qual))
    info = torch.zeros(n, dtype=torch.int64, device="cuda")
    score = torch.full((n,), float(np.float32(ora.NOSCORE_INIT)), dtype=torch.float32, device="cuda")
    m.match_unique(db, dq, patl=patl, info=info, score=score, n_reads=n)
    _compare_unique(info.cpu().numpy().view(np.uint64), score.cpu().numpy(), oinfo, oscore, 1)
    m.close()


@pytest.mark.parametrize("where", ["host", "device"])
def test_fresh_batch_ignores_what_the_record_arrays_hold(ora, where):
    """real_hip_batch.fresh: info / score are outputs only; every read -- matched by a lane, handed to the wave matcher
    (the poly-A reads), skipped (N reads, short reads) -- starts as uniqueinfo(numpat) does (UniqueMatchInfo.hpp:191)."""
    g, bases, qual, offsets, patl = _repeat_cliff_case(16, 20_000, 5_000)
    bases = bases.copy()
    bases[5 * patl + 3] = 4                                  # a read with an N: not matched at all
    p = ora.make_params(seedl=16, seedkmax=2, totalkmax=3, scores=1)
    oinfo, oscore, _ = _oracle_unique(ora, None, g.sym, g.frag_start, 16, 0, p, bases, qual, offsets)
    m = UniqueMatcher(_opts(16, 2, 3, 1), prefix_bits=13, table_kind=3)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    n = offsets.shape[0] - 1
    info = np.full(n, 0x7123456789abcdef, dtype=np.uint64)   # garbage that would win every fold
    score = np.full(n, 1e30, dtype=np.float32)
    if where == "device":
        import torch
        d = [torch.from_numpy(x).cuda() for x in (bases, qual, offsets.view(np.int64), info.view(np.int64), score)]
        m.match_unique(d[0], d[1], d[2], info=d[3], score=d[4], fresh=True)
        info, score = d[3].cpu().numpy().view(np.uint64), d[4].cpu().numpy()
    else:
        m.match_unique(bases, qual, offsets, info=info, score=score, fresh=True)
    assert m.counters()["handed_over"] > 0
    _compare_unique(info, score, oinfo, oscore, 1)
    m.close()


@pytest.mark.parametrize("seedl,kind,pb", [(32, 0, 0), (16, 3, 13)])
def test_repeat_cliff_match_all(ora, seedl, kind, pb):
    """matchAll on the same case: 10^5 hits per read.  The wave-cooperative matcher appends them behind ballots, the
    ordering pass sorts such a read's segment with a workgroup radix sort instead of the quadratic ranking."""
    import time
    g, bases, qual, offsets, patl = _repeat_cliff_case(seedl, 100_000, 60_000)
    p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=2, scores=1)
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, seedl)
    ohits, ooff, octr = ora.match_all(og, ix, p, bases, qual, offsets)
    m = AllMatcher(_opts(seedl, 2, 2, 1), prefix_bits=pb, table_kind=kind)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    cap = int(ohits.shape[0]) + 1024
    m.match_all(bases, qual, offsets, cap=cap)               # warm-up (allocations)
    m.counters(reset=True)
    for k in (1, 2, 4):
        m.kernel_time(k, reset=True)
    t0 = time.perf_counter()
    hits, hoff = m.match_all(bases, qual, offsets, cap=cap)
    dt = time.perf_counter() - t0
    assert np.array_equal(hoff, ooff), "per-read hit counts differ"
    assert int(np.diff(hoff.astype(np.int64)).max()) >= 90_000
    for x, y in zip(_hits_tuple(hits), _hits_tuple(ohits)):
        assert np.array_equal(x, y)
    c = m.counters()
    for k in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[k] == octr[k], (k, c[k], octr[k])
    assert dt < 1.0, "repeat-rich reads took %.3f s (lane-per-read kernel %.1f ms, wave-per-read kernel %.1f ms, ordering pass %.1f ms)" % (
        dt, m.kernel_time(1)[0], m.kernel_time(4)[0], m.kernel_time(2)[0])
    m.close()


@pytest.mark.parametrize("seedl,kind,pb", [(32, 0, 0), (32, 2, 29), (16, 3, 13), (16, 3, 12), (16, 2, 13), (16, 0, 0)])
def test_near_copies_reach_the_fold_in_the_reference_order(ora, seedl, kind, pb):
    """VERDICT r2 'weak' 1.  Scores on: three near-copies X, A, C of one read whose scores lie an epsilon-step apart, found
    through different lists (synth.near_copy_case).  The reference calls update() list by list
    (matchUniqueImplementation.cpp:407-497, match.hpp:383-413): X, A | X, C | X, A, C | ... and ends NonUnique; a matcher that
    delivers all events of a window at the window's first turn ends Straight at C.  The lane matcher queues a window once
    and must still deliver its events in the reference's order -- and must not dodge the case by handing the read over."""
    g, b, pos_c = synth.near_copy_case(seedl=seedl)
    p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=3, scores=1)
    oinfo, oscore, octr = _oracle_unique(ora, None, g.sym, g.frag_start, seedl, 0, p, b.bases, b.qual, b.offsets)
    st, fr, er, fi, po = unpack_info(oinfo)
    assert (int(st[0]), int(po[0])) == (4, pos_c), "the case is what it claims to be: NonUnique in the reference's order"
    m = UniqueMatcher(_opts(seedl, 2, 3, 1), prefix_bits=pb, table_kind=kind)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    # the read alone, and as every lane of two tiles (both strands: the reverse complement finds the same three windows)
    for reps in (1, 128):
        bases = np.concatenate([b.bases if i % 2 == 0 else synth.revcomp(b.bases) for i in range(reps)])
        qual = np.concatenate([b.qual if i % 2 == 0 else b.qual[::-1] for i in range(reps)])
        m.counters(reset=True)
        info, score = m.match_unique(bases, qual, patl=100)
        c = m.counters()
        assert c["handed_over"] == 0, c
        for i in range(reps):
            s_, f_, e_, fi_, p_ = unpack_info(info[i:i + 1])
            assert int(s_[0]) == 4 and int(e_[0]) == 1, "read %d: state %d (reference: NonUnique)" % (i, int(s_[0]))
            assert np.float32(score[i]).view(np.uint32) == oscore.view(np.uint32)[0]
        assert info[0] == oinfo[0]
    m.close()


def _grouped_delivery_differs(ora, ev, oinfo, oscore, eps):
    """number of reads whose record differs when every window's update() events are delivered at the window's first turn"""
    import ctypes as C
    n = oinfo.shape[0]
    evs = ev[np.argsort(ev["read"], kind="stable")]
    bounds = np.searchsorted(evs["read"], np.arange(n + 1))
    nd = 0
    for r in range(n):
        seen = {}
        for x in evs[bounds[r]:bounds[r + 1]]:
            seen.setdefault((int(x["inverted"]), int(x["pos"])), []).append(x)
        i = np.zeros(1, np.uint64)
        s = np.full(1, ora.NOSCORE_INIT, np.float32)
        for lst in seen.values():
            for x in lst:
                ora.lib().ora_update_unique(1, int(x["inverted"]), 0, int(x["pos"]), int(x["totalk"]), C.c_float(float(x["score"])), C.c_float(eps),
                                            int(x["frag"]), i.ctypes.data, s.ctypes.data)
        nd += int(i[0] != oinfo[r] or s.view(np.uint32)[0] != oscore.view(np.uint32)[r])
    return nd


@pytest.mark.parametrize("seedl,patl,k,kind,pb", [
    (32, 100, 3, 0, 0), (32, 100, 3, 2, 29), (16, 100, 3, 3, 13), (16, 60, 4, 3, 13), (16, 60, 4, 2, 13), (16, 60, 4, 0, 0),
    (64, 150, 5, 0, 0), (64, 150, 5, 2, 15), (64, 150, 5, 3, 13), (36, 80, 3, 3, 12), (20, 120, 2, 0, 0)])
def test_diverged_copies_random(ora, seedl, patl, k, kind, pb):
    """The same class at random: every read has 2..4 near-copies on either strand that differ from it in different seed
    segments / behind the seed, and random qualities 0..40 -- the copies are members of different lists' equal ranges
    and score differently, so the record depends on the order of the update() calls.  All table kinds, 32-bit and wider
    signatures; lane matcher and (where four locations do not suffice) wave matcher."""
    g = synth.random_genome(700_000, seed=900 + seedl, n_frag=3, n_runs=6)
    b = synth.diverged_copy_reads(g, 700, patl, seedl, seed=901 + patl + kind)
    p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=k, scores=1)
    og = ora.Genome(g.sym, g.frag_start)
    oinfo, oscore, octr, ev = ora.match_unique(og, ora.Index(og, seedl), p, b.bases, b.qual, b.offsets, want_events=True)
    ost = unpack_info(oinfo)[0]
    assert (ost == 4).sum() >= 50 and ((ost == 1) | (ost == 2)).sum() >= 20
    assert _grouped_delivery_differs(ora, ev, oinfo, oscore, float(np.float32(p.filter_mult * patl))) >= 2, "the case must hold reads whose record depends on the order"
    m = UniqueMatcher(_opts(seedl, 2, k, 1), prefix_bits=pb, table_kind=kind)
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    info, score = m.match_unique(b.bases, b.qual, patl=patl)
    _compare_unique(info, score, oinfo, oscore, 1)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    assert c["handed_over"] < b.n_reads // 4, c            # most of them are the lane matcher's
    m.close()


def _two_file_genome():
    """two genome files (file ids 0 and 1, UniqueMatchInfo.hpp:31) that share one 600-base stretch at the SAME position of the
    SAME fragment number: a read from it has two best hits that differ in the file id alone"""
    g0 = synth.random_genome(60_000, seed=501, n_runs=3, repeats=6)
    g1 = synth.random_genome(50_000, seed=502, n_runs=2, repeats=4)
    g0 = synth.Genome(sym=g0.sym, frag_start=np.array([0, 30_000, 60_000], dtype=np.uint64), frag_names=[" zero_0", " zero_1"])
    g1 = synth.Genome(sym=g1.sym, frag_start=np.array([0, 20_000, 35_000, 50_000], dtype=np.uint64), frag_names=[" one_0", " one_1", " one_2"])
    g1.sym[1000:1600] = g0.sym[1000:1600]
    g0.sym[1000:1600] = np.where(g0.sym[1000:1600] > 3, 0, g0.sym[1000:1600])        # (no N inside the shared stretch)
    g1.sym[1000:1600] = g0.sym[1000:1600]
    b0 = synth.sample_reads(g0, 900, 100, 0.02, seed=503)
    b1 = synth.sample_reads(g1, 700, 100, 0.02, seed=504)
    sh = synth.sample_reads(synth.Genome(sym=g0.sym[1000:1600].copy(), frag_start=np.array([0, 600], dtype=np.uint64)), 200, 100, 0.01, seed=505)
    return g0, g1, synth.concat_batches([b0, b1, sh])


@pytest.mark.parametrize("scores,kind,pb,seedl", [(1, 0, 0, 32), (0, 0, 0, 32), (1, 3, 13, 16), (1, 2, 29, 32)])
def test_two_genome_files_fold_through_the_file_id(ora, scores, kind, pb, seedl):
    """matchUniqueImplementation.cpp:1099-1118 walks the genome files one after the other over the same uniqueinfo[]; the
    fold compares the file id (:131, :219).  Reads of file 0, of file 1, and of a stretch both files hold at the same
    position and fragment (NonUnique through the file id alone); file 1 is matched with fileid 1 in the records."""
    g0, g1, b = _two_file_genome()
    info, score = new_unique_info(b.n_reads, scores)
    oinfo, oscore = new_unique_info(b.n_reads, scores)
    if oscore is None:
        oscore = np.full(b.n_reads, ora.NOSCORE_INIT, dtype=np.float32)
    m = UniqueMatcher(_opts(seedl, 2, 3, scores), prefix_bits=pb, table_kind=kind)
    for fid, g in enumerate((g0, g1)):
        p = ora.make_params(seedl=seedl, seedkmax=2, totalkmax=3, scores=scores, fileid=fid)
        og = ora.Genome(g.sym, g.frag_start)
        oinfo, oscore, _ = ora.match_unique(og, ora.Index(og, seedl), p, b.bases, b.qual, b.offsets, info=oinfo, score=oscore)
        m.set_text_symbols(fid, g.sym, g.frag_start)
        m.build_index_block()
        m.match_unique(b.bases, b.qual, b.offsets, info=info, score=score)
    _compare_unique(info, score, oinfo, oscore, scores)
    st, fr, er, fi, po = unpack_info(info)
    uniq = (st == 1) | (st == 2)
    assert (uniq & (fi == 0)).sum() > 500 and (uniq & (fi == 1)).sum() > 400, "both files have uniquely aligned reads"
    assert (st[-200:] == 4).sum() >= 150, "reads of the shared stretch are NonUnique through the file id"
    m.close()


@pytest.mark.parametrize("scores", [1, 0])
def test_second_pass_across_index_blocks_and_fresh_records(ora, scores):
    """Reads on six copies of a locus, twice: six copies inside the first index block and six more inside the third.  The
    first pass of the lane matcher hands such a read on (more than four locations), the second pass folds into the record
    the first pass started (block 1: `fresh` -- the arrays hold garbage that would win every fold) or the previous blocks
    left (blocks 2, 3): the records compose across the blocks exactly as the reference's uniqueinfo[] does
    (ListSetBlockReader.hpp:24-52), whoever matched the read in which block."""
    rng = np.random.default_rng(77)
    G = 96_000
    sym = rng.integers(0, 4, size=G, dtype=np.uint8)
    src = 2_000
    for c in range(1, 6):
        sym[src + 5_000 * c: src + 5_000 * c + 1000] = sym[src:src + 1000]
    for c in range(6):
        sym[66_000 + 4_500 * c: 66_000 + 4_500 * c + 1000] = sym[src:src + 1000]
    frag = np.array([0, 50_000, G], dtype=np.uint64)
    reads = []
    for i in range(150):
        r = sym[src + 20 + 5 * i: src + 120 + 5 * i].copy()
        if i % 2:
            r = synth.revcomp(r)
        if i % 4 == 0:
            r[(13 * i) % 100] = (r[(13 * i) % 100] + 1) & 3
        reads.append(r)
    extra = synth.sample_reads(synth.Genome(sym=sym, frag_start=frag), 400, 100, 0.02, seed=78)
    bases = np.concatenate([np.concatenate(reads), extra.bases]).astype(np.uint8)
    n = 150 + extra.n_reads
    qual = rng.integers(3, 40, size=bases.shape[0]).astype(np.uint8)
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(100)
    n_list = 32_000
    p = ora.make_params(seedl=16, seedkmax=2, totalkmax=3, scores=scores)
    oinfo, oscore, octr = _oracle_unique(ora, None, sym, frag, 16, n_list, p, bases, qual, offsets)
    m = UniqueMatcher(_opts(16, 2, 3, scores), prefix_bits=13, table_kind=3)
    m.set_text_symbols(0, sym, frag)
    info = np.full(n, 0x7123456789abcdef, dtype=np.uint64)
    score = np.full(n, 1e30, dtype=np.float32) if scores else None
    first, nxt, blocks = 0, True, 0
    m.counters(reset=True)
    while nxt:
        cnt, nxt = m.build_index_block(first, n_list)
        assert m.table_kind == 3
        first += cnt
        m.match_unique(bases, qual, offsets, info=info, score=score, fresh=(blocks == 0))
        blocks += 1
    assert blocks == 3
    _compare_unique(info, score, oinfo, oscore if scores else None, scores)
    c = m.counters()
    for kk in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[kk] == octr[kk], (kk, c[kk], octr[kk])
    if scores:
        assert c["handed_over"] >= 200, c            # the 150 reads on the copies, in two of the three blocks
    m.close()
