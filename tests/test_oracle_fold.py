"""The fold (UpdateUniqueInfo::update) and unifyMatches restatements.

These two pieces live in matchUniqueImplementation.cpp / matchAllImplementation.cpp,
which cannot be compiled from the reference here (they include the autoconf-generated
real_config.hpp), so no compiled-reference vector pins them ("parity unpinned",
DESIGN.md section 2).  The cases below are derived by hand from the source text
(matchUniqueImplementation.cpp:97-160, 179-248; matchAllImplementation.cpp:122-161).
"""
import ctypes as C

import numpy as np

NOM, STR, REV, GAP, NONU = 0, 1, 2, 3, 4
NEG = np.float32(-3.4028234663852886e38)


def fold(ora, scores, events, eps=0.0, info=0, score=NEG):
    """events: (inverted, fileid, pos, totalk, score, frag)"""
    i = np.array([info], dtype=np.uint64)
    s = np.array([score], dtype=np.float32)
    for (inv, fid, pos, k, sc, frag) in events:
        ora.lib().ora_update_unique(int(scores), int(inv), fid, pos, k, C.c_float(sc), C.c_float(eps), frag,
                                    i.ctypes.data, s.ctypes.data)
    st, fr, er, fi, po = ora.unpack_record(i)
    return int(st[0]), int(po[0]), int(er[0]), int(fr[0]), int(fi[0]), float(s[0])


def test_noscores_state_machine(ora):
    # NoMatch -> first hit is taken whatever its quality
    assert fold(ora, 0, [(0, 0, 100, 3, 1.0, 0)])[:3] == (STR, 100, 3)
    assert fold(ora, 0, [(1, 2, 100, 3, 1.0, 7)])[:5] == (REV, 100, 3, 7, 2)
    # fewer errors replace, more errors are ignored
    assert fold(ora, 0, [(0, 0, 100, 3, 1, 0), (1, 0, 500, 1, 1, 0)])[:3] == (REV, 500, 1)
    assert fold(ora, 0, [(0, 0, 100, 1, 1, 0), (1, 0, 500, 3, 1, 0)])[:3] == (STR, 100, 1)
    # same errors at a different place (pos, file or fragment) -> NonUnique, record keeps the first place
    assert fold(ora, 0, [(0, 0, 100, 2, 1, 0), (0, 0, 101, 2, 1, 0)])[:3] == (NONU, 100, 2)
    assert fold(ora, 0, [(0, 0, 100, 2, 1, 0), (0, 1, 100, 2, 1, 0)])[0] == NONU
    assert fold(ora, 0, [(0, 0, 100, 2, 1, 0), (0, 0, 100, 2, 1, 1)])[0] == NONU
    # the same place again is a no-op (also with the other strand: straight stays)
    assert fold(ora, 0, [(0, 0, 100, 2, 1, 0), (1, 0, 100, 2, 1, 0)])[:3] == (STR, 100, 2)
    # NonUnique is left only by strictly fewer errors
    assert fold(ora, 0, [(0, 0, 100, 2, 1, 0), (0, 0, 101, 2, 1, 0), (0, 0, 300, 2, 1, 0)])[0] == NONU
    assert fold(ora, 0, [(0, 0, 100, 2, 1, 0), (0, 0, 101, 2, 1, 0), (1, 0, 300, 1, 1, 0)])[:3] == (REV, 300, 1)
    # Gapped behaves like NoMatch
    gapped = int(ora.lib().ora_record_pack(GAP, 0, 5, 0, 9))
    assert fold(ora, 0, [(0, 0, 100, 7, 1, 0)], info=gapped)[:3] == (STR, 100, 7)


def test_scores_state_machine_and_order_dependence(ora):
    eps = 1.0
    # replace iff score > old + eps
    assert fold(ora, 1, [(0, 0, 100, 3, 10.0, 0), (0, 0, 200, 0, 11.5, 0)], eps)[:2] == (STR, 200)
    # within eps at a different place -> NonUnique, record (and score) keep the first
    st, pos, err, fr, fi, sc = fold(ora, 1, [(0, 0, 100, 3, 10.0, 0), (0, 0, 200, 0, 10.9, 0)], eps)
    assert (st, pos, sc) == (NONU, 100, 10.0)
    # clearly worse -> ignored
    assert fold(ora, 1, [(0, 0, 100, 3, 10.0, 0), (0, 0, 200, 0, 8.9, 0)], eps)[:2] == (STR, 100)
    # NonUnique is left only by score > old + eps
    assert fold(ora, 1, [(0, 0, 100, 3, 10.0, 0), (0, 0, 200, 0, 10.5, 0), (1, 0, 300, 0, 11.2, 0)], eps)[:2] == (REV, 300)
    assert fold(ora, 1, [(0, 0, 100, 3, 10.0, 0), (0, 0, 200, 0, 10.5, 0), (1, 0, 300, 0, 10.9, 0)], eps)[0] == NONU
    # order dependence: A,B,C in two orders give different results
    A, B, C3 = (0, 0, 100, 0, 10.0, 0), (0, 0, 200, 0, 10.9, 0), (0, 0, 300, 0, 11.8, 0)
    assert fold(ora, 1, [A, B, C3], eps)[:2] == (STR, 300)       # A, B~A -> NonUnique(10.0), C > 11.0 -> C
    assert fold(ora, 1, [C3, B, A], eps)[0] == NONU              # C, B within eps of C -> NonUnique
    # a *duplicate* of an earlier hit is not always a no-op (why the device replays every event):
    Z, X, Y = (0, 0, 10, 0, 10.0, 0), (0, 0, 20, 0, 10.9, 0), (0, 0, 30, 0, 11.1, 0)
    assert fold(ora, 1, [Z, X, Y], eps)[:2] == (STR, 30)
    assert fold(ora, 1, [Z, X, Y, X], eps)[0] == NONU
    # eps = 0: equal scores keep the first, different place -> NonUnique; same place -> no-op
    assert fold(ora, 1, [(0, 0, 100, 0, 5.0, 0), (0, 0, 100, 0, 5.0, 0)], 0.0)[:2] == (STR, 100)
    assert fold(ora, 1, [(0, 0, 100, 0, 5.0, 0), (0, 0, 101, 0, 5.0, 0)], 0.0)[0] == STR    # 5 > 5-0 is false
    # initial score is -FLT_MAX: the first hit is taken through the NoMatch case, not by comparison
    assert fold(ora, 1, [(0, 0, 7, 2, -500.0, 0)], eps)[:2] == (STR, 7)


def test_unify_matches_order_and_dedup(ora):
    """matchAll on a tiny genome with a planted exact repeat: hits come out ordered by
    (k, pos, ...) per read and each (strand,pos) once, although up to six lists find it."""
    from real_amd import synth
    rng = np.random.default_rng(5)
    sym = rng.integers(0, 4, size=4000, dtype=np.uint8)
    sym[3000:3100] = sym[1000:1100]                 # exact repeat
    sym[2000:2100] = sym[1000:1100]; sym[2050] ^= 1  # copy with one substitution
    frag = np.array([0, 4000], dtype=np.uint64)
    og = ora.Genome(sym, frag)
    ix = ora.Index(og, 32)
    read = sym[1000:1100].copy()
    bases = np.concatenate([read, synth.revcomp(read)])
    qual = np.full(200, 30, np.uint8)
    off = np.array([0, 100, 200], dtype=np.uint64)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=2, scores=1)
    hits, hoff, ctr = ora.match_all(og, ix, p, bases, qual, off)
    assert list(hoff) == [0, 3, 6]
    for r, inv in ((0, 0), (1, 1)):
        h = hits[int(hoff[r]):int(hoff[r + 1])]
        assert list(h["k"]) == [0, 0, 1] and list(h["pos"]) == [1000, 3000, 2000]
        assert set(h["inverted"]) == {inv}
    assert ctr["hits"] > 6                            # raw update() calls include the duplicates
    # the unique fold on the same reads says NonUnique
    info, score, _ = ora.match_unique(og, ix, p, bases, qual, off)
    assert list(ora.unpack_record(info)[0]) == [NONU, NONU]


def test_near_copies_event_order_decides_the_record(ora):
    """Scores on: three near-copies X, A, C of a read whose scores lie one epsilon-step apart.  In the reference's
    order (strand, list, ascending position: matchUniqueImplementation.cpp:407-497 calls ::match list by list,
    match.hpp:383-413 calls update() entry by entry) the record ends NonUnique; the same events grouped by window
    (every window's events at its first turn) end Straight at C.  Pins the oracle's event order on this input and shows
    that a matcher which regroups the events is caught by it (tests/test_gpu_parity.py::test_near_copies_*)."""
    from real_amd import synth
    g, b, pos_c = synth.near_copy_case()
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 32)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=3, scores=True)
    info, score, ctr, ev = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets, want_events=True)
    assert [(int(e["list"]), int(e["pos"])) for e in ev] == [
        (0, 10000), (0, 20000), (1, 10000), (1, 30000), (2, 10000), (2, 20000), (2, 30000), (3, 10000), (4, 10000), (4, 20000),
        (5, 10000), (5, 30000)]
    st, fr, er, fi, po = ora.unpack_record(info)
    assert (int(st[0]), int(po[0]), int(er[0])) == (NONU, pos_c, 1)
    eps = float(np.float32(p.filter_mult * 100))
    evs = [(int(e["inverted"]), 0, int(e["pos"]), int(e["totalk"]), float(e["score"]), int(e["frag"])) for e in ev]
    assert fold(ora, 1, evs, eps)[:2] == (NONU, pos_c)                       # the oracle's own fold, replayed
    grouped = sorted(evs, key=lambda e: e[2])                               # X*6, A*3, C*3: first-reach order of the windows
    assert fold(ora, 1, grouped, eps)[:2] == (STR, pos_c)
