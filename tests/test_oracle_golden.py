"""Pin oracle/real_oracle.c against golden vectors produced by the reference's
own header-only hot path (tests/golden/make_golden.py, oracle/_ref/ref_harness).

What these vectors pin: Scoring table, text packing, signature construction,
rest-word geometry, the six sorted lists with cross pointers, the 22-bit lookup
tables, and the ordered updater::update stream of all twelve ::match calls per
read (positions, strand, mismatch counts, fragment ids, float score bits), for
one-block and multi-block indexes, N runs, several fragments, repeats, ragged
read lengths, l = 12 / 32 / 48 / 64.  What they cannot pin (the reference's
matchUniqueImplementation.cpp needs the autoconf-generated real_config.hpp):
the UpdateUniqueInfo fold and unifyMatches -- see test_oracle_fold.py.
"""
import glob
import hashlib
import os

import numpy as np
import pytest

GOLDEN = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if os.path.basename(f) != "readers.npz")   # (readers.npz: tests/test_reader_golden.py)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sparse_lookup(lk):
    lk = lk.reshape(-1, 2)
    nz = np.nonzero((lk[:, 0] != 0) | (lk[:, 1] != 0))[0]
    out = np.empty((nz.shape[0], 3), dtype=np.uint64)
    out[:, 0] = nz
    out[:, 1:] = lk[nz]
    return out.reshape(-1)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_vectors(ora, path):
    z = np.load(path)
    seedl, seedkmax, totalkmax, scores, n_list = [int(x) for x in z["params"]]
    # scoring table: bit-exact doubles
    LL, _ = ora.scoring_table()
    assert np.array_equal(LL.view(np.uint64), z["LL"].view(np.uint64))
    # record packing
    recs = [ora.lib().ora_record_pack(s, s * 1001, s * 3, s * 7, 207 + 1000 * s) for s in range(5)]
    assert np.array_equal(np.array(recs, dtype=np.uint64), z["records"])
    g = ora.Genome(z["genome"], z["frag_start"])
    assert np.array_equal(g.text, z["text"])
    bases, qual, offsets = z["bases"], z["qual"], z["offsets"]
    # signatures + rest geometry
    import ctypes as C
    for row in z["sigs"]:
        r = int(row[0])
        rd = np.ascontiguousarray(bases[int(offsets[r]):int(offsets[r + 1])])
        m = np.zeros(4, np.uint32); im = np.zeros(4, np.uint32)
        s = np.zeros(6, np.uint64); rs = np.zeros(6, np.uint64)
        assert ora.lib().ora_signature_mapped(seedl, rd.ctypes.data, m.ctypes.data)
        assert ora.lib().ora_reverse_mapped_signature(seedl, rd.ctypes.data, im.ctypes.data)
        ora.lib().ora_signatures(seedl, m.ctypes.data, s.ctypes.data)
        ora.lib().ora_signatures(seedl, im.ctypes.data, rs.ctypes.data)
        assert list(m) == list(row[1:5]) and list(im) == list(row[5:9])
        assert list(s) == list(row[9:15]) and list(rs) == list(row[15:21])
        restlen = rd.shape[0] - seedl
        assert (restlen // 32, restlen % 32) == (int(row[21]), int(row[22]))
    p = ora.make_params(seedl=seedl, seedkmax=seedkmax, totalkmax=totalkmax, scores=scores, LL=LL)
    digests = {}
    for d in z["digests"]:
        blk, k, n, hs, hp, hq, hl = str(d).split()
        digests[(int(blk), int(k))] = (int(n), hs, hp, hq, hl)
    first = 0
    for blk in range(int(z["nblocks"])):
        ix = ora.Index(g, seedl, first_window=first, max_entries=(n_list if n_list else 1 << 62))
        for k in range(6):
            n, hs, hp, hq, hl = digests[(blk, k)]
            assert ix.n == n
            assert sha(ix.sign(k)) == hs, "sorted signatures of list %d differ" % k
            assert sha(ix.ptr(k)) == hp, "cross pointers of list %d differ" % k
            assert sha(ix.pos(k)) == hq, "positions of list %d differ" % k
            assert sha(sparse_lookup(ix.lookup(k))) == hl, "lookup table of list %d differs" % k
        n_reads = offsets.shape[0] - 1
        cap = 64 * n_reads + 1024
        ev = np.zeros(cap, dtype=ora.EVENT_DTYPE)
        nev = C.c_uint64(0)
        ctr = ora.OraCounters()
        rc = ora.lib().ora_match_events(g.h, ix.h, C.byref(p), bases.ctypes.data, qual.ctypes.data,
                                        np.ascontiguousarray(offsets).ctypes.data, n_reads,
                                        ev.ctypes.data, cap, C.byref(nev), C.byref(ctr))
        assert rc == 0
        ev = ev[:nev.value]
        ref = z["events_b%d" % blk]
        assert ev.shape[0] == ref.shape[0]
        got = np.stack([ev["read"], ev["list"].astype(np.uint64), ev["inverted"].astype(np.uint64),
                        ev["pos"].astype(np.uint64), ev["totalk"].astype(np.uint64), ev["frag"].astype(np.uint64),
                        ev["score"].view(np.uint32).astype(np.uint64)], axis=1)
        assert np.array_equal(got, ref), "update() stream differs from the reference in block %d" % blk
        assert ctr.hits == ref.shape[0]
        assert ix.have_next == (blk + 1 < int(z["nblocks"]))
        first += ix.n
