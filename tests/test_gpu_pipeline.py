"""Host-side forms of a batch on the GPU: 2-bit packed bases, the two-slot submit / wait pipeline over pinned memory,
the version-1 batch struct, and the loud error for an under-declared read length (all through the C ABI)."""
import ctypes as C

import numpy as np
import pytest

from real_amd import lib as rlib
from real_amd import synth
from real_amd.matcher import RealOptions, UniqueMatcher, new_unique_info

pytestmark = pytest.mark.gpu


def _case(n_reads=6000, ragged=True, seed=5):
    g = synth.random_genome(150_000, seed=seed, n_frag=3, n_runs=6, repeats=20)
    if ragged:
        parts = [synth.sample_reads(g, n_reads // 3, pl, 0.02, seed=seed + pl, n_read_prob=0.0005) for pl in (36, 77, 100)]
        bases = np.concatenate([p.bases for p in parts])
        qual = np.concatenate([p.qual for p in parts])
        lens = np.concatenate([np.diff(p.offsets.astype(np.int64)) for p in parts])
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    else:
        b = synth.sample_reads(g, n_reads, 100, 0.02, seed=seed + 1, n_read_prob=0.0005)
        bases, qual, offsets = b.bases, b.qual, b.offsets
    return g, bases, qual, offsets


def _matcher(g, scores=True):
    m = UniqueMatcher(RealOptions(seedl=32, seedkmax=2, totalkmax=3, scores=scores).normalise())
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    return m


@pytest.mark.parametrize("ragged", [True, False])
def test_packed_batch_equals_byte_batch(ragged):
    g, bases, qual, offsets = _case(ragged=ragged)
    assert (bases > 3).any()                                   # reads with N: flagged, skipped like the reference skips them
    m = _matcher(g)
    info, score = m.match_unique(bases, qual, offsets)
    pk, nf = synth.pack_bases(bases), synth.read_nflags(bases, offsets)
    if ragged:
        pinfo, pscore = m.match_unique(pk, qual, offsets, packed=True, nflags=nf)
    else:
        pinfo, pscore = m.match_unique(pk, qual, None, patl=100, n_reads=offsets.shape[0] - 1, packed=True, nflags=nf)
    assert np.array_equal(info, pinfo) and np.array_equal(score.view(np.uint32), pscore.view(np.uint32))
    assert ((info >> np.uint64(61)) != 0).mean() > 0.5
    # the same packed bytes handed over as device memory
    import torch
    dpk, dq, doff = torch.from_numpy(pk).cuda(), torch.from_numpy(qual).cuda(), torch.from_numpy(offsets.astype(np.int64)).cuda()
    dnf = torch.from_numpy(nf).cuda()
    di = torch.zeros(info.shape[0], dtype=torch.int64, device="cuda")
    ds = torch.full((info.shape[0],), float(np.finfo(np.float32).min), dtype=torch.float32, device="cuda")
    m.match_unique(dpk, dq, doff if ragged else None, patl=0 if ragged else 100, info=di, score=ds, n_reads=info.shape[0], packed=True, nflags=dnf)
    assert np.array_equal(di.cpu().numpy().view(np.uint64), info) and np.array_equal(ds.cpu().numpy().view(np.uint32), score.view(np.uint32))
    m.close()


@pytest.mark.parametrize("packed,fresh", [(True, True), (False, False), (True, False)])
def test_submit_wait_pipeline_equals_synchronous_call(packed, fresh):
    g, bases, qual, offsets = _case(n_reads=9000, ragged=False)
    n, patl = offsets.shape[0] - 1, 100
    m = _matcher(g)
    info, score = m.match_unique(bases, qual, offsets)
    # pinned host arrays, five chunks through two slots
    src = synth.pack_bases(bases) if packed else bases
    hb = m.host_alloc(src.shape, np.uint8); hb[:] = src
    hq = m.host_alloc(qual.shape, np.uint8); hq[:] = qual
    hi = m.host_alloc((n,), np.uint64)
    hs = m.host_alloc((n,), np.float32)
    i0, s0 = new_unique_info(n, True)
    hi[:] = i0 if fresh else 123                                # (fresh: whatever is in the arrays is not uploaded)
    hs[:] = s0 if fresh else 7.0
    if not fresh:
        hi[:] = i0; hs[:] = s0
    nf = synth.read_nflags(bases, offsets) if packed else None
    chunk = 2000                                                # a multiple of 8 reads (nflags bytes) and of 4 bases (packed bytes)
    cuts = list(range(0, n, chunk))
    for k, lo in enumerate(cuts):
        hi_ = min(n, lo + chunk)
        slot = k % 2
        m.wait(slot)
        bpr = patl // 4 if packed else patl
        m.submit_unique(slot, hb[lo * bpr:hi_ * bpr], hq[lo * patl:hi_ * patl], hi[lo:hi_], hs[lo:hi_], patl=patl, n_reads=hi_ - lo,
                        packed=packed, nflags=None if nf is None else nf[lo // 8:(hi_ + 7) // 8], fresh=fresh)
    with pytest.raises(rlib.RealHipError):                      # a slot in flight refuses another batch
        m.submit_unique((len(cuts) - 1) % 2, hb[:100 * (patl // 4 if packed else patl)], hq[:100 * patl], hi[:100], hs[:100], patl=patl, n_reads=100, packed=packed)
    m.wait(0); m.wait(1)
    assert np.array_equal(np.asarray(hi), info) and np.array_equal(np.asarray(hs).view(np.uint32), score.view(np.uint32))
    m.close()


def test_version_1_batch_struct_is_still_accepted():
    class BatchV1(C.Structure):
        _fields_ = rlib.RealHipBatch._fields_[:8]
    assert C.sizeof(BatchV1) == 48
    g, bases, qual, offsets = _case(n_reads=900, ragged=False)
    m = _matcher(g)
    info, score = m.match_unique(bases, qual, offsets)
    b = BatchV1()
    b.struct_size, b.on_device, b.n_reads = 48, 0, offsets.shape[0] - 1
    b.bases, b.qual, b.offsets, b.patl, b.max_patl = bases.ctypes.data, qual.ctypes.data, offsets.ctypes.data, 0, 0
    i1, s1 = new_unique_info(int(b.n_reads), True)
    rc = m._L.real_hip_match_unique(m._h, C.cast(C.byref(b), C.POINTER(rlib.RealHipBatch)), i1.ctypes.data, s1.ctypes.data)
    assert rc == 0 and np.array_equal(i1, info) and np.array_equal(s1.view(np.uint32), score.view(np.uint32))
    m.close()


def test_under_declared_read_length_is_harmless():
    """device offsets with a caller-supplied max_patl that is too small: the reads beyond it are not staged by the
    lane-per-read kernel (nothing is written past a wave's LDS region) but handed to the wave-per-read matcher --
    same records, only slower"""
    import torch
    g, bases, qual, offsets = _case(n_reads=3000, ragged=True)
    m = _matcher(g)
    db, dq, doff = torch.from_numpy(bases).cuda(), torch.from_numpy(qual).cuda(), torch.from_numpy(offsets.astype(np.int64)).cuda()
    n = offsets.shape[0] - 1
    ref_i, ref_s = m.match_unique(bases, qual, offsets)
    for declared in (64, 100):
        di = torch.zeros(n, dtype=torch.int64, device="cuda")
        ds = torch.full((n,), float(np.finfo(np.float32).min), dtype=torch.float32, device="cuda")
        m.counters(reset=True)
        m.match_unique(db, dq, doff, info=di, score=ds, max_patl=declared)        # the batch holds 36, 77 and 100 bp reads
        assert np.array_equal(di.cpu().numpy().view(np.uint64), ref_i) and np.array_equal(ds.cpu().numpy().view(np.uint32), ref_s.view(np.uint32))
        assert (m.counters()["handed_over"] > 1000) == (declared == 64)
    m.close()


@pytest.mark.parametrize("packed,scores", [(False, 1), (True, 1), (False, 0)])
def test_reads_longer_than_the_registers_hold(ora, packed, scores):
    """REAL_HIP_MAX_PATL (320) is the longest read a lane keeps in registers, not a limit of the library: longer reads
    (up to REAL_HIP_MAX_PATL_LONG) get a wave each and are read from LDS words.  A ragged batch of 60 ... 5000 bp reads,
    some with an N, against the oracle; then a batch of one long length."""
    g = synth.random_genome(400_000, seed=31, n_frag=2, n_runs=3, repeats=6)
    # (mismatches only, k <= 5: long reads get few errors, or nothing of them would align)
    parts = [synth.sample_reads(g, k, pl, 1.5 / pl, seed=300 + pl, n_read_prob=0.02 / pl) for k, pl in ((600, 100), (150, 400), (60, 321), (40, 1000), (12, 5000), (300, 60), (100, 300))]
    rng = np.random.default_rng(9)
    order = rng.permutation(sum(p.n_reads for p in parts))
    reads = [(p.bases[int(p.offsets[i]):int(p.offsets[i + 1])], p.qual[int(p.offsets[i]):int(p.offsets[i + 1])]) for p in parts for i in range(p.n_reads)]
    reads = [reads[i] for i in order]
    bases = np.concatenate([r[0] for r in reads]); qual = np.concatenate([r[1] for r in reads])
    offsets = np.concatenate([[0], np.cumsum([r[0].shape[0] for r in reads])]).astype(np.uint64)
    assert (bases > 3).any()
    og = ora.Genome(g.sym, g.frag_start)
    ix = ora.Index(og, 32)
    p = ora.make_params(seedl=32, seedkmax=2, totalkmax=5, scores=scores)
    oinfo, oscore, octr = ora.match_unique(og, ix, p, bases, qual, offsets)
    m = UniqueMatcher(RealOptions(seedl=32, seedkmax=2, totalkmax=5, scores=bool(scores)).normalise())
    m.set_text_symbols(0, g.sym, g.frag_start)
    m.build_index_block()
    m.counters(reset=True)
    if packed:
        info, score = m.match_unique(synth.pack_bases(bases), qual, offsets, packed=True, nflags=synth.read_nflags(bases, offsets))
    else:
        info, score = m.match_unique(bases, qual, offsets)
    assert np.array_equal(info, oinfo)
    if scores:
        assert np.array_equal(score.view(np.uint32), oscore.view(np.uint32))
    c = m.counters()
    for k in ("reads", "lookups", "candidates", "seedpass", "hits"):
        assert c[k] == octr[k], (k, c[k], octr[k])
    assert c["handed_over"] >= 150 + 60 + 40 + 12 - 10          # every long read (minus the ones with an N)
    st = (info >> np.uint64(61)).astype(int)
    lens = np.diff(offsets.astype(np.int64))
    assert ((st[lens >= 1000] == 1) | (st[lens >= 1000] == 2)).mean() > 0.6      # long reads do align
    # matchAll on the same batch
    hits, hoff = m.match_all(bases, qual, offsets)
    ohits, ooff, _ = ora.match_all(og, ix, p, bases, qual, offsets)
    assert np.array_equal(hoff, ooff) and np.array_equal(hits["pos"], ohits["pos"]) and np.array_equal(hits["k"], ohits["k"])
    # one long length for the whole batch: every read goes to the wave-per-read matcher
    b = synth.sample_reads(g, 300, 420, 0.005, seed=77)
    oi, os_, _ = ora.match_unique(og, ix, p, b.bases, b.qual, b.offsets)
    ui, us = m.match_unique(b.bases, b.qual, None, patl=420, n_reads=300)
    assert np.array_equal(ui, oi) and (not scores or np.array_equal(us.view(np.uint32), os_.view(np.uint32)))
    # beyond REAL_HIP_MAX_PATL_LONG: a loud error
    with pytest.raises(rlib.RealHipError) as e:
        m.match_unique(np.zeros(20000, np.uint8), np.zeros(20000, np.uint8), np.array([0, 20000], np.uint64))
    assert e.value.status == rlib.REAL_HIP_E_UNSUPPORTED
    m.close()


def test_c_abi_rccl_gather_single_rank():
    """real_hip_comm_* / real_hip_gather_*: the C-side form of the path's one collective (RCCL, one process per GPU),
    exercised here with a communicator of one rank -- counts exchange, grouped send/recv to self, rebasing kernels.
    (More ranks need more GPUs than this box has; the N-rank bench path runs over torch.distributed.)"""
    import torch
    from real_amd.matcher import AllMatcher, HipMatcher
    g, bases, qual, offsets = _case(n_reads=3000, ragged=False)
    n = offsets.shape[0] - 1
    m = _matcher(g)
    db, dq = torch.from_numpy(bases).cuda(), torch.from_numpy(qual).cuda()
    di = torch.zeros(n, dtype=torch.int64, device="cuda")
    ds = torch.full((n,), float(np.finfo(np.float32).min), dtype=torch.float32, device="cuda")
    m.match_unique(db, dq, patl=100, info=di, score=ds, n_reads=n)
    m.comm_init(HipMatcher.comm_id(), 0, 1)
    ai, as_ = torch.zeros(n + 5, dtype=torch.int64, device="cuda"), torch.zeros(n + 5, dtype=torch.float32, device="cuda")
    assert m.gather_records(0, di, ds, ai, as_) == n
    assert torch.equal(ai[:n], di) and torch.equal(as_[:n].view(torch.int32), ds.view(torch.int32))
    with pytest.raises(rlib.RealHipError) as e:                                 # a receive array that is too small: loud, on every rank
        m.gather_records(0, di, ds, ai[:10], as_[:10])
    assert e.value.status == rlib.REAL_HIP_E_OVERFLOW
    # null receive arrays on the root / a null send array: decided from the exchanged tuples (the same verdict on every rank,
    # before anyone sends -- ADVICE r2: a root that returned alone left its peers in ncclSend)
    import ctypes as C
    n_all = C.c_uint64(0)
    rc = m._L.real_hip_gather_records(m._h, 0, di.data_ptr(), ds.data_ptr(), n, None, None, n, C.byref(n_all))
    assert rc == rlib.REAL_HIP_E_INVALID and b"every rank" in m._L.real_hip_last_error(m._h)
    rc = m._L.real_hip_gather_records(m._h, 0, None, ds.data_ptr(), n, ai.data_ptr(), as_.data_ptr(), n + 5, C.byref(n_all))
    assert rc == rlib.REAL_HIP_E_INVALID
    assert m.gather_records(0, di, ds, ai, as_) == n                             # (and the communicator is still usable)
    # matchAll hit lists
    m.set_match_params(totalkmax=2)
    import ctypes as C
    cap = 8 * n
    hits = torch.zeros((cap, 4), dtype=torch.int32, device="cuda")
    hoff = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    b = m._batch(db, dq, None, 100, n)
    nout = C.c_uint64(0)
    m.sync_inputs(db)
    m._check(m._L.real_hip_match_all(m._h, C.byref(b), hits.data_ptr(), cap, C.byref(nout), hoff.data_ptr()))
    nh = int(nout.value)
    assert nh > n // 2
    ah = torch.zeros((nh + 7, 4), dtype=torch.int32, device="cuda")
    ao = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    assert m.gather_hits(0, hits, hoff, nh, ah, ao) == (n, nh)
    assert torch.equal(ah[:nh], hits[:nh]) and torch.equal(ao, hoff)
    nr_, nh_ = C.c_uint64(0), C.c_uint64(0)
    rc = m._L.real_hip_gather_hits(m._h, 0, hits.data_ptr(), hoff.data_ptr(), n, nh, ah.data_ptr(), nh + 7, None, n, C.byref(nr_), C.byref(nh_))
    assert rc == rlib.REAL_HIP_E_INVALID and b"every rank" in m._L.real_hip_last_error(m._h)
    assert m.gather_hits(0, hits, hoff, nh, ah, ao) == (n, nh)
    m.close()
