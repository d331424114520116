"""Host-side mirror of the reference's interface for the read-matching path.

Names follow the reference: ``RealOptions`` (RealOptions.hpp:26-78, parser
RealOptions.cpp:122-466), ``UniqueMatcher.match`` / ``AllMatcher.match``
(matchUniqueImplementation.cpp:369-500, matchAllImplementation.cpp:261-355) --
here they take a whole decoded pattern block instead of one pattern, because the
device boundary sits at "for z in block: UM.match(...)" (SURVEY 3.1).

Everything below calls the C ABI of include/real_hip.h; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lib as _lib
from .lib import (HIT_DTYPE, RealHipBatch, RealHipCounters, RealHipError, RealHipParams, _ptr)

NO_SCORE = np.float32(-np.finfo(np.float32).max)   # UniqueMatchInfo<true>() : score(-FLT_MAX), UniqueMatchInfo.hpp:191

# UniqueMatchInfoBase::MatchState, UniqueMatchInfo.hpp:71-78
NoMatch, Straight, Reverse, Gapped, NonUnique = 0, 1, 2, 3, 4


@dataclass
class RealOptions:
    """RealOptions.hpp:27-72; defaults :27-36."""
    textfilename: str = ""
    patternfilename: str = ""
    outputfilename: str = ""
    seedkmax: int = 2
    totalkmax: int = 5
    seedl: int = 32
    match_unique: bool = True
    fracmem: float = 0.75
    scores: bool = True
    qualityOffset: int = 0
    rewritepatterns: bool = True
    sort_threads: int = 2
    filter_level: int = 2
    similarity: float = 0.995
    err: float = 0.0
    trans: float = 0.71
    gc: float = 0.41
    gcmut_bias: float = 2.0
    gaps: bool = False

    def normalise(self) -> "RealOptions":
        """The clamps of RealOptions.cpp:172-180, 434-453."""
        if self.totalkmax > 15:
            self.totalkmax = 15
        if self.seedl > 64:
            self.seedl = 64
        if self.seedl % 4:
            self.seedl -= self.seedl % 4
        if self.seedl < 4:
            raise ValueError("cannot handle seed length < 4")
        if self.seedkmax > 2:
            self.seedkmax = 2
        return self

    @property
    def filter_mult(self) -> float:
        """RealOptions.cpp:455-463."""
        mult = {1: 0.5, 2: 1.0, 3: 2.0, 4: 3.0}.get(self.filter_level, 0.0) * self.totalkmax
        return mult / 70.0

    def getFilterValue(self, patl: int) -> float:
        """RealOptions.hpp:74-77."""
        return self.filter_mult * patl

    @classmethod
    def parse(cls, argv: Sequence[str]) -> "RealOptions":
        """The hand-rolled argv loop of RealOptions.cpp:140-396 (unknown arguments are ignored)."""
        o = cls()
        i = 0
        table = {"-t": ("textfilename", str), "-p": ("patternfilename", str), "-o": ("outputfilename", str),
                 "-s": ("seedkmax", int), "-e": ("totalkmax", int), "-l": ("seedl", int),
                 "-u": ("match_unique", lambda v: bool(int(v))), "-f": ("fracmem", float), "-m": ("fracmem", float),
                 "-q": ("scores", lambda v: bool(int(v))), "-Q": ("qualityOffset", int),
                 "-R": ("rewritepatterns", lambda v: bool(int(v))), "-T": ("sort_threads", int),
                 "-g": ("gaps", lambda v: bool(int(v))), "-similarity": ("similarity", float), "-err": ("err", float),
                 "-trans": ("trans", float), "-gc": ("gc", float), "-gcmut_bias": ("gcmut_bias", float),
                 "-filter_level": ("filter_level", int)}
        argv = list(argv)
        while i < len(argv):
            a = argv[i]
            if a in table:
                if i + 1 >= len(argv):
                    raise ValueError("Parameter for argument %s is missing." % a)
                name, conv = table[a]
                setattr(o, name, conv(argv[i + 1]))
                i += 2
            else:
                i += 1
        return o.normalise()


def new_unique_info(n: int, scores: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """AutoArray<UniqueMatchInfo<scores>> uniqueinfo(numpat), matchUniqueImplementation.cpp:1094-1097."""
    info = np.zeros(n, dtype=np.uint64)
    score = np.full(n, NO_SCORE, dtype=np.float32) if scores else None
    return info, score


def unpack_info(info: np.ndarray):
    """UniqueMatchInfo.hpp:29-39 -> state, fragment, errors, fileid, position."""
    rec = np.asarray(info, dtype=np.uint64)
    state = np.minimum(rec >> np.uint64(61), np.uint64(4)).astype(np.int64)
    frag = ((rec >> np.uint64(45)) & np.uint64(0xffff)).astype(np.int64)
    err = ((rec >> np.uint64(41)) & np.uint64(15)).astype(np.int64)
    fid = ((rec >> np.uint64(35)) & np.uint64(63)).astype(np.int64)
    pos = (rec & np.uint64((1 << 35) - 1)).astype(np.int64)
    return state, frag, err, fid, pos


class HipMatcher:
    """One context on one MI355X: resident text + index block + matching calls."""

    def __init__(self, opts: RealOptions, device: int = 0, prefix_bits: int = 0, LL: Optional[np.ndarray] = None,
                 table_kind: int = 0):
        self.opts = opts
        self._L = _lib.load()
        p = RealHipParams()
        p.struct_size = C.sizeof(RealHipParams)
        p.seedl, p.seedkmax, p.totalkmax = opts.seedl, opts.seedkmax, opts.totalkmax
        p.scores = int(bool(opts.scores))
        p.prefix_bits = prefix_bits
        p.table_kind = table_kind          # 0 auto, 1 bucket starts only, 2 directory entries (real_hip.h)
        p.device = device
        p.filter_mult = opts.filter_mult
        if LL is None:
            LL = _lib.scoring_table(opts.similarity, opts.gc, opts.trans, opts.err, opts.gcmut_bias)
        self.LL = np.ascontiguousarray(LL, dtype=np.float64)
        for i in range(1024):
            p.LL[i] = float(self.LL[i])
        h = C.c_void_p()
        rc = self._L.real_hip_create(C.byref(h), C.byref(p))
        if rc != 0:
            raise RealHipError(rc, self._L.real_hip_strerror(rc).decode() +
                               " (real_hip_create: is an MI355X visible? there is no CPU fallback)")
        self._h = h
        self.device = device
        self.n_entries = 0
        self.prefix_bits = 0

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            self._L.real_hip_destroy(self._h)
            self._h = None
        for p in getattr(self, "_pinned", []):
            self._L.real_hip_host_free(p)
        self._pinned = []

    def __del__(self):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise RealHipError(rc, self._L.real_hip_last_error(self._h).decode() or self._L.real_hip_strerror(rc).decode())

    def set_match_params(self, seedkmax: Optional[int] = None, totalkmax: Optional[int] = None, scores: Optional[bool] = None,
                         filter_level: Optional[int] = None):
        """-s / -e / -q / -filter_level for the calls that follow; the resident text and index stay (they depend on -l only)."""
        o = self.opts
        if seedkmax is not None: o.seedkmax = seedkmax
        if totalkmax is not None: o.totalkmax = totalkmax
        if scores is not None: o.scores = bool(scores)
        if filter_level is not None: o.filter_level = filter_level
        o.normalise()
        self._check(self._L.real_hip_set_match_params(self._h, o.seedkmax, o.totalkmax, int(bool(o.scores)), o.filter_mult))

    # -- text --
    def set_text(self, fileid: int, text2bit: np.ndarray, wildbits: np.ndarray, n_bases: int, frag_start: np.ndarray):
        t = np.ascontiguousarray(text2bit, dtype=np.uint64)
        w = np.ascontiguousarray(wildbits, dtype=np.uint64)
        f = np.ascontiguousarray(frag_start, dtype=np.uint64)
        self._check(self._L.real_hip_set_text(self._h, fileid, t.ctypes.data, w.ctypes.data, n_bases, f.ctypes.data, f.shape[0] - 1))

    def set_text_symbols(self, fileid: int, sym, frag_start: np.ndarray, n_bases: Optional[int] = None):
        """sym: numpy uint8 (host) or a torch uint8 tensor (host or device)."""
        f = np.ascontiguousarray(frag_start, dtype=np.uint64)
        on_device = bool(getattr(sym, "is_cuda", False))
        self.sync_inputs(sym)
        if isinstance(sym, np.ndarray):
            sym = np.ascontiguousarray(sym, dtype=np.uint8)
        n = int(n_bases if n_bases is not None else sym.shape[0])
        self._check(self._L.real_hip_set_text_symbols(self._h, fileid, _ptr(sym), n, int(on_device), f.ctypes.data, f.shape[0] - 1))

    # -- index --
    def set_index_block(self, sign: Sequence[np.ndarray], pos: Sequence[np.ndarray]):
        """Host-built sorted lists (ListSetBlockReader::readNextBlock, ListSetBlockReader.hpp:24-52)."""
        sdt = np.uint32 if self.opts.seedl <= 32 else np.uint64
        sg = [np.ascontiguousarray(s, dtype=sdt) for s in sign]
        ps = [np.ascontiguousarray(p, dtype=np.uint32) for p in pos]
        n = int(sg[0].shape[0])
        sa = (C.c_void_p * 6)(*[s.ctypes.data for s in sg])
        pa = (C.c_void_p * 6)(*[p.ctypes.data for p in ps])
        self._check(self._L.real_hip_set_index_block(self._h, n, sa, pa))
        self._refresh_index_info()

    def build_index_block(self, first_window: int = 0, max_entries: int = (1 << 62)) -> Tuple[int, bool]:
        n = C.c_uint64(0)
        nxt = C.c_int(0)
        self._check(self._L.real_hip_build_index_block(self._h, first_window, max_entries, C.byref(n), C.byref(nxt)))
        self._refresh_index_info()
        return int(n.value), bool(nxt.value)

    def index_build_stats(self, reset: bool = False) -> dict:
        """where the wall time of this context's index builds went (real_hip_index_build_stats)"""
        st = _lib.RealHipBuildStats()
        st.struct_size = C.sizeof(_lib.RealHipBuildStats)
        self._check(self._L.real_hip_index_build_stats(self._h, C.byref(st), int(reset)))
        return {k: (float(getattr(st, k)) if k.endswith("_ms") else int(getattr(st, k))) for k, _ in st._fields_ if k not in ("struct_size", "reserved")}

    def _refresh_index_info(self):
        n = C.c_uint64(0)
        pb = C.c_uint32(0)
        self._check(self._L.real_hip_index_info(self._h, C.byref(n), C.byref(pb)))
        self.n_entries, self.prefix_bits = int(n.value), int(pb.value)
        tk = C.c_uint32(0)
        self._check(self._L.real_hip_index_table_kind(self._h, C.byref(tk)))
        self.table_kind = int(tk.value)

    def index_download(self, k: int, want_buckets: bool = True):
        """device layout of list k: entries (n x {fingerprint, pos}) and bucket starts."""
        ent = np.zeros((self.n_entries, 2), dtype=np.uint32)
        bkt = np.zeros((1 << self.prefix_bits) + 1, dtype=np.uint32) if want_buckets else None
        self._check(self._L.real_hip_index_download(self._h, k, ent.ctypes.data, _ptr(bkt)))
        return ent, bkt

    def index_export(self, k: int):
        """list k in the reference's form: (sign[n], pos[n]) of the sorted Mask entries."""
        sdt = np.uint32 if self.opts.seedl <= 32 else np.uint64
        sign = np.zeros(self.n_entries, dtype=sdt)
        pos = np.zeros(self.n_entries, dtype=np.uint32)
        self._check(self._L.real_hip_index_export(self._h, k, sign.ctypes.data, pos.ctypes.data))
        return sign, pos

    # -- batches --
    @staticmethod
    def sync_inputs(*arrays):
        """The library runs on the context's own stream and does not know the caller's: device arrays handed over
        (inputs AND in/out records) must be complete before the call (include/real_hip.h, "Streams").  For torch
        tensors that means the producing stream has to drain -- done here, once per device."""
        seen = set()
        for a in arrays:
            if a is not None and getattr(a, "is_cuda", False) and a.device not in seen:
                import torch
                torch.cuda.current_stream(a.device).synchronize()
                seen.add(a.device)

    @staticmethod
    def _batch(bases, qual, offsets, patl: int, n_reads: Optional[int], max_patl: int = 0) -> RealHipBatch:
        b = RealHipBatch()
        b.struct_size = C.sizeof(RealHipBatch)
        on_device = bool(getattr(bases, "is_cuda", False))
        b.on_device = int(on_device)
        if offsets is not None:
            b.n_reads = int(offsets.shape[0]) - 1
        else:
            b.n_reads = int(n_reads if n_reads is not None else bases.shape[0] // patl)
        b.bases, b.qual, b.offsets = _ptr(bases), _ptr(qual), _ptr(offsets)
        b.patl, b.max_patl = int(patl), int(max_patl)
        return b

    def match_unique(self, bases, qual, offsets=None, patl: int = 0, info=None, score=None,
                     n_reads: Optional[int] = None, max_patl: int = 0, packed: bool = False, nflags=None,
                     fresh: bool = False):
        """UniqueMatcher::match over a pattern block, folding into info/score in place.
        Host numpy arrays or device torch tensors (all of one kind).  fresh: info / score are outputs only, every
        record starts as uniqueinfo(numpat) does (the first genome block of a run)."""
        if isinstance(bases, np.ndarray):
            bases = np.ascontiguousarray(bases, dtype=np.uint8)
            qual = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
            offsets = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        if packed and offsets is None and n_reads is None:
            raise ValueError("a packed batch of uniform length needs n_reads")
        b = self._batch(bases, qual, offsets, patl, n_reads, max_patl)
        b.packed = int(bool(packed))
        b.nflags = _ptr(nflags)
        b.fresh = int(bool(fresh))
        if info is None:
            info, score = new_unique_info(int(b.n_reads), self.opts.scores)
        self.sync_inputs(bases, qual, offsets, info, score)
        self._check(self._L.real_hip_match_unique(self._h, C.byref(b), _ptr(info), _ptr(score)))
        return info, score

    def match_all(self, bases, qual, offsets=None, patl: int = 0, n_reads: Optional[int] = None, cap: int = 0):
        """AllMatcher::match + unifyMatches over a host pattern block -> (hits, hit_offsets)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        qual = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
        offsets = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        b = self._batch(bases, qual, offsets, patl, n_reads)
        n = int(b.n_reads)
        cap = cap or max(1024, 4 * n)
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            hoff = np.zeros(n + 1, dtype=np.uint64)
            nout = C.c_uint64(0)
            rc = self._L.real_hip_match_all(self._h, C.byref(b), out.ctypes.data, cap, C.byref(nout), hoff.ctypes.data)
            if rc == _lib.REAL_HIP_E_OVERFLOW:      # caller retries with the size the library reports
                cap = int(nout.value)
                continue
            self._check(rc)
            return out[:int(nout.value)], hoff

    # -- pipelined host batches (submit / wait over two slots) --
    def host_alloc(self, shape, dtype) -> np.ndarray:
        """a numpy array over pinned host memory (real_hip_host_alloc): batches handed over from it cross PCIe by DMA
        without a staging copy, which is what makes submit asynchronous.  Freed with the matcher."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = self._L.real_hip_host_alloc(max(1, n * dt.itemsize))
        if not p:
            raise RealHipError(_lib.REAL_HIP_E_NOMEM, "real_hip_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_uint8 * (n * dt.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def submit_unique(self, slot: int, bases, qual, info, score, patl: int = 0, offsets=None, n_reads: Optional[int] = None,
                      packed: bool = False, nflags=None, fresh: bool = False):
        """real_hip_match_unique_submit: queue upload, kernels and download of one host batch; returns at once.  The
        arrays must stay alive and untouched until wait(slot)."""
        b = self._batch(bases, qual, offsets, patl, n_reads)
        b.on_device = 0
        b.packed = int(bool(packed))
        b.nflags = _ptr(nflags)
        self._check(self._L.real_hip_match_unique_submit(self._h, C.byref(b), _ptr(info), _ptr(score), slot, int(bool(fresh))))

    def wait(self, slot: int):
        self._check(self._L.real_hip_wait(self._h, slot))

    # -- multi-GPU: the C ABI's own RCCL gather (one process per GPU; torch.distributed is the other way, real_amd.distributed) --
    @staticmethod
    def comm_id() -> bytes:
        """rank 0: the 128 bytes every rank hands to comm_init (broadcast them by the launcher's own means)"""
        buf = (C.c_uint8 * 128)()
        rc = _lib.load().real_hip_comm_id(buf)
        if rc != 0:
            raise RealHipError(rc, "real_hip_comm_id (librccl could not be loaded?)")
        return bytes(buf)

    def comm_init(self, comm_id: bytes, rank: int, n_ranks: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(comm_id)
        self._check(self._L.real_hip_comm_init(self._h, buf, rank, n_ranks))

    def gather_records(self, root: int, info, score, info_all=None, score_all=None) -> int:
        """device tensors; returns the number of records on the root (real_hip_gather_records)"""
        self.sync_inputs(info, score, info_all, score_all)
        n_all = C.c_uint64(0)
        cap = int(info_all.shape[0]) if info_all is not None else 0
        self._check(self._L.real_hip_gather_records(self._h, root, _ptr(info), _ptr(score), int(info.shape[0]), _ptr(info_all), _ptr(score_all), cap, C.byref(n_all)))
        return int(n_all.value)

    def gather_hits(self, root: int, hits, hit_offsets, n_hits: int, hits_all=None, offsets_all=None):
        """device tensors: hits [n, 4] int32 (real_hip_hit records), hit_offsets [n_local + 1] int64; returns (reads, hits) on the root"""
        self.sync_inputs(hits, hit_offsets, hits_all, offsets_all)
        nr, nh = C.c_uint64(0), C.c_uint64(0)
        cap_h = int(hits_all.shape[0]) if hits_all is not None else 0
        cap_r = int(offsets_all.shape[0]) - 1 if offsets_all is not None else 0
        self._check(self._L.real_hip_gather_hits(self._h, root, _ptr(hits), _ptr(hit_offsets), int(hit_offsets.shape[0]) - 1, int(n_hits), _ptr(hits_all), cap_h,
                                                 _ptr(offsets_all), cap_r, C.byref(nr), C.byref(nh)))
        return int(nr.value), int(nh.value)

    # -- read ingestion on the device --
    def parse_reads(self, text, fastq: bool, quality_offset: int = 33):
        """FASTA / FASTQ text (bytes, numpy uint8 or a device torch tensor) -> RealHipParsed: device arrays owned by the
        context, valid until the next parse.  Raises RealHipError(E_UNSUPPORTED) for text that is not in
        one-line-per-field form (FastQReader.hpp:130-180 accepts more; the host reader handles that)."""
        on_dev = bool(getattr(text, "is_cuda", False))
        if not on_dev and not isinstance(text, np.ndarray):
            text = np.frombuffer(text, dtype=np.uint8)
        n = int(text.numel()) if on_dev else int(text.shape[0])
        self.sync_inputs(text)
        out = _lib.RealHipParsed()
        self._check(self._L.real_hip_parse_reads(self._h, _ptr(text), n, int(on_dev), int(bool(fastq)), int(quality_offset), C.byref(out)))
        return out

    def match_unique_parsed(self, parsed, info=None, score=None):
        """UniqueMatcher::match over the reads of a parse_reads() result (read arrays on the device, records on the host)."""
        b = RealHipBatch()
        b.struct_size = C.sizeof(RealHipBatch)
        b.on_device = 2
        b.n_reads = parsed.n_reads
        b.bases, b.qual, b.offsets = parsed.bases, parsed.qual, parsed.offsets
        b.patl, b.max_patl = 0, parsed.max_patl
        if info is None:
            info, score = new_unique_info(int(b.n_reads), self.opts.scores)
        self._check(self._L.real_hip_match_unique(self._h, C.byref(b), _ptr(info), _ptr(score)))
        return info, score

    def download(self, dev_ptr, count: int, dtype):
        """copy `count` items of a device array the library returned to the host (tests, id strings)"""
        out = np.zeros(count, dtype=dtype)
        if count:
            self._check(self._L.real_hip_download(self._h, C.c_void_p(int(dev_ptr)), out.ctypes.data, out.nbytes))
        return out

    # -- instrumentation --
    def counters(self, reset: bool = False) -> dict:
        c = RealHipCounters()
        self._check(self._L.real_hip_counters_get(self._h, C.byref(c), int(reset)))
        return c.as_dict()

    def kernel_time(self, which: int, reset: bool = False) -> Tuple[float, int]:
        ms = C.c_double(0)
        n = C.c_uint64(0)
        self._check(self._L.real_hip_kernel_time(self._h, which, C.byref(ms), C.byref(n), int(reset)))
        return float(ms.value), int(n.value)

    def timing_enable(self, on: bool):
        self._check(self._L.real_hip_timing_enable(self._h, int(on)))


# The reference's names for the two per-read matchers; the policy (fold vs. collect) is the
# only difference, exactly as UpdateUniqueInfo / VectorUpdater are for ::match.
class UniqueMatcher(HipMatcher):
    def match(self, bases, qual, offsets=None, patl: int = 0, info=None, score=None, **kw):
        return self.match_unique(bases, qual, offsets, patl, info, score, **kw)


class AllMatcher(HipMatcher):
    def match(self, bases, qual, offsets=None, patl: int = 0, **kw):
        return self.match_all(bases, qual, offsets, patl, **kw)
