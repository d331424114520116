"""ctypes wrapper of oracle/libreal_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (the oracle is the checker, never the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libreal_oracle.so")


class OraParams(C.Structure):
    _fields_ = [("seedl", C.c_uint32), ("seedkmax", C.c_uint32), ("totalkmax", C.c_uint32),
                ("scores", C.c_uint32), ("fileid", C.c_uint32), ("threads", C.c_uint32),
                ("filter_mult", C.c_double), ("LL", C.c_double * 1024)]


class OraCounters(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("lookups", C.c_uint64), ("probes", C.c_uint64),
                ("candidates", C.c_uint64), ("seedpass", C.c_uint64), ("hits", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


EVENT_DTYPE = np.dtype([("read", "<u8"), ("pos", "<u4"), ("frag", "<u4"), ("score", "<f4"),
                        ("inverted", "u1"), ("list", "u1"), ("totalk", "u1"), ("seedk", "u1")])
HIT_DTYPE = np.dtype([("read", "<u8"), ("pos", "<u4"), ("frag", "<u4"), ("score", "<f4"),
                      ("inverted", "u1"), ("k", "u1"), ("fileid", "<u2")])
assert EVENT_DTYPE.itemsize == 24 and HIT_DTYPE.itemsize == 24


class OraIndexStruct(C.Structure):
    _fields_ = [("seedl", C.c_uint), ("sig_bits", C.c_uint), ("shift", C.c_uint), ("n", C.c_uint64),
                ("first_window", C.c_uint64), ("have_next", C.c_int),
                ("sign", C.POINTER(C.c_uint64) * 6), ("ptr", C.POINTER(C.c_uint32) * 6),
                ("pos", C.POINTER(C.c_uint32) * 6), ("lookup", C.POINTER(C.c_uint64) * 6),
                ("csign", C.POINTER(C.c_uint32) * 6), ("cpos", C.POINTER(C.c_uint32) * 6), ("compact", C.c_int)]


class OraGenomeStruct(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_words", C.c_uint64), ("text", C.POINTER(C.c_uint64)),
                ("n_wwords", C.c_uint64), ("wild", C.POINTER(C.c_uint64)),
                ("wild_S", C.POINTER(C.c_uint64)), ("wild_M", C.POINTER(C.c_uint16)),
                ("n_wild", C.c_uint64), ("n_frag", C.c_uint32), ("frag_start", C.POINTER(C.c_uint64)),
                ("n_fwords", C.c_uint64), ("fbits", C.POINTER(C.c_uint64)),
                ("frag_S", C.POINTER(C.c_uint64)), ("frag_M", C.POINTER(C.c_uint16))]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc); building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(ORACLE_DIR, "real_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libreal_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.ora_genome_create.restype = C.POINTER(OraGenomeStruct)
        L.ora_genome_create.argtypes = [vp, u64, vp, u32]
        L.ora_genome_free.argtypes = [C.POINTER(OraGenomeStruct)]
        L.ora_index_build.restype = C.POINTER(OraIndexStruct)
        L.ora_index_build.argtypes = [C.POINTER(OraGenomeStruct), C.c_uint, u64, u64]
        L.ora_index_free.argtypes = [C.POINTER(OraIndexStruct)]
        L.ora_index_from_entries.restype = C.POINTER(OraIndexStruct)
        L.ora_index_from_entries.argtypes = [C.POINTER(OraGenomeStruct), C.c_uint, u64, C.POINTER(vp), C.POINTER(vp)]
        L.ora_index_getpos.restype = u32
        L.ora_index_getpos.argtypes = [C.POINTER(OraIndexStruct), C.c_int, u64]
        L.ora_get_text_word.restype = u64
        L.ora_get_text_word.argtypes = [C.POINTER(OraGenomeStruct), u64, C.c_uint]
        L.ora_signature_mapped.argtypes = [C.c_uint, vp, vp]
        L.ora_reverse_mapped_signature.argtypes = [C.c_uint, vp, vp]
        L.ora_signatures.argtypes = [C.c_uint, vp, vp]
        L.ora_scoring_table.argtypes = [C.c_double] * 5 + [vp, vp]
        L.ora_compute_score.restype = C.c_float
        L.ora_compute_score.argtypes = [C.POINTER(OraGenomeStruct), vp, C.c_int, vp, vp, u32, C.c_uint]
        L.ora_record_pack.restype = u64
        L.ora_record_pack.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, u64]
        L.ora_match_unique.argtypes = [C.POINTER(OraGenomeStruct), C.POINTER(OraIndexStruct), C.POINTER(OraParams),
                                       vp, vp, vp, u64, vp, vp, C.POINTER(OraCounters), vp, u64, C.POINTER(u64)]
        L.ora_match_events.argtypes = [C.POINTER(OraGenomeStruct), C.POINTER(OraIndexStruct), C.POINTER(OraParams),
                                       vp, vp, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(OraCounters)]
        L.ora_match_all.argtypes = [C.POINTER(OraGenomeStruct), C.POINTER(OraIndexStruct), C.POINTER(OraParams),
                                    vp, vp, vp, u64, vp, u64, C.POINTER(u64), vp, C.POINTER(OraCounters)]
        L.ora_update_unique.argtypes = [C.c_int, C.c_int, C.c_uint, u32, C.c_uint, C.c_float, C.c_float, C.c_uint, vp, vp]
        L.ora_diffcountpair32.restype = C.c_uint
        L.ora_diffcountpair32.argtypes = [u32, u32]
        L.ora_diffcountpair64.restype = C.c_uint
        L.ora_diffcountpair64.argtypes = [u64, u64]
        L.ora_is_position_valid.argtypes = [C.POINTER(OraGenomeStruct), u64, C.c_uint]
        L.ora_is_dontcare_free.argtypes = [C.POINTER(OraGenomeStruct), u64, C.c_uint]
        L.ora_position_to_range.restype = C.c_uint
        L.ora_position_to_range.argtypes = [C.POINTER(OraGenomeStruct), u64]
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def scoring_table(similarity=0.995, gc=0.41, trans=0.71, err=0.0, gcmut_bias=2.0) -> Tuple[np.ndarray, np.ndarray]:
    LL = np.zeros(1024, dtype=np.float64)
    odds = np.zeros(16, dtype=np.float64)
    lib().ora_scoring_table(similarity, gc, trans, err, gcmut_bias, _p(LL), _p(odds))
    return LL, odds.reshape(4, 4)


def filter_mult(filter_level: int, totalkmax: int) -> float:
    """RealOptions.cpp:455-463."""
    mult = {1: 0.5, 2: 1.0, 3: 2.0, 4: 3.0}.get(filter_level, 0.0) * totalkmax
    return mult / 70.0


class Genome:
    def __init__(self, sym: np.ndarray, frag_start: np.ndarray):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        fs = np.ascontiguousarray(frag_start, dtype=np.uint64)
        self.h = lib().ora_genome_create(_p(sym), sym.shape[0], _p(fs), fs.shape[0] - 1)
        self.n = int(sym.shape[0])

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_genome_free(self.h)
            self.h = None

    @property
    def text(self) -> np.ndarray:
        s = self.h.contents
        return np.ctypeslib.as_array(s.text, shape=(int(s.n_words),)).copy()

    @property
    def wild(self) -> np.ndarray:
        s = self.h.contents
        return np.ctypeslib.as_array(s.wild, shape=(int(s.n_wwords),)).copy()

    @property
    def n_wild(self) -> int:
        return int(self.h.contents.n_wild)


class Index:
    def __init__(self, g: Genome, seedl: int, first_window: int = 0, max_entries: int = (1 << 62)):
        self.g = g
        self.h = lib().ora_index_build(g.h, seedl, first_window, max_entries)
        if not self.h:
            raise ValueError("bad seed length %d" % seedl)
        self.n = int(self.h.contents.n)
        self.have_next = bool(self.h.contents.have_next)
        self.seedl = seedl

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_index_free(self.h)
            self.h = None

    def sign(self, k: int) -> np.ndarray:
        return np.ctypeslib.as_array(self.h.contents.sign[k], shape=(self.n,)).copy() if self.n else np.zeros(0, np.uint64)

    def ptr(self, k: int) -> np.ndarray:
        return np.ctypeslib.as_array(self.h.contents.ptr[k], shape=(self.n,)).copy() if self.n else np.zeros(0, np.uint32)

    def pos(self, k: int) -> np.ndarray:
        """window start of every entry of list k (Mask::getPos / BaseMask::getPos)."""
        if not self.n:
            return np.zeros(0, np.uint32)
        if k < 3:
            return np.ctypeslib.as_array(self.h.contents.pos[k], shape=(self.n,)).copy()
        own = np.ctypeslib.as_array(self.h.contents.pos[5 - k], shape=(self.n,))
        return own[self.ptr(k)].copy()

    def lookup(self, k: int) -> np.ndarray:
        return np.ctypeslib.as_array(self.h.contents.lookup[k], shape=(2 << 22,)).copy()


class CompactIndex:
    """CPU-baseline form: borrowed sorted sign[] / pos[] per list (seedl <= 32)."""

    def __init__(self, g: Genome, seedl: int, signs, poss):
        self.g = g
        self.signs = [np.ascontiguousarray(e, dtype=np.uint32) for e in signs]   # keep alive
        self.poss = [np.ascontiguousarray(e, dtype=np.uint32) for e in poss]
        self.n = int(self.signs[0].shape[0])
        sa = (C.c_void_p * 6)(*[e.ctypes.data for e in self.signs])
        pa = (C.c_void_p * 6)(*[e.ctypes.data for e in self.poss])
        self.h = lib().ora_index_from_entries(g.h, seedl, self.n, sa, pa)
        if not self.h:
            raise ValueError("compact index needs seedl <= 32")
        self.seedl = seedl
        self.have_next = False

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_index_free(self.h)
            self.h = None


def make_params(seedl=32, seedkmax=2, totalkmax=5, scores=True, filter_level=2, fileid=0, threads=0,
                LL: Optional[np.ndarray] = None) -> OraParams:
    p = OraParams()
    p.seedl, p.seedkmax, p.totalkmax, p.scores = seedl, seedkmax, totalkmax, int(bool(scores))
    p.fileid, p.threads = fileid, threads
    p.filter_mult = filter_mult(filter_level, totalkmax)
    if LL is None:
        LL, _ = scoring_table()
    for i in range(1024):
        p.LL[i] = float(LL[i])
    return p


NOSCORE_INIT = np.float32(-np.finfo(np.float32).max)   # UniqueMatchInfo.hpp:191


def match_unique(g: Genome, ix: Index, p: OraParams, bases, qual, offsets,
                 info: Optional[np.ndarray] = None, score: Optional[np.ndarray] = None,
                 want_events: bool = False):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.shape[0] - 1
    if info is None:
        info = np.zeros(n, dtype=np.uint64)
    if score is None:
        score = np.full(n, NOSCORE_INIT, dtype=np.float32)
    ctr = OraCounters()
    ev = None
    nev = C.c_uint64(0)
    cap = 0
    if want_events:
        cap = max(1024, 64 * n)
        ev = np.zeros(cap, dtype=EVENT_DTYPE)
    rc = lib().ora_match_unique(g.h, ix.h, C.byref(p), _p(bases), _p(qual), _p(offsets), n,
                                _p(info), _p(score), C.byref(ctr), _p(ev), cap, C.byref(nev))
    if rc != 0:
        raise RuntimeError("ora_match_unique rc=%d" % rc)
    if want_events:
        return info, score, ctr.as_dict(), ev[:nev.value]
    return info, score, ctr.as_dict()


def match_all(g: Genome, ix: Index, p: OraParams, bases, qual, offsets):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.shape[0] - 1
    cap = max(1024, 16 * n)
    while True:
        out = np.zeros(cap, dtype=HIT_DTYPE)
        hoff = np.zeros(n + 1, dtype=np.uint64)
        nout = C.c_uint64(0)
        ctr = OraCounters()
        rc = lib().ora_match_all(g.h, ix.h, C.byref(p), _p(bases), _p(qual), _p(offsets), n,
                                 _p(out), cap, C.byref(nout), _p(hoff), C.byref(ctr))
        if rc == -1:
            cap = int(nout.value) + 16
            continue
        if rc != 0:
            raise RuntimeError("ora_match_all rc=%d" % rc)
        return out[:nout.value], hoff, ctr.as_dict()


def unpack_record(rec):
    """UniqueMatchInfo.hpp:29-39 -> (state, frag, errors, fileid, pos) arrays."""
    rec = np.asarray(rec, dtype=np.uint64)
    state = np.minimum(rec >> np.uint64(61), np.uint64(4))
    return (state.astype(np.int64), ((rec >> np.uint64(45)) & np.uint64(0xffff)).astype(np.int64),
            ((rec >> np.uint64(41)) & np.uint64(15)).astype(np.int64),
            ((rec >> np.uint64(35)) & np.uint64(63)).astype(np.int64),
            (rec & np.uint64((1 << 35) - 1)).astype(np.int64))
