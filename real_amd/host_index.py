"""Host-side genome text packing and index build (numpy), for the host-built form
of the C ABI (real_hip_set_text / real_hip_set_index_block).

north_star keeps "the signature ListSet + ParallelRadixSort index build on the
host"; this module is that host build, vectorised: window enumeration
(MapTextFile.hpp:118-230), the six signatures per window (:211-216) and one
stable sort per list (ListSet.hpp:41-44; the reference's radix sort is stable,
ParallelRadixSort.hpp:160-203, so equal signatures keep ascending position).
The `ptr` cross links of Mask.hpp are not produced: the device re-reads the
partner segments from the text.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def pack_text(sym: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """AutoTextArray's getTextArray / getWildcardArray (AutoTextArray.hpp:28-61):
    2 bits per base MSB first, N stored as 0 plus a wildcard bit."""
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    n = sym.shape[0]
    nt = (2 * n + 63) // 64
    two = np.zeros(nt * 32, dtype=np.uint8)
    two[:n] = sym & 3
    two = two.reshape(nt, 32).astype(np.uint64)
    shifts = (62 - 2 * np.arange(32, dtype=np.uint64)).astype(np.uint64)
    text = np.bitwise_or.reduce(two << shifts[None, :], axis=1) if nt else np.zeros(0, np.uint64)
    nw = (n + 63) // 64
    wb = np.zeros(nw * 64, dtype=np.uint8)
    wb[:n] = sym > 3
    wild = np.packbits(wb.reshape(nw, 64), axis=1, bitorder="big").view(">u8").astype(np.uint64).reshape(-1) if nw else np.zeros(0, np.uint64)
    return text.astype(np.uint64), wild


def valid_windows(sym: np.ndarray, seedl: int) -> np.ndarray:
    """Start positions i with sym[i:i+seedl] free of N, ascending (the windows
    MapTextFile::readNextSignature emits; fragment boundaries do not cut windows)."""
    n = sym.shape[0]
    if n < seedl:
        return np.zeros(0, dtype=np.uint32)
    isn = (sym > 3).astype(np.int64)
    c = np.concatenate([[0], np.cumsum(isn)])
    bad = c[seedl:] - c[:n - seedl + 1]
    return np.nonzero(bad == 0)[0].astype(np.uint32)


def segment_values(sym: np.ndarray, wpos: np.ndarray, seedl: int) -> List[np.ndarray]:
    """m0..m3 of every window (2-bit MSB-first pack of seedl/4 bases each)."""
    q = seedl // 4
    out = []
    s64 = (sym & 3).astype(np.uint64)
    for j in range(4):
        m = np.zeros(wpos.shape[0], dtype=np.uint64)
        base = wpos.astype(np.int64) + j * q
        for t in range(q):
            m = (m << np.uint64(2)) | s64[base + t]
        out.append(m)
    return out


_PAIRS = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]   # s0..s5, SignatureConstruction.hpp:62-67


def build_lists(sym: np.ndarray, seedl: int, first_window: int = 0, max_entries: int = 1 << 62):
    """-> (sign[6], pos[6], n_entries, have_next) for the block of windows
    [first_window, first_window + max_entries)."""
    w = valid_windows(sym, seedl)
    total = w.shape[0]
    w = w[first_window:first_window + max_entries] if first_window < total else w[:0]
    m = segment_values(sym, w, seedl)
    bits = np.uint64(2 * (seedl // 4))
    sdt = np.uint32 if seedl <= 32 else np.uint64      # real.cpp:219-229
    sign, pos = [], []
    for (x, y) in _PAIRS:
        s = (m[x] << bits) | m[y]
        order = np.argsort(s, kind="stable")
        sign.append(s[order].astype(sdt))
        pos.append(w[order].astype(np.uint32))
    return sign, pos, int(w.shape[0]), bool(first_window + w.shape[0] < total)
