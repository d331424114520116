"""Seeded synthetic genomes and reads.

The reference only ships time(0)-seeded generators (randstr.cpp:27-53,
genpat.cpp:64-166); these reproduce their *distributions* reproducibly:

* genome: i.i.d. uniform ACGT (randstr.cpp), optionally with N runs and several
  fragments (FASTA records) so the N / fragment-boundary filters are exercised;
* reads: start positions uniform over n-patl+1 and sorted, strand flip p=0.5,
  per-base substitution with probability ``errprob`` to a different base, FASTQ
  quality 'D' (unchanged) / '*' (mutated) (genpat.cpp:96-157); ground truth in
  the id ``p<pos>[_inv][_<j><old><new>]...``.

Symbols are the reference's mapped alphabet: A,C,G,T -> 0..3, anything else -> 4
(acgtnMap.hpp:39-50).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

ALPHABET = "ACGTN"


@dataclass
class Genome:
    sym: np.ndarray                 # uint8 symbols 0..4, length n
    frag_start: np.ndarray          # uint64, n_frag+1 entries, last = n ("terminal", countReads.cpp:81)
    frag_names: List[str] = field(default_factory=list)

    @property
    def n(self) -> int:
        return int(self.sym.shape[0])

    @property
    def n_frag(self) -> int:
        return int(self.frag_start.shape[0]) - 1


@dataclass
class ReadBatch:
    bases: np.ndarray               # uint8 mapped symbols, concatenated
    qual: np.ndarray                # uint8 quality values (ASCII - offset), concatenated
    offsets: np.ndarray             # uint64, n_reads+1
    ids: Optional[List[str]] = None
    true_pos: Optional[np.ndarray] = None
    true_inv: Optional[np.ndarray] = None

    @property
    def n_reads(self) -> int:
        return int(self.offsets.shape[0]) - 1


def random_genome(n: int, seed: int, n_frag: int = 1, n_runs: int = 0, n_run_len: int = 7,
                  repeats: int = 0, repeat_len: int = 300) -> Genome:
    """Uniform ACGT genome.  ``n_runs`` runs of N and ``repeats`` copied segments
    (non-unique loci) can be planted; ``n_frag`` fragments of random sizes."""
    rng = np.random.default_rng(seed)
    sym = rng.integers(0, 4, size=n, dtype=np.uint8)
    for _ in range(repeats):
        if n <= 2 * repeat_len:
            break
        src = int(rng.integers(0, n - repeat_len))
        dst = int(rng.integers(0, n - repeat_len))
        seg = sym[src:src + repeat_len].copy()
        # a few substitutions so that the copies differ by 0..3 bases
        for _k in range(int(rng.integers(0, 4))):
            j = int(rng.integers(0, repeat_len))
            seg[j] = (seg[j] + 1 + rng.integers(0, 3)) & 3
        sym[dst:dst + repeat_len] = seg
    for _ in range(n_runs):
        ln = int(rng.integers(1, n_run_len + 1))
        p = int(rng.integers(0, max(1, n - ln)))
        sym[p:p + ln] = 4
    if n_frag <= 1:
        starts = np.array([0, n], dtype=np.uint64)
    else:
        cuts = np.sort(rng.choice(np.arange(1, n), size=n_frag - 1, replace=False))
        starts = np.concatenate([[0], cuts, [n]]).astype(np.uint64)
    names = [" random_%d_%d" % (n, i) for i in range(len(starts) - 1)]
    return Genome(sym=sym, frag_start=starts, frag_names=names)


_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def revcomp(mapped: np.ndarray) -> np.ndarray:
    """Pattern::computeMapped 'transposed' (Pattern.hpp:105-128, acgtnMap.hpp:58-68)."""
    return _COMP[mapped[::-1]]


def sample_reads(g: Genome, n_reads: int, patl: int, errprob: float, seed: int,
                 q_ok: int = 35, q_mut: int = 9, with_ids: bool = True,
                 n_read_prob: float = 0.0) -> ReadBatch:
    """genpat-style reads of uniform length ``patl`` (quality values already
    offset-free: 'D'-33 = 35, '*'-33 = 9)."""
    rng = np.random.default_rng(seed)
    n = g.n
    numpos = n - patl + 1
    assert numpos > 0
    pos = np.sort(rng.integers(0, numpos, size=n_reads, dtype=np.int64))
    inv = rng.integers(0, 2, size=n_reads).astype(bool)
    idx = pos[:, None] + np.arange(patl, dtype=np.int64)[None, :]
    rd = g.sym[idx]                                    # (n_reads, patl)
    rd[inv] = _COMP[rd[inv][:, ::-1]]
    mut = rng.random(size=rd.shape) < errprob
    delta = rng.integers(1, 4, size=rd.shape).astype(np.uint8)
    orig = rd.copy()
    isn = rd > 3
    rd = np.where(mut & ~isn, (rd + delta) & 3, rd).astype(np.uint8)
    if n_read_prob > 0:
        rd = np.where(rng.random(size=rd.shape) < n_read_prob, np.uint8(4), rd)
    qual = np.where(rd == orig, np.uint8(q_ok), np.uint8(q_mut)).astype(np.uint8)
    ids = None
    if with_ids:
        ids = []
        for i in range(n_reads):
            s = "p%d" % pos[i]
            if inv[i]:
                s += "_inv"
            for j in np.nonzero(rd[i] != orig[i])[0]:
                s += "_%d%s%s" % (j, ALPHABET[orig[i, j]], ALPHABET[rd[i, j]])
            ids.append(s)
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(patl))
    return ReadBatch(bases=rd.reshape(-1).copy(), qual=qual.reshape(-1).copy(), offsets=offsets, ids=ids,
                     true_pos=pos.astype(np.uint64), true_inv=inv)


def concat_batches(batches: Sequence[ReadBatch]) -> ReadBatch:
    """Ragged batch from several uniform-length ones."""
    bases = np.concatenate([b.bases for b in batches])
    qual = np.concatenate([b.qual for b in batches])
    offs = [np.zeros(1, dtype=np.uint64)]
    base = np.uint64(0)
    for b in batches:
        offs.append(b.offsets[1:] + base)
        base = base + b.offsets[-1]
    ids = None
    if all(b.ids is not None for b in batches):
        ids = [s for b in batches for s in b.ids]
    return ReadBatch(bases=bases, qual=qual, offsets=np.concatenate(offs).astype(np.uint64), ids=ids)


def genome_to_fasta(g: Genome, path: str, cols: int = 60) -> None:
    with open(path, "w") as f:
        for k in range(g.n_frag):
            name = g.frag_names[k] if g.frag_names else " frag%d" % k
            f.write(">" + name + "\n")
            s = "".join(ALPHABET[c] for c in g.sym[int(g.frag_start[k]):int(g.frag_start[k + 1])])
            for i in range(0, len(s), cols):
                f.write(s[i:i + cols] + "\n")


def reads_to_fastq(b: ReadBatch, path: str, offset: int = 33) -> None:
    with open(path, "w") as f:
        for i in range(b.n_reads):
            lo, hi = int(b.offsets[i]), int(b.offsets[i + 1])
            f.write("@" + (b.ids[i] if b.ids else "r%d" % i) + "\n")
            f.write("".join(ALPHABET[c] for c in b.bases[lo:hi]) + "\n+\n")
            f.write("".join(chr(int(q) + offset) for q in b.qual[lo:hi]) + "\n")


def reads_to_fasta(b: ReadBatch, path: str) -> None:
    with open(path, "w") as f:
        for i in range(b.n_reads):
            lo, hi = int(b.offsets[i]), int(b.offsets[i + 1])
            f.write(">" + (b.ids[i] if b.ids else "r%d" % i) + "\n")
            f.write("".join(ALPHABET[c] for c in b.bases[lo:hi]) + "\n")


def pack_bases(bases: np.ndarray):
    """mapped symbols (one per byte) -> the 2-bit packed form of a real_hip_batch with packed = 1: base g of the
    concatenated batch at bits 7-2(g%4)-1.. of byte g/4.  Symbols > 3 cannot be packed (they are stored as 0); the
    caller flags their reads in nflags."""
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    pad = (-b.shape[0]) % 4
    if pad:
        b = np.concatenate([b, np.zeros(pad, dtype=np.uint8)])
    q = (b & 3).reshape(-1, 4)
    return ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)


def read_nflags(bases: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """bit (i%8) of byte i/8 set iff read i holds a symbol > 3"""
    n = offsets.shape[0] - 1
    bad = np.zeros(n, dtype=bool)
    idx = np.nonzero(np.asarray(bases) > 3)[0]
    if idx.size:
        bad[np.searchsorted(offsets.astype(np.int64), idx, side="right") - 1] = True
    return np.packbits(bad, bitorder="little")
