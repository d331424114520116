#!/usr/bin/env python3
"""Generate tests/golden/readers.npz: tricky FASTA / FASTQ / genome texts and what the REFERENCE's own readers make of
them (oracle/_ref/ref_readers = FastQReader.hpp, FastAReader.hpp, Pattern.hpp, countReads.cpp compiled from
/root/reference/src by oracle/Makefile).  Runs in the build container only; the fixture is data (inputs + outputs).

    make -C oracle ref && python tests/golden/make_reader_golden.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_readers")

FQ = {
    "fq_canonical_q33": b"@r0\nACGTACGTAA\n+\nIIII*IIIII\n@r1 with spaces\tand tab\nTTGGCCAA\n+r1\n55555555\n",
    "fq_q64": b"@a\nACGT\n+\nhhhh\n@b\nGGCC\n+\nhh^h\n",
    "fq_crlf": b"@r0\r\nACGTAC\r\n+\r\nIIII*I\r\n@r1\r\nGGTTAA\r\n+\r\n555555\r\n",
    "fq_wrapped": b"@w0\nACGTAC\nGTAA\n+\nIIII*\nIIIII\n@w1\nAC\nGT\n+w1\nII\nII\n",
    "fq_lowercase_iupac_n": b"@x\nACGTNacgtRYKM\n+\nIIIIIIIIIIII*\n",
    "fq_at_in_quality": b"@q0\nACGTACGT\n+\n@III*III\n@q1\nTTTTCCCC\n+\nI@@@IIII\n",
    "fq_blank_lines": b"\n@b0\nACGT\n+\nIIII\n\n\n@b1\nGGCC\n+\n*III\n\n",
    "fq_no_final_newline": b"@n0\nACGT\n+\nIIII\n@n1\nGGCCA\n+\nII*II",
    "fq_truncated_last": b"@t0\nACGT\n+\nIIII\n@t1\nGGCCAA\n+\nII*\n",
    "fq_ragged_lengths": b"@a\nA\n+\n*\n@b\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII5\n@c\nAC\n+\nI5\n",
}
FA = {
    "fa_canonical": b">r0\nACGTACGTAA\n>r1 desc\nTTGGCCAA\n",
    "fa_wrapped_lowercase": b">w0\nACGTAC\nGTaa\nNN\n>w1\nAC\n\nGT\n",
    "fa_crlf": b">c0\r\nACGT\r\n>c1\r\nGG\r\nCC\r\n",
    "fa_no_final_newline": b">n0\nACGT\n>n1\nGGCC",
    "fa_leading_junk": b"junk line\n>j0\nAC GT\n>j1\nT\tT\n",
}
GENOME = {
    "g_simple": b"> random_20\nACGTACGTAC\nGTACGTACGT\n",
    "g_multi_lowercase_n": b">chr1 first\nACGTNNNNacgtACGT\nRYKMACGT\n>chr2\nTTTT\nGGGG\n>chr3\nA\n",
    "g_crlf": b">c1\r\nACGT\r\nNNAC\r\n>c2\r\nGGCC\r\n",
    "g_gt_inside_header": b">a>b name\nACGT\n>c\nGG>x\nTT\n",
    "g_no_final_newline": b">z1\nACGT\n>z2\nGGCC",
    "g_header_at_eof": b">h1\nACGTAC\n>h2",
}


def run(mode, text):
    d = tempfile.mkdtemp(prefix="rr_")
    fn = os.path.join(d, "in.txt")
    open(fn, "wb").write(text)
    subprocess.run([REF, mode, fn, d], check=True, stderr=subprocess.DEVNULL, timeout=60)
    rd = lambda name, dt: np.fromfile(os.path.join(d, name), dtype=dt)
    if mode == "genome":
        return {"names": rd("names.bin", np.uint8), "frag": rd("frag.u64", np.uint64), "sym": rd("sym.u8", np.uint8)}
    meta = [int(x) for x in open(os.path.join(d, "meta.txt")).read().split()]
    return {"ids": rd("ids.bin", np.uint8), "off": rd("off.u64", np.uint64), "bases": rd("bases.u8", np.uint8), "qual": rd("qual.u8", np.uint8),
            "meta": np.array(meta, dtype=np.int64)}         # countPatterns, detected quality offset, patterns read


def main():
    assert os.path.exists(REF), "build oracle/_ref/ref_readers first: make -C oracle ref"
    out = {}
    for mode, cases in (("fq", FQ), ("fa", FA), ("genome", GENOME)):
        for name, text in cases.items():
            out[name + "/input"] = np.frombuffer(text, dtype=np.uint8)
            for k, v in run(mode, text).items():
                out[name + "/" + k] = v
    np.savez_compressed(os.path.join(HERE, "readers.npz"), **out)
    print("wrote readers.npz: %d cases" % (len(FQ) + len(FA) + len(GENOME)))
    for name in list(FQ) + list(FA):
        print(name, out[name + "/meta"], bytes(out[name + "/ids"]).split(b"\0")[:-1], "".join(map(str, out[name + "/bases"])))
    for name in GENOME:
        print(name, bytes(out[name + "/names"]).split(b"\0")[:-1], out[name + "/frag"], "".join(map(str, out[name + "/sym"])))


if __name__ == "__main__":
    main()
