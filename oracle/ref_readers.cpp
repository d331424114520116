/*
 * ref_readers.cpp -- white-box tap on the REAL reference's input side, TEST INFRASTRUCTURE ONLY.
 *
 * This file is ours; it contains no reference code.  oracle/Makefile compiles it against the reference's readers
 * where they lie (-I/root/reference/src: FastQReader.hpp, FastAReader.hpp, Pattern.hpp, AsynchronousReader.hpp and
 * what they include) plus countReads.cpp (countLength / readFile; its real_config.hpp include is guarded by
 * HAVE_CONFIG_H, which stays undefined).  -DHAVE_PTHREADS is what the reference's configure defines when it finds
 * pthreads (this image has them): without it AsynchronousReader.hpp is empty.  Output goes to oracle/_ref/.
 * Used by tests/golden/make_golden.py to pin what the host-side readers of this repo (real_amd/host/ReadReader.cpp,
 * GenomeText.cpp) and the device parser must reproduce: ids, mapped symbols, qualities, the quality-offset
 * autodetection, the read count, fragment names and starts, the symbols of the genome.
 *
 * usage: ref_readers fq|fa <reads file> <outdir>    -> ids.bin (id bytes, each followed by \0) off.u64 bases.u8 qual.u8 meta.txt
 *        ref_readers genome <fasta> <outdir>        -> names.bin (each followed by \0) frag.u64 sym.u8
 */
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <stdint.h>

#include "AsynchronousReader.hpp"
#include "FastQReader.hpp"
#include "FastAReader.hpp"
#include "countReads.hpp"

template <typename T>
static void dump(std::string const &fn, std::vector<T> const &v)
{
    std::ofstream out(fn.c_str(), std::ios::binary);
    if (!v.empty()) out.write(reinterpret_cast<char const *>(&v[0]), v.size() * sizeof(T));
}

template <typename reader_type>
static void reads(std::string const &fn, std::string const &dir, bool fastq)
{
    int const off = fastq ? reader_type::getOffset(fn) : 0;
    u_int64_t const cnt = reader_type::countPatterns(fn);
    reader_type r(fn, off);
    typename reader_type::pattern_type p;
    std::vector<char> ids;
    std::vector<uint64_t> offs(1, 0);
    std::vector<uint8_t> bases, qual;
    uint64_t n = 0;
    while (r.getNextPatternUnlocked(p)) {
        p.computeMapped();
        ids.insert(ids.end(), p.sid.begin(), p.sid.end());
        ids.push_back(0);
        for (unsigned i = 0; i < p.patlen; ++i) {
            bases.push_back((uint8_t)p.mapped[i]);
            qual.push_back((uint8_t)p.getQuality(i));
        }
        offs.push_back(bases.size());
        if (p.patid != n) { std::cerr << "patid " << p.patid << " != " << n << std::endl; exit(3); }
        n++;
    }
    dump(dir + "/ids.bin", ids); dump(dir + "/off.u64", offs); dump(dir + "/bases.u8", bases); dump(dir + "/qual.u8", qual);
    std::ofstream m((dir + "/meta.txt").c_str());
    m << cnt << " " << off << " " << n << "\n";
}

int main(int argc, char *argv[])
{
    if (argc != 4) return 2;
    std::string const mode = argv[1], fn = argv[2], dir = argv[3];
    if (mode == "fq") reads<FastQReader>(fn, dir, true);
    else if (mode == "fa") reads<FastAReader>(fn, dir, false);
    else if (mode == "genome") {
        std::vector<std::pair<std::string, u_int64_t> > ranges;
        u_int64_t const n = countLength(fn, ranges);
        AutoArray<u_int8_t> A = readFile(fn, n);
        std::vector<char> names;
        std::vector<uint64_t> starts;
        for (size_t i = 0; i < ranges.size(); ++i) {
            if (i + 1 < ranges.size()) { names.insert(names.end(), ranges[i].first.begin(), ranges[i].first.end()); names.push_back(0); }
            starts.push_back(ranges[i].second);
        }
        std::vector<uint8_t> sym(A.get(), A.get() + n);
        dump(dir + "/names.bin", names); dump(dir + "/frag.u64", starts); dump(dir + "/sym.u8", sym);
    } else return 2;
    return 0;
}
