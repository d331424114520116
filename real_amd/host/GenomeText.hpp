// Genome FASTA loader: the surface of getText<sse4>() + RangeSet (getText.hpp:31-55,
// countReads.cpp:28-125, getFileList.cpp:145-174), rewritten.  One pass: every character
// other than A,C,G,T,N outside header lines is dropped (lowercase too -- reference quirk 4,
// coordinates are in the filtered text); fragment names are everything after '>' up to the
// newline; the last range is the "terminal" one.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct GenomeText {
    std::vector<uint8_t> sym;                 // 0..3 = ACGT, 4 = N
    std::vector<std::string> frag_names;      // n_frag
    std::vector<uint64_t> frag_start;         // n_frag + 1, last = sym.size()
    uint64_t n_wild = 0;
    void load(const std::string &fasta);      // throws std::runtime_error
    // AutoTextArray layout (AutoTextArray.hpp:28-61) for the host-packed form of the ABI
    void pack(std::vector<uint64_t> &text2bit, std::vector<uint64_t> &wildbits) const;
};

// -t is a file ending in ".fa" or a directory searched recursively in readdir order
void getFileList(const std::string &name, std::vector<std::string> &files, const std::string &suffix = ".fa");
