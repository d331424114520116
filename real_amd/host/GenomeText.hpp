// Genome FASTA loader: the surface of getText<sse4>() + RangeSet (getText.hpp:31-55,
// countReads.cpp:28-125, getFileList.cpp:145-174), rewritten.  One pass: every character
// other than A,C,G,T,N outside header lines is dropped (lowercase too -- reference quirk 4,
// coordinates are in the filtered text); fragment names are everything after '>' up to the
// newline; the last range is the "terminal" one.
#pragma once
#include <stdint.h>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

// the symbols of a genome: a byte array that is NOT zero-filled when it is sized (3 GB written once by all threads;
// std::vector would first clear it with one)
class SymArray {
public:
    SymArray() {}
    ~SymArray() { free(p_); }
    SymArray(const SymArray &) = delete;
    SymArray &operator=(const SymArray &) = delete;
    SymArray(SymArray &&o) : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    SymArray &operator=(SymArray &&o) { if (this != &o) { free(p_); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    void clear() { free(p_); p_ = nullptr; n_ = 0; }
    void resize_uninitialized(size_t n)
    {
        clear();
        if (n && !(p_ = (uint8_t *)malloc(n))) throw std::bad_alloc();
        n_ = n;
    }
    uint8_t *data() { return p_; }
    const uint8_t *data() const { return p_; }
    size_t size() const { return n_; }
    uint8_t operator[](size_t i) const { return p_[i]; }
private:
    uint8_t *p_ = nullptr;
    size_t n_ = 0;
};

struct GenomeText {
    SymArray sym;                             // 0..3 = ACGT, 4 = N
    std::vector<std::string> frag_names;      // n_frag
    std::vector<uint64_t> frag_start;         // n_frag + 1, last = sym.size()
    uint64_t n_wild = 0;
    void load(const std::string &fasta);      // throws std::runtime_error
    // AutoTextArray layout (AutoTextArray.hpp:28-61) for the host-packed form of the ABI
    void pack(std::vector<uint64_t> &text2bit, std::vector<uint64_t> &wildbits) const;
};

// -t is a file ending in ".fa" or a directory searched recursively in readdir order
void getFileList(const std::string &name, std::vector<std::string> &files, const std::string &suffix = ".fa");
