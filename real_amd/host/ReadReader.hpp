// Streaming FASTA / FASTQ read parser: the surface of FastAReader / FastQReader
// (FastAReader.hpp:107-138, FastQReader.hpp:130-240) + Pattern::computeMapped (Pattern.hpp:105-128).
// Produces decoded pattern blocks in the shape the C ABI takes: concatenated mapped symbols
// (A,C,G,T -> 0..3, anything else incl. lowercase -> 4), qualities = ASCII - offset, offsets, ids.
#pragma once
#include <stdint.h>
#include <cstdio>
#include <string>
#include <vector>

struct ReadBlock {
    std::vector<uint8_t> bases, qual;
    std::vector<uint64_t> offsets; // n+1
    std::vector<std::string> ids;
    uint64_t first_id = 0;         // patid of the first read
    uint64_t size() const { return offsets.empty() ? 0 : offsets.size() - 1; }
    void clear() { bases.clear(); qual.clear(); offsets.assign(1, 0); ids.clear(); }
};

class ReadReader {
public:
    // start_offset / first_id: begin at a byte offset of the file that is the start of record number first_id (the host
    // reader taking over from the device parser in the middle of a file)
    ReadReader(const std::string &filename, bool fastq, int qualityOffset, uint64_t start_offset = 0, uint64_t first_id = 0);
    ~ReadReader();
    // up to max_reads reads; returns the number read (0 at end of file)
    uint64_t fillBlock(ReadBlock &b, uint64_t max_reads, bool want_ids);
    static uint64_t countPatterns(const std::string &filename, bool fastq);
    // FastQReader::getOffset (FastQReader.hpp:221-239): first quality char <= 54 -> 33, >= 94 -> 64, else 0
    static int getOffset(const std::string &filename);
private:
    int getc_();
    bool next(std::string *id, std::vector<uint8_t> &bases, std::vector<uint8_t> *qual, bool raw_quality);
    void findNextMarker();
    FILE *f_;
    bool fastq_;
    int qoff_;
    char marker_;
    bool found_ = false;
    uint64_t nextid_ = 0;
    std::vector<unsigned char> buf_;
    size_t pos_ = 0, len_ = 0;
    friend struct ReadReaderAccess;
};
