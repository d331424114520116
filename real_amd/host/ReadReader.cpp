#include "ReadReader.hpp"

#include <cctype>
#include <stdexcept>

ReadReader::ReadReader(const std::string &filename, bool fastq, int qualityOffset, uint64_t start_offset, uint64_t first_id)
    : f_(nullptr), fastq_(fastq), qoff_(qualityOffset), marker_(fastq ? '@' : '>'), nextid_(first_id), buf_(1 << 22)
{
    f_ = (filename == "-") ? stdin : fopen(filename.c_str(), "rb");
    if (!f_) throw std::runtime_error("Unable to open pattern file.");
    if (start_offset && fseeko(f_, (off_t)start_offset, SEEK_SET) != 0) throw std::runtime_error("Unable to seek in the pattern file.");
    findNextMarker();
}
ReadReader::~ReadReader() { if (f_ && f_ != stdin) fclose(f_); }

int ReadReader::getc_()
{
    if (pos_ == len_) {
        len_ = fread(buf_.data(), 1, buf_.size(), f_);
        pos_ = 0;
        if (!len_) return -1;
    }
    return buf_[pos_++];
}

void ReadReader::findNextMarker()
{
    int c;
    while ((c = getc_()) >= 0 && c != marker_) {}
    found_ = (c == marker_);
}

// one record; sequence = every non-space character up to the next '>' (FASTA, may span lines) or
// up to '+' (FASTQ); FASTQ quality = the next patlen non-space characters
bool ReadReader::next(std::string *id, std::vector<uint8_t> &bases, std::vector<uint8_t> *qual, bool raw_quality)
{
    if (!found_) return false;
    found_ = false;
    int c;
    if (id) id->clear();
    while ((c = getc_()) >= 0 && c != '\n') if (id) id->push_back((char)c);
    if (c < 0) return false;
    const size_t b0 = bases.size();
    const char stop = fastq_ ? '+' : '>';
    while ((c = getc_()) >= 0 && c != stop) {
        if (isspace(c)) continue;
        uint8_t m;
        switch (c) { case 'A': m = 0; break; case 'C': m = 1; break; case 'G': m = 2; break; case 'T': m = 3; break; default: m = 4; }
        bases.push_back(m);
    }
    const size_t patlen = bases.size() - b0;
    if (!fastq_) { found_ = (c == '>'); nextid_++; return true; }
    while ((c = getc_()) >= 0 && c != '\n') {} // the '+' line
    if (c < 0) { bases.resize(b0); return false; }
    size_t got = 0;
    while (got < patlen && (c = getc_()) >= 0) {
        if (isspace(c)) continue;
        if (qual) qual->push_back(raw_quality ? (uint8_t)c : (uint8_t)(c - qoff_));
        got++;
    }
    if (got < patlen) { bases.resize(b0); if (qual) qual->resize(qual->size() - got); return false; }
    nextid_++;
    findNextMarker();
    return true;
}

uint64_t ReadReader::fillBlock(ReadBlock &b, uint64_t max_reads, bool want_ids)
{
    b.clear();
    b.first_id = nextid_;
    std::string id;
    uint64_t n = 0;
    while (n < max_reads) {
        const size_t before = b.bases.size();
        if (!next(want_ids ? &id : nullptr, b.bases, fastq_ ? &b.qual : nullptr, false)) break;
        if (!fastq_) b.qual.resize(b.bases.size(), 30); // PatternBase::getQuality, Pattern.hpp:42-45
        (void)before;
        b.offsets.push_back(b.bases.size());
        if (want_ids) b.ids.push_back(id);
        n++;
    }
    return n;
}

uint64_t ReadReader::countPatterns(const std::string &filename, bool fastq)
{
    ReadReader r(filename, fastq, 0);
    std::vector<uint8_t> bases, qual;
    uint64_t n = 0;
    while (true) {
        bases.clear(); qual.clear();
        if (!r.next(nullptr, bases, fastq ? &qual : nullptr, true)) break;
        n++;
    }
    return n;
}

int ReadReader::getOffset(const std::string &filename)
{
    ReadReader r(filename, true, 0);
    std::vector<uint8_t> bases, qual;
    while (true) {
        bases.clear(); qual.clear();
        if (!r.next(nullptr, bases, &qual, true)) break;
        for (uint8_t q : qual) {
            if (q <= 54) return 33; // Sanger
            if (q >= 94) return 64; // Illumina
        }
    }
    return 0;
}
