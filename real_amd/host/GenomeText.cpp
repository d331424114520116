#include "GenomeText.hpp"

#include <dirent.h>
#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

#include <cstdio>
#include <stdexcept>

// countLength + readFile (countReads.cpp:28-125) in one parallel sweep.  The reference's scanner has one bit of state
// -- "ignore the rest of the line", set by '>' (anywhere in a line), cleared by '\n' -- so the file, mapped into
// memory, is cut at line starts into one piece per thread: pass 1 counts the kept symbols of every piece and collects
// its header lines, a prefix sum places the pieces, pass 2 writes the symbols.  A header line's range starts at the
// number of symbols kept in front of it; its name is what follows the LAST '>' of the line (the reference restarts the
// name there); a header without a newline at the very end of the file opens no range (the reference pushes a range
// when it sees the newline).
namespace {
struct Piece {
    size_t lo = 0, hi = 0;
    uint64_t kept = 0, wild = 0;
    std::vector<std::pair<std::string, uint64_t>> headers; // name, symbols kept in front of it inside the piece
};
}

void GenomeText::load(const std::string &fasta)
{
    sym.clear(); frag_names.clear(); frag_start.clear(); n_wild = 0;
    const int fd = open(fasta.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Could not open text file " + fasta);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("Could not open text file " + fasta); }
    const size_t size = (size_t)st.st_size;
    const char *data = nullptr;
    std::vector<char> slurp; // (not a regular file: read it)
    void *map = MAP_FAILED;
    if (size) {
        map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map != MAP_FAILED) { data = (const char *)map; (void)madvise(map, size, MADV_SEQUENTIAL); }
    }
    size_t n = size;
    bool fd_open = true;
    if (!data) {
        fd_open = false; // (the stream owns the descriptor from here)
        FILE *f = fdopen(fd, "rb");
        if (!f) { close(fd); throw std::runtime_error("Could not open text file " + fasta); }
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) slurp.insert(slurp.end(), buf, buf + got);
        fclose(f);
        data = slurp.data(); n = slurp.size();
    }
    const int nt = std::max(1, omp_get_max_threads());
    std::vector<Piece> pc((size_t)nt);
    for (int t = 0; t < nt; ++t) { // cuts at line starts
        size_t lo = n * (size_t)t / (size_t)nt;
        if (t && lo > 0 && lo < n) { // (lo == 0: fewer bytes than threads -- offset 0 is a line start, and there is no byte in front of it)
            const void *nl = memchr(data + lo - 1, '\n', n - (lo - 1)); // (the line that holds byte lo-1 ends here)
            lo = nl ? (size_t)((const char *)nl - data) + 1 : n;
        }
        pc[(size_t)t].lo = lo;
        if (t) pc[(size_t)t - 1].hi = lo;
    }
    pc[(size_t)nt - 1].hi = n;
    for (int t = 1; t < nt; ++t) if (pc[(size_t)t].lo < pc[(size_t)t - 1].lo) pc[(size_t)t].lo = pc[(size_t)t - 1].lo; // (monotone by construction)
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; ++t) {
        Piece &P = pc[(size_t)t];
        bool ignore = false;
        std::string id;
        uint64_t kept = 0, wild = 0;
        for (size_t i = P.lo; i < P.hi; ++i) {
            const char c = data[i];
            if (c == '>') { ignore = true; id.clear(); }
            else if (c == '\n') { if (ignore) P.headers.emplace_back(id, kept); ignore = false; }
            else if (ignore) id.push_back(c);
            else if (c == 'A' || c == 'C' || c == 'G' || c == 'T') kept++;
            else if (c == 'N') { kept++; wild++; }
        }
        P.kept = kept; P.wild = wild;
    }
    uint64_t total = 0;
    std::vector<uint64_t> base((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        base[(size_t)t] = total;
        for (auto &h : pc[(size_t)t].headers) { frag_names.push_back(h.first); frag_start.push_back(total + h.second); }
        total += pc[(size_t)t].kept;
        n_wild += pc[(size_t)t].wild;
    }
    sym.resize_uninitialized(total);
    uint8_t *out = sym.data();
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; ++t) {
        const Piece &P = pc[(size_t)t];
        uint8_t *w = out + base[(size_t)t];
        bool ignore = false;
        for (size_t i = P.lo; i < P.hi; ++i) {
            const char c = data[i];
            if (c == '>') ignore = true;
            else if (c == '\n') ignore = false;
            else if (!ignore) {
                switch (c) {
                case 'A': *w++ = 0; break;
                case 'C': *w++ = 1; break;
                case 'G': *w++ = 2; break;
                case 'T': *w++ = 3; break;
                case 'N': *w++ = 4; break;
                default: break; // dropped (lowercase too: reference quirk 4)
                }
            }
        }
    }
    if (map != MAP_FAILED) munmap(map, size);
    if (fd_open) close(fd);
    if (frag_start.empty()) throw std::runtime_error("no FASTA record in " + fasta);
    frag_start.push_back(sym.size()); // "terminal"
    // empty records cannot be represented by the reference's RangeVector either (fillRange)
    for (size_t i = 0; i + 1 < frag_start.size(); ++i)
        if (frag_start[i + 1] <= frag_start[i]) throw std::runtime_error("empty FASTA record in " + fasta);
    if (sym.size() > 0xffffffffull) throw std::runtime_error("text longer than 2^32 positions (positions are 32 bit, Mask.hpp)");
    if (frag_names.size() > 65536) throw std::runtime_error("more than 65536 fragments in one file (UniqueMatchInfo.hpp:31-32)");
}

void GenomeText::pack(std::vector<uint64_t> &text, std::vector<uint64_t> &wild) const
{
    const uint64_t n = sym.size();
    text.assign((2 * n + 63) / 64, 0);
    wild.assign((n + 63) / 64, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t s = sym[i];
        text[i >> 5] |= (s & 3) << (62 - 2 * (i & 31));
        if (s > 3) wild[i >> 6] |= 1ull << (63 - (i & 63));
    }
}

static bool ends_on(const std::string &s, const std::string &suf)
{
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

static void enumerate(const std::string &dir, std::vector<std::string> &files, const std::string &suffix)
{
    DIR *d = opendir(dir.c_str());
    if (!d) throw std::runtime_error("Could not open file/directory");
    while (struct dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        const std::string path = dir + "/" + name;
        struct stat st;
        if (stat(path.c_str(), &st) != 0) continue;
        if (S_ISREG(st.st_mode)) { if (ends_on(path, suffix)) files.push_back(path); }
        else if (S_ISDIR(st.st_mode) && name != "." && name != "..") enumerate(path + "/", files, suffix);
    }
    closedir(d);
}

void getFileList(const std::string &name, std::vector<std::string> &files, const std::string &suffix)
{
    struct stat st;
    if (stat(name.c_str(), &st) != 0) return;
    if (S_ISREG(st.st_mode) && ends_on(name, suffix)) files.push_back(name);
    else if (S_ISDIR(st.st_mode)) enumerate(name, files, suffix);
}
