#include "GenomeText.hpp"

#include <dirent.h>
#include <sys/stat.h>

#include <cstdio>
#include <stdexcept>

void GenomeText::load(const std::string &fasta)
{
    FILE *f = fopen(fasta.c_str(), "rb");
    if (!f) throw std::runtime_error("Could not open text file " + fasta);
    sym.clear(); frag_names.clear(); frag_start.clear(); n_wild = 0;
    std::vector<char> buf(1 << 22);
    bool in_header = false;
    std::string id;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const char c = buf[i];
            if (c == '>') { in_header = true; id.clear(); frag_start.push_back(sym.size()); continue; } // countReads.cpp:44-50
            if (c == '\n') { if (in_header) frag_names.push_back(id); in_header = false; continue; }
            if (in_header) { id += c; continue; }
            switch (c) {
            case 'A': sym.push_back(0); break;
            case 'C': sym.push_back(1); break;
            case 'G': sym.push_back(2); break;
            case 'T': sym.push_back(3); break;
            case 'N': sym.push_back(4); ++n_wild; break;
            default: break; // dropped
            }
        }
    }
    fclose(f);
    if (in_header) frag_names.push_back(id);
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
