// host_selftest -- dumps what the host side computes, for the CPU tests (no GPU needed):
//   host_selftest genome <g.fa> <outdir>          -> sym.u8 frag.u64 names.txt text.u64 wild.u64
//   host_selftest reads <reads> <fastq:0|1> <qoff> <outdir> -> bases.u8 qual.u8 off.u64 ids.txt (+ count, offset detect)
//   host_selftest index <g.fa> <seedl> <first> <max> <threads> <outdir> -> l<k>_sign.bin l<k>_pos.u32 meta.txt
//   host_selftest options <args...>               -> prints the parsed RealOptions
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "FastFormat.hpp"
#include "GenomeText.hpp"
#include "HostIndex.hpp"
#include "ReadReader.hpp"
#include "RealOptions.hpp"

template <typename T>
static void dump(const std::string &fn, const T *p, size_t n)
{
    std::ofstream o(fn.c_str(), std::ios::binary);
    o.write(reinterpret_cast<const char *>(p), n * sizeof(T));
}

int main(int argc, char **argv)
{
    try {
        if (argc < 2) return 2;
        std::string cmd = argv[1];
        if (cmd == "genome" && argc == 4) {
            GenomeText G; G.load(argv[2]);
            std::string d = argv[3];
            dump(d + "/sym.u8", G.sym.data(), G.sym.size());
            dump(d + "/frag.u64", G.frag_start.data(), G.frag_start.size());
            std::ofstream n((d + "/names.txt").c_str());
            for (auto &s : G.frag_names) n << s << "\n";
            std::ofstream nb((d + "/names.bin").c_str(), std::ios::binary); // (names may hold any byte but NUL)
            for (auto &s : G.frag_names) { nb.write(s.data(), (std::streamsize)s.size()); nb.put('\0'); }
            std::vector<uint64_t> t, w; G.pack(t, w);
            dump(d + "/text.u64", t.data(), t.size()); dump(d + "/wild.u64", w.data(), w.size());
            return 0;
        }
        if (cmd == "reads" && argc == 6) {
            bool fq = atoi(argv[3]); int qoff = atoi(argv[4]); std::string d = argv[5];
            uint64_t cnt = ReadReader::countPatterns(argv[2], fq);
            int det = fq ? ReadReader::getOffset(argv[2]) : 0;
            ReadReader rr(argv[2], fq, qoff ? qoff : det);
            ReadBlock all, b; all.clear();
            std::ofstream ids((d + "/ids.txt").c_str());
            std::ofstream idb((d + "/ids.bin").c_str(), std::ios::binary);
            while (rr.fillBlock(b, 7, true)) { // tiny blocks: exercises the block boundaries
                for (uint64_t i = 0; i < b.size(); ++i) {
                    all.bases.insert(all.bases.end(), b.bases.begin() + b.offsets[i], b.bases.begin() + b.offsets[i + 1]);
                    all.qual.insert(all.qual.end(), b.qual.begin() + b.offsets[i], b.qual.begin() + b.offsets[i + 1]);
                    all.offsets.push_back(all.bases.size());
                    ids << b.ids[i] << "\n";
                    idb.write(b.ids[i].data(), (std::streamsize)b.ids[i].size()); idb.put('\0');
                }
            }
            dump(d + "/bases.u8", all.bases.data(), all.bases.size()); dump(d + "/qual.u8", all.qual.data(), all.qual.size());
            dump(d + "/off.u64", all.offsets.data(), all.offsets.size());
            std::ofstream m((d + "/meta.txt").c_str());
            m << cnt << " " << det << "\n";
            return 0;
        }
        if (cmd == "index" && argc == 8) {
            GenomeText G; G.load(argv[2]);
            unsigned l = atoi(argv[3]); uint64_t first = strtoull(argv[4], 0, 10), mx = strtoull(argv[5], 0, 10);
            int th = atoi(argv[6]); std::string d = argv[7];
            std::vector<uint32_t> w; enumerateWindows(G.sym, l, w);
            HostIndexBlock B; buildHostIndexBlock(G.sym, w, l, first, mx, th, B);
            for (int k = 0; k < 6; ++k) {
                char nm[64];
                snprintf(nm, sizeof nm, "/l%d_sign.bin", k);
                if (B.sig_bytes == 4) dump(d + nm, B.sign32[k].data(), B.sign32[k].size()); else dump(d + nm, B.sign64[k].data(), B.sign64[k].size());
                snprintf(nm, sizeof nm, "/l%d_pos.u32", k);
                dump(d + nm, B.pos[k].data(), B.pos[k].size());
            }
            std::ofstream m((d + "/meta.txt").c_str());
            m << B.n << " " << (B.have_next ? 1 : 0) << " " << B.sig_bytes << " " << w.size() << "\n";
            return 0;
        }
        if (cmd == "fmtcheck" && argc == 3) { // fastformat::fmt_g6 against printf's %g on argv[2] floats
            const uint64_t n = strtoull(argv[2], 0, 10);
            uint64_t bad = 0, s = 0x9E3779B97F4A7C15ull;
            auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
            auto check = [&](float f) {
                char a[64], b[64];
                const int la = fastformat::fmt_g6(f, a), lb = snprintf(b, sizeof b, "%g", (double)f);
                if (la != lb || memcmp(a, b, (size_t)la)) { if (bad++ < 20) { a[la] = 0; std::cerr << "fmt_g6 " << a << " != " << b << std::endl; } }
            };
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t r = next();
                uint32_t u = (uint32_t)r;
                float f;
                memcpy(&f, &u, 4);                                       // any bit pattern (nan, inf, denormals: the fallback)
                check(f);
                check((float)((double)(int64_t)(r >> 40) / 1000.0 - 8000.0)); // score-like values with a few decimals
                check((float)(int32_t)(r >> 44));                         // integers
                const int k = (int)((r >> 32) % 21) - 5;                   // around the powers of ten, from both sides
                double pw = 1; for (int j = 0; j < (k < 0 ? -k : k); ++j) pw *= 10;
                const double c = k < 0 ? 1 / pw : pw;
                float g = (float)c;
                uint32_t gu; memcpy(&gu, &g, 4);
                gu += (uint32_t)(r % 7) - 3; memcpy(&g, &gu, 4);
                check(g); check(-g);
                check((float)(c * 9.999995)); check((float)(c * 0.9999995)); check((float)(c * 1.2345675)); check((float)(c * 999999.5 / 1e5));
            }
            std::cout << bad << std::endl;
            return bad ? 1 : 0;
        }
        if (cmd == "options") {
            RealOptions o(argc - 1, argv + 1);
            std::cout << o.textfilename << " " << o.patternfilename << " " << o.outputfilename << " " << o.seedkmax << " " << o.totalkmax << " "
                      << o.seedl << " " << o.match_unique << " " << o.scores << " " << o.qualityOffset << " " << o.filter_level << " "
                      << o.filter_mult << " " << o.fastq << " " << o.gpus << " " << o.host_index << "\n";
            return 0;
        }
        return 2;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
