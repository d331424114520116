// real.cpp -- the `real` command line on top of the C ABI (include/real_hip.h).
//
// Host side of the drop-in: keeps the reference's CLI surface (real.cpp:357-375 main,
// cpuMain :295-354), RealOptions, FASTA/FASTQ input and the 11-column TSV output
// (printMatchUnlocked, matchUniqueImplementation.cpp:252-321; matchAll inline :481-518), and
// replaces the OpenMP per-read loops of EnumerateUniqueMatches::doMatching
// (matchUniqueImplementation.cpp:1082-1489) / EnumerateAllMatches::doMatching
// (matchAllImplementation.cpp:359-538) by calls into libreal_hip.so.  Loop structure as in the
// reference: genome files -> index blocks -> all reads re-streamed per block -> output.
//
// Deliberate differences from reference quirks (SURVEY 8a "quirks"): matchAll handles FASTQ input
// (quirk 1) and writes every line (quirk 2); the exit status is non-zero on errors (quirk 6).
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "GenomeText.hpp"
#include "HostIndex.hpp"
#include "ReadReader.hpp"
#include "RealOptions.hpp"
#include "real_hip.h"

namespace {

struct Ctx {
    real_hip_ctx *h = nullptr;
    ~Ctx() { if (h) real_hip_destroy(h); }
};

void check(real_hip_ctx *h, int rc, const char *what)
{
    if (rc == REAL_HIP_OK) return;
    std::string msg = std::string(what) + ": " + real_hip_strerror(rc);
    if (h && *real_hip_last_error(h)) msg += std::string(" (") + real_hip_last_error(h) + ")";
    if (rc == REAL_HIP_E_NOMEM) throw std::bad_alloc(); // "Insufficient memory", matchUniqueImplementation.cpp:1215-1219
    throw std::runtime_error(msg);
}

std::vector<std::unique_ptr<Ctx>> makeContexts(const RealOptions &o)
{
    real_hip_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.seedl = o.seedl; p.seedkmax = o.seedkmax; p.totalkmax = o.totalkmax; p.scores = o.scores;
    p.prefix_bits = o.prefix_bits; p.table_kind = o.table_kind; p.filter_mult = o.filter_mult;
    real_hip_scoring_table(o.similarity, o.gc, o.trans, o.err, o.gcmut_bias, p.LL); // Scoring(opts...) :1115
    std::vector<std::unique_ptr<Ctx>> v;
    for (int g = 0; g < o.gpus; ++g) {
        p.device = o.device + g;
        std::unique_ptr<Ctx> c(new Ctx);
        check(nullptr, real_hip_create(&c->h, &p), "real_hip_create (is an MI355X visible? there is no CPU fallback)");
        v.push_back(std::move(c));
    }
    return v;
}

// positions per index block from the HBM budget: 6 x 8 B entries + 16 B sort workspace per window,
// bucket tables on top (the device-side analogue of matchUniqueImplementation.cpp:1221-1244)
uint64_t blockEntries(const RealOptions &o, real_hip_ctx *h, uint64_t windows)
{
    if (o.block_entries) return o.block_entries;
    uint64_t fr = 0, tot = 0;
    check(h, real_hip_device_memory(h, &fr, &tot), "real_hip_device_memory");
    const double budget = o.fracmem * (double)fr - 6.0 * 4.0 * (double)(1ull << 30) - 2e9;
    uint64_t cap = budget > 0 ? (uint64_t)(budget / 64.0) : (1u << 20);
    if (cap < (1u << 20)) cap = 1u << 20;
    return windows < cap ? windows : cap;
}

char remapChar(uint8_t c) { return c < 4 ? "ACGT"[c] : 'N'; } // acgtnMap.hpp:24-35

std::string readString(const uint8_t *m, uint64_t n, bool inverted)
{
    std::string s((size_t)n, 'N');
    if (!inverted) for (uint64_t i = 0; i < n; ++i) s[i] = remapChar(m[i]);
    else for (uint64_t i = 0; i < n; ++i) { uint8_t c = m[n - 1 - i]; s[i] = remapChar(c < 4 ? 3 - c : 4); } // transposed
    return s;
}

struct Ranges { std::vector<std::vector<std::string>> names; std::vector<std::vector<uint64_t>> starts; };

// one output line; columns as printMatchUnlocked
void formatLine(std::ostringstream &out, const std::string &id, const std::string &seq, bool scores, float score, uint64_t patl,
                bool inverted, const std::string &fragname, uint64_t pos1, unsigned errors)
{
    out << id << "\t" << seq << "\t";
    if (scores) out << score;
    out << "\t" << 1 << "\t" << "a" << "\t" << patl << "\t" << (inverted ? "-" : "+") << "\t" << fragname << "\t" << pos1 << "\t"
        << "\t" << errors << "\n";
}

// text + one index block on every device
struct Resident {
    GenomeText G;
    std::vector<uint32_t> wpos; // host index only
    std::vector<uint64_t> text2bit, wildbits;
};

void setText(const RealOptions &o, std::vector<std::unique_ptr<Ctx>> &ctx, Resident &R, unsigned fi)
{
    if (o.host_index) { R.G.pack(R.text2bit, R.wildbits); enumerateWindows(R.G.sym, o.seedl, R.wpos); }
    for (auto &c : ctx) {
        if (o.host_index)
            check(c->h, real_hip_set_text(c->h, fi, R.text2bit.data(), R.wildbits.data(), R.G.sym.size(), R.G.frag_start.data(),
                                          (uint32_t)R.G.frag_names.size()), "real_hip_set_text");
        else
            check(c->h, real_hip_set_text_symbols(c->h, fi, R.G.sym.data(), R.G.sym.size(), 0, R.G.frag_start.data(),
                                                  (uint32_t)R.G.frag_names.size()), "real_hip_set_text_symbols");
    }
}

// returns entries of the block, sets have_next
uint64_t nextBlock(const RealOptions &o, std::vector<std::unique_ptr<Ctx>> &ctx, Resident &R, uint64_t first, uint64_t n_list, bool &have_next)
{
    uint64_t n = 0;
    if (o.host_index) {
        HostIndexBlock B;
        buildHostIndexBlock(R.G.sym, R.wpos, o.seedl, first, n_list, (int)o.sort_threads, B);
        const void *sg[6]; const uint32_t *ps[6];
        for (int k = 0; k < 6; ++k) { sg[k] = B.sign_ptr(k); ps[k] = B.pos[k].data(); }
        for (auto &c : ctx) check(c->h, real_hip_set_index_block(c->h, B.n, sg, ps), "real_hip_set_index_block");
        n = B.n; have_next = B.have_next;
    } else {
        for (auto &c : ctx) {
            int hn = 0;
            check(c->h, real_hip_build_index_block(c->h, first, n_list, &n, &hn), "real_hip_build_index_block");
            have_next = hn != 0;
        }
    }
    std::cerr << "Obtained " << n << " fragments of size " << o.seedl << std::endl; // ListSetBlockReader.hpp:36
    return n;
}

// The read file as raw text, cut into chunks of whole records for real_hip_parse_reads: a chunk ends behind a
// newline whose index is a multiple of the lines per record (4 FASTQ, 2 FASTA); only text in that form parses.
class RawChunker {
public:
    RawChunker(const std::string &fn, bool fastq, size_t chunk_bytes) : lpr_(fastq ? 4 : 2), cap_(chunk_bytes)
    {
        f_ = fopen(fn.c_str(), "rb");
        if (!f_) throw std::runtime_error("Unable to open pattern file.");
    }
    ~RawChunker() { if (f_) fclose(f_); }
    // false at the end of the file; text = whole records
    bool next(std::vector<char> &text)
    {
        text.swap(carry_);
        carry_.clear();
        const size_t have = text.size();
        text.resize(cap_);
        size_t got = eof_ ? 0 : fread(text.data() + have, 1, cap_ - have, f_);
        if (have + got < cap_) eof_ = true;
        text.resize(have + got);
        if (text.empty()) return false;
        if (eof_) return true; // the rest of the file (the parser accepts a last line without newline)
        size_t lines = 0, cut = 0;
        for (const char *p = text.data(), *e = p + text.size(); (p = (const char *)memchr(p, '\n', e - p)); ++p)
            if (++lines % lpr_ == 0) cut = (size_t)(p - text.data()) + 1;
        if (!cut) throw std::runtime_error("a read record longer than the text chunk");
        carry_.assign(text.begin() + cut, text.end());
        text.resize(cut);
        return true;
    }
private:
    FILE *f_ = nullptr;
    size_t lpr_, cap_;
    bool eof_ = false;
    std::vector<char> carry_;
};

real_hip_batch makeBatch(const ReadBlock &b)
{
    real_hip_batch rb;
    memset(&rb, 0, sizeof rb);
    rb.struct_size = sizeof rb; rb.on_device = 0; rb.n_reads = b.size();
    rb.bases = b.bases.data(); rb.qual = b.qual.data(); rb.offsets = b.offsets.data();
    return rb;
}

// ---- EnumerateUniqueMatches::doMatching -------------------------------------------------
int matchUnique(const RealOptions &o)
{
    const uint64_t numpat = ReadReader::countPatterns(o.patternfilename, o.fastq); // :1094
    std::cerr << "number of reads " << numpat << std::endl;
    int qoff = o.fastq ? (o.qualityOffset ? (int)o.qualityOffset : ReadReader::getOffset(o.patternfilename)) : 0;
    if (o.fastq && !qoff) throw std::runtime_error("Unable to automatically detect FastQ quality format."); // :1112
    std::vector<uint64_t> info(numpat, 0);                     // uniqueinfo(numpat), :1097
    std::vector<float> score(o.scores ? numpat : 0, -FLT_MAX); // UniqueMatchInfo.hpp:191
    std::vector<std::string> files;
    getFileList(o.textfilename, files);
    if (files.empty()) throw std::runtime_error("no .fa text file found at " + o.textfilename);
    if (files.size() > 64) throw std::runtime_error("more than 64 text files (6 bits of file id, UniqueMatchInfo.hpp:31)");
    auto ctx = makeContexts(o);
    Ranges RS;
    for (unsigned fi = 0; fi < files.size(); ++fi) {
        std::cerr << "Processing file " << files[fi] << ((fi + 1 == files.size()) ? " (last processed file)" : "") << std::endl;
        Resident R;
        R.G.load(files[fi]);
        RS.names.push_back(R.G.frag_names); RS.starts.push_back(R.G.frag_start);
        setText(o, ctx, R, fi);
        const uint64_t nwin_upper = R.G.sym.size();
        const uint64_t n_list = blockEntries(o, ctx[0]->h, nwin_upper ? nwin_upper : 1);
        uint64_t first = 0;
        bool have_next = true;
        while (have_next) {
            const uint64_t n = nextBlock(o, ctx, R, first, n_list, have_next);
            if (!n) break;
            first += n;
            uint64_t handled = 0;
            bool parsed_on_device = false;
            if (o.gpuparse) {
                // f2: the file goes to the device as text; one chunk per context and round.  Text that is not in
                // one-line-per-field form is refused by the library and read by the host reader below.
                RawChunker rc(o.patternfilename, o.fastq, (size_t)256 << 20);
                std::vector<std::vector<char>> txt(ctx.size());
                uint64_t next_id = 0;
                parsed_on_device = true;
                while (parsed_on_device) {
                    size_t used = 0;
                    for (; used < ctx.size(); ++used)
                        if (!rc.next(txt[used])) break;
                    if (!used) break;
                    std::vector<real_hip_parsed> pr(used);
                    std::vector<int> prc(used, 0);
                    {
                        std::vector<std::thread> th;
                        for (size_t g = 0; g < used; ++g)
                            th.emplace_back([&, g]() {
                                prc[g] = real_hip_parse_reads(ctx[g]->h, txt[g].data(), txt[g].size(), 0, o.fastq ? 1 : 0, qoff, &pr[g]);
                            });
                        for (auto &t : th) t.join();
                    }
                    for (size_t g = 0; g < used; ++g) {
                        if (prc[g] == REAL_HIP_E_UNSUPPORTED && handled == 0) { parsed_on_device = false; break; }
                        check(ctx[g]->h, prc[g], "real_hip_parse_reads");
                    }
                    if (!parsed_on_device) break;
                    std::vector<uint64_t> first(used);
                    for (size_t g = 0; g < used; ++g) { first[g] = next_id; next_id += pr[g].n_reads; }
                    if (next_id > numpat) throw std::runtime_error("device parser found more reads than countPatterns");
                    std::vector<std::thread> th;
                    std::vector<std::string> errs(used);
                    for (size_t g = 0; g < used; ++g)
                        th.emplace_back([&, g]() {
                            try {
                                real_hip_batch rb;
                                memset(&rb, 0, sizeof rb);
                                rb.struct_size = sizeof rb; rb.on_device = 2; rb.n_reads = pr[g].n_reads;
                                rb.bases = pr[g].bases; rb.qual = pr[g].qual; rb.offsets = pr[g].offsets; rb.max_patl = pr[g].max_patl;
                                if (rb.n_reads)
                                    check(ctx[g]->h, real_hip_match_unique(ctx[g]->h, &rb, info.data() + first[g],
                                                                           o.scores ? score.data() + first[g] : nullptr),
                                          "real_hip_match_unique");
                            } catch (const std::exception &e) { errs[g] = e.what(); }
                        });
                    for (auto &t : th) t.join();
                    for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
                    handled = next_id;
                    std::cerr << "\r                                                              \r" << (double)handled / (numpat ? numpat : 1) << std::flush;
                }
                if (parsed_on_device && handled != numpat) throw std::runtime_error("device parser and countPatterns disagree on the number of reads");
            }
            if (!parsed_on_device) {
            ReadReader rr(o.patternfilename, o.fastq, qoff);     // the whole read set is re-streamed per block, :1260
            std::vector<ReadBlock> blk(ctx.size());
            while (true) {
                size_t used = 0;
                for (; used < ctx.size(); ++used)
                    if (!rr.fillBlock(blk[used], o.batch_reads, false)) break;
                if (!used) break;
                std::vector<std::thread> th;
                std::vector<std::string> errs(used);
                for (size_t g = 0; g < used; ++g)
                    th.emplace_back([&, g]() {
                        try {
                            real_hip_batch rb = makeBatch(blk[g]);
                            check(ctx[g]->h, real_hip_match_unique(ctx[g]->h, &rb, info.data() + blk[g].first_id,
                                                                   o.scores ? score.data() + blk[g].first_id : nullptr),
                                  "real_hip_match_unique");
                        } catch (const std::exception &e) { errs[g] = e.what(); }
                    });
                for (auto &t : th) t.join();
                for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
                for (size_t g = 0; g < used; ++g) handled += blk[g].size();
                std::cerr << "\r                                                              \r" << (double)handled / (numpat ? numpat : 1) << std::flush;
            }
            }
            std::cerr << std::endl;
        }
    }
    std::cerr << "All done." << std::endl;
    // output, in read order (PatternIdReader re-stream, :1438-1486)
    FILE *out = (o.outputfilename == "-") ? stdout : fopen(o.outputfilename.c_str(), "wb");
    if (!out) throw std::runtime_error("cannot open output file " + o.outputfilename);
    ReadReader rr(o.patternfilename, o.fastq, qoff);
    ReadBlock b;
    uint64_t unique = 0;
    while (rr.fillBlock(b, 1u << 16, true)) {
        std::ostringstream os;
        for (uint64_t i = 0; i < b.size(); ++i) {
            const uint64_t rec = info[b.first_id + i];
            const unsigned st = (unsigned)(rec >> 61);
            if (st != 1 && st != 2) continue; // NoMatch / NonUnique / Gapped print nothing
            const unsigned frag = (rec >> 45) & 0xffff, errors = (rec >> 41) & 15, file = (rec >> 35) & 63;
            const uint64_t pos = rec & ((1ull << 35) - 1);
            const uint64_t lo = b.offsets[i], patl = b.offsets[i + 1] - lo;
            formatLine(os, b.ids[i], readString(&b.bases[lo], patl, st == 2), o.scores, o.scores ? score[b.first_id + i] : 0.f, patl,
                       st == 2, RS.names[file][frag], pos - RS.starts[file][frag] + 1, errors);
            unique++;
        }
        const std::string s = os.str();
        fwrite(s.data(), 1, s.size(), out);
    }
    if (out != stdout) fclose(out);
    std::cerr << "unique: " << unique << std::endl; // :1488
    return EXIT_SUCCESS;
}

// ---- EnumerateAllMatches::doMatching ----------------------------------------------------
int matchAll(const RealOptions &o)
{
    int qoff = o.fastq ? (o.qualityOffset ? (int)o.qualityOffset : ReadReader::getOffset(o.patternfilename)) : 0;
    if (o.fastq && !qoff) throw std::runtime_error("Unable to automatically detect FastQ quality format.");
    std::vector<std::string> files;
    getFileList(o.textfilename, files);
    if (files.empty()) throw std::runtime_error("no .fa text file found at " + o.textfilename);
    auto ctx = makeContexts(o);
    FILE *out = (o.outputfilename == "-") ? stdout : fopen(o.outputfilename.c_str(), "wb");
    if (!out) throw std::runtime_error("cannot open output file " + o.outputfilename);
    for (unsigned fi = 0; fi < files.size(); ++fi) {
        std::cerr << "Processing file " << files[fi] << std::endl;
        Resident R;
        R.G.load(files[fi]);
        setText(o, ctx, R, fi);
        const uint64_t n_list = blockEntries(o, ctx[0]->h, R.G.sym.size() ? R.G.sym.size() : 1);
        uint64_t first = 0;
        bool have_next = true;
        while (have_next) {
            const uint64_t n = nextBlock(o, ctx, R, first, n_list, have_next);
            if (!n) break;
            first += n;
            ReadReader rr(o.patternfilename, o.fastq, qoff); // hits are emitted per genome block, :451-535
            ReadBlock b;
            std::vector<real_hip_hit> hits(1u << 20);
            std::vector<uint64_t> hoff;
            while (rr.fillBlock(b, o.batch_reads, true)) {
                real_hip_batch rb = makeBatch(b);
                hoff.assign(b.size() + 1, 0);
                uint64_t nh = 0;
                int rc = real_hip_match_all(ctx[0]->h, &rb, hits.data(), hits.size(), &nh, hoff.data());
                if (rc == REAL_HIP_E_OVERFLOW) { // retry with the size the library reports
                    hits.resize(nh + 16);
                    rc = real_hip_match_all(ctx[0]->h, &rb, hits.data(), hits.size(), &nh, hoff.data());
                }
                check(ctx[0]->h, rc, "real_hip_match_all");
                std::ostringstream os;
                for (uint64_t i = 0; i < b.size(); ++i) {
                    const uint64_t lo = b.offsets[i], patl = b.offsets[i + 1] - lo;
                    for (uint64_t k = hoff[i]; k < hoff[i + 1]; ++k) {
                        const real_hip_hit &M = hits[k];
                        formatLine(os, b.ids[i], readString(&b.bases[lo], patl, M.inverted), o.scores, M.score, patl, M.inverted,
                                   R.G.frag_names[M.frag], (uint64_t)M.pos - R.G.frag_start[M.frag] + 1, M.k);
                    }
                }
                const std::string s = os.str();
                fwrite(s.data(), 1, s.size(), out);
            }
        }
    }
    if (out != stdout) fclose(out);
    std::cerr << "All done." << std::endl;
    return EXIT_SUCCESS;
}

} // namespace

int main(int argc, char *argv[])
{
    std::cerr << "This is real (MI355X read-matching path), ABI " << real_hip_abi_version() << "." << std::endl;
    try {
        RealOptions opts(argc, argv);
        return opts.match_unique ? matchUnique(opts) : matchAll(opts);
    } catch (const std::bad_alloc &) {
        std::cerr << "Insufficient memory." << std::endl;
        return EXIT_FAILURE;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
}
