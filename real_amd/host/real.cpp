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
// The read file travels as text: chunks of whole records are read into pinned buffers by a prefetch
// thread, parsed and matched on the device(s) (real_hip_parse_reads + real_hip_match_unique), and in the
// output pass parsed again for the id / sequence spans, from which the host formats the lines with all
// its cores and writes them with large sequential writes.  A file (or a later part of one) that is not in
// one-line-per-field form is read by the host reader from that record on.
//
// Deliberate differences from reference quirks (SURVEY 8a "quirks"): matchAll handles FASTQ input
// (quirk 1) and writes every line (quirk 2); the exit status is non-zero on errors (quirk 6).
#include <fcntl.h>
#include <omp.h>
#include <unistd.h>

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "FastFormat.hpp"
#include "GenomeText.hpp"
#include "HostIndex.hpp"
#include "ReadReader.hpp"
#include "RealOptions.hpp"
#include "real_hip.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// where the wall time went; printed as one "timing:" line on stderr at the end (bench_support/cli_midsize.py reads it)
struct Timers {
    double genome = 0, index = 0, read = 0, parse = 0, match = 0, format = 0, write = 0, total = 0;
    uint64_t reads = 0, lines = 0, out_bytes = 0;
    void print() const
    {
        fprintf(stderr, "timing: genome_load_s=%.3f index_s=%.3f read_file_s=%.3f parse_s=%.3f match_s=%.3f format_s=%.3f write_s=%.3f total_s=%.3f reads=%llu lines=%llu out_bytes=%llu\n",
                genome, index, read, parse, match, format, write, total, (unsigned long long)reads, (unsigned long long)lines, (unsigned long long)out_bytes);
    }
};

struct Ctx {
    real_hip_ctx *h = nullptr;
    ~Ctx() { if (h) real_hip_destroy(h); }
};
typedef std::vector<std::unique_ptr<Ctx>> CtxVec;

void check(real_hip_ctx *h, int rc, const char *what)
{
    if (rc == REAL_HIP_OK) return;
    std::string msg = std::string(what) + ": " + real_hip_strerror(rc);
    if (h && *real_hip_last_error(h)) msg += std::string(" (") + real_hip_last_error(h) + ")";
    if (rc == REAL_HIP_E_NOMEM) throw std::bad_alloc(); // "Insufficient memory", matchUniqueImplementation.cpp:1215-1219
    throw std::runtime_error(msg);
}

CtxVec makeContexts(const RealOptions &o)
{
    real_hip_params p;
    memset(&p, 0, sizeof p);
    p.struct_size = sizeof p;
    p.seedl = o.seedl; p.seedkmax = o.seedkmax; p.totalkmax = o.totalkmax; p.scores = o.scores;
    p.prefix_bits = o.prefix_bits; p.table_kind = o.table_kind; p.filter_mult = o.filter_mult;
    real_hip_scoring_table(o.similarity, o.gc, o.trans, o.err, o.gcmut_bias, p.LL); // Scoring(opts...) :1115
    CtxVec v;
    for (int g = 0; g < o.gpus; ++g) {
        p.device = o.gpus_share_device ? o.device : o.device + g;
        std::unique_ptr<Ctx> c(new Ctx);
        check(nullptr, real_hip_create(&c->h, &p), "real_hip_create (is an MI355X visible? there is no CPU fallback)");
        v.push_back(std::move(c));
    }
    return v;
}

// run f(g) for g in [0, n) on one host thread per context; the first exception is rethrown
template <class F>
void onEach(size_t n, F f)
{
    if (n == 1) { f((size_t)0); return; }
    std::vector<std::thread> th;
    std::vector<std::string> errs(n);
    std::vector<int> nomem(n, 0);
    for (size_t g = 0; g < n; ++g)
        th.emplace_back([&, g]() {
            try { f(g); } catch (const std::bad_alloc &) { nomem[g] = 1; } catch (const std::exception &e) { errs[g] = e.what(); if (errs[g].empty()) errs[g] = "error"; }
        });
    for (auto &t : th) t.join();
    for (size_t g = 0; g < n; ++g) if (nomem[g]) throw std::bad_alloc();
    for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
}

// positions per index block from the HBM budget: the tables of a block plus the transients of its build
// (the device-side analogue of matchUniqueImplementation.cpp:1221-1244)
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

struct Ranges { std::vector<std::vector<std::string>> names; std::vector<std::vector<uint64_t>> starts; };

// ---- output lines (printMatchUnlocked, matchUniqueImplementation.cpp:252-321) --------------------------
// id \t sequence as matched \t score|"" \t 1 \t a \t patl \t +|- \t fragment name \t 1-based position \t "" \t errors \n
// Written straight into the thread's buffer: table lookups for the sequence, hand-rolled decimal numbers, and the score by
// fastformat::fmt_g6 -- the digits of printf's %g, which is what operator<<(float) prints, from integer arithmetic.
struct SeqTables {
    char fwd[256], rc[256], map_fwd[5], map_rc[5];
    SeqTables()
    {
        for (int c = 0; c < 256; ++c) { fwd[c] = 'N'; rc[c] = 'N'; }   // anything but ACGT (lowercase too) maps to 4 and prints as N
        fwd['A'] = 'A'; fwd['C'] = 'C'; fwd['G'] = 'G'; fwd['T'] = 'T';
        rc['A'] = 'T'; rc['C'] = 'G'; rc['G'] = 'C'; rc['T'] = 'A';
        memcpy(map_fwd, "ACGTN", 5); memcpy(map_rc, "TGCAN", 5);       // remapChar, acgtnMap.hpp:24-35; transposed: 3 - c
    }
};
const SeqTables kSeq;

inline char *putUint(char *p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
// room for one line behind the current end of out; returns the write position
inline char *lineRoom(std::string &out, size_t bytes)
{
    const size_t at = out.size();
    if (out.capacity() < at + bytes) out.reserve(std::max(out.capacity() * 2, at + bytes));
    out.resize(at + bytes);
    return &out[at];
}
inline char *putTail(char *p, bool scores, float score, uint64_t patl, bool inverted, const std::string &fragname, uint64_t pos1, unsigned errors)
{
    *p++ = '\t';
    if (scores) p += fastformat::fmt_g6(score, p); // operator<<(float): %g, six significant digits
    memcpy(p, "\t1\ta\t", 5); p += 5;
    p = putUint(p, patl);
    *p++ = '\t'; *p++ = inverted ? '-' : '+'; *p++ = '\t';
    memcpy(p, fragname.data(), fragname.size()); p += fragname.size();
    *p++ = '\t';
    p = putUint(p, pos1);
    *p++ = '\t'; *p++ = '\t';
    p = putUint(p, errors);
    *p++ = '\n';
    return p;
}
// one line; the sequence column from the characters of the read file (seq_text) or from mapped symbols (seq_mapped)
inline void appendLine(std::string &out, const char *id, size_t idlen, const char *seq_text, const uint8_t *seq_mapped, uint64_t patl, bool scores,
                       float score, bool inverted, const std::string &fragname, uint64_t pos1, unsigned errors)
{
    const size_t at = out.size();
    char *p = lineRoom(out, idlen + patl + fragname.size() + 112), *p0 = p;
    memcpy(p, id, idlen); p += idlen;
    *p++ = '\t';
    if (seq_text) {
        if (!inverted) for (uint64_t i = 0; i < patl; ++i) p[i] = kSeq.fwd[(unsigned char)seq_text[i]];
        else for (uint64_t i = 0; i < patl; ++i) p[i] = kSeq.rc[(unsigned char)seq_text[patl - 1 - i]];
    } else {
        if (!inverted) for (uint64_t i = 0; i < patl; ++i) p[i] = kSeq.map_fwd[seq_mapped[i] < 4 ? seq_mapped[i] : 4];
        else for (uint64_t i = 0; i < patl; ++i) { const uint8_t c = seq_mapped[patl - 1 - i]; p[i] = kSeq.map_rc[c < 4 ? c : 4]; }
    }
    p += patl;
    p = putTail(p, scores, score, patl, inverted, fragname, pos1, errors);
    out.resize(at + (size_t)(p - p0));
}

struct Record { unsigned st, frag, errors, file; uint64_t pos; };
inline Record unpack(uint64_t rec)
{
    Record r;
    r.st = (unsigned)(rec >> 61); r.frag = (rec >> 45) & 0xffff; r.errors = (rec >> 41) & 15; r.file = (rec >> 35) & 63; r.pos = rec & ((1ull << 35) - 1);
    return r;
}

// The lines of reads [0, n) of one block, formatted by all host threads (each a contiguous range of reads into its
// own buffer) and written in read order with one large write per buffer.  line(i, out) appends read i's line(s).
struct alignas(128) LineBuf { std::string s; }; // (a cache line pair of its own: the threads update their string's length line by line)

template <class LineFn>
void formatAndWrite(uint64_t n, FILE *out, Timers &T, LineFn line)
{
    const int nt = std::max(1, omp_get_max_threads());
    static std::vector<LineBuf> buf; // (kept across calls: the pages of a buffer are touched once, not once per block)
    if ((int)buf.size() < nt) buf = std::vector<LineBuf>((size_t)nt);
    const double t0 = now_s();
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const uint64_t lo = n * (uint64_t)t / nt, hi = n * (uint64_t)(t + 1) / nt;
        std::string b;
        b.swap(buf[(size_t)t].s); // (worked on as a local: its length and pointer live in registers / this thread's stack)
        b.clear();
        for (uint64_t i = lo; i < hi; ++i) line(i, b);
        b.swap(buf[(size_t)t].s);
    }
    const double t1 = now_s();
    for (int t = 0; t < nt; ++t) {
        const std::string &b = buf[(size_t)t].s;
        if (!b.empty() && fwrite(b.data(), 1, b.size(), out) != b.size()) throw std::runtime_error("write to the output file failed");
        T.out_bytes += b.size();
    }
    T.format += t1 - t0; T.write += now_s() - t1;
}

// text + one index block on every device
struct Resident {
    GenomeText G;
    std::vector<uint32_t> wpos; // host index only
    std::vector<uint64_t> text2bit, wildbits;
};

void setText(const RealOptions &o, CtxVec &ctx, Resident &R, unsigned fi)
{
    if (o.host_index) { R.G.pack(R.text2bit, R.wildbits); enumerateWindows(R.G.sym, o.seedl, R.wpos); }
    onEach(ctx.size(), [&](size_t g) {
        real_hip_ctx *h = ctx[g]->h;
        if (o.host_index)
            check(h, real_hip_set_text(h, fi, R.text2bit.data(), R.wildbits.data(), R.G.sym.size(), R.G.frag_start.data(),
                                       (uint32_t)R.G.frag_names.size()), "real_hip_set_text");
        else
            check(h, real_hip_set_text_symbols(h, fi, R.G.sym.data(), R.G.sym.size(), 0, R.G.frag_start.data(),
                                               (uint32_t)R.G.frag_names.size()), "real_hip_set_text_symbols");
    });
}

// returns entries of the block, sets have_next
uint64_t nextBlock(const RealOptions &o, CtxVec &ctx, Resident &R, uint64_t first, uint64_t n_list, bool &have_next)
{
    uint64_t n = 0;
    if (o.host_index) {
        HostIndexBlock B;
        buildHostIndexBlock(R.G.sym, R.wpos, o.seedl, first, n_list, (int)o.sort_threads, B);
        const void *sg[6]; const uint32_t *ps[6];
        for (int k = 0; k < 6; ++k) { sg[k] = B.sign_ptr(k); ps[k] = B.pos[k].data(); }
        onEach(ctx.size(), [&](size_t g) { check(ctx[g]->h, real_hip_set_index_block(ctx[g]->h, B.n, sg, ps), "real_hip_set_index_block"); });
        n = B.n; have_next = B.have_next;
    } else {
        std::vector<uint64_t> ns(ctx.size(), 0);
        std::vector<int> hn(ctx.size(), 0);
        onEach(ctx.size(), [&](size_t g) {
            check(ctx[g]->h, real_hip_build_index_block(ctx[g]->h, first, n_list, &ns[g], &hn[g]), "real_hip_build_index_block");
        });
        n = ns[0]; have_next = hn[0] != 0;
    }
    std::cerr << "Obtained " << n << " fragments of size " << o.seedl << std::endl; // ListSetBlockReader.hpp:36
    return n;
}

// ---- the read file as text chunks ----------------------------------------------------------------------
// Chunks of whole records for real_hip_parse_reads: a chunk ends behind a newline whose index is a multiple of the
// lines per record (4 FASTQ, 2 FASTA) -- only text in one-line-per-field form parses, and only for such text the cuts
// are record boundaries.  Buffers are pinned (real_hip_host_alloc): they cross PCIe by DMA.  The next chunk is read by
// a prefetch thread while the devices work on the current ones.
struct Chunk {
    char *text = nullptr;
    size_t cap = 0, size = 0;
    uint64_t file_offset = 0; // of text[0]
};

class RawChunker {
public:
    RawChunker(const std::string &fn, bool fastq, size_t chunk_bytes, size_t n_buffers) : lpr_(fastq ? 4 : 2)
    {
        fd_ = open(fn.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("Unable to open pattern file.");
        for (size_t i = 0; i < n_buffers; ++i) {
            Chunk c;
            c.text = (char *)real_hip_host_alloc(chunk_bytes);
            if (!c.text) throw std::bad_alloc();
            c.cap = chunk_bytes;
            pool_.push_back(c);
            free_.push_back((int)i);
        }
        prefetch();
    }
    ~RawChunker()
    {
        if (next_.valid()) next_.wait();
        for (auto &c : pool_) real_hip_host_free(c.text);
        if (fd_ >= 0) close(fd_);
    }
    // the next chunk (the caller's until it hands it back with release); false at the end of the file
    bool next(Chunk &out, double &wait_s)
    {
        if (!next_.valid()) prefetch();
        if (!next_.valid()) return false;
        const double t0 = now_s();
        const int got = next_.get(); // (an exception of the reader thread surfaces here)
        wait_s += now_s() - t0;
        if (got < 0) { done_ = true; return false; }
        out = pool_[(size_t)got];
        prefetch();
        return true;
    }
    void release(const Chunk &c)
    {
        for (size_t i = 0; i < pool_.size(); ++i)
            if (pool_[i].text == c.text) free_.push_back((int)i);
        prefetch();
    }
private:
    // (only the thread that owns the chunker calls this, and only while no read is in flight: the reader thread is the
    // only one that touches the file state then)
    void prefetch()
    {
        if (done_ || next_.valid() || free_.empty()) return;
        const int id = free_.back();
        free_.pop_back();
        next_ = std::async(std::launch::async, [this, id]() { return fill(id) ? id : -1; });
    }
    // reads the next chunk with kReaders threads (pread of a slice each, then the newlines of the slice are counted
    // while it is hot), cuts it behind the last newline whose index is a multiple of the lines per record
    bool fill(int id)
    {
        Chunk &c = pool_[(size_t)id];
        const size_t have = carry_.size();
        if (have > c.cap) throw std::runtime_error("a read record longer than the text chunk");
        if (have) memcpy(c.text, carry_.data(), have);
        c.file_offset = offset_;
        carry_.clear();
        const size_t want = eof_ ? 0 : c.cap - have;
        constexpr int kReaders = 4;
        size_t got_s[kReaders] = {0, 0, 0, 0}, nl_s[kReaders] = {0, 0, 0, 0};
        bool err = false;
        auto slice = [&](int k) {
            const size_t lo = want * (size_t)k / kReaders, hi = want * (size_t)(k + 1) / kReaders;
            size_t done = 0;
            while (lo + done < hi) {
                const ssize_t r = pread(fd_, c.text + have + lo + done, hi - lo - done, (off_t)(fpos_ + lo + done));
                if (r < 0) { err = true; break; }
                if (r == 0) break; // end of file
                done += (size_t)r;
            }
            got_s[k] = done;
            size_t nl = 0;
            const char *b = c.text + have + lo, *e = b + done;
            if (k == 0) b = c.text; // (the carry belongs to the first slice)
            for (const char *q = b; (q = (const char *)memchr(q, '\n', (size_t)(e - q))); ++q) nl++;
            nl_s[k] = nl;
        };
        if (want) {
            std::thread th[kReaders - 1];
            for (int k = 1; k < kReaders; ++k) th[k - 1] = std::thread(slice, k);
            slice(0);
            for (auto &t : th) t.join();
        } else {
            slice(0); // (nothing to read: the newlines of the carry)
        }
        if (err) throw std::runtime_error("reading the pattern file failed");
        size_t got = 0, lines = 0;
        for (int k = 0; k < kReaders; ++k) { got += got_s[k]; lines += nl_s[k]; } // (a short slice = the end of the file: the slices behind it are empty)
        fpos_ += got;
        if (got < want || !want) eof_ = true;
        c.size = have + got;
        if (!c.size) return false;
        if (!eof_) {
            // behind newline number lines - lines % lpr: walk back over the lines % lpr newlines of the incomplete record
            size_t back = lines % lpr_, cut = c.size;
            const char *q = c.text + c.size;
            for (size_t k = 0; k <= back; ++k) {
                q = (const char *)memrchr(c.text, '\n', (size_t)(q - c.text));
                if (!q) { cut = 0; break; }
                cut = (size_t)(q - c.text) + 1;
            }
            if (lines < lpr_ || !cut) throw std::runtime_error("a read record longer than the text chunk");
            carry_.assign(c.text + cut, c.text + c.size);
            c.size = cut;
        }
        offset_ += c.size;
        return true;
    }
    int fd_ = -1;
    uint64_t fpos_ = 0; // file position behind the bytes read so far
    size_t lpr_;
    bool eof_ = false, done_ = false;
    uint64_t offset_ = 0;
    std::vector<char> carry_;
    std::vector<Chunk> pool_;
    std::vector<int> free_;
    std::future<int> next_;
};

real_hip_batch makeBatch(const ReadBlock &b)
{
    real_hip_batch rb;
    memset(&rb, 0, sizeof rb);
    rb.struct_size = sizeof rb; rb.on_device = 0; rb.n_reads = b.size();
    rb.bases = b.bases.data(); rb.qual = b.qual.data(); rb.offsets = b.offsets.data();
    return rb;
}
real_hip_batch makeBatch(const real_hip_parsed &p)
{
    real_hip_batch rb;
    memset(&rb, 0, sizeof rb);
    rb.struct_size = sizeof rb; rb.on_device = 2; rb.n_reads = p.n_reads;
    rb.bases = p.bases; rb.qual = p.qual; rb.offsets = p.offsets; rb.max_patl = p.max_patl;
    return rb;
}

// One pass over the read file.  Rounds of up to one chunk per context: parsed on the devices, then onChunks(first_id,
// chunks, parsed) sees them (in file order).  From the first chunk a device parser refuses (text not in one-line-per-
// field form) the host reader takes over at that record: onBlocks(blocks) sees rounds of host-parsed blocks.  A file
// that is refused from its first chunk on -- or -gpuparse 0 -- is read by the host reader alone.
// Returns the number of reads seen.
template <class OnChunks, class OnBlocks>
uint64_t streamReads(const RealOptions &o, CtxVec &ctx, int qoff, bool want_ids, Timers &T, OnChunks onChunks, OnBlocks onBlocks)
{
    uint64_t next_id = 0, takeover_at = 0;
    bool host = !o.gpuparse;
    if (!host) {
        RawChunker rc(o.patternfilename, o.fastq, o.chunk_bytes, 2 * ctx.size());
        while (!host) {
            std::vector<Chunk> ch;
            for (size_t g = 0; g < ctx.size(); ++g) {
                Chunk c;
                if (!rc.next(c, T.read)) break;
                ch.push_back(c);
            }
            if (ch.empty()) break;
            std::vector<real_hip_parsed> pr(ch.size());
            std::vector<int> prc(ch.size(), 0);
            const double t0 = now_s();
            onEach(ch.size(), [&](size_t g) {
                prc[g] = real_hip_parse_reads(ctx[g]->h, ch[g].text, ch[g].size, 0, o.fastq ? 1 : 0, qoff, &pr[g]);
            });
            T.parse += now_s() - t0;
            size_t good = 0;
            for (; good < ch.size(); ++good) {
                if (prc[good] == REAL_HIP_E_UNSUPPORTED) break;
                check(ctx[good]->h, prc[good], "real_hip_parse_reads");
            }
            if (good < ch.size()) { host = true; takeover_at = ch[good].file_offset; }
            ch.resize(good); pr.resize(good);
            if (good) {
                std::vector<uint64_t> first(good);
                for (size_t g = 0; g < good; ++g) { first[g] = next_id; next_id += pr[g].n_reads; }
                onChunks(first, ch, pr);
            }
            for (auto &c : ch) rc.release(c);
        }
        if (host) std::cerr << "read file leaves the one-line-per-field form at byte " << takeover_at << ": the host reader takes over from there" << std::endl;
    }
    if (host) {
        ReadReader rr(o.patternfilename, o.fastq, qoff, takeover_at, next_id); // the whole read set is re-streamed per block, :1260
        std::vector<ReadBlock> blk(ctx.size());
        while (true) {
            size_t used = 0;
            const double t0 = now_s();
            for (; used < ctx.size(); ++used)
                if (!rr.fillBlock(blk[used], o.batch_reads, want_ids)) break;
            T.parse += now_s() - t0;
            if (!used) break;
            onBlocks(blk, used);
            for (size_t g = 0; g < used; ++g) next_id += blk[g].size();
        }
    }
    return next_id;
}

void progress(uint64_t handled, uint64_t numpat)
{
    std::cerr << "\r                                                              \r" << (double)handled / (numpat ? numpat : 1) << std::flush;
}

// ---- EnumerateUniqueMatches::doMatching -------------------------------------------------
int matchUnique(const RealOptions &o)
{
    Timers T;
    const double t_begin = now_s();
    int qoff = o.fastq ? (o.qualityOffset ? (int)o.qualityOffset : ReadReader::getOffset(o.patternfilename)) : 0;
    if (o.fastq && !qoff) throw std::runtime_error("Unable to automatically detect FastQ quality format."); // :1112
    // uniqueinfo(numpat), :1094-1097.  The reference counts the reads in a pass of its own; here the arrays grow with the
    // first pass over the file (records start as NoMatch / -FLT_MAX, UniqueMatchInfo.hpp:191).
    std::vector<uint64_t> info;
    std::vector<float> score;
    uint64_t numpat = 0;
    bool counted = false;
    auto grow = [&](uint64_t n) {
        if (n > info.size()) {
            const uint64_t to = std::max<uint64_t>(n, info.size() + info.size() / 2);
            info.resize(to, 0);
            if (o.scores) score.resize(to, -FLT_MAX);
        }
    };
    std::vector<std::string> files;
    getFileList(o.textfilename, files);
    if (files.empty()) throw std::runtime_error("no .fa text file found at " + o.textfilename);
    if (files.size() > 64) throw std::runtime_error("more than 64 text files (6 bits of file id, UniqueMatchInfo.hpp:31)");
    CtxVec ctx = makeContexts(o);
    Ranges RS;
    for (unsigned fi = 0; fi < files.size(); ++fi) {
        std::cerr << "Processing file " << files[fi] << ((fi + 1 == files.size()) ? " (last processed file)" : "") << std::endl;
        Resident R;
        double t0 = now_s();
        R.G.load(files[fi]);
        T.genome += now_s() - t0;
        RS.names.push_back(R.G.frag_names); RS.starts.push_back(R.G.frag_start);
        t0 = now_s();
        setText(o, ctx, R, fi);
        T.index += now_s() - t0;
        const uint64_t nwin_upper = R.G.sym.size();
        const uint64_t n_list = blockEntries(o, ctx[0]->h, nwin_upper ? nwin_upper : 1);
        uint64_t first = 0;
        bool have_next = true;
        while (have_next) {
            t0 = now_s();
            const uint64_t n = nextBlock(o, ctx, R, first, n_list, have_next);
            T.index += now_s() - t0;
            if (!n) break;
            first += n;
            const uint64_t seen = streamReads(o, ctx, qoff, false, T,
                [&](const std::vector<uint64_t> &first_id, const std::vector<Chunk> &ch, const std::vector<real_hip_parsed> &pr) {
                    grow(first_id.back() + pr.back().n_reads);
                    const double tm = now_s();
                    onEach(ch.size(), [&](size_t g) {
                        real_hip_batch rb = makeBatch(pr[g]);
                        rb.fresh = !counted; // first pass over the reads: the records start on the device
                        if (rb.n_reads)
                            check(ctx[g]->h, real_hip_match_unique(ctx[g]->h, &rb, info.data() + first_id[g], o.scores ? score.data() + first_id[g] : nullptr),
                                  "real_hip_match_unique");
                    });
                    T.match += now_s() - tm;
                    if (counted) progress(first_id.back() + pr.back().n_reads, numpat);
                },
                [&](std::vector<ReadBlock> &blk, size_t used) {
                    grow(blk[used - 1].first_id + blk[used - 1].size());
                    const double tm = now_s();
                    onEach(used, [&](size_t g) {
                        real_hip_batch rb = makeBatch(blk[g]);
                        rb.fresh = !counted;
                        if (rb.n_reads)
                            check(ctx[g]->h, real_hip_match_unique(ctx[g]->h, &rb, info.data() + blk[g].first_id, o.scores ? score.data() + blk[g].first_id : nullptr),
                                  "real_hip_match_unique");
                    });
                    T.match += now_s() - tm;
                    if (counted) progress(blk[used - 1].first_id + blk[used - 1].size(), numpat);
                });
            if (!counted) { numpat = seen; counted = true; std::cerr << "number of reads " << numpat << std::endl; } // :1096
            else if (seen != numpat) throw std::runtime_error("the read file changed between two passes");
            std::cerr << std::endl;
        }
    }
    std::cerr << "All done." << std::endl;
    // output, in read order (PatternIdReader re-stream, :1438-1486)
    FILE *out = (o.outputfilename == "-") ? stdout : fopen(o.outputfilename.c_str(), "wb");
    if (!out) throw std::runtime_error("cannot open output file " + o.outputfilename);
    std::vector<char> obuf((size_t)8 << 20);
    setvbuf(out, obuf.data(), _IOFBF, obuf.size());
    uint64_t unique = 0;
    std::vector<uint32_t> id_start, id_len;
    std::vector<uint64_t> off;
    streamReads(o, ctx, qoff, true, T,
        [&](const std::vector<uint64_t> &first_id, const std::vector<Chunk> &ch, const std::vector<real_hip_parsed> &pr) {
            for (size_t g = 0; g < ch.size(); ++g) { // (in file order; the spans of one chunk at a time)
                const uint64_t n = pr[g].n_reads;
                if (!n) continue;
                id_start.resize(n); id_len.resize(n); off.resize(n + 1);
                const double td = now_s();
                check(ctx[g]->h, real_hip_download(ctx[g]->h, pr[g].id_start, id_start.data(), n * 4), "real_hip_download");
                check(ctx[g]->h, real_hip_download(ctx[g]->h, pr[g].id_len, id_len.data(), n * 4), "real_hip_download");
                check(ctx[g]->h, real_hip_download(ctx[g]->h, pr[g].offsets, off.data(), (n + 1) * 8), "real_hip_download");
                T.parse += now_s() - td;
                const char *text = ch[g].text;
                const uint64_t base = first_id[g];
                formatAndWrite(n, out, T, [&](uint64_t i, std::string &b) {
                    const Record r = unpack(info[base + i]);
                    if (r.st != 1 && r.st != 2) return; // NoMatch / NonUnique / Gapped print nothing
                    // the id is everything behind the marker up to the newline (a '\r' in front of it included, as the
                    // reference's reader keeps it); the sequence is the next line
                    uint64_t il = id_len[i];
                    if (text[id_start[i] + il] == '\r') il++;
                    const uint64_t patl = off[i + 1] - off[i];
                    appendLine(b, text + id_start[i], il, text + id_start[i] + il + 1, nullptr, patl, o.scores, o.scores ? score[base + i] : 0.f, r.st == 2,
                               RS.names[r.file][r.frag], r.pos - RS.starts[r.file][r.frag] + 1, r.errors);
                });
            }
        },
        [&](std::vector<ReadBlock> &blk, size_t used) {
            for (size_t g = 0; g < used; ++g) {
                const ReadBlock &b = blk[g];
                formatAndWrite(b.size(), out, T, [&](uint64_t i, std::string &s) {
                    const Record r = unpack(info[b.first_id + i]);
                    if (r.st != 1 && r.st != 2) return;
                    const uint64_t lo = b.offsets[i], patl = b.offsets[i + 1] - lo;
                    appendLine(s, b.ids[i].data(), b.ids[i].size(), nullptr, &b.bases[lo], patl, o.scores, o.scores ? score[b.first_id + i] : 0.f, r.st == 2,
                               RS.names[r.file][r.frag], r.pos - RS.starts[r.file][r.frag] + 1, r.errors);
                });
            }
        });
    if (fflush(out) != 0) throw std::runtime_error("write to the output file failed");
    if (out != stdout) fclose(out);
    // (counted here, once, and not line by line inside the formatter: sixteen threads bumping neighbouring counters
    // cost more than formatting the lines)
#pragma omp parallel for reduction(+ : unique) schedule(static)
    for (uint64_t i = 0; i < numpat; ++i) { const unsigned st = (unsigned)(info[i] >> 61); unique += (st == 1 || st == 2); }
    std::cerr << "unique: " << unique << std::endl; // :1488
    T.reads = numpat; T.lines = unique; T.total = now_s() - t_begin;
    T.print();
    return EXIT_SUCCESS;
}

// ---- EnumerateAllMatches::doMatching ----------------------------------------------------
// Hits are emitted per genome block (matchAllImplementation.cpp:451-535).  The read blocks of a round go to the
// contexts in parallel; their lines are written in read order.
int matchAll(const RealOptions &o)
{
    Timers T;
    const double t_begin = now_s();
    int qoff = o.fastq ? (o.qualityOffset ? (int)o.qualityOffset : ReadReader::getOffset(o.patternfilename)) : 0;
    if (o.fastq && !qoff) throw std::runtime_error("Unable to automatically detect FastQ quality format.");
    std::vector<std::string> files;
    getFileList(o.textfilename, files);
    if (files.empty()) throw std::runtime_error("no .fa text file found at " + o.textfilename);
    CtxVec ctx = makeContexts(o);
    FILE *out = (o.outputfilename == "-") ? stdout : fopen(o.outputfilename.c_str(), "wb");
    if (!out) throw std::runtime_error("cannot open output file " + o.outputfilename);
    std::vector<char> obuf((size_t)8 << 20);
    setvbuf(out, obuf.data(), _IOFBF, obuf.size());
    uint64_t n_reads = 0, n_lines = 0;
    for (unsigned fi = 0; fi < files.size(); ++fi) {
        std::cerr << "Processing file " << files[fi] << std::endl;
        Resident R;
        double t0 = now_s();
        R.G.load(files[fi]);
        T.genome += now_s() - t0;
        t0 = now_s();
        setText(o, ctx, R, fi);
        T.index += now_s() - t0;
        const uint64_t n_list = blockEntries(o, ctx[0]->h, R.G.sym.size() ? R.G.sym.size() : 1);
        uint64_t first = 0;
        bool have_next = true;
        while (have_next) {
            t0 = now_s();
            const uint64_t n = nextBlock(o, ctx, R, first, n_list, have_next);
            T.index += now_s() - t0;
            if (!n) break;
            first += n;
            std::vector<std::vector<real_hip_hit>> hits(ctx.size(), std::vector<real_hip_hit>(1u << 20));
            std::vector<std::vector<uint64_t>> hoff(ctx.size());
            auto matchOne = [&](size_t g, real_hip_batch rb) {
                hoff[g].assign(rb.n_reads + 1, 0);
                if (!rb.n_reads) return;
                uint64_t nh = 0;
                int rc = real_hip_match_all(ctx[g]->h, &rb, hits[g].data(), hits[g].size(), &nh, hoff[g].data());
                if (rc == REAL_HIP_E_OVERFLOW) { // retry with the size the library reports
                    hits[g].resize(nh + 16);
                    rc = real_hip_match_all(ctx[g]->h, &rb, hits[g].data(), hits[g].size(), &nh, hoff[g].data());
                }
                check(ctx[g]->h, rc, "real_hip_match_all");
            };
            std::vector<uint32_t> id_start, id_len;
            std::vector<uint64_t> off;
            n_reads = streamReads(o, ctx, qoff, true, T,
                [&](const std::vector<uint64_t> &, const std::vector<Chunk> &ch, const std::vector<real_hip_parsed> &pr) {
                    const double tm = now_s();
                    onEach(ch.size(), [&](size_t g) { matchOne(g, makeBatch(pr[g])); });
                    T.match += now_s() - tm;
                    for (size_t g = 0; g < ch.size(); ++g) { // (in file order)
                        const uint64_t n = pr[g].n_reads;
                        if (!n) continue;
                        id_start.resize(n); id_len.resize(n); off.resize(n + 1);
                        const double td = now_s();
                        check(ctx[g]->h, real_hip_download(ctx[g]->h, pr[g].id_start, id_start.data(), n * 4), "real_hip_download");
                        check(ctx[g]->h, real_hip_download(ctx[g]->h, pr[g].id_len, id_len.data(), n * 4), "real_hip_download");
                        check(ctx[g]->h, real_hip_download(ctx[g]->h, pr[g].offsets, off.data(), (n + 1) * 8), "real_hip_download");
                        T.parse += now_s() - td;
                        const char *text = ch[g].text;
                        n_lines += hoff[g][n];
                        formatAndWrite(n, out, T, [&](uint64_t i, std::string &s) {
                            uint64_t il = id_len[i];
                            if (text[id_start[i] + il] == '\r') il++;
                            const uint64_t patl = off[i + 1] - off[i];
                            for (uint64_t k = hoff[g][i]; k < hoff[g][i + 1]; ++k) {
                                const real_hip_hit &M = hits[g][k];
                                appendLine(s, text + id_start[i], il, text + id_start[i] + il + 1, nullptr, patl, o.scores, M.score, M.inverted != 0,
                                           R.G.frag_names[M.frag], (uint64_t)M.pos - R.G.frag_start[M.frag] + 1, M.k);
                            }
                        });
                    }
                },
                [&](std::vector<ReadBlock> &blk, size_t used) {
                    const double tm = now_s();
                    onEach(used, [&](size_t g) { matchOne(g, makeBatch(blk[g])); });
                    T.match += now_s() - tm;
                    for (size_t g = 0; g < used; ++g) {
                        const ReadBlock &b = blk[g];
                        n_lines += hoff[g][b.size()];
                        formatAndWrite(b.size(), out, T, [&](uint64_t i, std::string &s) {
                            const uint64_t lo = b.offsets[i], patl = b.offsets[i + 1] - lo;
                            for (uint64_t k = hoff[g][i]; k < hoff[g][i + 1]; ++k) {
                                const real_hip_hit &M = hits[g][k];
                                appendLine(s, b.ids[i].data(), b.ids[i].size(), nullptr, &b.bases[lo], patl, o.scores, M.score, M.inverted != 0,
                                           R.G.frag_names[M.frag], (uint64_t)M.pos - R.G.frag_start[M.frag] + 1, M.k);
                            }
                        });
                    }
                });
        }
    }
    if (fflush(out) != 0) throw std::runtime_error("write to the output file failed");
    if (out != stdout) fclose(out);
    std::cerr << "All done." << std::endl;
    T.reads = n_reads; T.lines = n_lines; T.total = now_s() - t_begin;
    T.print();
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
