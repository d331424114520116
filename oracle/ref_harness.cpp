/*
 * ref_harness.cpp -- white-box tap on the REAL reference, TEST INFRASTRUCTURE ONLY.
 *
 * This file is ours; it contains no reference code.  It is compiled by
 * oracle/Makefile against the reference's header-only hot path where it lies
 * (-I/root/reference/src: match.hpp, SignatureConstruction.hpp, MapTextFile.hpp,
 * u_sort.hpp/ParallelRadixSort.hpp, getLookupTable.hpp, AutoTextArray.hpp,
 * RangeVector.hpp, RestWordBuffer.hpp, ComputeScore.hpp, UniqueMatchInfo.hpp)
 * plus the three reference .cpp files that build without the autoconf-generated
 * real_config.hpp (Scoring.cpp, PopCountTable.cpp, StaticInitialization.cpp).
 * The output binary goes to oracle/_ref/ (git-ignored) and is used only to pin
 * oracle/real_oracle.c and to generate tests/golden/ fixtures.
 *
 * NOT compilable from the reference (they include the generated real_config.hpp
 * unconditionally): matchUniqueImplementation.cpp / matchAllImplementation.cpp,
 * i.e. UpdateUniqueInfo::update, UniqueMatcher::match, AllMatcher::match,
 * unifyMatches.  The harness therefore drives ::match itself in the call order
 * read from matchUniqueImplementation.cpp:407-497 and logs every
 * updater::update call; the fold is pinned only as a restatement.
 *
 * usage: ref_harness <dir> <seedl> <seedkmax> <totalkmax> <scores> <n_list>
 *   reads  <dir>/genome.u8 (symbols 0..4) <dir>/frag.u64 (n_frag+1 starts)
 *          <dir>/reads_off.u64 <dir>/reads_bases.u8 <dir>/reads_qual.u8
 *   writes <dir>/ref_*.bin / ref_events.txt
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>
#include <fstream>
#include <stdint.h>

#include "match.hpp"
#include "MapTextFile.hpp"
#include "u_sort.hpp"
#include "getLookupTable.hpp"
#include "getHistSize.hpp"
#include "getSampleBits.hpp"
#include "Pattern.hpp"
#include "Mask.hpp"

template<typename T>
static std::vector<T> slurp(std::string const & fn)
{
    std::ifstream in(fn.c_str(), std::ios::binary);
    if (!in) { std::cerr << "cannot open " << fn << std::endl; exit(2); }
    in.seekg(0, std::ios::end); size_t sz = in.tellg(); in.seekg(0);
    std::vector<T> v(sz / sizeof(T));
    if (sz) in.read(reinterpret_cast<char *>(&v[0]), sz);
    return v;
}
template<typename T>
static void dump(std::string const & fn, T const * p, size_t n)
{
    std::ofstream out(fn.c_str(), std::ios::binary);
    out.write(reinterpret_cast<char const *>(p), n * sizeof(T));
}

struct Event { unsigned read, call, inverted, pos, totalk, frag, fileid; float score; float eps; };

/* same shape as the reference's VectorUpdater: just logs update() calls */
struct LogUpdater
{
    typedef std::vector<Event> info_type;
    static unsigned cur_read, cur_call;
    static void update(bool const inverted, size_t const fileid, unsigned int const pos, unsigned int const totalk,
                       float const score, float const eps, unsigned int const fragid, info_type & V)
    {
        Event e; e.read = cur_read; e.call = cur_call; e.inverted = inverted; e.pos = pos; e.totalk = totalk;
        e.frag = fragid; e.fileid = fileid; e.score = score; e.eps = eps;
        V.push_back(e);
    }
};
unsigned LogUpdater::cur_read = 0;
unsigned LogUpdater::cur_call = 0;

template<typename signature_type>
static int run(std::string const & dir, unsigned seedl, unsigned seedkmax, unsigned totalkmax, bool scores_on, uint64_t n_list_req)
{
    static bool const sse4 = true;
    typedef unsigned int ptr_type;
    typedef Mask<signature_type, ptr_type> mask_type;
    typedef BaseMask<signature_type, ptr_type> base_mask_type;
    typedef PatternQualityBase pattern_type;

    std::vector<uint8_t> genome = slurp<uint8_t>(dir + "/genome.u8");
    std::vector<uint64_t> frag = slurp<uint64_t>(dir + "/frag.u64");
    std::vector<uint64_t> roff = slurp<uint64_t>(dir + "/reads_off.u64");
    std::vector<uint8_t> rbase = slurp<uint8_t>(dir + "/reads_bases.u8");
    std::vector<uint8_t> rqual = slurp<uint8_t>(dir + "/reads_qual.u8");
    size_t const n = genome.size();
    size_t const nreads = roff.size() ? roff.size() - 1 : 0;

    /* ---- scoring table ---- */
    Scoring scoring;
    {
        std::vector<double> LL(1024);
        for (unsigned c0 = 0; c0 < 4; ++c0) for (unsigned c1 = 0; c1 < 4; ++c1) for (unsigned q = 0; q < 64; ++q)
            LL[(c0 << 8) | (c1 << 6) | q] = scoring.getRawLogScoreTable(c0, c1, q);
        dump(dir + "/ref_LL.f64", &LL[0], LL.size());
    }

    /* ---- text ---- */
    AutoTextArray<sse4> ATA(genome.begin(), n);
    {
        std::vector<uint64_t> words((2 * n + 63) / 64);
        for (size_t i = 0; i < words.size(); ++i) words[i] = ATA.getTextWord(static_cast<unsigned int>(i));
        dump(dir + "/ref_text.u64", words.empty() ? 0 : &words[0], words.size());
        std::vector<uint8_t> sym(n);
        for (size_t i = 0; i < n; ++i) sym[i] = ATA[i];
        dump(dir + "/ref_sym.u8", n ? &sym[0] : 0, n);
    }
    std::vector< std::pair<std::string, u_int64_t> > ranges;
    for (size_t i = 0; i + 1 < frag.size(); ++i) ranges.push_back(std::make_pair(std::string("f"), (u_int64_t)frag[i]));
    ranges.push_back(std::make_pair(std::string("terminal"), (u_int64_t)frag.back()));
    RangeVector<sse4> RV(ranges);

    /* ---- record packing known answers (UniqueMatchInfo.hpp) ---- */
    {
        std::vector<uint64_t> recs;
        unsigned const states[] = { 0, 1, 2, 3, 4 };
        for (unsigned s = 0; s < 5; ++s) {
            UniqueMatchInfo<false> I;
            I.setState(static_cast<UniqueMatchInfoBase::MatchState>(states[s]));
            I.setPosition(207 + 1000 * s); I.setFileid(s * 7); I.setErrors(s * 3); I.setFragment(s * 1001);
            recs.push_back(I.data);
        }
        dump(dir + "/ref_records.u64", &recs[0], recs.size());
    }

    /* ---- lists, block by block (ListSetBlockReader.hpp:24-52, ListSet.hpp:41-63) ---- */
    SignatureConstruction<signature_type> SC(seedl, 4);
    MapTextFile<signature_type, sse4> MTF(ATA, seedl, 4);
    u_int64_t const n_list = n_list_req ? n_list_req : (n ? n : 1);

    AutoArray<mask_type> Afull[3];
    AutoArray<base_mask_type> Abase[3];
    for (unsigned i = 0; i < 3; ++i) { Afull[i] = AutoArray<mask_type>(n_list, false); Abase[i] = AutoArray<base_mask_type>(n_list, false); }

    std::vector<Event> events;
    std::ofstream sigout((dir + "/ref_sigs.txt").c_str());
    unsigned block = 0;
    bool have_next = false;
    while (true)
    {
        u_int64_t const masks = MTF.readLists(Afull[0].get(), Afull[1].get(), Afull[2].get(),
                                               Abase[0].get(), Abase[1].get(), Abase[2].get(), n_list, have_next);
        if (!masks) break;
        for (unsigned i = 0; i < 3; ++i)
            MaskSort<signature_type, unsigned int>::sort(Afull[i], Abase[2 - i], masks, 2);
        mask_type const * full[3]; base_mask_type const * base[3];
        for (unsigned i = 0; i < 3; ++i) { full[i] = Afull[i].get(); base[i] = Abase[i].get(); }
        unsigned int const histsize = getHistSize();
        AutoArray<size_t> Alookup[6];
        Alookup[0] = getLookupTable(full[0], masks, SC.s0shift(getSampleBits()), histsize);
        Alookup[1] = getLookupTable(full[1], masks, SC.s1shift(getSampleBits()), histsize);
        Alookup[2] = getLookupTable(full[2], masks, SC.s2shift(getSampleBits()), histsize);
        Alookup[3] = getLookupTable(base[0], masks, SC.s3shift(getSampleBits()), histsize);
        Alookup[4] = getLookupTable(base[1], masks, SC.s4shift(getSampleBits()), histsize);
        Alookup[5] = getLookupTable(base[2], masks, SC.s5shift(getSampleBits()), histsize);
        size_t const * lookup[6];
        for (unsigned i = 0; i < 6; ++i) lookup[i] = Alookup[i].get();

        /* dump the block's lists: sign (as u64), ptr, pos (through getPos) */
        {
            char tag[64];
            for (unsigned k = 0; k < 6; ++k) {
                std::vector<uint64_t> sg(masks); std::vector<uint32_t> pt(masks), ps(masks);
                for (u_int64_t j = 0; j < masks; ++j) {
                    if (k < 3) { sg[j] = full[k][j].sign; pt[j] = full[k][j].ptr; ps[j] = full[k][j].getPos(base[2 - k]); }
                    else       { sg[j] = base[k - 3][j].sign; pt[j] = base[k - 3][j].ptr; ps[j] = base[k - 3][j].getPos(full[5 - k]); }
                }
                snprintf(tag, sizeof tag, "/ref_b%u_l%u_", block, k);
                dump(dir + tag + "sign.u64", &sg[0], sg.size());
                dump(dir + tag + "ptr.u32", &pt[0], pt.size());
                dump(dir + tag + "pos.u32", &ps[0], ps.size());
                /* sparse lookup table: (prefix, low, high) for non-[0,0) entries */
                std::vector<uint64_t> lk;
                for (uint64_t p = 0; p < histsize; ++p)
                    if (lookup[k][2 * p] || lookup[k][2 * p + 1]) { lk.push_back(p); lk.push_back(lookup[k][2 * p]); lk.push_back(lookup[k][2 * p + 1]); }
                dump(dir + tag + "lookup.u64", lk.empty() ? 0 : &lk[0], lk.size());
            }
        }

        /* ---- per read: the 12 ::match calls of matchUniqueImplementation.cpp:407-497
               (= matchAllImplementation.cpp:308-349), all of them, no early-out ---- */
        RestWordBuffer<sse4> RWB(seedl);
        for (size_t z = 0; z < nreads; ++z)
        {
            unsigned const patl = roff[z + 1] - roff[z];
            std::string mapped(reinterpret_cast<char const *>(&rbase[roff[z]]), patl);
            std::string qual(reinterpret_cast<char const *>(&rqual[roff[z]]), patl);
            std::string transposed(patl, 0);
            for (unsigned i = 0; i < patl; ++i) transposed[i] = toollib::invertN(mapped[patl - 1 - i]);
            pattern_type pattern;
            pattern.patlen = patl; pattern.patid = z;
            pattern.mapped = mapped.c_str(); pattern.transposed = transposed.c_str(); pattern.quality = qual.c_str();
            if (patl < seedl || !pattern.isDontCareFree()) continue;
            RWB.setup(patl);
            u_int32_t m[4];
            if (!SC.signatureMapped(pattern.mapped, &m[0])) continue;
            RWB.setupStraight(pattern.mapped);
            signature_type const straight[] = { SC.s0(m[0], m[1]), SC.s1(m[0], m[2]), SC.s2(m[0], m[3]), SC.s3(m[1], m[2]), SC.s4(m[1], m[3]), SC.s5(m[2], m[3]) };
            u_int32_t im[4] = { 0, 0, 0, 0 };
            SC.reverseMappedSignature(pattern.mapped, &im[0]);
            signature_type const reverse[] = { SC.s0(im[0], im[1]), SC.s1(im[0], im[2]), SC.s2(im[0], im[3]), SC.s3(im[1], im[2]), SC.s4(im[1], im[3]), SC.s5(im[2], im[3]) };
            if (block == 0) {
                sigout << z;
                for (unsigned i = 0; i < 4; ++i) sigout << " " << m[i];
                for (unsigned i = 0; i < 4; ++i) sigout << " " << im[i];
                for (unsigned i = 0; i < 6; ++i) sigout << " " << (uint64_t)straight[i];
                for (unsigned i = 0; i < 6; ++i) sigout << " " << (uint64_t)reverse[i];
                sigout << " " << RWB.fullrestwords << " " << RWB.fracrestsyms;
                sigout << "\n";
            }
            float const epsilon = 0.0f; /* not used by ::match itself, only handed to update() */
            u_int64_t const fi = 0;
            LogUpdater::cur_read = z;
            for (unsigned inv = 0; inv < 2; ++inv)
            {
                signature_type const * s = inv ? reverse : straight;
                if (inv) RWB.setupReverse(pattern.mapped);
                bool const I = inv;
                #define CALLF(k,shiftf) LogUpdater::cur_call = inv*6+k; \
                    ::match<sse4, mask_type, base_mask_type, char const *, pattern_type, true, LogUpdater>(full[k], base[2-k], seedkmax, totalkmax, s[k], s[5-k], I, fi, SC.shiftf(getSampleBits()), lookup[k], RV, ATA, RWB, pattern, scoring, epsilon, events)
                #define CALLB(k,shiftf) LogUpdater::cur_call = inv*6+k; \
                    ::match<sse4, base_mask_type, mask_type, char const *, pattern_type, true, LogUpdater>(base[k-3], full[5-k], seedkmax, totalkmax, s[k], s[5-k], I, fi, SC.shiftf(getSampleBits()), lookup[k], RV, ATA, RWB, pattern, scoring, epsilon, events)
                #define CALLF0(k,shiftf) LogUpdater::cur_call = inv*6+k; \
                    ::match<sse4, mask_type, base_mask_type, char const *, pattern_type, false, LogUpdater>(full[k], base[2-k], seedkmax, totalkmax, s[k], s[5-k], I, fi, SC.shiftf(getSampleBits()), lookup[k], RV, ATA, RWB, pattern, scoring, epsilon, events)
                #define CALLB0(k,shiftf) LogUpdater::cur_call = inv*6+k; \
                    ::match<sse4, base_mask_type, mask_type, char const *, pattern_type, false, LogUpdater>(base[k-3], full[5-k], seedkmax, totalkmax, s[k], s[5-k], I, fi, SC.shiftf(getSampleBits()), lookup[k], RV, ATA, RWB, pattern, scoring, epsilon, events)
                if (scores_on) {
                    CALLF(0, s0shift); CALLF(1, s1shift); CALLF(2, s2shift);
                    CALLB(3, s3shift); CALLB(4, s4shift); CALLB(5, s5shift);
                } else {
                    CALLF0(0, s0shift); CALLF0(1, s1shift); CALLF0(2, s2shift);
                    CALLB0(3, s3shift); CALLB0(4, s4shift); CALLB0(5, s5shift);
                }
            }
        }
        /* tag this block's events */
        {
            char tag[64]; snprintf(tag, sizeof tag, "/ref_b%u_events.txt", block);
            FILE * f = fopen((dir + tag).c_str(), "w");
            for (size_t i = 0; i < events.size(); ++i) {
                Event const & e = events[i];
                uint32_t bits; memcpy(&bits, &e.score, 4);
                fprintf(f, "%u %u %u %u %u %u %u\n", e.read, e.call, e.inverted, e.pos, e.totalk, e.frag, bits);
            }
            fclose(f);
            events.clear();
        }
        block++;
        if (!have_next) break;
    }
    {
        FILE * f = fopen((dir + "/ref_meta.txt").c_str(), "w");
        fprintf(f, "blocks %u\n", block);
        fclose(f);
    }
    return 0;
}

int main(int argc, char * argv[])
{
    if (argc < 7) { std::cerr << "usage: ref_harness <dir> <seedl> <seedkmax> <totalkmax> <scores> <n_list>" << std::endl; return 1; }
    std::string dir = argv[1];
    unsigned seedl = atoi(argv[2]), seedkmax = atoi(argv[3]), totalkmax = atoi(argv[4]);
    bool scores = atoi(argv[5]);
    uint64_t n_list = strtoull(argv[6], 0, 10);
    /* real.cpp:219-229: seedl <= 32 -> u_int32_t signatures, else u_int64_t */
    if (seedl <= 32) return run<u_int32_t>(dir, seedl, seedkmax, totalkmax, scores, n_list);
    return run<u_int64_t>(dir, seedl, seedkmax, totalkmax, scores, n_list);
}
