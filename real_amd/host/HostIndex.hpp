// Host build of one genome index block, the form north_star keeps on the host: window
// enumeration (MapTextFile::readLists, MapTextFile.hpp:118-230), the six signatures per window
// (:211-216) and one stable LSD radix sort per list (ListSet::sort, ListSet.hpp:41-44;
// ParallelRadixSort.hpp -- stable, so equal signatures stay in ascending position).  The `ptr`
// cross links of Mask.hpp are not produced: the device re-reads the partner segments from the text.
#pragma once
#include <stdint.h>
#include <vector>

#include "GenomeText.hpp"

struct HostIndexBlock {
    unsigned sig_bytes = 4;              // 4 if seedl <= 32 else 8 (real.cpp:219-229)
    uint64_t n = 0;
    bool have_next = false;
    std::vector<uint32_t> sign32[6];
    std::vector<uint64_t> sign64[6];
    std::vector<uint32_t> pos[6];
    const void *sign_ptr(int k) const { return sig_bytes == 4 ? (const void *)sign32[k].data() : (const void *)sign64[k].data(); }
};

// N-free window starts in text order (all blocks); O(n)
void enumerateWindows(const SymArray &sym, unsigned seedl, std::vector<uint32_t> &wpos);
// block = windows [first, first+max_entries) of wpos
void buildHostIndexBlock(const SymArray &sym, const std::vector<uint32_t> &wpos, unsigned seedl,
                         uint64_t first, uint64_t max_entries, int threads, HostIndexBlock &out);
