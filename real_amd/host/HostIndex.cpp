#include "HostIndex.hpp"

#include <omp.h>

#include <algorithm>
#include <cstring>

void enumerateWindows(const SymArray &sym, unsigned l, std::vector<uint32_t> &wpos)
{
    wpos.clear();
    uint64_t run = 0;
    for (uint64_t i = 0; i < sym.size(); ++i) {
        run = (sym[i] > 3) ? 0 : run + 1;
        if (run >= l) wpos.push_back((uint32_t)(i + 1 - l));
    }
}

// stable LSD radix sort of (key, val) by the low `bits` bits of key, 11-bit digits as the reference
template <typename K>
static void radix_sort_pairs(std::vector<K> &key, std::vector<uint32_t> &val, unsigned bits, int threads)
{
    const uint64_t n = key.size();
    if (!n) return;
    std::vector<K> key2(n);
    std::vector<uint32_t> val2(n);
    const unsigned DB = 11, NB = 1u << DB;
    if (threads < 1) threads = 1;
    std::vector<uint64_t> hist((size_t)threads * NB);
    K *ka = key.data(), *kb = key2.data();
    uint32_t *va = val.data(), *vb = val2.data();
    for (unsigned sh = 0; sh < bits; sh += DB) {
        std::fill(hist.begin(), hist.end(), 0);
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
            const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
            uint64_t *h = &hist[(size_t)t * NB];
            for (uint64_t i = lo; i < hi; ++i) h[(ka[i] >> sh) & (NB - 1)]++;
#pragma omp barrier
#pragma omp single
            {
                uint64_t sum = 0; // digit-major, thread-minor: keeps the input order inside a digit (stability)
                for (unsigned d = 0; d < NB; ++d)
                    for (int tt = 0; tt < T; ++tt) { uint64_t c = hist[(size_t)tt * NB + d]; hist[(size_t)tt * NB + d] = sum; sum += c; }
            }
            for (uint64_t i = lo; i < hi; ++i) {
                const uint64_t p = h[(ka[i] >> sh) & (NB - 1)]++;
                kb[p] = ka[i]; vb[p] = va[i];
            }
        }
        std::swap(ka, kb); std::swap(va, vb);
    }
    if (ka != key.data()) { memcpy(key.data(), ka, n * sizeof(K)); memcpy(val.data(), va, n * 4); }
}

void buildHostIndexBlock(const SymArray &sym, const std::vector<uint32_t> &wpos, unsigned l, uint64_t first,
                         uint64_t max_entries, int threads, HostIndexBlock &out)
{
    const uint64_t total = wpos.size();
    const uint64_t n = (first < total) ? std::min<uint64_t>(max_entries, total - first) : 0;
    out.n = n; out.have_next = first + n < total; out.sig_bytes = l <= 32 ? 4 : 8;
    const unsigned q = l / 4, bb = 2 * q;
    static const int SA[6] = {0, 0, 0, 1, 1, 2}, SC[6] = {1, 2, 3, 2, 3, 3}; // s0..s5, SignatureConstruction.hpp:62-67
    std::vector<uint64_t> m[4];
    for (int j = 0; j < 4; ++j) m[j].resize(n);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const uint8_t *s = sym.data() + wpos[first + i];
        for (int j = 0; j < 4; ++j) {
            uint64_t v = 0;
            for (unsigned t = 0; t < q; ++t) v = (v << 2) | (s[j * q + t] & 3);
            m[j][i] = v;
        }
    }
    for (int k = 0; k < 6; ++k) {
        out.pos[k].assign(wpos.begin() + first, wpos.begin() + first + n);
        if (out.sig_bytes == 4) {
            out.sign32[k].resize(n); out.sign64[k].clear();
            for (uint64_t i = 0; i < n; ++i) out.sign32[k][i] = (uint32_t)((m[SA[k]][i] << bb) | m[SC[k]][i]);
            radix_sort_pairs(out.sign32[k], out.pos[k], l, threads);
        } else {
            out.sign64[k].resize(n); out.sign32[k].clear();
            for (uint64_t i = 0; i < n; ++i) out.sign64[k][i] = (m[SA[k]][i] << bb) | m[SC[k]][i];
            radix_sort_pairs(out.sign64[k], out.pos[k], l, threads);
        }
    }
}
