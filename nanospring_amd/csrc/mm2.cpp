// ---------------------------------------------------------------------------------------------------------------------
// Third-party notice.  The functions in this file reproduce, decision for decision, the behaviour of minimap2 v2.17
// (https://github.com/lh3/minimap2; files cited per function) -- bit-exact tie orders and thresholds are part of the
// contract of NanoSpring's on-disk format, so the order of the decisions is minimap2's by necessity.  minimap2 is
//   Copyright (c) 2018- Dana-Farber Cancer Institute, 2017-2018 Broad Institute, Inc.
// and distributed under the MIT License; its full text is in THIRD_PARTY_NOTICES.md at the root of this repository.
// ---------------------------------------------------------------------------------------------------------------------
// mm2.cpp -- see mm2.hpp.  Host-side decision chain of the aligner; all banded DP is
// delegated to the HIP kernel through DpCache.
#include "mm2.hpp"
#include <atomic>
#include <chrono>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <climits>

namespace nsgpu {
namespace mm2 {

namespace {

constexpr uint64_t U64MAX = ~0ull;

// minimap2/sketch.c:9-26 -- A/a 0, C/c 1, G/g 2, T/t/U/u 3, everything else 4
struct Nt4 {
    uint8_t t[256];
    Nt4() {
        memset(t, 4, sizeof(t));
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['U'] = t['u'] = 3;
        t[0] = 0; t[1] = 1; t[2] = 2; t[3] = 3;   // the table's first row maps raw codes onto themselves
    }
};
const Nt4 kNt4;

// minimap2/sketch.c:28-38 -- Thomas Wang's invertible integer hash, masked to 2k bits
inline uint64_t hash64_masked(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// minimap2/hit.c:40-50 -- the unmasked variant used to salt chain scores
inline uint64_t hash64_full(uint64_t key)
{
    key = (~key + (key << 21));
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8));
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4));
    key = key ^ key >> 28;
    key = (key + (key << 31));
    return key;
}

// minimap2/khash.h:400-409
inline uint32_t wang_hash32(uint32_t key)
{
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}

#ifdef NSGPU_HOST_CHAIN
inline int ilog2_32(uint32_t v) { return 31 - __builtin_clz(v); }   // chain.c:8-20 (v > 0)
#endif

inline int span_of(const Anchor &p) { return (int)(p.y >> 32 & 0xff); }

// ---- the library's in-place MSD byte radix sort (ksort.h:98-151).  It is NOT stable,
// and the order it leaves equal keys in decides tie-breaks downstream, so the
// permutation cycle walk is reproduced step by step. ----
template <class T, class KeyFn>
void rs_insertion(T *beg, T *end, KeyFn key)
{
    for (T *i = beg + 1; i < end; ++i)
        if (key(*i) < key(*(i - 1))) {
            T tmp = *i, *j;
            for (j = i; j > beg && key(tmp) < key(*(j - 1)); --j) *j = *(j - 1);
            *j = tmp;
        }
}

template <class T, class KeyFn>
void rs_msd(T *beg, T *end, int n_bits, int s, KeyFn key)
{
    struct Bucket { T *b, *e; };
    const int size = 1 << n_bits, m = size - 1;
    Bucket b[256], *be = b + size, *k;
    for (k = b; k != be; ++k) k->b = k->e = beg;
    for (T *i = beg; i != end; ++i) ++b[key(*i) >> s & m].e;
    for (k = b + 1; k != be; ++k) k->e += (k - 1)->e - beg, k->b = (k - 1)->e;
    for (k = b; k != be;) {
        if (k->b != k->e) {
            Bucket *l;
            if ((l = b + (key(*k->b) >> s & m)) != k) {
                T tmp = *k->b, swap;
                do {
                    swap = tmp; tmp = *l->b; *l->b++ = swap;
                    l = b + (key(tmp) >> s & m);
                } while (l != k);
                *k->b++ = tmp;
            } else ++k->b;
        } else ++k;
    }
    for (b->b = beg, k = b + 1; k != be; ++k) k->b = (k - 1)->e;
    if (s) {
        s = s > n_bits ? s - n_bits : 0;
        for (k = b; k != be; ++k)
            if (k->e - k->b > 64) rs_msd(k->b, k->e, n_bits, s, key);
            else if (k->e - k->b > 1) rs_insertion(k->b, k->e, key);
    }
}

template <class T, class KeyFn>
void radix_sort_generic(T *beg, T *end, KeyFn key)
{
    if (end - beg <= 64) rs_insertion(beg, end, key);
    else rs_msd(beg, end, 8, 56, key);
}

}  // namespace

void radix_sort_128x(Anchor *beg, Anchor *end) { radix_sort_generic(beg, end, [](const Anchor &p) { return p.x; }); }
void radix_sort_64(uint64_t *beg, uint64_t *end) { radix_sort_generic(beg, end, [](const uint64_t &p) { return p; }); }

// ---------------------------------------------------------------------------
// a14c  (w,k)-minimizers, no homopolymer compression (NanoSpring passes is_hpc = false)
// ---------------------------------------------------------------------------
void mm_sketch(const char *str, int len, int w, int k, uint32_t rid, std::vector<Anchor> &out)
{
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ull << 2 * k) - 1;
    uint64_t fw = 0, rv = 0;
    int run = 0, slot = 0, min_slot = 0, span = 0;
    Anchor ring[256], best = {U64MAX, U64MAX};
    for (int j = 0; j < w; ++j) ring[j].x = ring[j].y = U64MAX;
    auto emit_ties = [&](int from, int to, const Anchor &m) {       // identical hashes at other positions
        for (int j = from; j < to; ++j)
            if (m.x == ring[j].x && ring[j].y != m.y) out.push_back(ring[j]);
    };
    for (int i = 0; i < len; ++i) {
        const int c = kNt4.t[(uint8_t)str[i]];
        Anchor cur = {U64MAX, U64MAX};
        if (c < 4) {
            span = run + 1 < k ? run + 1 : k;
            fw = (fw << 2 | (uint64_t)c) & mask;
            rv = (rv >> 2) | (3ull ^ (uint64_t)c) << shift1;
            if (fw == rv) continue;                       // palindromic k-mer: strand unknown; the window does NOT advance
            const int strand = fw < rv ? 0 : 1;
            ++run;
            if (run >= k && span < 256) {
                cur.x = hash64_masked(strand ? rv : fw, mask) << 8 | (uint64_t)span;
                cur.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)strand;
            }
        } else run = 0, span = 0;
        ring[slot] = cur;
        if (run == w + k - 1 && best.x != U64MAX) {       // first full window: ties were not stored yet
            emit_ties(slot + 1, w, best);
            emit_ties(0, slot, best);
        }
        if (cur.x <= best.x) {                            // new minimum (<=: the right-most wins)
            if (run >= w + k && best.x != U64MAX) out.push_back(best);
            best = cur, min_slot = slot;
        } else if (slot == min_slot) {                    // the minimum slid out of the window
            if (run >= w + k - 1 && best.x != U64MAX) out.push_back(best);
            best.x = U64MAX;                              // .y keeps its old value, as in the reference
            for (int j = slot + 1; j < w; ++j) if (best.x >= ring[j].x) best = ring[j], min_slot = j;
            for (int j = 0; j <= slot; ++j) if (best.x >= ring[j].x) best = ring[j], min_slot = j;
            if (run >= w + k - 1 && best.x != U64MAX) {
                emit_ties(slot + 1, w, best);
                emit_ties(0, slot + 1, best);
            }
        }
        if (++slot == w) slot = 0;
    }
    if (best.x != U64MAX) out.push_back(best);
}

// ---------------------------------------------------------------------------
// a14b + a14a  single-sequence index and mid_occ
// ---------------------------------------------------------------------------
// Ascending sort of distinct 64-bit keys whose top bits are well mixed (hashes): three 8-bit counting passes over the top
// 24 bits leave the array sorted up to the rare keys that agree there, which one insertion pass puts right.
static void sort_unique_keys(std::vector<uint64_t> &v, std::vector<uint64_t> &tmp)
{
    const size_t n = v.size();
    if (n < 512) { std::sort(v.begin(), v.end()); return; }
    tmp.resize(n);
    uint64_t *a = v.data(), *b = tmp.data();
    for (int shift = 40; shift <= 56; shift += 8) {
        uint32_t cnt[257] = {0};
        for (size_t i = 0; i < n; ++i) ++cnt[((a[i] >> shift) & 0xff) + 1];
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (size_t i = 0; i < n; ++i) b[cnt[(a[i] >> shift) & 0xff]++] = a[i];
        std::swap(a, b);
    }
    // three passes: the result is in tmp
    for (size_t i = 1; i < n; ++i) {
        const uint64_t x = a[i];
        size_t j = i;
        for (; j > 0 && a[j - 1] > x; --j) a[j] = a[j - 1];
        a[j] = x;
    }
    v.swap(tmp);
}

// ASCII -> nt4 codes (A0 C1 G2 T3, anything else 4; the reference's seq_nt4_table).  Eight upper-case ACGT at a time:
// bits 1-2 of the letter are A 00, C 01, T 10, G 11; the word is accepted only if rebuilding the letters from those bits
// gives the input back, everything else goes through the table.
void nt4_codes(const char *s, size_t n, uint8_t *out)
{
    const uint64_t ones = 0x0101010101010101ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, s + i, 8);
        const uint64_t b0 = (w >> 1) & ones, b1 = (w >> 2) & ones;
        const uint64_t rebuilt = 0x41 * ones + 2 * b0 + 0x13 * b1 - 0x0f * (b0 & b1);
        if (rebuilt != w) { for (size_t j = i; j < i + 8; ++j) out[j] = kNt4.t[(uint8_t)s[j]]; continue; }
        const uint64_t code = (b0 ^ b1) | b1 << 1;          // 00->0, 01->1, 10->3, 11->2
        memcpy(out + i, &code, 8);
    }
    for (; i < n; ++i) out[i] = kNt4.t[(uint8_t)s[i]];
}

void RefIndex::build(const char *s, uint32_t n, int w_, int k_, float mid_occ_frac)
{
    std::vector<Anchor> mz;
    if (n > 0) mm_sketch(s, (int)n, w_ < 1 ? 1 : w_, k_, 0, mz);
    build_from_sketch(s, n, w_, k_, mid_occ_frac, mz.data(), mz.size());
}

void RefIndex::set_sequence(const char *s, uint32_t n, int w_, int k_)
{
    k = k_, w = w_ < 1 ? 1 : w_, len = n;
    seq.resize(n);
    nt4_codes(s, n, seq.data());
    has_table = false;
}

void RefIndex::set_sequence_from(const char *s, uint32_t n, int w_, int k_, size_t from)
{
    if (from > seq.size() || from > n) from = 0;
    k = k_, w = w_ < 1 ? 1 : w_, len = n;
    seq.resize(n);
    nt4_codes(s + from, n - from, seq.data() + from);
    has_table = false;
}

void RefIndex::set_sequence_spliced(const char *s, uint32_t n, int w_, int k_, size_t P, size_t S)
{
    const size_t old_n = seq.size();
    if (P + S > old_n || P + S > n) { set_sequence_from(s, n, w_, k_, 0); return; }
    k = k_, w = w_ < 1 ? 1 : w_, len = n;
    if (n > old_n) seq.resize(n);
    if (S && n != old_n) memmove(seq.data() + (n - S), seq.data() + (old_n - S), S);
    if (n < old_n) seq.resize(n);
    nt4_codes(s + P, n - S - P, seq.data() + P);
    has_table = false;
}

void RefIndex::build_from_sketch(const char *s, uint32_t n, int w_, int k_, float mid_occ_frac, const Anchor *mz_p, size_t mz_n)
{
    set_sequence(s, n, w_, k_);
    has_table = true;
    struct Span { const Anchor *p; size_t n; size_t size() const { return n; } const Anchor &operator[](size_t i) const { return p[i]; } } mz{mz_p, mz_n};
    // the bucketed hash tables of the reference (index.c:191-248) only define "hash -> positions
    // ascending"; a (hash, position) sort gives the same mapping
    keys.clear(); start.clear(); pos.resize(mz.size());
    uint32_t idx_bits = 1;
    while (((size_t)1 << idx_bits) < mz.size()) ++idx_bits;
    bool ascending = true;                 // mm_sketch emits a sequence's minimizers by position
    for (size_t i = 1; i < mz.size() && ascending; ++i) ascending = mz[i - 1].y < mz[i].y;
    if (ascending && 2 * (uint32_t)k + idx_bits <= 64) {
        // one 64-bit key per minimizer, the 2k-bit hash in the top bits (so that the radix sort's first byte already
        // separates the keys) above its rank in the sketch: the sorted order is the (hash, position) order
        const uint32_t hs = 64 - 2 * (uint32_t)k;
        sort_keys_.resize(mz.size());
        for (size_t i = 0; i < mz.size(); ++i) sort_keys_[i] = (mz[i].x >> 8) << hs | i;
        sort_unique_keys(sort_keys_, sort_tmp_);
        const uint64_t im = ((uint64_t)1 << idx_bits) - 1;
        uint64_t prev = 0;
        for (size_t i = 0; i < sort_keys_.size(); ++i) {
            const uint64_t h = sort_keys_[i] >> hs;
            if (i == 0 || h != prev) keys.push_back(h), start.push_back((uint32_t)i);
            prev = h;
            pos[i] = mz[sort_keys_[i] & im].y;
        }
    } else {
        std::vector<std::pair<uint64_t, uint64_t>> kv(mz.size());
        for (size_t i = 0; i < mz.size(); ++i) kv[i] = {mz[i].x >> 8, mz[i].y};
        std::sort(kv.begin(), kv.end());
        for (size_t i = 0; i < kv.size(); ++i) {
            if (i == 0 || kv[i].first != kv[i - 1].first) keys.push_back(kv[i].first), start.push_back((uint32_t)i);
            pos[i] = kv[i].second;
        }
    }
    start.push_back((uint32_t)mz.size());
    {
        uint32_t bits = 4;
        while (((size_t)1 << bits) < 2 * keys.size() + 2) ++bits;
        slot.assign((size_t)1 << bits, 0u);
        slot_shift = 64 - bits;
        const size_t m = slot.size() - 1;
        for (size_t i = 0; i < keys.size(); ++i) {
            size_t h = (size_t)((keys[i] * 0x9e3779b97f4a7c15ull) >> slot_shift);
            while (slot[h]) h = (h + 1) & m;
            slot[h] = (uint32_t)i + 1;
        }
    }
    // mm_idx_cal_max_occ(mi, 2e-4f) (index.c:164-185): (k-th smallest occurrence count) + 1,
    // k = (uint32_t)((1. - f) * n_distinct) with f a float promoted to double
    const size_t nd = keys.size();
    if (mid_occ_frac <= 0.f) mid_occ = INT32_MAX;
    else if (nd == 0) mid_occ = 1;   // the reference reads an empty array here; with no minimizers nothing can seed anyway
    else {
        std::vector<uint32_t> occ(nd);
        for (size_t i = 0; i < nd; ++i) occ[i] = start[i + 1] - start[i];
        const size_t kk = (uint32_t)((1. - mid_occ_frac) * nd);
        std::nth_element(occ.begin(), occ.begin() + kk, occ.end());
        mid_occ = (int32_t)(occ[kk] + 1);
    }
}

const uint64_t *RefIndex::get(uint64_t minier, int *n) const
{
    if (slot.empty()) { *n = 0; return nullptr; }
    const size_t m = slot.size() - 1;
    size_t h = (size_t)((minier * 0x9e3779b97f4a7c15ull) >> slot_shift);
    while (slot[h] && keys[slot[h] - 1] != minier) h = (h + 1) & m;
    if (!slot[h]) { *n = 0; return nullptr; }
    const size_t i = slot[h] - 1;
    *n = (int)(start[i + 1] - start[i]);
    return pos.data() + start[i];
}

// ---------------------------------------------------------------------------
// a14d  seeds (collect_matches + collect_seed_hits, map.c:90-123, 215-247) with
// MM_F_FOR_ONLY (skip_seed, map.c:139-145): only same-strand hits survive.
// ---------------------------------------------------------------------------
static void collect_seeds(const RefIndex &ri, const Anchor *mv, size_t n_mv, std::vector<Anchor> &a)
{
    a.clear();
    for (size_t i = 0; i < n_mv; ++i) {
        const Anchor &p = mv[i];
        const uint32_t q_pos = (uint32_t)p.y, q_span = (uint32_t)(p.x & 0xff);
        int t;
        const uint64_t *cr = ri.get(p.x >> 8, &t);
        if (t >= ri.mid_occ) continue;                  // too frequent on the reference
        bool tandem = false;
        if (i > 0 && p.x >> 8 == mv[i - 1].x >> 8) tandem = true;
        if (i + 1 < n_mv && p.x >> 8 == mv[i + 1].x >> 8) tandem = true;
        for (int k = 0; k < t; ++k) {
            const uint64_t r = cr[k];
            if ((r & 1) != (q_pos & 1)) continue;       // reverse-strand seed dropped
            Anchor s;
            s.x = (r & 0xffffffff00000000ull) | (uint32_t)((uint32_t)r >> 1);
            s.y = (uint64_t)q_span << 32 | (q_pos >> 1);
            if (tandem) s.y |= SEED_TANDEM;
            a.push_back(s);
        }
    }
    // minimap2 sorts the anchors with its (unstable) radix sort; where all keys differ every correct sort gives the same
    // array, and a comparison sort of a few hundred nearly sorted anchors is several times faster than the radix passes over
    // the empty high bytes.  Equal keys (one reference position hit by two query minimizers) fall back to the reference's
    // algorithm on the original order, whose tie order is what the chaining then sees.
    static thread_local std::vector<Anchor> orig;
    orig.assign(a.begin(), a.end());
    std::sort(a.begin(), a.end(), [](const Anchor &p, const Anchor &q) { return p.x < q.x; });
    bool ties = false;
    for (size_t i = 1; i < a.size(); ++i) if (a[i].x == a[i - 1].x) { ties = true; break; }
    if (ties) { a.assign(orig.begin(), orig.end()); radix_sort_128x(a.data(), a.data() + a.size()); }
}

// ---------------------------------------------------------------------------
// a14e  chaining (chain.c:22-164) for one segment, genomic mode.  The forward pass -- score f[i] and predecessor p[i] of every
// anchor, the O(n * window) part -- runs on the GPU (chain.hip, one wave per query); chain_forward_host below is the same
// recurrence as a plain loop and exists only in builds with NSGPU_HOST_CHAIN (the CPU test harness uses it to check the kernel's
// inputs and outputs against the live reference without a GPU).  chain_finish is the sequential remainder: peak scores, chain
// ends, backtracking, the order of the chains.
// ---------------------------------------------------------------------------
float chain_avg_qspan(const std::vector<Anchor> &a)
{
    uint64_t sum_qspan = 0;
    for (const Anchor &x : a) sum_qspan += x.y >> 32 & 0xff;
    return a.empty() ? 0.f : (float)sum_qspan / (int64_t)a.size();
}

#ifdef NSGPU_HOST_CHAIN
void chain_forward_host(const Opt &o, const std::vector<Anchor> &a, float avg_qspan, int32_t *f, int32_t *p)
{
    const int64_t n = (int64_t)a.size();
    const int max_dist_x = o.max_gap, max_dist_y = o.max_gap, bw = o.bw, max_skip = o.max_chain_skip, max_iter = o.max_chain_iter;
    std::vector<int32_t> t(n, 0);
    int64_t st = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t ri = a[i].x;
        int64_t max_j = -1;
        const int32_t qi = (int32_t)a[i].y, q_span = (int32_t)(a[i].y >> 32 & 0xff);
        int32_t max_f = q_span, n_skip = 0;
        while (st < i && ri > a[st].x + (uint64_t)max_dist_x) ++st;
        if (i - st > max_iter) st = i - max_iter;
        for (int64_t j = i - 1; j >= st; --j) {
            const int64_t dr = (int64_t)(ri - a[j].x);
            const int32_t dq = qi - (int32_t)a[j].y;
            if (dr == 0 || dq <= 0) continue;
            if (dq > max_dist_y || dq > max_dist_x) continue;
            const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
            if (dd > bw) continue;
            const int32_t min_d = dq < dr ? dq : (int32_t)dr;
            int32_t sc = min_d > q_span ? q_span : (dq < dr ? dq : (int32_t)dr);
            const int32_t log_dd = dd ? ilog2_32((uint32_t)dd) : 0;
            const int32_t gap_cost = (int)(dd * .01 * avg_qspan) + (log_dd >> 1);
            sc -= (int)((double)gap_cost * o.chain_gap_scale + .499);
            sc += f[j];
            if (sc > max_f) {
                max_f = sc, max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == (int32_t)i) {
                if (++n_skip > max_skip) break;
            }
            if (p[j] >= 0) t[p[j]] = (int32_t)i;
        }
        f[i] = max_f, p[i] = (int32_t)max_j;
    }
}
#endif

static void chain_finish(const Opt &o, std::vector<Anchor> &a, const int32_t *f, const int32_t *p, std::vector<uint64_t> &u)
{
    const int64_t n = (int64_t)a.size();
    const int min_cnt = o.min_cnt, min_sc = o.min_chain_score;
    u.clear();
    if (n == 0) return;
    std::vector<int32_t> t(n, 0), v(n);
    for (int64_t i = 0; i < n; ++i) v[i] = p[i] >= 0 && v[p[i]] > f[i] ? v[p[i]] : f[i];
    // chain ends
    for (int64_t i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
    std::vector<uint64_t> ends;
    for (int64_t i = 0; i < n; ++i)
        if (t[i] == 0 && v[i] >= min_sc) {
            int64_t j = i;
            while (j >= 0 && f[j] < v[j]) j = p[j];
            if (j < 0) j = i;
            ends.push_back((uint64_t)f[j] << 32 | (uint64_t)j);
        }
    if (ends.empty()) { a.clear(); return; }
    radix_sort_64(ends.data(), ends.data() + ends.size());
    std::reverse(ends.begin(), ends.end());
    // backtrack, best chain first
    std::fill(t.begin(), t.end(), 0);
    int32_t n_v = 0, k = 0;
    const int32_t n_u0 = (int32_t)ends.size();
    for (int32_t i = 0; i < n_u0; ++i) {
        const int32_t n_v0 = n_v, k0 = k;
        int64_t j = (int32_t)ends[i];
        do {
            v[n_v++] = (int32_t)j;
            t[j] = 1;
            j = p[j];
        } while (j >= 0 && t[j] == 0);
        if (j < 0) {
            if (n_v - n_v0 >= min_cnt) ends[k++] = ends[i] >> 32 << 32 | (uint64_t)(n_v - n_v0);
        } else if ((int32_t)(ends[i] >> 32) - f[j] >= min_sc) {
            if (n_v - n_v0 >= min_cnt) ends[k++] = (uint64_t)((ends[i] >> 32) - f[j]) << 32 | (uint64_t)(n_v - n_v0);
        }
        if (k0 == k) n_v = n_v0;
    }
    const int32_t n_u = k;
    std::vector<Anchor> b(n_v);
    for (int32_t i = 0, kk = 0; i < n_u; ++i) {
        const int32_t k0 = kk, ni = (int32_t)ends[i];
        for (int32_t j = 0; j < ni; ++j) b[kk] = a[v[k0 + (ni - j - 1)]], ++kk;
    }
    // order chains by the reference position of their first anchor (needed by the long join)
    std::vector<Anchor> wv(n_u);
    for (int32_t i = 0, kk = 0; i < n_u; ++i) {
        wv[i].x = b[kk].x, wv[i].y = (uint64_t)kk << 32 | (uint64_t)i;
        kk += (int32_t)ends[i];
    }
    radix_sort_128x(wv.data(), wv.data() + n_u);
    u.resize(n_u);
    std::vector<Anchor> out(n_v);
    for (int32_t i = 0, kk = 0; i < n_u; ++i) {
        const int32_t j = (int32_t)wv[i].y, nn = (int32_t)ends[j];
        u[i] = ends[j];
        memcpy(out.data() + kk, b.data() + (wv[i].y >> 32), (size_t)nn * sizeof(Anchor));
        kk += nn;
    }
    a.swap(out);
}

void chain_finish_scores(const Opt &o, std::vector<Anchor> &a, const int32_t *f, const int32_t *p, std::vector<uint64_t> &u) { chain_finish(o, a, f, p, u); }

// ---------------------------------------------------------------------------
// a14f  regions (hit.c)
// ---------------------------------------------------------------------------
static void reg_set_coor(Reg &r, int32_t qlen, const Anchor *a)        // hit.c:8-38 (forward strand only)
{
    const int32_t k = r.as, q_span = span_of(a[k]);
    (void)qlen;
    r.rev = (uint8_t)(a[k].x >> 63);
    r.rid = (int32_t)(a[k].x << 1 >> 33);
    r.rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r.re = (int32_t)a[k + r.cnt - 1].x + 1;
    r.qs = (int32_t)a[k].y + 1 - q_span;
    r.qe = (int32_t)a[k + r.cnt - 1].y + 1;
    r.mlen = r.blen = 0;
    if (r.cnt <= 0) return;
    r.mlen = r.blen = span_of(a[r.as]);
    for (int i = r.as + 1; i < r.as + r.cnt; ++i) {
        const int span = span_of(a[i]);
        const int tl = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        const int ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        r.blen += tl > ql ? tl : ql;
        r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
}

static void gen_regs(uint32_t hash, int qlen, const std::vector<uint64_t> &u, const std::vector<Anchor> &a, std::vector<Reg> &regs)   // hit.c:52-88
{
    const int n_u = (int)u.size();
    regs.clear();
    if (n_u == 0) return;
    std::vector<Anchor> z(n_u);
    for (int i = 0, k = 0; i < n_u; ++i) {
        const uint32_t h = (uint32_t)hash64_full((hash64_full(a[k].x) + hash64_full(a[k].y)) ^ hash);
        z[i].x = u[i] ^ h;
        z[i].y = (uint64_t)k << 32 | (uint64_t)(int64_t)(int32_t)u[i];
        k += (int32_t)u[i];
    }
    radix_sort_128x(z.data(), z.data() + n_u);
    std::reverse(z.begin(), z.end());
    regs.resize(n_u);
    for (int i = 0; i < n_u; ++i) {
        Reg &r = regs[i];
        r = Reg();
        r.id = i;
        r.parent = PARENT_UNSET;
        r.score = r.score0 = (int32_t)(z[i].x >> 32);
        r.hash = (uint32_t)z[i].x;
        r.cnt = (int32_t)z[i].y;
        r.as = (int32_t)(z[i].y >> 32);
        reg_set_coor(r, qlen, a.data());
    }
}

static void set_parent(const Opt &o, std::vector<Reg> &r)      // hit.c:125-186, no ALT contigs, soft mask level
{
    const int n = (int)r.size();
    if (n <= 0) return;
    const int sub_diff = o.a * 2 + o.b;
    for (int i = 0; i < n; ++i) r[i].id = i;
    std::vector<uint64_t> cov(n);
    std::vector<int> w(n);
    w[0] = 0, r[0].parent = 0;
    int k = 1;
    for (int i = 1; i < n; ++i) {
        Reg &ri = r[i];
        const int si = ri.qs, ei = ri.qe;
        int n_cov = 0, uncov_len = 0, j;
        for (j = 0; j < k; ++j) {
            const Reg &rp = r[w[j]];
            int sj = rp.qs, ej = rp.qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)sj << 32 | (uint64_t)ej;
        }
        bool is_primary = false;
        if (n_cov == 0) is_primary = true;
        else {
            int x = si;
            radix_sort_64(cov.data(), cov.data() + n_cov);
            for (int c = 0; c < n_cov; ++c) {
                if ((int)(cov[c] >> 32) > x) uncov_len += (int)(cov[c] >> 32) - x;
                x = (int32_t)cov[c] > x ? (int32_t)cov[c] : x;
            }
            if (ei > x) uncov_len += ei - x;
        }
        if (!is_primary) {
            for (j = 0; j < k; ++j) {
                Reg &rp = r[w[j]];
                const int sj = rp.qs, ej = rp.qe;
                if (ej <= si || sj >= ei) continue;
                const int mn = ej - sj < ei - si ? ej - sj : ei - si;
                const int mx = ej - sj > ei - si ? ej - sj : ei - si;
                const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / mn - (float)uncov_len / mx > o.mask_level && uncov_len <= o.mask_len) {
                    int cnt_sub = 0, sci = ri.score;
                    ri.parent = rp.parent;
                    rp.subsc = rp.subsc > sci ? rp.subsc : sci;
                    if (ri.cnt >= rp.cnt) cnt_sub = 1;
                    if (rp.has_p && ri.has_p && (rp.rid != ri.rid || rp.rs != ri.rs || rp.re != ri.re || ol != mn)) {
                        sci = ri.p.dp_max;
                        rp.p.dp_max2 = rp.p.dp_max2 > sci ? rp.p.dp_max2 : sci;
                        if (rp.p.dp_max - ri.p.dp_max <= sub_diff) cnt_sub = 1;
                    }
                    if (cnt_sub) ++rp.n_sub;
                    break;
                }
            }
            if (j == k) is_primary = true;
        }
        if (is_primary) w[k++] = i, ri.parent = i, ri.n_sub = 0;
    }
}

static void set_sam_pri(std::vector<Reg> &r)                    // hit.c:220-229
{
    int n_pri = 0;
    for (auto &x : r)
        if (x.id == x.parent) { ++n_pri; x.sam_pri = (n_pri == 1); }
        else x.sam_pri = 0;
}

static void sync_regs(std::vector<Reg> &regs)                   // hit.c:231-253
{
    const int n = (int)regs.size();
    if (n <= 0) return;
    int max_id = -1;
    for (auto &r : regs) max_id = max_id > r.id ? max_id : r.id;
    std::vector<int> tmp(max_id + 1 > 0 ? max_id + 1 : 0, -1);
    for (int i = 0; i < n; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
    for (int i = 0; i < n; ++i) {
        Reg &r = regs[i];
        r.id = i;
        if (r.parent == PARENT_TMP_PRI) r.parent = i;
        else if (r.parent >= 0 && tmp[r.parent] >= 0) r.parent = tmp[r.parent];
        else r.parent = PARENT_UNSET;
    }
    set_sam_pri(regs);
}

static void select_sub(const Opt &o, int min_diff, std::vector<Reg> &r)      // hit.c:255-272
{
    if (!(o.pri_ratio > 0.0f) || r.empty()) return;
    const int n = (int)r.size();
    int k = 0, n_2nd = 0;
    for (int i = 0; i < n; ++i) {
        const int p = r[i].parent;
        if (p == i || r[i].inv) {
            if (k != i) r[k] = r[i];
            ++k;
        } else if ((r[i].score >= r[p].score * o.pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < o.best_n) {
            if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) {
                if (k != i) r[k] = r[i];
                ++k, ++n_2nd;
            }
        }
    }
    r.resize(k);
    if (k != n) sync_regs(r);
}

static void filter_regs(const Opt &o, int qlen, std::vector<Reg> &regs)      // hit.c:274-293
{
    size_t k = 0;
    for (size_t i = 0; i < regs.size(); ++i) {
        Reg &r = regs[i];
        bool flt = false;
        if (!r.inv && r.cnt < o.min_cnt) flt = true;
        if (r.has_p) {
            if (r.mlen < o.min_chain_score) flt = true;
            else if (r.p.dp_max < o.min_dp_max) flt = true;
            else if (r.qs > qlen * o.max_clip_ratio && qlen - r.qe > qlen * o.max_clip_ratio) flt = true;
        }
        if (!flt) { if (k < i) regs[k] = regs[i]; ++k; }
    }
    regs.resize(k);
}

static int squeeze_a(std::vector<Reg> &regs, std::vector<Anchor> &a)         // hit.c:295-313
{
    const int n = (int)regs.size();
    std::vector<uint64_t> aux(n);
    for (int i = 0; i < n; ++i) aux[i] = (uint64_t)regs[i].as << 32 | (uint64_t)i;
    radix_sort_64(aux.data(), aux.data() + n);
    int as = 0;
    for (int i = 0; i < n; ++i) {
        Reg &r = regs[(int32_t)aux[i]];
        if (r.as != as) {
            memmove(&a[as], &a[r.as], (size_t)r.cnt * sizeof(Anchor));
            r.as = as;
        }
        as += r.cnt;
    }
    return as;
}

static void join_long(const Opt &o, int qlen, std::vector<Reg> &regs, std::vector<Anchor> &a)    // hit.c:315-371
{
    const int n_regs = (int)regs.size();
    if (n_regs < 2) return;
    squeeze_a(regs, a);
    std::vector<uint64_t> aux;
    for (int i = 0; i < n_regs; ++i)
        if (regs[i].parent == i || regs[i].parent < 0) aux.push_back((uint64_t)regs[i].as << 32 | (uint64_t)i);
    radix_sort_64(aux.data(), aux.data() + aux.size());
    int n_drop = 0;
    for (int i = (int)aux.size() - 1; i >= 1; --i) {
        Reg &r0 = regs[(int32_t)aux[i - 1]], &r1 = regs[(int32_t)aux[i]];
        if (r0.as + r0.cnt != r1.as) continue;
        if (r0.rid != r1.rid || r0.rev != r1.rev) continue;
        const Anchor &a0e = a[r0.as + r0.cnt - 1], &a1s = a[r1.as];
        if (a1s.x <= a0e.x || (int32_t)a1s.y <= (int32_t)a0e.y) continue;
        int max_gap, min_gap;
        max_gap = min_gap = (int32_t)a1s.y - (int32_t)a0e.y;
        max_gap = a0e.x + (uint64_t)(int64_t)max_gap > a1s.x ? max_gap : (int)(a1s.x - a0e.x);
        min_gap = a0e.x + (uint64_t)(int64_t)min_gap < a1s.x ? min_gap : (int)(a1s.x - a0e.x);
        if (max_gap > o.max_join_long || min_gap > o.max_join_short) continue;
        const int sc_thres = (int)((float)o.min_join_flank_sc / o.max_join_long * max_gap + .499);
        if (r0.score < sc_thres || r1.score < sc_thres) continue;
        const int min_flank_len = (int)(max_gap * o.min_join_flank_ratio);
        if (r0.re - r0.rs < min_flank_len || r0.qe - r0.qs < min_flank_len) continue;
        if (r1.re - r1.rs < min_flank_len || r1.qe - r1.qs < min_flank_len) continue;
        a[r1.as].y |= SEED_LONG_JOIN;
        r0.cnt += r1.cnt, r0.score += r1.score;
        reg_set_coor(r0, qlen, a.data());
        r1.cnt = 0;
        r1.parent = r0.id;
        ++n_drop;
    }
    if (n_drop > 0) {
        for (int i = 0; i < n_regs; ++i) {
            Reg &r = regs[i];
            if (r.parent >= 0 && r.id != r.parent)
                if (regs[r.parent].parent >= 0 && regs[r.parent].parent != r.parent) r.parent = regs[r.parent].parent;
        }
        filter_regs(o, qlen, regs);
        sync_regs(regs);
    }
}

static void hit_sort(std::vector<Reg> &r)                       // hit.c:188-218
{
    const int n = (int)r.size();
    if (n <= 1) return;
    std::vector<Anchor> aux;
    for (int i = 0; i < n; ++i)
        if (r[i].inv || r[i].cnt > 0) {
            const int score = r[i].has_p ? r[i].p.dp_max : r[i].score;
            Anchor x;
            x.x = (uint64_t)(int64_t)score << 32 | r[i].hash;
            x.y = (uint64_t)i;
            aux.push_back(x);
        }
    radix_sort_128x(aux.data(), aux.data() + aux.size());
    std::vector<Reg> t(aux.size());
    for (int i = (int)aux.size() - 1; i >= 0; --i) t[aux.size() - 1 - i] = r[aux[i].y];
    r.swap(t);
}

static void split_reg(Reg &r, Reg &r2, int n, int qlen, const Anchor *a)     // hit.c:106-123
{
    if (n <= 0 || n >= r.cnt) return;
    r2 = r;
    r2.id = -1;
    r2.sam_pri = 0;
    r2.has_p = false; r2.p = Extra();
    r2.split_inv = 0;
    r2.cnt = r.cnt - n;
    r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
    r2.as = r.as + n;
    if (r.parent == r.id) r2.parent = PARENT_TMP_PRI;
    reg_set_coor(r2, qlen, a);
    r.cnt -= r2.cnt;
    r.score -= r2.score;
    reg_set_coor(r, qlen, a);
    r.split |= 1, r2.split |= 2;
}

// ---------------------------------------------------------------------------
// a14g  alignment skeleton (align.c)
// ---------------------------------------------------------------------------
bool DpKey::operator<(const DpKey &o) const
{
    const int32_t l[8] = {qs, qe, rs, re, w, zdrop, end_bonus, flag}, r[8] = {o.qs, o.qe, o.rs, o.re, o.w, o.zdrop, o.end_bonus, o.flag};
    for (int i = 0; i < 8; ++i) if (l[i] != r[i]) return l[i] < r[i];
    return false;
}

static inline bool same_key(const DpKey &a, const DpKey &b) { return memcmp(&a, &b, sizeof(DpKey)) == 0; }

const DpResult *DpCache::find(const DpKey &k)
{
    const size_t n = keys.size();
    for (size_t t = 0; t < n; ++t) {
        size_t i = cursor + t;
        if (i >= n) i -= n;
        if (same_key(keys[i], k)) { cursor = i + 1 < n ? i + 1 : 0; return &vals[i]; }
    }
    return nullptr;
}

const DpResult *DpCache::get(const DpKey &k)
{
    if (const DpResult *r = find(k)) return r;
    if (std::find_if(missing.begin(), missing.end(), [&](const DpKey &m) { return same_key(m, k); }) == missing.end())
        missing.push_back(k);
    return nullptr;
}

const DpResult &DpCache::at(const DpKey &k)
{
    const DpResult *r = find(k);
    if (!r) { fprintf(stderr, "nsgpu: DP result missing from the job's cache\n"); abort(); }
    return *r;
}

void DpCache::put(const DpKey &k, const DpResult &scalars, const uint32_t *cigar, uint32_t n_cigar)
{
    DpResult r = scalars;
    r.cig_off = (uint32_t)pool.size(), r.n_cigar = n_cigar;
    pool.insert(pool.end(), cigar, cigar + n_cigar);
    keys.push_back(k);
    vals.push_back(r);
}

namespace {

constexpr int EZ_RIGHT = 0x02, EZ_APPROX_MAX = 0x08, EZ_EXTZ_ONLY = 0x40, EZ_REV_CIGAR = 0x80;

struct Mat5 { int8_t m[25]; };
Mat5 simple_mat(int a, int b, int sc_ambi)                     // align.c:9-22
{
    Mat5 o;
    a = a < 0 ? -a : a; b = b > 0 ? -b : b; sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) o.m[i * 5 + j] = (int8_t)(i == j ? a : b); o.m[i * 5 + 4] = (int8_t)sc_ambi; }
    for (int j = 0; j < 5; ++j) o.m[20 + j] = (int8_t)sc_ambi;
    return o;
}

// align.c:32-89 with MM_F_FOR_ONLY: the inversion branch is unreachable, returns 0 or 1
int test_zdrop(const Opt &o, const uint8_t *qseq, const uint8_t *tseq, CigSpan cigar, const int8_t *mat)
{
    int32_t score = 0, mx = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
    auto upd = [&](int32_t sc, int ii, int jj) {
        if (sc < mx) {
            const int li = ii - max_i, lj = jj - max_j, diff = li > lj ? li - lj : lj - li;
            const int z = mx - sc - diff * o.e;
            if (z > max_zdrop) max_zdrop = z;
        } else mx = sc, max_i = ii, max_j = jj;
    };
    const int a_match = mat[0];
    const bool fast = a_match > 0 && mat[6] == a_match && mat[12] == a_match && mat[18] == a_match;
    for (uint32_t c : cigar) {
        const uint32_t op = c & 0xf, len = c >> 4;
        if (op == 0) {
            uint32_t l = 0;
            while (l < len) {
                // Eight equal unambiguous bases at a time.  Over a run of matches the score climbs by a > 0 per base and
                // both coordinates advance together, so (i) no position of the run can raise max_zdrop above what the
                // update just before the run saw (same diagonal offset, higher score) and (ii) the maximum, if the run
                // reaches it, ends up at the run's last base: one update for that base leaves the same state.
                if (fast && l + 8 <= len) {
                    uint64_t wq, wt;
                    memcpy(&wq, qseq + j + l, 8), memcpy(&wt, tseq + i + l, 8);
                    if (wq == wt && !(wq & 0xfcfcfcfcfcfcfcfcull)) {
                        score += 8 * a_match;
                        l += 8;
                        if (score >= mx) mx = score, max_i = i + (int)l - 1, max_j = j + (int)l - 1;
                        continue;
                    }
                }
                score += mat[tseq[i + l] * 5 + qseq[j + l]];
                upd(score, i + (int)l, j + (int)l);
                ++l;
            }
            i += (int)len, j += (int)len;
        } else if (op == 1 || op == 2 || op == 3) {
            score -= o.q + o.e * (int)len;
            if (op == 1) j += (int)len; else i += (int)len;
            upd(score, i, j);
        }
    }
    return max_zdrop > o.zdrop ? 1 : 0;
}

void append_cigar(Reg &r, CigSpan cigar)  // align.c:288-311
{
    if (cigar.empty()) return;
    if (!r.has_p) { r.has_p = true; r.p = Extra(); }
    std::vector<uint32_t> &c = r.p.cigar;
    size_t from = 0;
    if (!c.empty() && (c.back() & 0xf) == (cigar[0] & 0xf)) { c.back() += cigar[0] >> 4 << 4; from = 1; }
    c.insert(c.end(), cigar.begin() + from, cigar.end());
}

// align.c:91-167 -- left-align indels, collapse I/D runs, drop leading I/D
void fix_cigar(Reg &r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
    std::vector<uint32_t> &cg = r.p.cigar;
    int32_t toff = 0, qoff = 0;
    bool to_shrink = false;
    *qshift = *tshift = 0;
    if (cg.size() <= 1) return;
    uint32_t n_cigar = (uint32_t)cg.size();
    for (uint32_t k = 0; k < n_cigar; ++k) {
        const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
        if (len == 0) to_shrink = true;
        if (op == 0) toff += (int)len, qoff += (int)len;
        else if (op == 1 || op == 2) {
            if (k > 0 && k < n_cigar - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
                int l;
                const int prev_len = (int)(cg[k - 1] >> 4);
                if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + (int)len - 1 - l]) break; }
                else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + (int)len - 1 - l]) break; }
                if (l > 0) cg[k - 1] -= (uint32_t)l << 4, cg[k + 1] += (uint32_t)l << 4, qoff -= l, toff -= l;
                if (l == prev_len) to_shrink = true;
            }
            if (op == 1) qoff += (int)len; else toff += (int)len;
        } else if (op == 3) toff += (int)len;
    }
    for (uint32_t k = 0; k + 2 < n_cigar; ++k) {           // 5I6D7I -> one I and one D  (k < n_cigar - 2, unsigned in the reference; n_cigar >= 2 here)
        if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k + 1] & 0xf) == 3) {
            uint32_t l, s[3] = {0, 0, 0};
            for (l = k; l < n_cigar; ++l) {
                const uint32_t op = cg[l] & 0xf;
                if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
                else break;
            }
            if (s[1] > 0 && s[2] > 0 && l - k > 2) {
                cg[k] = s[1] << 4 | 1;
                cg[k + 1] = s[2] << 4 | 2;
                for (k += 2; k < l; ++k) cg[k] &= 0xf;
                to_shrink = true;
            }
            k = l;
        }
    }
    if (to_shrink) {
        uint32_t l = 0;
        for (uint32_t k = 0; k < n_cigar; ++k) if (cg[k] >> 4 != 0) cg[l++] = cg[k];
        n_cigar = l;
        l = 0;
        for (uint32_t k = 0; k < n_cigar; ++k)
            if (k == n_cigar - 1 || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[l++] = cg[k];
            else cg[k + 1] += cg[k] >> 4 << 4;
        n_cigar = l;
    }
    cg.resize(n_cigar);
    if (!cg.empty() && ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2)) {
        const int32_t l = (int32_t)(cg[0] >> 4);
        if ((cg[0] & 0xf) == 1) { r.qs += l; *qshift = l; }      // forward strand only
        else r.rs += l, *tshift = l;
        cg.erase(cg.begin());
    }
}

// align.c:240-286
void update_extra(Reg &r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int q, int e)
{
    if (!r.has_p) return;
    int32_t s = 0, mx = 0, qshift, tshift, toff = 0, qoff = 0;
    fix_cigar(r, qseq, tseq, &qshift, &tshift);
    qseq += qshift, tseq += tshift;
    r.blen = r.mlen = 0;
    for (uint32_t c : r.p.cigar) {
        const uint32_t op = c & 0xf, len = c >> 4;
        if (op == 0) {
            int n_ambi = 0, n_diff = 0;
            uint32_t l = 0;
            // eight equal unambiguous bases at a time: the score only climbs over them (a > 0, s >= 0 on entry), so the
            // running maximum is the value after the last one -- the same numbers as the base-by-base loop below
            const int a_match = mat[0];
            if (a_match > 0 && mat[6] == a_match && mat[12] == a_match && mat[18] == a_match)
                while (l + 8 <= len) {
                    uint64_t wq, wt;
                    memcpy(&wq, qseq + qoff + l, 8), memcpy(&wt, tseq + toff + l, 8);
                    if (wq != wt || (wq & 0xfcfcfcfcfcfcfcfcull)) {
                        // scalar over this word
                        for (uint32_t e8 = l + 8; l < e8; ++l) {
                            const int cq = qseq[qoff + l], ct = tseq[toff + l];
                            if (ct > 3 || cq > 3) ++n_ambi;
                            else if (ct != cq) ++n_diff;
                            s += mat[ct * 5 + cq];
                            if (s < 0) s = 0; else mx = mx > s ? mx : s;
                        }
                        continue;
                    }
                    s += 8 * a_match;
                    mx = mx > s ? mx : s;
                    l += 8;
                }
            for (; l < len; ++l) {
                const int cq = qseq[qoff + l], ct = tseq[toff + l];
                if (ct > 3 || cq > 3) ++n_ambi;
                else if (ct != cq) ++n_diff;
                s += mat[ct * 5 + cq];
                if (s < 0) s = 0; else mx = mx > s ? mx : s;
            }
            r.blen += (int)len - n_ambi, r.mlen += (int)len - (n_ambi + n_diff), r.p.n_ambi += (uint32_t)n_ambi;
            toff += (int)len, qoff += (int)len;
        } else if (op == 1) {
            int n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l) if (qseq[qoff + l] > 3) ++n_ambi;
            r.blen += (int)len - n_ambi, r.p.n_ambi += (uint32_t)n_ambi;
            s -= q + e * (int)len;
            if (s < 0) s = 0;
            qoff += (int)len;
        } else if (op == 2) {
            int n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l) if (tseq[toff + l] > 3) ++n_ambi;
            r.blen += (int)len - n_ambi, r.p.n_ambi += (uint32_t)n_ambi;
            s -= q + e * (int)len;
            if (s < 0) s = 0;
            toff += (int)len;
        } else if (op == 3) toff += (int)len;
    }
    r.p.dp_max = mx;
}

inline int gap_of(const Anchor *a, int i)      // (dq) - (dr) between anchors i-1 and i, on the low 32 bits as the reference computes it
{
    return (int)((int32_t)a[i].y - (int32_t)a[i - 1].y) - (int)((int32_t)a[i].x - (int32_t)a[i - 1].x);
}

std::vector<int> collect_long_gaps(int as1, int cnt1, const Anchor *a, int min_gap)
{
    std::vector<int> K;
    for (int i = 1; i < cnt1; ++i) {
        const int gap = gap_of(a, as1 + i);
        if (gap < -min_gap || gap > min_gap) K.push_back(i);
    }
    if (K.size() <= 1) K.clear();
    return K;
}

// align.c:386-421
void filter_bad_seeds(int as1, int cnt1, Anchor *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
    const std::vector<int> K = collect_long_gaps(as1, cnt1, a, min_gap);
    const int n = (int)K.size();
    if (n == 0) return;
    int mx = 0, max_st = -1, max_en = -1;
    for (int k = 0;; ++k) {
        if (k == n || k >= max_en) {
            if (max_en > 0)
                for (int i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= SEED_IGNORE;
            mx = 0, max_st = max_en = -1;
            if (k == n) break;
        }
        int i = K[k], n_ins = 0, n_del = 0, max_diff = 0, max_diff_l = -1;
        int gap = gap_of(a, as1 + i);
        if (gap > 0) n_ins += gap; else n_del += -gap;
        const int qs = (int32_t)a[as1 + i - 1].y, rs = (int32_t)a[as1 + i - 1].x;
        for (int l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            const int j = K[l];
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = gap_of(a, as1 + j);
            if (gap > 0) n_ins += gap; else n_del += -gap;
            const int diff = n_ins + n_del - abs(n_ins - n_del);
            if (max_diff < diff) max_diff = diff, max_diff_l = l;
        }
        if (max_diff > diff_thres && max_diff > mx) mx = max_diff, max_st = k, max_en = max_diff_l;
    }
}

// align.c:423-457
void filter_bad_seeds_alt(int as1, int cnt1, Anchor *a, int min_gap, int max_ext)
{
    const std::vector<int> K = collect_long_gaps(as1, cnt1, a, min_gap);
    const int n = (int)K.size();
    for (int k = 0; k < n;) {
        const int i = K[k];
        int l;
        int gap1 = gap_of(a, as1 + i);
        int re1 = (int32_t)a[as1 + i].x, qe1 = (int32_t)a[as1 + i].y;
        gap1 = gap1 > 0 ? gap1 : -gap1;
        for (l = k + 1; l < n; ++l) {
            const int j = K[l];
            if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
            int gap2 = gap_of(a, as1 + j);
            const int q_span_pre = span_of(a[as1 + j - 1]);
            const int rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre, qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
            const int m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
            gap2 = gap2 > 0 ? gap2 : -gap2;
            if (m > gap1 + gap2) break;
            re1 = (int32_t)a[as1 + j].x, qe1 = (int32_t)a[as1 + j].y;
            gap1 = gap2;
        }
        if (l > k + 1) {
            const int end = K[l - 1];
            for (int j = K[k]; j < end; ++j) a[as1 + j].y |= SEED_IGNORE;
            a[as1 + end].y |= SEED_LONG_JOIN;
        }
        k = l;
    }
}

// align.c:459-493
void fix_bad_ends(const Reg &r, const Anchor *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
    *as = r.as, *cnt = r.cnt;
    if (r.cnt < 3) return;
    int32_t m, l;
    m = l = span_of(a[r.as]);
    for (int32_t i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
        const int32_t q_span = span_of(a[i]);
        if (a[i].y & SEED_LONG_JOIN) break;
        const int32_t lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *as = i;
        l += mn;
        m += mn < q_span ? mn : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
    }
    *cnt = r.as + r.cnt - *as;
    m = l = span_of(a[r.as + r.cnt - 1]);
    for (int32_t i = r.as + r.cnt - 2; i > *as; --i) {
        const int32_t q_span = span_of(a[i + 1]);
        if (a[i + 1].y & SEED_LONG_JOIN) break;
        const int32_t lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *cnt = i + 1 - *as;
        l += mn;
        m += mn < q_span ? mn : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
    }
}

}  // namespace

void AlignJob::start(const RefIndex *r, const char *q, int ql, const Opt &o)
{
    ref = r, qstr = q, qlen = ql, opt = o;
    finished = false, seeded = false, prepared = false, chained = false, cur = 0;
    cf = cp = nullptr, avg_qspan = 0.f;
    pre_mz = nullptr, n_pre_mz = 0;
    regs.clear(); a.clear(); cache.clear();
}

// One region through mm_align1 (align.c:565-795), plan-then-execute: the plan pass asks the
// cache for every DP the region can need; only when all are present does the execute pass
// run the reference's logic and commit.  Returns false while results are missing.
static bool align1(AlignJob &J, Reg &r_io, Reg &r2, bool plan_only = false)
{
    const Opt &opt = J.opt;
    const RefIndex &mi = *J.ref;
    Anchor *a = J.a.data();
    const int qlen = J.qlen, n_a = J.n_a;
    const uint8_t *qseq0 = J.qseq.data();
    r2 = Reg();
    r2.cnt = 0;
    if (r_io.cnt == 0) return true;
    const Mat5 mat = simple_mat(opt.a, opt.b, opt.sc_ambi);
    const int bw = (int)(opt.bw * 1.5 + 1.);
    const int ref_len = (int)mi.len;

    // the seed filters write flags into a[]; a later replay must start from the same a[]
    std::vector<uint64_t> saved(r_io.cnt);
    for (int i = 0; i < r_io.cnt; ++i) saved[i] = a[r_io.as + i].y;
    auto restore = [&]() { for (int i = 0; i < r_io.cnt; ++i) a[r_io.as + i].y = saved[i]; };

    Reg r = r_io;
    int32_t as1, cnt1, rs, qs, re, qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, l;
    fix_bad_ends(r, a, opt.bw, opt.min_chain_score * 2, &as1, &cnt1);
    filter_bad_seeds(as1, cnt1, a, 10, 40, opt.max_gap >> 1, 10);
    filter_bad_seeds_alt(as1, cnt1, a, 30, opt.max_gap >> 1);
    auto adjust = [&](const Anchor &p, int32_t *rr, int32_t *qq) { *rr = (int32_t)p.x - (mi.k >> 1); *qq = (int32_t)p.y - (mi.k >> 1); };   // mm_adjust_minier, no HPC
    adjust(a[as1], &rs, &qs);
    adjust(a[as1 + cnt1 - 1], &re, &qe);

    // DP windows (align.c:617-677)
    rs0 = (int32_t)a[r.as].x + 1 - span_of(a[r.as]);
    qs0 = (int32_t)a[r.as].y + 1 - span_of(a[r.as]);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    {
        int i;
        for (i = r.as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
            const int32_t x = (int32_t)a[i].x + 1 - span_of(a[i]), y = (int32_t)a[i].y + 1 - span_of(a[i]);
            if (x < rs0 && y < qs0) {
                if (++l > opt.min_cnt) {
                    l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                    rs1 = rs0 - l, qs1 = qs0 - l;
                    if (rs1 < 0) rs1 = 0;
                    break;
                }
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < opt.max_gap ? qs : opt.max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
        l = l < opt.max_gap ? l : opt.max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else rs0 = rs, qs0 = qs;
    re0 = (int32_t)a[r.as + r.cnt - 1].x + 1;
    qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
    re1 = ref_len, qe1 = qlen;
    {
        int i;
        for (i = r.as + r.cnt, l = 0; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
            const int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
            if (x > re0 && y > qe0) {
                if (++l > opt.min_cnt) {
                    l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                    re1 = re0 + l, qe1 = qe0 + l;
                    break;
                }
            }
        }
    }
    if (qe < qlen && re < ref_len) {
        l = qlen - qe < opt.max_gap ? qlen - qe : opt.max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * opt.a > opt.q ? (l * opt.a - opt.q) / opt.e : 0;
        l = l < opt.max_gap ? l : opt.max_gap;
        l = l < ref_len - re ? l : ref_len - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else re0 = re, qe0 = qe;

    const uint8_t *tseq_all = mi.seq.data();
    const bool do_left = qs > 0 && rs > 0;
    const DpKey left_key = {qs0, qs, rs0, rs, bw, r.split_inv ? opt.zdrop_inv : opt.zdrop, opt.end_bonus, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR};

    // the gap-fill windows depend only on the anchors
    struct Fill { int i; DpKey k1; bool zdrop = false; };
    std::vector<Fill> fills;
    {
        int32_t frs = rs, fqs = qs, fre, fqe;
        for (int i = 1; i < cnt1; ++i) {
            if ((a[as1 + i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
            adjust(a[as1 + i], &fre, &fqe);
            if (i == cnt1 - 1 || (a[as1 + i].y & SEED_LONG_JOIN) || (fqe - fqs >= opt.min_ksw_len && fre - frs >= opt.min_ksw_len)) {
                int bw1 = bw;
                if (a[as1 + i].y & SEED_LONG_JOIN) bw1 = fqe - fqs > fre - frs ? fqe - fqs : fre - frs;
                fills.push_back(Fill{i, {fqs, fqe, frs, fre, bw1, opt.zdrop, -1, EZ_APPROX_MAX}, false});
                frs = fre, fqs = fqe;
            }
        }
    }
    // ---- plan ----
    bool missing = false;
    if (do_left && !J.cache.get(left_key)) missing = true;
    for (Fill &f : fills) {
        const DpResult *r1 = J.cache.get(f.k1);
        if (!r1) { missing = true; continue; }
        f.zdrop = test_zdrop(opt, qseq0 + f.k1.qs, tseq_all + f.k1.rs, J.cache.cigar(*r1), mat.m) != 0;   // the execute pass below re-uses it
        if (f.zdrop) {
            DpKey k2 = f.k1;
            k2.flag = 0;
            if (!J.cache.get(k2)) missing = true;
        }
    }
    {
        // after the gap-fill loop qe/re are those of the last kept anchor (already computed above)
        if (qe < qe0 && re < re0) {
            const DpKey right_key = {qe, qe0, re, re0, bw, opt.zdrop, opt.end_bonus, EZ_EXTZ_ONLY};
            if (!J.cache.get(right_key)) missing = true;
        }
    }
    if (missing || plan_only) { restore(); return false; }

    // ---- execute ----
    bool dropped = false;
    if (do_left) {
        const DpResult &ez = J.cache.at(left_key);
        if (ez.n_cigar) { append_cigar(r, J.cache.cigar(ez)); r.p.dp_score += (int32_t)ez.max; }
        rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
    } else rs1 = rs, qs1 = qs;
    re1 = rs, qe1 = qs;
    size_t fi = 0;
    for (int i = 1; i < cnt1; ++i) {
        if ((a[as1 + i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
        adjust(a[as1 + i], &re, &qe);
        re1 = re, qe1 = qe;
        if (i == cnt1 - 1 || (a[as1 + i].y & SEED_LONG_JOIN) || (qe - qs >= opt.min_ksw_len && re - rs >= opt.min_ksw_len)) {
            const Fill &f = fills[fi++];
            const DpResult *ez = &J.cache.at(f.k1);
            if (f.zdrop) {                             // test_zdrop of this fill's first result, decided in the plan pass above (qs == f.k1.qs, rs == f.k1.rs)
                DpKey k2 = f.k1;
                k2.flag = 0;
                ez = &J.cache.at(k2);
            }
            if (ez->n_cigar) append_cigar(r, J.cache.cigar(*ez));
            if (ez->zdropped) {
                if (!r.has_p) { r.has_p = true; r.p = Extra(); }
                int j;
                for (j = i - 1; j >= 0; --j)
                    if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
                dropped = true;
                if (j < 0) j = 0;
                r.p.dp_score += (int32_t)ez->max;
                re1 = rs + (ez->max_t + 1);
                qe1 = qs + (ez->max_q + 1);
                if (cnt1 - (j + 1) >= opt.min_cnt) split_reg(r, r2, as1 + j + 1 - r.as, qlen, a);
                break;
            } else r.p.dp_score += ez->score;
            rs = re, qs = qe;
        }
    }
    if (!dropped && qe < qe0 && re < re0) {
        const DpKey right_key = {qe, qe0, re, re0, bw, opt.zdrop, opt.end_bonus, EZ_EXTZ_ONLY};
        const DpResult &ez = J.cache.at(right_key);
        if (ez.n_cigar) { append_cigar(r, J.cache.cigar(ez)); r.p.dp_score += (int32_t)ez.max; }
        re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
    }
    r.rs = rs1, r.re = re1;
    r.qs = qs1, r.qe = qe1;
    if (r.has_p) update_extra(r, qseq0 + qs1, tseq_all + rs1, mat.m, opt.q, opt.e);
    r_io = r;
    return true;
}

std::atomic<uint64_t> g_step_ns[6];
static inline uint64_t prof_now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// query codes, seeds (map.c:232-236 / collect_seed_hits) and what the chaining pass needs besides the anchors
void AlignJob::seed_prepare()
{
    if (prepared) return;
    prepared = true;
    qseq.resize(qlen);
    nt4_codes(qstr, (size_t)qlen, qseq.data());
}

void AlignJob::set_anchors(const Anchor *p, size_t n, float avg)
{
    seed_prepare();
    seeded = true;
    a.assign(p, p + n);
    avg_qspan = avg;
}

void AlignJob::seed()
{
    if (seeded) return;
    seed_prepare();
    seeded = true;
    if (!ref->has_table) { fprintf(stderr, "nsgpu: AlignJob::seed on an index without its lookup table (internal error)\n"); abort(); }
    if (pre_mz) collect_seeds(*ref, pre_mz, n_pre_mz, a);       // sketched by the caller (mm_sketch.hip)
    else {
        std::vector<Anchor> mv;
        if (qlen > 0) mm_sketch(qstr, qlen, ref->w, ref->k, 0, mv);
        collect_seeds(*ref, mv.data(), mv.size(), a);
    }
    avg_qspan = chain_avg_qspan(a);
}

bool AlignJob::step()
{
    if (finished) return true;
    cache.missing.clear();
    uint64_t t0 = prof_now();
    auto lap = [&](int k) { const uint64_t t1 = prof_now(); g_step_ns[k].fetch_add(t1 - t0, std::memory_order_relaxed); t0 = t1; };
    if (!seeded) seed();
    if (!chained) {
        chained = true;
        lap(0);
        if (!a.empty() && !(cf && cp)) {
#ifdef NSGPU_HOST_CHAIN
            own_f.resize(a.size()), own_p.resize(a.size());
            chain_forward_host(opt, a, avg_qspan, own_f.data(), own_p.data());
            cf = own_f.data(), cp = own_p.data();
#else
            fprintf(stderr, "nsgpu: AlignJob::step without the chaining scores of the GPU pass (chain.hip) -- there is no host path in this build\n");
            abort();
#endif
        }
        std::vector<uint64_t> u;
        chain_finish(opt, a, cf, cp, u);
        cf = cp = nullptr;
        lap(1);
        // map.c:290-292: query name is NULL
        uint32_t hash = 0;
        hash ^= wang_hash32((uint32_t)qlen) + wang_hash32((uint32_t)opt.seed);
        hash = wang_hash32(hash);
        gen_regs(hash, qlen, u, a, regs);
        // chain_post (map.c:249-258)
        set_parent(opt, regs);
        select_sub(opt, ref->k * 2, regs);
        join_long(opt, qlen, regs, a);
        if (regs.empty()) { finished = true; return true; }
        n_a = squeeze_a(regs, a);            // mm_align_skeleton prologue (align.c:880)
        cur = 0;
        lap(2);
    }
    while (cur < (int)regs.size()) {
        Reg r2;
        if (!align1(*this, regs[cur], r2)) {
            // results are missing: also collect the requests of the regions behind this one, so that
            // one DP launch serves them all (their anchors are disjoint from this region's)
            for (int j = cur + 1; j < (int)regs.size(); ++j) { Reg tmp; align1(*this, regs[j], tmp, true); }
            lap(3);
            return false;
        }
        if (r2.cnt > 0) regs.insert(regs.begin() + cur + 1, r2);     // mm_insert_reg
        ++cur;
    }
    filter_regs(opt, qlen, regs);
    hit_sort(regs);
    // align_regs epilogue (map.c:264-268)
    set_parent(opt, regs);
    select_sub(opt, ref->k * 2, regs);
    set_sam_pri(regs);
    finished = true;
    lap(4);
    return true;
}

// ConsensusGraph::alignRead after mm_map (src/ConsensusGraph.cpp:219-397): reg[0] -> Edit list + offsets
void edits_from_hit(int hits, int rs, int re, int qs, int qe, int blen, int mlen, int n_ambi, int dp_max, bool has_p,
                    const std::vector<uint32_t> &cigar, const char *ref, size_t ref_len, const char *s, size_t slen, AlnOut &out)
{
    out.reset();
    out.hits = hits;
    if (hits <= 0) return;
    out.rs = rs, out.re = re, out.qs = qs, out.qe = qe, out.blen = blen, out.mlen = mlen;
    out.n_ambi = n_ambi, out.dp_max = dp_max;
    out.cigar.assign(cigar.begin(), cigar.end());
    out.n_cigar = has_p ? (int32_t)cigar.size() : -1;
    const size_t edit_dis = (size_t)(blen - mlen + n_ambi);
    const int aligned_len = qe - qs;
    if (rs > 0 && re < (int64_t)ref_len)
        if (edit_dis / (double)aligned_len >= 1.0 || (double)aligned_len / slen <= 0.0) return;   // ok stays 0
    std::vector<EditOp> &ed = out.edits;
    ed.reserve(cigar.size() * 2 + (size_t)(blen - mlen) * 2 + 64);
    int qpos = qs, rpos = rs;
    out.rel_pos = (int64_t)rs - (int64_t)qs;
    if (rs > 0) {
        out.begin_offset = rs;
        for (int i = 0; i < qs; ++i) ed.push_back({1, (uint8_t)s[i], 0});
    } else out.begin_offset = -(int64_t)qs;
    for (uint32_t c : cigar) {
        const uint32_t op = c & 0xf, len = c >> 4;
        if (op == 0) {
            uint32_t same = 0;
            for (uint32_t k = 0; k < len; ++k) {
                while (k + 8 <= len) {                  // eight equal bases at a time
                    uint64_t wq, wt;
                    memcpy(&wq, s + qpos, 8), memcpy(&wt, ref + rpos, 8);
                    if (wq != wt) break;
                    same += 8, qpos += 8, rpos += 8, k += 8;
                }
                if (k >= len) break;
                if (s[qpos] == ref[rpos]) ++same;
                else {
                    if (same > 0) ed.push_back({0, 0, same});
                    same = 0;
                    ed.push_back({2, (uint8_t)ref[rpos], 0});
                    ed.push_back({1, (uint8_t)s[qpos], 0});
                }
                ++qpos, ++rpos;
            }
            if (same != 0) ed.push_back({0, 0, same});
        } else if (op == 1) {
            for (uint32_t k = 0; k < len; ++k) ed.push_back({1, (uint8_t)s[qpos++], 0});
        } else if (op == 2) {
            for (uint32_t k = 0; k < len; ++k) ed.push_back({2, (uint8_t)ref[rpos++], 0});
        }
    }
    if (re < (int64_t)ref_len) {
        out.end_offset = (int64_t)re - (int64_t)ref_len;
        for (size_t i = (size_t)qe; i < slen; ++i) ed.push_back({1, (uint8_t)s[i], 0});
    } else out.end_offset = (int64_t)slen - qe;
    size_t unchanged = 0;
    for (const EditOp &e : ed) if (e.type == 0) unchanged += e.num;
    out.ok = unchanged != 0;
}

void align_read_result(const AlignJob &job, const char *ref, size_t ref_len, AlnOut &out)
{
    if (job.regs.empty()) { out.reset(); return; }
    const Reg &r = job.regs[0];
    edits_from_hit((int)job.regs.size(), r.rs, r.re, r.qs, r.qe, r.blen, r.mlen, (int)r.p.n_ambi, r.p.dp_max, r.has_p, r.p.cigar, ref, ref_len,
                   job.qstr, (size_t)job.qlen, out);
}

}  // namespace mm2
}  // namespace nsgpu
