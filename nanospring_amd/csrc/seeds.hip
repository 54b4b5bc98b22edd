// Seeds on the GPU: for a batch of (reference minimizers, query minimizers) pairs, what minimap2's mm_idx_str + mm_idx_cal_max_occ
// + collect_matches / collect_seed_hits leave for the chaining (minimap2/index.c:164-248,
// minimap2/map.c:90-123, 215-247 with MM_F_FOR_ONLY, map.c:139-145) -- the per-reference index, the occurrence cut-off and the
// sorted anchor list of every pair -- one workgroup per pair, no host lookups.
//
// The reference builds a bucketed hash index of the consensus per candidate read, looks every query minimizer up, emits one
// anchor per same-strand occurrence and radix-sorts the anchors by reference position.  All of that only defines, per pair,
//   (1) the multiset  { (ref pos, query pos, span, tandem) : ref minimizer and query minimizer share a hash, same strand,
//                       occurrences of the hash on the reference < mid_occ },
//   (2) mid_occ = 1 + the (uint32)((1 - 2e-4f) * n_distinct)-th smallest occurrence count (0-based) over the distinct hashes, and
//   (3) the order of the list: ascending reference position -- unique when no two anchors share one.
// The kernel computes exactly these: an open-addressing table over the reference minimizers in global scratch (64-bit CAS claims a
// slot per distinct hash, occurrences are chained through a next[] array), the order statistic from a histogram of the slot counts,
// one count pass and one emit pass over the query minimizers, and a bitonic sort of (ref pos << 32 | emit index) keys in LDS.
// Pairs the kernel will not decide are FLAGGED and redone by the caller with the literal host code (mm2.cpp): two anchors on one
// reference position (the reference's radix sort is unstable and its tie order is what the chaining sees; needs a query with a
// repeated minimizer hitting the same spot -- tandem repeats), more than kSortCap anchors, an occurrence count above the histogram.
// tests/test_seeds_gpu.py compares anchors, mid_occ and mean span with the host code on the alignment cases and on synthetic
// repeat-rich pairs, and checks that the flags fire where they must.
#include "common.hpp"
#include "host_util.hpp"
#include "mm2.hpp"
#include <algorithm>
#include <cstring>

namespace nsgpu {
namespace {

constexpr uint32_t kSortCap = 4096;             // anchors per pair the LDS sort takes (32 KB of keys)
constexpr uint32_t kHistBins = 1024;

struct Slot { unsigned long long key; uint32_t count, head; };   // key = hash + 1 (0: empty); head = 1 + index of the last inserted occurrence

__device__ __forceinline__ uint32_t slot_of(unsigned long long key, uint32_t bits) { return (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> (64 - bits)); }

// The table is cleared with plain stores and filled with atomics, which execute in L2: reads go past the L1 as well (agent-scope
// loads), so that no line cached from the clearing pass can be seen.
template <class T> __device__ __forceinline__ T ld_l2(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// looks `key` up; returns the slot index or ~0u
__device__ __forceinline__ uint32_t find_slot(const Slot *tab, uint32_t bits, unsigned long long key)
{
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = slot_of(key, bits);
    for (;;) {
        const unsigned long long k = ld_l2(&tab[s].key);
        if (k == key) return s;
        if (k == 0) return ~0u;
        s = (s + 1) & mask;
    }
}

// kThreads: a workgroup per pair, and every phase is a loop of dependent L2 round trips per thread -- the more threads the shorter: 1024
// (index + seeds wait 0.58 instead of 0.71 s per cfg2 step at the one-group schedule, 80 pairs per launch on an empty chip; the
// 1024-builder schedule with 256 pairs per launch gains 1-3 % as well).
template <int kThreads>
__global__ __launch_bounds__(kThreads) void seed_kernel(const SeedPair *__restrict__ pairs, Slot *__restrict__ tabs, uint32_t *__restrict__ nexts, unsigned long long *__restrict__ ys_all,
                                                        mm2::Anchor *__restrict__ tmp, mm2::Anchor *__restrict__ out, unsigned long long *__restrict__ counter,
                                                        unsigned long long capacity, SeedResult *__restrict__ res, float mid_occ_frac)
{
    __shared__ uint32_t hist[kHistBins];
    __shared__ unsigned long long keys[kSortCap];
    __shared__ uint32_t s_nd, s_total, s_emit, s_mid, s_flag, s_span;
    __shared__ unsigned long long s_base;
    const SeedPair P = pairs[blockIdx.x];
    const int tid = threadIdx.x;
    Slot *tab = tabs + P.tab_off;
    uint32_t *next = nexts + P.next_off;
    unsigned long long *ys = ys_all + P.next_off;          // the occurrences' y (rid << 32 | pos << 1 | strand), next to the chain links
    const uint32_t n_slots = 1u << P.tab_bits;
    for (uint32_t s = tid; s < n_slots; s += kThreads) tab[s] = Slot{0ull, 0u, 0u};
    for (uint32_t b = tid; b < kHistBins; b += kThreads) hist[b] = 0;
    if (tid == 0) s_nd = s_total = s_emit = s_flag = s_span = 0;
    __syncthreads();
    // (1) the reference's minimizers into the table
    const uint32_t mask = n_slots - 1;
    for (uint32_t i = tid; i < P.n_ref; i += kThreads) {
        const mm2::Anchor rm = P.ref[i];
        const unsigned long long key = (rm.x >> 8) + 1;
        ys[i] = rm.y;
        uint32_t s = slot_of(key, P.tab_bits);
        for (;;) {
            const unsigned long long prev = atomicCAS(&tab[s].key, 0ull, key);
            if (prev == 0ull || prev == key) break;
            s = (s + 1) & mask;
        }
        atomicAdd(&tab[s].count, 1u);
        next[i] = atomicExch(&tab[s].head, i + 1);
    }
    __syncthreads();
    // (2) mid_occ (index.c:164-185): n_distinct and the histogram of the occurrence counts
    {
        uint32_t nd = 0;
        for (uint32_t s = tid; s < n_slots; s += kThreads) {
            const uint32_t cnt = ld_l2(&tab[s].count);
            if (cnt) { ++nd; atomicAdd(&hist[cnt < kHistBins ? cnt : kHistBins - 1], 1u); }
        }
        atomicAdd(&s_nd, nd);
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t nd = s_nd;
        uint32_t mid = 1;
        if (mid_occ_frac <= 0.f) mid = 0x7fffffffu;
        else if (nd) {
            uint32_t kk = (uint32_t)((1. - (double)mid_occ_frac) * (double)nd);           // 0-based rank, counted from the smallest
            if (kk >= nd) kk = nd - 1;
            uint32_t above = nd - 1 - kk;                                                 // counts strictly after it in ascending order
            uint32_t b = kHistBins - 1;
            while (hist[b] <= above) above -= hist[b], --b;                               // from the largest count down
            if (b == kHistBins - 1) s_flag |= SEED_FLAG_OCC;                              // the clipped bin: exact value unknown
            mid = b + 1;
        }
        s_mid = mid;
    }
    __syncthreads();
    const uint32_t mid_occ = s_mid;
    // (3) how many anchors: query minimizers whose hash occurs fewer than mid_occ times, one per same-strand occurrence
    {
        uint32_t cnt = 0, span = 0;
        for (uint32_t i = tid; i < P.n_qry; i += kThreads) {
            const mm2::Anchor q = P.qry[i];
            const uint32_t s = find_slot(tab, P.tab_bits, (q.x >> 8) + 1);
            if (s == ~0u || ld_l2(&tab[s].count) >= mid_occ) continue;
            const uint32_t q_pos = (uint32_t)q.y;
            for (uint32_t h = ld_l2(&tab[s].head); h; h = next[h - 1])
                if ((((uint32_t)ys[h - 1] ^ q_pos) & 1u) == 0) ++cnt, span += (uint32_t)(q.x & 0xff);
        }
        atomicAdd(&s_total, cnt);
        atomicAdd(&s_span, span);
    }
    __syncthreads();
    const uint32_t total = s_total;
    if (tid == 0) {
        s_base = atomicAdd(counter, (unsigned long long)total);
        if (total > kSortCap) s_flag |= SEED_FLAG_MANY;
        if (s_base + total > capacity) s_flag |= SEED_FLAG_CAPACITY;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    if (s_flag & (SEED_FLAG_MANY | SEED_FLAG_CAPACITY)) {
        if (tid == 0) res[blockIdx.x] = SeedResult{base, total, s_flag, (int32_t)mid_occ, 0.f};
        return;
    }
    // (4) emit (any order) into tmp, the sort keys into LDS
    for (uint32_t i = tid; i < P.n_qry; i += kThreads) {
        const mm2::Anchor q = P.qry[i];
        const unsigned long long hq = q.x >> 8;
        const uint32_t s = find_slot(tab, P.tab_bits, hq + 1);
        if (s == ~0u || ld_l2(&tab[s].count) >= mid_occ) continue;
        const uint32_t q_pos = (uint32_t)q.y;
        const bool tandem = (i > 0 && P.qry[i - 1].x >> 8 == hq) || (i + 1 < P.n_qry && P.qry[i + 1].x >> 8 == hq);
        const unsigned long long y = (unsigned long long)(q.x & 0xff) << 32 | (q_pos >> 1) | (tandem ? mm2::SEED_TANDEM : 0ull);
        for (uint32_t h = ld_l2(&tab[s].head); h; h = next[h - 1]) {
            const unsigned long long r = ys[h - 1];
            if ((((uint32_t)r ^ q_pos) & 1u) != 0) continue;                          // reverse-strand seed dropped (MM_F_FOR_ONLY)
            const uint32_t e = atomicAdd(&s_emit, 1u);
            const unsigned long long x = (r & 0xffffffff00000000ull) | ((uint32_t)r >> 1);
            tmp[base + e] = mm2::Anchor{x, y};
            if (x >> 32) atomicOr(&s_flag, SEED_FLAG_WIDE);                           // the sort key holds 32 bits of x
            keys[e] = x << 32 | e;
        }
    }
    __syncthreads();
    if (s_flag & SEED_FLAG_WIDE) {
        if (tid == 0) res[blockIdx.x] = SeedResult{base, total, s_flag, (int32_t)mid_occ, 0.f};
        return;
    }
    // (5) bitonic sort of the keys (padded to a power of two with the largest key)
    uint32_t n2 = 1;
    while (n2 < total) n2 <<= 1;
    for (uint32_t e = total + tid; e < n2; e += kThreads) keys[e] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < n2 / 2; t += kThreads) {
                const uint32_t lo = 2 * t - (t & (j - 1)), hi = lo + j;               // the t-th pair at distance j
                const bool up = (lo & k) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) keys[lo] = b, keys[hi] = a;
            }
            __syncthreads();
        }
    // (6) the sorted list; two anchors on one reference position: the reference's tie order is not a property of the set
    bool tie = false;
    for (uint32_t e = tid; e < total; e += kThreads) {
        const unsigned long long kx = keys[e];
        if (e && (keys[e - 1] >> 32) == (kx >> 32)) tie = true;
        out[base + e] = tmp[base + (uint32_t)kx];
    }
    if (tie) atomicOr(&s_flag, SEED_FLAG_TIES);
    __syncthreads();
    if (tid == 0) res[blockIdx.x] = SeedResult{base, total, s_flag, (int32_t)mid_occ, total ? (float)s_span / (float)(long long)total : 0.f};
}

}  // namespace

// Seeds of a batch of pairs.  ref / qry lists must be readable by the device (pinned host memory or device memory).  The sorted
// anchors of pair i are W.d_out[res[i].base .. + res[i].n) in DEVICE memory (for chain.hip); res (pinned) is valid after
// gpu_seeds_wait.  Pairs with res[i].flags != 0 have no usable list.
int gpu_seeds_launch(nsgpu_ctx *c, int ws, float mid_occ_frac, std::vector<SeedPair> &pairs)
{
    nsgpu_ctx::SeedWs &W = c->seed_ws[ws];
    const size_t n = pairs.size();
    W.pend = n;
    if (n == 0) return NSGPU_OK;
    if (!W.stream) NS_TRY(role_stream_create(&W.stream, "seeds"));
    uint64_t tab_total = 0, next_total = 0, qry_total = 0;
    for (SeedPair &p : pairs) {
        uint32_t bits = 4;
        while (((uint64_t)1 << bits) < 2 * (uint64_t)p.n_ref + 2) ++bits;
        NS_CHECK(bits <= 31, NSGPU_ERR_RANGE, "seeds: a reference with more than 2^30 minimizers");
        p.tab_bits = bits, p.tab_off = tab_total, p.next_off = next_total;
        tab_total += (uint64_t)1 << bits, next_total += p.n_ref, qry_total += p.n_qry;
    }
    // anchors: the engine's lists hold about one anchor per query minimizer; room for 4x, more after an overflow (the flagged pairs
    // of that batch go through the host code)
    // (NSGPU_SEED_CAP_FACTOR / NSGPU_SEED_CAP_SLACK: test switches that make the first launches overflow)
    static const uint64_t cap_factor = getenv("NSGPU_SEED_CAP_FACTOR") ? (uint64_t)atoll(getenv("NSGPU_SEED_CAP_FACTOR")) : 4;
    static const uint64_t cap_slack = getenv("NSGPU_SEED_CAP_SLACK") ? (uint64_t)atoll(getenv("NSGPU_SEED_CAP_SLACK")) : 65536;
    const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(W.cap_hint, cap_factor * qry_total + cap_slack), 16);
    NS_TRY(W.d_tab.reserve(tab_total * sizeof(Slot)));
    NS_TRY(W.d_next.reserve(next_total * sizeof(uint32_t) + 16));
    NS_TRY(W.d_ys.reserve(next_total * sizeof(uint64_t) + 16));
    NS_TRY(W.d_tmp.reserve(want * sizeof(mm2::Anchor)));
    NS_TRY(W.d_out.reserve(want * sizeof(mm2::Anchor)));
    NS_TRY(W.d_counter.reserve(16));
    NS_TRY(W.h_pairs.reserve(n * sizeof(SeedPair)));
    NS_TRY(W.h_res.reserve(n * sizeof(SeedResult) + 16));
    W.capacity = want;
    memcpy(W.h_pairs.p, pairs.data(), n * sizeof(SeedPair));
    NS_HIP(hipMemsetAsync(W.d_counter.p, 0, 8, W.stream));
    // the pair descriptors and the results are read / written in place in pinned memory
    hipLaunchKernelGGL(seed_kernel<1024>, dim3((unsigned)n), dim3(1024), 0, W.stream, W.h_pairs.as<SeedPair>(), W.d_tab.as<Slot>(), W.d_next.as<uint32_t>(),
                       W.d_ys.as<unsigned long long>(), W.d_tmp.as<mm2::Anchor>(), W.d_out.as<mm2::Anchor>(), W.d_counter.as<unsigned long long>(), (unsigned long long)want,
                       W.h_res.as<SeedResult>(), mid_occ_frac);
    NS_HIP(hipGetLastError());
    NS_HIP(hipMemcpyAsync(W.h_res.as<uint8_t>() + n * sizeof(SeedResult), W.d_counter.p, 8, hipMemcpyDeviceToHost, W.stream));
    return NSGPU_OK;
}

int gpu_seeds_wait(nsgpu_ctx *c, int ws, const SeedResult *&res, const mm2::Anchor *&d_anchors)
{
    nsgpu_ctx::SeedWs &W = c->seed_ws[ws];
    res = nullptr, d_anchors = nullptr;
    if (W.pend == 0) return NSGPU_OK;
    NS_HIP(stream_wait_short(W.stream));
    res = W.h_res.as<SeedResult>();
    d_anchors = W.d_out.as<mm2::Anchor>();
    unsigned long long used = 0;
    memcpy(&used, W.h_res.as<uint8_t>() + W.pend * sizeof(SeedResult), 8);
    if (used > W.capacity) W.cap_hint = used + used / 2;          // next batch
    return NSGPU_OK;
}

int gpu_chain_launch_seeded(nsgpu_ctx *c, int ws, hipStream_t stream, const mm2::Opt &opt, const mm2::Anchor *d_anchors, SeedResult *res, size_t n_pairs,
                            uint64_t capacity, uint32_t max_n_qry);
void gpu_chain_results_seeded(nsgpu_ctx *c, int ws, const mm2::Anchor *&a, const int32_t *&f, const int32_t *&p);

int gpu_seeds_chain_launch(nsgpu_ctx *c, int ws, int chain_ws, const mm2::Opt &opt, std::vector<SeedPair> &pairs)
{
    NS_TRY(gpu_seeds_launch(c, ws, opt.mid_occ_frac, pairs));
    nsgpu_ctx::SeedWs &W = c->seed_ws[ws];
    uint32_t max_q = 0;
    for (const SeedPair &p : pairs) max_q = std::max(max_q, p.n_qry);
    return gpu_chain_launch_seeded(c, chain_ws, W.stream, opt, W.d_out.as<mm2::Anchor>(), W.h_res.as<SeedResult>(), pairs.size(), W.capacity, max_q);
}

int gpu_seeds_chain_wait(nsgpu_ctx *c, int ws, int chain_ws, const SeedResult *&res, const mm2::Anchor *&a, const int32_t *&f, const int32_t *&p)
{
    const mm2::Anchor *d_a = nullptr;
    NS_TRY(gpu_seeds_wait(c, ws, res, d_a));
    gpu_chain_results_seeded(c, chain_ws, a, f, p);
    return NSGPU_OK;
}

}  // namespace nsgpu

using namespace nsgpu;

// Index + seeds for a batch of pairs on their own (the contig engine and nsgpu_align_batch run the same kernel): reference list r is
// ref_xy[2*j], ref_xy[2*j+1] for j in ref_off[r] .. ref_off[r+1], query list i likewise, pair i is (pair_ref[i], query i).
extern "C" int nsgpu_seed_anchors(nsgpu_ctx *c, const uint64_t *ref_xy, const uint64_t *ref_off, uint32_t n_refs, const uint64_t *qry_xy, const uint64_t *qry_off,
                                  const uint32_t *pair_ref, uint32_t n_pairs, uint64_t **xy_out, uint64_t **off_out, int32_t *mid_occ_out, uint32_t *flags_out,
                                  float *avg_out)
{
    NS_CHECK(c && ref_off && qry_off && xy_out && off_out && (n_pairs == 0 || pair_ref), NSGPU_ERR_ARG, "nsgpu_seed_anchors: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    nsgpu_ctx::SeedWs &W = c->seed_ws[0];
    const uint64_t n_r = ref_off[n_refs], n_q = qry_off[n_pairs];
    NS_TRY(W.h_ref.reserve((n_r + n_q) * sizeof(mm2::Anchor) + 16));
    mm2::Anchor *hr = W.h_ref.as<mm2::Anchor>(), *hq = hr + n_r;
    if (n_r) memcpy(hr, ref_xy, n_r * sizeof(mm2::Anchor));
    if (n_q) memcpy(hq, qry_xy, n_q * sizeof(mm2::Anchor));
    std::vector<SeedPair> pairs(n_pairs);
    for (uint32_t i = 0; i < n_pairs; ++i) {
        NS_CHECK(pair_ref[i] < n_refs, NSGPU_ERR_ARG, "pair %u refers to reference list %u of %u", i, pair_ref[i], n_refs);
        const uint32_t r = pair_ref[i];
        NS_CHECK(ref_off[r + 1] - ref_off[r] < (1ull << 30) && qry_off[i + 1] - qry_off[i] < (1ull << 31), NSGPU_ERR_RANGE, "list too long");
        pairs[i].ref = hr + ref_off[r], pairs[i].n_ref = (uint32_t)(ref_off[r + 1] - ref_off[r]);
        pairs[i].qry = hq + qry_off[i], pairs[i].n_qry = (uint32_t)(qry_off[i + 1] - qry_off[i]);
    }
    NS_TRY(gpu_seeds_launch(c, 0, 2e-4f, pairs));
    const SeedResult *res = nullptr;
    const mm2::Anchor *d_a = nullptr;
    NS_TRY(gpu_seeds_wait(c, 0, res, d_a));
    uint64_t *of = (uint64_t *)malloc(((size_t)n_pairs + 1) * 8);
    NS_CHECK(of, NSGPU_ERR_NOMEM, "malloc failed");
    of[0] = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        of[i + 1] = of[i] + (res[i].flags ? 0 : res[i].n);
        if (mid_occ_out) mid_occ_out[i] = res[i].mid_occ;
        if (flags_out) flags_out[i] = res[i].flags;
        if (avg_out) avg_out[i] = res[i].avg;
    }
    uint64_t *xy = (uint64_t *)malloc((of[n_pairs] * 2 + 1) * 8);
    if (!xy) { free(of); set_error("malloc failed"); return NSGPU_ERR_NOMEM; }
    for (uint32_t i = 0; i < n_pairs; ++i)
        if (!res[i].flags && res[i].n) NS_HIP(hipMemcpy(xy + 2 * of[i], d_a + res[i].base, (size_t)res[i].n * sizeof(mm2::Anchor), hipMemcpyDeviceToHost));
    *xy_out = xy, *off_out = of;
    return NSGPU_OK;
}
