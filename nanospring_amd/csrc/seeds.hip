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
// The kernels compute exactly these (round 5; until then one kernel rebuilt a table of the whole reference with positions per alignment): an
// open-addressing table hash -> occurrence count of the reference's minimizers that PERSISTS with the reference and is kept up to date with
// what a splice removed and added (count_update_kernel, which also keeps the histogram of the counts and derives mid_occ, the order
// statistic); the query's minimizers in an LDS table; the reference's list streamed past it, one anchor per same-strand hit; and a bitonic
// sort of (ref pos << 32 | emit index) keys in LDS (seed_kernel).
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
constexpr uint32_t kMaxQry = 2048;              // query minimizers per pair the LDS table takes (reads up to ~50 kb: 104 KB of LDS); beyond: SEED_FLAG_MANY

__device__ __forceinline__ uint32_t slot_of(unsigned long long key, uint32_t bits) { return (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> (64 - bits)); }
// The tables are filled with atomics, which execute in L2: reads go past the L1 as well (agent-scope loads)
template <class T> __device__ __forceinline__ T ld_l2(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- the occurrence counts of a reference's minimizers (mm_idx_str's buckets reduced to what collect_seed_hits needs of them) ---------------
// An open-addressing table hash -> count over the reference's minimizer list, the histogram of the counts and the number of distinct hashes,
// from which mid_occ (mm_idx_cal_max_occ, index.c:164-185) is an order statistic.  The table PERSISTS with its reference (a contig of the
// engine keeps one in HBM): after a splice of the list only the minimizers that left it and the ones that came are applied (a few hundred
// of tens of thousands); rebuilt from the whole list when the reference is new or the table has grown too full (keys whose count went to
// zero stay as keys).  Until round 5 every alignment rebuilt a table of all of the reference's minimizers WITH their positions (0.20 ms per
// launch: 4 MB cleared and 40 k atomic insertions for a 1 Mb contig); positions need no table at all -- seed_kernel streams the list.
// meta: [0] distinct hashes, [1] mid_occ, [2] SEED_FLAG_OCC when the order statistic fell into the histogram's clipped last bin.
// One workgroup owns a job's table for the launch, so the histogram's moves and the number of distinct hashes are kept in LDS (h_delta, nd_delta) and
// folded into the persistent words once: in global memory every insertion hit the same two words -- bin 1 and the count of distinct hashes --, and
// same-address atomics are served one after the other (a rebuild of a 19 k list: 0.2 ms, in the slot's critical path; the kernel trace).
__device__ __forceinline__ void count_change(CountSlot *tab, uint32_t bits, int32_t *h_delta, int32_t *nd_delta, unsigned long long key, bool add)
{
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = slot_of(key, bits);
    for (;;) {
        unsigned long long k = ld_l2(&tab[s].key);
        if (k == 0ull && add) { k = atomicCAS(&tab[s].key, 0ull, key); if (k == 0ull) k = key; }
        if (k == key) break;
        if (k == 0ull) return;                       // (removing a hash that is not there: never for a consistent caller)
        s = (s + 1) & mask;
    }
    if (add) {
        const uint32_t old = atomicAdd(&tab[s].count, 1u);
        if (old) atomicSub(&h_delta[old < kHistBins ? old : kHistBins - 1], 1); else atomicAdd(nd_delta, 1);
        atomicAdd(&h_delta[old + 1 < kHistBins ? old + 1 : kHistBins - 1], 1);
    } else {
        const uint32_t old = atomicSub(&tab[s].count, 1u);
        atomicSub(&h_delta[old < kHistBins ? old : kHistBins - 1], 1);
        if (old > 1) atomicAdd(&h_delta[old - 1 < kHistBins ? old - 1 : kHistBins - 1], 1); else atomicSub(nd_delta, 1);
    }
}

__global__ __launch_bounds__(1024) void count_update_kernel(const CountJob *__restrict__ jobs, float mid_occ_frac)
{
    __shared__ int32_t h_delta[kHistBins];
    __shared__ uint32_t suf[kHistBins];
    __shared__ int32_t nd_delta;
    const CountJob J = jobs[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    h_delta[tid] = 0;
    if (tid == 0) nd_delta = 0;
    if (J.rebuild) {
        const uint32_t n_slots = 1u << J.bits;
        // (cleared with agent-scope stores, which go through to memory like the atomics that follow; a plain store would sit in this XCD's L2, and
        // __threadfence() between the two is a write-back and an invalidation of that whole L2: `buffer_wbl2 sc1` / `buffer_inv sc1`)
        for (uint32_t s = tid; s < n_slots; s += 1024) {
            __hip_atomic_store(&J.tab[s].key, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(&J.tab[s].count), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();               // (the workgroup's stores have been acknowledged: s_waitcnt vmcnt(0) in front of the barrier)
        for (uint32_t i = tid; i < J.n_all; i += 1024) count_change(J.tab, J.bits, h_delta, &nd_delta, (J.all[i].x >> 8) + 1, true);
    } else {
        __syncthreads();
        // (removals first: a hash that leaves and comes back keeps its slot either way; the histogram moves are order-independent)
        for (uint32_t i = tid; i < J.n_rem; i += 1024) count_change(J.tab, J.bits, h_delta, &nd_delta, J.rem[i] + 1, false);
        for (uint32_t i = tid; i < J.n_add; i += 1024) count_change(J.tab, J.bits, h_delta, &nd_delta, J.add[i] + 1, true);
    }
    __syncthreads();
    // the persistent histogram and the number of distinct hashes: what they were (nothing, for a rebuilt table) plus what this launch moved
    const uint32_t h = (J.rebuild ? 0u : ld_l2(&J.hist[tid])) + (uint32_t)h_delta[tid];
    if (J.rebuild || h_delta[tid]) __hip_atomic_store(&J.hist[tid], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t nd = (J.rebuild ? 0u : ld_l2(&J.meta[0])) + (uint32_t)nd_delta;
    if (tid == 0) __hip_atomic_store(&J.meta[0], nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // mid_occ (index.c:164-185): 1 + the (uint32)((1 - f) n_distinct)-th smallest count, 0-based = the bin b with more than `above` counts in the
    // bins from b up and at most `above` in the bins above it (a suffix sum over the 1024 bins, one per thread)
    suf[tid] = h;
    __syncthreads();
    for (uint32_t d = 1; d < kHistBins; d <<= 1) {
        const uint32_t v = tid + d < kHistBins ? suf[tid + d] : 0u;
        __syncthreads();
        suf[tid] += v;
        __syncthreads();
    }
    auto put = [&](uint32_t mid, uint32_t flag) {
        __hip_atomic_store(&J.meta[1], mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&J.meta[2], flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (mid_occ_frac <= 0.f || nd == 0) { if (tid == 0) put(mid_occ_frac <= 0.f ? 0x7fffffffu : 1u, 0u); return; }
    uint32_t kk = (uint32_t)((1. - (double)mid_occ_frac) * (double)nd);
    if (kk >= nd) kk = nd - 1;
    const uint32_t above = nd - 1 - kk;                                                   // counts strictly after it in ascending order
    const uint32_t up = tid + 1 < kHistBins ? suf[tid + 1] : 0u;
    if (suf[tid] > above && up <= above) put(tid + 1, tid == kHistBins - 1 ? SEED_FLAG_OCC : 0u);      // (the clipped bin: exact value unknown)
}

// ---- the anchors of a pair: the query's minimizers in an LDS table, the reference's list streamed past it --------------------------------------
// One workgroup per pair.  (1) every query minimizer into an LDS table hash -> chain of query indices (a hash twice in the query: tandem
// repeats), a hash the reference has at least mid_occ times (its persistent count table) is dropped; (2) the reference's list, 16 bytes per
// minimizer and coalesced, past that table: every same-strand hit is an anchor, written to the pair's own stretch of scratch with its sort key
// in LDS; (3) a bitonic sort of the keys (reference position << 32 | emit index), the list written in order.  What the kernel does not decide it
// flags, as before: two anchors on one reference position (the reference's unstable radix sort orders those), more than kSortCap anchors or
// kMaxQry query minimizers, a reference position beyond 32 bits, no room in the output.
constexpr int kSeedThreads = 512;
struct SeedLds { uint32_t n_slots; };

__global__ __launch_bounds__(kSeedThreads) void seed_kernel(const SeedPair *__restrict__ pairs, mm2::Anchor *__restrict__ tmp, mm2::Anchor *__restrict__ out,
                                                            unsigned long long *__restrict__ counter, unsigned long long capacity, SeedResult *__restrict__ res, uint32_t q_slots,
                                                            uint32_t sort_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // LDS: keys[kSortCap] u64 | qkey[q_slots] u64 | qhead[q_slots] u32 | qnext[q_cap] u32 | qy[q_cap] u32 | qspan[q_cap] u8 (q_cap = q_slots / 2)
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(lds);
    unsigned long long *qkey = keys + kSortCap;
    uint32_t *qhead = reinterpret_cast<uint32_t *>(qkey + q_slots);
    uint32_t *qnext = qhead + q_slots;
    uint32_t *qy = qnext + q_slots / 2;
    uint32_t *qtan = qy + q_slots / 2;           // span | tandem << 8
    __shared__ uint32_t s_emit, s_flag, s_span;
    __shared__ unsigned long long s_base;
    const SeedPair P = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint32_t mid_occ = ld_l2(&P.cnt_meta[1]);
    uint32_t flag0 = ld_l2(&P.cnt_meta[2]);
    if (P.n_qry > q_slots / 2) flag0 |= SEED_FLAG_MANY;
    for (uint32_t s = tid; s < q_slots; s += kSeedThreads) qkey[s] = 0ull, qhead[s] = 0u;
    if (tid == 0) s_emit = 0, s_flag = flag0, s_span = 0;
    __syncthreads();
    const uint32_t qmask = q_slots - 1;
    uint32_t qbits = 0;
    while ((1u << qbits) < q_slots) ++qbits;
    // (1) the query: hash -> chain of indices (index + 1; 0 ends a chain)
    if (!(flag0 & SEED_FLAG_MANY))
    for (uint32_t i = tid; i < P.n_qry; i += kSeedThreads) {
        const mm2::Anchor q = P.qry[i];
        const unsigned long long hq = q.x >> 8, key = hq + 1;
        const bool tandem = (i > 0 && P.qry[i - 1].x >> 8 == hq) || (i + 1 < P.n_qry && P.qry[i + 1].x >> 8 == hq);
        qy[i] = (uint32_t)q.y;
        qtan[i] = (uint32_t)(q.x & 0xff) | (tandem ? 256u : 0u);
        // dropped when the reference has the hash mid_occ times or more (or not at all: nothing to find)
        bool keep = false;
        {
            const uint32_t mask = (1u << P.cnt_bits) - 1;
            uint32_t s = slot_of(key, P.cnt_bits);
            for (;;) {
                const unsigned long long k = ld_l2(&P.cnt_tab[s].key);
                if (k == key) { const uint32_t cnt = ld_l2(&P.cnt_tab[s].count); keep = cnt > 0 && cnt < mid_occ; break; }
                if (k == 0ull) break;
                s = (s + 1) & mask;
            }
        }
        if (!keep) continue;
        uint32_t s = slot_of(key, qbits);
        for (;;) {
            const unsigned long long prev = atomicCAS(&qkey[s], 0ull, key);
            if (prev == 0ull || prev == key) break;
            s = (s + 1) & qmask;
        }
        qnext[i] = atomicExch(&qhead[s], i + 1);
    }
    __syncthreads();
    // (2) the reference's list past the table
    mm2::Anchor *mine = tmp + (size_t)blockIdx.x * kSortCap;
    if (!(flag0 & SEED_FLAG_MANY))
    for (uint32_t i = tid; i < P.n_ref; i += kSeedThreads) {
        const mm2::Anchor rm = P.ref[i];
        const unsigned long long key = (rm.x >> 8) + 1;
        uint32_t s = slot_of(key, qbits);
        uint32_t h = 0;
        for (;;) {
            const unsigned long long k = qkey[s];
            if (k == key) { h = qhead[s]; break; }
            if (k == 0ull) break;
            s = (s + 1) & qmask;
        }
        const unsigned long long r = rm.y;
        for (; h; h = qnext[h - 1]) {
            const uint32_t q_pos = qy[h - 1];
            if ((((uint32_t)r ^ q_pos) & 1u) != 0) continue;                          // reverse-strand seed dropped (MM_F_FOR_ONLY)
            const uint32_t e = atomicAdd(&s_emit, 1u);
            const uint32_t st = qtan[h - 1];
            atomicAdd(&s_span, st & 0xffu);
            if (e >= kSortCap) continue;                                              // (counted: the pair is flagged below)
            const unsigned long long x = (r & 0xffffffff00000000ull) | ((uint32_t)r >> 1);
            mine[e] = mm2::Anchor{x, (unsigned long long)(st & 0xffu) << 32 | (q_pos >> 1) | ((st & 256u) ? mm2::SEED_TANDEM : 0ull)};
            if (x >> 32) atomicOr(&s_flag, SEED_FLAG_WIDE);                           // the sort key holds 32 bits of x
            keys[e] = x << 32 | e;
        }
    }
    __syncthreads();
    const uint32_t total = s_emit;
    if (tid == 0) {
        s_base = atomicAdd(counter, (unsigned long long)total);
        if (total > sort_cap) s_flag |= SEED_FLAG_MANY;          // (sort_cap <= kSortCap: the engine's rule for deferred alignments hands shorter lists back, nsgpu_set_defer)
        if (s_base + total > capacity) s_flag |= SEED_FLAG_CAPACITY;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    if (s_flag & (SEED_FLAG_MANY | SEED_FLAG_CAPACITY | SEED_FLAG_WIDE | SEED_FLAG_OCC)) {
        if (tid == 0) res[blockIdx.x] = SeedResult{base, total, s_flag, (int32_t)mid_occ, 0.f};
        return;
    }
    // (3) bitonic sort of the keys (padded to a power of two with the largest key)
    uint32_t n2 = 1;
    while (n2 < total) n2 <<= 1;
    for (uint32_t e = total + tid; e < n2; e += kSeedThreads) keys[e] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < n2 / 2; t += kSeedThreads) {
                const uint32_t lo = 2 * t - (t & (j - 1)), hi = lo + j;               // the t-th pair at distance j
                const bool up = (lo & k) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) keys[lo] = b, keys[hi] = a;
            }
            __syncthreads();
        }
    // the sorted list; two anchors on one reference position: the reference's tie order is not a property of the set
    bool tie = false;
    for (uint32_t e = tid; e < total; e += kSeedThreads) {
        const unsigned long long kx = keys[e];
        if (e && (keys[e - 1] >> 32) == (kx >> 32)) tie = true;
        out[base + e] = mine[(uint32_t)kx];
    }
    if (tie) atomicOr(&s_flag, SEED_FLAG_TIES);
    __syncthreads();
    if (tid == 0) res[blockIdx.x] = SeedResult{base, total, s_flag, (int32_t)mid_occ, total ? (float)s_span / (float)(long long)total : 0.f};
}

size_t seed_lds_bytes(uint32_t q_slots) { return (size_t)kSortCap * 8 + (size_t)q_slots * 8 + (size_t)q_slots * 4 + 3 * (size_t)(q_slots / 2) * 4 + 64; }

}  // namespace

// Count tables for references that do not keep one (direct API calls; the pairs the caller gave no table): built in the workspace's scratch by
// the same kernel the persistent ones are kept up to date with.  Launches on `st`.
int gpu_count_tables_launch(hipStream_t st, const CountJob *jobs_pinned, uint32_t n_jobs, float mid_occ_frac)
{
    if (n_jobs == 0) return NSGPU_OK;
    hipLaunchKernelGGL(count_update_kernel, dim3(n_jobs), dim3(1024), 0, st, jobs_pinned, mid_occ_frac);
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

uint32_t count_table_bits(uint64_t n_keys)
{
    uint32_t bits = 6;
    while (((uint64_t)1 << bits) < 4 * n_keys + 16) ++bits;
    return bits;
}

// Seeds of a batch of pairs.  ref / qry lists must be readable by the device (pinned host memory or device memory).  The sorted
// anchors of pair i are W.d_out[res[i].base .. + res[i].n) in DEVICE memory (for chain.hip); res (pinned) is valid after
// gpu_seeds_wait.  Pairs with res[i].flags != 0 have no usable list.  A pair without a count table of its reference (cnt_tab == nullptr)
// gets one built here, in the workspace's scratch.
int gpu_seeds_launch(nsgpu_ctx *c, int ws, float mid_occ_frac, std::vector<SeedPair> &pairs, bool with_total)
{
    nsgpu_ctx::SeedWs &W = c->seed_ws[ws];
    const size_t n = pairs.size();
    W.pend = n;
    if (n == 0) return NSGPU_OK;
    if (!W.stream) NS_TRY(role_stream_create(&W.stream, "seeds"));
    uint64_t tab_total = 0, qry_total = 0;
    uint32_t max_q = 0, n_tmp = 0;
    for (SeedPair &p : pairs) {
        NS_CHECK(p.n_ref < (1u << 30), NSGPU_ERR_RANGE, "seeds: a reference with more than 2^30 minimizers");
        if (!p.cnt_tab) { p.cnt_bits = count_table_bits(p.n_ref); p.tab_off = tab_total; tab_total += (uint64_t)1 << p.cnt_bits; ++n_tmp; }
        qry_total += p.n_qry, max_q = std::max(max_q, p.n_qry);
    }
    // anchors: the engine's lists hold about one anchor per query minimizer; room for 4x, more after an overflow (the flagged pairs
    // of that batch go through the host code)
    // (NSGPU_SEED_CAP_FACTOR / NSGPU_SEED_CAP_SLACK: test switches that make the first launches overflow)
    static const uint64_t cap_factor = getenv("NSGPU_SEED_CAP_FACTOR") ? (uint64_t)atoll(getenv("NSGPU_SEED_CAP_FACTOR")) : 4;
    static const uint64_t cap_slack = getenv("NSGPU_SEED_CAP_SLACK") ? (uint64_t)atoll(getenv("NSGPU_SEED_CAP_SLACK")) : 65536;
    const uint64_t want = std::max<uint64_t>(std::max<uint64_t>(W.cap_hint, cap_factor * qry_total + cap_slack), 16);
    if (n_tmp) {
        NS_TRY(W.d_tab.reserve(tab_total * sizeof(CountSlot) + 16));
        NS_TRY(W.d_next.reserve((size_t)n_tmp * (kHistBins + 4) * 4 + 16));           // histogram + meta of every temporary table
        NS_TRY(W.h_jobs.reserve((size_t)n_tmp * sizeof(CountJob)));
        CountJob *jobs = W.h_jobs.as<CountJob>();
        uint32_t j = 0;
        for (SeedPair &p : pairs) {
            if (p.cnt_tab) continue;
            uint32_t *hm = W.d_next.as<uint32_t>() + (size_t)j * (kHistBins + 4);
            p.cnt_tab = W.d_tab.as<CountSlot>() + p.tab_off, p.cnt_meta = hm + kHistBins;
            jobs[j++] = CountJob{const_cast<CountSlot *>(p.cnt_tab), p.cnt_bits, 1u, hm, hm + kHistBins, nullptr, nullptr, p.ref, 0u, 0u, p.n_ref, 0u};
        }
        NS_TRY(gpu_count_tables_launch(W.stream, jobs, n_tmp, mid_occ_frac));
    }
    NS_TRY(W.d_tmp.reserve(n * kSortCap * sizeof(mm2::Anchor)));
    NS_TRY(W.d_out.reserve(want * sizeof(mm2::Anchor)));
    const bool fresh_counter = W.d_counter.p == nullptr;
    NS_TRY(W.d_counter.reserve(16));
    NS_TRY(W.h_pairs.reserve(n * sizeof(SeedPair)));
    NS_TRY(W.h_res.reserve(n * sizeof(SeedResult) + 16));
    W.capacity = want;
    memcpy(W.h_pairs.p, pairs.data(), n * sizeof(SeedPair));
    // (the counter is cleared behind its read-out, gpu_seeds_total: not in front of the seeding kernel, where the fill kernel was 6-9 us of every slot's chain)
    if (fresh_counter) NS_HIP(hipMemsetAsync(W.d_counter.p, 0, 8, W.stream));
    uint32_t q_slots = 256;
    while (q_slots < 2 * std::min<uint32_t>(max_q, kMaxQry)) q_slots <<= 1;
    const size_t lds = seed_lds_bytes(q_slots);
    static LdsAttr attr;
    if (lds > 32768) NS_TRY(attr.raise(lds, reinterpret_cast<const void *>(seed_kernel)));
    // the pair descriptors and the results are read / written in place in pinned memory
    hipLaunchKernelGGL(seed_kernel, dim3((unsigned)n), dim3(kSeedThreads), lds, W.stream, W.h_pairs.as<SeedPair>(), W.d_tmp.as<mm2::Anchor>(), W.d_out.as<mm2::Anchor>(),
                       W.d_counter.as<unsigned long long>(), (unsigned long long)want, W.h_res.as<SeedResult>(), q_slots,
                       c->defer_slots && c->defer_anchors < kSortCap ? c->defer_anchors : kSortCap);
    NS_HIP(hipGetLastError());
    if (with_total) NS_TRY(gpu_seeds_total(c, ws));
    return NSGPU_OK;
}

// the launch's anchor total to the host (behind the results) and the counter cleared for the workspace's next launch; a caller that puts the chaining
// kernel right behind the seeding kernel asks for it behind that (gpu_seeds_chain_launch)
int gpu_seeds_total(nsgpu_ctx *c, int ws)
{
    nsgpu_ctx::SeedWs &W = c->seed_ws[ws];
    if (W.pend == 0) return NSGPU_OK;
    NS_HIP(hipMemcpyAsync(W.h_res.as<uint8_t>() + W.pend * sizeof(SeedResult), W.d_counter.p, 8, hipMemcpyDeviceToHost, W.stream));
    NS_HIP(hipMemsetAsync(W.d_counter.p, 0, 8, W.stream));
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

// with_total = false: the caller puts more kernels behind the chaining kernel (the alignment plan) and calls gpu_seeds_total behind those
int gpu_seeds_chain_launch(nsgpu_ctx *c, int ws, int chain_ws, const mm2::Opt &opt, std::vector<SeedPair> &pairs, bool with_total)
{
    NS_TRY(gpu_seeds_launch(c, ws, opt.mid_occ_frac, pairs, false));
    nsgpu_ctx::SeedWs &W = c->seed_ws[ws];
    uint32_t max_q = 0;
    for (const SeedPair &p : pairs) max_q = std::max(max_q, p.n_qry);
    NS_TRY(gpu_chain_launch_seeded(c, chain_ws, W.stream, opt, W.d_out.as<mm2::Anchor>(), W.h_res.as<SeedResult>(), pairs.size(), W.capacity, max_q));
    if (with_total) NS_TRY(gpu_seeds_total(c, ws));
    return NSGPU_OK;
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
