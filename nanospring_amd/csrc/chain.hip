// Chaining scores on the GPU: the forward pass of minimap2's mm_chain_dp (chain.c:43-92 of the version the reference vendors;
// SURVEY §8 row a14e) for a batch of anchor lists, one wavefront per list.
//
// For anchor i the reference walks its predecessors j = i-1 .. st in order and keeps
//     max_f  = the running best of  f[j] + gain(i, j)            (a new best only on a STRICTLY larger score),
//     n_skip = a counter: -1 (floored at 0) on a new best, +1 when j is not a new best and some earlier-visited j' had p[j'] == j,
// and stops as soon as n_skip exceeds max_skip.  The walk is sequential in j, but every quantity is a prefix function of
// per-j values that do not depend on the walk:
//     * new best at j       <=>  score(j) > max(q_span, max over visited j' of score(j'))              -- an exclusive prefix max;
//     * "marked" at j       <=>  some admissible j' > j in the window has p[j'] == j                        (*);
//     * n_skip after j      =    the composition of  n -> max(n - 1, 0)  (new best),  n -> n + 1  (marked, not a new best) and
//                                the identity: maps of the form n -> max(n + a, b), which are closed under composition
//                                ((a1,b1) then (a2,b2) = (a1 + a2, max(b1 + a2, b2))) -- an inclusive prefix scan of pairs.
// So a wave takes 64 predecessors at a time (lane l holds j = hi - l), computes the scans across its lanes, finds the first lane
// where n_skip would exceed max_skip, and commits the last new best in front of it.  Everything behind the break point is
// discarded, which is exactly what the sequential loop never looked at.
// (*) the reference sets t[p[j']] = i only for j' that passed the admissibility tests (`continue` skips the store): the kernel
// does the same -- the store is predicated on `ok`.  Marks written by lanes behind the break point are harmless: a mark with
// value i is only ever compared with i during this anchor's own walk, by lanes even further behind.
//
// Two kernels.  chain_forward_lds_kernel (lists of <= kFastAnchors anchors whose reference coordinate fits 31 bits and bw <=
// kFastBw -- every list of the contig engine): lane l keeps anchor i-1-l (position, f, p) in registers and the window slides by one
// lane per anchor (wave_shr:1), so the 64 nearest predecessors -- where almost every walk ends, max_skip being 25 -- never come
// from memory; older predecessors, the marks and the gap-cost table (a function of |dr - dq| <= bw and the list's mean span,
// tabulated once per list: no double-precision instruction in the loop) are in LDS; the prefix maxima are v_max_i32_dpp row shifts
// + row broadcasts, and the skip counter is the reflected +-1 walk  D_l + max(n0, max_{u<=l} -D_u)  with D from v_mbcnt of two
// ballots -- one more prefix max instead of the pair scan.  One wave per block: its LDS operations complete in program order, so
// its lanes need compiler ordering only and no s_barrier.  The lists are read from and the results written to pinned host memory
// directly (one launch, no copy operations).  chain_forward_general_kernel (anything else): the textbook form of the same steps
// with 64-bit coordinates, f / p / marks in global memory behind agent-scope atomics, ds_bpermute shuffles, the pair scan.
// Bound: one wave per list is a dependent chain of ~130 instructions per anchor (~0.2 us): a launch lasts as long as its longest
// list (cfg2: ~250 lists of ~300 anchors per launch, longest ~1000-1500: 0.30 ms on average, profiles/r02_chain_gpu_ab.txt);
// algorithmic bytes 16 B in + 8 B out per anchor, irrelevant next to the latency.
//
// Bit-exactness: the gain is integer arithmetic plus two double-precision products that the reference also evaluates in double
// ((int)(dd * .01 * avg_qspan) and (int)((double)gap_cost * chain_gap_scale + .499)); the build runs with -ffp-contract=off, and
// v_mul_f64 / v_add_f64 / v_cvt are IEEE.  tests/test_chain_gpu.py compares f and p with the plain loop on the anchor lists of the
// alignment cases and on synthetic lists (both kernels; max_skip / max_chain_iter paths); tests/test_align_gpu.py compares whole
// alignments with the reference's minimap2.
#include "common.hpp"
#include "host_util.hpp"
#include "mm2.hpp"
#include <algorithm>
#include <cstring>

namespace nsgpu {
namespace {

struct ChainParams {
    int32_t max_dist, bw, max_skip, max_iter;
    float gap_scale;
};

constexpr int kNegInf = -(1 << 28);

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---- the general kernel: f / p / marks in global memory, read and written past the L1 with agent-scope atomics (one wave is the
// only reader and writer, the barrier orders its lanes) ----
__device__ __forceinline__ int32_t ld(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(64) void chain_forward_general_kernel(const mm2::Anchor *__restrict__ anchors, const uint64_t *__restrict__ off, const float *__restrict__ avg,
                                                                   const uint32_t *__restrict__ jobs, int32_t *__restrict__ f_out, int32_t *__restrict__ p_out,
                                                                   int32_t *__restrict__ t_glob, ChainParams P)
{
    const uint32_t job = jobs[blockIdx.x];
    const uint64_t base = off[job];
    const int32_t n = (int32_t)(off[job + 1] - base);
    const mm2::Anchor *a = anchors + base;
    int32_t *F = f_out + base, *Pp = p_out + base, *T = t_glob + base;
    const int lane = lane_id();
    for (int32_t i = lane; i < n; i += 64) st(T + i, 0);
    const double avg_qspan = (double)avg[job], gap_scale = (double)P.gap_scale;
    int32_t stt = 0;
    for (int32_t i = 0; i < n; ++i) {
        __syncthreads();                                   // f[i-1] / p[i-1] (and the marks of the first pass) are visible
        const uint64_t ri = a[i].x, yi = a[i].y;
        const int32_t qi = (int32_t)yi, q_span = (int32_t)(yi >> 32 & 0xff);
        while (stt < i && ri > a[stt].x + (uint64_t)P.max_dist) ++stt;
        if (i - stt > P.max_iter) stt = i - P.max_iter;
        int32_t max_f = q_span, max_j = -1, n_skip = 0;
        bool stop = false;
        for (int32_t hi = i - 1; hi >= stt && !stop; hi -= 64) {
            const int32_t j = hi - lane;
            const bool in = j >= stt;
            bool ok = false;
            int32_t sc = kNegInf, pj = -1;
            if (in) {
                const uint64_t xj = a[j].x, yj = a[j].y;
                const int64_t dr = (int64_t)(ri - xj);
                const int32_t dq = qi - (int32_t)yj;
                ok = dr != 0 && dq > 0 && dq <= P.max_dist;
                const int32_t dd = (int32_t)(dr > dq ? dr - dq : dq - dr);
                ok = ok && dd <= P.bw;
                if (ok) {
                    const int32_t min_d = dq < dr ? dq : (int32_t)dr;
                    sc = min_d > q_span ? q_span : min_d;
                    const int32_t log_dd = dd ? 31 - __builtin_clz((uint32_t)dd) : 0;
                    const int32_t gap_cost = (int)((double)dd * .01 * avg_qspan) + (log_dd >> 1);
                    sc -= (int)((double)gap_cost * gap_scale + .499);
                    sc += ld(F + j);
                    pj = ld(Pp + j);
                    if (pj >= 0) st(T + pj, i);
                }
            }
            __syncthreads();                               // this chunk's marks are visible
            const bool marked = ok && ld(T + j) == i;
            // exclusive prefix max of the scores in front of this lane, seeded with the running best
            int32_t pm = sc;
            for (int d = 1; d < 64; d <<= 1) { const int32_t y = __shfl_up(pm, d); if (lane >= d) pm = max(pm, y); }
            pm = __shfl_up(pm, 1);
            pm = lane ? max(pm, max_f) : max_f;
            const bool rec = ok && sc > pm;
            const bool skp = ok && !rec && marked;
            // n_skip behind this lane: inclusive scan of the maps n -> max(n + ca, cb)
            int32_t ca = rec ? -1 : (skp ? 1 : 0), cb = rec ? 0 : kNegInf;
            for (int d = 1; d < 64; d <<= 1) {
                const int32_t ya = __shfl_up(ca, d), yb = __shfl_up(cb, d);
                if (lane >= d) { cb = max(yb + ca, cb); ca += ya; }
            }
            const int32_t n_after = max(n_skip + ca, cb);
            const uint64_t brk = __ballot(skp && n_after > P.max_skip);
            const int first_brk = brk ? __ffsll((unsigned long long)brk) - 1 : 64;
            const uint64_t recs = __ballot(rec) & (first_brk >= 64 ? ~0ull : (1ull << first_brk) - 1);
            if (recs) {
                const int L = 63 - __clzll((long long)recs);
                max_f = __shfl(sc, L);
                max_j = hi - L;
            }
            n_skip = __shfl(n_after, 63);
            stop = brk != 0;
        }
        if (lane == 0) {
            st(F + i, max_f), st(Pp + i, max_j);
        }
    }
}


// ---- the LDS kernel ----
constexpr uint32_t kFastAnchors = 7400;         // 21 B of LDS per anchor + the gap-cost table: < 160 KB
constexpr int32_t kFastBw = 1023;               // the gap-cost table has bw + 1 entries
constexpr size_t lds_bytes(uint32_t n, int32_t bw) { return (((size_t)n * 21 + 15) & ~(size_t)15) + ((size_t)bw + 1) * 4; }

template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ int32_t dpp(int32_t ident, int32_t v)
{
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
}
constexpr int kWaveShr1 = 0x138;

// inclusive prefix max over the wave (lane order): v_max_i32_dpp with destination = both sources -- a lane without a source
// (row start, row outside the row mask; bound_ctrl off) is not written and keeps its value.  Inline assembly because the compiler
// emits mov + dpp-mov + max for the builtin; the s_nop covers the VALU-write -> DPP-read hazard, which it does not see in here.
#define NS_MAX_DPP(v, ctrl) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ int32_t scan_max(int32_t v)
{
    NS_MAX_DPP(v, "row_shr:1 row_mask:0xf");
    NS_MAX_DPP(v, "row_shr:2 row_mask:0xf");
    NS_MAX_DPP(v, "row_shr:4 row_mask:0xf");
    NS_MAX_DPP(v, "row_shr:8 row_mask:0xf");
    NS_MAX_DPP(v, "row_bcast:15 row_mask:0xa");
    NS_MAX_DPP(v, "row_bcast:31 row_mask:0xc");
    return v;
}
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ int32_t count_upto(uint64_t mask, bool self)      // set bits of `mask` in lanes <= this one (self = this lane's bit)
{
    return (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)) + (self ? 1 : 0);
}

// The running state of one anchor's walk over its predecessors, and one step of it over 64 predecessors (lane l: predecessor j, its
// admissibility `ok`, score `sc`, and whether an earlier admissible predecessor points at it).  The skip counter is a +-1 walk
// reflected at 0: after lane l it is D_l + max(n0, max over u <= l of -D_u) with D = (#marked non-best) - (#new best) up to the
// lane (Lindley's recursion solved; for +1 steps the reflection never binds) -- the special case of the pair scan in the header
// that needs one prefix max.
struct Walk { int32_t max_f, max_j, n_skip; bool stop; };
__device__ __forceinline__ void walk_chunk(Walk &w, int32_t hi, bool ok, int32_t sc, bool marked, int32_t max_skip)
{
    int32_t pm = scan_max(ok ? sc : kNegInf);
    pm = dpp<kWaveShr1>(kNegInf, pm);              // exclusive
    pm = max(pm, w.max_f);
    const bool rec = ok && sc > pm;
    const bool skp = ok && !rec && marked;
    const uint64_t recs_all = __ballot(rec), skps = __ballot(skp);
    const int32_t D = count_upto(skps, skp) - count_upto(recs_all, rec);
    const int32_t n_after = D + max(w.n_skip, scan_max(-D));
    const uint64_t brk = __ballot(skp && n_after > max_skip);
    const int first_brk = brk ? __ffsll((unsigned long long)brk) - 1 : 64;
    const uint64_t recs = recs_all & (first_brk >= 64 ? ~0ull : (1ull << first_brk) - 1);
    if (recs) {
        const int L = 63 - __clzll((long long)recs);
        w.max_f = __builtin_amdgcn_readlane(sc, L);
        w.max_j = hi - L;
    }
    w.n_skip = __builtin_amdgcn_readlane(n_after, 63);
    w.stop = brk != 0;
}

// list q: anchors[L.beg .. + L.n) in, f / p (and, with a_out, a copy of the anchors: lists that were seeded on the GPU reach the host
// this way) out at [L.obeg .. + L.n)
// With `seeded` (lists == jobs == nullptr) block b takes the list that seeds.hip left for pair b -- launched right behind the
// seeding kernel on its stream, no host round trip in between -- skips pairs that kernel flagged, and flags the ones that do not
// fit this launch's LDS (lds_anchors) itself.
__global__ __launch_bounds__(64) void chain_forward_lds_kernel(const mm2::Anchor *__restrict__ anchors, const ChainList *__restrict__ lists,
                                                               const uint32_t *__restrict__ jobs, int32_t *__restrict__ f_out, int32_t *__restrict__ p_out,
                                                               mm2::Anchor *__restrict__ a_out, SeedResult *seeded, uint32_t lds_anchors, ChainParams P,
                                                               int32_t *__restrict__ f_dev, int32_t *__restrict__ p_dev)
{
    extern __shared__ int32_t lds[];
    ChainList L;
    if (seeded) {
        const SeedResult r = seeded[blockIdx.x];
        if (r.flags || r.n == 0) return;
        if (r.n > lds_anchors) { if (threadIdx.x == 0) seeded[blockIdx.x].flags = SEED_FLAG_MANY; return; }
        L = ChainList{r.base, r.base, r.n, r.avg};
    } else L = lists[jobs[blockIdx.x]];
    const uint64_t base = L.obeg;
    const int32_t n = (int32_t)L.n;
    const mm2::Anchor *a = anchors + L.beg;
    int32_t *F = lds, *Pp = lds + n, *T = lds + 2 * n, *R = lds + 3 * n, *Q = lds + 4 * n;
    uint8_t *S = reinterpret_cast<uint8_t *>(lds + 5 * n);
    const int lane = lane_id();
    // (the list may lie in pinned host memory: four loads in flight per lane)
    for (int32_t i0 = 0; i0 < n; i0 += 256) {
        uint64_t x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int32_t i = i0 + u * 64 + lane; if (i < n) x[u] = a[i].x, y[u] = a[i].y; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t i = i0 + u * 64 + lane;
            if (i < n) {
                T[i] = 0, R[i] = (int32_t)x[u], Q[i] = (int32_t)y[u], S[i] = (uint8_t)(y[u] >> 32);      // (lists with x >= 2^31 go to the general kernel)
                if (a_out) a_out[base + i] = mm2::Anchor{x[u], y[u]};
            }
        }
    }
    lds_order();
    // what a gap of dd = |dr - dq| <= bw costs (chain.c:68-71): depends on dd and the list's mean span only -- tabulated once, so that
    // no double-precision instruction is left in the per-anchor loop
    int32_t *G = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(lds) + (((size_t)n * 21 + 15) & ~(size_t)15));
    {
        const double avg_qspan = (double)L.avg, gap_scale = (double)P.gap_scale;
        for (int32_t dd = lane; dd <= P.bw; dd += 64) {
            const int32_t log_dd = dd ? 31 - __builtin_clz((uint32_t)dd) : 0;
            const int32_t gap_cost = (int)((double)dd * .01 * avg_qspan) + (log_dd >> 1);
            G[dd] = (int)((double)gap_cost * gap_scale + .499);
        }
    }
    lds_order();
    // admissibility and gain of predecessor (rj, qj) for the anchor (ri, qi, q_span): chain.c:58-75 without f[j]
    auto gain = [&](int32_t ri, int32_t qi, int32_t q_span, int32_t rj, int32_t qj, bool in, int32_t &sc) -> bool {
        const int32_t dr = ri - rj, dq = qi - qj;
        const int32_t dd = dr > dq ? dr - dq : dq - dr;
        const bool ok = in && dr != 0 && dq > 0 && dq <= P.max_dist && dd <= P.bw;
        sc = min(min(dq, dr), q_span) - G[ok ? dd : 0];
        return ok;
    };
    // Lane l keeps anchor i-1-l -- position, f, p -- in registers: the 64 nearest predecessors of anchor i, which is where the walk
    // of almost every anchor ends (max_skip = 25 marked predecessors), never come from LDS.  After anchor i the window slides by one
    // lane (wave_shr:1) and lane 0 takes anchor i itself.  Only the marks go through LDS.
    int32_t rj = 0, qj = 0, fj = 0, pj = -1;
    int32_t ri = n ? R[0] : 0, qi = n ? Q[0] : 0, q_span = n ? S[0] : 0;
    for (int32_t i = 0; i < n; ++i) {
        // anchor i+1's own values arrive while anchor i is worked on
        const int32_t i1 = i + 1 < n ? i + 1 : i;
        const int32_t ri_n = R[i1], qi_n = Q[i1], sp_n = S[i1];
        Walk w{q_span, -1, 0, false};
        // the window: predecessors within max_dist on the reference (the list is sorted by it) and at most max_chain_iter back --
        // what chain.c's `st` pointer amounts to, as a test each lane can make on its own
        const int32_t j_min = i - P.max_iter > 0 ? i - P.max_iter : 0;
        {
            const int32_t j = i - 1 - lane;
            const bool in = j >= j_min && ri - rj <= P.max_dist;          // (both below 2^31)
            int32_t sc;
            const bool ok = gain(ri, qi, q_span, rj, qj, in, sc);
            sc += fj;
            if (ok && pj >= 0) T[pj] = i;
            lds_order();                                   // this chunk's marks are visible to the lanes behind
            const bool marked = ok && T[j] == i;
            const bool more = __builtin_amdgcn_readlane((int)in, 63) != 0;     // the window goes on behind lane 63
            walk_chunk(w, i - 1, ok, sc, marked, P.max_skip);
            if (!more) w.stop = true;
        }
        for (int32_t hi = i - 65; hi >= j_min && !w.stop; hi -= 64) {
            const int32_t j = hi - lane;
            const bool in0 = j >= j_min;
            const int32_t rj2 = in0 ? R[j] : 0, qj2 = in0 ? Q[j] : 0, fj2 = in0 ? F[j] : 0, pj2 = in0 ? Pp[j] : -1;
            const bool in = in0 && ri - rj2 <= P.max_dist;
            int32_t sc;
            const bool ok = gain(ri, qi, q_span, rj2, qj2, in, sc);
            sc += fj2;
            if (ok && pj2 >= 0) T[pj2] = i;
            lds_order();
            const bool marked = ok && T[j] == i;
            const bool more = __builtin_amdgcn_readlane((int)in, 63) != 0;
            walk_chunk(w, hi, ok, sc, marked, P.max_skip);
            if (!more) w.stop = true;
        }
        if (lane == 0) F[i] = w.max_f, Pp[i] = w.max_j;
        lds_order();
        rj = dpp<kWaveShr1>(ri, rj), qj = dpp<kWaveShr1>(qi, qj), fj = dpp<kWaveShr1>(w.max_f, fj), pj = dpp<kWaveShr1>(w.max_j, pj);
        ri = ri_n, qi = qi_n, q_span = sp_n;
    }
    for (int32_t i = lane; i < n; i += 64) f_out[base + i] = F[i], p_out[base + i] = Pp[i];
    // a copy in device memory for the plan kernel that follows on the stream (plan.hip); f_out / p_out are pinned host memory
    if (f_dev) for (int32_t i = lane; i < n; i += 64) f_dev[base + i] = F[i], p_dev[base + i] = Pp[i];
}

// ---- the ring kernel: lists of ANY length at the LDS kernel's pace (round 3) ----
// A read across a tandem repeat has 10^4 - 10^5 anchors ((repeat length / unit)^2 of them); such lists used to go to the general kernel,
// whose f / p / marks live in global memory behind L2-latency atomics (~12 us per anchor when the walk runs its full max_chain_iter
// predecessors, as it does in a dense repeat).  But an anchor only ever looks max_chain_iter back (and marks predecessors' predecessors,
// at most twice that): f, p, the marks and the coordinates live in rings of kRing entries here, anchors stream in 64 at a time, f / p stream
// out 64 at a time.  Same steps as chain_forward_lds_kernel otherwise (one wave per list, the 64 nearest predecessors in registers).
// Needs max_chain_iter <= kRingIter, bw <= kFastBw, coordinates < 2^31.  A stale mark in a re-used ring slot is an anchor index from at
// least kRing anchors ago: it never equals the current one.
constexpr int32_t kRing = 1024, kRingMask = kRing - 1, kRingIter = 416;       // 2 * kRingIter + 64 + 128 <= kRing
constexpr size_t ring_lds_bytes(int32_t bw) { return (size_t)kRing * 21 + 16 + ((size_t)bw + 1) * 4; }

__global__ __launch_bounds__(64) void chain_forward_ring_kernel(const mm2::Anchor *__restrict__ anchors, const ChainList *__restrict__ lists, const uint32_t *__restrict__ jobs,
                                                                int32_t *__restrict__ f_out, int32_t *__restrict__ p_out, ChainParams P)
{
    extern __shared__ int32_t lds[];
    const ChainList L = lists[jobs[blockIdx.x]];
    const uint64_t base = L.obeg;
    const int32_t n = (int32_t)L.n;
    const mm2::Anchor *a = anchors + L.beg;
    int32_t *F = lds, *Pp = lds + kRing, *T = lds + 2 * kRing, *R = lds + 3 * kRing, *Q = lds + 4 * kRing;
    uint8_t *S = reinterpret_cast<uint8_t *>(lds + 5 * kRing);
    int32_t *G = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(lds) + (((size_t)kRing * 21 + 15) & ~(size_t)15));
    const int lane = lane_id();
    for (int32_t i = lane; i < kRing; i += 64) T[i] = -1;
    // anchors [0, 128) into the rings; block [i + 64, i + 128) follows while anchors i .. i + 63 are worked on
    for (int32_t i0 = 0; i0 < 128; i0 += 64) {
        const int32_t i = i0 + lane;
        if (i < n) { const uint64_t x = a[i].x, y = a[i].y; R[i] = (int32_t)x, Q[i] = (int32_t)y, S[i] = (uint8_t)(y >> 32); }
    }
    {
        const double avg_qspan = (double)L.avg, gap_scale = (double)P.gap_scale;
        for (int32_t dd = lane; dd <= P.bw; dd += 64) {
            const int32_t log_dd = dd ? 31 - __builtin_clz((uint32_t)dd) : 0;
            const int32_t gap_cost = (int)((double)dd * .01 * avg_qspan) + (log_dd >> 1);
            G[dd] = (int)((double)gap_cost * gap_scale + .499);
        }
    }
    lds_order();
    auto gain = [&](int32_t ri, int32_t qi, int32_t q_span, int32_t rj, int32_t qj, bool in, int32_t &sc) -> bool {
        const int32_t dr = ri - rj, dq = qi - qj;
        const int32_t dd = dr > dq ? dr - dq : dq - dr;
        const bool ok = in && dr != 0 && dq > 0 && dq <= P.max_dist && dd <= P.bw;
        sc = min(min(dq, dr), q_span) - G[ok ? dd : 0];
        return ok;
    };
    int32_t rj = 0, qj = 0, fj = 0, pj = -1;
    int32_t ri = n ? R[0] : 0, qi = n ? Q[0] : 0, q_span = n ? S[0] : 0;
    uint64_t pf_x = 0, pf_y = 0;                       // the block in flight from memory
    for (int32_t i = 0; i < n; ++i) {
        if ((i & 63) == 0) { const int32_t k = i + 64 + lane; if (i > 0 && k < n) pf_x = a[k].x, pf_y = a[k].y; }
        if ((i & 63) == 32) { const int32_t k = (i & ~63) + 64 + lane; if (i > 63 && k < n) { R[k & kRingMask] = (int32_t)pf_x, Q[k & kRingMask] = (int32_t)pf_y, S[k & kRingMask] = (uint8_t)(pf_y >> 32); } }
        const int32_t i1 = i + 1 < n ? i + 1 : i;
        const int32_t ri_n = R[i1 & kRingMask], qi_n = Q[i1 & kRingMask], sp_n = S[i1 & kRingMask];
        Walk w{q_span, -1, 0, false};
        const int32_t j_min = i - P.max_iter > 0 ? i - P.max_iter : 0;
        {
            const int32_t j = i - 1 - lane;
            const bool in = j >= j_min && ri - rj <= P.max_dist;
            int32_t sc;
            const bool ok = gain(ri, qi, q_span, rj, qj, in, sc);
            sc += fj;
            if (ok && pj >= 0) T[pj & kRingMask] = i;
            lds_order();
            const bool marked = ok && T[j & kRingMask] == i;
            const bool more = __builtin_amdgcn_readlane((int)in, 63) != 0;
            walk_chunk(w, i - 1, ok, sc, marked, P.max_skip);
            if (!more) w.stop = true;
        }
        for (int32_t hi = i - 65; hi >= j_min && !w.stop; hi -= 64) {
            const int32_t j = hi - lane;
            const bool in0 = j >= j_min;
            const int32_t jm = j & kRingMask;
            const int32_t rj2 = in0 ? R[jm] : 0, qj2 = in0 ? Q[jm] : 0, fj2 = in0 ? F[jm] : 0, pj2 = in0 ? Pp[jm] : -1;
            const bool in = in0 && ri - rj2 <= P.max_dist;
            int32_t sc;
            const bool ok = gain(ri, qi, q_span, rj2, qj2, in, sc);
            sc += fj2;
            if (ok && pj2 >= 0) T[pj2 & kRingMask] = i;
            lds_order();
            const bool marked = ok && T[jm] == i;
            const bool more = __builtin_amdgcn_readlane((int)in, 63) != 0;
            walk_chunk(w, hi, ok, sc, marked, P.max_skip);
            if (!more) w.stop = true;
        }
        if (lane == 0) F[i & kRingMask] = w.max_f, Pp[i & kRingMask] = w.max_j;
        lds_order();
        if ((i & 63) == 63 || i == n - 1) {            // the block of results that is complete goes out
            const int32_t k = (i & ~63) + lane;
            if (k <= i) f_out[base + k] = F[k & kRingMask], p_out[base + k] = Pp[k & kRingMask];
        }
        rj = dpp<kWaveShr1>(ri, rj), qj = dpp<kWaveShr1>(qi, qj), fj = dpp<kWaveShr1>(w.max_f, fj), pj = dpp<kWaveShr1>(w.max_j, pj);
        ri = ri_n, qi = qi_n, q_span = sp_n;
    }
}

// ---- the level kernel: the long lists of repeats on a whole workgroup (round 4) ----
// A read across a tandem repeat has 10^4 - 10^5 anchors because every reference minimizer of the repeat is hit from every copy in the read:
// tens to hundreds of anchors SHARE a reference position, and anchors with dr == 0 are never each other's predecessors (chain.c:58).  The
// anchors of one reference position -- a level -- therefore depend only on the levels before them: up to kLvWaves of them are walked at
// the same time, one wave each, over the shared rings of f / p / coordinates; marks are per wave (a mark is "anchor i saw this
// predecessor", and the waves work on different i).  One workgroup barrier per batch of a level.  Same walk, same results as the ring kernel;
// a list without repeated positions degenerates to one anchor per barrier, so only long lists are sent here (gpu_chain_launch).
constexpr int kLvWaves = 16;
constexpr int32_t kLvAhead = 272;              // anchors kept loaded beyond the batch's first one
constexpr size_t level_lds_bytes(int32_t bw) { return (size_t)kRing * 17 + (size_t)kLvWaves * kRing * 4 + 16 + ((size_t)bw + 1) * 4; }

__global__ __launch_bounds__(kLvWaves * 64) void chain_forward_level_kernel(const mm2::Anchor *__restrict__ anchors, const ChainList *__restrict__ lists, const uint32_t *__restrict__ jobs,
                                                                             int32_t *__restrict__ f_out, int32_t *__restrict__ p_out, ChainParams P)
{
    extern __shared__ int32_t lds[];
    const ChainList L = lists[jobs[blockIdx.x]];
    const uint64_t base = L.obeg;
    const int32_t n = (int32_t)L.n;
    const mm2::Anchor *a = anchors + L.beg;
    int32_t *F = lds, *Pp = lds + kRing, *R = lds + 2 * kRing, *Q = lds + 3 * kRing;
    uint8_t *S = reinterpret_cast<uint8_t *>(lds + 4 * kRing);
    int32_t *T_all = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(lds) + (size_t)kRing * 17 + 16 - ((size_t)kRing * 17) % 16);
    int32_t *G = T_all + (size_t)kLvWaves * kRing;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t *T = T_all + (size_t)wv * kRing;
    for (int32_t i = lane; i < kRing; i += 64) T[i] = -1;
    {
        const double avg_qspan = (double)L.avg, gap_scale = (double)P.gap_scale;
        for (int32_t dd = tid; dd <= P.bw; dd += kLvWaves * 64) {
            const int32_t log_dd = dd ? 31 - __builtin_clz((uint32_t)dd) : 0;
            const int32_t gap_cost = (int)((double)dd * .01 * avg_qspan) + (log_dd >> 1);
            G[dd] = (int)((double)gap_cost * gap_scale + .499);
        }
    }
    auto gain = [&](int32_t ri, int32_t qi, int32_t q_span, int32_t rj, int32_t qj, bool in, int32_t &sc) -> bool {
        const int32_t dr = ri - rj, dq = qi - qj;
        const int32_t dd = dr > dq ? dr - dq : dq - dr;
        const bool ok = in && dr != 0 && dq > 0 && dq <= P.max_dist && dd <= P.bw;
        sc = min(min(dq, dr), q_span) - G[ok ? dd : 0];
        return ok;
    };
    int32_t loaded = 0;                 // anchors [0, loaded) have been through the rings
    int32_t i0 = 0;                     // first anchor of the batch (all of this is uniform over the workgroup)
    while (i0 < n) {
        // 256 more anchors into the rings (their slots held anchors more than max_chain_iter behind the batch: loaded - 768 < i0 - 416)
        while (loaded < n && loaded < i0 + kLvAhead) {
            const int32_t k = loaded + tid;
            if (tid < 256 && k < n) { const uint64_t x = a[k].x, y = a[k].y; R[k & kRingMask] = (int32_t)x, Q[k & kRingMask] = (int32_t)y, S[k & kRingMask] = (uint8_t)(y >> 32); }
            loaded = loaded + 256 < n ? loaded + 256 : n;
            __syncthreads();
        }
        // the batch: the anchors from i0 on that share its reference position, at most one per wave
        int32_t cnt;
        {
            const int32_t k = i0 + lane;
            const bool same = lane < kLvWaves && k < n && R[k & kRingMask] == R[i0 & kRingMask];
            const uint64_t m = __ballot(same);
            cnt = (int32_t)__builtin_ctzll(~m);
        }
        if (wv < cnt) {
            const int32_t i = i0 + wv;
            const int32_t ri = R[i & kRingMask], qi = Q[i & kRingMask], q_span = S[i & kRingMask];
            Walk w{q_span, -1, 0, false};
            const int32_t j_min = i - P.max_iter > 0 ? i - P.max_iter : 0;
            // The whole window at once: the loads of all its chunks of 64 predecessors are issued together, then all gap costs, then all
            // marks, then all mark tests -- three LDS round trips per ANCHOR instead of three per chunk (a chunk's walk is a dependent chain
            // of ~1000 cycles otherwise, nearly all of it LDS latency); what is left per chunk is the scan of walk_chunk.  Chunks behind the
            // one where the walk stops were loaded and marked in vain: a mark only ever matters to predecessors further back, which the
            // stopped walk does not visit either.
            constexpr int kCh = (kRingIter + 63) / 64;
            const int32_t nch = (i - j_min + 63) >> 6;                 // <= kCh (max_chain_iter <= kRingIter)
            int32_t sc[kCh], tm[kCh];
            bool ok[kCh], in[kCh];
            {
                int32_t rj[kCh], qj[kCh], fj[kCh], pj[kCh];
#pragma unroll
                for (int c = 0; c < kCh; ++c) {
                    const int32_t j = i - 1 - 64 * c - lane;
                    const bool in0 = c < nch && j >= j_min;
                    const int32_t jm = j & kRingMask;
                    // (a predecessor inside the batch is being written by its wave: it has dr == 0, its f / p are read and never used)
                    rj[c] = in0 ? R[jm] : 0, qj[c] = in0 ? Q[jm] : 0, fj[c] = in0 ? F[jm] : 0, pj[c] = in0 ? Pp[jm] : -1;
                    in[c] = in0;
                }
#pragma unroll
                for (int c = 0; c < kCh; ++c) {
                    in[c] = in[c] && ri - rj[c] <= P.max_dist;
                    ok[c] = gain(ri, qi, q_span, rj[c], qj[c], in[c], sc[c]);
                    sc[c] += fj[c];
                }
#pragma unroll
                for (int c = 0; c < kCh; ++c) if (ok[c] && pj[c] >= 0) T[pj[c] & kRingMask] = i;
            }
            lds_order();
#pragma unroll
            for (int c = 0; c < kCh; ++c) tm[c] = ok[c] ? T[(i - 1 - 64 * c - lane) & kRingMask] : -1;
#pragma unroll
            for (int c = 0; c < kCh; ++c) {
                if (c < nch && !w.stop) {
                    const bool more = __builtin_amdgcn_readlane((int)in[c], 63) != 0;
                    walk_chunk(w, i - 1 - 64 * c, ok[c], sc[c], ok[c] && tm[c] == i, P.max_skip);
                    if (!more) w.stop = true;
                }
            }
            if (lane == 0) F[i & kRingMask] = w.max_f, Pp[i & kRingMask] = w.max_j;
        }
        __syncthreads();
        const int32_t i1 = i0 + cnt;
        // the block of 64 results that the batch completed goes out (the last wave is the one most often without an anchor)
        if (wv == kLvWaves - 1 && ((i1 >> 6) != (i0 >> 6) || i1 == n)) {
            const int32_t b0 = (i1 >> 6) != (i0 >> 6) ? (i0 & ~63) : (i1 & ~63);
            const int32_t k = b0 + lane;
            if (k < i1) f_out[base + k] = F[k & kRingMask], p_out[base + k] = Pp[k & kRingMask];
            if ((i1 >> 6) != (i0 >> 6) && i1 == n && (i1 & 63)) {       // the batch crossed a boundary AND ended the list: the tail block as well
                const int32_t k2 = (i1 & ~63) + lane;
                if (k2 < i1) f_out[base + k2] = F[k2 & kRingMask], p_out[base + k2] = Pp[k2 & kRingMask];
            }
        }
        i0 = i1;
    }
}

}  // namespace

// f[i] / p[i] of mm_chain_dp's first loop for the anchor lists a[off[q] .. off[q+1]) (host pointers; avg[q] = mean query span):
// gpu_chain_launch stages the lists and starts the kernel(s) on the workspace's stream, gpu_chain_wait waits for them.  The results
// land in W.h_out: f at [0, total), p at [total, 2 total), valid until the workspace's next launch.
int gpu_chain_launch(nsgpu_ctx *c, int ws, const mm2::Opt &opt, const std::vector<const mm2::Anchor *> &lists, const std::vector<uint64_t> &off,
                     const std::vector<float> &avg)
{
    nsgpu_ctx::ChainWs &W = c->cws[ws];
    const size_t nq = lists.size();
    const uint64_t total = off[nq];
    W.pend_total = total;
    if (total == 0) return NSGPU_OK;
    const double t0 = now_ms();
    if (!W.stream) NS_TRY(role_stream_create(&W.stream, "seeds"));
    // pinned staging: anchors | offsets | mean spans | job lists (LDS kernel's first) | list descriptors of the LDS kernel
    const size_t b_anch = total * sizeof(mm2::Anchor), b_off = (nq + 1) * sizeof(uint64_t), b_avg = (nq * sizeof(float) + 7) & ~(size_t)7, b_jobs = (nq * sizeof(uint32_t) + 7) & ~(size_t)7;
    NS_TRY(W.h_in.reserve(b_anch + b_off + b_avg + b_jobs + nq * sizeof(ChainList)));
    NS_TRY(W.h_out.reserve(total * 2 * sizeof(int32_t)));
    uint8_t *h = W.h_in.as<uint8_t>();
    mm2::Anchor *ha = reinterpret_cast<mm2::Anchor *>(h);
    uint64_t *ho = reinterpret_cast<uint64_t *>(h + b_anch);
    float *hv = reinterpret_cast<float *>(h + b_anch + b_off);
    uint32_t *hj = reinterpret_cast<uint32_t *>(h + b_anch + b_off + b_avg);
    ChainList *hl = reinterpret_cast<ChainList *>(h + b_anch + b_off + b_avg + b_jobs);
    memcpy(ho, off.data(), b_off);
    memcpy(hv, avg.data(), nq * sizeof(float));
    for (size_t q = 0; q < nq; ++q) hl[q] = ChainList{off[q], off[q], (uint32_t)(off[q + 1] - off[q]), avg[q]};
    // which kernel takes which list; the longest lists first (a wave's time grows with its list: the tail of the launch should be
    // the short ones)
    uint32_t n_lds = 0, n_big = 0, n_ring = 0, n_level = 0, max_lds = 0;
    std::vector<uint8_t> &fast = W.h_fast;                // 1: the LDS kernel, 2: the ring kernel (same conditions, any length), 3: the level kernel (long lists), 0: the general kernel
    fast.assign(nq, 0);
    // lists from this length on go to the level kernel (a workgroup per list; 0 = never): such lists come from repeats
    static const uint64_t level_min = [] { const char *e = getenv("NSGPU_CHAIN_LEVEL_MIN"); return e ? (uint64_t)atoll(e) : (uint64_t)768; }();
    par_for(nq, [&](size_t q) {
        const uint64_t n = off[q + 1] - off[q];
        if (n == 0) return;
        memcpy(ha + off[q], lists[q], (size_t)n * sizeof(mm2::Anchor));
        uint64_t hi_bits = 0;
        for (uint64_t i = 0; i < n; ++i) hi_bits |= lists[q][i].x;
        const bool small_coords = (hi_bits >> 31) == 0 && opt.bw >= 0 && opt.bw <= kFastBw;
        const bool ring_ok = small_coords && opt.max_chain_iter <= kRingIter && opt.max_chain_iter >= 0;
        fast[q] = ring_ok && level_min && n >= level_min ? 3 : small_coords && n <= kFastAnchors ? 1 : ring_ok ? 2 : 0;
    });
    for (size_t q = 0; q < nq; ++q) if (fast[q] == 1) hj[n_lds++] = (uint32_t)q, max_lds = std::max<uint32_t>(max_lds, (uint32_t)(off[q + 1] - off[q]));
    for (size_t q = 0; q < nq; ++q) if (!fast[q] && off[q + 1] > off[q]) hj[n_lds + n_big++] = (uint32_t)q;
    for (size_t q = 0; q < nq; ++q) if (fast[q] == 2) hj[n_lds + n_big + n_ring++] = (uint32_t)q;
    for (size_t q = 0; q < nq; ++q) if (fast[q] == 3) hj[n_lds + n_big + n_ring + n_level++] = (uint32_t)q;
    std::sort(hj + n_lds + n_big + n_ring, hj + n_lds + n_big + n_ring + n_level, [&](uint32_t x, uint32_t y) { const uint64_t nx = off[x + 1] - off[x], ny = off[y + 1] - off[y]; return nx != ny ? nx > ny : x < y; });
    std::sort(hj + n_lds + n_big, hj + n_lds + n_big + n_ring, [&](uint32_t x, uint32_t y) { const uint64_t nx = off[x + 1] - off[x], ny = off[y + 1] - off[y]; return nx != ny ? nx > ny : x < y; });
    std::sort(hj, hj + n_lds, [&](uint32_t x, uint32_t y) { const uint64_t nx = off[x + 1] - off[x], ny = off[y + 1] - off[y]; return nx != ny ? nx > ny : x < y; });
    const double t1 = now_ms();
    // The LDS kernel reads its lists straight from the pinned staging buffer and stores f / p straight into the pinned result buffer
    // (each anchor crosses PCIe once each way, as a coalesced read / a fire-and-forget write): one launch and one wait, no copy
    // operations whose scheduling behind the DP streams' traffic cost more than the kernel.  Only the general kernel, which reads
    // anchors repeatedly, works on device copies.
    const ChainParams P{opt.max_gap, opt.bw, opt.max_chain_skip, opt.max_chain_iter, opt.chain_gap_scale};
    int32_t *hf = W.h_out.as<int32_t>(), *hp = hf + total;
    if (n_level) {      // a workgroup of sixteen waves per long list (chain_forward_level_kernel), beside the other launches
        static LdsAttr level_attr;
        NS_TRY(level_attr.raise(level_lds_bytes(kFastBw), reinterpret_cast<const void *>(chain_forward_level_kernel)));
        if (!W.stream2) NS_TRY(role_stream_create(&W.stream2, "seeds"));
        hipLaunchKernelGGL(chain_forward_level_kernel, dim3(n_level), dim3(kLvWaves * 64), level_lds_bytes(opt.bw), W.stream2, ha, hl, hj + n_lds + n_big + n_ring, hf, hp, P);
        NS_HIP(hipGetLastError());
        W.ring_used = true;
    }
    if (n_ring) {       // the longest lists first; like the LDS kernel it reads the pinned staging buffer and writes the pinned results itself
        static LdsAttr ring_attr;
        NS_TRY(ring_attr.raise(ring_lds_bytes(kFastBw), reinterpret_cast<const void *>(chain_forward_ring_kernel)));
        if (!W.stream2) NS_TRY(role_stream_create(&W.stream2, "seeds"));
        hipLaunchKernelGGL(chain_forward_ring_kernel, dim3(n_ring), dim3(64), ring_lds_bytes(opt.bw), W.stream2, ha, hl, hj + n_lds + n_big, hf, hp, P);
        NS_HIP(hipGetLastError());
        W.ring_used = true;
    }
    if (n_lds) {
        if (!W.lds_set) {
            NS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_forward_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(kFastAnchors, kFastBw)));
            W.lds_set = lds_bytes(kFastAnchors, kFastBw);
        }
        hipLaunchKernelGGL(chain_forward_lds_kernel, dim3(n_lds), dim3(64), lds_bytes(max_lds, opt.bw), W.stream, ha, hl, hj, hf, hp, (mm2::Anchor *)nullptr, (SeedResult *)nullptr, 0u, P,
                           (int32_t *)nullptr, (int32_t *)nullptr);
    }
    if (n_big) {
        NS_TRY(W.d_in.reserve(b_anch + b_off + b_avg + b_jobs));
        NS_TRY(W.d_out.reserve(total * 2 * sizeof(int32_t)));
        NS_TRY(W.d_marks.reserve(total * sizeof(int32_t)));
        uint8_t *d = W.d_in.as<uint8_t>();
        NS_HIP(hipMemcpyAsync(d, h, b_anch + b_off + b_avg + b_jobs, hipMemcpyHostToDevice, W.stream));
        int32_t *df = W.d_out.as<int32_t>(), *dp = df + total;
        hipLaunchKernelGGL(chain_forward_general_kernel, dim3(n_big), dim3(64), 0, W.stream, reinterpret_cast<const mm2::Anchor *>(d), reinterpret_cast<const uint64_t *>(d + b_anch),
                           reinterpret_cast<const float *>(d + b_anch + b_off), reinterpret_cast<const uint32_t *>(d + b_anch + b_off + b_avg) + n_lds, df, dp, W.d_marks.as<int32_t>(), P);
        for (uint32_t k = 0; k < n_big; ++k) {
            const uint32_t q = hj[n_lds + k];
            const size_t o = (size_t)off[q], nb = (size_t)(off[q + 1] - off[q]) * sizeof(int32_t);
            NS_HIP(hipMemcpyAsync(hf + o, df + o, nb, hipMemcpyDeviceToHost, W.stream));
            NS_HIP(hipMemcpyAsync(hp + o, dp + o, nb, hipMemcpyDeviceToHost, W.stream));
        }
    }
    NS_HIP(hipGetLastError());
    W.ms_stage += t1 - t0, W.ms_enqueue += now_ms() - t1, ++W.calls;
    return NSGPU_OK;
}

// The same for the lists seeds.hip has just been asked for, on the seeding workspace's stream right behind its kernel: res / d_anchors
// are that launch's (device-visible) results, capacity its anchor capacity, max_n_qry the longest query minimizer list (the LDS of
// the launch is sized for about twice as many anchors; a list beyond that is flagged and goes the host code's way).  The kernel
// also copies the anchors out: after the stream has been waited for, anchors / f / p of pair q are at a + res[q].base etc. (pinned).
int gpu_chain_launch_seeded(nsgpu_ctx *c, int ws, hipStream_t stream, const mm2::Opt &opt, const mm2::Anchor *d_anchors, SeedResult *res, size_t n_pairs,
                            uint64_t capacity, uint32_t max_n_qry)
{
    nsgpu_ctx::ChainWs &W = c->cws[ws];
    NS_CHECK(opt.bw >= 0 && opt.bw <= kFastBw, NSGPU_ERR_ARG, "chaining of GPU-seeded lists: bw above %d", kFastBw);
    W.pend_total = n_pairs ? capacity : 0;
    if (n_pairs == 0) return NSGPU_OK;
    const double t0 = now_ms();
    NS_TRY(W.h_out.reserve(capacity * (sizeof(mm2::Anchor) + 2 * sizeof(int32_t))));
    const ChainParams P{opt.max_gap, opt.bw, opt.max_chain_skip, opt.max_chain_iter, opt.chain_gap_scale};
    mm2::Anchor *ha = W.h_out.as<mm2::Anchor>();
    int32_t *hf = reinterpret_cast<int32_t *>(ha + capacity), *hp = hf + capacity;
    if (!W.lds_set) {
        NS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_forward_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(kFastAnchors, kFastBw)));
        W.lds_set = lds_bytes(kFastAnchors, kFastBw);
    }
    const uint32_t lds_anchors = (uint32_t)std::min<uint64_t>(kFastAnchors, 2 * (uint64_t)max_n_qry + 256);
    NS_TRY(W.d_out.reserve(capacity * 2 * sizeof(int32_t)));
    int32_t *df = W.d_out.as<int32_t>(), *dp = df + capacity;
    W.seeded_lds_anchors = lds_anchors, W.seeded_capacity = capacity;
    hipLaunchKernelGGL(chain_forward_lds_kernel, dim3((unsigned)n_pairs), dim3(64), lds_bytes(lds_anchors, opt.bw), stream, d_anchors, (const ChainList *)nullptr,
                       (const uint32_t *)nullptr, hf, hp, ha, res, lds_anchors, P, df, dp);
    NS_HIP(hipGetLastError());
    W.ms_enqueue += now_ms() - t0, ++W.calls;
    return NSGPU_OK;
}

// (after the stream of the launch above has been waited for)
void gpu_chain_results_seeded(nsgpu_ctx *c, int ws, const mm2::Anchor *&a, const int32_t *&f, const int32_t *&p)
{
    nsgpu_ctx::ChainWs &W = c->cws[ws];
    a = nullptr, f = p = nullptr;
    if (W.pend_total == 0) return;
    a = W.h_out.as<mm2::Anchor>();
    f = reinterpret_cast<const int32_t *>(a + W.pend_total), p = f + W.pend_total;
}

SeedChainDev gpu_seeds_chain_dev(nsgpu_ctx *c, int ws, int chain_ws)
{
    nsgpu_ctx::SeedWs &S = c->seed_ws[ws];
    nsgpu_ctx::ChainWs &W = c->cws[chain_ws];
    SeedChainDev d;
    d.anchors = S.d_out.as<mm2::Anchor>(), d.res = S.h_res.as<SeedResult>();
    d.f = W.d_out.as<int32_t>(), d.p = d.f + W.seeded_capacity;
    d.stream = S.stream, d.lds_anchors = W.seeded_lds_anchors;
    return d;
}

int gpu_chain_wait(nsgpu_ctx *c, int ws, const int32_t *&f, const int32_t *&p)
{
    nsgpu_ctx::ChainWs &W = c->cws[ws];
    f = p = nullptr;
    if (W.pend_total == 0) return NSGPU_OK;
    const double t0 = now_ms();
    NS_HIP(stream_wait_short(W.stream));
    if (W.ring_used) { NS_HIP(stream_wait(W.stream2)); W.ring_used = false; }
    W.ms_wait += now_ms() - t0;
    f = W.h_out.as<int32_t>(), p = f + W.pend_total;
    return NSGPU_OK;
}

}  // namespace nsgpu
