// ---------------------------------------------------------------------------------------------------------------------
// Third-party notice.  The decisions in this file are those of minimap2 v2.17 (https://github.com/lh3/minimap2; files cited per
// step) -- bit-exact thresholds and orders are part of the contract of NanoSpring's on-disk format.  minimap2 is
//   Copyright (c) 2018- Dana-Farber Cancer Institute, 2017-2018 Broad Institute, Inc.
// and distributed under the MIT License; its full text is in THIRD_PARTY_NOTICES.md at the root of this repository.
// ---------------------------------------------------------------------------------------------------------------------
// plan.hip -- the alignment plan on the DEVICE (second half: an alignment's dependency chain stays on the GPU).
//
// Between the chaining kernel (chain.hip: f[] / p[] of mm_chain_dp's forward pass) and the DP launch (ksw2_reg.hip) minimap2 decides,
// per read: the chains (chain.c:94-164), the regions (hit.c:52-88, 125-186, 255-272, 315-371) and, per region, which banded DP problems
// the alignment skeleton needs (align.c:565-795: left extension, one gap fill per >= min_ksw_len stretch between kept anchors, right
// extension).  The host code does this in mm2.cpp; every slot of the contig stage used to wait for the chaining kernel, run that code,
// pack the DP sequences, upload them and launch.  Here ONE WORKGROUP PER ALIGNMENT does it behind the chaining kernel on its stream, writes
// the DP task descriptors and their sequences (gathered from the contig's consensus, resident in HBM and updated in place --
// consensus_driver.hip cons_update_kernel -- and from the candidate read where the sketch batch staged it) and appends every task to the
// launch list of its kernel class; the DP kernels start behind it without a host round trip.
//
// The device plan is a PREFETCH, the host remains the authority: the host runs its own plan (mm2.cpp AlignJob::step) while the DP kernels
// are in flight and then looks every DP problem it needs up among the device's results BY KEY (the eight integers of mm2::DpKey).  What it
// does not find it asks for in a later DP round, exactly as before -- a case the kernel declines (below) or a wrong device plan costs
// time, never correctness.  nsgpu_align_stats.plan_* count hits and misses; tests assert the hit rate on the ordinary inputs.
//
// What the kernel takes: alignments whose anchors end in ONE chain (every alignment on a genome without repeats; chain_finish below is
// complete, the region bookkeeping for several chains stays on the host), without long-gap seed filtering (align.c:386-457 acts on chains
// with two or more indels > 10 bp between adjacent anchors) and whose DP problems the narrow register kernels serve (a problem for the
// <8,5> class -- targets beyond 1536 -- is left to the host: its launch would hold LDS for every alignment that does not need it).
// Everything else is flagged per alignment (PlanOut.flags) and planned by the host as before: 30 of 94 345 alignments of a cfg2 step.
#include "common.hpp"
#include "host_util.hpp"
#include "mm2.hpp"
#include "ksw2.hpp"
#include "ksw_class.hpp"

namespace nsgpu {
namespace {

constexpr int kMaxEnds = 256, kMaxTasks = 256, kThreads = 256;
constexpr int EZ_RIGHT = 0x02, EZ_APPROX_MAX = 0x08, EZ_EXTZ_ONLY = 0x40, EZ_REV_CIGAR = 0x80;
constexpr int32_t KNEG = -0x40000000;

__device__ __forceinline__ int nt4_of(uint8_t ch)
{
    if (ch < 4) return ch;
    switch (ch | 0x20) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': case 'u': return 3;
    }
    return 4;
}

struct Lds {
    int32_t *X, *Y, *F, *P, *V, *T;
    uint8_t *SP, *FL;
    unsigned long long *E, *E2;
    PlanKey *K;
    uint32_t *tq;        // per task: class | rank inside the alignment's tasks of that class << 8
    uint32_t *tp, *tc, *ts;      // per task: traceback bytes / CIGAR entries / sequence bytes (then: their exclusive sums)
    int32_t *sh;         // shared scalars
};

}  // namespace

// One workgroup of four waves per alignment: the passes over the anchors are spread over all of them (a launch lasts as long as its longest
// list), the few sequential steps run on thread 0.
__global__ __launch_bounds__(kThreads) void align_plan_kernel(const SeedResult *__restrict__ seeded, const mm2::Anchor *__restrict__ anchors, const int32_t *__restrict__ f_in,
                                                              const int32_t *__restrict__ p_in, const PlanPair *__restrict__ pairs, PlanOut *__restrict__ out,
                                                              PlanKey *__restrict__ keys_out, PlanDp dp, PlanCfg cfg, uint32_t lds_anchors)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t b = blockIdx.x;
    const SeedResult sr = seeded[b];
    const PlanPair pp = pairs[b];
    // the alignment's task slots start out empty (all-zero task and result: what the collecting kernel and the host take for "unused")
    for (uint32_t i = tid; i < pp.task_cap * (uint32_t)(sizeof(KswTask) / 4); i += kThreads) reinterpret_cast<uint32_t *>(dp.tasks + pp.task_base)[i] = 0u;
    for (uint32_t i = tid; i < pp.task_cap * (uint32_t)(sizeof(KswResult) / 4); i += kThreads) reinterpret_cast<uint32_t *>(dp.res + pp.task_base)[i] = 0u;
    if (sr.flags || sr.n == 0 || sr.n > lds_anchors || pp.task_cap == 0) {
        if (tid == 0) out[b] = PlanOut{0u, PLAN_NONE, 0u, 0u};
        return;
    }
    const int32_t n = (int32_t)sr.n;
    Lds L;
    {
        uint8_t *p = lds_raw;
        auto carve = [&](size_t bytes) { uint8_t *q = p; p += (bytes + 15) & ~(size_t)15; return q; };
        L.X = (int32_t *)carve(4 * (size_t)n), L.Y = (int32_t *)carve(4 * (size_t)n), L.F = (int32_t *)carve(4 * (size_t)n), L.P = (int32_t *)carve(4 * (size_t)n);
        L.V = (int32_t *)carve(4 * ((size_t)n + 64)), L.T = (int32_t *)carve(4 * (size_t)n);
        L.SP = carve((size_t)n), L.FL = carve((size_t)n);
        L.E = (unsigned long long *)carve(8 * kMaxEnds), L.E2 = (unsigned long long *)carve(8 * kMaxEnds);
        L.K = (PlanKey *)carve(sizeof(PlanKey) * kMaxTasks);
        L.tq = (uint32_t *)carve(4 * kMaxTasks), L.tp = (uint32_t *)carve(4 * kMaxTasks), L.tc = (uint32_t *)carve(4 * kMaxTasks), L.ts = (uint32_t *)carve(4 * kMaxTasks);
        L.sh = (int32_t *)carve(4 * 48);
    }
    auto fail = [&](uint32_t why) { if (tid == 0) out[b] = PlanOut{0u, why, 0u, 0u}; };       // (every thread takes the same exits: the conditions are uniform)
    const mm2::Anchor *a = anchors + sr.base;
    const int32_t *fi = f_in + sr.base, *pi = p_in + sr.base;
    if (tid < 48) L.sh[tid] = 0;
    __syncthreads();
    for (int32_t i = tid; i < n; i += kThreads) {
        const uint64_t x = a[i].x, y = a[i].y;
        L.X[i] = (int32_t)x, L.Y[i] = (int32_t)y, L.SP[i] = (uint8_t)(y >> 32), L.FL[i] = (uint8_t)(y >> 40);
        const int32_t pj = pi[i];
        L.F[i] = fi[i], L.P[i] = pj, L.V[i] = fi[i], L.T[i] = pj;
        if ((x >> 31) != 0) L.sh[47] = 1;           // rid != 0, reverse strand or a position beyond 2^31: not the engine's lists
    }
    __syncthreads();
    if (L.sh[47]) { fail(PLAN_COMPLEX); return; }
    const int min_cnt = cfg.min_cnt, min_sc = cfg.min_sc;

    // ---- chain_finish (chain.c:94-164) ----
    // v[i] = the best f on the way back from i along the predecessor links (chain.c:97: v[i] = max(v[p[i]], f[i]), sequential there).  Here by
    // pointer jumping: J[i] (in T) starts at p[i]; a round takes v[J[i]] into v[i] and sets J[i] = J[J[i]]: after round k v[i] covers the 2^k
    // nearest ancestors.  Reads and writes of a round are separated by a barrier: log2(n) rounds instead of n dependent steps of one lane.
    for (;;) {
        int32_t nv[8], nj[8];            // (n <= 8 * kThreads is not assumed: the loop below re-reads for longer lists)
        bool more = false;
        if (n <= 8 * kThreads) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int32_t i = tid + u * kThreads;
                nv[u] = 0, nj[u] = -2;
                if (i < n) { const int32_t j = L.T[i]; if (j >= 0) { nv[u] = L.V[j], nj[u] = L.T[j]; } }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int32_t i = tid + u * kThreads;
                if (nj[u] != -2) { if (nv[u] > L.V[i]) L.V[i] = nv[u]; L.T[i] = nj[u]; more |= nj[u] >= 0; }
            }
        } else {
            // longer lists: strip by strip, ascending (a strip reads pairs (v[J], J[J]) that are either both old or both new: see above)
            for (int32_t i0 = 0; i0 < n; i0 += kThreads) {
                const int32_t i = i0 + tid;
                const int32_t j = i < n ? L.T[i] : -1;
                const int32_t vj = j >= 0 ? L.V[j] : 0, jj = j >= 0 ? L.T[j] : -1;
                __syncthreads();
                if (j >= 0) { if (vj > L.V[i]) L.V[i] = vj; L.T[i] = jj; more |= jj >= 0; }
                __syncthreads();
            }
        }
        if (!__syncthreads_or(more ? 1 : 0)) break;
    }
    for (int32_t i = tid; i < n; i += kThreads) L.T[i] = 0;
    __syncthreads();
    for (int32_t i = tid; i < n; i += kThreads) { const int32_t pj = L.P[i]; if (pj >= 0) L.T[pj] = 1; }
    __syncthreads();
    // chain ends: nobody's predecessor, good enough; each walks back to the last anchor that holds the best score
    for (int32_t i = tid; i < n; i += kThreads)
        if (L.T[i] == 0 && L.V[i] >= min_sc) {
            int32_t j = i;
            while (j >= 0 && L.F[j] < L.V[j]) j = L.P[j];
            if (j < 0) j = i;
            const int32_t at = atomicAdd(&L.sh[0], 1);
            if (at < kMaxEnds) L.E[at] = (unsigned long long)(uint32_t)L.F[j] << 32 | (uint32_t)j;
        }
    __syncthreads();
    const int32_t n_e = L.sh[0];
    if (n_e > kMaxEnds) { fail(PLAN_COMPLEX); return; }
    if (n_e == 0) { fail(0u); return; }                            // no chain: no hit, nothing to align
    // best first (radix_sort_64 + reversal in the reference: equal keys are identical values, any order of them is the same array)
    for (int32_t e = tid; e < n_e; e += kThreads) {
        const unsigned long long me = L.E[e];
        int32_t rank = 0;
        for (int32_t x = 0; x < n_e; ++x) { const unsigned long long o = L.E[x]; rank += (o > me || (o == me && x < e)) ? 1 : 0; }
        L.E2[rank] = me;
    }
    // The backtrack (chain.c:110-126) of the BEST end walks to the root -- nothing is used yet -- so its chain is the set of ancestors of
    // that end: marked by pointer jumping again (a marked node marks its current jump target, then everybody jumps: after round k the
    // ancestors at every distance < 2^(k+1) are marked; the jumps are synchronous -- two buffers, T and V -- so that they stay powers of two).
    // The marks live in bit 30 of p[] (no parent = kNoPar): the sequential walks of the other ends below get the next anchor and whether it
    // is used from one load.
    constexpr int32_t kMark = 1 << 30, kNoPar = kMark - 1;
    __syncthreads();
    for (int32_t i = tid; i < n; i += kThreads) { const int32_t pj = L.P[i]; L.T[i] = pj; L.P[i] = pj < 0 ? kNoPar : pj; }
    __syncthreads();
    const int32_t j0 = (int32_t)(uint32_t)L.E2[0], sc0 = (int32_t)(L.E2[0] >> 32);
    if (tid == 0) L.P[j0] |= kMark;
    __syncthreads();
    {
        int32_t *Jc = L.T, *Jn = L.V;
        for (;;) {
            bool more = false;
            for (int32_t i = tid; i < n; i += kThreads) {
                const int32_t j = Jc[i];
                if (j >= 0 && (L.P[i] & kMark)) atomicOr(&L.P[j], kMark);          // (distinct nodes of one path have distinct targets; the atomic keeps the OR whole)
                const int32_t jj = j >= 0 ? Jc[j] : -1;
                Jn[i] = jj;
                more |= jj >= 0;
            }
            int32_t *t_ = Jc; Jc = Jn; Jn = t_;
            if (!__syncthreads_or(more ? 1 : 0)) break;
        }
        // one more marking round with the last jump targets (they were computed, not yet used)
        for (int32_t i = tid; i < n; i += kThreads) { const int32_t j = Jc[i]; if (j >= 0 && (L.P[i] & kMark)) atomicOr(&L.P[j], kMark); }
        __syncthreads();
    }
    // its length; the other ends, sequentially, only to see whether any of them makes a second chain (then the host's region bookkeeping decides)
    {
        int32_t c = 0;
        for (int32_t i = tid; i < n; i += kThreads) c += (L.P[i] & kMark) != 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) atomicAdd(&L.sh[1], c);
    }
    __syncthreads();
    const int32_t cnt = L.sh[1];
    if (tid == 0) {
        int32_t k = cnt >= min_cnt ? 1 : 0;                              // (the best end reaches the root: j < 0 in chain.c:119)
        const int32_t k_first = k;
        for (int32_t i = 1; i < n_e && k == k_first; ++i) {
            const int32_t sc = (int32_t)(L.E2[i] >> 32);
            int32_t j = (int32_t)(uint32_t)L.E2[i], steps = 0;
            int32_t raw = L.P[j];
            for (;;) {                                        // do { v[n_v++] = j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
                ++steps;
                L.P[j] = raw | kMark;
                const int32_t par = raw & kNoPar;
                if (par == kNoPar) { j = -1; break; }
                j = par;
                raw = L.P[j];
                if (raw & kMark) break;
            }
            if (j < 0) { if (steps >= min_cnt) ++k; }
            else if (sc - L.F[j] >= min_sc) { if (steps >= min_cnt) ++k; }
        }
        L.sh[2] = (k != k_first || k_first == 0) ? 1 : 0;      // a second chain, or the best end's chain is too short while ... : not the one-chain case
        if (k_first == 0 && k == 0) L.sh[2] = 2;               // no chain at all
    }
    __syncthreads();
    if (L.sh[2] == 2) { fail(0u); return; }
    if (L.sh[2]) { fail(PLAN_COMPLEX); return; }
    (void)sc0;
    // the chain in ascending order (p[i] < i: the order of the indices) into dense arrays, re-using F / P / T for x / y / span + flags: anchor
    // i of the chain goes to its rank among the marked.  (mm_chain_dp leaves exactly these anchors in a[]; mm_gen_regs / chain_post /
    // mm_squeeze_a are the identity on one chain.)  Strips ascending; a strip's writes land at or below its own first index.
    {
        int32_t base = 0;
        for (int32_t i0 = 0; i0 < n; i0 += kThreads) {
            const int32_t i = i0 + tid;
            const bool mk = i < n && (L.P[i] & kMark);
            const int32_t x = mk ? L.X[i] : 0, y = mk ? L.Y[i] : 0, sf = mk ? ((int32_t)L.SP[i] | (int32_t)L.FL[i] << 8) : 0;
            const unsigned long long bal = __ballot(mk);
            if (lane == 0) L.sh[24 + wv] = __popcll(bal);
            __syncthreads();
            int32_t before = 0;
            for (int w2 = 0; w2 < wv; ++w2) before += L.sh[24 + w2];
            const int32_t total = L.sh[24] + L.sh[25] + L.sh[26] + L.sh[27];
            const int32_t m = base + before + __popcll(bal & ((1ull << lane) - 1));
            __syncthreads();
            if (mk) L.F[m] = x, L.P[m] = y, L.T[m] = sf;
            base += total;
            __syncthreads();
        }
    }
    int32_t *CX = L.F, *CY = L.P, *CS = L.T;
#define CSPAN(i) (CS[(i)] & 0xff)
    // mm_reg_set_coor's match length (hit.c:8-38): fix_bad_ends looks at it.  A sum over the chain.
    {
        int32_t ml = 0;
        for (int32_t i = 1 + tid; i < cnt; i += kThreads) {
            const int span = CSPAN(i), tl = CX[i] - CX[i - 1], ql = CY[i] - CY[i - 1];
            ml += tl > span && ql > span ? span : tl < ql ? tl : ql;
        }
        for (int o = 32; o > 0; o >>= 1) ml += __shfl_xor(ml, o, 64);
        if (lane == 0) atomicAdd(&L.sh[28], ml);
    }
    __syncthreads();
    const int32_t mlen = L.sh[28] + CSPAN(0);
    // ---- one region through the plan half of mm_align1 (align.c:565-795) ----
    const int kk = cfg.k >> 1;
    const int bw = (int)(cfg.bw * 1.5 + 1.);
    // mm_fix_bad_ends (align.c:459-493): two short walks from the ends (they stop after ~2 bw bases); thread 0
    if (tid == 0) {
        int32_t as1 = 0, cnt1 = cnt;
        if (cnt >= 3) {
            const int min_match = cfg.min_sc * 2;
            int32_t m, l;
            m = l = CSPAN(0);
            for (int32_t i = 1; i < cnt - 1; ++i) {
                const int32_t q_span = CSPAN(i);
                const int32_t lr = CX[i] - CX[i - 1], lq = CY[i] - CY[i - 1];
                const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
                if (mx - mn > l >> 1) as1 = i;
                l += mn;
                m += mn < q_span ? mn : q_span;
                if (l >= cfg.bw << 1 || (m >= min_match && m >= cfg.bw) || m >= mlen >> 1) break;
            }
            cnt1 = cnt - as1;
            m = l = CSPAN(cnt - 1);
            for (int32_t i = cnt - 2; i > as1; --i) {
                const int32_t q_span = CSPAN(i + 1);
                const int32_t lr = CX[i + 1] - CX[i], lq = CY[i + 1] - CY[i];
                const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
                if (mx - mn > l >> 1) cnt1 = i + 1 - as1;
                l += mn;
                m += mn < q_span ? mn : q_span;
                if (l >= cfg.bw << 1 || (m >= min_match && m >= cfg.bw) || m >= mlen >> 1) break;
            }
        }
        L.sh[6] = as1, L.sh[7] = cnt1;
    }
    __syncthreads();
    const int32_t as1 = L.sh[6], cnt1 = L.sh[7];
    // mm_filter_bad_seeds / _alt (align.c:386-457) act only with two or more long gaps between adjacent anchors: those chains stay the host's
    {
        int32_t k10 = 0, k30 = 0;
        for (int32_t i = 1 + tid; i < cnt1; i += kThreads) {
            const int gap = (CY[as1 + i] - CY[as1 + i - 1]) - (CX[as1 + i] - CX[as1 + i - 1]);
            k10 += gap < -10 || gap > 10, k30 += gap < -30 || gap > 30;
        }
        for (int o = 32; o > 0; o >>= 1) k10 += __shfl_xor(k10, o, 64), k30 += __shfl_xor(k30, o, 64);
        if (lane == 0) { atomicAdd(&L.sh[29], k10); atomicAdd(&L.sh[30], k30); }
    }
    // The gap fills (align.c:709-765) cut the stretch between the first and the last kept anchor greedily: from the anchor the last fill ended
    // at (b) to the first later kept anchor that is the last one, carries MM_SEED_LONG_JOIN, or lies >= min_ksw_len further on BOTH sequences.
    // That next cut is a function of b alone: every thread finds it for its anchors (a handful of steps), thread 0 then follows the cuts.
    // The seed flags sit in the byte above the span: 1 = MM_SEED_LONG_JOIN, 2 = MM_SEED_IGNORE (both only ever set by code that does not run
    // for this chain), 4 = MM_SEED_TANDEM.
    int32_t *NXT = L.V;
    for (int32_t b0 = tid; b0 < cnt1 - 1; b0 += kThreads) {
        const int32_t xb = CX[as1 + b0], yb = CY[as1 + b0];
        int32_t i = b0 + 1;
        for (; i < cnt1 - 1; ++i) {
            const int fl = CS[as1 + i] >> 8;
            if (fl & 0x06) continue;
            if ((fl & 0x01) || (CY[as1 + i] - yb >= cfg.min_ksw_len && CX[as1 + i] - xb >= cfg.min_ksw_len)) break;
        }
        NXT[b0] = i;
    }
    __syncthreads();
    if (tid == 0) {
        int32_t flags = (L.sh[29] > 1 || L.sh[30] > 1) ? (int32_t)PLAN_BADSEEDS : 0, n_t = 0;
        const int qlen = (int)pp.qlen, ref_len = (int)pp.ref_len;
        int32_t rs, qs, re, qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, l;
        rs = CX[as1] - kk, qs = CY[as1] - kk;                                   // mm_adjust_minier without HPC (align.c:350-365)
        re = CX[as1 + cnt1 - 1] - kk, qe = CY[as1 + cnt1 - 1] - kk;
        // DP windows (align.c:617-677); with one chain no anchor lies outside the region, so rs1 / qs1 / re1 / qe1 start from their defaults
        rs0 = CX[0] + 1 - CSPAN(0);
        qs0 = CY[0] + 1 - CSPAN(0);
        if (rs0 < 0) rs0 = 0;
        rs1 = qs1 = 0;
        if (qs > 0 && rs > 0) {
            l = qs < cfg.max_gap ? qs : cfg.max_gap;
            qs1 = qs1 > qs - l ? qs1 : qs - l;
            qs0 = qs0 < qs1 ? qs0 : qs1;
            l += l * cfg.a > cfg.q ? (l * cfg.a - cfg.q) / cfg.e : 0;
            l = l < cfg.max_gap ? l : cfg.max_gap;
            l = l < rs ? l : rs;
            rs1 = rs1 > rs - l ? rs1 : rs - l;
            rs0 = rs0 < rs1 ? rs0 : rs1;
            rs0 = rs0 < rs ? rs0 : rs;
        } else rs0 = rs, qs0 = qs;
        re0 = CX[cnt - 1] + 1;
        qe0 = CY[cnt - 1] + 1;
        re1 = ref_len, qe1 = qlen;
        if (qe < qlen && re < ref_len) {
            l = qlen - qe < cfg.max_gap ? qlen - qe : cfg.max_gap;
            qe1 = qe1 < qe + l ? qe1 : qe + l;
            qe0 = qe0 > qe1 ? qe0 : qe1;
            l += l * cfg.a > cfg.q ? (l * cfg.a - cfg.q) / cfg.e : 0;
            l = l < cfg.max_gap ? l : cfg.max_gap;
            l = l < ref_len - re ? l : ref_len - re;
            re1 = re1 < re + l ? re1 : re + l;
            re0 = re0 > re1 ? re0 : re1;
        } else re0 = re, qe0 = qe;
        auto push = [&](int kqs, int kqe, int krs, int kre, int w, int end_bonus, int flag) {
            if (n_t >= kMaxTasks || (uint32_t)n_t >= pp.task_cap) { flags |= PLAN_FULL; return; }
            PlanKey &K = L.K[n_t++];
            K.qs = kqs, K.qe = kqe, K.rs = krs, K.re = kre, K.w = w, K.zdrop = cfg.zdrop, K.end_bonus = end_bonus, K.flag = flag;
        };
        if (qs > 0 && rs > 0) push(qs0, qs, rs0, rs, bw, cfg.end_bonus, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR);
        for (int32_t b0 = 0; b0 < cnt1 - 1;) {
            const int32_t i = NXT[b0];
            const int32_t frs = CX[as1 + b0] - kk, fqs = CY[as1 + b0] - kk, fre = CX[as1 + i] - kk, fqe = CY[as1 + i] - kk;
            int bw1 = bw;
            if ((CS[as1 + i] >> 8) & 0x01) bw1 = fqe - fqs > fre - frs ? fqe - fqs : fre - frs;
            push(fqs, fqe, frs, fre, bw1, -1, EZ_APPROX_MAX);
            b0 = i;
        }
        if (qe < qe0 && re < re0) push(qe, qe0, re, re0, bw, cfg.end_bonus, EZ_EXTZ_ONLY);
        L.sh[3] = flags, L.sh[4] = n_t;
        L.sh[31] = 0x7fffffff, L.sh[32] = -1, L.sh[33] = 0x7fffffff, L.sh[34] = -1, L.sh[35] = 0, L.sh[36] = 0, L.sh[37] = 0;
    }
    __syncthreads();
    int32_t flags = L.sh[3];
    const int32_t n_t = L.sh[4];
    // ---- per task: kernel class, scratch needs, span check (threads over tasks) ----
    // The tasks' sequences are laid out once per alignment, not per task: the gap fills tile the stretch between the first and the last kept
    // anchor and the right extension continues it, so ONE forward copy of the query hull and one of the target hull (0..4 codes) serve all of
    // them (a task's qoff / toff point into the hulls); the left extension, whose sequences are reversed (align.c:693-696), gets its own two.
    for (int32_t t = tid; t < n_t; t += kThreads) {
        const PlanKey K = L.K[t];
        const int ql = K.qe - K.qs, tl = K.re - K.rs;
        uint32_t cls = 255, pb = 0, cg = 0, sq = 0, bad = 0;
        if (ql < 0 || tl < 0 || K.qs < 0 || K.qe > (int)pp.qlen) bad |= PLAN_COMPLEX;
        else if (ql > 0 && tl > 0) {
            if (K.rs < (int)pp.ref_lo || K.re > (int)(pp.ref_lo + pp.ref_n)) bad |= PLAN_SPAN;
            const int c = ksw_launch_class_hd(ql, tl, K.w, K.flag, cfg.kp, cfg.kc, cfg.two_phase != 0);
            const size_t pbytes = (ksw_p_bytes_hd(ql, tl, K.w) + 63) & ~(size_t)63;
            // (the widest class, targets beyond 1536 columns: a handful of problems per thousand slots on an iid genome, two per slot on one with
            // repeats -- launched from a device list of its own with a small grid, ksw2.hip dev_class_grid: what does not fit is the host's)
            if (c < 0 || c == 7 || ql > cfg.q_max || pbytes >= (1ull << 32)) bad |= PLAN_CLASS;
            else {
                cls = (uint32_t)c, pb = (uint32_t)pbytes, cg = (uint32_t)(ql + tl + 2);
                if (ksw_class_is_slow(c)) atomicOr(&L.sh[37], 1);
                if (K.flag & EZ_RIGHT) { sq = (uint32_t)(ql + tl); atomicAdd(&L.sh[35], ql + tl); }       // its own reversed copies
                else { atomicMin(&L.sh[31], K.qs); atomicMax(&L.sh[32], K.qe); atomicMin(&L.sh[33], K.rs); atomicMax(&L.sh[34], K.re); }
            }
        }
        if (bad) atomicOr(&L.sh[36], (int32_t)bad);
        L.tq[t] = cls, L.tp[t] = pb, L.tc[t] = cg, L.ts[t] = sq;
    }
    __syncthreads();
    flags |= L.sh[36];
    if (flags) { fail((uint32_t)flags); return; }
    if (n_t == 0) { fail(0u); return; }
    const int32_t qa = L.sh[31], ta = L.sh[33], n_rev = L.sh[35];
    const uint32_t len_q = L.sh[32] > qa ? (uint32_t)(L.sh[32] - qa) : 0u, len_t = L.sh[34] > ta ? (uint32_t)(L.sh[34] - ta) : 0u;
    // ---- room: one atomic per cursor and per class for the whole alignment (all of them in flight at once: one lane each) ----
    if (tid == 0) {
        unsigned long long sp = 0; uint32_t sc = 0, ss = len_q + len_t;
        uint32_t seen[KSW_REG_CLASSES];
        for (int c = 0; c < KSW_REG_CLASSES; ++c) seen[c] = 0;
        for (int32_t t = 0; t < n_t; ++t) {
            const uint32_t pb = L.tp[t], cg = L.tc[t], sq = L.ts[t], cls = L.tq[t];
            L.tp[t] = (uint32_t)sp, L.tc[t] = sc, L.ts[t] = ss;         // exclusive sums (the traceback of one alignment stays below 4 GiB: checked below)
            sp += pb, sc += cg, ss += sq;
            if (cls < KSW_REG_CLASSES) L.tq[t] = cls | (seen[cls]++ << 8);       // + its rank among the alignment's tasks of the class
        }
        reinterpret_cast<unsigned long long *>(L.sh + 8)[0] = sp;
        L.sh[10] = (int32_t)sc, L.sh[11] = (int32_t)ss;
        for (int c = 0; c < KSW_REG_CLASSES; ++c) L.sh[12 + c] = (int32_t)seen[c];
    }
    __syncthreads();
    if (wv == 0) {
        const unsigned long long sp = reinterpret_cast<unsigned long long *>(L.sh + 8)[0];
        const uint32_t sc = (uint32_t)L.sh[10], ss = (uint32_t)L.sh[11];
        const uint32_t cnt_c = lane >= 3 && lane < 3 + KSW_REG_CLASSES ? (uint32_t)L.sh[12 + lane - 3] : 0u;
        unsigned long long got = 0;
        bool full = sp >= (1ull << 32);
        if (lane == 0) { got = atomicAdd(&dp.cursors[0], sp); full |= got + sp > dp.p_cap; }
        else if (lane == 1) { got = atomicAdd(&dp.cursors[1], (unsigned long long)sc); full |= got + sc > dp.cig_cap; }
        else if (lane == 2) { got = atomicAdd(&dp.cursors[2], (unsigned long long)ss); full |= got + ss > dp.seq_cap; }
        else if (cnt_c) {
            // a class list is served by a grid of dp.class_grid[c] workgroups: what does not fit is left to the host, and the entries this
            // alignment reserved inside the grid are voided (the DP kernels skip ~0u)
            got = atomicAdd(&dp.class_cnt[lane - 3], cnt_c);
            const uint32_t cap_c = dp.class_grid[lane - 3];
            if (got + cnt_c > cap_c) { full = true; for (unsigned long long e = got; e < got + cnt_c && e < cap_c; ++e) dp.class_list[(size_t)(lane - 3) * dp.n_slots + e] = ~0u; }
        }
        const bool any_full = __ballot(full) != 0;
        if (lane == 0) reinterpret_cast<unsigned long long *>(L.sh + 8)[0] = got;
        else if (lane == 1) L.sh[10] = (int32_t)(uint32_t)got;
        else if (lane == 2) L.sh[11] = (int32_t)(uint32_t)got;
        else if (lane < 3 + KSW_REG_CLASSES) L.sh[12 + lane - 3] = (int32_t)(uint32_t)got;
        if (any_full) {
            // (classes of this alignment that did fit keep their reservation: void it as well)
            if (lane >= 3 && cnt_c && !full) for (unsigned long long e = got; e < got + cnt_c; ++e) dp.class_list[(size_t)(lane - 3) * dp.n_slots + e] = ~0u;
            if (lane == 0) L.sh[5] = 1;
        }
    }
    __syncthreads();
    if (L.sh[5]) { fail(PLAN_FULL); return; }
    const unsigned long long p0 = reinterpret_cast<unsigned long long *>(L.sh + 8)[0];
    const uint32_t c0 = (uint32_t)L.sh[10], s0 = (uint32_t)L.sh[11];
    for (int32_t t = tid; t < n_t; t += kThreads) {
        const PlanKey K = L.K[t];
        const uint32_t slot = pp.task_base + (uint32_t)t;
        const int ql = K.qe - K.qs, tl = K.re - K.rs;
        const uint32_t cls = L.tq[t] & 0xffu, rank = L.tq[t] >> 8;
        KswTask tk;
        if (K.flag & EZ_RIGHT) tk.qoff = s0 + L.ts[t], tk.toff = tk.qoff + (uint32_t)(ql > 0 ? ql : 0);
        else tk.qoff = s0 + (uint32_t)(K.qs - qa), tk.toff = s0 + len_q + (uint32_t)(K.rs - ta);
        if (cls >= KSW_REG_CLASSES) tk.qoff = tk.toff = s0;           // an empty problem reads nothing
        tk.qlen = ql, tk.tlen = tl, tk.w = K.w, tk.zdrop = K.zdrop, tk.end_bonus = K.end_bonus, tk.flag = K.flag | cfg.kc.flag_or | ((K.flag & EZ_APPROX_MAX) ? cfg.approx_flag_or : (K.flag & EZ_EXTZ_ONLY) ? cfg.ext_flag_or : 0);
        tk.p_off = p0 + L.tp[t], tk.cig_off = c0 + L.tc[t], tk.out_idx = slot;
        dp.tasks[slot] = tk;
        dp.task_pair[slot] = b;
        keys_out[slot] = K;
        if (cls < KSW_REG_CLASSES) dp.class_list[(size_t)cls * dp.n_slots + (uint32_t)L.sh[12 + cls] + rank] = slot;
        else {
            // an empty problem: what ksw_reset_extz leaves (ksw2.h:103-110), as the reference returns at once
            KswResult o;
            o.max = 0, o.zdropped = 0, o.max_q = o.max_t = o.mqe_t = o.mte_q = -1, o.mqe = o.mte = o.score = KNEG, o.n_cigar = 0, o.reach_end = 0;
            dp.res[slot] = o;
            // (no DP kernel will count it on its alignment: ksw_collect.hpp's hand-over waits for n_tasks problems.  The DP kernels run behind
            // this one; an alignment ALL of whose problems are empty is handed over by nobody and comes with the batch's end)
            if (dp.pair_done) atomicAdd(&dp.pair_done[b], 1u);
        }
    }
    // ---- the sequences: 16-byte loads, four in flight per thread; dst[x] (or dst[n - 1 - x]) = code of src[x] ----
    auto copy_codes = [&](uint8_t *dst, const uint8_t *src, int nb, bool rev) {
        auto put = [&](int x, uint32_t ch) { dst[rev ? nb - 1 - x : x] = (uint8_t)nt4_of((uint8_t)ch); };
        auto put4 = [&](int x, uint32_t v) { put(x, v & 0xff), put(x + 1, v >> 8 & 0xff), put(x + 2, v >> 16 & 0xff), put(x + 3, v >> 24); };
        int h = (int)((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15);
        h = h < nb ? h : nb;
        if (tid < h) put(tid, src[tid]);
        const int nd = (nb - h) >> 4;
        const uint4 *s16 = reinterpret_cast<const uint4 *>(src + h);
        for (int d0 = 0; d0 < nd; d0 += 4 * kThreads) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int d = d0 + u * kThreads + tid; v[u] = d < nd ? s16[d] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = d0 + u * kThreads + tid;
                if (d < nd) { const int x = h + 16 * d; put4(x, v[u].x), put4(x + 4, v[u].y), put4(x + 8, v[u].z), put4(x + 12, v[u].w); }
            }
        }
        const int t0 = h + 16 * nd;
        if (tid < nb - t0) put(t0 + tid, src[t0 + tid]);
    };
    if (len_q) copy_codes(dp.seqs + s0, pp.qry + qa, (int)len_q, false);
    if (len_t) copy_codes(dp.seqs + s0 + len_q, pp.ref + (ta - (int)pp.ref_lo), (int)len_t, false);
    if (n_rev)
        for (int32_t t = 0; t < n_t; ++t) {
            const PlanKey K = L.K[t];
            if (!(K.flag & EZ_RIGHT) || (L.tq[t] & 0xffu) >= KSW_REG_CLASSES) continue;
            const int ql = K.qe - K.qs, tl = K.re - K.rs;
            uint8_t *dq = dp.seqs + s0 + L.ts[t];
            copy_codes(dq, pp.qry + K.qs, ql, true);
            copy_codes(dq + ql, pp.ref + (K.rs - (int)pp.ref_lo), tl, true);
        }
    if (tid == 0) out[b] = PlanOut{(uint32_t)n_t, 0u, (uint32_t)L.sh[37], 0u};
#undef CSPAN
}

size_t plan_lds_bytes(uint32_t lds_anchors)
{
    const size_t n = lds_anchors;
    auto r16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return 4 * r16(4 * n) + r16(4 * (n + 64)) + r16(4 * n) + 2 * r16(n) + 2 * r16(8 * kMaxEnds) + r16(sizeof(PlanKey) * kMaxTasks) + 4 * r16(4 * kMaxTasks) + r16(4 * 48) + 64;
}

int plan_launch(hipStream_t st, uint32_t n_pairs, uint32_t lds_anchors, const SeedResult *seeded, const mm2::Anchor *anchors, const int32_t *f, const int32_t *p,
                const PlanPair *pairs, PlanOut *out, PlanKey *keys_out, const PlanDp &dp, const PlanCfg &cfg)
{
    if (n_pairs == 0) return NSGPU_OK;
    // a workgroup has 160 KB of LDS: lists beyond what that takes (reads of 70 kb and more: ~6 000 anchors) are the host's to plan -- the
    // kernel leaves every alignment with more anchors than `lds_anchors` alone
    while (lds_anchors > 64 && plan_lds_bytes(lds_anchors) > (size_t)152 << 10) lds_anchors -= lds_anchors / 8;
    const size_t lds = plan_lds_bytes(lds_anchors);
    static LdsAttr attr;
    if (lds > 32768) NS_TRY(attr.raise(lds, reinterpret_cast<const void *>(align_plan_kernel)));
    hipLaunchKernelGGL(align_plan_kernel, dim3(n_pairs), dim3(kThreads), lds, st, seeded, anchors, f, p, pairs, out, keys_out, dp, cfg, lds_anchors);
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

}  // namespace nsgpu
