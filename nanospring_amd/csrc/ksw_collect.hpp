// ksw_collect.hpp -- how the results of a device-planned DP batch reach the host (ksw2.hip ksw_dev_launch): per alignment, its problems' results as
// flat words and their CIGARs compacted into a pinned arena at an offset taken from a device-side cursor, then ONE status word in pinned memory.
// The body is shared by the collecting kernel (ksw2.hip: one workgroup per alignment behind a whole launch) and by the DP kernels themselves
// (ksw2_reg.hip): the problem that completes an alignment -- every problem counts itself on its alignment when it is done -- hands that alignment
// over at once, so that the host can finish and apply it while the launch's longer problems are still running.
#pragma once
#include <cstdint>
#include <cstddef>
#include <hip/hip_runtime.h>
#include "ksw2.hpp"
#include "host_util.hpp"

namespace nsgpu {

constexpr uint32_t kDvCtrlWords = 18;                       // class counters + flags in front of the 64-bit fields
// `done`: written last by the batch's closing kernel into the pinned copy -- everything of the batch has been handed over
struct DvCtrl { uint32_t class_cnt[16]; uint32_t cig_overflow, done; unsigned long long cursors[3]; unsigned long long cells, alg_bytes, cig_out; };
static_assert(offsetof(DvCtrl, cursors) == kDvCtrlWords * 4, "layout");

struct DvCollect {
    const PlanPair *pairs; const PlanOut *outs; const KswTask *tasks; const KswResult *res; const uint32_t *pool;      // the batch, in device memory
    const uint32_t *task_pair;          // task slot -> alignment
    uint32_t *pair_done;                // alignment -> problems completed so far (nullptr: the DP kernels do not hand over, the collecting kernel does)
    KswResult *h_res; uint64_t *h_off; uint32_t *h_cig; uint64_t h_cig_cap; uint32_t *h_status;                         // pinned landing zones
    uint32_t *h_check;                  // per alignment: dv_check_* over every word handed over (the host recomputes it from what it reads before it believes the status word)
    uint32_t epoch;                     // the batch's number on its workspace: seeds the check, so that a previous batch's self-consistent words never pass
    DvCtrl *ctrl;
};

// The check word of an alignment's hand-over: seeded with the batch's epoch and the alignment's index, every word weighted by its POSITION in the
// hand-over (result words, then the CIGAR entries in task order, then the offsets) -- words that arrive permuted, or a previous batch's words
// under this batch's status word, do not add up.  Host (align_batch.hip) and device compute the same sum.
__host__ __device__ inline uint32_t dv_check_seed(uint32_t epoch, uint32_t pair) { return epoch * 0x9E3779B1u ^ (pair + 1u) * 0x85EBCA6Bu; }
__host__ __device__ inline uint32_t dv_check_term(uint32_t idx, uint32_t word) { return word * (2u * idx + 1u); }

// One wave (lane 0 .. 63) hands alignment b over.  s_off: 256 words of LDS of the caller's.
__device__ inline void dev_collect_pair(const DvCollect &dc, uint32_t b, uint32_t lane, uint32_t *s_off)
{
    const PlanOut o = dc.outs[b];
    const uint32_t s0 = dc.pairs[b].task_base, n = o.n_tasks;          // (n <= 256: plan.hip's kMaxTasks)
    // CIGAR entries of the alignment's tasks, their exclusive sums (tasks over lanes, 64 at a time)
    unsigned long long total = 0, cells = 0, alg = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 64) {
        const uint32_t t = t0 + lane;
        const uint32_t c = t < n ? (uint32_t)dc.res[s0 + t].n_cigar : 0u;
        uint32_t inc = c;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)inc, d, 64); if ((int)lane >= d) inc += x; }
        if (t < n) {
            s_off[t] = (uint32_t)total + inc - c;         // relative to the alignment's base
            const KswTask tk = dc.tasks[s0 + t];
            if (tk.qlen > 0 && tk.tlen > 0) cells += (unsigned long long)tk.qlen * (unsigned long long)tk.tlen, alg += (unsigned long long)tk.qlen + tk.tlen + 4ull * c + sizeof(KswResult);
        }
        total += (unsigned long long)(uint32_t)__shfl((int)inc, 63, 64);
    }
    for (int d = 32; d > 0; d >>= 1) cells += __shfl_xor((long long)cells, d, 64), alg += __shfl_xor((long long)alg, d, 64);
    unsigned long long base = 0;
    if (lane == 0) {
        base = atomicAdd(&dc.ctrl->cig_out, total);
        atomicAdd(&dc.ctrl->cells, cells);
        atomicAdd(&dc.ctrl->alg_bytes, alg);
    }
    base = (unsigned long long)__shfl((long long)base, 0, 64);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                       // s_off is this wave's
    if (base + total > dc.h_cig_cap) {                     // the host redoes this alignment's problems (and sizes the arena for the total next time)
        if (lane == 0) dc.h_status[b] = 2u;
        return;
    }
    uint32_t chk = 0;
    const uint32_t n_rw = n * (uint32_t)(sizeof(KswResult) / 4);
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(dc.res + s0);
        uint32_t *dst = reinterpret_cast<uint32_t *>(dc.h_res + s0);
        for (uint32_t i = lane; i < n_rw; i += 64) { const uint32_t v = src[i]; dst[i] = v; chk += dv_check_term(i, v); }
    }
    for (uint32_t t = 0; t < n; ++t) {
        const uint32_t c = (uint32_t)dc.res[s0 + t].n_cigar;
        const unsigned long long at = base + s_off[t];
        const uint32_t *src = dc.pool + dc.tasks[s0 + t].cig_off;
        for (uint32_t k = lane; k < c; k += 64) { const uint32_t v = src[k]; dc.h_cig[at + k] = v; chk += dv_check_term(n_rw + s_off[t] + k, v); }
    }
    for (uint32_t t = lane; t < n; t += 64) { dc.h_off[s0 + t] = base + s_off[t]; chk += dv_check_term(n_rw + (uint32_t)total + t, (uint32_t)(base + s_off[t])); }
    for (int d = 32; d > 0; d >>= 1) chk += (uint32_t)__shfl_xor((int)chk, d, 64);
    if (lane == 0) dc.h_check[b] = chk + dv_check_seed(dc.epoch, b);
    __threadfence_system();
    // The host may be watching the status word while this kernel is still running (ksw_dev_poll): the word must not overtake the data on the
    // way to host memory.  Every lane reads one word of what it wrote back from host memory (a system-scope load: a read request does not pass
    // the posted writes in front of it on the link), and only then does lane 0 raise the word.
    {
        uint32_t probe = 0;
        if (lane < n) probe = __hip_atomic_load(reinterpret_cast<const uint32_t *>(dc.h_off + s0 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (total) probe += __hip_atomic_load(dc.h_cig + base + (total - 1 - (lane % (total < 64 ? total : 64))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        probe += __hip_atomic_load(reinterpret_cast<const uint32_t *>(dc.h_res + s0) + lane % (n * (uint32_t)(sizeof(KswResult) / 4)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("" :: "v"(probe) : "memory");
    }
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(dc.h_status + b, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// At the end of a DP kernel's workgroup, by ALL its threads: the problem in task slot ti is complete -- its result and CIGAR are in device memory.
// Counts it for its alignment; the problem that completes the alignment hands it over.  lds: at least 1028 bytes of the workgroup's LDS, free now.
__device__ inline void dev_problem_done(const DvCollect &dc, uint32_t ti, uint8_t *lds)
{
    if (!dc.pair_done) return;
    __threadfence();                                       // every thread's own stores (CIGAR entries, the result)
    __syncthreads();
    uint32_t *s_off = reinterpret_cast<uint32_t *>(lds);
    uint32_t *s_last = s_off + 256;
    if (threadIdx.x == 0) {
        const uint32_t b = dc.task_pair[ti];
        const PlanOut o = dc.outs[b];
        const uint32_t old = atomicAdd(&dc.pair_done[b], 1u);
        *s_last = o.flags == 0 && old + 1 == o.n_tasks ? b + 1 : 0u;
    }
    __syncthreads();
    const uint32_t last = *s_last;
    if (last && threadIdx.x < 64) {
        __threadfence();                                   // the other problems' stores, counted before this one
        dev_collect_pair(dc, last - 1, threadIdx.x, s_off);
    }
}

}  // namespace nsgpu
