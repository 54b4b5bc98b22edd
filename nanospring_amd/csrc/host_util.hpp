// host_util.hpp -- host thread pool helpers and the internal align-request interface.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>
#include "mm2.hpp"
#include "ksw2.hpp"
#include "ksw_class.hpp"

struct nsgpu_ctx;

namespace nsgpu {

unsigned host_threads();
double now_ms();
void parallel_for_impl(size_t n, const std::function<void(size_t)> &fn);
template <class F> inline void par_for(size_t n, F fn) { parallel_for_impl(n, std::function<void(size_t)>(fn)); }
void parallel_for_pinned_impl(size_t n, const std::function<void(size_t)> &fn);
template <class F> inline void par_for_pinned(size_t n, F fn) { parallel_for_pinned_impl(n, std::function<void(size_t)>(fn)); }
// background tasks of the lowest priority (run when no parallel loop has work) and the wait for all of them
void pool_bind_to_gpu_node(const char *pci_bus_id);   // pool threads (and pool_bind_this_thread callers) stay on the GPU's NUMA node
void pool_bind_this_thread();
void pool_post(std::function<void()> fn);
// debug accounting (NSGPU_CONS_DEBUG): loops submitted by this thread are booked under `name` until restored
int pool_tag(const char *name);
void pool_tag_restore(int t);
void pool_prof_print();
uint64_t pool_thread_cpu_ns();      // this thread's CPU time / the part of it spent inside pool loops
uint64_t pool_thread_work_ns();
struct PoolTag { int prev; explicit PoolTag(const char *n) : prev(pool_tag(n)) {} ~PoolTag() { pool_tag_restore(prev); } };
template <class F> inline void par_for(const char *tag, size_t n, F fn) { PoolTag t(tag); parallel_for_impl(n, std::function<void(size_t)>(fn)); }
template <class F> inline void par_for_pinned(const char *tag, size_t n, F fn) { PoolTag t(tag); parallel_for_pinned_impl(n, std::function<void(size_t)>(fn)); }
void pool_drain();

struct CountSlot { unsigned long long key; uint32_t count, pad; };       // seeds.hip: key = hash + 1 (0: empty)

struct AlignReq {
    mm2::RefIndex *idx;           // index of `ref` (owned by the caller, reusable across batches): at least set_sequence(); the lookup
                                  // table is built on demand for the pairs the GPU seeding hands back (from ref_mz)
    const char *ref; size_t ref_len;
    const char *qry; size_t qry_len;
    const mm2::Anchor *qry_mz = nullptr;   // the query's minimizers when the caller already has them (gpu_mm_sketch): PINNED memory
    size_t n_qry_mz = 0;
    const mm2::Anchor *ref_mz = nullptr;   // the reference's minimizers, PINNED memory as well; both given = seeds on the GPU (seeds.hip)
    size_t n_ref_mz = 0;
    // the reference's occurrence-count table, when the caller keeps one up to date (seeds.hip count_update_kernel); else the seeding launch builds one
    const CountSlot *ref_cnt = nullptr; const uint32_t *ref_cnt_meta = nullptr; uint32_t ref_cnt_bits = 0;
    const mm2::Anchor *ref_mz_dev = nullptr;   // the same list resident in DEVICE memory (the contig engine keeps one per contig): the seeding kernel
                                               // reads this one, ref_mz (any host memory then) serves the pairs the kernel hands back to the host code
    // for the plan kernel (plan.hip): the query and the stretch [ref_dev_lo, ref_dev_lo + ref_dev_n) of the reference as ASCII in DEVICE memory
    // (sketch_dev_seq: they travelled with the sketch batch); both given = the DP problems are planned and launched on the device
    const uint8_t *qry_dev = nullptr, *ref_dev = nullptr;
    uint32_t ref_dev_lo = 0, ref_dev_n = 0;
};
struct SketchReq { const char *ptr; size_t len; };
// (w,k)-minimizers of a batch of sequences, computed on the GPU (mm_sketch.hip).  Sequence i's minimizers are
// out[out_off[i] .. out_off[i+1]) in mm_sketch's order; `out` points into a pinned buffer owned by the context and
// stays valid until the next call.
// `out` points into the pinned buffer of sketch workspace `ws` (0 or 1) and stays valid until the next call with the same workspace;
// calls with different workspaces may run concurrently
// n_stage_only: the last so many requests are copied to HBM with the batch but not sketched (their lists are empty); sketch_dev_seq(c, ws, i) is
// request i's bytes in DEVICE memory (nullptr when the batch did not go through the fused path's staging), valid until the workspace's next call
int gpu_mm_sketch(nsgpu_ctx *c, const std::vector<SketchReq> &reqs, int w, int k, const mm2::Anchor *&out, std::vector<uint64_t> &out_off, int ws = 0, size_t n_stage_only = 0);
const uint8_t *sketch_dev_seq(const nsgpu_ctx *c, int ws, size_t i);
int align_requests(nsgpu_ctx *c, std::vector<AlignReq> &reqs, std::vector<mm2::AlnOut> &outs, int ws_index = 0);
// the same in two parts, the DP kernels in flight between them (state of one batch)
// seeds.hip: index + seeds of a batch of (reference minimizers, query minimizers) pairs on the GPU
struct SeedPair {
    const mm2::Anchor *ref; const mm2::Anchor *qry;      // device-readable (pinned host or device memory), mm_sketch order
    uint32_t n_ref, n_qry;
    // the reference's occurrence-count table (seeds.hip count_update_kernel: a contig keeps one in HBM) and its meta words {distinct hashes,
    // mid_occ, flags}; nullptr: gpu_seeds_launch builds one in scratch for this launch
    const CountSlot *cnt_tab = nullptr; const uint32_t *cnt_meta = nullptr;
    uint32_t cnt_bits = 0; uint32_t pad_ = 0;
    uint64_t tab_off = 0;                                 // filled by gpu_seeds_launch: the pair's slice of the scratch table
};
// seeds.hip: one job of count_update_kernel (pinned array, one workgroup per job) -- rebuild = 1: the table cleared and filled from all[0 .. n_all);
// else the hashes rem[] leave and add[] join (Anchor::x >> 8 each)
struct CountJob {
    CountSlot *tab; uint32_t bits, rebuild; uint32_t *hist; uint32_t *meta;          // hist: 1024 words; meta: {distinct hashes, mid_occ, flags, -}
    const unsigned long long *rem; const unsigned long long *add; const mm2::Anchor *all; uint32_t n_rem, n_add, n_all, pad;
};
int gpu_count_tables_launch(hipStream_t st, const CountJob *jobs_pinned, uint32_t n_jobs, float mid_occ_frac);
uint32_t count_table_bits(uint64_t n_keys);
struct SeedResult { unsigned long long base; uint32_t n; uint32_t flags; int32_t mid_occ; float avg; };
constexpr uint32_t SEED_FLAG_TIES = 1, SEED_FLAG_MANY = 2, SEED_FLAG_CAPACITY = 4, SEED_FLAG_OCC = 8, SEED_FLAG_WIDE = 16;
struct ChainList { uint64_t beg, obeg; uint32_t n; float avg; };

// ---- the alignment plan on the device (plan.hip) and the DP launch that follows it without a host round trip (ksw2.hip) ----
// One entry per alignment of a batch.  ref: the staged span [ref_lo, ref_lo + ref_n) of the reference (consensus) as ASCII in DEVICE memory;
// qry: the whole query likewise.  The alignment may use the task slots [task_base, task_base + task_cap).
struct PlanPair { const uint8_t *ref; const uint8_t *qry; uint32_t ref_lo, ref_n, ref_len, qlen, task_base, task_cap; };
struct PlanOut { uint32_t n_tasks, flags, slow, pad; };        // slow: one of its problems runs in a class that finishes late (ksw_class_is_slow): its results come with the second part
// why an alignment was left to the host's plan: no anchors / pair flagged by the kernels before; several chains; long-gap seed filtering;
// a target window outside the staged span; out of task slots or scratch; a DP problem the register kernels do not serve
constexpr uint32_t PLAN_NONE = 1, PLAN_COMPLEX = 2, PLAN_BADSEEDS = 4, PLAN_SPAN = 8, PLAN_FULL = 16, PLAN_CLASS = 32;
struct PlanKey { int32_t qs, qe, rs, re, w, zdrop, end_bonus, flag; };      // = mm2::DpKey: what the host looks a device-planned problem up by
struct PlanDp {            // where the plan kernel puts its DP tasks (buffers of a DP workspace, ksw2.hip ksw_dev_prepare)
    KswTask *tasks; KswResult *res; uint32_t *class_list; uint32_t *class_cnt; uint32_t n_slots;
    uint32_t *task_pair;                                 // task slot -> alignment (for the DP kernels' own hand-over, ksw_collect.hpp)
    uint32_t class_grid[KSW_REG_CLASSES];               // workgroups the class's launch has: entries of its list beyond that are never run
    uint32_t *pair_done;                                 // alignment -> problems that are complete (ksw_collect.hpp): the plan kernel counts the EMPTY problems it resolves itself
    uint8_t *seqs; unsigned long long *cursors;          // cursors: traceback bytes, CIGAR entries, sequence bytes handed out so far
    unsigned long long p_cap; uint32_t cig_cap, seq_cap;
};
struct PlanCfg {           // minimap2's options the plan depends on (mm2::Opt) + the DP kernels' parameters and class rule
    int32_t k, min_cnt, min_sc, bw, max_gap, min_ksw_len, zdrop, end_bonus, a, q, e, q_max, two_phase, approx_flag_or, ext_flag_or;
    KswParams kp; KswClassCfg kc;
};
int plan_launch(hipStream_t st, uint32_t n_pairs, uint32_t lds_anchors, const SeedResult *seeded, const mm2::Anchor *anchors, const int32_t *f, const int32_t *p,
                const PlanPair *pairs, PlanOut *out, PlanKey *keys_out, const PlanDp &dp, const PlanCfg &cfg);
// ksw2.hip: a DP batch whose tasks are written by the plan kernel.  prepare: buffers for n_slots task slots (results zeroed, cursors and class
// counters reset) on the workspace's stream; launch: every register class over its device-side list, CIGAR compaction, results / CIGARs into
// pinned memory -- all behind `after` (an event on the stream that ran the plan kernel); collect: waits, then res / coff / cig point into the
// pinned results (valid until the workspace's next prepare) and class_cnt[KSW_REG_CLASSES] tells what was launched.
// The results come in two parts: the alignments none of whose problems runs in a late class (part 0: behind the bulk of one-wave problems on
// the main stream), then the rest (part 1: behind everything).  res[slot] / cig + coff[slot] are valid for the task slots of the alignments
// whose status[pair] == 1 (2: the pinned CIGAR arena overflowed -- those alignments are the host's to redo).
// cig_cap / n_slots: how many CIGAR entries / task slots the landing zones hold (nothing beyond is read); epoch: the batch's number on its workspace
// (seeds the hand-over's check word, ksw_collect.hpp)
struct KswDevResults { const KswResult *res; const uint64_t *coff; const uint32_t *cig; const uint32_t *status; const uint32_t *check; uint64_t cig_cap; uint32_t n_slots, epoch; };
int ksw_dev_prepare(nsgpu_ctx *c, int ws_index, uint32_t n_slots, uint32_t n_pairs, uint64_t seq_bytes_bound, hipStream_t st, PlanDp &dp);
int ksw_dev_launch(nsgpu_ctx *c, int ws_index, int max_qlen, const KswParams &pr, hipEvent_t after, const PlanPair *pairs, const PlanOut *outs, uint32_t n_pairs, bool two_phase);
int ksw_dev_collect(nsgpu_ctx *c, int ws_index, int part, KswDevResults &out);

struct AlignBatch {
    std::vector<AlignReq> reqs;
    std::vector<mm2::AlignJob> jobs;
    std::vector<uint32_t> live;
    std::vector<KswTask> tasks;
    std::vector<KswResult> res;
    std::vector<uint32_t> cig;
    std::vector<uint64_t> coff;
    std::vector<size_t> t_off, b_off;
    std::vector<mm2::DpKey> task_keys;       // key of every task of the host-planned round in flight
    size_t nb = 0;
    int ws_index = 0;
    bool in_flight = false;
    bool prestepped = false;             // align_prestep has started every job and run its first step (seeds, chains, DP plan)
    double host_ms = 0, dp_ms = 0, chain_ms = 0;
    std::vector<SeedPair> seed_pairs;      // scratch of the seeding launch (seeds.hip)
    uint64_t dp_tasks = 0, rounds = 0;
    // the device plan of the batch (plan.hip): per seeding pair its PlanPair / PlanOut, per task slot its key (all pinned); per request its first slot
    PinBuf plan_pairs, plan_out, plan_keys;
    std::vector<uint32_t> plan_base, plan_pair;          // request -> first task slot / seeding pair (~0u: none)
    int plan_ws = -1;                                    // DP workspace the device-planned batch runs on; -1: none in flight
    hipEvent_t plan_ev = nullptr;
    hipEvent_t plan_wait_ev = nullptr;                   // set by the caller: the plan kernel's inputs (the device copies of the references) are ready behind it
    bool plan_two_part = false;                          // set by the caller: fetch the device-planned results in two parts (align_finish_early first)
    std::vector<uint8_t> plan_delivered;                 // per request: its device-planned results are in the job's cache
    std::vector<uint8_t> early_done;                     // per request: finished by align_finish_early (its AlnOut is final)
    // Deferred alignments (the contig engine's rule for reads across long repeats, engine.hpp DeferBatch): with defer_anchors > 0 a pair the
    // seeding kernel handed back whose host-seeded list has more anchors than that is NOT chained and aligned with the batch: its request index
    // goes to `deferred`, skip[i] = 1 takes it out of everything that follows (no step, no DP round, no result), and the caller moves its job out
    // (B.jobs[i], seeded, anchors in J.a) before the batch's next use.
    uint32_t defer_anchors = 0;
    std::vector<uint32_t> deferred;
    std::vector<uint8_t> skip;
    AlignBatch() = default;
    AlignBatch(const AlignBatch &) = delete;
    AlignBatch &operator=(const AlignBatch &) = delete;
    ~AlignBatch() { plan_pairs.release(); plan_out.release(); plan_keys.release(); if (plan_ev) (void)hipEventDestroy(plan_ev); }
};
// first host step (seeds / chains / regions / DP plan) of the requests [lo, hi) of B.reqs, ahead of align_begin: lets the caller
// overlap it with the GPU sketch of the batch's other requests.  All requests must have been pre-stepped before align_begin.
int align_prestep(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws);
// the same in two parts: seeds + launch of the chaining kernel / wait + first step -- the caller pipelines ranges (different chain_ws)
// (align_prestep_start + the caller's own loop calling B.jobs[i].seed() + launch(..., true): seeds inside another per-builder loop)
int align_prestep_start(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi);
int align_prestep_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws, bool started_and_seeded = false, int dp_ws = -1);   // dp_ws >= 0: + the device plan and its DP launch
int align_prestep_finish(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws);
// a batch of jobs that were started and seeded by the host code elsewhere (AlignBatch::deferred of another batch, moved in): their chaining scores
// on chaining workspace chain_ws, then align_begin .. align_finish on DP workspace dp_ws.  Safe beside other batches on other workspaces.
int align_seeded_jobs(nsgpu_ctx *c, AlignBatch &B, int chain_ws, int dp_ws, std::vector<mm2::AlnOut> &outs);
// chain.hip: mm_chain_dp's forward pass for a batch of anchor lists (host pointers in, pinned host results out)
int gpu_chain_launch(nsgpu_ctx *c, int ws, const mm2::Opt &opt, const std::vector<const mm2::Anchor *> &lists, const std::vector<uint64_t> &off,
                     const std::vector<float> &avg);
int gpu_chain_wait(nsgpu_ctx *c, int ws, const int32_t *&f, const int32_t *&p);

int gpu_seeds_launch(nsgpu_ctx *c, int ws, float mid_occ_frac, std::vector<SeedPair> &pairs, bool with_total = true);
int gpu_seeds_total(nsgpu_ctx *c, int ws);
int gpu_seeds_wait(nsgpu_ctx *c, int ws, const SeedResult *&res, const mm2::Anchor *&d_anchors);
// seeds + the chaining kernel behind them on one stream (chain_ws: the chaining workspace whose pinned buffer takes anchors / f / p)
int gpu_seeds_chain_launch(nsgpu_ctx *c, int ws, int chain_ws, const mm2::Opt &opt, std::vector<SeedPair> &pairs, bool with_total = true);
// what the launch above left in DEVICE memory (valid on the seeding workspace's stream, behind its kernels): the sorted anchors, the pairs' results
// (pinned, device-visible) and f / p of every anchor -- the inputs of the plan kernel (plan.hip)
struct SeedChainDev { const mm2::Anchor *anchors; const SeedResult *res; const int32_t *f, *p; hipStream_t stream; uint32_t lds_anchors; };
SeedChainDev gpu_seeds_chain_dev(nsgpu_ctx *c, int ws, int chain_ws);
int gpu_seeds_chain_wait(nsgpu_ctx *c, int ws, int chain_ws, const SeedResult *&res, const mm2::Anchor *&a, const int32_t *&f, const int32_t *&p);
int align_begin(nsgpu_ctx *c, AlignBatch &B, int ws_index);
int align_finish(nsgpu_ctx *c, AlignBatch &B, std::vector<mm2::AlnOut> &outs);
// a two-part batch (B.plan_two_part): waits for the first part of the device-planned DP results -- the alignments none of whose problems runs in a
// late kernel class -- and finishes those alignments (skeleton execution, alignRead's conversion into outs[i]); ready[i] = 1 for the requests that
// are final now.  align_finish must follow; it leaves the early ones alone.
int align_finish_early(nsgpu_ctx *c, AlignBatch &B, std::vector<mm2::AlnOut> &outs, std::vector<uint8_t> &ready, int part);
// the same in steps, for callers that run the per-request part inside their own per-builder tasks (consensus_driver.hip engine_early_updates)
int batch_plan_wait(nsgpu_ctx *c, AlignBatch &B, int part, KswDevResults &R);
bool ksw_dev_poll(nsgpu_ctx *c, int ws_index, KswDevResults &out, const volatile uint32_t *&done);      // ksw2.hip
uint32_t batch_plan_deliver_one(AlignBatch &B, const KswDevResults &R, size_t i, int part, bool own_part_only);
bool align_early_one(AlignBatch &B, size_t i, mm2::AlnOut &out);
int filter_strings_device(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq);   // api.hip: results stay in c->f_off / c->f_ids

}  // namespace nsgpu
