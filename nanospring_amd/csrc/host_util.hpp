// host_util.hpp -- host thread pool helpers and the internal align-request interface.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>
#include "mm2.hpp"
#include "ksw2.hpp"

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

struct AlignReq {
    mm2::RefIndex *idx;           // index of `ref` (owned by the caller, reusable across batches): at least set_sequence(); the lookup
                                  // table is built on demand for the pairs the GPU seeding hands back (from ref_mz)
    const char *ref; size_t ref_len;
    const char *qry; size_t qry_len;
    const mm2::Anchor *qry_mz = nullptr;   // the query's minimizers when the caller already has them (gpu_mm_sketch): PINNED memory
    size_t n_qry_mz = 0;
    const mm2::Anchor *ref_mz = nullptr;   // the reference's minimizers, PINNED memory as well; both given = seeds on the GPU (seeds.hip)
    size_t n_ref_mz = 0;
    const mm2::Anchor *ref_mz_dev = nullptr;   // the same list resident in DEVICE memory (the contig engine keeps one per contig): the seeding kernel
                                               // reads this one, ref_mz (any host memory then) serves the pairs the kernel hands back to the host code
};
struct SketchReq { const char *ptr; size_t len; };
// (w,k)-minimizers of a batch of sequences, computed on the GPU (mm_sketch.hip).  Sequence i's minimizers are
// out[out_off[i] .. out_off[i+1]) in mm_sketch's order; `out` points into a pinned buffer owned by the context and
// stays valid until the next call.
// `out` points into the pinned buffer of sketch workspace `ws` (0 or 1) and stays valid until the next call with the same workspace;
// calls with different workspaces may run concurrently
int gpu_mm_sketch(nsgpu_ctx *c, const std::vector<SketchReq> &reqs, int w, int k, const mm2::Anchor *&out, std::vector<uint64_t> &out_off, int ws = 0);
int align_requests(nsgpu_ctx *c, std::vector<AlignReq> &reqs, std::vector<mm2::AlnOut> &outs, int ws_index = 0);
// the same in two parts, the DP kernels in flight between them (state of one batch)
// seeds.hip: index + seeds of a batch of (reference minimizers, query minimizers) pairs on the GPU
struct SeedPair {
    const mm2::Anchor *ref; const mm2::Anchor *qry;      // device-readable (pinned host or device memory), mm_sketch order
    uint32_t n_ref, n_qry;
    uint32_t tab_bits; uint32_t pad_ = 0;                 // filled by gpu_seeds_launch: the pair's slice of the scratch buffers
    uint64_t tab_off, next_off;
};
struct SeedResult { unsigned long long base; uint32_t n; uint32_t flags; int32_t mid_occ; float avg; };
constexpr uint32_t SEED_FLAG_TIES = 1, SEED_FLAG_MANY = 2, SEED_FLAG_CAPACITY = 4, SEED_FLAG_OCC = 8, SEED_FLAG_WIDE = 16;
struct ChainList { uint64_t beg, obeg; uint32_t n; float avg; };

struct AlignBatch {
    std::vector<AlignReq> reqs;
    std::vector<mm2::AlignJob> jobs;
    std::vector<uint32_t> live;
    std::vector<KswTask> tasks;
    std::vector<KswResult> res;
    std::vector<uint32_t> cig;
    std::vector<uint64_t> coff;
    std::vector<size_t> t_off, b_off;
    size_t nb = 0;
    int ws_index = 0;
    bool in_flight = false;
    bool prestepped = false;             // align_prestep has started every job and run its first step (seeds, chains, DP plan)
    double host_ms = 0, dp_ms = 0, chain_ms = 0;
    std::vector<SeedPair> seed_pairs;      // scratch of the seeding launch (seeds.hip)
    uint64_t dp_tasks = 0, rounds = 0;
};
// first host step (seeds / chains / regions / DP plan) of the requests [lo, hi) of B.reqs, ahead of align_begin: lets the caller
// overlap it with the GPU sketch of the batch's other requests.  All requests must have been pre-stepped before align_begin.
int align_prestep(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws);
// the same in two parts: seeds + launch of the chaining kernel / wait + first step -- the caller pipelines ranges (different chain_ws)
// (align_prestep_start + the caller's own loop calling B.jobs[i].seed() + launch(..., true): seeds inside another per-builder loop)
int align_prestep_start(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi);
int align_prestep_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws, bool started_and_seeded = false);
int align_prestep_finish(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws);
// chain.hip: mm_chain_dp's forward pass for a batch of anchor lists (host pointers in, pinned host results out)
int gpu_chain_launch(nsgpu_ctx *c, int ws, const mm2::Opt &opt, const std::vector<const mm2::Anchor *> &lists, const std::vector<uint64_t> &off,
                     const std::vector<float> &avg);
int gpu_chain_wait(nsgpu_ctx *c, int ws, const int32_t *&f, const int32_t *&p);

int gpu_seeds_launch(nsgpu_ctx *c, int ws, float mid_occ_frac, std::vector<SeedPair> &pairs);
int gpu_seeds_wait(nsgpu_ctx *c, int ws, const SeedResult *&res, const mm2::Anchor *&d_anchors);
// seeds + the chaining kernel behind them on one stream (chain_ws: the chaining workspace whose pinned buffer takes anchors / f / p)
int gpu_seeds_chain_launch(nsgpu_ctx *c, int ws, int chain_ws, const mm2::Opt &opt, std::vector<SeedPair> &pairs);
int gpu_seeds_chain_wait(nsgpu_ctx *c, int ws, int chain_ws, const SeedResult *&res, const mm2::Anchor *&a, const int32_t *&f, const int32_t *&p);
int align_begin(nsgpu_ctx *c, AlignBatch &B, int ws_index);
int align_finish(nsgpu_ctx *c, AlignBatch &B, std::vector<mm2::AlnOut> &outs);
int filter_strings_device(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq);   // api.hip: results stay in c->f_off / c->f_ids

}  // namespace nsgpu
