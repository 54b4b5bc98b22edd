// common.hpp -- context, error handling and device-buffer helpers shared by the
// translation units of libnsgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <algorithm>
#include "../../include/nsgpu.h"
#include "consensus.hpp"

namespace nsgpu {

void set_error(const char *fmt, ...);

#define NS_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            nsgpu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return NSGPU_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define NS_CHECK(cond, code, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            nsgpu::set_error(__VA_ARGS__);        \
            return (code);                        \
        }                                         \
    } while (0)

#define NS_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != NSGPU_OK) return rc_; \
    } while (0)

// Waits for a stream without holding a core.  hipStreamSynchronize (and hipEventSynchronize, blocking-sync flag or not)
// busy-waits on this runtime: a waiting thread shows 100 % CPU for the whole wait (tools/spin_probe.py), and the contig stage
// is bound by the host cores it shares with those waits.  So: poll the stream, spin only for the first ~20 us, then sleep
// between polls.
hipError_t stream_wait(hipStream_t s);
hipError_t event_wait(hipEvent_t ev);            // the same for a point inside a stream's work
// a non-blocking stream for one of the roles "sketch", "seeds" (seeding + chaining kernels), "dp_side" (long DP problems), "dp" (DP workspaces 1-3):
// priority from NSGPU_PRIO_<ROLE>=lo|mid|hi, else the role's measured default (api.hip)
int role_stream_create(hipStream_t *st, const char *role);
int mirror_finalize(nsgpu_ctx *c, bool force_packed = false);                                      // api.hip: ASCII or packed host copy of the reads
const char *mirror_read(const nsgpu_ctx *c, uint32_t r, std::string &buf);
hipError_t stream_wait_short(hipStream_t s);     // busy-wait (the runtime's): for waits inside a chain of short kernels on a slot's critical path

// Growable device allocation (never shrinks). No hipMalloc happens inside a
// stage once the buffers have reached their steady-state size.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return NSGPU_OK;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return NSGPU_ERR_NOMEM;
        }
        cap = want;
        return NSGPU_OK;
    }
    void release() { if (p) { hipError_t e = hipFree(p); (void)e; } p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Growable pinned host buffer.  Device-to-host copies must land in pinned memory: a hipMemcpyAsync into pageable memory
// (a std::vector) turns into a blocking staged copy inside the runtime, which busy-waits for everything queued before it.
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return NSGPU_OK;
        if (p) { hipError_t e = hipHostFree(p); (void)e; p = nullptr; cap = 0; }
        const size_t want = bytes + (bytes >> 1) + 4096;
        const hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) { set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return NSGPU_ERR_NOMEM; }
        cap = want;
        return NSGPU_OK;
    }
    void release() { if (p) { hipError_t e = hipHostFree(p); (void)e; } p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// A set of 2-bit packed sequences in HBM.  Row r starts at byte poff[r]
// (16-byte aligned, zero padded to the next 16 bytes plus 16 more so that
// kernels may over-read two dwords) and holds len[r] bases, MSB-first.
struct SeqStore {
    DevBuf packed, poff, len;
    uint32_t n = 0;
    uint64_t packed_bytes = 0;
    uint64_t n_bases = 0;
    uint32_t max_len = 0;
    std::vector<uint64_t> h_poff;   // host mirrors (needed by get_read and job planning)
    std::vector<uint32_t> h_len;
    void release() { packed.release(); poff.release(); len.release(); n = 0; }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): launches come from several host threads and, with
// several contexts in one process, for several devices.  One of these per kernel: raise(bytes, kernel) sets the attribute when the
// current device has not seen that many bytes for the kernel yet.
struct LdsAttr {
    std::mutex m;
    size_t cap[64] = {};
    int raise(size_t bytes, const void *kernel)
    {
        int dev = 0;
        NS_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) dev = 0, bytes = bytes > 65536 ? bytes : 65536;
        std::lock_guard<std::mutex> lk(m);
        if (bytes > cap[dev]) {
            NS_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            cap[dev] = bytes;
        }
        return NSGPU_OK;
    }
};

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    int init() { NS_HIP(hipEventCreate(&a)); NS_HIP(hipEventCreate(&b)); return NSGPU_OK; }
    void destroy() { if (a) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } a = b = nullptr; }
};

}  // namespace nsgpu

namespace nsgpu { namespace mm2 { struct Anchor; } }

struct nsgpu_ctx {
    nsgpu_params prm;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int n_cu = 256;
    nsgpu::SeqStore reads;       // the loaded reads
    nsgpu::SeqStore queries;     // scratch store for window/query strings
    nsgpu::DevBuf ascii, aoff;   // staging for ASCII uploads
    nsgpu::DevBuf salts;         // n u64
    bool have_salts = false, have_sketch = false, have_index = false, have_filter_all = false;
    nsgpu::DevBuf sketch;        // N*n u64, row-major by read (fwd)
    nsgpu::DevBuf sketch_rc;     // N*n u64 (whole-read RC queries)
    nsgpu::DevBuf qsketch;       // Q*n u64 (string queries)
    // bucket index: per slot j, keys sorted ascending (idx_keys[j*N+i]) with the
    // read ids in the same order (idx_ids[j*N+i], ascending within equal keys)
    nsgpu::DevBuf idx_keys, idx_ids, idx_tmp_k, idx_tmp_v, idx_tmp_e, idx_tmp_e2, idx_sort_ws;
    // filter results
    nsgpu::DevBuf f_pool, f_qstart, f_qcnt, f_qm, f_off, f_ids, f_ctrl, f_ovf_list, f_ovf_cnt, f_scan_ws;
    size_t f_pool_cap = 0;       // ids
    uint64_t f_total = 0;        // candidates of the last filter call
    bool filter_stats = true;    // also total the matched list lengths (timing.filter_matches): one more scan + read-back per call; the contig engine switches it off
    uint32_t f_nq = 0;
    nsgpu::DevBuf rep_flags;
    nsgpu::DevBuf fq_cnt, fq_base, fq_nlpos, fq_len;   // FASTQ ingest (fastq.hip)
    double fastq_ms = 0;
    // ksw2 batches
    // two workspaces: the contig engine aligns two half batches from two host threads, so that one half's host work
    // (seeding, chaining, CIGAR bookkeeping) overlaps the other half's DP kernels
    struct KswWs {
        nsgpu::DevBuf k_tasks, k_order, k_seqs, k_p, k_cig, k_res, k_slab, k_ncig, k_coff, k_cig2, scan_ws;
        std::vector<hipEvent_t> ev;                                  // start/end event pairs, one pair per launch of a batch
        std::vector<int> ev_class;                                   // register class of each pair (-1: another kernel)
        uint8_t *h_pool = nullptr; size_t h_pool_cap = 0;            // pinned staging of the DP sequence pool
        std::vector<uint8_t> h_bucket; std::vector<uint32_t> h_tmp;   // scratch of the launch-order bucketing
        std::vector<uint32_t> h_flat;                                // launch order of the batch in flight
        std::vector<uint32_t> h_pbytes; std::vector<uint8_t> h_class;  // per-problem traceback bytes / kernel class (scratch of the launch)
        nsgpu::PinBuf h_res, h_coff, h_cig;                            // pinned landing zones of the results, CIGAR offsets and CIGARs
        size_t pend_n = 0, pend_n_ev = 0; uint64_t pend_n_launch = 0; // batch launched, not yet collected (ksw_batch_launch / _collect)
        hipStream_t stream = nullptr;                                // workspace 0 runs on the context's stream
        hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
        hipEvent_t side_done[3] = {nullptr, nullptr, nullptr}, side_fork = nullptr, t_a = nullptr, t_b = nullptr;
        hipStream_t bulk_stream = nullptr; hipEvent_t bulk_done = nullptr;   // a two-part device batch: the <1,4> bulk beside the <1,2> bulk
        // a batch whose tasks the plan kernel writes (plan.hip, ksw_dev_*): buffers of its own, so that a host-planned batch of the same
        // workspace may follow in the same slot.  dv_ctrl: [0..15] class counters, [16..17] overflow flags, then 8 u64: cursors (traceback bytes,
        // CIGAR entries, sequence bytes), cells, algorithmic bytes
        nsgpu::DevBuf dv_tasks, dv_list, dv_ctrl, dv_seqs, dv_p, dv_cig, dv_res, dv_coff, dv_scan_ws, dv_tpair, dv_pdone;
        bool dv_inline = false;                  // this batch's DP kernels hand every alignment over themselves (ksw_collect.hpp)
        nsgpu::PinBuf hv_res, hv_coff, hv_cig, hv_ctrl, hv_status, hv_check;
        hipEvent_t dv_part0 = nullptr;                                  // behind the first part of a two-part batch's results
        uint32_t dv_npairs_launched = 0; bool dv_two_phase = false;
        uint32_t dv_epoch = 0;                   // device batches launched on this workspace so far (seed of the hand-over's check words)
        uint32_t dv_slots = 0, dv_pairs = 0, dv_classes = 0;                         // dv_classes: bit k = class k was launched
        uint64_t dv_p_hint = 0, dv_hcig_hint = 0, dv_hcig_cap = 0;
        bool dv_pending = false;
        std::vector<hipEvent_t> dv_ev;
        hipEvent_t dv_a = nullptr, dv_b = nullptr, dv_clear_ev = nullptr;
    } kws[8];                                                       // 0: the context's stream (direct API calls); 1-3: own streams (contig engine, one per batch in flight)
    // batched minimizer sketches (mm_sketch.hip): device buffers + pinned staging both ways
    struct SketchWs {
        nsgpu::DevBuf seqs, soff, len, sob, vf, mk, vr, linv, npf, pushf, npr, pr, V, hk, PX, PY, PRUN, PSEQ, rm, nout, oscan, off, out, scan_ws;
        uint8_t *h_seqs = nullptr; size_t h_cap = 0;
        uint8_t *h_out = nullptr; size_t h_out_cap = 0;
        nsgpu::PinBuf h_meta;                                          // pinned landing zone of the small read-backs (push count, offsets)
        nsgpu::PinBuf h_concat;                                        // results of an oversize batch sketched piece by piece (gpu_mm_sketch)
        nsgpu::PinBuf h_compact; std::vector<uint64_t> h_toff;         // the fused path's tile slots laid out back to back / the tiles' offsets there
        const uint32_t *staged_soff = nullptr; size_t staged_n = 0;   // offsets of the last batch's requests in `seqs` (sketch_dev_seq)
        std::vector<uint32_t> staged_soff_v;
        hipStream_t stream = nullptr;
    } sws[2];                                                       // two workspaces: the contig engine sketches the two halves of a batch concurrently
    // chaining scores (chain.hip): anchors in, f / p out.  0: direct API calls; 1..: two per group of the contig engine (the halves of a batch are pipelined)
    struct ChainWs {
        nsgpu::DevBuf d_in, d_out, d_marks;
        nsgpu::PinBuf h_in, h_out;
        std::vector<uint8_t> h_fast;                                   // per list: the LDS kernel takes it
        size_t lds_set = 0;
        uint64_t pend_total = 0;                                       // anchors of the launch in flight
        std::vector<const nsgpu::mm2::Anchor *> lists; std::vector<uint64_t> off; std::vector<float> avg;   // the launch's lists (host side)
        double ms_stage = 0, ms_enqueue = 0, ms_wait = 0; uint64_t calls = 0;   // host wall of the calls: staging / enqueue / wait for the results
        hipStream_t stream = nullptr;
        hipStream_t stream2 = nullptr; bool ring_used = false;         // the ring kernel's long lists run beside the LDS kernel's launch
        uint32_t seeded_lds_anchors = 0; uint64_t seeded_capacity = 0; // of the last seeded launch (d_out then holds f / p for the plan kernel)
    } cws[26];                                                      // per batch workspace w: 2w the lists seeded on the GPU, 2w + 1 the ones seeded by the host code
    // index + seeds (seeds.hip): scratch tables, anchors (device), pair descriptors and results (pinned)
    struct SeedWs {
        nsgpu::DevBuf d_tab, d_next, d_ys, d_tmp, d_out, d_counter;
        nsgpu::PinBuf h_pairs, h_res, h_ref, h_jobs;                  // h_ref: staging of reference minimizer lists that live in pageable memory; h_jobs: count-table jobs
        size_t pend = 0; uint64_t capacity = 0, cap_hint = 0;
        std::vector<uint32_t> pair_of, fb, late;                       // job -> pair (~0u: host-seeded), the host-seeded jobs, the jobs the kernels handed back
        const void *res = nullptr;                                    // results of the launch in flight (SeedResult[])
        double ms_wait = 0; uint64_t calls = 0, pairs = 0, fallbacks = 0;
        // (debug print) the pairs handed back: by flag bit, their anchors in sum and the longest list, wall-ms of the host seeding and of their chaining launch
        uint64_t late_flag[5] = {}, late_anchors = 0, late_longest = 0, late_calls = 0; double late_seed_ms = 0, late_chain_ms = 0;
        hipStream_t stream = nullptr;
    } seed_ws[9];
    std::vector<uint64_t> wq_off; std::vector<uint32_t> wq_ids;      // the last fused window-query batch's candidate lists (run_window_queries_fast)
    nsgpu::PinBuf pin_wq, pin_wq_out;                                // the engine's window queries: staging of the strings / candidate lists as the last kernel writes them
    nsgpu::PinBuf pin_small, pin_foff, pin_fids;                     // pinned landing zones: the filter's scalars / the engine's candidate CSR
    double sketch_mm_ms = 0;                                         // wall of the batched mm_sketch calls
    std::mutex stat_m;                                               // guards the ksw_* / aln_* counters below
    double ksw_kernel_ms = 0, ksw_cells = 0, ksw_alg_bytes = 0;     // kernel_ms: wall of the (overlapping) DP launches per batch
    double ksw_kernel_sum_ms = 0;                                    // sum of the individual kernel durations (what rocprof reports)
    uint64_t ksw_launches = 0;
    double ksw_class_ms[16] = {}; uint64_t ksw_class_n[16] = {};    // per register class: summed launch durations / launches (debug print of the contig stage)
    // align batches
    uint64_t aln_pairs = 0, aln_dp_tasks = 0, aln_rounds = 0, aln_seed_gpu = 0, aln_seed_host = 0;
    uint64_t plan_pairs_dev = 0, plan_pairs_host = 0, plan_hits = 0, plan_misses = 0, plan_extra = 0;      // the alignment plan on the device (plan.hip)
    uint64_t plan_why[8] = {};                                     // alignments left to the host, by PLAN_* bit (debug print)
    double aln_index_ms = 0, aln_host_ms = 0, aln_dp_ms = 0;
    // host copy of the reads as ReadData::getRead returns them (A/T/C/G after the 2-bit folding)
    std::vector<char> h_bases;
    std::vector<uint64_t> h_off;
    // ... or, for large inputs (NSGPU_PACKED_MIRROR, api.hip mirror_finalize), the 2-bit rows as they lie in HBM (0.25 B/base): the
    // engine decodes a read when it copies it anyway (seed read, candidate, emission walk)
    std::vector<uint8_t> h_packed;
    bool packed_mirror = false;
    // consensus run
    bool have_cons = false;
    uint32_t read_id_base = 0;   // global id of local read 0 (multi-GPU shards)
    // schedule of the contig stage (nsgpu_set_schedule): pipeline groups (1, 2 or 4) and the conflict-aware seed rule (0 = off)
    uint32_t sched_groups = 4, seed_bucket_depth = 0, seed_rings = 1, seed_tail_rings = 1;
    bool defer_set = false; uint64_t cons_n_deferred = 0;            // (cons_n_deferred: alignments deferred in the last contig stage)
    uint32_t defer_anchors = 0, defer_slots = 0;     // nsgpu_set_defer (the automatic schedule: 4096 / 2): alignments with longer anchor lists take defer_slots more slots
    bool sched_auto = false, sched_set = false;      // nsgpu_set_schedule_auto / an explicit nsgpu_set_schedule: 0 builders with neither = the automatic schedule
    void *cons_engine = nullptr;                 // resumable contig engine (consensus_driver.hip)
    void (*cons_engine_free)(void *) = nullptr;
    uint32_t graph_mode = 0; bool graph_mode_set = false;          // nsgpu_set_graph
    uint32_t graph_used = 0;                     // what the last contig stage used (NSGPU_GRAPH_HOST / _DEVICE)
    void *graph_shared = nullptr;                // pools and streams of the consensus graphs in HBM (graph_dev.hpp), made on first use
    void (*graph_shared_free)(void *) = nullptr;
    uint64_t cons_n_reads_out = 0;               // reads covered by this context's output streams
    nsgpu_consensus_stats cons_stats;
    std::vector<nsgpu::cons::StreamSet> cons_out;
    nsgpu::Timer t_stage, t_kernel;
    nsgpu_timing timing;
};

namespace nsgpu {

// kernels_minhash.hip
int launch_pack_ascii(nsgpu_ctx *c, const char *d_ascii, const uint64_t *d_aoff, SeqStore &st);
int launch_sketch(nsgpu_ctx *c, const SeqStore &st, uint64_t *d_out_fwd, uint64_t *d_out_rc);
int launch_sketch_range(nsgpu_ctx *c, const SeqStore &st, uint32_t lo, uint32_t hi, uint64_t *d_out_fwd);   // rows lo..hi only
int launch_repetitive(nsgpu_ctx *c, const SeqStore &st, uint8_t *d_flags);
int run_filter(nsgpu_ctx *c, const uint64_t *d_q_even, const uint64_t *d_q_odd, uint32_t nq, bool interleave);
int run_window_queries_fast(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq, const uint64_t *&off_out, const uint32_t *&ids_out, bool *redo);
// index.hip
int build_index(nsgpu_ctx *c);
int scan_u32_to_u64(nsgpu_ctx *c, const uint32_t *d_in, uint64_t *d_out, uint32_t n);  // exclusive, n+1 outputs
int scan_u32_to_u64(DevBuf &scratch, hipStream_t stream, const uint32_t *d_in, uint64_t *d_out, uint32_t n);

int store_layout(SeqStore &st, const uint32_t *len, uint32_t n);   // fills h_poff/h_len, allocs
}  // namespace nsgpu
