// engine.hpp -- the contig engine's state (consensus_driver.hip: the engine's phases; consensus_schedule.hip: seeds, claims, the automatic
// schedule; consensus_dist.hip: the same loop over the ranks of a communicator; consensus_debug.hip: the NSGPU_CONS_DEBUG reports).
#pragma once
#include "common.hpp"
#include <dirent.h>
#include <unistd.h>
#include "consensus.hpp"
#include "graph_dev.hpp"
#include "host_util.hpp"
#include "dist.hpp"
#include <memory>
#include <atomic>
#include <thread>
#include <sched.h>
#include <pthread.h>
#include <sys/resource.h>
#include <sys/prctl.h>
#include <time.h>

namespace nsgpu {


// A finished contig waiting for its edit emission (consensus + one edit script per read into the seven streams).  The
// emission touches nothing but the contig's own graph, so it is taken off the builder's critical path: the builder moves
// on to its next contig at once and the emission runs as a task of its own in the next host phase.
constexpr int kMaxGroups = 4;
// Pipeline groups (see run_consensus): host phase | batches part 1 (sketches + index, seeds / chains / DP launch) | alignment
// DP in flight | batches part 2.  (Measured and dropped: part 1 as two pipeline stages, a fifth group -- DESIGN.md.)
static inline int n_groups(const nsgpu_ctx *c) { return (int)c->sched_groups; }

struct FinishedContig {
    std::unique_ptr<GraphBase> g;             // null once emitted
    cons::StreamSet out;
    double write_ms = 0, free_ms = 0;
};

struct Builder {
    // DEFERRED: its alignment was taken out of its slot's batch (a read across a long repeat, Engine::DeferBatch) and is delivered at the end of
    // slot defer_due; until then the builder takes part in nothing
    enum State { NEED_CONTIG, ADVANCE, WAIT_FILTER, WAIT_ALIGN, GOT_FILTER, ALIGNED, GOT_ALIGN, DONE, DEFERRED };
    State st = NEED_CONTIG;
    uint32_t defer_due = 0;
    uint32_t id = 0, gid = 0;                 // local index / global builder id
    int group = 0;                            // pipeline group (a function of gid only, so that it does not depend on the rank count)
    std::unique_ptr<GraphBase> g;             // the contig's consensus DAG: in HBM (DevGraph) or, NSGPU_GRAPH=host, the pointer graph on the host
    bool graph_flying = false;                // an update of the graph is in flight on the GPU (engine_early_updates)
    read_t cursor = 0;
    // contig walk (src/Consensus.cpp:51-95)
    ssize_t init_start = 0, len = 0, cur_pos = 0;
    bool right_phase = true, edges_too_many = false, window_open = false;
    // window
    std::string win[2];
    std::vector<read_t> cand[2];
    int strand = 0;
    size_t ci = 0;
    bool strand_counted = false;
    // pending alignment
    read_t pend = 0;
    std::string query;
    mm2::AlnOut aln;
    bool accepted = false;
    bool early_updated = false, early_result = false;      // the graph was updated / the result taken over ahead of the slot's end (engine_early_updates)
    // cached index of the current main path
    mm2::RefIndex idx;
    bool idx_valid = false;
    bool sp_ready = false;               // plan_splice has run for the consensus as it is now (right behind the update, on the thread that made it)
    // incremental consensus sketch: the minimizers of mz_str (the main path they were computed for).  When the path changes only the
    // stretch that differs (+ a margin on both sides) is sketched again and spliced in -- see engine_batches_sketch
    std::vector<mm2::Anchor> mz;
    std::string mz_str;
    struct Splice { bool full = true, have_common = false; size_t a = 0, B_sub = 0, A = 0, B = 0, P = 0, S = 0, cp = 0, cs = 0; ssize_t delta = 0; } sp;      // P / S: common prefix / suffix with the string before
    // the contig's consensus resident in HBM (what the plan kernel gathers the DP targets from): main_path as of the last alignment batch the
    // builder was in, at [dc_beg, dc_beg + dc_len) of d_cons -- room on both sides, a contig grows at its ends (cons_update_kernel)
    DevBuf d_cons;
    size_t dc_beg = 0, dc_len = 0;
    bool dc_valid = false;
    // the contig's minimizer list resident in HBM (what the seeding kernel reads): mz[0 .. d_mz_n) as of the last upload.  After a splice only
    // the entries from the first changed one on travel (a contig grows at its ends: a few hundred entries of tens of thousands).
    DevBuf d_mz;
    size_t d_mz_n = 0;
    // the occurrence counts of the list's hashes resident in HBM (seeds.hip count_update_kernel: what mm_idx_str's buckets are reduced to --
    // mid_occ and "does the consensus have this hash fewer than mid_occ times"): kept up to date with what a splice removed and added
    DevBuf d_cnt, d_cnt_hm;                    // the table; its histogram (1024 words) + meta (4 words)
    uint32_t cnt_bits = 0;
    uint64_t cnt_keys = 0;                     // keys inserted since the table was built (an upper bound of its load)
    bool cnt_valid = false;
    std::vector<uint64_t> cnt_rem, cnt_add;    // what a splice changed: the hashes that left the list and the ones that came (a hash on both sides cancelled)
    size_t chg_lb = 0;                         // the main path agrees with mz_str (and idx's base codes) on [0, chg_lb): from ContigGraph::path_changed_from
    std::vector<std::unique_ptr<FinishedContig>> contigs;     // in the order they were finished = output order
    size_t n_queued = 0;                                      // contigs[0 .. n_queued) were handed to the emission queue
    uint64_t n_minhash = 0, n_minhash_new = 0, n_aligner = 0, n_align_calls = 0, n_contigs = 0, n_lone = 0;
    double last_u = 0, last_m = 0, apply_t0 = 0;
    double cpu_ms = 0, max_ms = 0, last_ms = 0, write_ms = 0, dbg_w1 = 0, dbg_w2 = 0, dbg_w3 = 0, dbg_u = 0, dbg_m = 0, dbg_cyc = 0, dbg_init = 0, dbg_rc = 0, dbg_win = 0, dbg_start = 0, dbg_max_u = 0, dbg_max_m = 0, dbg_long_ms = 0;
    uint64_t dbg_long_n = 0;
    uint64_t dbg_c[6] = {0, 0, 0, 0, 0, 0};
    uint64_t dbg_g[2] = {0, 0};               // graph kernels: splits, excursions taken one at a time
};

struct Driver {
    nsgpu_ctx *c;
    uint32_t N, id_base = 0;
    uint64_t edge_thr;
    size_t offset;                     // avgReadLen / 4 (src/Consensus.cpp:54)
    std::vector<uint8_t> in_graph, rep;
    std::vector<Builder> B;
    DevGraphShared *gsh = nullptr;     // pools and streams of the graphs in HBM (lives with the context)
    std::atomic<int> graph_rc{NSGPU_OK};     // first failure of a graph update (checked between the engine's phases)
    std::string graph_err;
    std::mutex graph_err_m;
    void graph_failed(int rc) { std::lock_guard<std::mutex> lk(graph_err_m); if (graph_rc.load() == NSGPU_OK) { graph_err = nsgpu_last_error(); graph_rc.store(rc); } }

    // read r as ReadData::getRead returns it; with the packed host mirror decoded into a per-thread buffer (valid until the thread's next call)
    const char *read_ptr(read_t r) const { static thread_local std::string buf; return mirror_read(c, r, buf); }
    size_t read_len(read_t r) const { return (size_t)(c->h_off[r + 1] - c->h_off[r]); }

    // createGraph (src/Consensus.cpp:388-403) for the seed read r the builder was granted
    void start_contig(Builder &b, read_t r)
    {
        const double s0 = now_ms();
        start_contig_inner(b, r);
        b.dbg_start += now_ms() - s0;
    }
    void start_contig_inner(Builder &b, read_t r)
    {
        if (gsh) b.g.reset(new DevGraph(gsh, b.id)); else b.g.reset(new HostGraph());
        b.g->path_mut().assign(read_ptr(r), read_len(r));
        b.g->set_span(0, (ssize_t)read_len(r));
        b.g->first_read = r + id_base;       // graph / stream ids are global, array indices local
        b.cursor = r + 1;
        b.init_start = 0;
        b.len = b.g->end_pos() - b.g->start_pos();
        b.cur_pos = b.g->start_pos();
        b.graph_flying = false;
        b.right_phase = true, b.edges_too_many = false, b.window_open = false;
        b.idx_valid = false, b.sp_ready = false;
        b.mz.clear(), b.mz_str.clear();          // a new consensus: nothing to splice into
        b.chg_lb = 0;
        b.d_mz_n = 0;                            // (the resident list's memory stays with the builder)
        b.cnt_valid = false;
        b.dc_valid = false;
        b.st = Builder::ADVANCE;
    }

    void finish_contig(Builder &b)
    {
        GraphBase &g = *b.g;
        std::unique_ptr<FinishedContig> fc(new FinishedContig());
        if (g.num_reads() == 0) {
            g.write_read_lone(fc->out);
            fc->out.lone_ids.push_back(g.first_read);
            fc->out.reads_in_contig.push_back(1);
            ++b.n_lone;
            b.g.reset();
        } else {
            b.dbg_cyc += g.dbg_cycles_ms;
            for (int i = 0; i < 6; ++i) b.dbg_c[i] += g.dbg[i];
            b.dbg_g[0] += g.dbg[6], b.dbg_g[1] += g.dbg[7];
            // (a graph in HBM: its arrays start their way back to the host now; the emission task waits for them)
            const int rc = g.emit_begin();
            if (rc != NSGPU_OK) graph_failed(rc);
            fc->g = std::move(b.g);
        }
        b.contigs.push_back(std::move(fc));
        ++b.n_contigs;
        b.st = Builder::NEED_CONTIG;
    }
    // edit emission of one finished contig (ConsensusGraph::writeMainPath + writeReads, src/ConsensusGraph.cpp:979-1012)
    void emit_contig(FinishedContig &fc)
    {
        if (!fc.g) return;
        const double t0 = now_ms();
        GraphBase &g = *fc.g;
        g.write_main_path(fc.out);
        const std::function<cons::ReadBases(cons::read_t)> src = [this](cons::read_t id) {
            const read_t r = id - id_base;
            return cons::ReadBases{read_ptr(r), read_len(r)};
        };
        g.write_reads(fc.out, &src);
        fc.out.reads_in_contig.push_back((read_t)g.num_reads());
        const double t1 = now_ms();
        fc.g.reset();
        fc.write_ms = t1 - t0, fc.free_ms = now_ms() - t1;
    }

    // opens the window at cur_pos (addRelatedReads prologue, src/Consensus.cpp:168-184); false = nothing to query
    bool open_window(Builder &b)
    {
        const double w0 = now_ms();
        const bool r = open_window_inner(b);
        b.dbg_win += now_ms() - w0;
        return r;
    }
    bool open_window_inner(Builder &b)
    {
        GraphBase &g = *b.g;
        const std::string &mp = g.path();
        const ssize_t off = b.cur_pos - g.start_pos();
        if (b.len == 0 || off < 0 || off >= (ssize_t)mp.size()) return false;
        const size_t n = (ssize_t)mp.size() >= off + b.len ? (size_t)b.len : mp.size() - (size_t)off;
        b.win[0].assign(mp, (size_t)off, n);
        cons::reverse_complement(b.win[0], b.win[1]);
        b.strand = 0, b.ci = 0, b.strand_counted = false;
        b.window_open = true;
        b.st = Builder::WAIT_FILTER;
        return true;
    }

    // the two while loops of generateAndWriteConsensus as a resumable walk; returns when a window was
    // opened (state WAIT_FILTER) or the contig is finished (state NEED_CONTIG)
    void walk(Builder &b, bool window_just_done)
    {
        GraphBase &g = *b.g;
        const bool usable = b.len >= 32 && !rep[g.first_read - id_base];
        for (;;) {
            if (b.right_phase) {
                if (window_just_done) {
                    b.cur_pos += (ssize_t)offset;
                    window_just_done = false;
                    if (b.cur_pos + b.len > g.end_pos()) b.right_phase = false;
                    else if (g.num_edges() >= edge_thr) b.edges_too_many = true, b.right_phase = false;
                    if (!b.right_phase) { b.cur_pos = b.init_start - (ssize_t)offset; continue; }
                }
                if (!usable) { b.right_phase = false; b.cur_pos = b.init_start - (ssize_t)offset; continue; }
                if (open_window(b)) return;
                window_just_done = true;         // addRelatedReads returned immediately
            } else {
                if (window_just_done) { b.cur_pos -= (ssize_t)offset; window_just_done = false; }
                if (!(usable && !b.edges_too_many)) break;
                if (b.cur_pos < g.start_pos()) break;
                if (g.num_edges() >= edge_thr) { b.edges_too_many = true; break; }
                if (open_window(b)) return;
                window_just_done = true;
            }
        }
        finish_contig(b);
    }

    // candidate loop of addRelatedReads (src/Consensus.cpp:185-246) up to the next alignment request
    void next_candidate(Builder &b)
    {
        GraphBase &g = *b.g;
        for (; b.strand < 2; ++b.strand, b.ci = 0, b.strand_counted = false) {
            if (!b.strand_counted) { b.n_minhash += b.cand[b.strand].size(); b.strand_counted = true; }
            for (; b.ci < b.cand[b.strand].size(); ++b.ci) {
                const read_t r = b.cand[b.strand][b.ci];
                if (g.num_edges() >= edge_thr) { b.window_open = false; walk(b, true); return; }   // `return` out of addRelatedReads
                if (rep[r]) continue;
                if (in_graph[r]) continue;
                ++b.n_minhash_new;
                if (read_len(r) < 32) continue;
                const double r0 = now_ms();
                if (b.strand) cons::reverse_complement(read_ptr(r), read_len(r), b.query);
                else b.query.assign(read_ptr(r), read_len(r));
                b.dbg_rc += now_ms() - r0;
                b.pend = r;
                b.st = Builder::WAIT_ALIGN;
                return;
            }
        }
        b.window_open = false;
        walk(b, true);
    }

    // parallel phase: consume what the last round delivered and run to the next request
    void advance(Builder &b)
    {
        if (b.st != Builder::ADVANCE && b.st != Builder::GOT_FILTER && b.st != Builder::GOT_ALIGN) return;
        const double t0 = now_ms();
        advance_inner(b);
        const double dt = now_ms() - t0;
        b.cpu_ms += dt;
        b.last_ms = dt;
        if (dt > b.max_ms) b.max_ms = dt;
        if (dt > 3.0) ++b.dbg_long_n, b.dbg_long_ms += dt;
    }
    void advance_inner(Builder &b)
    {
        if (b.st == Builder::ADVANCE) walk(b, false);
        else if (b.st == Builder::GOT_FILTER) next_candidate(b);
        else if (b.st == Builder::GOT_ALIGN) {
            if (b.accepted) {
                if (!b.early_updated) apply_alignment(b);
                b.early_updated = false;
                b.accepted = false;
            }
            ++b.ci;
            next_candidate(b);
        }
    }
    // the accepted read into the contig's graph, the new consensus (src/Consensus.cpp:319-331).  Touches nothing but the builder's own graph:
    // the engine may run it as soon as the alignment is there and its claim cannot fail (engine_early_updates), ahead of the host phase.
    // In two halves: the graph in HBM takes the read (one kernel on a stream of its own) and reports when it is done.
    void apply_submit(Builder &b)
    {
        b.apply_t0 = now_ms();
        const int rc = b.g->submit(b.query, b.aln, b.pend + id_base, b.strand == 1);
        if (rc != NSGPU_OK) graph_failed(rc);
        b.graph_flying = rc == NSGPU_OK;
    }
    void apply_complete(Builder &b)
    {
        GraphBase &g = *b.g;
        if (b.graph_flying) {
            const int rc = g.complete();
            if (rc != NSGPU_OK) graph_failed(rc);
            b.graph_flying = false;
        }
        if (g.path_changed_from < b.chg_lb) b.chg_lb = g.path_changed_from;
        g.path_changed_from = (size_t)-1;
        const double dt = now_ms() - b.apply_t0;
        b.dbg_u += dt, b.last_u = dt;
        if (dt > b.dbg_max_u) b.dbg_max_u = dt;
        b.idx_valid = false, b.sp_ready = false;
    }
    void apply_alignment(Builder &b) { apply_submit(b); apply_complete(b); }
};

// ---------------------------------------------------------------------------
// The engine as resumable phases.  A single process (nsgpu_consensus_run) and a multi-GPU job
// (one process per GPU, nanospring_amd/dist.py) run the SAME phases; in the multi-GPU job every rank
// holds all reads and the whole bucket index (replicated by all-gather), owns the builders with
// gid % world == rank, and the two kinds of claims are resolved on a replicated in_graph[] from
// all-gathered request lists, strictly in global builder order -- so the result does not depend on
// the number of ranks.
// ---------------------------------------------------------------------------
// lists of a batch: per builder the tail that changed goes from the pinned staging buffer into the contig's resident list
struct TailCopy { const mm2::Anchor *src; mm2::Anchor *dst; uint32_t n; uint32_t pad; };
// cons_update_kernel's job: see there
struct ConsJob { uint8_t *buf; const uint8_t *mid; uint64_t beg_old, len_old, beg_new, len_new, P, S; uint32_t full, pad; };


struct Engine {
    Driver D;
    uint32_t rank = 0, world = 1, n_total = 0;     // global builder count
    uint64_t n_done_global = 0;
    double t0 = 0;
    std::string qbuf;
    std::vector<uint64_t> qoff, foff;
    std::vector<uint32_t> fids;
    double p1_align_ms = 0, p1_host_ms = 0, p1_launch_ms = 0;
    uint64_t slot_long_n[4] = {0, 0, 0, 0};
    double slot_long_ms[4] = {0, 0, 0, 0};
    double g1_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // one-group schedule, wall time of a slot's steps (debug report)
    uint64_t role_serial_ns[4] = {0, 0, 0, 0};
    std::vector<FinishedContig *> emit_queue;      // finished contigs whose emission has not run yet
    // Scratch of one alignment batch in the making (engine_batches_sketch .. engine_align_finish).  One per batch that can be in flight at a
    // time: a group's (index = group; one group: index 0).
    struct Lane {
        std::vector<uint32_t> who;
        std::vector<uint64_t> mz_off;                   // minimizer offsets of the sketch batch
        std::vector<SketchReq> sk_reqs;                 // requests of a sketch batch: changed consensus stretches, then the candidates
        std::vector<uint32_t> sk_ref;                   // per builder of the batch: its consensus request (~0u: consensus unchanged)
        std::vector<uint64_t> stage_off;                // offsets of the builders' consensus minimizer lists in the seeding kernel's staging buffer
        std::vector<size_t> tail_from;                  // per builder of the batch: first list entry that travels this time
        std::vector<TailCopy> tail_jobs;
        PinBuf pin_tail;                                // the scatter kernel's job descriptors
        std::vector<uint8_t> cons_changed;              // per builder of the batch: its consensus changed since the batch before
        std::vector<ConsJob> cons_jobs;
        PinBuf pin_cons;                                // cons_update_kernel's job descriptors
        hipStream_t cons_stream = nullptr; hipEvent_t cons_ev = nullptr;
        std::vector<char> cons_check;                   // NSGPU_CONS_CHECK: a device copy read back
        std::vector<DevBuf> retired;                    // resident lists replaced by larger ones, freed at the lane's next batch
        std::vector<mm2::AlnOut> outs;
        std::vector<uint8_t> early_sure;                // per request: its claim cannot fail (engine_early_updates)
        std::vector<const uint8_t *> staged;            // per request: the changed stretch's text in device memory (cons_update_kernel's source)
        std::vector<CountJob> cnt_jobs; PinBuf pin_cnt; // the count tables' update jobs of the batch: descriptors, then the removed minimizers (pinned)
            double sk_ms[6] = {0, 0, 0, 0, 0, 0};           // engine_batches_sketch: splice plan, requests, sketch call, index loop, enqueue of seeds..DP, wait + first step
        int sketch_ws = 0;                              // mm_sketch workspace of the lane's batches
        void release()
        {
            for (DevBuf &d : retired) d.release();
            retired.clear();
            pin_tail.release(), pin_cons.release(), pin_cnt.release();
            if (cons_stream) { (void)hipStreamSynchronize(cons_stream); (void)hipStreamDestroy(cons_stream); (void)hipEventDestroy(cons_ev); cons_stream = nullptr; }
        }
    } lane[kMaxGroups];
    std::vector<uint32_t> global_pends;             // several ranks, one-group schedule: the reads ALL ranks' builders align in this slot, sorted (run_consensus_dist)
    bool have_global_pends = false;
    std::vector<uint32_t> early_pends; std::vector<int32_t> early_widx; std::vector<int8_t> early_lane;      // scratch of engine_early_updates
    std::vector<uint32_t> early_cand; std::vector<std::atomic<uint8_t>> early_claim; uint64_t n_early_stolen = 0; double early_last_claim_ms = 0;      // the watch's builders / who runs which (a flag each) / tasks run by a thread other than the builder's
    double early_part_ms[2] = {0, 0}, early_task_ms = 0, early_task_max_ms = 0, early_conv_ms = 0;     // debug report: wall of the two parts' loops, sum / per-slot maximum of their tasks, skeleton + conversion inside
    uint64_t n_early = 0, n_early_retry = 0; double early_ms = 0;      // (retry: a status word seen before all of its data, ksw_collect.hpp)       // graph updates run ahead of the slot's end / wall of that (debug print)
    double crit_u_ms = 0, crit_m_ms = 0;              // sum over host phases of the slowest update_graph / main-path recompute (debug print)
    std::vector<uint32_t> dbg_batch_sizes;          // alignments per batch, in order (debug print: how full the slots are over the run)
    uint64_t n_wq_exact = 0;                          // window-query batches that went the exact multi-step way
    int deferred_fresh = -1;                          // group whose freshly started contigs take their first steps with the next host phase
    AlignBatch ab[kMaxGroups];                        // alignment batch of each group, DP kernels in flight between part 1 and part 2
    // ---- deferred alignments (one-group schedule; nsgpu_set_defer) ----
    // A read across a tandem repeat has 10^4 - 10^5 anchors; chaining its list and running its DP problems take 5 - 10 ms, twice a whole slot,
    // and in lock step every builder of the slot waited for it.  The rule (the lock-step oracle's, oracle/consensus_oracle.cpp LockStep::VT::extra):
    // an alignment whose anchor list -- collect_seed_hits's, before chaining -- is longer than `defer_anchors` takes `defer_slots` MORE slots
    // than the others: its result is delivered, and its claim made, at the end of slot s + defer_slots.  The builder's consensus does not change
    // meanwhile, so the result is the one the slot's batch would have had; the jobs leave the batch (AlignBatch::deferred) for a batch of their own
    // that a thread of its own takes through chaining, DP rounds and conversion on workspaces of its own, joined when due.
    struct DeferBatch {
        AlignBatch AB;
        std::vector<uint32_t> builder;                // per request: the builder
        std::vector<std::vector<mm2::Anchor>> qmz;    // per request: the candidate's minimizers (the batch's copy lies in a sketch workspace that is reused)
        std::vector<mm2::AlnOut> outs;
        uint32_t due = 0;
        bool busy = false;
        std::thread th;
        int rc = NSGPU_OK;
        std::string err;
    } defer[4];
    uint64_t dbg_cnt_rebuilds = 0, dbg_cnt_rebuild_slots = 0, dbg_cnt_rebuild_max = 0, dbg_cnt_updates = 0, dbg_cnt_keys = 0;      // count tables (debug print)
    uint32_t cur_slot = 0;
    uint64_t n_deferred = 0; double defer_join_ms = 0, defer_run_ms = 0;
    std::vector<uint32_t> fwho;                    // builders of the window-query batch
    std::vector<uint32_t> awho[kMaxGroups];           // builders of that batch
    Builder *local(uint32_t gid) { return gid % world == rank ? &D.B[gid / world] : nullptr; }
    // ---- conflict-aware seeds (nsgpu_set_schedule; SURVEY 8e "assign seed reads by MinHash bucket locality") ----
    // Reads are grouped once into buckets of the whole-read filter graph (x -- y when y is a filter result of x or of its reverse
    // complement, nsgpu_filter_all_reads): in id order every read without a bucket opens one and takes every bucketless read within
    // `depth` hops (breadth first).  Buckets are adjacent when an edge joins them.  A contig in flight occupies the buckets of its seed
    // and of every read it claimed; a seed must lie in a bucket that is neither occupied nor within `rings` adjacency steps of an
    // occupied one, and the lowest unclaimed read that qualifies is taken, by the waiting builders in global builder order.  A builder
    // that finds none although unclaimed reads exist asks again at its group's next slot.  All of it is a function of replicated data.
    struct SeedPolicy {
        uint32_t depth = 0, rings = 1, tail_rings = 1;        // tail_rings: the exclusion radius while more than half of all builders are waiting for a seed
        uint32_t rings_now = 1;
        std::vector<uint32_t> bucket_of;                 // read -> bucket
        // the filter's answer to every whole read, both strands (what the buckets are built from): kept, because a fresh contig's first window
        // IS its seed read -- that query need not go to the GPU again (engine_window_queries)
        std::vector<uint64_t> wr_off; std::vector<uint32_t> wr_ids;
        std::vector<uint64_t> adj_off; std::vector<uint32_t> adj;     // bucket adjacency (CSR, ascending, without itself)
        std::vector<uint64_t> bk_off; std::vector<uint32_t> bk_reads; // reads of every bucket, ascending
        std::vector<uint32_t> bk_next;                   // per bucket: index into its reads of the first one not known to be claimed
        std::vector<uint32_t> occ;                       // per bucket: members of contigs in flight
        std::vector<std::vector<uint32_t>> members;      // per GLOBAL builder: seed + claimed reads of its contig in flight
        uint64_t n_unclaimed = 0, n_idle = 0;
        uint64_t n_filter_results = 0; bool have_wr = false;      // whole_read_filter has run for this stage (wr_off / wr_ids hold its answers)
        std::vector<uint8_t> blocked;                    // scratch of a seed round: bucket within `rings` steps of an occupied one
        std::vector<uint32_t> q_cur, q_nxt, stamp;
        uint32_t epoch = 0;
        // marks every bucket within `rings` adjacency steps of the buckets in q_cur (a breadth-first walk of its own: a bucket that is
        // already blocked from elsewhere may still be a step on the way)
        void spread()
        {
            if (stamp.size() != blocked.size()) stamp.assign(blocked.size(), 0), epoch = 0;
            ++epoch;
            for (uint32_t x : q_cur) stamp[x] = epoch, blocked[x] = 1;
            for (uint32_t d = 0; d < rings_now && !q_cur.empty(); ++d) {
                q_nxt.clear();
                for (uint32_t x : q_cur)
                    for (uint64_t i = adj_off[x]; i < adj_off[x + 1]; ++i) if (stamp[adj[i]] != epoch) { stamp[adj[i]] = epoch; blocked[adj[i]] = 1; q_nxt.push_back(adj[i]); }
                q_cur.swap(q_nxt);
            }
        }
    } sp;
};

static inline bool in_group(const Builder &b, int group) { return group < 0 || b.group == group; }

// consensus_driver.hip
int engine_begin(nsgpu_ctx *c, uint32_t n_builders_total, uint32_t rank, uint32_t world);
void engine_advance(nsgpu_ctx *c, bool only_fresh, int group);
int engine_slot(nsgpu_ctx *c, uint32_t slot, int part = 0);
int engine_window_loop(nsgpu_ctx *c, int group);
int engine_batches(nsgpu_ctx *c, int group);
int engine_finish(nsgpu_ctx *c, uint32_t n_threads_out);
int run_consensus(nsgpu_ctx *c, uint32_t n_builders, uint32_t n_threads_out);
// consensus_schedule.hip
struct AutoSchedule { uint32_t builders, depth, rings, tail; };
AutoSchedule auto_schedule(uint64_t n_reads, uint64_t n_bases, uint64_t n_filter_results);
int whole_read_filter(nsgpu_ctx *c, Engine *E);
int seed_policy_init(nsgpu_ctx *c, Engine *E);
void engine_seed_requests(nsgpu_ctx *c, std::vector<uint32_t> &gids, std::vector<uint32_t> &cursors, int group);
uint32_t engine_seed_resolve(nsgpu_ctx *c, const uint32_t *gids, const uint32_t *cursors, uint32_t n);
void engine_claim_requests(nsgpu_ctx *c, std::vector<uint32_t> &gids, std::vector<uint32_t> &reads, int group);
void engine_claim_resolve(nsgpu_ctx *c, const uint32_t *gids, const uint32_t *reads, uint32_t n);
// consensus_debug.hip
void debug_report_slots(nsgpu_ctx *c, Engine *E);
void debug_report_stage(nsgpu_ctx *c, Engine *E, const struct rusage &ru0, double w_begin, double w_slot, double w_seed, double w_claim, double tf);

}  // namespace nsgpu
