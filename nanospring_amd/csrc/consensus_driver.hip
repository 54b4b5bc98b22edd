// consensus_driver.hip -- a11/a12: the greedy contig driver (Consensus::generateAndWriteConsensus,
// addRelatedReads, createGraph, getRead; src/Consensus.cpp:21-340, 388-403, 444-468) restated as a
// batch engine for one GPU.
//
// The reference runs numThr OpenMP threads, each building one contig at a time and claiming reads
// optimistically (inGraph[] + try_lock); its output depends on thread timing for numThr > 1.  Here
// the "threads" are VIRTUAL builders advanced in lock-step rounds:
//   1. every builder runs on a host thread until it needs a candidate list (window queries) or an
//      alignment, or finishes its contig;
//   2. builders that need a new contig claim their seed read, strictly in builder order;
//   3. all window queries of the round go through ONE sketch+filter launch sequence, all
//      alignments through ONE batched alignRead (all DP in one kernel launch per DP round);
//   4. successful alignments claim their read strictly in builder order (lowest builder wins).
// Any such schedule is one of the interleavings the reference's numThr-thread run can produce
// (no try_lock ever fails); with ONE builder it is the reference's deterministic -t 1 schedule.
// The result is deterministic for a given (reads, salts, n_builders).
// The builders form four groups that go through these steps a quarter of a period apart (host phase |
// batches part 1 with the DP launch | DP in flight | batches part 2), so that host cores and GPU work
// at the same time: see run_consensus / engine_slot.
#include "engine.hpp"

namespace nsgpu {
static void graph_shared_free(void *p) { delete static_cast<DevGraphShared *>(p); }
// pools and streams of the consensus graphs in HBM: made once per context (pinned memory takes its time to map), reused by every stage
static int graph_shared_get(nsgpu_ctx *c, DevGraphShared **out)
{
    *out = nullptr;
    uint32_t mode = c->graph_mode & 0xffu;
    bool check = (c->graph_mode & NSGPU_GRAPH_CHECK) != 0 || getenv("NSGPU_GRAPH_CHECK") != nullptr;
    if (!c->graph_mode_set) { const char *e = getenv("NSGPU_GRAPH"); if (e) mode = !strcmp(e, "host") ? NSGPU_GRAPH_HOST : !strcmp(e, "device") ? NSGPU_GRAPH_DEVICE : NSGPU_GRAPH_AUTO; }
    // (the measured cross-over, profiles/r06_graph_placement_by_threads.txt: with 16 threads the pointer graph's updates ride on the DP phase for nothing, with 2 they ARE the step; the two placements meet at 6)
    if (mode == NSGPU_GRAPH_AUTO) mode = host_threads() <= 5 ? NSGPU_GRAPH_DEVICE : NSGPU_GRAPH_HOST;
    c->graph_used = mode;
    if (mode != NSGPU_GRAPH_DEVICE) return NSGPU_OK;
    if (!c->graph_shared) {
        DevGraphShared *sh = new DevGraphShared();
        c->graph_shared = sh, c->graph_shared_free = graph_shared_free;
        NS_TRY(role_stream_create(&sh->serve_stream, "graph"));
        NS_TRY(role_stream_create(&sh->copy_stream, "graph_copy"));
        sh->dev.set_slab_bytes((size_t)1 << 30);
        sh->pin.set_slab_bytes((size_t)256 << 20);
    }
    DevGraphShared *sh = static_cast<DevGraphShared *>(c->graph_shared);
    sh->max_ops = 2 * c->reads.max_len + 64;
    sh->check = check;
    { const char *e = getenv("NSGPU_SOA_DEBUG_FLAGS"); sh->dbg_flags = e ? (uint32_t)atoi(e) : 0; }
    sh->n_updates = 0, sh->n_launches = 0, sh->n_grow = 0, sh->n_mid_copies = 0, sh->kernel_wait_ns = 0, sh->bytes_back = 0, sh->update_ns = 0, sh->final_wait_ns = 0, sh->n_seq_updates = 0, sh->n_full_walks = 0, sh->n_splits = 0, sh->n_regrow = 0;
    sh->aborted = false;
    sh->edge_thr = c->prm.edge_threshold;
    for (auto &t : sh->phase_ticks) t = 0;
    for (auto &t : sh->hist) t = 0;
    for (auto &t : sh->slow_phase) t = 0;
    for (auto &t : sh->cnt) t = 0;
    for (auto &t : sh->cyc) t = 0;
    for (auto &t : sh->rt) t = 0;
    *out = sh;
    return NSGPU_OK;
}

static void engine_free(void *p)
{
    pool_drain();                                     // no emission task may outlive the engine
    Engine *E = static_cast<Engine *>(p);
    for (Builder &b : E->D.B) { b.d_mz.release(); b.d_cons.release(); b.d_cnt.release(); b.d_cnt_hm.release(); }
    for (Engine::Lane &L : E->lane) L.release();
    for (Engine::DeferBatch &DB : E->defer) if (DB.th.joinable()) DB.th.join();       // (a run that ended in an error may leave one in flight)
    delete E;
}


int engine_begin(nsgpu_ctx *c, uint32_t n_builders_total, uint32_t rank, uint32_t world)
{
    NS_CHECK(c->have_index && c->have_salts, NSGPU_ERR_ARG, "consensus: call nsgpu_sketch and nsgpu_build_index first");
    const bool auto_now = c->sched_auto || (n_builders_total == 0 && !c->sched_set);
    NS_CHECK((n_builders_total >= 1 || auto_now) && world >= 1 && rank < world, NSGPU_ERR_ARG, "consensus: bad builder / rank arguments (0 builders = the library's choice needs the automatic schedule: nsgpu_set_schedule_auto)");
    NS_CHECK(c->h_off.size() == (size_t)c->reads.n + 1, NSGPU_ERR_ARG, "consensus: reads must be loaded first");
    if (c->cons_engine) { c->cons_engine_free(c->cons_engine); c->cons_engine = nullptr; }
    Engine *E = new Engine();
    c->cons_engine = E, c->cons_engine_free = engine_free;
    Driver &D = E->D;
    D.c = c;
    D.N = c->reads.n;
    D.edge_thr = c->prm.edge_threshold;
    D.id_base = c->read_id_base;
    NS_CHECK((uint64_t)D.id_base + D.N <= 0xFFFFFFFFull, NSGPU_ERR_RANGE, "read id base + reads exceeds read_t");
    D.offset = D.N ? (size_t)(c->reads.n_bases / D.N) / 4 : 0;      // avgReadLen is integer-truncated (src/ReadData.cpp:207)
    if (D.offset == 0) D.offset = 1;                                  // the reference would never terminate with a zero stride
    D.in_graph.assign(D.N, 0);
    D.rep.assign((size_t)D.N + 1, 0);
    NS_TRY(graph_shared_get(c, &D.gsh));
    E->t0 = now_ms();
    if (D.N) NS_TRY(nsgpu_check_repetitive(c, D.rep.data()));
    if (!auto_now && !c->defer_set) c->defer_anchors = c->defer_slots = 0;
    if (!auto_now && !c->sched_set) c->sched_groups = 4, c->seed_bucket_depth = 0, c->seed_rings = 1, c->seed_tail_rings = 1;      // (nothing chosen, builders given: the defaults, whatever an automatic run before derived)
    if (auto_now) {
        c->sched_groups = 1;
        if (D.N) NS_TRY(whole_read_filter(c, E));
        const AutoSchedule a = auto_schedule(D.N, c->reads.n_bases, E->sp.n_filter_results);
        c->seed_bucket_depth = a.depth, c->seed_rings = a.rings, c->seed_tail_rings = a.tail;
        if (!c->defer_set) c->defer_anchors = 4096, c->defer_slots = 2;        // (reads across long repeats: Engine::DeferBatch)
        if (n_builders_total == 0) n_builders_total = a.builders;
        if (getenv("NSGPU_CONS_DEBUG")) fprintf(stderr, "[cons] automatic schedule: %.1f filter results per read -> %u builders, one group, buckets of depth %u, %u rings (%u in the tail)\n",
                                                D.N ? (double)E->sp.n_filter_results / D.N : 0.0, n_builders_total, a.depth, a.rings, a.tail);
    }
    if (n_builders_total > D.N && D.N > 0) n_builders_total = D.N;
    if (D.N == 0) n_builders_total = 1;
    {   // one group's alignments of one slot share a DP sequence pool addressed with 32 bits (align_batch.hip): an alignment needs at most a few
        // times its read and the stretch of consensus under it, so this many builders per group and rank can never overflow it.  A function of
        // replicated values only (every rank holds all reads): all ranks clamp alike.
        const uint64_t per_aln = 6ull * std::max<uint64_t>(c->reads.max_len, 1024);
        const uint64_t cap = std::max<uint64_t>(8, (3500ull << 20) / per_aln) * (uint64_t)n_groups(c) * world;
        if (n_builders_total > cap) n_builders_total = (uint32_t)cap;
    }
    E->rank = rank, E->world = world, E->n_total = n_builders_total;
    const uint32_t n_local = n_builders_total > rank ? (n_builders_total - rank + world - 1) / world : 0;
    D.B.resize(n_local);
    for (uint32_t i = 0; i < n_local; ++i) D.B[i].id = i, D.B[i].gid = rank + i * world, D.B[i].group = (int)((D.B[i].gid >> 3) % (uint32_t)n_groups(c));
    memset(&c->cons_stats, 0, sizeof(c->cons_stats));
    c->cons_stats.n_builders = n_builders_total;
    c->have_cons = false;
    return seed_policy_init(c, E);
}

// phase 1/3: consume deliveries and run every local builder to its next request

static void plan_splice(Builder &b, int w, int k);
void engine_advance(nsgpu_ctx *c, bool only_fresh, int group)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    const double a0 = now_ms();
    // (with the host phase of a group also the first steps of the contigs that the group of the slot before started at the slot
    // boundary: see run_consensus)
    const int dg = only_fresh ? -1 : E->deferred_fresh;
    if (!only_fresh) E->deferred_fresh = -1;
    // The accepted reads this phase puts into graphs in HBM (what the slot's early updates left: contested reads, deferred alignments; with several
    // groups: every accepted read): ONE launch, one workgroup per graph, handed their scripts by the loop below
    std::vector<DevGraph *> armed;
    if (D.gsh && !only_fresh) {
        for (Builder &b : D.B) {
            if (!in_group(b, group) || b.st != Builder::GOT_ALIGN || !b.accepted || b.early_updated || !b.g) continue;
            DevGraph *g = b.g->dev();
            if (!g) continue;
            const int rc = g->prepare(b.query.size());
            if (rc != NSGPU_OK) { D.graph_failed(rc); break; }
            armed.push_back(g);
        }
        if (D.graph_rc.load() == NSGPU_OK && armed.size() > 1) { const int rc = graph_serve_launch(D.gsh, armed.data(), armed.size()); if (rc != NSGPU_OK) D.graph_failed(rc); }
    }
    par_for_pinned("host.phase", D.B.size(), [&](size_t i) {
        Builder &b = D.B[i];
        if ((in_group(b, group) && (!only_fresh || b.st == Builder::ADVANCE)) || (dg >= 0 && in_group(b, dg) && b.st == Builder::ADVANCE)) {
            D.advance(b);
            // which stretch of the changed consensus has to be sketched again: here, while the strings are in this thread's cache, and not as
            // a loop of its own at the head of the GPU chain (engine_batches_sketch)
            if (b.st == Builder::WAIT_ALIGN && !b.idx_valid && !b.sp_ready) { plan_splice(b, (int)c->prm.m_w, (int)c->prm.m_k); b.sp_ready = true; }
        }
    });
    for (DevGraph *g : armed) if (g->armed()) g->cancel();
    // the edit emission of the contigs finished in this phase: background tasks of the host pool, picked up whenever a
    // thread has nothing else to do (nothing waits for them before the end of the stage)
    for (Builder &b : D.B)
        for (; b.n_queued < b.contigs.size(); ++b.n_queued) {
            FinishedContig *fc = b.contigs[b.n_queued].get();
            if (!fc->g) continue;
            pool_post([&D, fc] { D.emit_contig(*fc); });
        }
    c->cons_stats.graph_ms += now_ms() - a0;
    if (D.graph_rc.load() != NSGPU_OK) set_error("%s", D.graph_err.c_str());      // (reported by the caller: engine_slot checks graph_rc)
    double mx = 0, mxu = 0, mxm = 0;
    for (Builder &b : D.B) if (in_group(b, group)) { if (b.last_ms > mx) mx = b.last_ms; b.last_ms = 0; if (b.last_u > mxu) mxu = b.last_u; if (b.last_m > mxm) mxm = b.last_m; b.last_u = b.last_m = 0; }
    c->cons_stats.graph_crit_ms += mx;       // sum over phases of the slowest builder step: the floor of the phase wall
    E->crit_u_ms += mxu, E->crit_m_ms += mxm;
}

// phases 4+5: window queries and alignments of the local builders (GPU batches); no claims yet
// batches, part 1: for the builders that wait for an alignment, minimizer sketches, index, seeds / chains / DP plan and the
// launch of the DP kernels -- which stay in flight until part 2
// batches, part 1a: for the builders that wait for an alignment, the minimizers of every changed consensus and of every
// candidate read (GPU, mm_sketch.hip), the consensus indexes, and the first host step of the alignments (seeds / chains /
// DP plan).  (Optionally the batch is cut in two halves whose sketches run concurrently on the GPU -- two workspaces, two
// streams -- with the host work of the first half overlapping the sketch of the second: see below.)
// ---- incremental consensus sketch ---------------------------------------------------------------------------
// Whether position x is a minimizer of a string depends on the bases within w + k of x only (mm_sketch's state after a push is
// the right-most minimum of the last w pushes -- mm_sketch.hip -- and an element is emitted at most w pushes after its own).
// So after an accepted read changed the consensus between a common prefix of P and a common suffix of S bases, only
// new[P - 2m, Ln - S + 2m), m = w + k + 2, is sketched again; of its minimizers those at least m inside are exact (the artificial
// start / end of the substring cannot reach them; a substring that starts at 0 or ends at Ln is exact up to that end), the old
// minimizers left of P - m stay, those right of the old end of the change move by the length difference.
// NSGPU_SKETCH_CHECK=1 compares every spliced list with a sketch of the whole string.
static void plan_splice(Builder &b, int w, int k)
{
    const std::string &nw = b.g->path(), &od = b.mz_str;
    Builder::Splice &sp = b.sp;
    sp = Builder::Splice();
    if (od.empty() || b.mz.empty()) return;
    const size_t Ln = nw.size(), Lo = od.size(), mn = std::min(Ln, Lo);
    // the graph knows from where on its main path changed (a lower bound of the common prefix): a multi-megabase consensus is compared
    // from there, not from its first base
    size_t P = std::min(b.chg_lb, mn);
    {
        static const bool check = getenv("NSGPU_SKETCH_CHECK") != nullptr;
        if (check && P && memcmp(nw.data(), od.data(), P) != 0) { fprintf(stderr, "nsgpu: the main path changed in front of path_changed_from (internal error)\n"); abort(); }
    }
    while (P + 8 <= mn && memcmp(nw.data() + P, od.data() + P, 8) == 0) P += 8;
    while (P < mn && nw[P] == od[P]) ++P;
    size_t S = 0;
    while (S + 8 <= mn - P && memcmp(nw.data() + Ln - S - 8, od.data() + Lo - S - 8, 8) == 0) S += 8;
    while (S < mn - P && nw[Ln - 1 - S] == od[Lo - 1 - S]) ++S;
    sp.cp = P, sp.cs = S, sp.have_common = true;                 // (whatever is decided below: the base codes are spliced with these)
    const size_t m = (size_t)(w + k + 2);
    const size_t a = P > 2 * m ? P - 2 * m : 0, e = Ln - S + 2 * m < Ln ? Ln - S + 2 * m : Ln;
    if ((e - a) * 2 > Ln) return;                                // most of it changed: a whole sketch is as cheap
    {   // The locality argument above counts POSITIONS, mm_sketch's window counts PUSHES: a k-mer equal to its own reverse complement
        // pushes nothing (minimap2/sketch.c:108), so every such k-mer between a kept minimizer and the change stretches the window by one
        // position -- (AT)n, (CG)n, (ACGT)n runs have many.  With one anywhere within 3m of the changed stretch the whole string is sketched.
        const size_t lo = P > 3 * m ? P - 3 * m : 0, hi = Ln - S + 3 * m < Ln ? Ln - S + 3 * m : Ln;
        if ((k & 1) == 0 && hi - lo >= (size_t)k)
            for (size_t i = lo; i + (size_t)k <= hi; ++i) {
                bool sym = true;
                for (int j = 0; j < k / 2 && sym; ++j) {
                    const char x = nw[i + (size_t)j], y = nw[i + (size_t)(k - 1 - j)];
                    sym = (x == 'A' && y == 'T') || (x == 'T' && y == 'A') || (x == 'C' && y == 'G') || (x == 'G' && y == 'C');
                }
                if (sym) return;
            }
    }
    sp.full = false;
    sp.P = P, sp.S = S;
    sp.a = a, sp.B_sub = e;
    sp.A = a == 0 ? 0 : P - m;                                   // new minimizers are taken from [A, B)
    sp.B = e == Ln ? Ln : Ln - S + m;
    sp.delta = (ssize_t)Ln - (ssize_t)Lo;
}

// returns the index of the first list entry that may differ from the list before (everything in front of it was kept)
static size_t apply_splice(Builder &b, const mm2::Anchor *sub, size_t n_sub, int w, int k)
{
    size_t first_diff = 0;
    auto pos_of = [](const mm2::Anchor &x) { return (size_t)((x.y & 0xffffffffull) >> 1); };
    const Builder::Splice &sp = b.sp;
    const std::string &nw = b.g->path();
    b.cnt_rem.clear(), b.cnt_add.clear();
    if (sp.full) { b.mz.assign(sub, sub + n_sub); b.cnt_valid = false; }       // (a new list: its count table is rebuilt)
    else {
        std::vector<mm2::Anchor> out;
        out.reserve(b.mz.size() + n_sub);
        size_t i = 0;
        {   // the kept prefix ends at the first old minimizer at or behind A (positions ascend: binary search, then one block copy)
            size_t lo = 0, hi = b.mz.size();
            while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (pos_of(b.mz[mid]) < sp.A) lo = mid + 1; else hi = mid; }
            i = lo;
            out.insert(out.end(), b.mz.begin(), b.mz.begin() + (ptrdiff_t)i);
        }
        first_diff = i;
        for (size_t j = 0; j < n_sub; ++j) {
            const size_t x = pos_of(sub[j]) + sp.a;
            if (x >= sp.A && x < sp.B) { mm2::Anchor t = sub[j]; t.y = (t.y & ~0xffffffffull) | ((uint64_t)x << 1 | (t.y & 1)); out.push_back(t); }
        }
        for (size_t j = first_diff; j < out.size(); ++j) b.cnt_add.push_back(out[j].x >> 8);         // what came ...
        const size_t B_old = (size_t)((ssize_t)sp.B - sp.delta);
        for (; i < b.mz.size() && pos_of(b.mz[i]) < B_old; ++i) b.cnt_rem.push_back(b.mz[i].x >> 8);      // ... and what left: the count table's update
        for (; i < b.mz.size(); ++i) { mm2::Anchor t = b.mz[i]; const size_t x = (size_t)((ssize_t)pos_of(t) + sp.delta); t.y = (t.y & ~0xffffffffull) | ((uint64_t)x << 1 | (t.y & 1)); out.push_back(t); }
        b.mz.swap(out);
        // (a re-sketched stretch gives back most of the minimizers it had: a hash on both sides is no change of its count)
        if (!b.cnt_rem.empty() && !b.cnt_add.empty()) {
            std::sort(b.cnt_rem.begin(), b.cnt_rem.end()), std::sort(b.cnt_add.begin(), b.cnt_add.end());
            size_t x = 0, y = 0, nx = 0, ny = 0;
            while (x < b.cnt_rem.size() && y < b.cnt_add.size()) {
                if (b.cnt_rem[x] == b.cnt_add[y]) ++x, ++y;
                else if (b.cnt_rem[x] < b.cnt_add[y]) b.cnt_rem[nx++] = b.cnt_rem[x++];
                else b.cnt_add[ny++] = b.cnt_add[y++];
            }
            while (x < b.cnt_rem.size()) b.cnt_rem[nx++] = b.cnt_rem[x++];
            while (y < b.cnt_add.size()) b.cnt_add[ny++] = b.cnt_add[y++];
            b.cnt_rem.resize(nx), b.cnt_add.resize(ny);
        }
    }
    {   // mz_str = nw, copying only what can differ
        const size_t keep = std::min(std::min(b.chg_lb, b.mz_str.size()), nw.size());
        b.mz_str.resize(keep);
        b.mz_str.append(nw, keep, std::string::npos);
    }
    static const bool check = getenv("NSGPU_SKETCH_CHECK") != nullptr;
    if (check && b.mz_str != nw) { fprintf(stderr, "nsgpu: incremental copy of the main path differs (internal error)\n"); abort(); }
    if (check) {
        std::vector<mm2::Anchor> full;
        mm2::mm_sketch(nw.data(), (int)nw.size(), w, k, 0, full);
        bool same = full.size() == b.mz.size();
        for (size_t i = 0; same && i < full.size(); ++i) same = full[i].x == b.mz[i].x && full[i].y == b.mz[i].y;
        if (!same) {
            fprintf(stderr, "SKETCH SPLICE MISMATCH: builder %u, path %zu bases, full %zu vs spliced %zu minimizers (a %zu e %zu A %zu B %zu delta %zd full %d)\n", b.gid, nw.size(), full.size(),
                    b.mz.size(), sp.a, sp.B_sub, sp.A, sp.B, sp.delta, (int)sp.full);
            abort();
        }
    }
    return first_diff;
}

// ---- the consensus resident in HBM -----------------------------------------------------------------------------------------------------
// After an accepted read the new consensus differs from the old one between a common prefix of P and a common suffix of S bases, and the
// bytes in between travelled with the sketch batch anyway (the re-sketched stretch contains them).  The device copy is brought up to date
// in place: the SHORTER of prefix and suffix moves by the length difference (a contig is edited near the end it is growing at, so that is
// a few kilobytes of a megabase string), the middle is written from the staged stretch.  One workgroup per contig; a move is done chunk by
// chunk in the direction that never overwrites bytes it has yet to read, every chunk loaded whole before it is stored.

__global__ __launch_bounds__(256) void cons_update_kernel(const ConsJob *__restrict__ jobs)
{
    const ConsJob J = jobs[blockIdx.y];
    const int tid = (int)threadIdx.x;
    uint8_t *buf = J.buf;
    if (J.full) {                                           // the whole string was staged: every workgroup of the row takes its share
        uint8_t *d = buf + J.beg_new;
        const uint8_t *sm = J.mid;
        // (dwords where both sides allow it: the staging buffer and the copy's start are 16-byte aligned)
        if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(sm)) & 3) == 0) {
            const uint64_t nw = J.len_new >> 2;
            for (uint64_t i = (uint64_t)blockIdx.x * 256 + tid; i < nw; i += (uint64_t)gridDim.x * 256) reinterpret_cast<uint32_t *>(d)[i] = reinterpret_cast<const uint32_t *>(sm)[i];
            if (blockIdx.x == 0 && tid < (int)(J.len_new & 3)) d[(nw << 2) + tid] = sm[(nw << 2) + tid];
        } else for (uint64_t i = (uint64_t)blockIdx.x * 256 + tid; i < J.len_new; i += (uint64_t)gridDim.x * 256) d[i] = sm[i];
        return;
    }
    if (blockIdx.x) return;                                 // an in-place update is one workgroup's job (it orders its own reads and writes)
    const uint64_t M = J.len_new - J.P - J.S;               // new middle
    // the part that moves: prefix (kept suffix in place) or suffix (kept prefix in place)
    const bool move_prefix = J.beg_new != J.beg_old;
    const uint64_t n_mv = move_prefix ? J.P : J.S;
    const uint64_t src = move_prefix ? J.beg_old : J.beg_old + J.len_old - J.S, dst = move_prefix ? J.beg_new : J.beg_new + J.len_new - J.S;
    if (src != dst && n_mv) {
        constexpr int kPer = 64;                            // bytes per thread and chunk: loads of a chunk all in flight, then a barrier, then its stores
        constexpr uint64_t kChunk = 256 * kPer;
        const uint64_t n_ch = (n_mv + kChunk - 1) / kChunk;
        for (uint64_t k = 0; k < n_ch; ++k) {
            const uint64_t ch = dst > src ? n_ch - 1 - k : k;          // moving right: from the far end
            uint8_t v[kPer];
            // byte j of thread t: offset ch * kChunk + j * 256 + t (consecutive threads, consecutive bytes)
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const uint64_t o = ch * kChunk + (uint64_t)u * 256 + tid; v[u] = o < n_mv ? buf[src + o] : (uint8_t)0; }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const uint64_t o = ch * kChunk + (uint64_t)u * 256 + tid; if (o < n_mv) buf[dst + o] = v[u]; }
            __syncthreads();
        }
    }
    __syncthreads();
    for (uint64_t i = tid; i < M; i += 256) buf[J.beg_new + J.P + i] = J.mid[i];
}

__global__ void mz_tail_scatter_kernel(const TailCopy *__restrict__ jobs, uint32_t n_jobs)
{
    const uint32_t j = blockIdx.y;
    if (j >= n_jobs) return;
    const TailCopy t = jobs[j];
    const uint4 *s = reinterpret_cast<const uint4 *>(t.src);
    uint4 *d = reinterpret_cast<uint4 *>(t.dst);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// The contigs' consensus strings in HBM brought up to date for the builders of an alignment batch (after the sketch batch staged the changed
// stretches, before the plan kernel reads them), and AlignReq.ref_dev pointed at them.  A builder whose copy cannot be updated (nothing
// staged to update it from) goes without the device plan this time.
// ---- deferred alignments (Engine::DeferBatch) ----
// The jobs AB.deferred of the slot's batch (started and seeded by the host code, anchors in place) move into a batch of their own, which a thread of
// its own takes through chaining, DP rounds and conversion; their builders wait until slot cur_slot + defer_slots.
static int engine_defer_start(nsgpu_ctx *c, Engine *E, AlignBatch &AB, const std::vector<uint32_t> &who)
{
    Driver &D = E->D;
    Engine::DeferBatch &DB = E->defer[E->cur_slot & 3];
    NS_CHECK(!DB.busy && c->defer_slots >= 1 && c->defer_slots <= 3, NSGPU_ERR_ARG, "contig engine: a deferred batch is still in flight after %u slots (internal error)", c->defer_slots);
    const size_t n = AB.deferred.size();
    DB.AB.reqs.clear(), DB.builder.clear();
    if (DB.AB.jobs.size() < n) DB.AB.jobs.resize(n);
    if (DB.qmz.size() < n) DB.qmz.resize(n);
    if (DB.outs.size() < n) DB.outs.resize(n);
    for (size_t k = 0; k < n; ++k) {
        const uint32_t i = AB.deferred[k];
        Builder &b = D.B[who[i]];
        AlignReq r = AB.reqs[i];
        // (what stays valid while the builder waits: its consensus, its index, its host minimizer list, its query string; the candidate's minimizers
        // lie in the slot's sketch workspace and are copied)
        DB.qmz[k].assign(r.qry_mz, r.qry_mz + r.n_qry_mz);
        r.qry_mz = DB.qmz[k].data();
        r.ref_mz_dev = nullptr, r.ref_cnt = nullptr, r.ref_cnt_meta = nullptr, r.qry_dev = nullptr, r.ref_dev = nullptr;
        DB.AB.reqs.push_back(r);
        std::swap(DB.AB.jobs[k], AB.jobs[i]);
        DB.AB.jobs[k].pre_mz = DB.qmz[k].data();
        DB.builder.push_back(who[i]);
        b.st = Builder::DEFERRED, b.defer_due = E->cur_slot + c->defer_slots;
    }
    DB.due = E->cur_slot + c->defer_slots, DB.busy = true, DB.rc = NSGPU_OK, DB.err.clear();
    const int chain_ws = 18 + (int)(E->cur_slot & 3), dp_ws = 4 + (int)(E->cur_slot & 3);
    Engine::DeferBatch *dbp = &DB;
    Engine *Ep = E;
    DB.th = std::thread([c, dbp, Ep, chain_ws, dp_ws] {
        pool_bind_this_thread();
        const double t0 = now_ms();
        dbp->rc = hipSetDevice(c->prm.device) == hipSuccess ? align_seeded_jobs(c, dbp->AB, chain_ws, dp_ws, dbp->outs) : NSGPU_ERR_HIP;
        if (dbp->rc != NSGPU_OK) dbp->err = nsgpu_last_error();
        Ep->defer_run_ms += now_ms() - t0;        // (debug print; one deferred batch ends at a time in practice)
    });
    E->n_deferred += n;
    return NSGPU_OK;
}
// the deferred batches due at the end of this slot: their builders are ALIGNED like the slot's own
static int engine_defer_deliver(nsgpu_ctx *c, Engine *E)
{
    Driver &D = E->D;
    for (Engine::DeferBatch &DB : E->defer) {
        if (!DB.busy || DB.due != E->cur_slot) continue;
        const double t0 = now_ms();
        if (DB.th.joinable()) DB.th.join();
        E->defer_join_ms += now_ms() - t0;
        DB.busy = false;
        if (DB.rc != NSGPU_OK) { set_error("%s", DB.err.empty() ? "contig engine: a deferred alignment batch failed" : DB.err.c_str()); return DB.rc; }
        for (size_t k = 0; k < DB.builder.size(); ++k) {
            Builder &b = D.B[DB.builder[k]];
            std::swap(b.aln, DB.outs[k]);
            b.early_result = false, b.early_updated = false;
            ++b.n_align_calls;
            b.accepted = false;
            b.st = Builder::ALIGNED;
        }
    }
    return NSGPU_OK;
}

static int engine_cons_update(nsgpu_ctx *c, Engine *E, AlignBatch &AB, const std::vector<uint32_t> &who, const std::vector<uint8_t> &changed, const std::vector<const uint8_t *> &staged_of, Engine::Lane &L)
{
    Driver &D = E->D;
    std::vector<ConsJob> &jobs = L.cons_jobs;
    jobs.clear();
    const size_t n = who.size();
    for (size_t w = 0; w < n; ++w) {
        Builder &b = D.B[who[w]];
        const size_t Ln = b.g->path().size();
        const uint8_t *staged = changed[w] ? staged_of[w] : nullptr;
        if (changed[w]) {
            const bool full = b.sp.full || !b.dc_valid;
            if (!staged || (full && !b.sp.full) || Ln >= (1ull << 31)) { b.dc_valid = false; continue; }      // nothing whole to (re)build the copy from
            ConsJob J{};
            J.len_new = Ln, J.full = full;
            if (full) {
                const size_t want = 2 * Ln + (256u << 10);
                if (b.d_cons.cap < want) { if (b.d_cons.p) L.retired.push_back(b.d_cons); b.d_cons = DevBuf(); NS_TRY(b.d_cons.reserve(2 * want)); }
                J.buf = b.d_cons.as<uint8_t>(), J.mid = staged, J.beg_new = (b.d_cons.cap - Ln) / 2;
            } else {
                const size_t Lo = b.dc_len, P = b.sp.P, S = b.sp.S;
                if (P + S > Lo || P + S > Ln || P < b.sp.a || Ln - S > b.sp.B_sub) { fprintf(stderr, "nsgpu: consensus splice out of range (internal error)\n"); abort(); }
                // which side moves: the shorter one, if the buffer has the room on that side
                const ssize_t delta = (ssize_t)Ln - (ssize_t)Lo;
                const bool pre_fits = (ssize_t)b.dc_beg - delta >= 0, suf_fits = b.dc_beg + Ln <= b.d_cons.cap;
                bool move_prefix = P <= S;
                if (move_prefix && !pre_fits) move_prefix = false;
                if (!move_prefix && !suf_fits) move_prefix = pre_fits;
                if (!(move_prefix ? pre_fits : suf_fits)) {
                    // out of room on both sides: a larger buffer, the old content re-centred in it (device to device, on the stream the update runs on)
                    DevBuf bigger;
                    NS_TRY(bigger.reserve(4 * Ln + (512u << 10)));
                    const size_t nb = (bigger.cap - Lo) / 2;
                    if (!L.cons_stream) { NS_TRY(role_stream_create(&L.cons_stream, "seeds")); NS_HIP(hipEventCreateWithFlags(&L.cons_ev, hipEventDisableTiming)); }
                    NS_HIP(hipMemcpyAsync(bigger.as<uint8_t>() + nb, b.d_cons.as<uint8_t>() + b.dc_beg, Lo, hipMemcpyDeviceToDevice, L.cons_stream));
                    L.retired.push_back(b.d_cons);
                    b.d_cons = bigger, b.dc_beg = nb;
                    move_prefix = P <= S;
                }
                J.buf = b.d_cons.as<uint8_t>(), J.mid = staged + (P - b.sp.a);
                J.beg_old = b.dc_beg, J.len_old = Lo, J.P = P, J.S = S;
                J.beg_new = move_prefix ? (uint64_t)((ssize_t)b.dc_beg - delta) : b.dc_beg;
            }
            b.dc_beg = J.beg_new, b.dc_len = Ln, b.dc_valid = true;
            jobs.push_back(J);
        }
        if (b.dc_valid && AB.reqs[w].qry_dev) AB.reqs[w].ref_dev = b.d_cons.as<uint8_t>() + b.dc_beg, AB.reqs[w].ref_dev_lo = 0, AB.reqs[w].ref_dev_n = (uint32_t)b.dc_len;
        else AB.reqs[w].qry_dev = nullptr;
    }
    if (!jobs.empty()) {
        NS_TRY(L.pin_cons.reserve(jobs.size() * sizeof(ConsJob)));
        memcpy(L.pin_cons.p, jobs.data(), jobs.size() * sizeof(ConsJob));
        // on a stream of its own beside the seeding and chaining kernels: only the plan kernel behind them reads the copies (it waits for cons_ev)
        if (!L.cons_stream) { NS_TRY(role_stream_create(&L.cons_stream, "seeds")); NS_HIP(hipEventCreateWithFlags(&L.cons_ev, hipEventDisableTiming)); }
        hipLaunchKernelGGL(cons_update_kernel, dim3(16, (uint32_t)jobs.size()), dim3(256), 0, L.cons_stream, L.pin_cons.as<ConsJob>());
        NS_HIP(hipGetLastError());
        NS_HIP(hipEventRecord(L.cons_ev, L.cons_stream));
        AB.plan_wait_ev = L.cons_ev;
    } else AB.plan_wait_ev = nullptr;
    static const bool check = getenv("NSGPU_CONS_CHECK") != nullptr;          // every device copy against the host's string (the contig tests run under it)
    if (check) {
        if (L.cons_stream) NS_HIP(hipStreamSynchronize(L.cons_stream));
        for (size_t w = 0; w < n; ++w) {
            Builder &b = D.B[who[w]];
            if (!b.dc_valid) continue;
            L.cons_check.resize(b.dc_len);
            NS_HIP(hipMemcpy(L.cons_check.data(), b.d_cons.as<uint8_t>() + b.dc_beg, b.dc_len, hipMemcpyDeviceToHost));
            if (b.dc_len != b.g->path().size() || memcmp(L.cons_check.data(), b.g->path().data(), b.dc_len) != 0) {
                fprintf(stderr, "CONSENSUS COPY MISMATCH: builder %u, %zu bases (P %zu S %zu full %d)\n", b.gid, b.dc_len, b.sp.P, b.sp.S, (int)b.sp.full);
                abort();
            }
        }
    }
    return NSGPU_OK;
}

static int engine_batches_sketch(nsgpu_ctx *c, int group, int dp_ws)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    nsgpu_consensus_stats &S = c->cons_stats;
    const int gi = group < 0 ? 0 : group;
    Engine::Lane &L = E->lane[gi];
    std::vector<uint32_t> &who = L.who;
    who.clear();
    for (Builder &b : D.B) if (in_group(b, group) && b.st == Builder::WAIT_ALIGN) who.push_back(b.id);
    AlignBatch &AB = E->ab[gi];
    AB.reqs.clear();
    AB.host_ms = 0;
    static const bool no_early = getenv("NSGPU_NO_EARLY_UPDATES") != nullptr;        // A/B switch: results in one part, graph updates in the host phase, as before
    AB.plan_two_part = n_groups(c) == 1 && !no_early;
    E->awho[gi].clear();
    if (who.empty()) return NSGPU_OK;
    const double g0 = now_ms();
    const size_t n = who.size();
    // ONE sketch call and ONE chaining launch per batch.  Both were measured in halves (two sketch calls side by side; the second half's
    // index builds while the GPU chains the first) and were slower: a sketch call costs what its ~25 launches and its round trips cost
    // whatever its size (sketch + index 2.6 instead of 2.0 s per cfg2 step), and a chaining launch lasts as long as its longest list.
    // The requests: the changed stretch of every changed consensus (plan_splice, run where the consensus changed), then the candidate reads.
    std::vector<SketchReq> &sk = L.sk_reqs;
    std::vector<uint32_t> &sk_ref = L.sk_ref;
    sk.clear();
    sk_ref.assign(n, ~0u);
    bool unplanned = false;
    for (size_t w = 0; w < n && !unplanned; ++w) { const Builder &b = D.B[who[w]]; unplanned = !b.idx_valid && !b.sp_ready; }
    if (unplanned) par_for("sketch.plan", n, [&](size_t w) { Builder &b = D.B[who[w]]; if (!b.idx_valid && !b.sp_ready) plan_splice(b, (int)c->prm.m_w, (int)c->prm.m_k); });
    double tk = now_ms();
    L.sk_ms[0] += tk - g0;
    // (Measured in round 5 and removed: every read's minimizers, both strands, resident in HBM for the whole stage -- no candidate in the slots'
    // sketch batches -- and every changed stretch sketched by its builder's own launch right behind the graph update, collected here through an
    // event.  The sketch call went from 0.25 to 0.07 ms per slot and was gone in 60 % of the slots, but the launches cost the slowest
    // builder's task as much as the call had cost the batch, and the lists take 0.13 s per step to build: 6.93 / 6.70 s per step with both /
    // with the resident lists only, against 6.77 s without either.)
    for (size_t w = 0; w < n; ++w) {
        Builder &b = D.B[who[w]];
        if (b.idx_valid) continue;
        sk_ref[w] = (uint32_t)sk.size();
        if (b.sp.full) sk.push_back(SketchReq{b.g->path().data(), b.g->path().size()});
        else sk.push_back(SketchReq{b.g->path().data() + b.sp.a, b.sp.B_sub - b.sp.a});
    }
    const size_t q_base = sk.size();
    for (size_t w = 0; w < n; ++w) sk.push_back(SketchReq{D.B[who[w]].query.data(), D.B[who[w]].query.size()});
    AB.reqs.resize(n);
    if (AB.jobs.size() < n) AB.jobs.resize(n);
    L.sk_ms[1] += now_ms() - tk; tk = now_ms();
    const mm2::Anchor *mz = nullptr;
    std::vector<uint64_t> &mo = L.mz_off;
    NS_TRY(gpu_mm_sketch(c, sk, (int)c->prm.m_w, (int)c->prm.m_k, mz, mo, L.sketch_ws));
    L.sk_ms[2] += now_ms() - tk; tk = now_ms();
    constexpr bool resident_lists = true;           // (round 3's A/B, settled: the whole list through the pinned buffer cost the seeding kernel 16 MB over PCIe per slot)
    std::vector<size_t> &tail_from = L.tail_from;
    tail_from.assign(n, 0);
    // the plan kernel (plan.hip) reads the candidate where the sketch batch staged it and the consensus from the contig's resident copy
    const bool use_dev_plan = dp_ws >= 0;
    std::vector<uint8_t> &changed = L.cons_changed;
    changed.assign(n, 0);
    for (size_t w = 0; w < n; ++w) changed[w] = !D.B[who[w]].idx_valid;
    // the changed stretches' text where the sketch batch staged it in HBM (the source of the resident consensus copies' updates)
    std::vector<const uint8_t *> &staged = L.staged;
    staged.assign(n, nullptr);
    for (size_t w = 0; w < n; ++w) if (changed[w] && sk_ref[w] != ~0u) staged[w] = sketch_dev_seq(c, L.sketch_ws, sk_ref[w]);
    // lists retired by an earlier call: their last reader (that call's seeding kernel) has been waited for since
    for (DevBuf &d : L.retired) d.release();
    L.retired.clear();
    // The consensus minimizers of every builder go to the seeding kernel through one pinned staging buffer: room for each list
    // is set aside from an upper bound of its length after the splice (old list + newly sketched stretch).
    const int sws_i = 1 + 2 * gi;
    std::vector<uint64_t> &so = L.stage_off;
    so.assign(n + 1, 0);
    auto n_sub_of = [&](size_t w) -> uint64_t { return sk_ref[w] != ~0u ? mo[sk_ref[w] + 1] - mo[sk_ref[w]] : 0; };
    auto sub_of = [&](size_t w) -> const mm2::Anchor * { return sk_ref[w] != ~0u ? mz + mo[sk_ref[w]] : nullptr; };
    for (size_t w = 0; w < n; ++w) {
        const Builder &b = D.B[who[w]];
        uint64_t bound = b.mz.size();
        if (!b.idx_valid) bound += n_sub_of(w);
        so[w + 1] = so[w] + bound;
    }
    NS_TRY(c->seed_ws[sws_i].h_ref.reserve(so[n] * sizeof(mm2::Anchor) + 16));
    mm2::Anchor *stage = c->seed_ws[sws_i].h_ref.as<mm2::Anchor>();
    // the query minimizers stay in the pinned buffer of the sketch workspace until the alignments have been seeded
    for (size_t w = 0; w < n; ++w) {
        Builder &b = D.B[who[w]];
        const size_t qi = q_base + w;
        AB.reqs[w] = AlignReq{&b.idx, b.g->path().data(), b.g->path().size(), b.query.data(), b.query.size(), mz + mo[qi], (size_t)(mo[qi + 1] - mo[qi]), stage + so[w], 0};
        if (use_dev_plan) AB.reqs[w].qry_dev = sketch_dev_seq(c, L.sketch_ws, qi);       // (the consensus side: filled in below, once the device copies are up to date)
    }
    NS_TRY(align_prestep_start(c, AB, 0, n));
    // one loop over the builders: the minimizers of the changed consensus (splice), its base codes, the list into the staging
    // buffer, the candidate's base codes.  The index proper -- lookup table, occurrence cut-off -- and the seeds are the GPU's.
    par_for("index.build", n, [&](size_t w) {
        Builder &b = D.B[who[w]];
        size_t first_diff = b.d_mz_n;                 // nothing to upload when the consensus did not change
        if (!b.idx_valid) {
            first_diff = apply_splice(b, sub_of(w), (size_t)n_sub_of(w), (int)c->prm.m_w, (int)c->prm.m_k);
            // the base codes: prefix kept, suffix moved, the middle coded (a contig that grows at its left end used to be coded whole: 0.1 ms)
            if (b.sp.have_common) b.idx.set_sequence_spliced(b.g->path().data(), (uint32_t)b.g->path().size(), (int)c->prm.m_w, (int)c->prm.m_k, b.sp.cp, b.sp.cs);
            else b.idx.set_sequence_from(b.g->path().data(), (uint32_t)b.g->path().size(), (int)c->prm.m_w, (int)c->prm.m_k, b.chg_lb);
            {
                static const bool check = getenv("NSGPU_SKETCH_CHECK") != nullptr;
                if (check) {
                    mm2::RefIndex whole;
                    whole.set_sequence(b.g->path().data(), (uint32_t)b.g->path().size(), (int)c->prm.m_w, (int)c->prm.m_k);
                    if (whole.seq != b.idx.seq) { fprintf(stderr, "nsgpu: spliced base codes differ from the whole string's (internal error)\n"); abort(); }
                }
            }
            b.chg_lb = (size_t)-1;
            b.idx_valid = true, b.sp_ready = false;
        }
        if (b.mz.size() > so[w + 1] - so[w]) { fprintf(stderr, "nsgpu: spliced minimizer list longer than its bound (internal error)\n"); abort(); }
        // The contig's list is resident in HBM (Builder::d_mz): only the entries from the first changed one on go through the pinned
        // staging buffer (resident_lists off: the whole list, read by the seeding kernel where it lies in pinned memory, as before).
        if (first_diff > b.d_mz_n) first_diff = b.d_mz_n;
        const size_t from = resident_lists ? first_diff : 0;
        if (b.mz.size() > from) memcpy(stage + so[w], b.mz.data() + from, (b.mz.size() - from) * sizeof(mm2::Anchor));
        tail_from[w] = from;
        if (resident_lists) AB.reqs[w].ref_mz = b.mz.data();      // host copy: for the pairs the kernel hands back to the host code
        AB.reqs[w].n_ref_mz = b.mz.size();
        AB.jobs[w].seed_prepare();
    });
    L.sk_ms[3] += now_ms() - tk; tk = now_ms();
    if (resident_lists) {
        // room in the resident lists (growing one copies what it keeps), then ONE kernel moves every tail into place, on the stream the
        // seeding kernel is launched on right behind it
        nsgpu_ctx::SeedWs &SW = c->seed_ws[sws_i];
        if (!SW.stream) NS_TRY(role_stream_create(&SW.stream, "seeds"));
        std::vector<TailCopy> &jobs = L.tail_jobs;
        jobs.clear();
        uint32_t max_n = 0;
        for (size_t w = 0; w < n; ++w) {
            Builder &b = D.B[who[w]];
            const size_t n_new = b.mz.size(), from = tail_from[w];
            if (n_new * sizeof(mm2::Anchor) > b.d_mz.cap) {
                DevBuf bigger;
                NS_TRY(bigger.reserve(std::max<size_t>(2 * n_new, 16384) * sizeof(mm2::Anchor)));
                if (from) NS_HIP(hipMemcpyAsync(bigger.p, b.d_mz.p, from * sizeof(mm2::Anchor), hipMemcpyDeviceToDevice, SW.stream));
                L.retired.push_back(b.d_mz);         // freed once the stream is known to be past this slot (engine_batches_sketch's next call)
                b.d_mz = bigger;
            }
            if (n_new > from) { jobs.push_back(TailCopy{stage + so[w], b.d_mz.as<mm2::Anchor>() + from, (uint32_t)(n_new - from), 0}); max_n = std::max(max_n, (uint32_t)(n_new - from)); }
            b.d_mz_n = n_new;
            AB.reqs[w].ref_mz_dev = b.d_mz.as<mm2::Anchor>();
        }
        if (!jobs.empty()) {
            NS_TRY(L.pin_tail.reserve(jobs.size() * sizeof(TailCopy)));
            memcpy(L.pin_tail.p, jobs.data(), jobs.size() * sizeof(TailCopy));
            const uint32_t gx = std::max<uint32_t>(1, std::min<uint32_t>(64, (max_n + 255) / 256));
            hipLaunchKernelGGL(mz_tail_scatter_kernel, dim3(gx, (uint32_t)jobs.size()), dim3(256), 0, SW.stream, L.pin_tail.as<TailCopy>(), (uint32_t)jobs.size());
            NS_HIP(hipGetLastError());
        }
        // the contigs' count tables (seeds.hip): what the splice removed and added, behind the scatter on the same stream -- or the whole list
        // when the contig is new, was re-sketched whole, or its table has filled up with hashes that left
        std::vector<CountJob> &cj = L.cnt_jobs;
        cj.clear();
        size_t n_keys = 0;
        for (size_t w = 0; w < n; ++w) if (changed[w]) n_keys += D.B[who[w]].cnt_rem.size() + D.B[who[w]].cnt_add.size();
        NS_TRY(L.pin_cnt.reserve(n * sizeof(CountJob) + n_keys * sizeof(uint64_t) + 16));
        unsigned long long *key_stage = reinterpret_cast<unsigned long long *>(L.pin_cnt.as<CountJob>() + n);
        for (size_t w = 0; w < n; ++w) {
            Builder &b = D.B[who[w]];
            const size_t n_mz = b.mz.size();
            if (changed[w] || !b.cnt_valid) {
                // (keys whose count went to zero stay in the table: rebuilt once a third of the slots may be taken)
                const bool rebuild = !b.cnt_valid || (b.cnt_keys + b.cnt_add.size()) * 3 > ((uint64_t)1 << b.cnt_bits);
                CountJob j{};
                if (rebuild) {
                    const uint32_t bits = count_table_bits(n_mz + n_mz / 2);
                    if ((sizeof(CountSlot) << bits) > b.d_cnt.cap) {
                        if (b.d_cnt.p) L.retired.push_back(b.d_cnt), b.d_cnt = DevBuf();
                        NS_TRY(b.d_cnt.reserve(sizeof(CountSlot) << bits));
                    }
                    if (!b.d_cnt_hm.p) NS_TRY(b.d_cnt_hm.reserve((1024 + 4) * sizeof(uint32_t)));
                    b.cnt_bits = bits;
                    j.rebuild = 1, j.all = b.d_mz.as<mm2::Anchor>(), j.n_all = (uint32_t)n_mz;
                    b.cnt_keys = n_mz, b.cnt_valid = true;
                    ++E->dbg_cnt_rebuilds, E->dbg_cnt_rebuild_slots += (uint64_t)1 << bits, E->dbg_cnt_rebuild_max = std::max<uint64_t>(E->dbg_cnt_rebuild_max, n_mz);
                } else {
                    memcpy(key_stage, b.cnt_rem.data(), b.cnt_rem.size() * sizeof(uint64_t));
                    j.rem = key_stage, j.n_rem = (uint32_t)b.cnt_rem.size(), key_stage += b.cnt_rem.size();
                    memcpy(key_stage, b.cnt_add.data(), b.cnt_add.size() * sizeof(uint64_t));
                    j.add = key_stage, j.n_add = (uint32_t)b.cnt_add.size(), key_stage += b.cnt_add.size();
                    b.cnt_keys += b.cnt_add.size();
                    ++E->dbg_cnt_updates, E->dbg_cnt_keys += j.n_rem + j.n_add;
                }
                j.tab = b.d_cnt.as<CountSlot>(), j.bits = b.cnt_bits;
                j.hist = b.d_cnt_hm.as<uint32_t>(), j.meta = b.d_cnt_hm.as<uint32_t>() + 1024;
                if (j.rebuild || j.n_rem || j.n_add) cj.push_back(j);
                b.cnt_rem.clear(), b.cnt_add.clear();
            }
            AB.reqs[w].ref_cnt = b.d_cnt.as<CountSlot>(), AB.reqs[w].ref_cnt_meta = b.d_cnt_hm.as<uint32_t>() + 1024, AB.reqs[w].ref_cnt_bits = b.cnt_bits;
        }
        if (!cj.empty()) {
            memcpy(L.pin_cnt.p, cj.data(), cj.size() * sizeof(CountJob));
            NS_TRY(gpu_count_tables_launch(SW.stream, L.pin_cnt.as<CountJob>(), (uint32_t)cj.size(), mm2::Opt().mid_occ_frac));
        }
        static const bool check_cnt = getenv("NSGPU_SKETCH_CHECK") != nullptr;
        if (check_cnt) {
            // every table against the counts of the host's list: the same hashes with the same counts, no other hash with a count, the
            // same number of distinct hashes and the host's mid_occ (mm2.cpp RefIndex::build_from_sketch)
            NS_HIP(hipStreamSynchronize(SW.stream));
            for (size_t w = 0; w < n; ++w) {
                Builder &b = D.B[who[w]];
                std::vector<CountSlot> tab((size_t)1 << b.cnt_bits);
                uint32_t meta[4];
                NS_HIP(hipMemcpy(tab.data(), b.d_cnt.p, tab.size() * sizeof(CountSlot), hipMemcpyDeviceToHost));
                NS_HIP(hipMemcpy(meta, b.d_cnt_hm.as<uint32_t>() + 1024, sizeof(meta), hipMemcpyDeviceToHost));
                std::vector<uint64_t> h(b.mz.size());
                for (size_t i = 0; i < h.size(); ++i) h[i] = b.mz[i].x >> 8;
                std::sort(h.begin(), h.end());
                std::vector<uint32_t> occ;
                size_t n_bad = 0, n_live = 0;
                for (size_t i = 0; i < h.size();) {
                    size_t j = i;
                    while (j < h.size() && h[j] == h[i]) ++j;
                    occ.push_back((uint32_t)(j - i));
                    uint32_t sl = (uint32_t)(((h[i] + 1) * 0x9e3779b97f4a7c15ull) >> (64 - b.cnt_bits)), cnt = 0;
                    for (;; sl = (sl + 1) & ((1u << b.cnt_bits) - 1)) { if (tab[sl].key == h[i] + 1) { cnt = tab[sl].count; break; } if (!tab[sl].key) break; }
                    n_bad += cnt != j - i;
                    i = j;
                }
                for (const CountSlot &t : tab) n_live += t.key && t.count;
                uint32_t mid = 1;
                if (!occ.empty()) {
                    const size_t kk = (uint32_t)((1. - mm2::Opt().mid_occ_frac) * occ.size());
                    std::nth_element(occ.begin(), occ.begin() + kk, occ.end());
                    mid = occ[kk] + 1;
                }
                if (n_bad || n_live != occ.size() || meta[0] != occ.size() || (meta[1] != mid && !meta[2])) {
                    fprintf(stderr, "nsgpu: a contig's count table differs from its minimizer list's counts (internal error): %zu wrong counts, %zu / %u / %zu hashes, mid_occ %u / %u\n",
                            n_bad, n_live, meta[0], occ.size(), meta[1], mid);
                    abort();
                }
            }
        }
    }
    if (use_dev_plan) NS_TRY(engine_cons_update(c, E, AB, who, changed, staged, L));
    else for (size_t w = 0; w < n; ++w) if (changed[w]) D.B[who[w]].dc_valid = false;          // (a batch that does not update the device copies leaves them stale)
    AB.defer_anchors = n_groups(c) == 1 && c->defer_slots ? c->defer_anchors : 0;
    NS_TRY(align_prestep_launch(c, AB, 0, n, sws_i, true, use_dev_plan ? dp_ws : -1));
    L.sk_ms[4] += now_ms() - tk; tk = now_ms();
    NS_TRY(align_prestep_finish(c, AB, 0, n, sws_i));
    if (!AB.deferred.empty()) NS_TRY(engine_defer_start(c, E, AB, who));
    L.sk_ms[5] += now_ms() - tk;
    E->awho[gi] = who;
    E->dbg_batch_sizes.push_back((uint32_t)who.size());
    { std::lock_guard<std::mutex> lk(c->stat_m); S.index_ms += now_ms() - g0; }
    return NSGPU_OK;
}

// batches, part 1: seeds / chains / DP plan of the alignments sketched in the stage before, and the launch of the DP kernels
// -- which stay in flight until part 2
static int engine_batches_begin(nsgpu_ctx *c, int group, int ws_index)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    nsgpu_consensus_stats &S = c->cons_stats;
    const int gi = group < 0 ? 0 : group;
    if (E->awho[gi].empty()) return NSGPU_OK;
    const double g1 = now_ms();
    AlignBatch &AB = E->ab[gi];
    NS_TRY(align_begin(c, AB, ws_index));
    { std::lock_guard<std::mutex> lk(c->stat_m); S.align_ms += now_ms() - g1; E->p1_align_ms += now_ms() - g1; E->p1_host_ms += AB.host_ms; E->p1_launch_ms += AB.dp_ms; }
    return NSGPU_OK;
}


// batches, part 2: the group's window queries; DP results, execution of the alignment skeletons, alignRead's conversion (the
// builders become ALIGNED)
static int engine_window_queries(nsgpu_ctx *c, int group)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    nsgpu_consensus_stats &S = c->cons_stats;
    {   // the group's window queries (builders that wait for a candidate list): independent of its alignments, done in this
        // part because part 1 is the longer one
        std::vector<uint32_t> &who = E->fwho;
        who.clear();
        for (Builder &b : D.B) if (in_group(b, group) && b.st == Builder::WAIT_FILTER) who.push_back(b.id);
        if (!who.empty() && !E->sp.wr_off.empty()) {
            // A fresh contig's first window is its whole seed read, and the filter is a function of the query string and the tables: the
            // answer was computed for every read when the seed buckets were built.  (Most seed rounds start a few contigs and nothing else
            // waits for a window then: the slot saves a GPU round trip.)
            const Engine::SeedPolicy &P = E->sp;
            size_t keep = 0, n_tab = 0;
            for (uint32_t bi : who) {
                Builder &b = D.B[bi];
                bool from_table = false;
                if (b.g && b.g->num_reads() == 0 && b.g->first_read >= D.id_base) {
                    const read_t r = b.g->first_read - D.id_base;
                    if (r < D.N && b.win[0].size() == D.read_len(r) && memcmp(b.win[0].data(), D.read_ptr(r), b.win[0].size()) == 0) {
                        for (int sd = 0; sd < 2; ++sd) b.cand[sd].assign(P.wr_ids.begin() + (ptrdiff_t)P.wr_off[2 * (size_t)r + sd], P.wr_ids.begin() + (ptrdiff_t)P.wr_off[2 * (size_t)r + sd + 1]);
                        b.st = Builder::GOT_FILTER;
                        from_table = true, ++n_tab;
                    }
                }
                if (!from_table) who[keep++] = bi;
            }
            who.resize(keep);
            if (n_tab) { std::lock_guard<std::mutex> lk(c->stat_m); S.n_windows += n_tab; }
        }
        if (!who.empty()) {
            const double f0 = now_ms();
            E->qbuf.clear(); E->qoff.assign(1, 0);
            for (uint32_t bi : who) for (int s = 0; s < 2; ++s) { E->qbuf += D.B[bi].win[s]; E->qoff.push_back(E->qbuf.size()); }
            const uint32_t nq = (uint32_t)(2 * who.size());
            // one launch sequence, one wait, candidate lists straight into pinned memory (run_window_queries_fast); the exact
            // multi-step path when a buffer sized in advance did not fit (it grows them), or with NSGPU_WQ_EXACT=1
            static const bool wq_exact = getenv("NSGPU_WQ_EXACT") != nullptr;
            const uint64_t *foff = nullptr;
            const uint32_t *fids = nullptr;
            bool redo = wq_exact;
            if (!wq_exact) NS_TRY(run_window_queries_fast(c, E->qbuf.data(), E->qoff.data(), nq, foff, fids, &redo));
            if (redo) {
                c->filter_stats = false;                       // nobody reads the match totals of the engine's window queries
                const int frc = filter_strings_device(c, E->qbuf.data(), E->qoff.data(), nq);
                c->filter_stats = true;
                NS_TRY(frc);
                NS_TRY(c->pin_foff.reserve(((size_t)nq + 1) * 8));
                NS_TRY(c->pin_fids.reserve((c->f_total + 1) * 4));
                NS_HIP(hipMemcpyAsync(c->pin_foff.p, c->f_off.p, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost, c->stream));
                if (c->f_total) NS_HIP(hipMemcpyAsync(c->pin_fids.p, c->f_ids.p, c->f_total * 4, hipMemcpyDeviceToHost, c->stream));
                NS_HIP(stream_wait_short(c->stream));
                foff = c->pin_foff.as<uint64_t>(), fids = c->pin_fids.as<uint32_t>();
                ++E->n_wq_exact;
            }
            E->foff.assign(foff, foff + nq + 1);
            E->fids.assign(fids, fids + c->f_total);
            for (size_t w = 0; w < who.size(); ++w) {
                Builder &b = D.B[who[w]];
                for (int s = 0; s < 2; ++s) b.cand[s].assign(E->fids.begin() + E->foff[2 * w + s], E->fids.begin() + E->foff[2 * w + s + 1]);
                b.st = Builder::GOT_FILTER;
            }
            { std::lock_guard<std::mutex> lk(c->stat_m); S.filter_ms += now_ms() - f0; S.n_windows += who.size(); ++S.n_filter_rounds; }
        }
    }
    return NSGPU_OK;
}

// ONE group: the first part of a two-part batch's DP results is there while the late kernel classes still run (align_finish_early).  The
// builders whose alignment is final take their result now, and those whose claim cannot fail -- the alignment succeeded, nobody else in the
// batch aligns the same read, the read is unclaimed (claims and seed grants are only ever written between slots) -- update their graph and
// consensus at once: the slot's host phase, which used to wait for the slowest DP problem, is left with the stragglers.  Same result: the
// update touches the builder's own graph only, and the claim it anticipates is the one the slot's end resolves.
static int engine_early_updates(nsgpu_ctx *c, const int *gis, int n_gi)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    // the batches this call serves: those with alignments and results in two parts
    int act[kMaxGroups], n_act = 0;
    for (int k = 0; k < n_gi; ++k) if (!E->awho[gis[k]].empty() && E->ab[gis[k]].plan_two_part) act[n_act++] = gis[k];
    if (!n_act) return NSGPU_OK;
    const double g0 = now_ms();
    // contested reads: two builders of the slot align the same one (the lower builder's claim decides) -- over ALL the slot's batches
    std::vector<uint32_t> &pends = E->early_pends;
    pends.clear();
    for (int k = 0; k < n_gi; ++k) for (uint32_t bi : E->awho[gis[k]]) pends.push_back(D.B[bi].pend);
    for (const Builder &b : D.B) if (b.st == Builder::DEFERRED && b.defer_due == E->cur_slot) pends.push_back(b.pend);      // (deferred alignments that claim with this slot's)
    // (several ranks: the other ranks' builders count as well -- their candidate reads came with the slot's exchange, run_consensus_dist)
    const bool global_known = E->world == 1 || E->have_global_pends;
    std::vector<uint32_t> sorted(E->world == 1 || !E->have_global_pends ? pends : E->global_pends);
    if (E->world == 1 || !E->have_global_pends) std::sort(sorted.begin(), sorted.end());
    E->have_global_pends = false;
    // builder -> its batch and its request in it, and whether its claim cannot fail
    std::vector<int32_t> &widx = E->early_widx;
    std::vector<int8_t> &lane_of = E->early_lane;
    widx.assign(D.B.size(), -1);
    lane_of.assign(D.B.size(), -1);
    for (int k = 0; k < n_act; ++k) {
        const int gi = act[k];
        const std::vector<uint32_t> &who = E->awho[gi];
        Engine::Lane &L = E->lane[gi];
        AlignBatch &AB = E->ab[gi];
        const size_t n = who.size();
        if (L.outs.size() < n) L.outs.resize(n);
        AB.early_done.assign(n, 0);
        L.early_sure.assign(n, 0);
        for (size_t w = 0; w < n; ++w) {
            widx[who[w]] = (int32_t)w, lane_of[who[w]] = (int8_t)gi;
            const uint32_t pend = D.B[who[w]].pend;
            const auto range = std::equal_range(sorted.begin(), sorted.end(), pend);
            L.early_sure[w] = range.second - range.first == 1 && !D.in_graph[pend] && global_known;
        }
    }
    // One part of the results (0: the alignments without a problem in a late class, there behind the bulk classes; 1: the others): ONE task per
    // builder, on the thread its graph lives with -- its problems' results into its job, the skeleton to the end, the conversion, and when
    // its claim cannot fail the graph update.  The two parts touch disjoint builders; the second part's thread starts its loop when the late
    // classes are done, some 0.2 ms behind the first, and the two loops share the pool: a builder of the second part is not held up by the
    // first part's slowest update (it used to wait for that, then for the claims, and was updated in the next host phase).
    std::atomic<uint64_t> n_updates{0}, n_tasks{0}, n_retry{0};
    std::atomic<uint64_t> task_ns{0}, task_max_ns{0}, conv_ns{0};
    struct Tk { std::atomic<uint64_t> &sum, &mx; double k0; ~Tk() { const uint64_t d = (uint64_t)((now_ms() - k0) * 1e6); sum += d; uint64_t m = mx.load(); while (d > m && !mx.compare_exchange_weak(m, d)) {} } };
    // a builder whose alignment has been delivered: the skeleton to its end, the conversion, and -- its claim cannot fail -- the graph update
    // the graphs in HBM of the slot's builders: ONE launch (behind the DP kernels' launches) whose workgroups wait for the accepted reads' scripts;
    // whatever gets none -- an alignment that failed, a contested read -- is released when this function is left
    std::vector<DevGraph *> armed;
    struct Disarm { std::vector<DevGraph *> &v; ~Disarm() { for (DevGraph *g : v) if (g->armed()) g->cancel(); } } disarm{armed};
    static const bool solo = getenv("NSGPU_GRAPH_SOLO") != nullptr;      // tests: no launch for a whole slot, every update a launch of its own
    if (D.gsh && !solo) {
        for (int k = 0; k < n_act; ++k)
            for (uint32_t bi : E->awho[act[k]]) {
                Builder &b = D.B[bi];
                DevGraph *dg = b.g ? b.g->dev() : nullptr;
                if (!dg) continue;
                NS_TRY(dg->prepare(b.query.size()));
                armed.push_back(dg);
            }
        NS_TRY(graph_serve_launch(D.gsh, armed.data(), armed.size()));
    }
    // the second half of a builder's task: the graph has taken the read -- the new consensus, and which stretch of it has to be sketched again
    auto finish_update = [&](Builder &b) {
        const double t0 = now_ms();
        D.apply_complete(b);
        plan_splice(b, (int)c->prm.m_w, (int)c->prm.m_k);
        b.sp_ready = true;
        b.early_updated = true;
        b.cpu_ms += now_ms() - t0;
        n_updates += 1;
    };
    auto finish_builder = [&](size_t i, int gi, size_t w, double k0, bool async) {
        AlignBatch &AB = E->ab[gi];
        Engine::Lane &L = E->lane[gi];
        if (!align_early_one(AB, w, L.outs[w])) return;
        conv_ns += (uint64_t)((now_ms() - k0) * 1e6);
        Builder &b = D.B[i];
        std::swap(b.aln, L.outs[w]);
        b.early_result = true;
        if (!(b.aln.ok && L.early_sure[w])) return;
        const double t0 = now_ms();
        D.apply_submit(b);                            // (the graph in HBM: a kernel on its way; the pointer graph: done)
        b.cpu_ms += now_ms() - t0;
        if (!async || !b.graph_flying || b.g->ready()) finish_update(b);
    };
    // The DP kernels hand every alignment over the moment its last problem is done (ksw_collect.hpp): no parts -- every pool thread keeps looking at
    // the status words of ITS builders' alignments (the pinned assignment of the host phase) and runs a builder's task as soon as its word is
    // up, sleeping 10 us when nothing of its own is ready; the batches' closing words end the watch for whatever was not handed over (a full
    // CIGAR arena: align_finish asks for those problems again).  An alignment whose longest problem is short is finished and applied while the
    // launch's longest problems are still running.
    KswDevResults Rp[kMaxGroups];
    const volatile uint32_t *donep[kMaxGroups];
    const PlanOut *pop[kMaxGroups];
    bool polled[kMaxGroups];
    int n_polled = 0;
    for (int gi = 0; gi < kMaxGroups; ++gi) polled[gi] = false, donep[gi] = nullptr, pop[gi] = nullptr;
    for (int k = 0; k < n_act; ++k) {
        const int gi = act[k];
        AlignBatch &AB = E->ab[gi];
        if (AB.plan_ws >= 0 && ksw_dev_poll(c, AB.plan_ws, Rp[gi], donep[gi])) { polled[gi] = true, pop[gi] = AB.plan_out.as<PlanOut>(), ++n_polled; }
    }
    if (n_polled) {
        const size_t T = std::max<size_t>(1, (size_t)host_threads());
        const double p0 = now_ms();
        // Every builder of the watch has ONE owner thread (the pinned assignment of the host phase: its graph lives in that thread's caches); a thread
        // whose own builders are all done does not leave, though: it takes over whatever alignment is handed over while its owner is busy with
        // another builder's update -- the last results of a slot used to queue up behind each other on one thread while fifteen were idle.
        // A builder is run by whoever sets its claim flag first (NSGPU_CONS_NO_STEAL=1: owners only, the A/B switch).
        static const bool no_steal = getenv("NSGPU_CONS_NO_STEAL") != nullptr;
        std::vector<uint32_t> &cand = E->early_cand;
        cand.clear();
        for (size_t i = 0; i < D.B.size(); ++i) {
            const int32_t w = widx[i];
            const int gi = lane_of[i];
            if (w < 0 || gi < 0 || !polled[gi]) continue;
            const AlignBatch &AB = E->ab[gi];
            if ((size_t)w >= AB.plan_pair.size() || AB.plan_pair[w] == ~0u || AB.plan_delivered[w]) continue;
            const PlanOut o = pop[gi][AB.plan_pair[w]];
            if (o.flags || o.n_tasks == 0) continue;
            cand.push_back((uint32_t)i);
        }
        if (E->early_claim.size() < D.B.size()) E->early_claim = std::vector<std::atomic<uint8_t>>(D.B.size());
        for (uint32_t i : cand) E->early_claim[i].store(0, std::memory_order_relaxed);
        std::atomic<uint32_t> n_open{(uint32_t)cand.size()};      // builders of the watch nobody has taken yet
        std::atomic<uint64_t> n_stolen{0}, last_claim_us{0};
        par_for_pinned("host.early", T, [&](size_t t) {
            static thread_local const int slack_set = prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);      // (a 10 us sleep is 10 us, not 60)
            (void)slack_set;
            static thread_local std::vector<uint32_t> mine;          // (any number of builders per thread: few threads, many builders)
            static thread_local std::vector<uint32_t> flying;        // builders whose graph update this thread has launched and not yet taken over
            flying.clear();
            mine.clear();
            for (uint32_t i : cand) if (i % T == t) mine.push_back(i);
            size_t n_mine = mine.size();
            // one builder whose status word is up: claimed, delivered, finished.  0: not ready (or its data not all there yet), 1: taken by this call, 2: by somebody else
            auto try_builder = [&](size_t i) -> int {
                const int32_t w = widx[i];
                const int gi = lane_of[i];
                AlignBatch &AB = E->ab[gi];
                if (E->early_claim[i].load(std::memory_order_acquire)) return 2;
                const uint32_t st = __atomic_load_n(&Rp[gi].status[AB.plan_pair[w]], __ATOMIC_ACQUIRE);
                if (st == 0) return 0;
                uint8_t zero = 0;
                if (!E->early_claim[i].compare_exchange_strong(zero, 1, std::memory_order_acq_rel)) return 2;
                if (st == 1u && !AB.plan_delivered[w]) {
                    // (the status word is up: is everything it announces here?  If not, look again in a moment)
                    const uint32_t got = batch_plan_deliver_one(AB, Rp[gi], (size_t)w, 1, false);
                    if (got == ~0u) { n_retry += 1; E->early_claim[i].store(0, std::memory_order_release); return 0; }
                    n_tasks += got;
                }
                n_open -= 1;
                if (st != 1u) return 1;                                       // (not handed over: align_finish's rounds)
                const double k0 = now_ms();
                { uint64_t m = last_claim_us.load(); const uint64_t v = (uint64_t)((k0 - p0) * 1e3); while (v > m && !last_claim_us.compare_exchange_weak(m, v)) {} }
                Tk tk{task_ns, task_max_ns, k0};
                finish_builder(i, gi, (size_t)w, k0, true);
                if (D.B[i].graph_flying) flying.push_back((uint32_t)i);
                return 1;
            };
            // graph updates in flight: whichever has reported is taken over (the consensus patched, the splice planned)
            auto poll_flying = [&]() -> bool {
                bool any = false;
                for (size_t k = 0; k < flying.size();) {
                    Builder &b = D.B[flying[k]];
                    if (!b.g->ready()) { ++k; continue; }
                    const double k0 = now_ms();
                    Tk tk{task_ns, task_max_ns, k0};
                    finish_update(b);
                    flying[k] = flying.back(), flying.pop_back();
                    any = true;
                }
                return any;
            };
            bool closing = false;
            // (a batch that never closes -- a GPU fault -- must not hold the pool for ever: align_finish's wait reports it)
            static const double give_up_ms = [] { const char *e = getenv("NSGPU_WAIT_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return (v > 0 ? v : 120.0) * 1e3; }();
            while (n_mine || !flying.empty() || (!no_steal && n_open.load(std::memory_order_acquire))) {
                if (now_ms() - p0 > give_up_ms) break;
                bool progressed = poll_flying();
                for (size_t k = 0; k < n_mine;) {
                    const int r = try_builder(mine[k]);
                    if (r == 0) { ++k; continue; }
                    mine[k] = mine[--n_mine];
                    progressed = true;
                }
                if (!n_mine && !no_steal) {
                    // nothing of its own left: the others' builders, one at a time (then back to the top: the closing test)
                    for (uint32_t i : cand) {
                        if (i % T == t) continue;
                        const int r = try_builder(i);
                        if (r == 1) { progressed = true; n_stolen += 1; break; }
                    }
                }
                if (progressed) continue;
                if (closing && flying.empty()) break;                         // one more look after the closing words, then leave
                bool all_done = true;
                for (int gi = 0; gi < kMaxGroups; ++gi) if (polled[gi] && !*donep[gi]) all_done = false;
                if (all_done && !closing) { closing = true; continue; }
                timespec ts = {0, 10000};
                nanosleep(&ts, nullptr);
            }
            // (a watch that gave up: nothing may stay in flight behind it)
            for (uint32_t i : flying) finish_update(D.B[i]);
            flying.clear();
        });
        E->n_early_stolen += n_stolen.load();
        E->early_last_claim_ms += last_claim_us.load() / 1e3;       // (debug print: when the slot's last alignment was taken up, from the watch's start)
        E->early_part_ms[0] += now_ms() - p0;
        E->n_early_retry += n_retry.load();
    }
    // batches whose results come from collecting kernels behind the launches (NSGPU_KSW_NO_INLINE_COLLECT=1): part 0 here, part 1 on a second thread
    int rc_all = NSGPU_OK;
    for (int k = 0; k < n_act && rc_all == NSGPU_OK; ++k) {
        const int gi = act[k];
        if (polled[gi]) continue;
        AlignBatch &AB = E->ab[gi];
        auto run_part = [&](int part, const KswDevResults &R) {
            const double p0 = now_ms();
            struct Fin { Engine *E; int part; double p0; ~Fin() { E->early_part_ms[part] += now_ms() - p0; } } fin{E, part, p0};
            // (pinned like the host phase: a builder's graph stays with one thread's caches)
            par_for_pinned("host.early", D.B.size(), [&](size_t i) {
                const int32_t w = widx[i];
                if (w < 0 || lane_of[i] != gi) return;
                // (only what THIS call delivered is touched: the other part's builders are the other thread's)
                const double k0 = now_ms();
                Tk tk{task_ns, task_max_ns, k0};
                const uint32_t got = batch_plan_deliver_one(AB, R, (size_t)w, part, true);
                if (!got || got == ~0u) return;
                n_tasks += got;
                finish_builder(i, gi, (size_t)w, k0, false);
            });
        };
        int rc1 = NSGPU_OK;
        std::string err1;
        std::thread t1;
        // (the second part's wait begins once the first part has been waited for: the two never race for the workspace's state)
        KswDevResults R0;
        const int rc0 = batch_plan_wait(c, AB, 0, R0);
        if (rc0 == NSGPU_OK)
            t1 = std::thread([&] {
                pool_bind_this_thread();
                KswDevResults R1;
                rc1 = hipSetDevice(c->prm.device) == hipSuccess ? batch_plan_wait(c, AB, 1, R1) : NSGPU_ERR_HIP;
                if (rc1 != NSGPU_OK) err1 = nsgpu_last_error();
                else if (R1.res) run_part(1, R1);
            });
        if (rc0 == NSGPU_OK && R0.res) run_part(0, R0);
        if (t1.joinable()) t1.join();
        if (rc0 != NSGPU_OK) rc_all = rc0;
        else if (rc1 != NSGPU_OK) { set_error("%s", err1.empty() ? "contig engine: the second part of the results failed" : err1.c_str()); rc_all = rc1; }
    }
    if (rc_all != NSGPU_OK) return rc_all;
    if (D.graph_rc.load() != NSGPU_OK) { set_error("%s", D.graph_err.c_str()); return D.graph_rc.load(); }
    E->n_early += n_updates.load();
    E->early_task_ms += task_ns.load() / 1e6, E->early_task_max_ms += task_max_ns.load() / 1e6, E->early_conv_ms += conv_ns.load() / 1e6;
    { std::lock_guard<std::mutex> lk(c->stat_m); c->aln_dp_tasks += n_tasks.load(); c->cons_stats.graph_ms += now_ms() - g0; }
    E->early_ms += now_ms() - g0;
    return NSGPU_OK;
}
static int engine_early_updates(nsgpu_ctx *c, int group) { const int gi = group < 0 ? 0 : group; return engine_early_updates(c, &gi, 1); }

static int engine_align_finish(nsgpu_ctx *c, int group)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    nsgpu_consensus_stats &S = c->cons_stats;
    const int gi = group < 0 ? 0 : group;
    std::vector<uint32_t> &who = E->awho[gi];
    if (who.empty()) return engine_defer_deliver(c, E);
    Engine::Lane &L = E->lane[gi];
    const double g1 = now_ms();
    NS_TRY(align_finish(c, E->ab[gi], L.outs));
    NS_TRY(engine_defer_deliver(c, E));
    for (size_t w = 0; w < who.size(); ++w) {
        Builder &b = D.B[who[w]];
        if (b.st == Builder::DEFERRED) continue;                      // (its alignment left the batch: engine_defer_start)
        if (!b.early_result) std::swap(b.aln, L.outs[w]);              // the builder's previous result goes back into the pool of result objects
        b.early_result = false;
        ++b.n_align_calls;
        b.accepted = false;
        b.st = Builder::ALIGNED;
    }
    who.clear();
    E->ab[gi].reqs.clear();
    { std::lock_guard<std::mutex> lk(c->stat_m); S.align_ms += now_ms() - g1; ++S.n_align_rounds; }
    return NSGPU_OK;
}

static int engine_batches_finish(nsgpu_ctx *c, int group)
{
    NS_TRY(engine_window_queries(c, group));
    return engine_align_finish(c, group);
}

int engine_batches(nsgpu_ctx *c, int group)
{
    NS_TRY(engine_batches_sketch(c, group, 1));
    NS_TRY(engine_batches_begin(c, group, 1));
    return engine_batches_finish(c, group);
}

int engine_finish(nsgpu_ctx *c, uint32_t n_threads_out)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    NS_CHECK(E && n_threads_out >= 1, NSGPU_ERR_ARG, "consensus: nothing to finish");
    Driver &D = E->D;
    nsgpu_consensus_stats &S = c->cons_stats;
    // merge builders into the requested number of output "threads" (Compressor expects exactly numThr
    // file sets, src/Compressor.cpp:123-124; Decompressor reads numThr from metaData)
    c->cons_out.assign(n_threads_out, cons::StreamSet());
    c->cons_n_deferred = E->n_deferred;
    pool_drain();                                 // all edit emissions
    for (size_t i = 0; i < D.B.size(); ++i)
        for (auto &fc : D.B[i].contigs) c->cons_out[i * n_threads_out / (D.B.empty() ? 1 : D.B.size())].append(fc->out);
    double dbg_w[6] = {0, 0, 0, 0, 0, 0}, dbg_x[4] = {0, 0, 0, 0};
    uint64_t dbg_c[6] = {0, 0, 0, 0, 0, 0};
    for (Builder &b : D.B) {
        S.n_contigs += b.n_contigs; S.n_lone += b.n_lone; S.count_minhash += b.n_minhash; S.count_minhash_not_in_graph += b.n_minhash_new;
        S.count_aligner += b.n_aligner; S.n_align_calls += b.n_align_calls;
        S.graph_cpu_ms += b.cpu_ms; if (b.max_ms > S.graph_max_ms) S.graph_max_ms = b.max_ms;
        for (auto &fc : b.contigs) S.write_cpu_ms += fc->write_ms, dbg_w[1] += fc->write_ms, dbg_w[2] += fc->free_ms;
        dbg_w[0] += b.dbg_w1, dbg_w[3] += b.dbg_u, dbg_w[4] += b.dbg_m, dbg_w[5] += b.dbg_cyc;
        for (int k = 0; k < 6; ++k) dbg_c[k] += b.dbg_c[k];
        dbg_x[0] += b.dbg_init, dbg_x[1] += b.dbg_rc, dbg_x[2] += b.dbg_win, dbg_x[3] += b.dbg_start;
    }
    if (getenv("NSGPU_CONS_DEBUG")) {
        double mu = 0, mm = 0, lm = 0; uint64_t ln = 0;
        for (Builder &b : D.B) { if (b.dbg_max_u > mu) mu = b.dbg_max_u; if (b.dbg_max_m > mm) mm = b.dbg_max_m; lm += b.dbg_long_ms; ln += b.dbg_long_n; }
        fprintf(stderr, "[cons] longest update_graph %.1f ms, longest main path %.1f ms; builder steps > 3 ms: %llu, %.0f ms in total; crit %.0f ms (slowest update_graph per phase, summed: %.0f; slowest main path: %.0f)\n", mu, mm, (unsigned long long)ln, lm, S.graph_crit_ms, static_cast<Engine *>(c->cons_engine)->crit_u_ms, static_cast<Engine *>(c->cons_engine)->crit_m_ms);
    }
    if (getenv("NSGPU_CONS_DEBUG")) fprintf(stderr, "[cons] cpu-ms: graph total %.0f; initialize+first main path %.0f, query copy/revcomp %.0f, open_window %.0f, start_contig %.0f\n", S.graph_cpu_ms, dbg_x[0], dbg_x[1], dbg_x[2], dbg_x[3]);
    if (getenv("NSGPU_CONS_DEBUG"))
        fprintf(stderr, "[cons] cpu-ms: update_graph %.0f main_path %.0f (remove_cycles %.0f) write_main %.0f write_reads %.0f graph_free %.0f; main-path calls %llu cycles-skipped %llu detours %llu cycle-scans-without-split %llu walked-nodes %llu cycle-scans-from-list %llu\n",
                dbg_w[3], dbg_w[4], dbg_w[5], dbg_w[0], dbg_w[1], dbg_w[2], (unsigned long long)dbg_c[0], (unsigned long long)dbg_c[1], (unsigned long long)dbg_c[2], (unsigned long long)dbg_c[3], (unsigned long long)dbg_c[4], (unsigned long long)dbg_c[5]);
    S.total_ms = now_ms() - E->t0;
    c->cons_n_reads_out = 0;
    for (auto &t : c->cons_out) for (read_t x : t.reads_in_contig) c->cons_n_reads_out += x;
    c->have_cons = true;
    c->cons_engine_free(c->cons_engine);
    c->cons_engine = nullptr;
    return NSGPU_OK;
}

// ---- ONE group: the slot's batch -------------------------------------------------------------------------------------------------------------
// (Measured in round 5 and removed: the slot's alignments in TWO batches -- the builders that know their next alignment right after the host
// phase launched at once on a thread of their own, the windows and seeds in their shadow, the builders those release as a second batch.  Same
// schedule, same streams, but slower: 7.86 / 8.73 against 7.11 / 7.26 s per cfg2 step.  A batch's front is latency, not volume -- the second
// batch's sketch .. DP launch took as long as the whole batch's (1.16 against 1.19 ms) and still started behind the windows and the seeds, so
// the slot's critical path kept every step it had, and the two batches' kernels got in each other's way: seeds + chaining 0.70 instead of
// 0.48 ms per launch, windows 0.52 instead of 0.37 ms.)
static int engine_g1_batches(nsgpu_ctx *c)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    double t0 = now_ms(), t1;
    NS_TRY(engine_batches_sketch(c, -1, 1));
    t1 = now_ms(); E->g1_ms[3] += t1 - t0; t0 = t1;
    NS_TRY(engine_batches_begin(c, -1, 1));
    t1 = now_ms(); E->g1_ms[4] += t1 - t0; t0 = t1;
    NS_TRY(engine_early_updates(c, -1));
    t1 = now_ms(); E->g1_ms[6] += t1 - t0; t0 = t1;
    NS_TRY(engine_align_finish(c, -1));
    E->g1_ms[7] += now_ms() - t0;
    return NSGPU_OK;
}

// One pipeline slot: the host phase of group h = slot % G, part 1 (sketches + index + seeds / chains, then the DP launch) of
// the group that was there one slot earlier, part 2 of group (slot + 1) % G -- whose DP launch has been in flight for a
// slot -- all concurrently.
// part: 0 = the whole slot; with ONE group the seeds are granted between the host phase (part 1) and the batches (part 2): see run_consensus
static int engine_slot_inner(nsgpu_ctx *c, uint32_t slot, int part);
int engine_slot(nsgpu_ctx *c, uint32_t slot, int part)
{
    const int rc = engine_slot_inner(c, slot, part);
    Driver &D = static_cast<Engine *>(c->cons_engine)->D;
    if (rc == NSGPU_OK && D.graph_rc.load() != NSGPU_OK) { set_error("%s", D.graph_err.c_str()); return D.graph_rc.load(); }     // (a graph update that failed inside a pool loop)
    return rc;
}
static int engine_slot_inner(nsgpu_ctx *c, uint32_t slot, int part)
{
    const uint32_t G = (uint32_t)n_groups(c);
    const int host_group = (int)(slot % G), begin_group = (int)((slot + G - 1) % G), finish_group = (int)((slot + 1) % G);
    const int ws_index = 1 + (int)(slot % 3);
    static_cast<Engine *>(c->cons_engine)->cur_slot = slot;
    static const bool serial = getenv("NSGPU_NO_OVERLAP") != nullptr;      // debugging aid: one after the other
    if (G <= 2) {
        // One or two groups (nsgpu_set_schedule): a builder's step takes G slots instead of four -- the schedule for few builders,
        // where the length of a slot is the latency of its GPU round trips and not its volume.  G = 1: host phase, then the batches
        // (the window queries on a second thread beside sketches / seeds / DP launch), then the DP results.  G = 2: the same chain
        // for the group that had its host phase in the slot before, beside the host phase of the other group.
        struct Rebind {
            cpu_set_t old; bool ok;
            Rebind() { ok = pthread_getaffinity_np(pthread_self(), sizeof(old), &old) == 0; pool_bind_this_thread(); }
            ~Rebind() { if (ok) (void)pthread_setaffinity_np(pthread_self(), sizeof(old), &old); }
        } rebind;
        int rc1 = NSGPU_OK, rc2 = NSGPU_OK;
        std::string err1, err2;
        auto chain = [&] {
            std::thread tw;
            auto wq = [&] {
                rc2 = hipSetDevice(c->prm.device) == hipSuccess ? engine_window_queries(c, finish_group) : NSGPU_ERR_HIP;
                if (rc2 != NSGPU_OK) err2 = nsgpu_last_error();
            };
            // NSGPU_NO_OVERLAP=1 (debugging aid): the same steps in the same order on this thread -- the schedule, hence the result, is the same
            if (serial) wq();
            else tw = std::thread([&] { pool_bind_this_thread(); wq(); });
            Engine *E = static_cast<Engine *>(c->cons_engine);
            double t0 = now_ms(), t1;
            rc1 = engine_batches_sketch(c, begin_group, ws_index);
            t1 = now_ms(); E->g1_ms[3] += t1 - t0; t0 = t1;
            if (rc1 == NSGPU_OK) rc1 = engine_batches_begin(c, begin_group, ws_index);
            if (rc1 != NSGPU_OK) err1 = nsgpu_last_error();
            t1 = now_ms(); E->g1_ms[4] += t1 - t0; t0 = t1;
            if (tw.joinable()) tw.join();
            t1 = now_ms(); E->g1_ms[5] += t1 - t0; t0 = t1;
            if (rc1 == NSGPU_OK && rc2 == NSGPU_OK && G == 1) { rc1 = engine_early_updates(c, finish_group); if (rc1 != NSGPU_OK) err1 = nsgpu_last_error(); }
            t1 = now_ms(); E->g1_ms[6] += t1 - t0; t0 = t1;
            if (rc1 == NSGPU_OK && rc2 == NSGPU_OK) { rc1 = engine_align_finish(c, finish_group); if (rc1 != NSGPU_OK) err1 = nsgpu_last_error(); }
            E->g1_ms[7] += now_ms() - t0;
        };
        if (G == 1) {
            // (part 0: the whole slot in one call, for callers that drive the phases themselves -- its window queries are answered here and
            // consumed by the next slot's host phase, as with more groups)
            if (part != 2) engine_advance(c, false, host_group);
            if (part == 0) NS_TRY(engine_window_queries(c, finish_group));
            if (part != 1) return engine_g1_batches(c);
        }
        else if (serial) { engine_advance(c, false, host_group); chain(); }
        else {
            std::thread t1([&] { pool_bind_this_thread(); if (hipSetDevice(c->prm.device) == hipSuccess) chain(); else rc1 = NSGPU_ERR_HIP; });
            engine_advance(c, false, host_group);
            t1.join();
        }
        if (rc1 != NSGPU_OK) { set_error("%s", err1.empty() ? "contig engine: a batch thread failed" : err1.c_str()); return rc1; }
        if (rc2 != NSGPU_OK) { set_error("%s", err2.empty() ? "contig engine: the window-query thread failed" : err2.c_str()); return rc2; }
        return NSGPU_OK;
    }
    if (serial) {
        engine_advance(c, false, host_group);
        NS_TRY(engine_batches_finish(c, finish_group));
        NS_TRY(engine_batches_sketch(c, begin_group, ws_index));
        return engine_batches_begin(c, begin_group, ws_index);
    }
    // the calling thread works in the host phase's loops: it joins the pool's threads on the GPU's NUMA node for the slot
    struct Rebind {
        cpu_set_t old; bool ok;
        Rebind() { ok = pthread_getaffinity_np(pthread_self(), sizeof(old), &old) == 0; pool_bind_this_thread(); }
        ~Rebind() { if (ok) (void)pthread_setaffinity_np(pthread_self(), sizeof(old), &old); }
    } rebind;
    int rc[3] = {NSGPU_OK, NSGPU_OK, NSGPU_OK};
    std::string role_err[3];              // set_error() is thread-local: a role thread's message is re-issued on the calling thread below
    double d[3] = {0, 0, 0};
    uint64_t ser[3] = {0, 0, 0};          // CPU time of the role threads outside the pool's loops (debug breakdown)
    auto role = [&](int i, const std::function<int()> &fn) {
        pool_bind_this_thread();
        const double x = now_ms();
        const uint64_t c0 = pool_thread_cpu_ns();
        rc[i] = hipSetDevice(c->prm.device) == hipSuccess ? fn() : NSGPU_ERR_HIP;
        if (rc[i] != NSGPU_OK) role_err[i] = nsgpu_last_error();
        d[i] = now_ms() - x;
        ser[i] = pool_thread_cpu_ns() - c0 - pool_thread_work_ns();
    };
    std::thread t1([&] { role(1, [&] { const int r = engine_batches_sketch(c, begin_group, ws_index); return r != NSGPU_OK ? r : engine_batches_begin(c, begin_group, ws_index); }); });
    std::thread t2([&] { role(2, [&] { return engine_batches_finish(c, finish_group); }); });
    const double h0 = now_ms();
    const uint64_t hc0 = pool_thread_cpu_ns(), hw0 = pool_thread_work_ns();
    engine_advance(c, false, host_group);
    ser[0] = pool_thread_cpu_ns() - hc0 - (pool_thread_work_ns() - hw0);
    d[0] = now_ms() - h0;
    t1.join(); t2.join();
    {   // which of the roles set the length of the slot (debug breakdown)
        Engine *E = static_cast<Engine *>(c->cons_engine);
        int w = 0;
        for (int i = 1; i < 3; ++i) if (d[i] > d[w]) w = i;
        for (int i = 0; i < 3; ++i) E->role_serial_ns[i] += ser[i];
        ++E->slot_long_n[w];
        E->slot_long_ms[w] += d[w];
    }
    for (int i = 1; i < 3; ++i)
        if (rc[i] != NSGPU_OK) { set_error("%s", role_err[i].empty() ? "contig engine: a batch thread failed" : role_err[i].c_str()); return rc[i]; }
    return NSGPU_OK;
}

// ONE group: a window costs no slot of its own.  The builders that opened a window get their candidate lists at once and go on (to their
// next alignment request, their next window, or the end of their contig) until nobody waits for a window any more: one GPU round trip per
// pass, typically one pass.  Reads inGraph[] only.
int engine_window_loop(nsgpu_ctx *c, int group)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    for (;;) {
        bool any = false;
        for (const Builder &b : E->D.B) if (in_group(b, group) && b.st == Builder::WAIT_FILTER) { any = true; break; }
        if (!any) return NSGPU_OK;
        NS_TRY(engine_window_queries(c, group));
        engine_advance(c, false, group);                 // (only builders that hold a delivery move: here the ones with a candidate list)
    }
}

static int run_consensus_inner(nsgpu_ctx *c, uint32_t n_builders, uint32_t n_threads_out);
int run_consensus(nsgpu_ctx *c, uint32_t n_builders, uint32_t n_threads_out)
{
    const int rc = run_consensus_inner(c, n_builders, n_threads_out);
    // a failed stage leaves no engine behind (nsgpu_set_schedule would refuse with "a contig stage is in progress")
    if (rc != NSGPU_OK && c->cons_engine) { c->cons_engine_free(c->cons_engine); c->cons_engine = nullptr; }
    return rc;
}
static int run_consensus_inner(nsgpu_ctx *c, uint32_t n_builders, uint32_t n_threads_out)
{
    NS_CHECK(n_threads_out >= 1, NSGPU_ERR_ARG, "n_threads_out must be >= 1");
    NS_TRY(engine_begin(c, n_builders, 0, 1));
    Engine *E = static_cast<Engine *>(c->cons_engine);
    std::vector<uint32_t> ga, gb;
    double w_slot = 0, w_seed = 0, w_claim = 0;
    struct rusage ru0;
    getrusage(RUSAGE_SELF, &ru0);
    const double w_begin = now_ms() - E->t0;
    // G builder groups (4), a G-th of a period apart.  In slot s group h = s % G runs its host phase (graph updates up to
    // the next window / alignment request), group (s + G - 1) % G -- which did that in the slot before -- part 1 of its GPU
    // batches (minimizer sketches, consensus indexes, seeds / chains, launch of the alignment DP), the DP kernels of group
    // (s + 2) % G stay in flight for this slot (their longest problems take about as long as a slot), and group (s + 1) % G
    // runs part 2 (its window queries; DP results, alignment skeletons, edit scripts): the cores are not idle during
    // kernels, nor the GPU during graph work, nor either during the other's bookkeeping.  At the slot boundary the part-2
    // group's read claims and then the host group's seed requests are resolved, in global builder order: the schedule is a
    // function of the data only.
    for (uint32_t slot = 0;; ++slot) {
        const int G = n_groups(c);
        const int h = (int)(slot % G), b = (int)((slot + 1) % G);
        double t = now_ms();
        if (G == 1) {
            // ONE group: host phase (windows are answered at once: engine_window_loop), seeds (the fresh contigs take their first steps
            // and get their first window at once, so that their first alignment is in this slot's batches), batches, claims -- a read
            // granted as a seed here cannot be claimed by an alignment of this slot
            NS_TRY(engine_slot(c, slot, 1));
            double t1 = now_ms();
            E->g1_ms[0] += t1 - t;
            NS_TRY(engine_window_loop(c, h));
            E->g1_ms[1] += now_ms() - t1; t1 = now_ms();
            engine_seed_requests(c, ga, gb, h);
            if (!ga.empty() && engine_seed_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size()) != 0) { engine_advance(c, true, h); NS_TRY(engine_window_loop(c, h)); }
            E->g1_ms[2] += now_ms() - t1;
            NS_TRY(engine_slot(c, slot, 2));
            w_slot += now_ms() - t;
            engine_claim_requests(c, ga, gb, b);
            engine_claim_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size());
            if (E->n_done_global >= E->n_total) break;
            continue;
        }
        NS_TRY(engine_slot(c, slot));
        w_slot += now_ms() - t;
        t = now_ms();
        engine_claim_requests(c, ga, gb, b);
        engine_claim_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size());
        w_claim += now_ms() - t;
        t = now_ms();
        // one round of seeds per slot: a fresh contig that ends at once (nothing to look up) asks again at its group's next
        // boundary -- a multi-GPU driver then needs one collective per slot for both request lists
        // The first steps of the contigs started here (graph of the seed read, first main path, first window) are not run at the
        // boundary, where every other thread would wait for them (170 ms per cfg2 step): the group's next role is part 1 of the
        // batches, which only looks at builders that wait for an alignment, and nobody needs a fresh contig's window request
        // before the group's part 2, two slots on -- they run with the next slot's host phase.
        engine_seed_requests(c, ga, gb, h);
        if (!ga.empty() && engine_seed_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size()) != 0) {
            if (G < 4) engine_advance(c, true, h);      // (two groups: the group's batches run in the very next slot)
            else E->deferred_fresh = h;
        }
        w_seed += now_ms() - t;
        if (E->n_done_global >= E->n_total) break;
    }
    if (getenv("NSGPU_CONS_DEBUG")) debug_report_slots(c, E);
    const double tf = now_ms();
    const int rc = engine_finish(c, n_threads_out);
    if (getenv("NSGPU_CONS_DEBUG")) debug_report_stage(c, E, ru0, w_begin, w_slot, w_seed, w_claim, tf);
    return rc;
}

}  // namespace nsgpu

using namespace nsgpu;

static int give_u32(const std::vector<uint32_t> &v, uint32_t **out)
{
    uint32_t *p = (uint32_t *)malloc((v.size() + 1) * 4);
    NS_CHECK(p, NSGPU_ERR_NOMEM, "malloc failed");
    if (!v.empty()) memcpy(p, v.data(), v.size() * 4);
    *out = p;
    return NSGPU_OK;
}

extern "C" {

// ---- the contig engine phase by phase (multi-GPU jobs drive these and put a collective between the
//      *_requests and *_resolve calls; see nanospring_amd/dist.py) ----
int nsgpu_cons_begin(nsgpu_ctx *c, uint32_t n_builders_total, uint32_t rank, uint32_t world)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(c->sched_groups != 1, NSGPU_ERR_ARG, "nsgpu_cons_begin: the phase calls drive the 2- and 4-group schedules; the one-group schedule (seeds between the host phase and "
                                                  "the batches) runs inside nsgpu_consensus_run / nsgpu_dist_consensus_run");
    NS_HIP(hipSetDevice(c->prm.device));
    return engine_begin(c, n_builders_total, rank, world);
}

int nsgpu_cons_advance(nsgpu_ctx *c, int only_fresh, int group)
{
    NS_CHECK(c && c->cons_engine, NSGPU_ERR_ARG, "nsgpu_cons_advance: call nsgpu_cons_begin first");
    engine_advance(c, only_fresh != 0, group);
    return NSGPU_OK;
}

uint32_t nsgpu_cons_groups(void) { return (uint32_t)kMaxGroups; }      // the default; nsgpu_get_schedule for a context's own

int nsgpu_cons_slot(nsgpu_ctx *c, uint32_t slot)
{
    NS_CHECK(c && c->cons_engine, NSGPU_ERR_ARG, "nsgpu_cons_slot: call nsgpu_cons_begin first");
    NS_HIP(hipSetDevice(c->prm.device));
    return engine_slot(c, slot);
}

int nsgpu_cons_seed_requests(nsgpu_ctx *c, int group, uint32_t **gids_out, uint32_t **cursors_out, uint32_t *n_out)
{
    NS_CHECK(c && c->cons_engine && gids_out && cursors_out && n_out, NSGPU_ERR_ARG, "nsgpu_cons_seed_requests: bad argument");
    std::vector<uint32_t> a, b;
    engine_seed_requests(c, a, b, group);
    NS_TRY(give_u32(a, gids_out));
    NS_TRY(give_u32(b, cursors_out));
    *n_out = (uint32_t)a.size();
    return NSGPU_OK;
}

int nsgpu_cons_seed_resolve(nsgpu_ctx *c, const uint32_t *gids, const uint32_t *cursors, uint32_t n, uint32_t *n_started_out, uint32_t *all_done_out)
{
    NS_CHECK(c && c->cons_engine && (n == 0 || (gids && cursors)) && n_started_out && all_done_out, NSGPU_ERR_ARG, "nsgpu_cons_seed_resolve: bad argument");
    *n_started_out = engine_seed_resolve(c, gids, cursors, n);
    Engine *E = static_cast<Engine *>(c->cons_engine);
    *all_done_out = E->n_done_global >= E->n_total;
    return NSGPU_OK;
}

int nsgpu_cons_batches(nsgpu_ctx *c, int group)
{
    NS_CHECK(c && c->cons_engine, NSGPU_ERR_ARG, "nsgpu_cons_batches: call nsgpu_cons_begin first");
    NS_HIP(hipSetDevice(c->prm.device));
    return engine_batches(c, group);
}

int nsgpu_cons_claim_requests(nsgpu_ctx *c, int group, uint32_t **gids_out, uint32_t **reads_out, uint32_t *n_out)
{
    NS_CHECK(c && c->cons_engine && gids_out && reads_out && n_out, NSGPU_ERR_ARG, "nsgpu_cons_claim_requests: bad argument");
    std::vector<uint32_t> a, b;
    engine_claim_requests(c, a, b, group);
    NS_TRY(give_u32(a, gids_out));
    NS_TRY(give_u32(b, reads_out));
    *n_out = (uint32_t)a.size();
    return NSGPU_OK;
}

int nsgpu_cons_claim_resolve(nsgpu_ctx *c, const uint32_t *gids, const uint32_t *reads, uint32_t n, uint32_t *all_done_out)
{
    NS_CHECK(c && c->cons_engine && (n == 0 || (gids && reads)) && all_done_out, NSGPU_ERR_ARG, "nsgpu_cons_claim_resolve: bad argument");
    engine_claim_resolve(c, gids, reads, n);
    Engine *E = static_cast<Engine *>(c->cons_engine);
    *all_done_out = E->n_done_global >= E->n_total;
    return NSGPU_OK;
}

int nsgpu_cons_finish(nsgpu_ctx *c, uint32_t n_threads_out, nsgpu_consensus_stats *stats_out)
{
    NS_CHECK(c && c->cons_engine, NSGPU_ERR_ARG, "nsgpu_cons_finish: call nsgpu_cons_begin first");
    NS_TRY(engine_finish(c, n_threads_out));
    if (stats_out) *stats_out = c->cons_stats;
    return NSGPU_OK;
}

int nsgpu_set_schedule(nsgpu_ctx *c, uint32_t groups, uint32_t seed_bucket_depth, uint32_t seed_rings) { return nsgpu_set_schedule2(c, groups, seed_bucket_depth, seed_rings, seed_rings); }

int nsgpu_set_schedule2(nsgpu_ctx *c, uint32_t groups, uint32_t seed_bucket_depth, uint32_t seed_rings, uint32_t seed_tail_rings)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(groups == 1 || groups == 2 || groups == 4, NSGPU_ERR_ARG, "nsgpu_set_schedule: groups must be 1, 2 or 4");
    NS_CHECK(seed_bucket_depth <= 64 && seed_rings <= 8, NSGPU_ERR_ARG, "nsgpu_set_schedule: bucket depth at most 64, rings at most 8");
    NS_CHECK(!c->cons_engine, NSGPU_ERR_ARG, "nsgpu_set_schedule: a contig stage is in progress");
    NS_CHECK(seed_tail_rings <= seed_rings, NSGPU_ERR_ARG, "nsgpu_set_schedule2: seed_tail_rings must not exceed seed_rings");
    c->sched_groups = groups, c->seed_bucket_depth = seed_bucket_depth, c->seed_rings = seed_rings, c->seed_tail_rings = seed_tail_rings;
    c->sched_set = true, c->sched_auto = false;
    return NSGPU_OK;
}

int nsgpu_set_schedule_auto(nsgpu_ctx *c)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(!c->cons_engine, NSGPU_ERR_ARG, "nsgpu_set_schedule_auto: a contig stage is in progress");
    c->sched_auto = true, c->sched_set = false;
    return NSGPU_OK;
}

int nsgpu_set_defer(nsgpu_ctx *c, uint32_t anchors, uint32_t slots)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(!c->cons_engine, NSGPU_ERR_ARG, "nsgpu_set_defer: a contig stage is in progress");
    NS_CHECK(slots <= 3 && (slots == 0 || anchors >= 1), NSGPU_ERR_ARG, "nsgpu_set_defer: 0 .. 3 slots, at least one anchor");
    c->defer_anchors = slots ? anchors : 0, c->defer_slots = slots, c->defer_set = true;
    return NSGPU_OK;
}

int nsgpu_set_graph(nsgpu_ctx *c, uint32_t mode)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(!c->cons_engine, NSGPU_ERR_ARG, "nsgpu_set_graph: a contig stage is in progress");
    NS_CHECK((mode & 0xffu) <= NSGPU_GRAPH_DEVICE && !(mode & ~(0xffu | NSGPU_GRAPH_CHECK)), NSGPU_ERR_ARG, "nsgpu_set_graph: NSGPU_GRAPH_AUTO / _HOST / _DEVICE, optionally | NSGPU_GRAPH_CHECK");
    c->graph_mode = mode, c->graph_mode_set = true;
    return NSGPU_OK;
}

int nsgpu_get_graph_stats(const nsgpu_ctx *c, nsgpu_graph_stats *o)
{
    NS_CHECK(c && o, NSGPU_ERR_ARG, "nsgpu_get_graph_stats: null argument");
    memset(o, 0, sizeof(*o));
    o->placement = c->graph_used;
    if (c->graph_used != NSGPU_GRAPH_DEVICE || !c->graph_shared) return NSGPU_OK;
    const DevGraphShared &G = *static_cast<const DevGraphShared *>(c->graph_shared);
    o->checked = G.check;
    o->n_updates = G.n_updates.load(), o->n_launches = G.n_launches.load(), o->n_array_growths = G.n_grow.load(), o->n_long_reports = G.n_mid_copies.load();
    o->n_sequential_updates = G.n_seq_updates.load(), o->n_full_walks = G.n_full_walks.load(), o->n_split_calls = G.n_splits.load();
    for (int i = 0; i < 8; ++i) o->kernel_ms[i] = G.phase_ticks[i].load() / 1e5, o->by_duration[i] = G.hist[i].load();
    o->report_ms = G.update_ns.load() / 1e6, o->host_wait_first_ms = G.kernel_wait_ns.load() / 1e6, o->host_wait_second_ms = G.final_wait_ns.load() / 1e6;
    o->gb_copied_back = G.bytes_back.load() / 1e9, o->hbm_peak_gb = G.dev.peak() / 1e9, o->hbm_mapped_gb = G.dev.mapped() / 1e9, o->pinned_peak_gb = G.pin.peak() / 1e9, o->pinned_mapped_gb = G.pin.mapped() / 1e9;
    return NSGPU_OK;
}

int nsgpu_get_defer(const nsgpu_ctx *c, uint32_t *anchors, uint32_t *slots, uint64_t *n_deferred)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    if (anchors) *anchors = c->defer_anchors;
    if (slots) *slots = c->defer_slots;
    if (n_deferred) *n_deferred = c->cons_n_deferred;
    return NSGPU_OK;
}

int nsgpu_get_schedule2(const nsgpu_ctx *c, uint32_t *groups, uint32_t *seed_bucket_depth, uint32_t *seed_rings, uint32_t *seed_tail_rings, uint32_t *builders)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    if (groups) *groups = c->sched_groups;
    if (seed_bucket_depth) *seed_bucket_depth = c->seed_bucket_depth;
    if (seed_rings) *seed_rings = c->seed_rings;
    if (seed_tail_rings) *seed_tail_rings = c->seed_tail_rings;
    if (builders) *builders = c->cons_stats.n_builders;
    return NSGPU_OK;
}

int nsgpu_get_schedule(const nsgpu_ctx *c, uint32_t *groups, uint32_t *seed_bucket_depth, uint32_t *seed_rings)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    if (groups) *groups = c->sched_groups;
    if (seed_bucket_depth) *seed_bucket_depth = c->seed_bucket_depth;
    if (seed_rings) *seed_rings = c->seed_rings;
    return NSGPU_OK;
}

int nsgpu_set_read_id_base(nsgpu_ctx *c, uint32_t base)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    c->read_id_base = base;
    return NSGPU_OK;
}

int nsgpu_consensus_run(nsgpu_ctx *c, uint32_t n_builders, uint32_t n_threads_out, nsgpu_consensus_stats *stats_out)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_HIP(hipSetDevice(c->prm.device));
    NS_CHECK(c->h_off.size() == (size_t)c->reads.n + 1, NSGPU_ERR_ARG, "nsgpu_consensus_run: reads must be loaded first");
    NS_TRY(run_consensus(c, n_builders, n_threads_out));
    if (stats_out) *stats_out = c->cons_stats;
    return NSGPU_OK;
}

static const char *kExt[7] = {".genome", ".lone", ".id", ".pos", ".type", ".base", ".complement"};

static std::string stream_of(const cons::StreamSet &s, int which)
{
    switch (which) {
    case 0: return s.genome; case 1: return s.lone; case 2: return s.id_bytes(); case 3: return s.pos;
    case 4: return s.type; case 5: return s.base; default: return s.complement;
    }
}

int nsgpu_consensus_stream(nsgpu_ctx *c, uint32_t thread, uint32_t which, uint8_t **data_out, size_t *len_out)
{
    NS_CHECK(c && data_out && len_out, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(c->have_cons, NSGPU_ERR_ARG, "nsgpu_consensus_stream: run nsgpu_consensus_run first");
    NS_CHECK(which <= 7 && (which == 7 || thread < c->cons_out.size()), NSGPU_ERR_ARG, "no such stream");
    const std::string s = which == 7 ? cons::meta_data(c->cons_n_reads_out, c->cons_out) : stream_of(c->cons_out[thread], (int)which);
    uint8_t *p = (uint8_t *)malloc(s.size() + 1);
    NS_CHECK(p, NSGPU_ERR_NOMEM, "malloc failed");
    memcpy(p, s.data(), s.size());
    *data_out = p;
    *len_out = s.size();
    return NSGPU_OK;
}

int nsgpu_consensus_write(nsgpu_ctx *c, const char *temp_dir, const char *temp_file_name)
{
    NS_CHECK(c && temp_dir && temp_file_name, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(c->have_cons, NSGPU_ERR_ARG, "nsgpu_consensus_write: run nsgpu_consensus_run first");
    auto put = [&](const std::string &path, const std::string &bytes) -> int {
        FILE *f = fopen(path.c_str(), "wb");
        NS_CHECK(f, NSGPU_ERR_ARG, "cannot open %s", path.c_str());
        const size_t w = bytes.empty() ? 0 : fwrite(bytes.data(), 1, bytes.size(), f);
        fclose(f);
        NS_CHECK(w == bytes.size(), NSGPU_ERR_ARG, "short write to %s", path.c_str());
        return NSGPU_OK;
    };
    for (size_t t = 0; t < c->cons_out.size(); ++t)
        for (int k = 0; k < 7; ++k)       // same names as Consensus.cpp:36 + ConsensusGraphWriter
            NS_TRY(put(std::string(temp_dir) + temp_file_name + ".tid." + std::to_string(t) + kExt[k], stream_of(c->cons_out[t], k)));
    return put(std::string(temp_dir) + "metaData", cons::meta_data(c->cons_n_reads_out, c->cons_out));
}

int nsgpu_consensus_verify(nsgpu_ctx *c, uint64_t *n_bad_out)
{
    NS_CHECK(c && n_bad_out, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(c->have_cons, NSGPU_ERR_ARG, "nsgpu_consensus_verify: run nsgpu_consensus_run first");
    const uint32_t N = c->reads.n;
    std::vector<uint8_t> seen(N, 0);
    uint64_t bad = 0;
    for (const cons::StreamSet &s : c->cons_out) {
        std::vector<std::pair<cons::read_t, std::string>> rd;
        std::string err;
        if (!cons::decode_streams(s, rd, err)) { set_error("stream set does not decode: %s", err.c_str()); return NSGPU_ERR_ARG; }
        for (auto &pr : rd) {
            const uint32_t r = pr.first - c->read_id_base;      // streams carry global ids
            if (pr.first < c->read_id_base || r >= N || seen[r]) { ++bad; continue; }
            seen[r] = 1;
            const size_t L = (size_t)(c->h_off[r + 1] - c->h_off[r]);
            std::string rbuf;
            if (pr.second.size() != L || memcmp(pr.second.data(), mirror_read(c, r, rbuf), L) != 0) ++bad;
        }
    }
    if (c->cons_n_reads_out == N)                       // a multi-GPU rank only holds its builders' share of the reads
        for (uint32_t r = 0; r < N; ++r) bad += !seen[r];
    *n_bad_out = bad;
    return NSGPU_OK;
}

}  // extern "C"
