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
#include "common.hpp"
#include "consensus.hpp"
#include "host_util.hpp"
#include <memory>

namespace nsgpu {

using cons::read_t;

struct Builder {
    enum State { NEED_CONTIG, ADVANCE, WAIT_FILTER, WAIT_ALIGN, GOT_FILTER, GOT_ALIGN, DONE };
    State st = NEED_CONTIG;
    uint32_t id = 0;
    std::unique_ptr<cons::ContigGraph> g;
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
    // cached index of the current main path
    mm2::RefIndex idx;
    bool idx_valid = false;
    cons::StreamSet out;
    uint64_t n_minhash = 0, n_minhash_new = 0, n_aligner = 0, n_align_calls = 0, n_contigs = 0, n_lone = 0;
    double cpu_ms = 0, max_ms = 0;
};

struct Driver {
    nsgpu_ctx *c;
    uint32_t N, id_base = 0;
    uint64_t edge_thr;
    size_t offset;                     // avgReadLen / 4 (src/Consensus.cpp:54)
    std::vector<uint8_t> in_graph, rep;
    std::vector<Builder> B;

    const char *read_ptr(read_t r) const { return c->h_bases.data() + c->h_off[r]; }
    size_t read_len(read_t r) const { return (size_t)(c->h_off[r + 1] - c->h_off[r]); }

    // Consensus::getRead + createGraph (src/Consensus.cpp:388-403, 444-468); sequential
    void claim_seed(Builder &b)
    {
        read_t r = b.cursor;
        while (r < N && in_graph[r]) ++r;
        if (r >= N) { b.st = Builder::DONE; return; }
        in_graph[r] = 1;
        b.g.reset(new cons::ContigGraph());
        b.g->main_path.assign(read_ptr(r), read_len(r));
        b.g->start_pos = 0;
        b.g->end_pos = (ssize_t)read_len(r);
        b.g->first_read = r + id_base;       // graph / stream ids are global, array indices local
        b.cursor = r + 1;
        b.init_start = 0;
        b.len = b.g->end_pos - b.g->start_pos;
        b.cur_pos = b.g->start_pos;
        b.right_phase = true, b.edges_too_many = false, b.window_open = false;
        b.idx_valid = false;
        b.st = Builder::ADVANCE;
    }

    void finish_contig(Builder &b)
    {
        cons::ContigGraph &g = *b.g;
        if (g.num_reads() == 0) {
            g.write_read_lone(b.out);
            b.out.lone_ids.push_back(g.first_read);
            b.out.reads_in_contig.push_back(1);
            ++b.n_lone;
        } else {
            g.write_main_path(b.out);
            g.write_reads(b.out);
            b.out.reads_in_contig.push_back((read_t)g.num_reads());
        }
        ++b.n_contigs;
        b.g.reset();
        b.st = Builder::NEED_CONTIG;
    }

    // opens the window at cur_pos (addRelatedReads prologue, src/Consensus.cpp:168-184); false = nothing to query
    bool open_window(Builder &b)
    {
        cons::ContigGraph &g = *b.g;
        const ssize_t off = b.cur_pos - g.start_pos;
        if (b.len == 0 || off < 0 || off >= (ssize_t)g.main_path.size()) return false;
        const size_t n = (ssize_t)g.main_path.size() >= off + b.len ? (size_t)b.len : g.main_path.size() - (size_t)off;
        b.win[0].assign(g.main_path, (size_t)off, n);
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
        cons::ContigGraph &g = *b.g;
        const bool usable = b.len >= 32 && !rep[g.first_read - id_base];
        for (;;) {
            if (b.right_phase) {
                if (window_just_done) {
                    b.cur_pos += (ssize_t)offset;
                    window_just_done = false;
                    if (b.cur_pos + b.len > g.end_pos) b.right_phase = false;
                    else if (g.num_edges() >= edge_thr) b.edges_too_many = true, b.right_phase = false;
                    if (!b.right_phase) { b.cur_pos = b.init_start - (ssize_t)offset; continue; }
                }
                if (!usable) { b.right_phase = false; b.cur_pos = b.init_start - (ssize_t)offset; continue; }
                if (open_window(b)) return;
                window_just_done = true;         // addRelatedReads returned immediately
            } else {
                if (window_just_done) { b.cur_pos -= (ssize_t)offset; window_just_done = false; }
                if (!(usable && !b.edges_too_many)) break;
                if (b.cur_pos < g.start_pos) break;
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
        cons::ContigGraph &g = *b.g;
        for (; b.strand < 2; ++b.strand, b.ci = 0, b.strand_counted = false) {
            if (!b.strand_counted) { b.n_minhash += b.cand[b.strand].size(); b.strand_counted = true; }
            for (; b.ci < b.cand[b.strand].size(); ++b.ci) {
                const read_t r = b.cand[b.strand][b.ci];
                if (g.num_edges() >= edge_thr) { b.window_open = false; walk(b, true); return; }   // `return` out of addRelatedReads
                if (rep[r]) continue;
                if (in_graph[r]) continue;
                ++b.n_minhash_new;
                if (read_len(r) < 32) continue;
                if (b.strand) { std::string fwd(read_ptr(r), read_len(r)); cons::reverse_complement(fwd, b.query); }
                else b.query.assign(read_ptr(r), read_len(r));
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
        if (dt > b.max_ms) b.max_ms = dt;
    }
    void advance_inner(Builder &b)
    {
        if (b.st == Builder::ADVANCE) walk(b, false);
        else if (b.st == Builder::GOT_FILTER) next_candidate(b);
        else if (b.st == Builder::GOT_ALIGN) {
            cons::ContigGraph &g = *b.g;
            if (b.accepted) {
                if (g.num_reads() == 0) {                       // src/Consensus.cpp:319-324
                    const std::string seed = g.main_path;
                    g.main_path.clear();
                    g.initialize(seed, g.first_read, 0);
                    g.calculate_main_path_greedy();
                }
                g.update_graph(b.query, b.aln.edits, (ssize_t)b.aln.begin_offset, (ssize_t)b.aln.end_offset, b.pend + id_base, (long)b.aln.rel_pos, b.strand == 1);
                g.calculate_main_path_greedy();
                b.idx_valid = false;
                b.accepted = false;
            }
            ++b.ci;
            next_candidate(b);
        }
    }
};

static int run_consensus(nsgpu_ctx *c, uint32_t n_builders, uint32_t n_threads_out)
{
    NS_CHECK(c->have_index && c->have_salts, NSGPU_ERR_ARG, "nsgpu_consensus_run: call nsgpu_sketch and nsgpu_build_index first");
    NS_CHECK(n_builders >= 1 && n_threads_out >= 1, NSGPU_ERR_ARG, "n_builders and n_threads_out must be >= 1");
    Driver D;
    D.c = c;
    D.N = c->reads.n;
    D.edge_thr = c->prm.edge_threshold;
    D.id_base = c->read_id_base;
    NS_CHECK((uint64_t)D.id_base + D.N <= 0xFFFFFFFFull, NSGPU_ERR_RANGE, "read id base + reads exceeds read_t");
    D.offset = D.N ? (size_t)(c->reads.n_bases / D.N) / 4 : 0;      // avgReadLen is integer-truncated (src/ReadData.cpp:207)
    if (D.offset == 0) D.offset = 1;                                  // the reference would never terminate with a zero stride
    D.in_graph.assign(D.N, 0);
    D.rep.assign((size_t)D.N + 1, 0);
    const double t0 = now_ms();
    if (D.N) NS_TRY(nsgpu_check_repetitive(c, D.rep.data()));
    if (n_builders > D.N && D.N > 0) n_builders = D.N;
    if (D.N == 0) n_builders = 1;
    D.B.resize(n_builders);
    for (uint32_t i = 0; i < n_builders; ++i) D.B[i].id = i;
    nsgpu_consensus_stats &S = c->cons_stats;
    memset(&S, 0, sizeof(S));
    S.n_builders = n_builders;
    std::vector<uint32_t> who;
    std::string qbuf;
    std::vector<uint64_t> qoff;
    std::vector<uint64_t> foff;
    std::vector<uint32_t> fids;
    std::vector<AlignReq> reqs;
    std::vector<mm2::AlnOut> outs;
    for (;;) {
        // 1. parallel: consume deliveries, run to the next request
        double a0 = now_ms();
        par_for_pinned(D.B.size(), [&](size_t i) { D.advance(D.B[i]); });
        // 2. sequential seed claims, then 3. parallel: open their first window
        bool any_new = false;
        for (Builder &b : D.B) if (b.st == Builder::NEED_CONTIG) { D.claim_seed(b); any_new |= b.st == Builder::ADVANCE; }
        if (any_new) par_for_pinned(D.B.size(), [&](size_t i) { if (D.B[i].st == Builder::ADVANCE) D.advance(D.B[i]); });
        // builders whose fresh contig finished at once (lone, repetitive or short seeds) claim again
        for (int guard = 0; guard < 1 << 30; ++guard) {
            bool again = false;
            for (Builder &b : D.B) if (b.st == Builder::NEED_CONTIG) { D.claim_seed(b); again |= b.st == Builder::ADVANCE; }
            if (!again) break;
            par_for_pinned(D.B.size(), [&](size_t i) { if (D.B[i].st == Builder::ADVANCE) D.advance(D.B[i]); });
        }
        S.graph_ms += now_ms() - a0;
        // 4. window queries of this round
        who.clear();
        for (Builder &b : D.B) if (b.st == Builder::WAIT_FILTER) who.push_back(b.id);
        if (!who.empty()) {
            double f0 = now_ms();
            qbuf.clear(); qoff.assign(1, 0);
            for (uint32_t bi : who) for (int s = 0; s < 2; ++s) { qbuf += D.B[bi].win[s]; qoff.push_back(qbuf.size()); }
            const uint32_t nq = (uint32_t)(2 * who.size());
            NS_TRY(filter_strings_device(c, qbuf.data(), qoff.data(), nq));
            foff.resize((size_t)nq + 1);
            fids.resize(c->f_total + 1);
            NS_HIP(hipMemcpyAsync(foff.data(), c->f_off.p, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost, c->stream));
            if (c->f_total) NS_HIP(hipMemcpyAsync(fids.data(), c->f_ids.p, c->f_total * 4, hipMemcpyDeviceToHost, c->stream));
            NS_HIP(hipStreamSynchronize(c->stream));
            for (size_t w = 0; w < who.size(); ++w) {
                Builder &b = D.B[who[w]];
                for (int s = 0; s < 2; ++s) b.cand[s].assign(fids.begin() + foff[2 * w + s], fids.begin() + foff[2 * w + s + 1]);
                b.st = Builder::GOT_FILTER;
            }
            S.filter_ms += now_ms() - f0;
            S.n_windows += who.size();
            ++S.n_filter_rounds;
        }
        // 5. alignments of this round
        who.clear();
        for (Builder &b : D.B) if (b.st == Builder::WAIT_ALIGN) who.push_back(b.id);
        if (!who.empty()) {
            double g0 = now_ms();
            par_for(who.size(), [&](size_t w) {
                Builder &b = D.B[who[w]];
                if (!b.idx_valid) {
                    b.idx.build(b.g->main_path.data(), (uint32_t)b.g->main_path.size(), (int)c->prm.m_w, (int)c->prm.m_k, 2e-4f);
                    b.idx_valid = true;
                }
            });
            const double g1 = now_ms();
            S.index_ms += g1 - g0;
            reqs.resize(who.size());
            for (size_t w = 0; w < who.size(); ++w) {
                Builder &b = D.B[who[w]];
                reqs[w] = AlignReq{&b.idx, b.g->main_path.data(), b.g->main_path.size(), b.query.data(), b.query.size()};
            }
            NS_TRY(align_requests(c, reqs, outs));
            // 6. claims, strictly in builder order (src/Consensus.cpp:256-277 without lock contention)
            for (size_t w = 0; w < who.size(); ++w) {
                Builder &b = D.B[who[w]];
                b.aln = std::move(outs[w]);
                ++b.n_align_calls;
                b.accepted = false;
                if (b.aln.ok && !D.in_graph[b.pend]) { D.in_graph[b.pend] = 1; b.accepted = true; ++b.n_aligner; }
                b.st = Builder::GOT_ALIGN;
            }
            S.align_ms += now_ms() - g1;
            ++S.n_align_rounds;
        }
        bool active = false;
        for (Builder &b : D.B) active |= b.st != Builder::DONE;
        if (!active) break;
        ++S.n_rounds;
    }
    // merge builders into the requested number of output "threads" (Compressor expects exactly numThr
    // file sets, src/Compressor.cpp:123-124; Decompressor reads numThr from metaData)
    c->cons_out.assign(n_threads_out, cons::StreamSet());
    for (size_t i = 0; i < D.B.size(); ++i) c->cons_out[i * n_threads_out / D.B.size()].append(D.B[i].out);
    for (Builder &b : D.B) {
        S.n_contigs += b.n_contigs; S.n_lone += b.n_lone; S.count_minhash += b.n_minhash; S.count_minhash_not_in_graph += b.n_minhash_new;
        S.count_aligner += b.n_aligner; S.n_align_calls += b.n_align_calls;
        S.graph_cpu_ms += b.cpu_ms; if (b.max_ms > S.graph_max_ms) S.graph_max_ms = b.max_ms;
    }
    S.total_ms = now_ms() - t0;
    c->have_cons = true;
    return NSGPU_OK;
}

}  // namespace nsgpu

using namespace nsgpu;

extern "C" {

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
    const std::string s = which == 7 ? cons::meta_data(c->reads.n, c->cons_out) : stream_of(c->cons_out[thread], (int)which);
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
    return put(std::string(temp_dir) + "metaData", cons::meta_data(c->reads.n, c->cons_out));
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
            if (pr.second.size() != L || memcmp(pr.second.data(), c->h_bases.data() + c->h_off[r], L) != 0) ++bad;
        }
    }
    for (uint32_t r = 0; r < N; ++r) bad += !seen[r];
    *n_bad_out = bad;
    return NSGPU_OK;
}

}  // extern "C"
