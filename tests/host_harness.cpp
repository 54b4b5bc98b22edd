// host_harness.cpp -- TEST-ONLY driver for the host-side aligner logic (nanospring_amd/csrc/mm2.cpp).
// It runs AlignJob with the DP requests answered by the CPU oracle (oracle/ksw2_oracle.c), so that
// the decision chain (sketch, index, seeds, chaining, regions, skeleton, CIGAR fixing, edit
// conversion) can be checked against the reference's minimap2 without a GPU.  Built by
// tests/host_lib.py into tests/_build/; never part of libnsgpu.so.
#include <cstring>
#include <vector>
#include <algorithm>
#include "../nanospring_amd/csrc/mm2.hpp"

extern "C" {
typedef struct {
    uint32_t max; int32_t zdropped;
    int32_t max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} oracle_ez_t;
int oracle_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t sc_mch, int8_t sc_mis, int8_t sc_ambi_mat,
                     int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                     oracle_ez_t *ez, uint32_t *cigar_out, int cigar_cap);
}

using namespace nsgpu::mm2;

static void answer(AlignJob &J)
{
    std::vector<DpKey> miss = J.cache.missing;
    for (const DpKey &k : miss) {
        const int ql = k.qe - k.qs, tl = k.re - k.rs;
        std::vector<uint8_t> q(J.qseq.begin() + k.qs, J.qseq.begin() + k.qe), t(J.ref->seq.begin() + k.rs, J.ref->seq.begin() + k.re);
        if (k.flag & 0x02) { std::reverse(q.begin(), q.end()); std::reverse(t.begin(), t.end()); }   // left extension
        std::vector<uint32_t> cig(ql + tl + 4);
        oracle_ez_t ez;
        const Opt &o = J.opt;
        int n = oracle_ksw_extd2(ql, q.data(), tl, t.data(), (int8_t)o.a, (int8_t)-o.b, (int8_t)-o.sc_ambi, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2,
                                 (int8_t)o.e2, k.w, k.zdrop, k.end_bonus, k.flag, &ez, cig.data(), (int)cig.size());
        DpResult r;
        r.max = ez.max; r.zdropped = ez.zdropped; r.max_q = ez.max_q; r.max_t = ez.max_t; r.mqe = ez.mqe; r.mqe_t = ez.mqe_t;
        r.mte = ez.mte; r.mte_q = ez.mte_q; r.score = ez.score; r.reach_end = ez.reach_end;
        J.cache.put(k, r, cig.data(), (uint32_t)(n > 0 ? n : 0));
    }
}

extern "C" {

typedef struct {
    int32_t ok, hits;
    int64_t rel_pos, begin_offset, end_offset;
    int32_t rs, re, qs, qe, blen, mlen, n_ambi, dp_max, n_cigar, mid_occ, n_rounds, n_dp;
} harness_aln_t;

// returns the number of edits; cigar/edits receive at most *_cap entries. edits: type | base<<8 | num<<16 (u64)
int harness_align(const char *ref, int rl, const char *qry, int ql, int k, int w, int max_chain_iter, harness_aln_t *out,
                  uint32_t *cigar, int cigar_cap, uint64_t *edits, int edit_cap)
{
    RefIndex ri;
    ri.build(ref, (uint32_t)rl, w, k, 2e-4f);
    Opt o;
    o.k = k, o.w = w, o.max_chain_iter = max_chain_iter;
    AlignJob J;
    J.start(&ri, qry, ql, o);
    int rounds = 0, n_dp = 0;
    while (!J.step()) { n_dp += (int)J.cache.missing.size(); answer(J); ++rounds; }
    AlnOut ao;
    align_read_result(J, ref, (size_t)rl, ao);
    out->ok = ao.ok; out->hits = ao.hits; out->rel_pos = ao.rel_pos; out->begin_offset = ao.begin_offset; out->end_offset = ao.end_offset;
    out->rs = ao.rs; out->re = ao.re; out->qs = ao.qs; out->qe = ao.qe; out->blen = ao.blen; out->mlen = ao.mlen; out->n_ambi = ao.n_ambi;
    out->dp_max = ao.dp_max; out->n_cigar = ao.n_cigar; out->mid_occ = ri.mid_occ; out->n_rounds = rounds; out->n_dp = n_dp;
    for (int i = 0; i < (int)ao.cigar.size() && i < cigar_cap; ++i) cigar[i] = ao.cigar[i];
    for (int i = 0; i < (int)ao.edits.size() && i < edit_cap; ++i)
        edits[i] = (uint64_t)ao.edits[i].type | (uint64_t)ao.edits[i].base << 8 | (uint64_t)ao.edits[i].num << 16;
    return (int)ao.edits.size();
}

int harness_sketch(const char *s, int len, int w, int k, uint64_t *xy, int cap)
{
    std::vector<Anchor> v;
    mm_sketch(s, len, w, k, 0, v);
    for (int i = 0; i < (int)v.size() && i < cap; ++i) { xy[2 * i] = v[i].x; xy[2 * i + 1] = v[i].y; }
    return (int)v.size();
}

void harness_nt4(const char *s, int64_t n, uint8_t *out) { nt4_codes(s, (size_t)n, out); }

// RefIndex::build_from_sketch over caller-supplied minimizers: keys / CSR starts / positions, mid_occ, and for every key
// what get() answers (count and first position)
int64_t harness_index(const char *s, int len, int w, int k, const uint64_t *xy, int64_t n_mz, uint64_t *keys, uint32_t *start, uint64_t *pos, int64_t cap,
                      int32_t *mid_occ, uint64_t *get_first, int32_t *get_n)
{
    RefIndex ri;
    ri.build_from_sketch(s, (uint32_t)len, w, k, 2e-4f, (const Anchor *)xy, (size_t)n_mz);
    *mid_occ = ri.mid_occ;
    const int64_t nk = (int64_t)ri.keys.size();
    if (nk > cap || (int64_t)ri.pos.size() > cap) return -1;
    for (int64_t i = 0; i < nk; ++i) {
        keys[i] = ri.keys[i], start[i] = ri.start[i];
        int n = 0;
        const uint64_t *p = ri.get(ri.keys[i], &n);
        get_n[i] = n, get_first[i] = p ? p[0] : ~0ull;
    }
    start[nk] = ri.start[nk];
    for (size_t i = 0; i < ri.pos.size(); ++i) pos[i] = ri.pos[i];
    int n = 7;
    if (ri.get(0xfffffffffffull, &n) != nullptr || n != 0) return -2;      // a key that no 2k-bit hash can be
    return nk;
}

// the anchors mm_map_frag hands to mm_chain_dp for one (reference, query) pair (sorted), and the chaining recurrence over a
// list of anchors as the plain loop (chain_forward_host): the checker of chain.hip
int64_t harness_seeds2(const char *ref, int rl, const char *qry, int ql, int k, int w, uint64_t *xy, int64_t cap, int32_t *mid_occ, float *avg)
{
    RefIndex ri;
    ri.build(ref, (uint32_t)rl, w, k, 2e-4f);
    Opt o;
    o.k = k, o.w = w;
    AlignJob J;
    J.start(&ri, qry, ql, o);
    J.seed();
    *mid_occ = ri.mid_occ, *avg = J.avg_qspan;
    for (int64_t i = 0; i < (int64_t)J.a.size() && i < cap; ++i) xy[2 * i] = J.a[i].x, xy[2 * i + 1] = J.a[i].y;
    return (int64_t)J.a.size();
}
int64_t harness_seeds(const char *ref, int rl, const char *qry, int ql, int k, int w, uint64_t *xy, int64_t cap)
{
    RefIndex ri;
    ri.build(ref, (uint32_t)rl, w, k, 2e-4f);
    Opt o;
    o.k = k, o.w = w;
    AlignJob J;
    J.start(&ri, qry, ql, o);
    J.seed();
    for (int64_t i = 0; i < (int64_t)J.a.size() && i < cap; ++i) xy[2 * i] = J.a[i].x, xy[2 * i + 1] = J.a[i].y;
    return (int64_t)J.a.size();
}
void harness_chain_forward(const uint64_t *xy, int64_t n, int max_chain_iter, int32_t *f, int32_t *p)
{
    Opt o;
    o.max_chain_iter = max_chain_iter;
    std::vector<Anchor> a((const Anchor *)xy, (const Anchor *)xy + n);
    chain_forward_host(o, a, chain_avg_qspan(a), f, p);
}

// the product's chain_finish (backtracking, chain order) on forward-pass scores computed elsewhere (the GPU kernel): u[] and the reordered anchors
int64_t harness_chain_finish(uint64_t *xy, int64_t n, int max_chain_iter, const int32_t *f, const int32_t *p, uint64_t *u_out, int64_t *n_a_out)
{
    Opt o;
    o.max_chain_iter = max_chain_iter;
    std::vector<Anchor> a((const Anchor *)xy, (const Anchor *)xy + n);
    std::vector<uint64_t> u;
    chain_finish_scores(o, a, f, p, u);
    for (size_t i = 0; i < a.size(); ++i) xy[2 * i] = a[i].x, xy[2 * i + 1] = a[i].y;
    for (size_t i = 0; i < u.size(); ++i) u_out[i] = u[i];
    *n_a_out = (int64_t)a.size();
    return (int64_t)u.size();
}

void harness_radix_sort_128x(uint64_t *xy, int64_t n) { radix_sort_128x((Anchor *)xy, (Anchor *)xy + n); }
void harness_radix_sort_64(uint64_t *x, int64_t n) { radix_sort_64(x, x + n); }

}

// ---------------------------------------------------------------------------
// Sequential restatement of the reference's `-t 1` contig loop (src/Consensus.cpp:21-340) as plain
// nested loops, with candidates from the CPU oracle filter and DP from the CPU oracle.  It is the
// checker for the GPU batch engine (consensus_driver.hip) run with ONE builder, and it exercises the
// product's host-side graph / emission / decoder code on the CPU.
// ---------------------------------------------------------------------------
#include <string>
#include <chrono>
#include "../nanospring_amd/csrc/consensus.hpp"
#include "../nanospring_amd/csrc/consensus_soa.hpp"
static double hnow() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" {
void oracle_sketch_reads(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, const uint64_t *salts, uint64_t *sketches);
void oracle_index_build(const uint64_t *sketches, uint32_t N, uint32_t n, uint64_t *keys, uint32_t *start, uint32_t *ids, uint32_t *nkeys);
uint64_t oracle_filter_string(const char *s, uint64_t len, uint32_t k, uint32_t N, uint32_t n, uint32_t thr, const uint64_t *salts,
                              const uint64_t *keys, const uint32_t *start, const uint32_t *ids, const uint32_t *nkeys, uint32_t *out,
                              uint64_t cap, uint64_t *n_matches);
int oracle_check_repetitive(const char *s, uint64_t len);
}

using namespace nsgpu::cons;

// optional: answer alignments with the REFERENCE's minimap2 (ref_mm2_align of oracle/_ref/libmm2ref.so);
// used for the CPU baseline of bench.py (the reference's own aligner at SSE speed)
typedef struct { int32_t hits, rs, re, qs, qe, blen, mlen, n_ambi, dp_max, dp_score, score, cnt, rev, mid_occ, n_cigar; } ref_aln_t;
typedef int (*ref_align_fn)(const char *, int, const char *, int, int, int, int, ref_aln_t *, uint32_t *, int);
static ref_align_fn g_ref_align = nullptr;
extern "C" void harness_set_ref_align(void *fn) { g_ref_align = (ref_align_fn)fn; }

static bool harness_align1(const std::string &ref, const std::string &q, int k, int w, int mci, AlnOut &ao)
{
    if (g_ref_align) {
        ref_aln_t ra;
        std::vector<uint32_t> cig(ref.size() + q.size() + 8);
        g_ref_align(ref.c_str(), (int)ref.size(), q.c_str(), (int)q.size(), k, w, mci, &ra, cig.data(), (int)cig.size());
        cig.resize(ra.n_cigar > 0 ? ra.n_cigar : 0);
        edits_from_hit(ra.hits, ra.rs, ra.re, ra.qs, ra.qe, ra.blen, ra.mlen, ra.n_ambi, ra.dp_max, ra.n_cigar >= 0, cig, ref.data(), ref.size(), q.data(),
                       q.size(), ao);
        return ao.ok != 0;
    }
    RefIndex ri;
    ri.build(ref.data(), (uint32_t)ref.size(), w, k, 2e-4f);
    Opt o;
    o.k = k, o.w = w, o.max_chain_iter = mci;
    AlignJob J;
    J.start(&ri, q.data(), (int)q.size(), o);
    while (!J.step()) answer(J);
    align_read_result(J, ref.data(), ref.size(), ao);
    return ao.ok != 0;
}

extern uint64_t g_dbg_calls, g_dbg_skipped, g_dbg_spliced, g_dbg_spliced_nodes;
extern "C" uint64_t harness_dbg(int which) { return which == 0 ? g_dbg_calls : which == 1 ? g_dbg_skipped : which == 2 ? g_dbg_spliced : g_dbg_spliced_nodes; }
extern "C" {

typedef struct { uint64_t n_contigs, n_lone, count_minhash, count_minhash_not_in_graph, count_aligner, n_align_calls, n_bad_roundtrip, n_graph_check_fail; double update_ms, mainpath_ms, write_ms; } harness_cons_stats;

}  // extern "C"

// NSGPU_HARNESS_GRAPH=both: the pointer graph (consensus.cpp) and the structure-of-arrays graph (dgraph.hpp with a team of one) side by side on every
// update -- consensus, positions and sizes must agree after each step; the streams come from the SoA graph's emission
struct ShadowGraph {
    ContigGraph a;
    SoaGraph b;
    // NSGPU_HARNESS_SOA2_FLAGS: a second structure-of-arrays graph with these debug flags beside the first: whichever way a split is taken (step by
    // step, stretches by the team, by routes; from which copy on) the nodes, the edges and every list must come out the same, id for id
    SoaGraph c;
    bool have_c = false;
    void compare_soa(const char *where)
    {
        namespace dg = nsgpu::dg;
        using dg::Hdr; using dg::Chunk; using dg::kChunkIds; using dg::kOutInl; using dg::kInInl; using dg::kEdgeInl;
        nsgpu::cons::SoaStore &X = b.store(), &Y = c.store();
        const Hdr &hx = X.hdr, &hy = Y.hdr;
        auto list_at = [](const std::vector<Chunk> &ch, const uint32_t *inl, uint32_t n_inl, uint32_t ext, uint32_t k) { if (k < n_inl) return inl[k]; k -= n_inl; uint32_t cc = ext; while (k >= kChunkIds) cc = ch[cc].next, k -= kChunkIds; return ch[cc].v[k]; };
        auto say = [&](const char *what, uint64_t i, uint64_t x, uint64_t y) { if (mismatches()++ < 5) fprintf(stderr, "SOA2 MISMATCH after %s (read count %zu): %s %llu: %llu / %llu (nodes %u / %u, edges %u / %u, routes %u / %u)\n", where, a.num_reads(), what, (unsigned long long)i, (unsigned long long)x, (unsigned long long)y, hx.n_nodes, hy.n_nodes, hx.n_edges, hy.n_edges, hx.st_routes, hy.st_routes); };
        if (hx.n_nodes != hy.n_nodes || hx.n_edges != hy.n_edges || hx.live_nodes != hy.live_nodes || hx.live_edges != hy.live_edges || hx.m != hy.m || hx.n_multi != hy.n_multi) { say("counts", 0, hx.n_nodes, hy.n_nodes); return; }
        for (uint32_t i = 0; i <= hx.m; ++i) if (X.pn[hx.path_off + i] != Y.pn[hy.path_off + i]) { say("path node", i, X.pn[hx.path_off + i], Y.pn[hy.path_off + i]); return; }
        for (uint32_t n = 0; n < hx.n_nodes; ++n) {
            const dg::Node &x = X.nodes[n], &y = Y.nodes[n];
            if (x.n_out != y.n_out || x.n_in != y.n_in || x.base != y.base || x.on_main != y.on_main) { say("header of node", n, ((uint64_t)x.n_out << 24) | (x.n_in << 16) | (x.base << 8) | x.on_main, ((uint64_t)y.n_out << 24) | (y.n_in << 16) | (y.base << 8) | y.on_main); return; }
            for (uint32_t k = 0; k < x.n_out; ++k) if (list_at(X.chunks, x.out, kOutInl, x.out_ext, k) != list_at(Y.chunks, y.out, kOutInl, y.out_ext, k)) { say("an out reference of node", n, list_at(X.chunks, x.out, kOutInl, x.out_ext, k), list_at(Y.chunks, y.out, kOutInl, y.out_ext, k)); return; }
            for (uint32_t k = 0; k < x.n_in; ++k) if (list_at(X.chunks, x.in, kInInl, x.in_ext, k) != list_at(Y.chunks, y.in, kInInl, y.in_ext, k)) { say("an in reference of node", n, list_at(X.chunks, x.in, kInInl, x.in_ext, k), list_at(Y.chunks, y.in, kInInl, y.in_ext, k)); return; }
        }
        for (uint32_t e = 0; e < hx.n_edges; ++e) {
            const dg::Edge &x = X.edges[e], &y = Y.edges[e];
            if (x.src != y.src || x.sink != y.sink || x.count != y.count) { say("header of edge", e, ((uint64_t)x.src << 32) | x.sink, ((uint64_t)y.src << 32) | y.sink); return; }
            for (uint32_t k = 0; k < x.count; ++k) if (list_at(X.chunks, x.ids, kEdgeInl, x.head, k) != list_at(Y.chunks, y.ids, kEdgeInl, y.head, k)) { say("a read id of edge", e, list_at(X.chunks, x.ids, kEdgeInl, x.head, k), list_at(Y.chunks, y.ids, kEdgeInl, y.head, k)); return; }
        }
    }
    ssize_t start_pos = 0, end_pos = 0;
    std::string main_path;
    read_t first_read = 0;
    size_t path_changed_from = 0;
    uint64_t dbg_cycles_calls = 0, dbg_cycles_skipped = 0, dbg_spliced = 0, dbg_spliced_nodes = 0;
    static uint64_t &mismatches() { static uint64_t v = 0; return v; }
    void sync_out(const char *where)
    {
        if (a.main_path != b.main_path || a.start_pos != b.start_pos || a.end_pos != b.end_pos || a.num_edges() != b.num_edges() || a.num_nodes() != b.num_nodes()) {
            if (mismatches()++ < 5) {
                size_t d = 0; while (d < a.main_path.size() && d < b.main_path.size() && a.main_path[d] == b.main_path[d]) ++d;
                fprintf(stderr, "SHADOW MISMATCH after %s (read count %zu): path %zu / %zu bases (first difference at %zu), start %zd / %zd, end %zd / %zd, edges %zu / %zu, nodes %zu / %zu\n", where,
                        a.num_reads(), a.main_path.size(), b.main_path.size(), d, a.start_pos, b.start_pos, a.end_pos, b.end_pos, a.num_edges(), b.num_edges(), a.num_nodes(), b.num_nodes());
            }
        }
        main_path = b.main_path, start_pos = b.start_pos, end_pos = b.end_pos;
        path_changed_from = std::min(a.path_changed_from, b.path_changed_from);
    }
    void initialize(const std::string &seed, read_t id, long pos)
    {
        a.main_path.clear(); a.first_read = b.first_read = first_read; a.initialize(seed, id, pos); b.initialize(seed, id, pos);
        static const char *f2 = getenv("NSGPU_HARNESS_SOA2_FLAGS");
        have_c = f2 != nullptr;
        if (have_c) { c.dbg_flags_override = atoi(f2); c.first_read = first_read; c.initialize(seed, id, pos); }
    }
    void update_graph(const std::string &s, const std::vector<nsgpu::mm2::EditOp> &script, ssize_t bo, ssize_t eo, read_t id, long pos, bool rc)
    {
        a.update_graph(s, script, bo, eo, id, pos, rc);
        b.update_graph(s, script, bo, eo, id, pos, rc);
        if (have_c) c.update_graph(s, script, bo, eo, id, pos, rc);
    }
    void calculate_main_path_greedy()
    {
        a.path_changed_from = b.path_changed_from = path_changed_from;
        a.calculate_main_path_greedy();
        b.calculate_main_path_greedy();
        if (have_c) { c.path_changed_from = path_changed_from; c.calculate_main_path_greedy(); compare_soa("calculate_main_path_greedy"); }
        sync_out("calculate_main_path_greedy");
    }
    size_t num_reads() const { return b.num_reads(); }
    size_t num_edges() const { return b.num_edges(); }
    void write_main_path(StreamSet &o) const { b.write_main_path(o); }
    void write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source = nullptr)
    {
        StreamSet oa;
        a.write_reads(oa, source);
        StreamSet ob;
        b.write_reads(ob, source);
        if (oa.pos != ob.pos || oa.type != ob.type || oa.base != ob.base || oa.complement != ob.complement || oa.id_contigs != ob.id_contigs) { if (mismatches()++ < 5) fprintf(stderr, "SHADOW MISMATCH in the emission of a contig of %zu reads\n", a.num_reads()); }
        o.append(ob);
    }
    void write_read_lone(StreamSet &o) const { o.lone += main_path; o.lone.push_back('\n'); }
    bool read_string(read_t id, std::string &out) { return b.read_string(id, out); }
    bool has_cycle() { return b.has_cycle(); }
};
template <class GT> struct GraphDbg { static void take(GT &) {} };
uint64_t g_dbg_calls = 0, g_dbg_skipped = 0, g_dbg_spliced = 0, g_dbg_spliced_nodes = 0;
uint64_t g_soa_stats[16];
template <> struct GraphDbg<SoaGraph> { static void take(SoaGraph &g) { const nsgpu::dg::Hdr &h = g.store().hdr; g_soa_stats[0] += h.st_splits, g_soa_stats[1] += h.st_detours, g_soa_stats[2] += h.st_walked, g_soa_stats[3] += h.st_seq_exc, g_soa_stats[4] += h.st_cycles_run, g_soa_stats[5] += h.st_full_walk, g_soa_stats[6] += h.st_dis, g_soa_stats[7] += h.n_nodes - h.live_nodes, g_soa_stats[8] += h.st_routes, g_soa_stats[9] += h.st_route_ctx, g_soa_stats[10] += h.st_ctx, g_soa_stats[11] += h.st_regrow, g_soa_stats[12] += h.st_unreach_par; } };
extern "C" uint64_t harness_soa_stat(int i) { return g_soa_stats[i]; }
template <> struct GraphDbg<ShadowGraph> { static void take(ShadowGraph &g) { GraphDbg<SoaGraph>::take(g.b); } };
template <> struct GraphDbg<ContigGraph> { static void take(ContigGraph &g) { g_dbg_calls += g.dbg_cycles_calls; g_dbg_skipped += g.dbg_cycles_skipped; g_dbg_spliced += g.dbg_spliced; g_dbg_spliced_nodes += g.dbg_spliced_nodes; } };

// streams_out[0..6] = genome, lone, id, pos, type, base, complement; streams_out[7] = metaData (malloc'ed)
template <class GT>
static int harness_consensus_t(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, uint32_t thr, const uint64_t *salts,
                      int m_k, int m_w, int mci, uint64_t edge_thr, int run_checks, uint8_t **streams_out, uint64_t *lens_out,
                      harness_cons_stats *st, uint32_t id_base)
{
    std::vector<std::string> reads(N);
    static const char dna[4] = {'A', 'T', 'C', 'G'};
    std::string folded(bases + off[0], bases + off[N]);
    for (auto &c : folded) c = dna[(c & 2) | ((c & 4) >> 2)];
    uint64_t total = 0;
    for (uint32_t r = 0; r < N; ++r) { reads[r] = folded.substr(off[r] - off[0], off[r + 1] - off[r]); total += reads[r].size(); }
    std::vector<uint64_t> sk((size_t)N * n), keys((size_t)N * n);
    std::vector<uint32_t> start((size_t)(N + 1) * n), ids((size_t)N * n), nkeys(n);
    std::vector<uint64_t> foff(N + 1);
    for (uint32_t r = 0; r <= N; ++r) foff[r] = off[r] - off[0];
    oracle_sketch_reads(folded.data(), foff.data(), N, k, n, salts, sk.data());
    oracle_index_build(sk.data(), N, n, keys.data(), start.data(), ids.data(), nkeys.data());
    std::vector<uint8_t> in_graph(N, 0), rep(N, 0);
    for (uint32_t r = 0; r < N; ++r) rep[r] = reads[r].empty() ? 0 : (uint8_t)oracle_check_repetitive(reads[r].data(), reads[r].size());
    const size_t offset = N ? std::max<size_t>(1, (total / N) / 4) : 1;
    StreamSet out;
    memset(st, 0, sizeof(*st));
    std::vector<uint32_t> cand(N + 1);
    read_t cursor = 0;
    for (;;) {
        read_t first = cursor;
        while (first < N && in_graph[first]) ++first;
        if (first >= N) break;
        in_graph[first] = 1;
        cursor = first + 1;
        GT g;
        g.main_path = reads[first];
        g.start_pos = 0, g.end_pos = (ssize_t)reads[first].size(), g.first_read = first + id_base;
        const ssize_t init_start = g.start_pos, len = g.end_pos - g.start_pos;
        auto add_related = [&](ssize_t cur_pos) {
            const ssize_t o = cur_pos - g.start_pos;
            if (len == 0 || o < 0 || o >= (ssize_t)g.main_path.size()) return;
            const std::string fwd = g.main_path.substr((size_t)o, (size_t)len);
            std::string rcs;
            reverse_complement(fwd, rcs);
            for (int strand = 0; strand < 2; ++strand) {
                const std::string &w = strand ? rcs : fwd;
                uint64_t m;
                const uint64_t nc = oracle_filter_string(w.data(), w.size(), k, N, n, thr, salts, keys.data(), start.data(), ids.data(), nkeys.data(),
                                                         cand.data(), N, &m);
                st->count_minhash += nc;
                for (uint64_t ci = 0; ci < nc; ++ci) {
                    const read_t r = cand[ci];
                    if (g.num_edges() >= edge_thr) return;
                    if (rep[r] || in_graph[r]) continue;
                    ++st->count_minhash_not_in_graph;
                    if (reads[r].size() < 32) continue;
                    std::string q;
                    if (strand) reverse_complement(reads[r], q); else q = reads[r];
                    AlnOut ao;
                    ++st->n_align_calls;
                    if (!harness_align1(g.main_path, q, m_k, m_w, mci, ao)) continue;
                    in_graph[r] = 1;
                    ++st->count_aligner;
                    if (g.num_reads() == 0) {
                        const std::string seed = g.main_path;
                        g.main_path.clear();
                        g.initialize(seed, g.first_read, 0);
                        g.calculate_main_path_greedy();
                    }
                    double t0 = hnow();
                    g.update_graph(q, ao.edits, (ssize_t)ao.begin_offset, (ssize_t)ao.end_offset, r + id_base, (long)ao.rel_pos, strand == 1);
                    st->update_ms += hnow() - t0;
                    if (run_checks) {            // Consensus::checkRead / checkNoCycle under -DCHECKS (src/Consensus.cpp:328-337)
                        std::string back;
                        if (!g.read_string(r + id_base, back) || back != q) ++st->n_graph_check_fail;
                    }
                    t0 = hnow();
                    const std::string path_before = run_checks ? g.main_path : std::string();
                    g.path_changed_from = (size_t)-1;
                    g.calculate_main_path_greedy();
                    st->mainpath_ms += hnow() - t0;
                    if (run_checks) {            // path_changed_from is a lower bound of the prefix the recompute kept (the contig engine relies on it)
                        const size_t keep = std::min(std::min(g.path_changed_from, path_before.size()), g.main_path.size());
                        if (memcmp(path_before.data(), g.main_path.data(), keep) != 0) ++st->n_graph_check_fail;
                        if (g.path_changed_from == (size_t)-1 && path_before != g.main_path) ++st->n_graph_check_fail;
                    }
                    if (run_checks) {
                        std::string back;
                        if (!g.read_string(r + id_base, back) || back != q || g.has_cycle()) ++st->n_graph_check_fail;
                    }
                }
            }
        };
        bool too_many = false;
        const bool usable = len >= 32 && !rep[first];
        ssize_t cur_pos = g.start_pos;
        while (usable) {
            add_related(cur_pos);
            cur_pos += (ssize_t)offset;
            if (cur_pos + len > g.end_pos) break;
            else if (g.num_edges() >= edge_thr) { too_many = true; break; }
        }
        cur_pos = init_start - (ssize_t)offset;
        while (usable && !too_many) {
            if (cur_pos < g.start_pos) break;
            else if (g.num_edges() >= edge_thr) { too_many = true; break; }
            add_related(cur_pos);
            cur_pos -= (ssize_t)offset;
        }
        if (g.num_reads() == 0) {
            g.write_read_lone(out);
            out.lone_ids.push_back(first + id_base);
            out.reads_in_contig.push_back(1);
            ++st->n_lone;
        } else {
            double t0 = hnow();
            g.write_main_path(out);
            // the engine's route (walks guided by the reads' own bases, several reads in flight) unless NSGPU_EMIT_NO_SOURCE=1 asks
            // for the walk along the edges' read lists
            static const bool no_source = getenv("NSGPU_EMIT_NO_SOURCE") != nullptr;
            const std::function<ReadBases(read_t)> src = [&](read_t id) { const std::string &s = reads[id - id_base]; return ReadBases{s.data(), s.size()}; };
            if (no_source) g.write_reads(out);
            else g.write_reads(out, &src);
            st->write_ms += hnow() - t0;
            out.reads_in_contig.push_back((read_t)g.num_reads());
        }
        ++st->n_contigs;
        GraphDbg<GT>::take(g);
    }
    // round trip through the decoder
    {
        std::vector<std::pair<read_t, std::string>> rd;
        std::string err;
        std::vector<uint8_t> seen(N, 0);
        if (!decode_streams(out, rd, err)) st->n_bad_roundtrip = N + 1;
        else {
            for (auto &p : rd) { const uint32_t r = p.first - id_base; if (p.first < id_base || r >= N || seen[r] || p.second != reads[r]) ++st->n_bad_roundtrip; else seen[r] = 1; }
            for (uint32_t r = 0; r < N; ++r) st->n_bad_roundtrip += !seen[r];
        }
    }
    const std::string parts[8] = {out.genome, out.lone, out.id_bytes(), out.pos, out.type, out.base, out.complement,
                                  meta_data(N, std::vector<StreamSet>(1, out))};
    for (int i = 0; i < 8; ++i) {
        streams_out[i] = (uint8_t *)malloc(parts[i].size() + 1);
        memcpy(streams_out[i], parts[i].data(), parts[i].size());
        lens_out[i] = parts[i].size();
    }
    return 0;
}

extern "C" {
int harness_consensus(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, uint32_t thr, const uint64_t *salts,
                      int m_k, int m_w, int mci, uint64_t edge_thr, int run_checks, uint8_t **streams_out, uint64_t *lens_out,
                      harness_cons_stats *st, uint32_t id_base)
{
    const char *e = getenv("NSGPU_HARNESS_GRAPH");
    if (e && !strcmp(e, "soa")) return harness_consensus_t<SoaGraph>(bases, off, N, k, n, thr, salts, m_k, m_w, mci, edge_thr, run_checks, streams_out, lens_out, st, id_base);
    if (e && !strcmp(e, "both")) {
        const int rc = harness_consensus_t<ShadowGraph>(bases, off, N, k, n, thr, salts, m_k, m_w, mci, edge_thr, run_checks, streams_out, lens_out, st, id_base);
        st->n_graph_check_fail += ShadowGraph::mismatches();
        ShadowGraph::mismatches() = 0;
        return rc;
    }
    return harness_consensus_t<ContigGraph>(bases, off, N, k, n, thr, salts, m_k, m_w, mci, edge_thr, run_checks, streams_out, lens_out, st, id_base);
}

void harness_free(void *p) { free(p); }

}

// a15: the product's Edit::optimizeEditScript restatement (consensus.cpp) on a raw script (types 0 SAME 1 INSERT 2 DELETE)
extern "C" int64_t harness_optimize_edits(const uint8_t *types, const uint8_t *bases, const uint32_t *nums, uint32_t n, uint8_t *otypes, uint8_t *obases,
                                          uint32_t *onums, uint32_t cap, uint64_t *dis_out)
{
    std::vector<nsgpu::mm2::EditOp> in(n), out;
    for (uint32_t i = 0; i < n; ++i) in[i] = nsgpu::mm2::EditOp{types[i], bases[i], nums[i]};
    *dis_out = nsgpu::cons::optimize_edit_script(in, out);
    if (out.size() > cap) return -1;
    for (size_t i = 0; i < out.size(); ++i) otypes[i] = out[i].type, obases[i] = out[i].base, onums[i] = out[i].num;
    return (int64_t)out.size();
}
