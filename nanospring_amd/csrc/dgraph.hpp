// dgraph.hpp -- the consensus DAG (ConsensusGraph, src/ConsensusGraph.cpp:135-159 initialize, :400-557 updateGraph,
// :559-615 calculateMainPathGreedy, :617-651 clearMainPath, :653-714 removeCycles / walkAndPrune, :716-807 splitPath, :809-897 the
// node / edge bookkeeping; tie rules :33-91) as a structure of arrays with 32-bit ids, written ONCE for the host and for gfx950:
// one workgroup of 512 threads per accepted read runs these functions on the graph where it lies in HBM (graph_dev.hip); the
// CPU test harness and the emission of a finished contig (consensus_soa.cpp) run the very same code with a team of one.
//
// Layout (all indices 32 bit, position independent: a graph moves between HBM and host memory by plain copies)
//   Node 32 B: out[3] | out_ext | in[2] | in_ext | n_out n_in base on_main.  An out reference carries the base code of the edge's
//              sink in its top three bits (which way out has base b needs neither the edge nor the sink).  Lists beyond the inline
//              slots continue in chunks.
//   Edge 64 B: src sink count head tail ids[11].  count == number of read ids; ids beyond 11 continue in chunks of 15 (head .. tail).
//              The reference keeps the ids ordered (a sorted vector); nothing observable depends on the order (membership, the
//              smallest id, intersections, differences), so they are kept in arrival order.
//   path     : pe[] edge ids, pn[] node ids (m + 1), ps[] bases (m + 1), a deque inside arrays of cap_path entries (path_off).
// Ids are handed out in a fixed order (prefix sums, never atomics), so a team of 512 and a team of one build byte-identical arrays:
// the GPU tests compare the device arrays with the host run's after every update.
//
// What is parallel on the device: the appends of a read along the runs of the main path it follows (lanes over edges), the side
// excursions of a read (lanes over excursions: each leaves the path at a node of its own and comes back at another), the creation
// of new nodes and edges (lanes over nodes), the greedy choice at every node of the stretch the reference walks again (lanes over
// nodes; only where the choice differs from the path the walk is followed step by step), the copies of path stretches, the
// comparison of the old and the new consensus, the private copies of splitPath (by the reads' routes: split_routes_run; stretches of nodes
// with one way out: split_chain_run), the probes along chains with consecutive ids.  Sequential (thread 0): the stitching of detours, the
// walks of removeCycles, splits with more than 32 reads on their first edge.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define DG_HD __host__ __device__ inline
#define DG_COLD __host__ __device__ inline
#else
#define DG_HD inline
#define DG_COLD inline
#endif

namespace nsgpu {
namespace dg {

constexpr uint32_t NIL = 0xffffffffu;
constexpr uint32_t kEdgeInl = 11, kChunkIds = 15, kOutInl = 3, kInInl = 2;
constexpr uint32_t kRefMask = 0x1fffffffu;         // edge id of an out reference (the sink's base code above it)

struct Node { uint32_t out[kOutInl], out_ext, in[kInInl], in_ext; uint8_t n_out, n_in, base, on_main; };
struct Edge { uint32_t src, sink, count, head, tail, ids[kEdgeInl]; };
struct Chunk { uint32_t next, v[kChunkIds]; };
static_assert(sizeof(Node) == 32 && sizeof(Edge) == 64 && sizeof(Chunk) == 64, "graph records are 32 / 64 bytes");

enum : uint32_t {
    ERR_CAP = 1u,        // an array is full (the host grows the arrays before a launch from worst-case bounds: a bug or a pathological read)
    ERR_SCRIPT = 2u,     // an edit script that runs off the path
    ERR_DEGREE = 4u,     // more than 255 edges at one node
    ERR_WALK = 8u,       // a detour of the greedy walk that does not come back behind where it left (a cycle through the path)
    ERR_SCRATCH = 16u,   // the work area is too small
    ERR_ROOM = 32u,      // removeCycles stopped in front of a split whose copies do not fit (Hdr::need_*): nothing of that split has been done, the
                         // graph is whole -- the host makes room and runs removeCycles again (the one error that is an answer, not a failure)
};

// one op of the script: type (0 SAME, 1 INSERT, 2 DELETE) | base << 2 | num << 10
DG_HD uint32_t op_make(uint32_t type, uint32_t base, uint32_t num) { return type | (base << 2) | (num << 10); }
DG_HD uint32_t op_type(uint32_t o) { return o & 3u; }
DG_HD uint32_t op_base(uint32_t o) { return (o >> 2) & 0xffu; }
DG_HD uint32_t op_num(uint32_t o) { return o >> 10; }
constexpr uint32_t kOpMaxNum = (1u << 22) - 1;

struct Hdr {
    uint32_t n_nodes, n_edges, n_chunks;             // ids handed out so far
    uint32_t live_nodes, live_edges;                  // numNodes / numEdges of the reference
    uint32_t cap_nodes, cap_edges, cap_chunks, cap_path, cap_wk, cap_multi;
    uint32_t pos_bias;                                // added to pidx[] entries: the arrays' moves by the host (re-centring) leave pidx[] valid
    uint32_t path_off, m;                             // the path's edges are pe[path_off .. path_off + m), its nodes pn / ps[path_off .. path_off + m]
    uint32_t right_off, left_off, right_unch, left_unch;     // rightMostUnchangedNodeOffset / leftMost... and the nodes
    uint32_t n_multi, multi_n;                        // side nodes with more than one edge in; entries of multi_list (a superset, may hold stale ids)
    uint32_t epoch;
    uint32_t err;
    // the last update, for the main-path recompute behind it: its tables stay in wk[0 .. upd_wk)
    uint32_t upd_wk, upd_n_ops, upd_n_exc, upd_nodes0, upd_edges0;
    uint32_t upd_touch_lo, upd_touch_hi, have_touch;   // path nodes [lo, hi] are the ones whose edges the update changed (NIL: none)
    uint32_t cons_from;                               // path nodes from this index on are known to choose the path's edge (NIL: nothing known)
    // what an update + recompute reports to the host
    uint32_t initial;                                 // the read's first node
    uint32_t P, S, old_len, new_len;                  // the consensus before and after share P bases in front and S at the end
    uint32_t ending_id, starting_id;                  // the smallest read id on the path's last / first edge
    uint32_t st_splits, st_detours, st_walked, st_seq_exc, st_cycles_run, st_full_walk, st_dis;
    uint32_t stage;                                   // of the recompute: 2 = choosing (nothing changed yet), 3 = changing the graph
    uint32_t dbg_flags;                               // tests: 1 = excursions one at a time, 2 = removeCycles by the reference's full walk, 4 = a longer path always moves its left part, 8 = chain runs of splitPath and probes with a team of one too, 16 = no probes, 32 = no chain runs, 64 = no splits by routes, 128 = a split goes over to routes only after 24 copies (step by step and by stretches before)
    uint32_t err_line;                                // where the first error was raised (dgraph.hpp line)
    uint32_t err_info[4];                             // what it was about (node / index ...)
    uint32_t st_search, st_steps, st_idscan, st_ctx;   // thread 0's loops: rejoin searches (entries looked at), detour steps, read ids compared in splitPath, its contexts
    uint32_t st_gap, st_ended, st_last[6];             // by-passed nodes in sum; walks that ended; the last recompute's R, Lf, m, la, lenF, touch_hi
    uint32_t st_pops, st_probes, st_anc, st_probed;    // removeCycles' walk: edges popped; probes asked; ancestor steps; nodes the probes covered
    uint32_t st_cyc[6];                               // removeCycles in parts (thread 0's clock): marking, finding the roots, splitPath: looking for stretches / stretches / the rest, the walks
    uint32_t st_tm[8];                                // ticks of the team's clock by phase: tables, runs, excursions, choices, stitching, writing, flags + P/S, removeCycles
    uint32_t st_routes, st_route_ctx;                 // splits by routes (split_routes_run) and the contexts they made
    uint32_t unreach_n, unreach_list_n, st_unreach_par, pad_;      // n_multi and multi_n as removeCycles left them when it found that NO side node with several ways in can be reached from the path (NIL: not known)
    uint32_t need_nodes, need_edges, need_chunks, st_regrow;     // ERR_ROOM: what the split in front of which removeCycles stopped needs; how often that happened
    uint32_t st_rt[4];                                // ... in parts (thread 0's clock): walking the routes, comparing them, the copies, taking the reads off the old edges + the rest
};
static_assert(sizeof(Hdr) % 16 == 0, "header is whole 16-byte words");

struct G {
    Hdr *h;
    Node *nodes; Edge *edges; Chunk *chunks; uint32_t *mark;
    uint32_t *pidx;                                   // per node: where it was last written into pn[] (minus Hdr::pos_bias); believed only if pn[] there still holds the node
    uint32_t *pe, *pn; uint8_t *ps;
    uint32_t *sv_e, *sv_n; uint8_t *sv_s;             // the path as it was, where the walk goes over it again (cap_path entries each, same positions)
    uint32_t *multi_list;
    uint32_t *wk;                                     // work area, cap_wk words
};

DG_HD uint32_t code_of(uint32_t b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; }

// ---------------------------------------------------------------------------------------------------------------------------
// teams: the functions below are written for a team of threads that all run the same statements (every loop bound and every
// branch around a team call is the same for all of them); `tid == 0` marks what one thread does alone.  HostTeam is the team of one.
// ---------------------------------------------------------------------------------------------------------------------------
struct HostTeam {
    uint32_t shared_[1];
    DG_HD uint32_t *shared() { return shared_; }                                    // a few KB the threads of a team share (the device's LDS)
    DG_HD uint32_t shared_words() const { return 0; }
    DG_HD uint32_t tid() const { return 0; }
    DG_HD uint32_t size() const { return 1; }
    DG_HD void sync() {}
    DG_HD uint32_t bcast(uint32_t v) { return v; }                                  // thread 0's value
    DG_HD uint32_t scan(uint32_t v, uint32_t &total) { total = v; return 0; }       // exclusive prefix sum over the team's threads
    DG_HD uint32_t min_all(uint32_t v) { return v; }
    DG_HD uint32_t max_all(uint32_t v) { return v; }
    DG_HD uint32_t clock() const { return 0; }                                      // a free-running counter (debug report)
    DG_HD void add_to(uint32_t *p, uint32_t v) { *p += v; }                         // a sum several threads contribute to
    DG_HD void min_to(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }              // a minimum ...
    DG_HD uint32_t peek(const uint32_t *p) const { return *p; }                      // ... and how such a word is read (on the device: past the first-level cache, which the others' atomic updates do not reach)
    DG_HD bool cas(uint32_t *p, uint32_t expect, uint32_t v) { if (*p != expect) return false; *p = v; return true; }      // one of several threads wins
    // Thread 0 working alone while the others wait for its orders (removeCycles): on the device a barrier belongs to a whole wavefront, so the
    // lanes that share thread 0's wavefront sit such a stretch out and the team that takes the orders is thread 0 + the other wavefronts.
    DG_HD bool helper() const { return false; }                                      // this thread takes thread 0's orders
    DG_HD uint32_t crew_rank() const { return 0; }                                   // rank among thread 0 + the helpers
    DG_HD uint32_t crew_size() const { return 1; }
};

// excursion record (8 words) in the update's tables
enum { X_A = 0, X_CUR = 1, X_K0 = 2, X_NODE0 = 3, X_EDGE0 = 4, X_FLAGS = 5, X_B = 6, X_NINS = 7, X_WORDS = 8 };
// piece of a re-walked stretch (4 words): kind, a, b, c
enum { PC_OLD = 0, PC_EDGE = 1, PC_CHAIN = 2 };

template <class T> struct Ops {
    G g;
    T &team;
    DG_HD Ops(const G &gg, T &t) : g(gg), team(t) {}

    DG_HD void fail_at(uint32_t line, uint32_t e) { if (!g.h->err) g.h->err_line = line; g.h->err |= e; }
    DG_HD bool failed() const { return g.h->err != 0; }

    // ---- allocation (sequential contexts: one thread; the parallel phases compute their ids from prefix sums) ----
    DG_HD uint32_t new_chunk() { Hdr &h = *g.h; if (h.n_chunks >= h.cap_chunks) { fail_at(__LINE__, ERR_CAP); return 0; } const uint32_t c = h.n_chunks++; g.chunks[c].next = NIL; return c; }
    DG_HD uint32_t new_node(uint32_t base)
    {
        Hdr &h = *g.h;
        if (h.n_nodes >= h.cap_nodes) { fail_at(__LINE__, ERR_CAP); return 0; }
        const uint32_t n = h.n_nodes++;
        Node &x = g.nodes[n];
        x.out_ext = x.in_ext = NIL, x.n_out = x.n_in = 0, x.base = (uint8_t)base, x.on_main = 0;
        g.mark[n] = 0, g.pidx[n] = NIL;
        ++h.live_nodes;
        return n;
    }

    // ---- a node's edge lists ----
    DG_HD uint32_t list_get(const uint32_t *inl, uint32_t n_inl, uint32_t ext, uint32_t i) const
    {
        if (i < n_inl) return inl[i];
        i -= n_inl;
        uint32_t c = ext;
        while (i >= kChunkIds) c = g.chunks[c].next, i -= kChunkIds;
        return g.chunks[c].v[i];
    }
    DG_HD void list_set(uint32_t *inl, uint32_t n_inl, uint32_t &ext, uint32_t i, uint32_t v)
    {
        if (i < n_inl) { inl[i] = v; return; }
        i -= n_inl;
        if (ext == NIL) ext = new_chunk();
        uint32_t c = ext;
        while (i >= kChunkIds) { if (g.chunks[c].next == NIL) { const uint32_t nc = new_chunk(); g.chunks[c].next = nc; } c = g.chunks[c].next; i -= kChunkIds; }
        g.chunks[c].v[i] = v;
    }
    DG_HD uint32_t out_ref(const Node &x, uint32_t i) const { return list_get(x.out, kOutInl, x.out_ext, i); }
    DG_HD uint32_t in_ref(const Node &x, uint32_t i) const { return list_get(x.in, kInInl, x.in_ext, i); }
    // a push whose list may need one more chunk, handed in by the caller (the parallel phases get their chunk ids from prefix sums)
    DG_HD static bool push_needs_chunk(uint32_t n_inl, uint32_t cnt) { return cnt >= n_inl && (cnt - n_inl) % kChunkIds == 0; }
    DG_HD void list_push_with(uint32_t *inl, uint32_t n_inl, uint32_t &ext, uint32_t cnt, uint32_t v, uint32_t fresh)
    {
        if (cnt < n_inl) { inl[cnt] = v; return; }
        const uint32_t i = cnt - n_inl, s = i % kChunkIds;
        uint32_t hops = i / kChunkIds;
        if (s == 0) {
            g.chunks[fresh].next = NIL;
            if (hops == 0) ext = fresh;
            else { uint32_t c = ext; while (--hops) c = g.chunks[c].next; g.chunks[c].next = fresh; }
            g.chunks[fresh].v[0] = v;
            return;
        }
        uint32_t c = ext;
        while (hops--) c = g.chunks[c].next;
        g.chunks[c].v[s] = v;
    }
    DG_HD void out_push_with(uint32_t n, uint32_t ref, uint32_t fresh)
    {
        Node &x = g.nodes[n];
        if (x.n_out == 255) { fail_at(__LINE__, ERR_DEGREE); return; }
        list_push_with(x.out, kOutInl, x.out_ext, x.n_out, ref, fresh);
        ++x.n_out;
    }
    DG_HD void in_push_with(uint32_t n, uint32_t e, uint32_t fresh)
    {
        Node &x = g.nodes[n];
        if (x.n_in == 255) { fail_at(__LINE__, ERR_DEGREE); return; }
        list_push_with(x.in, kInInl, x.in_ext, x.n_in, e, fresh);
        ++x.n_in;
    }
    DG_HD void out_push(uint32_t n, uint32_t ref)
    {
        Node &x = g.nodes[n];
        if (x.n_out == 255) { fail_at(__LINE__, ERR_DEGREE); return; }
        list_set(x.out, kOutInl, x.out_ext, x.n_out, ref);
        ++x.n_out;
    }
    DG_HD void in_push(uint32_t n, uint32_t e)
    {
        Node &x = g.nodes[n];
        if (x.n_in == 255) { fail_at(__LINE__, ERR_DEGREE); return; }
        list_set(x.in, kInInl, x.in_ext, x.n_in, e);
        ++x.n_in;
    }
    DG_HD void out_erase(uint32_t n, uint32_t i)
    {
        Node &x = g.nodes[n];
        for (uint32_t j = i; j + 1 < x.n_out; ++j) list_set(x.out, kOutInl, x.out_ext, j, list_get(x.out, kOutInl, x.out_ext, j + 1));
        --x.n_out;
    }
    DG_HD void in_erase(uint32_t n, uint32_t i)
    {
        Node &x = g.nodes[n];
        for (uint32_t j = i; j + 1 < x.n_in; ++j) list_set(x.in, kInInl, x.in_ext, j, list_get(x.in, kInInl, x.in_ext, j + 1));
        --x.n_in;
    }

    // ---- an edge's read ids ----
    DG_HD static bool append_needs_chunk(uint32_t count) { return count >= kEdgeInl && (count - kEdgeInl) % kChunkIds == 0; }
    // Edge::addRead (:24-28) with the chunk an id behind the last full one needs handed in
    DG_HD void append_id(uint32_t ei, uint32_t id, uint32_t fresh_chunk)
    {
        Edge &e = g.edges[ei];
        const uint32_t c = e.count;
        if (c < kEdgeInl) e.ids[c] = id;
        else {
            const uint32_t s = (c - kEdgeInl) % kChunkIds;
            if (s == 0) {
                g.chunks[fresh_chunk].next = NIL;
                if (c == kEdgeInl) e.head = fresh_chunk; else g.chunks[e.tail].next = fresh_chunk;
                e.tail = fresh_chunk;
            }
            g.chunks[e.tail].v[s] = id;
        }
        e.count = c + 1;
    }
    DG_HD void add_read_seq(uint32_t ei, uint32_t id) { const uint32_t c = append_needs_chunk(g.edges[ei].count) ? new_chunk() : NIL; append_id(ei, id, c); }
    // the ids of an edge into a list in the work area; returns their number
    DG_HD uint32_t ids_copy(const Edge &e, uint32_t *dst) const
    {
        const uint32_t n = e.count, ni = n < kEdgeInl ? n : kEdgeInl;
        for (uint32_t p = 0; p < ni; ++p) dst[p] = e.ids[p];
        uint32_t left = n - ni, c = e.head, w = ni;
        while (left) { const Chunk &k = g.chunks[c]; const uint32_t t = left < kChunkIds ? left : kChunkIds; for (uint32_t p = 0; p < t; ++p) dst[w++] = k.v[p]; left -= t; c = k.next; }
        return n;
    }
    DG_HD bool edge_has(const Edge &e, uint32_t id) const
    {
        const uint32_t n = e.count, ni = n < kEdgeInl ? n : kEdgeInl;
        for (uint32_t p = 0; p < ni; ++p) if (e.ids[p] == id) return true;
        uint32_t left = n - ni, c = e.head;
        while (left) { const Chunk &k = g.chunks[c]; const uint32_t t = left < kChunkIds ? left : kChunkIds; for (uint32_t p = 0; p < t; ++p) if (k.v[p] == id) return true; left -= t; c = k.next; }
        return false;
    }
    DG_HD uint32_t edge_min_id(const Edge &e) const
    {
        uint32_t best = NIL;
        const uint32_t n = e.count, ni = n < kEdgeInl ? n : kEdgeInl;
        for (uint32_t p = 0; p < ni; ++p) if (e.ids[p] < best) best = e.ids[p];
        uint32_t left = n - ni, c = e.head;
        while (left) { const Chunk &k = g.chunks[c]; const uint32_t t = left < kChunkIds ? left : kChunkIds; for (uint32_t p = 0; p < t; ++p) if (k.v[p] < best) best = k.v[p]; left -= t; c = k.next; }
        return best;
    }

    // ---- Node::getEdgeTo / getEdgeToSide / getBestEdgeOut / getBestEdgeIn (src/ConsensusGraph.cpp:33-81) ----
    DG_HD uint32_t edge_to(uint32_t n, uint32_t target) const
    {
        const Node &x = g.nodes[n];
        for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t e = out_ref(x, i) & kRefMask; if (g.edges[e].sink == target) return e; }
        return NIL;
    }
    DG_HD uint32_t edge_to_side(uint32_t n, uint32_t base) const
    {
        const Node &x = g.nodes[n];
        const uint32_t want = code_of(base);
        for (uint32_t i = 0; i < x.n_out; ++i) {
            const uint32_t r = out_ref(x, i), e = r & kRefMask, c = r >> 29;
            if (c != want) continue;
            const Node &s = g.nodes[g.edges[e].sink];
            if (!s.on_main && s.base == base) return e;
        }
        return NIL;
    }
    DG_HD uint32_t best_out(uint32_t n) const
    {
        const Node &x = g.nodes[n];
        uint32_t best = NIL, c = 0;
        for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t e = out_ref(x, i) & kRefMask, k = g.edges[e].count; if (k > c) c = k, best = e; }
        return best;
    }
    // getBestEdgeOut of up to U nodes at once (NIL: none): the nodes' records, then all their edges' counts, loaded side by side
    template <uint32_t U> DG_HD void best_out_many(const uint32_t *n, uint32_t *best) const
    {
        uint32_t refs[U][kOutInl], no[U], k[U][kOutInl];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (uint32_t q = 0; q < U; ++q) {
            no[q] = 0;
            if (n[q] != NIL) { const Node &x = g.nodes[n[q]]; no[q] = x.n_out; for (uint32_t i = 0; i < kOutInl; ++i) refs[q][i] = x.out[i]; }
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (uint32_t q = 0; q < U; ++q)
            for (uint32_t i = 0; i < kOutInl; ++i) k[q][i] = i < no[q] && no[q] <= kOutInl ? g.edges[refs[q][i] & kRefMask].count : 0u;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (uint32_t q = 0; q < U; ++q) {
            if (n[q] == NIL) { best[q] = NIL; continue; }
            if (no[q] > kOutInl) { best[q] = best_out(n[q]); continue; }
            uint32_t b = NIL, c = 0;
            for (uint32_t i = 0; i < kOutInl; ++i) if (i < no[q] && k[q][i] > c) c = k[q][i], b = refs[q][i] & kRefMask;
            best[q] = b;
        }
    }
    DG_HD uint32_t best_in(uint32_t n) const
    {
        const Node &x = g.nodes[n];
        uint32_t best = NIL, c = 0;
        for (uint32_t i = 0; i < x.n_in; ++i) { const uint32_t e = in_ref(x, i), k = g.edges[e].count; if (k > c) c = k, best = e; }
        return best;
    }

    // ---- side nodes with more than one edge in (what removeCycles acts on) ----
    DG_HD bool multi_in_side(uint32_t n) const { const Node &x = g.nodes[n]; return !x.on_main && x.n_in > 1; }
    DG_HD void note_multi(uint32_t n) { Hdr &h = *g.h; if (h.multi_n < h.cap_multi) g.multi_list[h.multi_n++] = n; else h.multi_n = h.cap_multi + 1; }     // (a full list: the full walk decides)
    DG_HD void set_on_main(uint32_t n, bool v)
    {
        const bool was = multi_in_side(n);
        g.nodes[n].on_main = v ? 1 : 0;
        const bool is = multi_in_side(n);
        g.h->n_multi += (uint32_t)is - (uint32_t)was;
        if (!was && is) note_multi(n);
    }
    // createEdge (:823-839) with the ids copied from a list in the work area
    DG_HD uint32_t new_edge(uint32_t s, uint32_t t, const uint32_t *ids, uint32_t n_ids)
    {
        Hdr &h = *g.h;
        if (h.n_edges >= h.cap_edges) { fail_at(__LINE__, ERR_CAP); return 0; }
        const uint32_t e = h.n_edges++;
        Edge &x = g.edges[e];
        x.src = s, x.sink = t, x.count = 0, x.head = x.tail = NIL;
        const uint32_t ni = n_ids < kEdgeInl ? n_ids : kEdgeInl;
        for (uint32_t i = 0; i < ni; ++i) x.ids[i] = ids[i];
        x.count = ni;
        for (uint32_t i = ni; i < n_ids; ++i) add_read_seq(e, ids[i]);
        const bool was = multi_in_side(t);
        out_push(s, e | (code_of(g.nodes[t].base) << 29));
        in_push(t, e);
        const bool is = multi_in_side(t);
        h.n_multi += (uint32_t)is - (uint32_t)was;
        if (!was && is) note_multi(t);
        ++h.live_edges;
        return e;
    }
    // removeEdge (:861-876): the source side drops the FIRST out-edge that leads to the same sink.  The quiet form leaves the graph's counters
    // to the caller (returns the change of n_multi): the lanes of a chain run (split_chain_run) sum theirs up.
    DG_HD int32_t remove_edge_quiet(uint32_t e, bool keep_in_source, bool keep_in_sink)
    {
        Edge &x = g.edges[e];
        int32_t d_multi = 0;
        if (!keep_in_source) {
            const Node &s = g.nodes[x.src];
            for (uint32_t i = 0; i < s.n_out; ++i) if (g.edges[out_ref(s, i) & kRefMask].sink == x.sink) { out_erase(x.src, i); break; }
        }
        if (!keep_in_sink) {
            const bool was = multi_in_side(x.sink);
            const Node &t = g.nodes[x.sink];
            for (uint32_t i = 0; i < t.n_in; ++i) if (in_ref(t, i) == e) { in_erase(x.sink, i); break; }
            d_multi = (int32_t)multi_in_side(x.sink) - (int32_t)was;
        }
        x.count = 0, x.src = x.sink = NIL;       // (ids are not handed out again: a dead edge stays dead)
        return d_multi;
    }
    DG_HD void remove_edge(uint32_t e, bool keep_in_source, bool keep_in_sink)
    {
        g.h->n_multi += (uint32_t)remove_edge_quiet(e, keep_in_source, keep_in_sink);
        --g.h->live_edges;
    }
    // removeNode (:878-897)
    DG_HD void remove_node(uint32_t n)
    {
        Hdr &h = *g.h;
        Node &x = g.nodes[n];
        if (x.base == 0) return;                  // (gone already: base 0 marks a node that was taken away)
        h.n_multi -= (uint32_t)multi_in_side(n);
        x.on_main = 1;                            // keeps the counter untouched while the node's edges go away
        for (uint32_t i = 0; i < x.n_in; ++i) remove_edge(in_ref(x, i), false, true);
        for (uint32_t i = 0; i < x.n_out; ++i) remove_edge(out_ref(x, i) & kRefMask, true, false);
        x.n_in = x.n_out = 0, x.on_main = 0, x.base = 0;
        --h.live_nodes;
    }
    // removeReadsFromEdge (:843-859): the ids of `rm` leave the edge; an edge without reads goes
    DG_HD uint32_t drop_reads(uint32_t ei, const uint32_t *rm, uint32_t n_rm)        // the ids left
    {
        Edge &e = g.edges[ei];
        const uint32_t n = e.count;
        if (n_rm >= n) { e.count = 0, e.head = e.tail = NIL; return 0; }      // (the callers' lists are sub-sets of the edge's: as many as it has = all of them)
        // compaction in place, position by position (the write position never overtakes the read position)
        uint32_t w = 0, rc = e.head, wc = e.head;            // chunks of the read / write positions once they are beyond the inline slots
        for (uint32_t p = 0; p < n; ++p) {
            uint32_t v;
            if (p < kEdgeInl) v = e.ids[p];
            else { const uint32_t s = (p - kEdgeInl) % kChunkIds; if (s == 0 && p != kEdgeInl) rc = g.chunks[rc].next; v = g.chunks[rc].v[s]; }
            bool drop = false;
            for (uint32_t q = 0; q < n_rm; ++q) if (rm[q] == v) { drop = true; break; }
            if (drop) continue;
            if (w < kEdgeInl) e.ids[w] = v;
            else { const uint32_t s = (w - kEdgeInl) % kChunkIds; if (s == 0 && w != kEdgeInl) wc = g.chunks[wc].next; g.chunks[wc].v[s] = v; }
            ++w;
        }
        e.count = w;
        e.tail = w > kEdgeInl ? wc : NIL;
        if (w <= kEdgeInl) e.head = NIL;
        return w;
    }
    DG_HD void remove_reads_from_edge(uint32_t ei, const uint32_t *rm, uint32_t n_rm) { if (drop_reads(ei, rm, n_rm) == 0) remove_edge(ei, false, false); }

    DG_HD uint32_t path_node(uint32_t i) const { return g.pn[g.h->path_off + i]; }
    // where node n lies on the path, if what was noted when it was written there still holds (NIL: look for it)
    DG_HD uint32_t path_index_of(uint32_t n) const
    {
        const Hdr &h = *g.h;
        const uint32_t at = g.pidx[n] + h.pos_bias;
        if (g.pidx[n] == NIL || at < h.path_off || at > h.path_off + h.m || g.pn[at] != n) return NIL;
        return at - h.path_off;
    }
    DG_HD uint32_t *order() const { return g.wk + g.h->cap_wk - 64; }       // what thread 0 tells its helpers (the last words of the work area)

    // ================================================================================================================
    // initialize (:135-159) + the first calculateMainPathGreedy: the seed read as a chain, all of it the main path
    // ================================================================================================================
    DG_HD void initialize(const uint8_t *seed, uint32_t len, uint32_t id)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        if (len == 0 || len > h.cap_nodes || len - 1 > h.cap_edges || len + 2 > h.cap_path) { if (tid == 0) fail_at(__LINE__, ERR_CAP); team.sync(); return; }
        const uint32_t off = (h.cap_path - len) / 2;
        for (uint32_t i = tid; i < len; i += nt) {
            Node &x = g.nodes[i];
            x.out_ext = x.in_ext = NIL, x.base = seed[i], x.on_main = 1;
            x.n_out = i + 1 < len ? 1 : 0, x.n_in = i ? 1 : 0;
            if (i + 1 < len) x.out[0] = i | (code_of(seed[i + 1]) << 29);
            if (i) x.in[0] = i - 1;
            g.mark[i] = 0;
            if (i + 1 < len) { Edge &e = g.edges[i]; e.src = i, e.sink = i + 1, e.count = 1, e.head = e.tail = NIL, e.ids[0] = id; g.pe[off + i] = i; }
            g.pn[off + i] = i, g.ps[off + i] = seed[i];
            g.pidx[i] = off + i;
        }
        team.sync();
        if (tid == 0) {
            h.n_nodes = h.live_nodes = len, h.n_edges = h.live_edges = len - 1, h.n_chunks = 0;
            h.path_off = off, h.m = len - 1, h.pos_bias = 0;
            h.right_off = h.m, h.left_off = 0, h.right_unch = len - 1, h.left_unch = 0;
            h.n_multi = h.multi_n = 0, h.epoch = 0;
            h.upd_wk = h.upd_n_ops = h.upd_n_exc = 0, h.upd_nodes0 = len, h.upd_edges0 = len - 1;
            h.upd_touch_lo = h.upd_touch_hi = NIL, h.have_touch = 0, h.cons_from = 0;      // (a chain: every node has its one way out)
            h.initial = 0, h.P = 0, h.S = 0, h.old_len = 0, h.new_len = len, h.ending_id = h.starting_id = id;
        }
        team.sync();
    }

    // ================================================================================================================
    // updateGraph (:400-557)
    // ================================================================================================================
    // words of the update's tables for a script of n ops
    DG_HD static uint32_t wk_update_words(uint32_t n_ops) { return 14 * (n_ops + 2); }

    // The read `id` with the edit script ops[0 .. n_ops) against the main path.  The script is the aligner's with the read's overhangs
    // written out: -begin_offset INSERTs in front when the read starts left of the path, end_offset INSERTs at the end when it runs
    // beyond it (:437-452, :533-541 do exactly insertNode for those); the walk starts at edge max(begin_offset, 0).
    DG_HD void update(const uint32_t *ops_in, uint32_t n_ops, int64_t begin_offset, int64_t end_offset, uint32_t id)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        const uint32_t m = h.m;
        if (wk_update_words(n_ops) > h.cap_wk) { if (tid == 0) fail_at(__LINE__, ERR_SCRATCH); team.sync(); return; }
        // ---- the unchanged stretch (:409-433) ----
        if (tid == 0) {
            if (begin_offset >= 0 || end_offset >= 0) {
                int64_t r = (int64_t)h.right_off < begin_offset ? (int64_t)h.right_off : begin_offset;
                if (r < (int64_t)h.left_off) r = (int64_t)h.left_off;
                h.right_off = (uint32_t)r;
                h.right_unch = path_node(h.right_off);
            } else {
                // (size_t arithmetic in the reference: a read that ends left of the path's first base wraps to a huge value)
                const int64_t v = (int64_t)m + end_offset;
                uint32_t l = v < 0 ? NIL : (uint32_t)v;
                if (l < h.left_off) l = h.left_off;
                if (l > h.right_off) l = h.right_off;
                h.left_off = l;
                h.left_unch = path_node(h.left_off);
            }
        }
        uint32_t tk = team.clock();
        auto lap = [&](int i) { const uint32_t now = team.clock(); if (tid == 0) h.st_tm[i] += now - tk; tk = now; };
        uint32_t *W = g.wk;
        uint32_t *ops = W;                        // [n_ops]      the script (copied: it may lie in host memory)
        uint32_t *op_at = ops + n_ops + 1;        // [n_ops + 1]  edgeInPath at the start of op k
        uint32_t *run_off = op_at + n_ops + 1;    // [n_ops + 1]  main-path appends in front of op k's
        uint32_t *ins_ord = run_off + n_ops + 1;  // [n_ops + 1]  INSERTs in front of op k
        uint32_t *ins_op = ins_ord + n_ops + 1;   // [n_ops]      op of the q-th INSERT
        uint32_t *app = ins_op + n_ops + 1;       // [n_ops + 1]  the existing side / junction edge op k follows (NIL: none); slot k of a SAME op = its first base's edge
        uint32_t *ex = app + n_ops + 1;           // [n_ops + 1] excursion records
        const uint32_t ei0 = begin_offset > 0 ? (begin_offset < (int64_t)m ? (uint32_t)begin_offset : m) : 0;
        for (uint32_t k = tid; k < n_ops; k += nt) ops[k] = ops_in[k];
        team.sync();
        // ---- pass A: positions.  An excursion is a maximal run of INSERT / DELETE ops; one starts at op 0 and behind every SAME (possibly
        // with no op at all: its junction is then the step from one SAME's last node to the next one's first) ----
        uint32_t carry_e = ei0, carry_r = 0, carry_i = 0, n_exc = 0;
        for (uint32_t base = 0; base < n_ops + 1; base += nt) {
            const uint32_t k = base + tid;
            uint32_t de = 0, dr = 0, di = 0, is_start = 0;
            if (k < n_ops) {
                const uint32_t o = ops[k], t = op_type(o);
                de = t == 0 ? op_num(o) : t == 2 ? 1u : 0u;
                dr = t == 0 && op_num(o) > 1 ? op_num(o) - 1 : 0u;
                di = t == 1 ? 1u : 0u;
            }
            if (k <= n_ops) is_start = k == 0 || op_type(ops[k - 1]) == 0 ? 1u : 0u;
            uint32_t te, tr, ti, tx;
            const uint32_t pe_ = team.scan(de, te), pr_ = team.scan(dr, tr), pi_ = team.scan(di, ti), px_ = team.scan(is_start, tx);
            if (k <= n_ops) {
                const uint64_t e2 = (uint64_t)carry_e + pe_;
                op_at[k] = e2 < m ? (uint32_t)e2 : m;
                run_off[k] = carry_r + pr_;
                ins_ord[k] = carry_i + pi_;
                if (di) ins_op[carry_i + pi_] = k;
                app[k] = NIL;
                if (is_start) ex[X_WORDS * (n_exc + px_) + X_A] = k;
            }
            { const uint64_t c = (uint64_t)carry_e + te; carry_e = c > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)c; }
            carry_r += tr, carry_i += ti, n_exc += tx;
        }
        team.sync();
        const uint32_t n_run = carry_r;
        for (uint32_t x = tid; x < n_exc; x += nt) {
            uint32_t *r = ex + X_WORDS * x;
            const uint32_t b = x + 1 < n_exc ? ex[X_WORDS * (x + 1) + X_A] - 1 : n_ops;
            r[X_B] = b, r[X_NINS] = ins_ord[b] - ins_ord[r[X_A]];
        }
        // a SAME run that leaves the path is not a script of this path
        {
            uint32_t bad = 0;
            for (uint32_t k = tid; k < n_ops; k += nt) { const uint32_t o = ops[k]; if (op_type(o) == 0 && ((uint64_t)op_at[k] + op_num(o) - 1 > m || op_num(o) == 0)) bad = 1; }
            if (team.max_all(bad)) { if (tid == 0) fail_at(__LINE__, ERR_SCRIPT); team.sync(); return; }
        }
        {   // the first path node the read's SAMEs touch (:470 `touch`), and how far the walk came
            uint32_t first_same = NIL;
            for (uint32_t k = tid; k < n_ops; k += nt) if (op_type(ops[k]) == 0) { first_same = k; break; }
            first_same = team.min_all(first_same);
            if (tid == 0) {
                h.upd_wk = wk_update_words(n_ops), h.upd_n_ops = n_ops, h.upd_n_exc = n_exc, h.upd_nodes0 = h.n_nodes, h.upd_edges0 = h.n_edges;
                h.upd_touch_lo = first_same != NIL ? op_at[first_same] : NIL, h.upd_touch_hi = op_at[n_ops], h.have_touch = 1;
            }
        }
        team.sync();
        lap(0);
        // ---- pass B: the read along the runs of the path it follows (the per-base loop of :486-500 on main-path edges) ----
        append_runs(n_ops, op_at, run_off, n_run, id);
        if (failed()) return;
        lap(1);
        // ---- pass C: the excursions ----
        // every side node an excursion can come to has one way in then: no two excursions can meet.  (Side nodes with several ways in that cannot be
        // reached from the path -- the heads of reads that start left of a contig whose front has moved on -- are out of every excursion's way: an
        // excursion starts at a path node and follows edges, or creates what it needs.  removeCycles has said so if nothing has changed since.)
        const bool parallel_exc = (h.n_multi == 0 || (h.n_multi == h.unreach_n && h.multi_n == h.unreach_list_n)) && !(h.dbg_flags & 1u);
        if (tid == 0 && parallel_exc && h.n_multi) ++h.st_unreach_par;
        const uint32_t nodes_base = h.n_nodes, edges_base = h.n_edges;
        if (parallel_exc) {
            uint32_t tot_n = 0, tot_e = 0;
            for (uint32_t base = 0; base < n_exc; base += nt) {
                const uint32_t x = base + tid;
                uint32_t nn = 0, ne = 0;
                if (x < n_exc) exc_follow(ops, n_ops, op_at, ins_ord, ins_op, app, ex + X_WORDS * x, nn, ne);
                uint32_t tn, te2;
                const uint32_t pn_ = team.scan(nn, tn), pe2 = team.scan(ne, te2);
                if (x < n_exc) { uint32_t *r = ex + X_WORDS * x; r[X_NODE0] = nodes_base + tot_n + pn_, r[X_EDGE0] = edges_base + tot_e + pe2; }
                tot_n += tn, tot_e += te2;
            }
            team.sync();
            if ((uint64_t)nodes_base + tot_n > h.cap_nodes || (uint64_t)edges_base + tot_e > h.cap_edges) { if (tid == 0) fail_at(__LINE__, ERR_CAP); team.sync(); return; }
            exc_create_range(ops, op_at, ins_ord, ins_op, ex, 0, n_exc, nodes_base, tot_n, id);
            if (tid == 0) h.n_nodes += tot_n, h.live_nodes += tot_n, h.n_edges += tot_e, h.live_edges += tot_e;
            team.sync();
        } else {
            // in the reference's order, one excursion at a time (a side node with two ways in may be reached by two of them)
            for (uint32_t x = 0; x < n_exc; ++x) {
                uint32_t nn = 0, ne = 0;
                uint32_t *r = ex + X_WORDS * x;
                if (tid == 0) {
                    exc_follow(ops, n_ops, op_at, ins_ord, ins_op, app, r, nn, ne);
                    r[X_NODE0] = h.n_nodes, r[X_EDGE0] = h.n_edges;
                    if ((uint64_t)h.n_nodes + nn > h.cap_nodes || (uint64_t)h.n_edges + ne > h.cap_edges) fail_at(__LINE__, ERR_CAP);
                }
                team.sync();
                if (failed()) return;
                nn = team.bcast(nn), ne = team.bcast(ne);
                exc_create_range(ops, op_at, ins_ord, ins_op, ex, x, x + 1, r[X_NODE0], nn, id);
                if (tid == 0) h.n_nodes += nn, h.live_nodes += nn, h.n_edges += ne, h.live_edges += ne, ++h.st_seq_exc;
                team.sync();
            }
        }
        if (failed()) return;
        // ---- pass D: the read on the existing side and junction edges it follows ----
        append_listed(app, n_ops + 1, id, !parallel_exc);
        lap(2);
        // ---- the read's first node (`initialNode`) ----
        if (tid == 0) {
            uint32_t ini = NIL;
            if (n_ops && op_type(ops[0]) == 0) ini = path_node(op_at[0]);
            else if (n_exc) {
                const uint32_t *r = ex;                                     // excursion 0 starts at op 0 without a node to start from
                if (r[X_NINS]) ini = r[X_NODE0];
                else if (r[X_B] < n_ops) ini = path_node(op_at[r[X_B]]);
            }
            h.initial = ini;
        }
        team.sync();
    }

    // pass B
    DG_HD void append_runs(uint32_t n_ops, const uint32_t *op_at, const uint32_t *run_off, uint32_t n_run, uint32_t id)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        uint32_t chunks_base = h.n_chunks;
        // (the table the items search: in the team's shared memory when it fits -- a dozen dependent reads per item otherwise)
        const uint32_t *ro = run_off;
        if (n_ops + 1 <= team.shared_words()) {
            uint32_t *sh = team.shared();
            team.sync();
            for (uint32_t k = tid; k <= n_ops; k += nt) sh[k] = run_off[k];
            team.sync();
            ro = sh;
        }
        // (four items per lane and round: the loads of four edges in flight at once -- a round is as long as its chain of dependent loads)
        constexpr uint32_t U = 4;
        for (uint32_t base = 0; base < n_run; base += nt * U) {
            const uint32_t t0 = base + tid * U;
            uint32_t e[U], need[U], nsum = 0, lo = 0;
            if (t0 < n_run) {
                // the op whose run holds item t: run_off[lo] <= t < run_off[hi]   (run_off[n_ops] = n_run)
                uint32_t hi = n_ops;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ro[mid] <= t0) lo = mid; else hi = mid; }
            }
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (uint32_t q = 0; q < U; ++q) {
                const uint32_t t = t0 + q;
                e[q] = NIL, need[q] = 0;
                if (t < n_run) { while (lo + 1 < n_ops && ro[lo + 1] <= t) ++lo; e[q] = g.pe[h.path_off + op_at[lo] + (t - ro[lo])]; }
            }
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (uint32_t q = 0; q < U; ++q) if (e[q] != NIL) { need[q] = append_needs_chunk(g.edges[e[q]].count) ? 1u : 0u; nsum += need[q]; }
            uint32_t tot;
            uint32_t p = team.scan(nsum, tot);
            if ((uint64_t)chunks_base + tot > h.cap_chunks) { if (tid == 0) fail_at(__LINE__, ERR_CAP); team.sync(); return; }
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (uint32_t q = 0; q < U; ++q) if (e[q] != NIL) { append_id(e[q], id, need[q] ? chunks_base + p : NIL); p += need[q]; }
            chunks_base += tot;
        }
        team.sync();
        if (tid == 0) h.n_chunks = chunks_base;
        team.sync();
    }
    // pass D (in order on one thread when two excursions may share an edge)
    DG_HD void append_listed(const uint32_t *app, uint32_t n, uint32_t id, bool sequential)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        if (sequential) {
            if (tid == 0) for (uint32_t k = 0; k < n; ++k) if (app[k] != NIL) add_read_seq(app[k], id);
            team.sync();
            return;
        }
        uint32_t chunks_base = h.n_chunks;
        for (uint32_t base = 0; base < n; base += nt) {
            const uint32_t k = base + tid;
            const uint32_t e = k < n ? app[k] : NIL;
            const uint32_t need = e != NIL && append_needs_chunk(g.edges[e].count) ? 1u : 0u;
            uint32_t tot;
            const uint32_t p = team.scan(need, tot);
            if ((uint64_t)chunks_base + tot > h.cap_chunks) { if (tid == 0) fail_at(__LINE__, ERR_CAP); team.sync(); return; }
            if (e != NIL) append_id(e, id, need ? chunks_base + p : NIL);
            chunks_base += tot;
        }
        team.sync();
        if (tid == 0) h.n_chunks = chunks_base;
        team.sync();
    }

    // One excursion [a, b): where it starts, how far it follows side edges that exist (insertNode's getEdgeToSide, :454-468), what it
    // has to create, and the edge on which it comes back to the path (the first base of the SAME behind it, :472-484).
    // flags: 1 junction edge to create, 2 junction edge exists
    DG_HD void exc_follow(const uint32_t *ops, uint32_t n_ops, const uint32_t *op_at, const uint32_t *ins_ord, const uint32_t *ins_op, uint32_t *app, uint32_t *r, uint32_t &nn, uint32_t &ne)
    {
        const uint32_t a = r[X_A], b = r[X_B], n_ins = r[X_NINS], q0 = ins_ord[a];
        // the node the excursion leaves from: the last node of the SAME in front of it
        uint32_t cur = NIL;
        if (a > 0) { const uint32_t o = ops[a - 1]; cur = path_node(op_at[a - 1] + op_num(o) - 1); }
        uint32_t k0 = 0;
        if (cur != NIL)
            for (; k0 < n_ins; ++k0) {
                const uint32_t k = ins_op[q0 + k0];
                const uint32_t e = edge_to_side(cur, op_base(ops[k]));
                if (e == NIL) break;
                app[k] = e;
                cur = g.edges[e].sink;
            }
        nn = n_ins - k0;
        ne = nn;
        if (a == 0 && nn) --ne;                       // the read's first node has no edge in
        uint32_t fl = 0;
        r[X_CUR] = cur, r[X_K0] = k0;
        // the junction (b is a SAME op unless the script ends here)
        if (b < n_ops) {
            const uint32_t target = path_node(op_at[b]);
            if (nn) fl = 1, ++ne;
            else if (cur != NIL) {
                const uint32_t e = edge_to(cur, target);
                if (e != NIL) { app[b] = e; fl = 2; } else fl = 1, ++ne;
            }
        }
        r[X_FLAGS] = fl;
    }

    // creates what the excursions [x0, x1) have to create: lanes over the new nodes (each with its edge in), then over the excursions
    // (the way out of the node the follow ended at, the junction edge).  New edges of an excursion: [edges into its new nodes][junction].
    DG_HD void exc_create_range(const uint32_t *ops, const uint32_t *op_at, const uint32_t *ins_ord, const uint32_t *ins_op, uint32_t *ex, uint32_t x0, uint32_t x1,
                                uint32_t nodes_base, uint32_t tot_n, uint32_t id)
    {
        const uint32_t tid = team.tid(), nt = team.size();
        for (uint32_t t = tid; t < tot_n; t += nt) {
            const uint32_t nid = nodes_base + t;
            // the excursion that owns node nid: the last x with NODE0 <= nid (those that create nothing share NODE0 with the next that does)
            uint32_t lo = x0, hi = x1;
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ex[X_WORDS * mid + X_NODE0] <= nid) lo = mid; else hi = mid; }
            const uint32_t *r = ex + X_WORDS * lo;
            const uint32_t a = r[X_A], b = r[X_B], k0 = r[X_K0], nn = r[X_NINS] - k0, j = nid - r[X_NODE0];
            const uint32_t q = ins_ord[a] + k0 + j;                         // this node's INSERT
            const uint32_t base = op_base(ops[ins_op[q]]);
            const bool first_has_in = r[X_CUR] != NIL;
            const uint32_t e0 = r[X_EDGE0];
            const uint32_t e_in = first_has_in ? e0 + j : (j ? e0 + j - 1 : NIL);
            const uint32_t n_in_edges = first_has_in ? nn : nn - 1;
            Node &nd = g.nodes[nid];
            nd.out_ext = nd.in_ext = NIL, nd.base = (uint8_t)base, nd.on_main = 0;
            g.mark[nid] = 0, g.pidx[nid] = NIL;
            nd.n_in = e_in != NIL ? 1 : 0;
            if (e_in != NIL) nd.in[0] = e_in;
            if (j + 1 < nn) nd.n_out = 1, nd.out[0] = (first_has_in ? e0 + j + 1 : e0 + j) | (code_of(op_base(ops[ins_op[q + 1]])) << 29);
            else if (r[X_FLAGS] & 1u) nd.n_out = 1, nd.out[0] = (e0 + n_in_edges) | (code_of(g.nodes[path_node(op_at[b])].base) << 29);
            else nd.n_out = 0;
            if (e_in != NIL) { Edge &e = g.edges[e_in]; e.src = j ? nid - 1 : r[X_CUR], e.sink = nid, e.count = 1, e.head = e.tail = NIL, e.ids[0] = id; }
        }
        team.sync();
        Hdr &h = *g.h;
        uint32_t chunks_base = h.n_chunks;
        for (uint32_t base = x0; base < x1; base += nt) {
            const uint32_t x = base + tid;
            uint32_t need_o = 0, need_i = 0, nn = 0, e0 = 0, n_in_edges = 0, target = NIL, b = 0;
            bool do_out = false, do_in = false;
            const uint32_t *r = ex + X_WORDS * (x < x1 ? x : x0);
            if (x < x1) {
                b = r[X_B], nn = r[X_NINS] - r[X_K0], e0 = r[X_EDGE0];
                const bool first_has_in = r[X_CUR] != NIL;
                n_in_edges = nn ? (first_has_in ? nn : nn - 1) : 0;
                do_in = (r[X_FLAGS] & 1u) != 0;
                do_out = (nn && first_has_in) || (do_in && !nn);          // the node the follow ended at gets its one new way out
                if (do_in) target = path_node(op_at[b]);
                need_o = do_out && push_needs_chunk(kOutInl, g.nodes[r[X_CUR]].n_out) ? 1u : 0u;
                need_i = do_in && push_needs_chunk(kInInl, g.nodes[target].n_in) ? 1u : 0u;
            }
            uint32_t tot;
            const uint32_t p = team.scan(need_o + need_i, tot);
            if ((uint64_t)chunks_base + tot > h.cap_chunks) { if (tid == 0) fail_at(__LINE__, ERR_CAP); team.sync(); return; }
            if (x < x1) {
                if (nn && do_out) out_push_with(r[X_CUR], e0 | (code_of(op_base(ops[ins_op[ins_ord[r[X_A]] + r[X_K0]]])) << 29), chunks_base + p);
                if (do_in) {
                    const uint32_t je = e0 + n_in_edges;
                    Edge &e = g.edges[je];
                    e.src = nn ? r[X_NODE0] + nn - 1 : r[X_CUR], e.sink = target, e.count = 1, e.head = e.tail = NIL, e.ids[0] = id;
                    if (!nn) out_push_with(r[X_CUR], je | (code_of(g.nodes[target].base) << 29), chunks_base + p);
                    in_push_with(target, je, chunks_base + p + need_o);              // (a main-path node is the target of at most one junction per read)
                }
            }
            chunks_base += tot;
        }
        team.sync();
        if (tid == 0) h.n_chunks = chunks_base;
        team.sync();
    }

    // ================================================================================================================
    // calculateMainPathGreedy (:559-615) with clearMainPath (:617-651) and removeCycles behind it
    // ================================================================================================================
    // The reference drops the path right of rightMostUnchangedNode and left of leftMostUnchangedNode and walks both parts again, one
    // getBestEdgeOut / getBestEdgeIn after the other.  Here every node of the two stretches is asked for its choice at once (lanes over
    // nodes); where the choice is the path's old edge the walk would have taken it, so only the nodes that choose differently are
    // followed step by step (thread 0) until the walk is back on the old path -- through nodes this very update created in whole
    // chains, whose ids are consecutive and whose choice is forced (one edge in, one edge out).
    // piece: kind, a, b, c, offset (5 words); gap of by-passed old nodes: lo, hi, offset (3 words)
    DG_HD uint32_t flat_find(const uint32_t *recs, uint32_t stride, uint32_t off_word, uint32_t n, uint32_t t) const
    {
        uint32_t lo = 0, hi = n;                 // recs[lo].off <= t < recs[hi].off
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (recs[stride * mid + off_word] <= t) lo = mid; else hi = mid; }
        return lo;
    }
    // first position where a and b differ, at most n (all threads get the result)
    DG_HD uint32_t lcp(const uint8_t *a, const uint8_t *b, uint32_t n)
    {
        const uint32_t tid = team.tid(), nt = team.size();
        for (uint32_t base = 0; base < n; base += nt * 16) {
            uint32_t first = NIL;
            const uint32_t lo = base + tid * 16, hi = lo + 16 < n ? lo + 16 : n;
            for (uint32_t i = lo; i < hi; ++i) if (a[i] != b[i]) { first = i; break; }
            first = team.min_all(first);
            if (first != NIL) return first;
        }
        return n;
    }
    // number of equal bytes at the ends of a[0 .. n) and b[0 .. n) read backwards from a_end / b_end (exclusive ends)
    DG_HD uint32_t lcs(const uint8_t *a_end, const uint8_t *b_end, uint32_t n)
    {
        const uint32_t tid = team.tid(), nt = team.size();
        for (uint32_t base = 0; base < n; base += nt * 16) {
            uint32_t first = NIL;
            const uint32_t lo = base + tid * 16, hi = lo + 16 < n ? lo + 16 : n;
            for (uint32_t i = lo; i < hi; ++i) if (a_end[-(ptrdiff_t)i - 1] != b_end[-(ptrdiff_t)i - 1]) { first = i; break; }
            first = team.min_all(first);
            if (first != NIL) return first;
        }
        return n;
    }

    struct Stitch { uint32_t *pc; uint32_t cap_pc, n_pc, len; uint32_t *gp; uint32_t cap_gp, n_gp, gap_len; bool ended; };
    DG_HD void emit_piece(Stitch &S, uint32_t kind, uint32_t a, uint32_t b, uint32_t c, uint32_t l)
    {
        if (S.n_pc >= S.cap_pc) { fail_at(__LINE__, ERR_SCRATCH); return; }
        uint32_t *p = S.pc + 5 * S.n_pc++;
        p[0] = kind, p[1] = a, p[2] = b, p[3] = c, p[4] = S.len;
        S.len += l;
    }
    DG_HD void emit_gap(Stitch &S, uint32_t lo, uint32_t hi)
    {
        if (lo >= hi) return;
        if (S.n_gp >= S.cap_gp) { fail_at(__LINE__, ERR_SCRATCH); return; }
        uint32_t *p = S.gp + 3 * S.n_gp++;
        p[0] = lo, p[1] = hi, p[2] = S.gap_len;
        S.gap_len += hi - lo;
    }
    // the excursion of this update that created edge e
    DG_HD const uint32_t *exc_of_edge(uint32_t e) const
    {
        const Hdr &h = *g.h;
        const uint32_t n_ops = h.upd_n_ops;
        const uint32_t *ex = g.wk + 6 * (n_ops + 1);
        uint32_t lo = 0, hi = h.upd_n_exc;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ex[X_WORDS * mid + X_EDGE0] <= e) lo = mid; else hi = mid; }
        return ex + X_WORDS * lo;
    }
    DG_HD const uint32_t *upd_ops() const { return g.wk; }
    DG_HD const uint32_t *upd_op_at() const { return g.wk + (g.h->upd_n_ops + 1); }

    // Thread 0 asks its helpers: how far does the stretch of side nodes behind node X0 go on with CONSECUTIVE ids -- node X0 + i with one way on
    // (edge E0 + i, which leads to node X0 + i + 1)?  Chains are made with consecutive ids (an excursion's nodes, the copies of a split), so a
    // walk along one -- the greedy walk turning into a side branch, splitPath looking for the end of a by-passed stretch -- need not chase
    // pointers node by node: the helpers look at a few hundred nodes at once.  kind 0: forward, nodes with one edge in and one out; 1: backward
    // (in-edges E0 - i from node X0 - i - 1; 3: the same, and the stretch's nodes get the epoch's mark); 2: forward, any number of edges in, and the stretch is written into the chain arrays (from index
    // order[7], the edge into X0 being order[5]).  Returns the number of nodes of the stretch (0: X0 itself does not qualify).
    DG_HD uint32_t probe(uint32_t kind, uint32_t X0, uint32_t E0, uint32_t maxn, uint32_t e_into, uint32_t fill_at)
    {
        uint32_t *ord = order();
        ord[1] = kind, ord[2] = X0, ord[3] = E0, ord[4] = maxn, ord[5] = e_into, ord[6] = maxn, ord[7] = fill_at;
        (void)team.bcast(2);
        probe_run();
        const uint32_t L = team.peek(&ord[6]);
        ++g.h->st_probes, g.h->st_probed += L;
        return L;
    }
    DG_COLD void probe_run()
    {
        const Hdr &h = *g.h;
        uint32_t *ord = order();
        const uint32_t kind = ord[1], X0 = ord[2], E0 = ord[3], maxn = ord[4];
        const uint32_t cr = team.crew_rank(), nc = team.crew_size();
        const bool bwd = kind == 1 || kind == 3;
        for (uint32_t base = 0; base < maxn; base += nc) {
            const uint32_t i = base + cr;
            if (i < maxn) {
                bool ok = false;
                if (!bwd) {
                    const uint64_t n = (uint64_t)X0 + i, e = (uint64_t)E0 + i;
                    if (n < h.n_nodes && e < h.n_edges) {
                        const Node &nd = g.nodes[n];
                        ok = !nd.on_main && nd.n_out == 1 && (nd.out[0] & kRefMask) == e && (kind == 2 || nd.n_in == 1) && g.edges[e].sink == n + 1;
                    }
                } else if (i < X0 && i <= E0) {
                    const uint32_t n = X0 - i, e = E0 - i;
                    const Node &nd = g.nodes[n];
                    ok = !nd.on_main && nd.n_in == 1 && nd.in[0] == e && g.edges[e].src == n - 1;
                }
                if (!ok) team.min_to(&ord[6], i);
            }
            team.sync();
            if (team.peek(&ord[6]) < base + nc) break;
        }
        team.sync();
        if (kind == 3) {                                     // (the stretch's nodes are ancestors of a node removeCycles works on)
            const uint32_t L = team.peek(&ord[6]);
            for (uint32_t i = cr; i < L; i += nc) g.mark[X0 - i] = h.epoch;
            team.sync();
        }
        if (kind == 2) {
            const CycWk K = cyc_wk();
            const uint32_t L = team.peek(&ord[6]), at = ord[7];
            for (uint32_t i = cr; i < L; i += nc) K.chain[at + i] = i ? E0 + i - 1 : ord[5], K.chain[K.cap_chain + at + i] = X0 + i;
            team.sync();
        }
    }
    // the threads that take thread 0's orders while it works alone (1: a chain run of splitPath, 2: a probe, 3: a split by routes; 0: done)
    DG_HD void helpers_loop()
    {
        for (;;) {
            const uint32_t o = team.bcast(0);
            if (o == 0) break;
            if (o == 1) split_chain_run(cyc_wk()); else if (o == 3) split_routes_run(cyc_wk()); else probe_run();
        }
    }
    DG_HD bool probing() const { return (team.crew_size() > 1 && !(g.h->dbg_flags & 16u)) || (g.h->dbg_flags & 8u); }

    // thread 0: the walk from old node R to the right.  D: (index, chosen edge) of the old nodes in [R, m] whose choice is not the path's edge, ascending.
    DG_COLD void stitch_forward(Stitch &S, const uint32_t *D, uint32_t n_dis, uint32_t R, uint32_t m, uint32_t off)
    {
        Hdr &h = *g.h;
        uint32_t pos = R;
        for (uint32_t q = 0; q < n_dis && !S.ended && !failed(); ++q) {
            const uint32_t c = D[2 * q];
            uint32_t e = D[2 * q + 1];
            if (c < pos) continue;                              // by-passed by an earlier detour
            if (c > pos) emit_piece(S, PC_OLD, pos, c, 0, c - pos);
            pos = c;
            ++h.st_detours;
            uint32_t streak = 0;
            for (;;) {
                if (e == NIL) { S.ended = true; emit_gap(S, pos + 1, m + 1); break; }
                if (e >= h.upd_edges0) {                        // created by this update: a chain whose choices are forced
                    const uint32_t *r = exc_of_edge(e);
                    const uint32_t nn = r[X_NINS] - r[X_K0], e0 = r[X_EDGE0];
                    const bool first_has_in = r[X_CUR] != NIL;
                    const uint32_t n_in_edges = nn ? (first_has_in ? nn : nn - 1) : 0;
                    if (e < e0 + n_in_edges) {
                        const uint32_t t = e - e0, cnt = n_in_edges - t;
                        emit_piece(S, PC_CHAIN, e, cnt, r[X_NODE0] + (first_has_in ? t : t + 1), cnt);
                        h.st_walked += cnt;
                        e = (r[X_FLAGS] & 1u) ? e0 + n_in_edges : NIL;
                        continue;
                    }
                    const uint32_t j = upd_op_at()[r[X_B]];        // the junction: back on the path at this index
                    emit_piece(S, PC_EDGE, e, 1, g.edges[e].sink, 1);
                    if (j <= pos || j > m || g.pn[off + j] != g.edges[e].sink) { fail_at(__LINE__, ERR_WALK); break; }
                    emit_gap(S, pos + 1, j);
                    pos = j;
                    break;
                }
                const uint32_t nx = g.edges[e].sink;
                const uint32_t was_on = g.nodes[nx].on_main;
                emit_piece(S, PC_EDGE, e, was_on, nx, 1);
                ++h.st_walked;
                if (!was_on && probing()) {                    // a side branch: a stretch of it at once when its ids run on
                    const Node &sn = g.nodes[nx];
                    if (sn.n_out == 1 && sn.n_in == 1 && ++streak >= 2) {
                        const uint32_t o0 = sn.out[0] & kRefMask;
                        const uint32_t L = probe(0, nx, o0, 4 * team.crew_size() + 60, 0, 0);
                        streak = 0;
                        if (L >= 2) { emit_piece(S, PC_CHAIN, o0, L - 1, nx + 1, L - 1); h.st_walked += L - 1; e = o0 + L - 1; continue; }
                    }
                }
                if (was_on) {
                    uint32_t j = path_index_of(nx);
                    if (j == NIL || j <= pos || j > m) { j = pos + 1; while (j <= m && g.pn[off + j] != nx) ++j; h.st_search += j - pos; }
                    if (j > m) { h.err_info[0] = nx, h.err_info[1] = pos, h.err_info[2] = g.pidx[nx] + h.pos_bias - off, h.err_info[3] = e; fail_at(__LINE__, ERR_WALK); break; }
                    emit_gap(S, pos + 1, j);
                    pos = j;
                    break;
                }
                e = best_out(nx);
                ++h.st_steps;
                if (failed()) break;
            }
        }
        if (!S.ended && !failed() && pos < m) emit_piece(S, PC_OLD, pos, m, 0, m - pos);
    }
    // thread 0: the walk from old node Lf to the left.  D: (index, chosen edge in) of the old nodes in [0, Lf] whose choice is not the path's edge, descending.
    // Pieces in walk order (right to left): OLD (a, b) = old edges b-1 down to a; EDGE c = the edge's source; CHAIN = edges a, a-1, ..., sources c, c-1, ...
    DG_COLD void stitch_backward(Stitch &S, const uint32_t *D, uint32_t n_dis, uint32_t Lf, uint32_t off)
    {
        Hdr &h = *g.h;
        uint32_t pos = Lf;
        for (uint32_t q = 0; q < n_dis && !S.ended && !failed(); ++q) {
            const uint32_t c = D[2 * q];
            uint32_t e = D[2 * q + 1];
            if (c > pos) continue;
            if (c < pos) emit_piece(S, PC_OLD, c, pos, 0, pos - c);
            pos = c;
            ++h.st_detours;
            uint32_t streak = 0;
            for (;;) {
                if (e == NIL) { S.ended = true; emit_gap(S, 0, pos); break; }
                uint32_t nx = NIL, known_idx = NIL;
                if (e >= h.upd_edges0) {
                    const uint32_t *r = exc_of_edge(e);
                    const uint32_t nn = r[X_NINS] - r[X_K0], e0 = r[X_EDGE0];
                    const bool first_has_in = r[X_CUR] != NIL;
                    const uint32_t n_in_edges = nn ? (first_has_in ? nn : nn - 1) : 0;
                    if (e == e0 + n_in_edges && nn) {           // the junction edge out of the excursion's last new node
                        emit_piece(S, PC_EDGE, e, 0, r[X_NODE0] + nn - 1, 1);
                        ++h.st_walked;
                        e = n_in_edges ? e0 + n_in_edges - 1 : NIL;     // that node's one edge in (none: the read, and the path, start there)
                        continue;
                    }
                    if (e < e0 + n_in_edges) {                  // an edge into new node j: back along the chain
                        const uint32_t t = e - e0;
                        if (first_has_in) {
                            if (t) { emit_piece(S, PC_CHAIN, e, t, r[X_NODE0] + t - 1, t); h.st_walked += t; }
                            e = e0;                              // the edge from the node the follow ended at: an ordinary step below
                        } else {
                            emit_piece(S, PC_CHAIN, e, t + 1, r[X_NODE0] + t, t + 1);
                            h.st_walked += t + 1;
                            e = NIL;
                            continue;
                        }
                    }
                    nx = r[X_CUR];                               // (junction straight from an old node, or the chain's first edge)
                    if (r[X_K0] == 0 && r[X_A] > 0) known_idx = upd_op_at()[r[X_A] - 1] + op_num(upd_ops()[r[X_A] - 1]) - 1;
                } else nx = g.edges[e].src;
                const uint32_t was_on = g.nodes[nx].on_main;
                emit_piece(S, PC_EDGE, e, was_on, nx, 1);
                ++h.st_walked;
                if (!was_on && probing()) {
                    const Node &sn = g.nodes[nx];
                    if (sn.n_in == 1 && ++streak >= 2) {
                        const uint32_t i0 = sn.in[0];
                        const uint32_t L = probe(1, nx, i0, 4 * team.crew_size() + 60, 0, 0);
                        streak = 0;
                        if (L >= 2) { emit_piece(S, PC_CHAIN, i0, L - 1, nx - 1, L - 1); h.st_walked += L - 1; e = i0 - (L - 1); continue; }
                    }
                }
                if (was_on) {
                    uint32_t j = known_idx;
                    if (j == NIL) { j = path_index_of(nx); if (j != NIL && j >= pos) j = NIL; }
                    if (j == NIL) { j = pos; while (j > 0 && g.pn[off + j - 1] != nx) --j; h.st_search += pos - j; j = j > 0 ? j - 1 : NIL; }
                    if (j == NIL || j >= pos || g.pn[off + j] != nx) { fail_at(__LINE__, ERR_WALK); break; }
                    emit_gap(S, j + 1, pos);
                    pos = j;
                    break;
                }
                e = best_in(nx);
                ++h.st_steps;
                if (failed()) break;
            }
        }
        if (!S.ended && !failed() && pos > 0) emit_piece(S, PC_OLD, 0, pos, 0, pos);
    }

    // by-passed old nodes leave the path (they become side nodes: one with several edges in is work for removeCycles), the nodes of
    // the detours join it.  Gaps and chains: lanes over nodes; single detour nodes: thread 0.  Runs before the path arrays change.
    DG_COLD void apply_flags(const Stitch &S, uint32_t off, bool backward)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        for (uint32_t base = 0; base < S.gap_len; base += nt) {
            const uint32_t t = base + tid;
            uint32_t n = NIL, became = 0;
            if (t < S.gap_len) {
                const uint32_t *p = S.gp + 3 * flat_find(S.gp, 3, 2, S.n_gp, t);
                n = g.pn[off + p[0] + (t - p[2])];
                g.nodes[n].on_main = 0;
                became = g.nodes[n].n_in > 1 ? 1u : 0u;
            }
            uint32_t tot;
            const uint32_t w = team.scan(became, tot);
            if (became) { const uint32_t at = h.multi_n + w; if (at < h.cap_multi) g.multi_list[at] = n; }
            team.sync();
            if (tid == 0) { h.n_multi += tot; h.multi_n = h.multi_n + tot <= h.cap_multi ? h.multi_n + tot : h.cap_multi + 1; }
            team.sync();
        }
        for (uint32_t i = 0; i < S.n_pc; ++i) {
            const uint32_t *p = S.pc + 5 * i;
            if (p[0] == PC_CHAIN) for (uint32_t u = tid; u < p[2]; u += nt) g.nodes[backward ? p[3] - u : p[3] + u].on_main = 1;        // (new nodes: one edge in)
            else if (p[0] == PC_EDGE && !p[2] && tid == 0) set_on_main(p[3], true);
        }
        team.sync();
    }
    // entry t of a stitched stretch: forward -- edge, its sink, the sink's base; backward (walk order) -- edge, its source, the source's base
    DG_HD void piece_entry(const Stitch &S, uint32_t t, uint32_t off, bool backward, uint32_t &e, uint32_t &n, uint8_t &b) const
    {
        const uint32_t *p = S.pc + 5 * flat_find(S.pc, 5, 4, S.n_pc, t);
        const uint32_t u = t - p[4];
        if (p[0] == PC_OLD) {
            if (!backward) e = g.pe[off + p[1] + u], n = g.pn[off + p[1] + u + 1], b = g.ps[off + p[1] + u + 1];
            else e = g.pe[off + p[2] - 1 - u], n = g.pn[off + p[2] - 1 - u], b = g.ps[off + p[2] - 1 - u];
        } else if (p[0] == PC_EDGE) e = p[1], n = p[3], b = g.nodes[n].base;
        else { e = backward ? p[1] - u : p[1] + u, n = backward ? p[3] - u : p[3] + u, b = g.nodes[n].base; }
    }

    DG_HD void main_path()
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        const uint32_t m = h.m, R = h.right_off, Lf = h.left_off, off = h.path_off;
        if (tid == 0) h.stage = 2;
        uint32_t tk = team.clock();
        auto lap = [&](int i) { const uint32_t now = team.clock(); if (tid == 0) h.st_tm[i] += now - tk; tk = now; };
        if (Lf > R || R > m || h.upd_wk + 4096 > h.cap_wk) { if (tid == 0) fail_at(__LINE__, Lf > R || R > m ? ERR_WALK : ERR_SCRATCH); team.sync(); return; }
        uint32_t *W = g.wk + h.upd_wk;
        const uint32_t wcap = h.cap_wk - h.upd_wk - 64;
        uint32_t *D = W;                                     // disagreements, 2 words each
        const uint32_t cap_d = wcap / 4;                     // entries
        Stitch F, B;
        F.pc = W + wcap / 2, F.cap_pc = wcap / 8 / 5, F.gp = F.pc + wcap / 8, F.cap_gp = wcap / 16 / 3;
        B.pc = F.gp + wcap / 16, B.cap_pc = wcap / 8 / 5, B.gp = B.pc + wcap / 8, B.cap_gp = wcap / 16 / 3;
        F.n_pc = F.len = F.n_gp = F.gap_len = 0, F.ended = false;
        B.n_pc = B.len = B.n_gp = B.gap_len = 0, B.ended = false;
        // ---- to the right of old node R ----
        // Which nodes can choose differently from the path: the path from cons_from on was chosen by getBestEdgeOut on the counts of its time, and
        // an update since then changed the out-edges of the nodes [touch_lo, touch_hi] only (along path edges it raises the count of the edge that was
        // the first maximum already).  Nodes the update touched inside the part the reference keeps as it is stop being known-consistent; everything
        // in [R, m] below cons_from or at most touch_hi is asked, and the path's last node (the path may go on behind it).
        uint32_t cf = h.cons_from;
        const bool touched = h.have_touch != 0;
        if (touched && h.upd_touch_lo != NIL && R >= 1 && cf != NIL) {
            const uint32_t hi = h.upd_touch_hi < R - 1 ? h.upd_touch_hi : R - 1, lo = h.upd_touch_lo > Lf ? h.upd_touch_lo : Lf;
            if (lo <= hi && cf < hi + 1) cf = hi + 1;
        }
        uint32_t hi_f = m;                                   // ask [R, hi_f] and m
        if (cf != NIL) {
            uint32_t need = cf;                              // exclusive
            if (touched && h.upd_touch_hi != NIL && h.upd_touch_hi + 1 > need) need = h.upd_touch_hi + 1;
            if (!touched) need = m + 1;
            hi_f = need > m ? m : (need > R ? need - 1 : R);
        }
        uint32_t n_dis = 0;
        for (int part = 0; part < 2; ++part) {
            const uint32_t lo_i = part ? m : R, hi_i = part ? m : hi_f;
            if (part && hi_f >= m) break;
            constexpr uint32_t U = 4;                             // (four nodes per lane and round: their edges' counts are loaded side by side)
            for (uint64_t base = lo_i; base <= hi_i; base += (uint64_t)nt * U) {
                const uint64_t i0 = base + (uint64_t)tid * U;
                uint32_t nn[U], ch[U], dsum = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (uint32_t q = 0; q < U; ++q) nn[q] = i0 + q <= hi_i ? g.pn[off + (uint32_t)(i0 + q)] : NIL;
                best_out_many<U>(nn, ch);
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (uint32_t q = 0; q < U; ++q) {
                    const uint64_t i = i0 + q;
                    const bool d = i <= hi_i && ch[q] != (i < m ? g.pe[off + (uint32_t)i] : NIL);
                    if (!d) nn[q] = NIL; else ++dsum;              // (nn[q] != NIL from here on: a disagreement)
                }
                uint32_t tot;
                uint32_t p = team.scan(dsum, tot);
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (uint32_t q = 0; q < U; ++q) if (nn[q] != NIL) { if (n_dis + p < cap_d) D[2 * (n_dis + p)] = (uint32_t)(i0 + q), D[2 * (n_dis + p) + 1] = ch[q]; ++p; }
                n_dis += tot;
            }
        }
        team.sync();
        if (n_dis > cap_d) { if (tid == 0) fail_at(__LINE__, ERR_SCRATCH); team.sync(); return; }
        lap(3);
        if (tid == 0) { h.st_dis += n_dis; stitch_forward(F, D, n_dis, R, m, off); (void)team.bcast(0); } else if (team.helper()) helpers_loop();
        team.sync();
        if (failed()) return;
        F.n_pc = team.bcast(F.n_pc), F.len = team.bcast(F.len), F.n_gp = team.bcast(F.n_gp), F.gap_len = team.bcast(F.gap_len), F.ended = team.bcast(F.ended ? 1u : 0u) != 0;
        lap(4);
        // ---- to the left of old node Lf ----
        n_dis = 0;
        for (uint32_t base = 0; base <= Lf; base += nt) {
            const uint32_t k = base + tid;
            uint32_t d = 0, ch = NIL, i = 0;
            if (k <= Lf) { i = Lf - k; ch = best_in(g.pn[off + i]); d = ch != (i > 0 ? g.pe[off + i - 1] : NIL) ? 1u : 0u; }
            uint32_t tot;
            const uint32_t p = team.scan(d, tot);
            if (d && n_dis + p < cap_d) D[2 * (n_dis + p)] = i, D[2 * (n_dis + p) + 1] = ch;
            n_dis += tot;
        }
        team.sync();
        if (n_dis > cap_d) { if (tid == 0) fail_at(__LINE__, ERR_SCRATCH); team.sync(); return; }
        lap(3);
        if (tid == 0) { h.st_dis += n_dis; stitch_backward(B, D, n_dis, Lf, off); (void)team.bcast(0); } else if (team.helper()) helpers_loop();
        team.sync();
        if (failed()) return;
        B.n_pc = team.bcast(B.n_pc), B.len = team.bcast(B.len), B.n_gp = team.bcast(B.n_gp), B.gap_len = team.bcast(B.gap_len), B.ended = team.bcast(B.ended ? 1u : 0u) != 0;
        lap(4);
        // ---- the new path.  Only what changes is written: the stretches between the first and the last detour of each side, and -- when the
        // path's length changes -- whichever of the two parts next to the change is the shorter one moves (a contig grows at its ends: the read
        // lies near one of them, the megabase behind it stays where it is).  Everything is staged in the sv arrays at its final position first. ----
        const uint32_t la = B.len, lenF = F.len;
        if (tid == 0) { h.st_gap += F.gap_len + B.gap_len, h.st_ended += (F.ended ? 1u : 0u) + (B.ended ? 2u : 0u); h.st_last[0] = R, h.st_last[1] = Lf, h.st_last[2] = m, h.st_last[3] = la, h.st_last[4] = lenF, h.st_last[5] = h.upd_touch_hi; }
        const bool bwd_same = B.n_pc == 0 ? Lf == 0 : (B.n_pc == 1 && B.pc[0] == PC_OLD && B.pc[1] == 0 && B.pc[2] == Lf && !B.ended);
        const bool fwd_same = F.n_pc == 0 ? R == m : (F.n_pc == 1 && F.pc[0] == PC_OLD && F.pc[1] == R && F.pc[2] == m && !F.ended);
        uint32_t pre_nb = 0, pre_n = 0, suf_n = 0;
        if (!bwd_same && B.n_pc && B.pc[0] == PC_OLD && B.pc[2] == Lf) pre_nb = B.pc[2] - B.pc[1];
        if (!fwd_same && F.n_pc && F.pc[0] == PC_OLD && F.pc[1] == R) pre_n = F.pc[2] - F.pc[1];
        if (!fwd_same && !F.ended && F.n_pc > 1) { const uint32_t *p = F.pc + 5 * (F.n_pc - 1); if (p[0] == PC_OLD && p[2] == m) suf_n = p[2] - p[1]; }
        const int64_t delta = (int64_t)lenF - (int64_t)(m - R);
        const uint32_t left_size = la + (R - Lf) + pre_n;                 // edges left of the forward change
        const bool move_left = !fwd_same && suf_n && delta != 0 && (suf_n > left_size || (h.dbg_flags & 4u));
        const uint32_t off2 = off + Lf - la;                              // the path's start once the front is written
        const int64_t shift = move_left ? -delta : 0;                     // of everything left of the forward change's end
        if (off + Lf < la || (int64_t)off2 + shift < 0 || (uint64_t)off + R + lenF + 1 > h.cap_path) { if (tid == 0) fail_at(__LINE__, ERR_CAP); team.sync(); return; }
        if (tid == 0) h.stage = 3;
        team.sync();
        apply_flags(F, off, false);
        apply_flags(B, off, true);
        lap(6);
        if (!bwd_same) {
            for (uint32_t t = pre_nb + tid; t < la; t += nt) {
                uint32_t e, n; uint8_t b;
                piece_entry(B, t, off, true, e, n, b);
                const uint32_t at = off + Lf - 1 - t;                     // the t-th edge of the walk; its source is the node in front of it
                g.sv_e[at] = e, g.sv_n[at] = n, g.sv_s[at] = b;
            }
            team.sync();
            for (uint32_t t = pre_nb + tid; t < la; t += nt) { const uint32_t at = off + Lf - 1 - t; g.pe[at] = g.sv_e[at], g.pn[at] = g.sv_n[at], g.ps[at] = g.sv_s[at]; g.pidx[g.sv_n[at]] = at - h.pos_bias; }
            team.sync();
        }
        if (!fwd_same) {
            const uint32_t w_end = lenF - (move_left || delta == 0 ? suf_n : 0);      // forward entries [pre_n, w_end) are written
            if (move_left) {
                // the front and the kept part move by `shift` (source: the live arrays, which the front's write has just completed)
                const uint32_t n_left = la + (R - Lf) + pre_n;                         // edges; nodes: one more
                for (uint32_t t = tid; t <= n_left; t += nt) {
                    const uint32_t from = off2 + t, to = (uint32_t)((int64_t)from + shift);
                    g.sv_n[to] = g.pn[from], g.sv_s[to] = g.ps[from];
                    if (t < n_left) g.sv_e[to] = g.pe[from];
                }
            }
            for (uint32_t t = pre_n + tid; t < w_end; t += nt) {
                uint32_t e, n; uint8_t b;
                piece_entry(F, t, off, false, e, n, b);
                const uint32_t at = (uint32_t)((int64_t)(off + R + t) + shift);
                g.sv_e[at] = e, g.sv_n[at + 1] = n, g.sv_s[at + 1] = b;
            }
            team.sync();
            if (move_left) {
                const uint32_t n_left = la + (R - Lf) + pre_n;
                for (uint32_t t = tid; t <= n_left; t += nt) { const uint32_t to = (uint32_t)((int64_t)(off2 + t) + shift); g.pn[to] = g.sv_n[to], g.ps[to] = g.sv_s[to]; g.pidx[g.sv_n[to]] = to - h.pos_bias; if (t < n_left) g.pe[to] = g.sv_e[to]; }
            }
            for (uint32_t t = pre_n + tid; t < w_end; t += nt) { const uint32_t at = (uint32_t)((int64_t)(off + R + t) + shift); g.pe[at] = g.sv_e[at], g.pn[at + 1] = g.sv_n[at + 1], g.ps[at + 1] = g.sv_s[at + 1]; g.pidx[g.sv_n[at + 1]] = at + 1 - h.pos_bias; }
            team.sync();
        }
        lap(5);
        const uint32_t m2 = la + (R - Lf) + lenF;
        const uint32_t off3 = (uint32_t)((int64_t)off2 + shift);
        if (tid == 0) {
            h.path_off = off3, h.m = m2;
            h.right_off = la + (R - Lf), h.left_off = la;
            // every node from R on has just been asked or was known to be consistent; the kept part keeps what was known of it; nodes chosen
            // by getBestEdgeIn (the new front) are not trusted
            { uint32_t c2 = cf == NIL || cf > R ? R : cf; if (c2 < Lf) c2 = Lf; h.cons_from = c2 - Lf + la; }
            h.have_touch = 0;
            h.ending_id = edge_min_id(g.edges[g.pe[off3 + m2 - 1]]);
            h.starting_id = edge_min_id(g.edges[g.pe[off3]]);
            // what the consensus kept for sure: P bases in front (same index), S at the end (same distance from the end)
            const uint32_t Lo = m + 1, Ln = m2 + 1;
            uint32_t P, S;
            if (bwd_same && fwd_same) P = Ln, S = 0;
            else {
                P = bwd_same ? R + pre_n + 1 : 0;
                S = fwd_same ? m - (Lf - pre_nb) + 1 : suf_n;
            }
            const uint32_t mn = Lo < Ln ? Lo : Ln;
            if (P > mn) P = mn;
            if (P + S > mn) S = mn - P;
            h.P = P, h.S = S, h.old_len = Lo, h.new_len = Ln;
        }
        team.sync();
        lap(6);
    }
    // removeCycles behind the recompute (:606), and the unchanged stretch is the whole path again (:608-613).  Apart from the graph's size nothing
    // the caller sees changes here -- the consensus, the contig's span and the path are final when main_path() returns -- so the kernel reports
    // those first and the caller goes on while this part runs.
    DG_HD void finish_path()
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid();
        uint32_t tk = team.clock();
        remove_cycles();
        if (tid == 0) {
            h.st_tm[7] += team.clock() - tk;
            if (!h.err) {                                    // (ERR_ROOM: the same call again when there is room)
                h.right_unch = g.pn[h.path_off + h.m], h.right_off = h.m;
                h.left_unch = g.pn[h.path_off], h.left_off = 0;
            }
        }
        team.sync();
    }

    // ================================================================================================================
    // removeCycles (:653-691), walkAndPrune (:693-714), splitPath (:716-807)
    // ================================================================================================================
    struct CycWk { uint32_t *todo, *roots, *hits, *estack, *ctx, *lists, *copy, *defer, *chain, *route, *order; uint32_t cap_todo, cap_roots, cap_hits, cap_estack, cap_ctx, cap_lists, cap_copy, cap_defer, cap_chain, cap_route; };
    DG_HD CycWk cyc_wk() const
    {
        const Hdr &h = *g.h;
        uint32_t *W = g.wk + h.upd_wk;
        const uint32_t wcap = h.cap_wk - h.upd_wk - 64, u = wcap / 32;
        CycWk c;
        c.todo = W, c.cap_todo = u;
        c.roots = W + u, c.cap_roots = u;
        c.hits = W + 2 * u, c.cap_hits = 2 * u;            // pairs
        c.estack = W + 4 * u, c.cap_estack = 4 * u;
        c.ctx = W + 8 * u, c.cap_ctx = 4 * u / 8;          // 8 words each
        c.lists = W + 12 * u, c.cap_lists = 4 * u;
        c.copy = W + 16 * u, c.cap_copy = u;
        c.defer = W + 17 * u, c.cap_defer = 2 * u;
        c.chain = W + 19 * u, c.cap_chain = 2 * u / 3;        // a run of a chain: its edges, its nodes, its masks
        c.route = W + 21 * u, c.cap_route = 11 * u;           // the routes of a split's reads (split_routes_run)
        c.order = order();
        return c;
    }

    // Gives the reads of edge e0 (a side branch entering a node that has other ways in) a private copy of everything downstream until
    // the main path is reached again.  The reference's two-visit context stack, iteratively; read lists live in the work area.
    // ctx: new_pre, e, in_off, in_n, own_off, own_n, visited, old_cur.  Two things keep the memory at what the FORKS of the copied part
    // need, not its length (a by-passed stretch of the path is a chain of thousands of nodes): a context's list is given back when the
    // context is popped (its descendants are done by then: stack discipline), and a node with ONE way out is not a new context at all --
    // the walk goes on in the same one, its list filtered in place; whether the old node is left without edges (the reference looks at
    // the context's second visit) is looked at when the split is over (nothing can reach such a node in between).
    DG_COLD void split_path(const CycWk &K, uint32_t new_pre0, uint32_t e0)
    {
        Hdr &h = *g.h;
        ++h.st_splits;
        const uint32_t sp0 = team.clock();
        uint32_t lists_top = 0, n_ctx = 0, n_def = 0, n_done = 0;
        if (g.edges[e0].count > K.cap_lists) { fail_at(__LINE__, ERR_SCRATCH); return; }
        lists_top = ids_copy(g.edges[e0], K.lists);          // (a copy: e0's list dies with e0 during the first visit)
        {
            uint32_t *c = K.ctx;
            c[0] = new_pre0, c[1] = e0, c[2] = 0, c[3] = lists_top, c[4] = lists_top, c[5] = 0, c[6] = 0, c[7] = NIL;
            n_ctx = 1;
        }
        while (n_ctx && !failed()) {
            uint32_t *c = K.ctx + 8 * (n_ctx - 1);
            if (c[6]) {
                const uint32_t oc = c[7];
                lists_top = c[4];
                --n_ctx;
                if (oc != NIL && g.nodes[oc].n_in == 0 && g.nodes[oc].n_out == 0) remove_node(oc);
                continue;
            }
            // first visit: the reads of this branch that go down edge c[1]
            uint32_t new_pre = c[0], e = c[1];
            const uint32_t own_off = lists_top;
            uint32_t own_n = 0;
            if (n_ctx == 1 && c[2] == 0) {                       // the split's first edge: its own reads, all of them
                if (own_off + c[3] > K.cap_lists) { fail_at(__LINE__, ERR_SCRATCH); return; }
                for (uint32_t i = 0; i < c[3]; ++i) K.lists[own_off + i] = K.lists[i];
                own_n = c[3];
            } else {
                const Edge &ed = g.edges[e];
                for (uint32_t i = 0; i < c[3]; ++i) {
                    const uint32_t id = K.lists[c[2] + i];
                    if (ed.src != NIL && edge_has(ed, id)) { if (own_off + own_n >= K.cap_lists) { fail_at(__LINE__, ERR_SCRATCH); return; } K.lists[own_off + own_n++] = id; }
                }
            }
            lists_top = own_off + own_n;
            h.st_idscan += c[3] * g.edges[e].count; ++h.st_ctx;
            c[4] = own_off, c[5] = own_n, c[6] = 1;
            bool no_routes = (h.dbg_flags & 64u) != 0;
            while (own_n && !failed()) {
                // a split that has proven large: what hangs below this edge is taken by the whole team, read by read (split_routes_run)
                if (!no_routes && n_done >= ((h.dbg_flags & 128u) ? 24u : kRouteAfter) && own_n <= kRouteReads) {
                    const uint32_t r0 = team.clock();
                    K.order[11] = new_pre, K.order[12] = e, K.order[13] = own_off, K.order[14] = own_n, K.order[15] = 0, K.order[16] = 0, K.order[20] = 0, K.order[21] = 0, K.order[22] = 0, K.order[24] = n_done == 0 && n_ctx == 1 ? 1u : 0u;
                    (void)team.bcast(3);
                    split_routes_run(K);
                    h.st_cyc[3] += team.clock() - r0;
                    if (failed()) return;
                    if (K.order[16]) { n_done += K.order[17]; own_n = 0, lists_top = own_off; c[5] = 0; break; }
                    no_routes = true;                            // (a route longer than the work area: step by step)
                }
                // A stretch of nodes with one way out each (a by-passed piece of the old path, typically): not one step of this loop after the
                // other -- the whole team takes the stretch at once (split_chain_run)
                if (own_n <= kEdgeInl && ((team.crew_size() > 1 && !(h.dbg_flags & 32u)) || (h.dbg_flags & 8u))) {
                    uint32_t k = 0, ce = e;
                    const uint32_t d0 = team.clock();
                    const uint32_t lim = K.cap_chain < 4096 ? K.cap_chain : 4096;
                    uint32_t since = 0;
                    while (k < lim) {
                        const uint32_t X = g.edges[ce].sink;
                        const Node &xn = g.nodes[X];
                        if (xn.on_main || xn.n_out != 1) break;
                        const uint32_t o0 = xn.out[0] & kRefMask;
                        if (++since >= 3 && probing()) {          // the rest of the stretch by its ids, a few hundred nodes at a time
                            const uint32_t L = probe(2, X, o0, lim - k < 4 * team.crew_size() + 60 ? lim - k : 4 * team.crew_size() + 60, ce, k);
                            since = L >= 32 ? 3 : 0;
                            if (L >= 1) { k += L; ce = o0 + L - 1; continue; }
                        }
                        K.chain[k] = ce, K.chain[K.cap_chain + k] = X;
                        ++k;
                        ce = o0;
                    }
                    const uint32_t d1 = team.clock();
                    h.st_cyc[2] += d1 - d0;
                    if (k >= 8) {
                        if (n_def + k > K.cap_defer || (uint64_t)h.n_nodes + k > h.cap_nodes || (uint64_t)h.n_edges + k > h.cap_edges) { fail_at(__LINE__, n_def + k > K.cap_defer ? ERR_SCRATCH : ERR_CAP); return; }
                        K.order[1] = new_pre, K.order[2] = k, K.order[3] = own_off, K.order[4] = own_n, K.order[5] = n_def;
                        (void)team.bcast(1);
                        split_chain_run(K);
                        h.st_cyc[3] += team.clock() - d1;
                        const uint32_t k_eff = K.order[6];
                        n_def += k_eff;
                        h.st_ctx += k_eff, n_done += k_eff;
                        if (k_eff) {
                            new_pre = h.n_nodes - 1;                      // the copy of the stretch's last node
                            e = k_eff < k ? K.chain[k_eff] : ce;
                            // what is left of the reads behind the stretch, and which of them go down the next edge
                            const uint32_t mk = K.chain[2 * K.cap_chain + k_eff - 1];
                            const Edge &ed = g.edges[e];
                            uint32_t w = 0;
                            for (uint32_t i = 0; i < own_n; ++i) { const uint32_t id = K.lists[own_off + i]; if (((mk >> i) & 1u) && edge_has(ed, id)) K.lists[own_off + w++] = id; }
                            own_n = w, lists_top = own_off + w;
                            c[5] = own_n;
                            continue;
                        }
                    }
                }
                const uint32_t old_cur = g.edges[e].sink;
                ++n_done;
                remove_reads_from_edge(e, K.lists + own_off, own_n);
                if (g.nodes[old_cur].on_main) { new_edge(new_pre, old_cur, K.lists + own_off, own_n); c[7] = old_cur; break; }
                const uint32_t new_cur = new_node(g.nodes[old_cur].base);
                new_edge(new_pre, new_cur, K.lists + own_off, own_n);
                const Node &oc = g.nodes[old_cur];
                const uint32_t no = oc.n_out;
                if (no == 1) {                                   // one way on: the same context goes on
                    if (n_def >= K.cap_defer) { fail_at(__LINE__, ERR_SCRATCH); return; }
                    K.defer[n_def++] = old_cur;
                    e = out_ref(oc, 0) & kRefMask, new_pre = new_cur;
                    const Edge &ed = g.edges[e];
                    uint32_t w = 0;
                    for (uint32_t i = 0; i < own_n; ++i) { const uint32_t id = K.lists[own_off + i]; if (edge_has(ed, id)) K.lists[own_off + w++] = id; }
                    h.st_idscan += own_n * ed.count; ++h.st_ctx;
                    own_n = w, lists_top = own_off + w;
                    c[5] = own_n;
                    continue;
                }
                c[7] = old_cur;
                if (n_ctx + no > K.cap_ctx) { fail_at(__LINE__, ERR_SCRATCH); return; }
                for (uint32_t i = 0; i < no; ++i) {
                    uint32_t *d = K.ctx + 8 * n_ctx++;
                    d[0] = new_cur, d[1] = out_ref(oc, i) & kRefMask, d[2] = own_off, d[3] = own_n, d[4] = lists_top, d[5] = 0, d[6] = 0, d[7] = NIL;
                }
                break;
            }
        }
        for (uint32_t i = 0; i < n_def && !failed(); ++i) { const uint32_t oc = K.defer[i]; if (g.nodes[oc].n_in == 0 && g.nodes[oc].n_out == 0 && g.nodes[oc].on_main == 0) remove_node(oc); }
        h.st_cyc[4] += team.clock() - sp0;
    }
    // The team's part of split_path for a stretch of k nodes with one way out each: edges chain[0 .. k), nodes chain[cap + i], the reads O = lists[own_off ..
    // own_off + own_n) (at most 11), the node in front of the copy new_pre.  What the loop in split_path does node by node -- take the reads off the old
    // edge, copy the node, join the copy to the copy before it -- the lanes do for all nodes at once; which reads are still on the stretch at node i is the
    // AND of the edges' membership masks up to i (thread 0, one pass over k words).  Ids as the loop would hand them out: in order along the stretch.
    // order[]: 1 new_pre, 2 k, 3 own_off, 4 own_n, 5 n_def; out: 6 k_eff (the nodes copied: up to where the reads run out)
    DG_COLD void split_chain_run(const CycWk &K)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), cr = team.crew_rank(), nc = team.crew_size();
        const uint32_t new_pre = K.order[1], k = K.order[2], own_off = K.order[3], own_n = K.order[4], n_def = K.order[5];
        const uint32_t *CE = K.chain, *CX = K.chain + K.cap_chain;
        uint32_t *M = K.chain + 2 * K.cap_chain;
        uint32_t O[kEdgeInl];
        for (uint32_t q = 0; q < own_n; ++q) O[q] = K.lists[own_off + q];
        for (uint32_t i = cr; i < k; i += nc) {
            const Edge &ed = g.edges[CE[i]];
            uint32_t mk = 0;
            for (uint32_t q = 0; q < own_n; ++q) if (edge_has(ed, O[q])) mk |= 1u << q;
            M[i] = mk;
        }
        team.sync();
        if (tid == 0) {
            uint32_t acc = (1u << own_n) - 1u, k_eff = 0;
            for (; k_eff < k; ++k_eff) { acc &= M[k_eff]; if (!acc) break; M[k_eff] = acc; }
            K.order[6] = k_eff, K.order[7] = 0, K.order[8] = 0;
        }
        team.sync();
        const uint32_t k_eff = K.order[6];
        const uint32_t nbase = h.n_nodes, ebase = h.n_edges;
        uint32_t removed = 0, unmulti = 0;
        for (uint32_t i = cr; i < k_eff; i += nc) {
            const uint32_t mk = M[i], X = CX[i], oe = CE[i];
            uint32_t rm[kEdgeInl], n_rm = 0;
            for (uint32_t q = 0; q < own_n; ++q) if ((mk >> q) & 1u) rm[n_rm++] = O[q];
            if (drop_reads(oe, rm, n_rm) == 0) { const int32_t d = remove_edge_quiet(oe, false, false); ++removed; if (d < 0) ++unmulti; }
            const uint32_t nid = nbase + i, ne = ebase + i;
            Node &nd = g.nodes[nid];
            nd.out_ext = nd.in_ext = NIL, nd.base = g.nodes[X].base, nd.on_main = 0;
            g.mark[nid] = 0, g.pidx[nid] = NIL;
            nd.n_in = 1, nd.in[0] = ne;
            if (i + 1 < k_eff) nd.n_out = 1, nd.out[0] = (ne + 1) | (code_of(g.nodes[CX[i + 1]].base) << 29);
            else nd.n_out = 0;
            Edge &x = g.edges[ne];
            x.src = i ? nid - 1 : new_pre, x.sink = nid, x.count = n_rm, x.head = x.tail = NIL;
            for (uint32_t q = 0; q < n_rm; ++q) x.ids[q] = rm[q];
            K.defer[n_def + i] = X;
        }
        if (removed) team.add_to(&K.order[7], removed);
        if (unmulti) team.add_to(&K.order[8], unmulti);
        team.sync();
        if (tid == 0 && k_eff) {
            out_push(new_pre, ebase | (code_of(g.nodes[CX[0]].base) << 29));
            h.n_nodes += k_eff, h.live_nodes += k_eff, h.n_edges += k_eff, h.live_edges += k_eff - team.peek(&K.order[7]);
            h.n_multi -= team.peek(&K.order[8]);
        }
        team.sync();
    }

    // ---- the split by routes ------------------------------------------------------------------------------------------------------
    // What splitPath builds below an edge e with reads O is the prefix tree of the reads' ROUTES: the edges read r goes down from e until the sink is on the
    // main path (or the read ends); reads share a copy as long as their routes agree.  The reference walks that tree depth first, the ways out of a node last
    // to first, and hands out node and edge ids as it goes.  A by-passed stretch of the path makes the tree thousands of nodes deep and a few reads wide, so
    // here the reads are walked at the same time (one lane each: P1), the tree comes from comparing routes -- lcp(a, b) edges in common, and which of the
    // two the walk takes first where they part (P2) -- and the walk's order from that: the reads in depth-first order pi, read pi[t] owning the contexts
    // lp[t] .. len - 1 of its route (lp = what it shares with its predecessor), which are consecutive in the walk's order; so every context knows its edge id,
    // its node id, the chunks its id list needs and the copy in front of it without looking at another (P4: prefix sums over at most 32 reads), and all copies
    // are made at once (P5).  The reads leave the old edges read by read (the contexts of ONE read touch different edges and different nodes' lists: P6);
    // what is appended to lists that exist (the first copy to the node in front, the forks of the tree, the edges back into the path) thread 0 does in the
    // walk's order (P7); old nodes left without edges go (P8).  order[]: 11 new_pre, 12 e, 13 own_off, 14 n; out: 16 done (0: a route does not fit,
    // nothing was changed), 17 the contexts.
    static constexpr uint32_t kRouteReads = 32, kRouteAfter = 0;       // (from the first edge: the split's size is known before anything is changed)
    struct RouteWk { uint32_t *lc, *bf, *len, *lm, *pi, *lp, *cb, *nb, *kb, *d12, *d27, *ids, *st, *rt; uint32_t cap_rt; };
    DG_HD RouteWk route_wk(const CycWk &K, uint32_t n0) const
    {
        RouteWk R;
        uint32_t *p = K.route;
        R.lc = p, p += kRouteReads * kRouteReads;
        R.bf = p, p += kRouteReads * kRouteReads;
        R.len = p, p += kRouteReads; R.lm = p, p += kRouteReads; R.pi = p, p += kRouteReads; R.lp = p, p += kRouteReads;
        R.cb = p, p += kRouteReads; R.nb = p, p += kRouteReads; R.kb = p, p += kRouteReads; R.d12 = p, p += kRouteReads;
        R.d27 = p, p += kRouteReads; R.ids = p, p += kRouteReads; R.st = p, p += 8 * kRouteReads;
        R.rt = p;
        const uint32_t fixed = (uint32_t)(p - K.route);
        R.cap_rt = K.cap_route > fixed && n0 ? (K.cap_route - fixed) / n0 : 0;
        return R;
    }
    // the ids with a bit in `mask` (bit s: ids[s]) leave edge ei; returns the ids left
    DG_HD uint32_t drop_reads_masked(uint32_t ei, const uint32_t *ids, uint32_t mask, uint32_t n_rm)
    {
        Edge &e = g.edges[ei];
        const uint32_t n = e.count;
        if (n_rm >= n) { e.count = 0, e.head = e.tail = NIL; return 0; }
        uint32_t w = 0, rc = e.head, wc = e.head;
        for (uint32_t p = 0; p < n; ++p) {
            uint32_t v;
            if (p < kEdgeInl) v = e.ids[p];
            else { const uint32_t s = (p - kEdgeInl) % kChunkIds; if (s == 0 && p != kEdgeInl) rc = g.chunks[rc].next; v = g.chunks[rc].v[s]; }
            bool drop = false;
            for (uint32_t mk = mask, q = 0; mk; mk >>= 1, ++q) if ((mk & 1u) && ids[q] == v) { drop = true; break; }
            if (drop) continue;
            if (w < kEdgeInl) e.ids[w] = v;
            else { const uint32_t s = (w - kEdgeInl) % kChunkIds; if (s == 0 && w != kEdgeInl) wc = g.chunks[wc].next; g.chunks[wc].v[s] = v; }
            ++w;
        }
        e.count = w;
        e.tail = w > kEdgeInl ? wc : NIL;
        if (w <= kEdgeInl) e.head = NIL;
        return w;
    }
    DG_HD static uint32_t clamp3(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : v > hi ? hi : v; }
    DG_COLD void split_routes_run(const CycWk &K)
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), cr = team.crew_rank(), nc = team.crew_size();
        uint32_t *ord = K.order;
        const uint32_t new_pre = ord[11], e0 = ord[12], own_off = ord[13], n0 = ord[14];
        const RouteWk R = route_wk(K, n0);
        constexpr uint32_t W = kRouteReads;
        // P1: the routes
        uint32_t tk = team.clock();
        auto lap = [&](uint32_t i) { if (tid == 0) { const uint32_t t = team.clock(); h.st_rt[i] += t - tk; tk = t; } };
        // A route mostly runs along chains whose edges were made one after the other -- the by-passed stretch of the path (a read's own chain once), the
        // read's own side branches -- so the lanes of a read's group try the edges ce + 1, ce + 2, ... at once: edge ce + j + 1 is the read's next one behind
        // ce + j when it starts where that one ends and holds the read (a read is on one way out of a node).  Where the ids do not run on, the group's
        // first lane looks the next edge up.  Rounds until every route has ended.
        const uint32_t G = (h.dbg_flags & 8u) ? 16u : nc / n0 ? nc / n0 : 1u;      // lanes per read (a team of one: the routes step by step; tests: 16 lanes' work in turn)
        // per read: ce, len, done, the round's first failing link -- two copies: a round reads one and writes the other (no lane of a group may see the
        // group's next state before it has used the present one)
        uint32_t par = 0;
        for (uint32_t b = cr; b < n0; b += nc) {
            R.ids[b] = K.lists[own_off + b];
            if (R.cap_rt) R.rt[(size_t)b * R.cap_rt] = e0;
            uint32_t *st = R.st + 4 * b;
            st[0] = e0, st[1] = 1, st[2] = 0, st[3] = G;
            R.len[b] = 0, R.lm[b] = 0;
        }
        if (tid == 0) { ord[23] = 0; if (!R.cap_rt) ord[15] = 1; }
        team.sync();
        while (team.peek(&ord[23]) < n0 && !team.peek(&ord[15])) {
            uint32_t *cur = R.st + par * 4 * kRouteReads, *nxt = R.st + (par ^ 1u) * 4 * kRouteReads;
            for (uint32_t v = cr; v < n0 * G; v += nc) {
                const uint32_t b = v / G, jx = v % G;
                if (cur[4 * b + 2]) continue;
                const uint64_t e1 = (uint64_t)cur[4 * b] + jx, e2 = e1 + 1;
                bool ok = false;
                if (e2 < h.n_edges) {
                    const uint32_t mid = g.edges[e1].sink;
                    const Edge &x2 = g.edges[e2];
                    ok = mid != NIL && x2.src == mid && !g.nodes[mid].on_main && edge_has(x2, R.ids[b]);
                }
                if (!ok) team.min_to(&cur[4 * b + 3], jx);
            }
            team.sync();
            for (uint32_t v = cr; v < n0 * G; v += nc) {
                const uint32_t b = v / G, jx = v % G;
                if (cur[4 * b + 2]) { if (jx == 0) nxt[4 * b + 2] = 1; continue; }
                const uint32_t L = team.peek(&cur[4 * b + 3]), ce = cur[4 * b], len = cur[4 * b + 1];
                uint32_t *rt = R.rt + (size_t)b * R.cap_rt;
                if ((uint64_t)len + L + 1 > R.cap_rt) { if (jx == 0) { nxt[4 * b + 2] = 1; team.add_to(&ord[15], 1); team.add_to(&ord[23], 1); } continue; }
                if (jx < L) rt[len + jx] = ce + jx + 1;
                if (jx) continue;
                // the group's first lane: where the ids stop running on, the next edge by looking it up
                uint32_t c2 = ce + L, l2 = len + L, done = 0, lm = 0;
                if (L < G) {
                    const Node &x = g.nodes[g.edges[c2].sink];
                    if (x.on_main) done = 1, lm = 1;
                    else {
                        uint32_t next = NIL;
                        const uint32_t r = R.ids[b];
                        for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t o = out_ref(x, i) & kRefMask; if (edge_has(g.edges[o], r)) { next = o; break; } }
                        if (next == NIL) done = 1;               // the read ends at this node
                        else rt[l2++] = next, c2 = next;
                    }
                }
                nxt[4 * b] = c2, nxt[4 * b + 1] = l2, nxt[4 * b + 2] = done, nxt[4 * b + 3] = G;
                if (done) { R.len[b] = l2, R.lm[b] = lm; team.add_to(&ord[23], 1); }
            }
            team.sync();
            par ^= 1u;
        }
        team.sync();
        lap(0);
        if (team.peek(&ord[15])) { team.sync(); return; }
        // P2: what two routes share, and which of the two the walk takes first
        for (uint32_t p = cr; p < n0 * n0; p += nc) {
            const uint32_t a = p / n0, b = p % n0;
            if (a == b) { R.lc[a * W + a] = R.len[a]; R.bf[a * W + a] = 0; continue; }
            if (a > b) continue;
            const uint32_t *ra = R.rt + (size_t)a * R.cap_rt, *rb = R.rt + (size_t)b * R.cap_rt;
            const uint32_t la = R.len[a], lb = R.len[b], mn = la < lb ? la : lb;
            uint32_t d = 1;                                      // (both start with e)
            while (d < mn && ra[d] == rb[d]) ++d;
            bool a_first;
            if (d == mn) a_first = la <= lb;                     // one runs out where the other goes on (or both do): the shorter one first
            else {
                const Node &x = g.nodes[g.edges[ra[d - 1]].sink];
                uint32_t ia = 0, ib = 0;
                for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t o = out_ref(x, i) & kRefMask; if (o == ra[d]) ia = i; if (o == rb[d]) ib = i; }
                a_first = ia > ib;                               // the reference's stack: the last way out first
            }
            R.lc[a * W + b] = R.lc[b * W + a] = d;
            R.bf[a * W + b] = a_first ? 1u : 0u, R.bf[b * W + a] = a_first ? 0u : 1u;
        }
        team.sync();
        // P3: each read's place in the walk's order; up to which depth it travels with at least 11 / 26 others (id lists beyond the inline slots)
        for (uint32_t a = cr; a < n0; a += nc) {
            uint32_t rank = 0;
            for (uint32_t b = 0; b < n0; ++b) if (b != a) rank += R.bf[b * W + a];
            R.pi[rank] = a;
            uint32_t d12 = 0, d27 = 0;
            if (n0 > kEdgeInl) for (uint32_t s = 0; s < n0; ++s) {
                const uint32_t v = R.lc[a * W + s];
                uint32_t cnt = 0;
                for (uint32_t s2 = 0; s2 < n0; ++s2) cnt += R.lc[a * W + s2] >= v ? 1u : 0u;
                if (cnt > kEdgeInl && v > d12) d12 = v;
                if (cnt > kEdgeInl + kChunkIds && v > d27) d27 = v;
            }
            R.d12[a] = d12, R.d27[a] = d27;
        }
        team.sync();
        // P4: the contexts each read owns, and where their ids start
        if (tid == 0) {
            uint32_t C = 0, Nn = 0, Kc = 0;
            for (uint32_t t = 0; t < n0; ++t) {
                const uint32_t r = R.pi[t], len = R.len[r];
                const uint32_t lp = t ? R.lc[R.pi[t - 1] * W + r] : 0, cnt = len - lp;
                R.lp[t] = lp, R.cb[t] = C, R.nb[t] = Nn, R.kb[t] = Kc;
                C += cnt, Nn += cnt - (cnt && R.lm[r] ? 1u : 0u);
                const uint32_t a = clamp3(R.d27[r], lp, len), b = clamp3(R.d12[r], lp, len);
                Kc += 2 * (a - lp) + (b - a);
            }
            ord[17] = C, ord[18] = Nn, ord[19] = Kc;
            if ((uint64_t)h.n_nodes + Nn > h.cap_nodes || (uint64_t)h.n_edges + C > h.cap_edges || (uint64_t)h.n_chunks + Kc + 2 * n0 + 8 > h.cap_chunks) {
                // a split taken by routes from its first edge has changed nothing yet: the caller can make room and come again
                if (ord[24]) { h.need_nodes = Nn, h.need_edges = C, h.need_chunks = Kc + 2 * n0 + 8; ++h.st_regrow; fail_at(__LINE__, ERR_ROOM); }
                else fail_at(__LINE__, ERR_CAP);
                ord[15] = 1;
            }
        }
        team.sync();
        if (team.peek(&ord[15])) { team.sync(); return; }
        const uint32_t nbase = h.n_nodes, ebase = h.n_edges, kbase = h.n_chunks;
        lap(1);
        // P5: the copies
        for (uint32_t t = 0; t < n0; ++t) {
            const uint32_t r = R.pi[t], len = R.len[r], lp = R.lp[t];
            const uint32_t *rt = R.rt + (size_t)r * R.cap_rt, *lcr = R.lc + r * W;
            const uint32_t a27 = clamp3(R.d27[r], lp, len), a12 = clamp3(R.d12[r], lp, len);
            for (uint32_t d = lp + cr; d < len; d += nc) {
                const uint32_t q = R.cb[t] + (d - lp), ne = ebase + q;
                const bool leaf_main = d + 1 == len && R.lm[r];
                const uint32_t nid = nbase + R.nb[t] + (d - lp);
                uint32_t parent;
                if (d == 0) parent = new_pre;
                else if (d > lp) parent = nid - 1;
                else {                                           // the copy in front belongs to the first read of the walk that travels with this one that far
                    uint32_t t2 = 0;
                    while (R.lc[R.pi[t2] * W + r] < d) ++t2;
                    parent = nbase + R.nb[t2] + (d - 1 - R.lp[t2]);
                }
                const uint32_t X = g.edges[rt[d]].sink;
                const uint32_t dm = d < a27 ? d : a27, dn = d < a12 ? d : a12;
                const uint32_t ck = kbase + R.kb[t] + 2 * (dm - lp) + (dn - dm);
                Edge &x = g.edges[ne];
                x.src = parent, x.sink = leaf_main ? X : nid, x.head = x.tail = NIL;
                uint32_t w = 0;
                for (uint32_t s = 0; s < n0; ++s) {
                    if (lcr[s] <= d) continue;
                    const uint32_t id = R.ids[s];
                    if (w < kEdgeInl) x.ids[w] = id;
                    else g.chunks[ck + (w - kEdgeInl) / kChunkIds].v[(w - kEdgeInl) % kChunkIds] = id;
                    ++w;
                }
                x.count = w;
                if (w > kEdgeInl) {
                    const uint32_t nk = (w - kEdgeInl + kChunkIds - 1) / kChunkIds;
                    x.head = ck, x.tail = ck + nk - 1;
                    for (uint32_t i = 0; i < nk; ++i) g.chunks[ck + i].next = i + 1 < nk ? ck + i + 1 : NIL;
                }
                if (!leaf_main) {
                    Node &nd = g.nodes[nid];
                    nd.out_ext = nd.in_ext = NIL, nd.base = g.nodes[X].base, nd.on_main = 0;
                    g.mark[nid] = 0, g.pidx[nid] = NIL;
                    nd.n_in = 1, nd.in[0] = ne;
                    if (d + 1 < len) nd.n_out = 1, nd.out[0] = (ne + 1) | (code_of(g.nodes[g.edges[rt[d + 1]].sink].base) << 29);
                    else nd.n_out = 0;
                }
            }
        }
        team.sync();
        lap(2);
        // P6: the reads leave the old edges, one read of the walk after the other
        uint32_t removed = 0, unmulti = 0;
        for (uint32_t t = 0; t < n0; ++t) {
            const uint32_t r = R.pi[t], len = R.len[r], lp = R.lp[t];
            uint32_t *rt = R.rt + (size_t)r * R.cap_rt;
            const uint32_t *lcr = R.lc + r * W;
            for (uint32_t d = lp + cr; d < len; d += nc) {
                uint32_t mask = 0, n_rm = 0;
                for (uint32_t s = 0; s < n0; ++s) if (lcr[s] > d) mask |= 1u << s, ++n_rm;
                const uint32_t oe = rt[d], X = g.edges[oe].sink;
                if (drop_reads_masked(oe, R.ids, mask, n_rm) == 0) { const int32_t dd = remove_edge_quiet(oe, false, false); ++removed; if (dd < 0) ++unmulti; }
                rt[d] = d + 1 == len && R.lm[r] ? NIL : X;       // (from here on: the old node the context has copied)
            }
            if (lp < len) team.sync();
        }
        if (removed) team.add_to(&ord[20], removed);
        if (unmulti) team.add_to(&ord[21], unmulti);
        team.sync();
        // P7: what is appended to lists that were there before
        if (tid == 0) {
            // (the copies' ids are taken first: the pushes below may hand out chunks of their own)
            h.n_nodes += ord[18], h.live_nodes += ord[18], h.n_edges += ord[17], h.live_edges += ord[17] - team.peek(&ord[20]), h.n_chunks += ord[19];
            h.n_multi -= team.peek(&ord[21]);
            out_push(new_pre, ebase | (code_of(g.nodes[g.edges[ebase].sink].base) << 29));
            for (uint32_t t = 1; t < n0; ++t) {
                const uint32_t r = R.pi[t], d = R.lp[t];
                if (d >= R.len[r]) continue;
                uint32_t t2 = 0;
                while (R.lc[R.pi[t2] * W + r] < d) ++t2;
                const uint32_t ne = ebase + R.cb[t];
                out_push(nbase + R.nb[t2] + (d - 1 - R.lp[t2]), ne | (code_of(g.nodes[g.edges[ne].sink].base) << 29));
            }
            for (uint32_t t = 0; t < n0; ++t) {
                const uint32_t r = R.pi[t];
                if (R.lp[t] >= R.len[r] || !R.lm[r]) continue;
                const uint32_t ne = ebase + R.cb[t] + (R.len[r] - R.lp[t]) - 1;
                in_push(g.edges[ne].sink, ne);
            }
            h.st_ctx += ord[17], ++h.st_routes, h.st_route_ctx += ord[17];
        }
        team.sync();
        // P8: old nodes nothing leads to or from any more
        uint32_t gone = 0;
        for (uint32_t t = 0; t < n0; ++t) {
            const uint32_t r = R.pi[t], len = R.len[r], lp = R.lp[t];
            const uint32_t *rt = R.rt + (size_t)r * R.cap_rt;
            for (uint32_t d = lp + cr; d < len; d += nc) {
                const uint32_t X = rt[d];
                if (X == NIL) continue;
                Node &x = g.nodes[X];
                if (x.n_in || x.n_out || x.on_main || !x.base) continue;
                uint32_t *wd = reinterpret_cast<uint32_t *>(&x.n_out);          // n_out n_in base on_main: one word
                const uint32_t was = (uint32_t)x.base << 16;
                if (team.cas(wd, was, 0u)) ++gone;
            }
        }
        if (gone) team.add_to(&ord[22], gone);
        team.sync();
        if (tid == 0) { h.live_nodes -= team.peek(&ord[22]); ord[16] = 1; }
        lap(3);
        team.sync();
    }
    DG_COLD void walk_and_prune(const CycWk &K, uint32_t e0, bool marked_only)
    {
        const Hdr &h = *g.h;
        uint32_t n = 0, streak = 0;
        K.estack[n++] = e0;
        // (a side node with more than one edge in is what a split needs: when the last of them is gone -- the splits so far have taken them apart -- the
        // rest of the reference's walk finds nothing to do)
        while (n && !failed() && h.n_multi) {
            const uint32_t curr = K.estack[--n];
            ++g.h->st_pops;
            const uint32_t sink = g.edges[curr].sink, source = g.edges[curr].src;
            if (sink == NIL) continue;                           // (an edge a split before this one took away)
            if (g.nodes[sink].on_main) continue;
            // (nothing below an unmarked node can be split.  A node the walk has gone down from before -- it comes to a node once per edge in -- carries
            // ~epoch: the edge it comes by now is split like any other, but below the node every edge has been looked at, and what the walk found there
            // was either split away or led to a node with one way in, which it still is: the reference's second pass over the same branch changes nothing)
            const uint32_t mk = g.mark[sink], seen = ~h.epoch;
            if (marked_only && mk != h.epoch && mk != seen) continue;
            if (g.nodes[sink].n_in > 1) split_path(K, source, curr);
            if (mk == seen) continue;
            g.mark[sink] = seen;
            const Node &s = g.nodes[sink];
            if (s.n_out == 1 && s.n_in == 1 && probing()) {
                if (++streak >= 3) {                          // down a branch without forks or ways in: nothing to split on a stretch of it
                    const uint32_t o0 = s.out[0] & kRefMask;
                    const uint32_t L = probe(0, sink, o0, 4 * team.crew_size() + 60, 0, 0);
                    if (L >= 2) { K.estack[n++] = o0 + L - 1; continue; }
                    streak = 0;
                }
            } else streak = 0;
            if (n + s.n_out > K.cap_estack) { fail_at(__LINE__, ERR_SCRATCH); return; }
            for (uint32_t i = 0; i < s.n_out; ++i) K.estack[n++] = out_ref(s, i) & kRefMask;
        }
    }
    DG_HD void run_node(const CycWk &K, uint32_t n, bool marked_only)
    {
        const Node &x = g.nodes[n];
        const uint32_t no = x.n_out;
        if (no > K.cap_copy) { fail_at(__LINE__, ERR_SCRATCH); return; }
        for (uint32_t i = 0; i < no; ++i) K.copy[i] = out_ref(x, i) & kRefMask;      // a copy: the walk edits the node's list
        for (uint32_t i = 0; i < no && !failed() && g.h->n_multi; ++i) walk_and_prune(K, K.copy[i], marked_only);
    }

    // removeCycles.  The reference walks every side branch of the two stretches just re-walked and splits at every side edge whose sink
    // has another way in.  The nodes with that property are known (multi_list: every node that became one was noted); every edge a split
    // can happen at lies on a way from the path to one of them.  So: mark them and their ancestors (edges in, backwards), find the path
    // nodes the marked branches hang off (lanes over the path's node array), and run the reference's walk from those nodes only, in
    // its order, descending only into marked nodes.  When the list does not account for every such node the reference's full walk runs.
    DG_COLD void remove_cycles()
    {
        Hdr &h = *g.h;
        const uint32_t tid = team.tid(), nt = team.size();
        if (h.n_multi == 0) { if (tid == 0) h.multi_n = 0, h.unreach_n = NIL; team.sync(); return; }
        const CycWk K = cyc_wk();
        uint32_t mode = 0, n_roots = 0;                        // 0 nothing to do, 1 from the list, 2 the full walk
        const uint32_t c0 = team.clock();
        if (tid == 0) {
            ++h.st_cycles_run;
            ++h.epoch;
            h.unreach_n = NIL;
            mode = 1;
            if (h.epoch == 0 || h.multi_n > h.cap_multi || (h.dbg_flags & 2u)) { mode = 2; if (h.epoch == 0) h.epoch = 1; }
            if (mode == 1) {
                uint32_t k = 0;
                for (uint32_t i = 0; i < h.multi_n; ++i) { const uint32_t n = g.multi_list[i]; if (multi_in_side(n) && g.mark[n] != h.epoch) { g.mark[n] = h.epoch; g.multi_list[k++] = n; } }
                h.multi_n = k;
                if (k != h.n_multi) mode = 2;
            }
            if (mode == 1) {
                uint32_t n_todo = 0;
                for (uint32_t i = 0; i < h.multi_n && n_todo < K.cap_todo; ++i) K.todo[n_todo++] = g.multi_list[i];
                if (h.multi_n > K.cap_todo) mode = 2;
                uint32_t streak = 0;
                while (n_todo && mode == 1) {
                    uint32_t n = K.todo[--n_todo];
                    ++h.st_anc;
                    if (g.nodes[n].n_in == 1 && probing()) {
                        if (++streak >= 3) {                      // up a branch without forks: a stretch of it at once when its ids run on
                            const uint32_t L = probe(3, n, g.nodes[n].in[0], 4 * team.crew_size() + 60, 0, 0);
                            if (L >= 2) n = n - (L - 1);          // (its nodes are marked; go on from the last of them)
                            else streak = 0;
                        }
                    } else streak = 0;
                    const Node &x = g.nodes[n];
                    for (uint32_t i = 0; i < x.n_in; ++i) {
                        const uint32_t s = g.edges[in_ref(x, i)].src;
                        if (g.mark[s] == h.epoch) continue;
                        g.mark[s] = h.epoch;
                        if (g.nodes[s].on_main) { if (n_roots < K.cap_roots) K.roots[n_roots++] = s; else mode = 2; }
                        else { if (n_todo < K.cap_todo) K.todo[n_todo++] = s; else mode = 2; }
                    }
                }
                if (mode == 1 && n_roots == 0) { mode = 0; h.unreach_n = h.n_multi, h.unreach_list_n = h.multi_n; }        // not reachable from the path: the full walk would find nothing either (and no excursion of the next read can meet one of them: update())
            }
            if (mode == 2) ++h.st_full_walk;
            h.st_cyc[0] += team.clock() - c0;
            (void)team.bcast(0);
        } else if (team.helper()) helpers_loop();
        team.sync();
        mode = team.bcast(mode), n_roots = team.bcast(n_roots);
        const uint32_t m = h.m, off = h.path_off;
        if (mode == 1) {
            // where the roots lie on the path: node indices [0, left_end) and [right_off, m].  Each node remembers where it was written last;
            // only when that does not hold for one of them is the path looked through.
            uint32_t known = 0;
            if (tid == 0) {
                known = 1;
                const uint32_t left_end0 = h.left_off < m ? h.left_off + 1 : m;
                uint32_t nh = 0;
                for (uint32_t r = 0; r < n_roots && known; ++r) {
                    const uint32_t i = path_index_of(K.roots[r]);
                    if (i == NIL) { known = 0; break; }
                    if (!(i < left_end0 || i >= h.right_off)) continue;
                    uint32_t q = nh++;                              // insertion by index (a handful of roots)
                    for (; q > 0 && K.hits[2 * (q - 1)] > i; --q) K.hits[2 * q] = K.hits[2 * (q - 1)], K.hits[2 * q + 1] = K.hits[2 * (q - 1) + 1];
                    K.hits[2 * q] = i, K.hits[2 * q + 1] = K.roots[r];
                }
                if (known) K.order[9] = nh;
            }
            known = team.bcast(known);
            uint64_t bloom = 0;
            for (uint32_t i = 0; i < n_roots; ++i) bloom |= 1ull << (K.roots[i] & 63u);
            const uint32_t left_end = h.left_off < m ? h.left_off + 1 : m;
            uint32_t lo1 = 0, hi1 = left_end, lo2 = h.right_off, hi2 = m + 1;
            if (h.right_off < left_end) lo1 = 0, hi1 = m + 1, lo2 = hi2 = 0;
            uint32_t n_hits = 0;
            if (known) { n_hits = K.order[9]; lo1 = hi1 = lo2 = hi2 = 0; }
            for (int part = 0; part < 2; ++part) {
                const uint32_t lo = part ? lo2 : lo1, hi = part ? hi2 : hi1;
                for (uint32_t base = lo; base < hi; base += nt) {
                    const uint32_t i = base + tid;
                    uint32_t hit = 0, n = NIL;
                    if (i < hi) {
                        n = g.pn[off + i];
                        if ((bloom >> (n & 63u)) & 1u) for (uint32_t r = 0; r < n_roots; ++r) if (K.roots[r] == n) { hit = 1; break; }
                    }
                    uint32_t tot;
                    const uint32_t p = team.scan(hit, tot);
                    if (hit && n_hits + p < K.cap_hits / 2) K.hits[2 * (n_hits + p)] = i, K.hits[2 * (n_hits + p) + 1] = n;
                    n_hits += tot;
                }
            }
            team.sync();
            if (n_hits > K.cap_hits / 2) { if (tid == 0) fail_at(__LINE__, ERR_SCRATCH); team.sync(); return; }
            const uint32_t c1 = team.clock();
            if (tid == 0) {
                h.st_cyc[1] += c1 - c0;
                // first loop of the reference: nodes right_off .. m in path order; second: nodes min(left_off, m - 1) .. 0, backwards
                for (uint32_t i = 0; i < n_hits && !failed() && h.n_multi; ++i) if (K.hits[2 * i] >= h.right_off) run_node(K, K.hits[2 * i + 1], true);
                const uint32_t l0 = h.left_off < m ? h.left_off : m - 1;
                for (uint32_t i = n_hits; i-- > 0 && !failed() && h.n_multi;) if (K.hits[2 * i] <= l0 && K.hits[2 * i] < m) run_node(K, K.hits[2 * i + 1], true);
                h.st_cyc[5] += team.clock() - c1;
                (void)team.bcast(0);
            } else if (team.helper()) helpers_loop();
        } else if (mode == 2) {
            if (tid == 0) {
                for (uint32_t i = h.right_off; i <= m && !failed() && h.n_multi; ++i) run_node(K, g.pn[off + i], false);
                for (uint32_t i = (h.left_off < m ? h.left_off : m - 1) + 1; i-- > 0 && !failed() && h.n_multi;) run_node(K, g.pn[off + i], false);
                // (what is left are nodes the walks cannot reach; the list starts over with them)
                if (h.multi_n > h.cap_multi) h.multi_n = 0;
                (void)team.bcast(0);
            } else if (team.helper()) helpers_loop();
        }
        team.sync();
    }
};

}  // namespace dg
}  // namespace nsgpu
