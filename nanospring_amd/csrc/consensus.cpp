// consensus.cpp -- see consensus.hpp.
#include "consensus.hpp"
#include <algorithm>
#include <cassert>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <iterator>
#include <new>
#include <set>
#include <mutex>
#include <atomic>
#include <sys/mman.h>

#ifdef NSGPU_PROF
#include <chrono>
double g_prof[8];
#define PROF_T(x) const double x = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count()
#define PROF_ADD(i, v) g_prof[i] += (v)
#else
#define PROF_T(x)
#define PROF_ADD(i, v)
#endif

namespace nsgpu {
namespace cons {

using mm2::EditOp;

// ---------------------------------------------------------------------------
// pool
// ---------------------------------------------------------------------------
namespace {
struct SlabCache {
    std::vector<unsigned char *> free;     // region-backed slabs are not handed back at thread exit (process-lifetime regions)
};
thread_local SlabCache t_slabs;
}  // namespace


// Slabs are carved out of 32 MiB regions aligned to 2 MiB and advised as huge pages: a thread's graphs spread over
// hundreds of MB that are walked edge by edge, and with 4 KiB pages nearly every edge is also a TLB miss.  Regions are
// never returned to the system (slabs circulate through the per-thread caches for the life of the process).
namespace {
struct Region { unsigned char *p = nullptr; size_t used = 0; };
thread_local Region t_region;
constexpr size_t kRegionBytes = 32u << 20;
}  // namespace

// Graphs are built on one thread and often freed on another (the edit emission runs as a background task of whichever worker
// is idle): a thread's cache hands its surplus to a shared pool and refills from it before new memory is mapped, so slabs
// circulate instead of piling up on the freeing side.
namespace {
std::mutex g_slab_pool_m;
std::vector<unsigned char *> g_slab_pool;
constexpr size_t kLocalSlabs = 64, kSlabBatch = 32;
}  // namespace

std::atomic<int64_t> g_slabs_in_use{0}, g_slabs_peak{0}, g_slabs_mapped{0};     // debug print: slabs handed out now / at most / ever carved
unsigned char *slab_acquire(size_t bytes)
{
    if (bytes == kSlabBytes) { const int64_t u = ++g_slabs_in_use; int64_t pk = g_slabs_peak.load(); while (u > pk && !g_slabs_peak.compare_exchange_weak(pk, u)) {} }
    if (t_slabs.free.empty() && bytes == kSlabBytes) {
        std::lock_guard<std::mutex> lk(g_slab_pool_m);
        for (size_t i = 0; i < kSlabBatch && !g_slab_pool.empty(); ++i) { t_slabs.free.push_back(g_slab_pool.back()); g_slab_pool.pop_back(); }
    }
    if (!t_slabs.free.empty()) { unsigned char *p = t_slabs.free.back(); t_slabs.free.pop_back(); return p; }
    if (bytes <= kRegionBytes) {
        if (!t_region.p || t_region.used + bytes > kRegionBytes) {
            void *r = mmap(nullptr, kRegionBytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (r == MAP_FAILED) throw std::bad_alloc();
            unsigned char *a = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(r) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
            (void)madvise(a, kRegionBytes, MADV_HUGEPAGE);
            t_region.p = a, t_region.used = 0;
        }
        unsigned char *p = t_region.p + t_region.used;
        t_region.used += bytes;
        ++g_slabs_mapped;
        return p;
    }
    void *p = nullptr;
    if (posix_memalign(&p, 64, bytes) != 0) throw std::bad_alloc();
    return static_cast<unsigned char *>(p);
}

void slab_release(unsigned char *p, size_t bytes)
{
    if (bytes == kSlabBytes) --g_slabs_in_use;
    // region-backed slabs never reach free(): they circulate through the caches
    t_slabs.free.push_back(p);
    if (t_slabs.free.size() > kLocalSlabs) {
        std::lock_guard<std::mutex> lk(g_slab_pool_m);
        for (size_t i = 0; i < kSlabBatch; ++i) { g_slab_pool.push_back(t_slabs.free.back()); t_slabs.free.pop_back(); }
    }
}

template <class T> Pool<T>::~Pool()
{
    // the objects own nothing outside the graph's Arena: only the slabs go back
    for (unsigned char *slab : slabs_) slab_release(slab, kSlabBytes);
}
template <class T> template <class... A> T *Pool<T>::make(A &&...a)
{
    T *p;
    if (!free_.empty()) { p = free_.back(); free_.pop_back(); }
    else {
        if (slabs_.empty() || used_in_last_ == kPerSlab) { slabs_.push_back(slab_acquire(kSlabBytes)); used_in_last_ = 0; }
        p = reinterpret_cast<T *>(slabs_.back()) + used_in_last_++;
    }
    ++live;
    return new (p) T(std::forward<A>(a)...);
}
template <class T> void Pool<T>::free(T *p) { free_.push_back(p); --live; }
template class Pool<Node>;
template class Pool<Edge>;

Arena::~Arena()
{
    for (unsigned char *slab : slabs_) slab_release(slab, kSlabBytes);
    for (void *p : big_) ::free(p);
}
void *Arena::alloc(unsigned cls)
{
    const size_t bytes = (size_t)32 << cls;
    if (cls >= kClasses) {
        void *p = malloc(bytes);
        if (!p) throw std::bad_alloc();
        big_.push_back(p);
        return p;
    }
    if (void *p = free_[cls]) { memcpy(&free_[cls], p, sizeof(void *)); return p; }
    if (used_ + bytes > kSlabBytes) {
        // the unused rest of the slab is wasted (at most one block of the largest class per slab)
        slabs_.push_back(slab_acquire(kSlabBytes));
        used_ = 0;
    }
    void *p = slabs_.back() + used_;
    used_ += bytes;
    return p;
}
void Arena::release(void *p, unsigned cls)
{
    if (cls >= kClasses) {
        auto it = std::find(big_.begin(), big_.end(), p);
        if (it != big_.end()) { *it = big_.back(); big_.pop_back(); }
        ::free(p);
        return;
    }
    memcpy(p, &free_[cls], sizeof(void *));
    free_[cls] = p;
}

// ---------------------------------------------------------------------------
// nodes and edges
// ---------------------------------------------------------------------------
void Edge::add_read(Arena &a, read_t r)
{
    ++count;
    const uint32_t n = reads.n;
    if (sorted_n == n && (n == 0 || reads.data()[n - 1] < r)) sorted_n = n + 1;      // still ascending
    reads.push_back(a, r);
}

void Edge::sort_reads()
{
    const uint32_t n = reads.n;
    if (sorted_n == n) return;
    read_t *b = reads.data();
    std::sort(b + sorted_n, b + n);
    if (sorted_n) std::inplace_merge(b, b + sorted_n, b + n);
    sorted_n = n;
}

Edge *Node::edge_to(Node *n) const { for (Edge *e : out) if (e->sink == n) return e; return nullptr; }
Edge *Node::edge_to_side(char b) const { for (const OutRef &o : out) if (o.sink_base_is(b) && !o->sink->on_main) return o; return nullptr; }
Edge *Node::best_out() const
{
    if (out.size() == 1) return out[0]->count ? out[0].get() : nullptr;
    Edge *best = nullptr; read_t c = 0;
    for (Edge *e : out) if (e->count > c) c = e->count, best = e;
    return best;
}
Edge *Node::best_in() const { Edge *best = nullptr; read_t c = 0; for (Edge *e : in) if (e->count > c) c = e->count, best = e; return best; }
Edge *Node::edge_in_read(read_t r) const
{
    for (Edge *e : out) {
        e->sort_reads();
        const read_t *b = e->reads.data();
        size_t n = e->reads.size();
        if (n == 0 || r < b[0] || r > b[n - 1]) continue;
        while (n > 1) { const size_t h = n >> 1; b = b[h] <= r ? b + h : b; n -= h; }   // branch-free lower bound on a short sorted list
        if (*b == r) return e;
    }
    return nullptr;
}

Node *ContigGraph::create_node(char b) { ++n_nodes_; return nodes_.make(b); }
Edge *ContigGraph::create_edge(Node *s, Node *t, read_t r)
{
    Edge *e = edges_.make();
    e->source = s, e->sink = t, e->count = 1, e->reads.push_back(arena_, r), e->sorted_n = 1;
    const bool was_multi = multi_in_side(t);
    n_multi_in_side_ -= was_multi;
    s->out.push_back(arena_, OutRef(e, t->base)), t->in.push_back(arena_, e);
    n_multi_in_side_ += multi_in_side(t);
    note_multi(t, was_multi);
    ++n_edges_;
    return e;
}
Edge *ContigGraph::create_edge(Node *s, Node *t, const std::vector<read_t> &rs)
{
    Edge *e = edges_.make();
    e->source = s, e->sink = t, e->reads.assign(arena_, rs.data(), rs.size()), e->count = (read_t)rs.size(), e->sorted_n = (uint32_t)rs.size();    // rs is ascending
    const bool was_multi = multi_in_side(t);
    n_multi_in_side_ -= was_multi;
    s->out.push_back(arena_, OutRef(e, t->base)), t->in.push_back(arena_, e);
    n_multi_in_side_ += multi_in_side(t);
    note_multi(t, was_multi);
    ++n_edges_;
    return e;
}
// removeEdge (:861-876): the source side drops the FIRST out-edge that leads to the same sink
void ContigGraph::remove_edge(Edge *e, bool keep_in_source, bool keep_in_sink)
{
    if (!keep_in_source) {
        auto &v = e->source->out;
        v.erase_at((size_t)(std::find_if(v.begin(), v.end(), [&](const Edge *p) { return p->sink == e->sink; }) - v.begin()));
    }
    if (!keep_in_sink) {
        auto &v = e->sink->in;
        n_multi_in_side_ -= multi_in_side(e->sink);
        v.erase_at((size_t)(std::find(v.begin(), v.end(), e) - v.begin()));
        n_multi_in_side_ += multi_in_side(e->sink);
    }
    e->reads.release(arena_);
    edges_.free(e);
    --n_edges_;
}
void ContigGraph::remove_node(Node *n)
{
    for (size_t i = 0; i < multi_in_list_.size();)          // the node's memory is recycled: no stale pointer may stay behind
        if (multi_in_list_[i] == n) multi_in_list_[i] = multi_in_list_.back(), multi_in_list_.pop_back(); else ++i;
    n_multi_in_side_ -= multi_in_side(n);
    n->on_main = true;      // keeps the counter untouched while the node's edges go away
    for (Edge *e : std::vector<Edge *>(n->in.begin(), n->in.end())) remove_edge(e, false, true);
    for (Edge *e : std::vector<Edge *>(n->out.begin(), n->out.end())) remove_edge(e, true, false);
    n->in.release(arena_), n->out.release(arena_);
    nodes_.free(n);
    --n_nodes_;
}
void ContigGraph::remove_reads_from_edge(Edge *e, const std::vector<read_t> &rs)
{
    // in place: the output never overtakes the input of a set difference
    e->sort_reads();
    read_t *w = std::set_difference(e->reads.begin(), e->reads.end(), rs.begin(), rs.end(), e->reads.begin());
    e->reads.n = (uint32_t)(w - e->reads.begin());
    e->sorted_n = e->reads.n;
    e->count = (read_t)e->reads.size();
    if (e->count == 0) remove_edge(e);
}

// ---------------------------------------------------------------------------
// graph construction
// ---------------------------------------------------------------------------
void ContigGraph::initialize(const std::string &seed, read_t id, long pos)
{
    Node *cur = create_node(seed[0]);
    reads.insert(std::make_pair(id, GraphRead{pos, cur, seed.length(), false}));
    right_unchanged_ = left_unchanged_ = cur;
    right_off_ = left_off_ = 0;
    path_changed_from = 0;
    main_path.push_back(cur->base);
    set_on_main(cur, true);
    cur->cum_weight = 0;
    for (size_t i = 1; i < seed.length(); ++i) {
        Node *nx = create_node(seed[i]);
        create_edge(cur, nx, id);
        cur = nx;
    }
    start_pos = pos;
    end_pos = pos + 1;     // only one base is on the main path until calculate_main_path_greedy runs
}

static inline uint64_t emit_now();
std::atomic<uint64_t> g_mp_cnt[3];    // diagnostic: main-path edges copied out by the tail re-use, calls that cut the path, path lengths at those calls
void ContigGraph::update_graph(const std::string &s, const std::vector<EditOp> &script, ssize_t begin_offset, ssize_t end_offset, read_t id,
                               long pos, bool rc)
{
    const size_t n_path_edges = main_edges.size();
    std::vector<uint64_t> dbg_before;
    static const bool dbg = getenv("NSGPU_SPLICE_CHECK") != nullptr;      // debugging aid for the tail re-use shortcut
    auto out_sig = [](const Node *n) { uint64_t h = n->out.size(); for (Edge *e : n->out) h = h * 1000003u + e->count; return h; };
    if (dbg) for (size_t i = 0; i < n_path_edges; ++i) dbg_before.push_back(out_sig(main_edges[i]->sink));
    size_t ei = 0;                                   // edgeInPath
    Node *node_in_path = main_edges[0]->source;
    Node *cur = nullptr, *initial = nullptr;
    // main-path indices (node j = main_edges[j-1]->sink) where this read leaves the path over a side edge: the only
    // nodes whose greedy choice the update can change (see calculate_main_path_greedy)
    diverged_.clear();
    touch_lo_ = (size_t)-1;                          // first main-path node whose out-edges this read changes (its first SAME)
    ssize_t cur_main = -1;                           // main-path index of `cur` while it is a main-path node reached by SAME

    if (begin_offset >= 0 || end_offset >= 0) {
        right_off_ = (size_t)std::max((ssize_t)left_off_, std::min((ssize_t)right_off_, begin_offset));
        right_unchanged_ = right_off_ > 0 ? main_edges[right_off_ - 1]->sink : main_edges[0]->source;
    } else {
        left_off_ = std::min(right_off_, std::max(left_off_, main_path.size() - 1 + (size_t)end_offset));
        left_unchanged_ = left_off_ > 0 ? main_edges[left_off_ - 1]->sink : main_edges[0]->source;
    }
    auto advance = [&]() {
        if (ei == n_path_edges) return;
        node_in_path = main_edges[ei]->sink;
        ++ei;
    };
    if (begin_offset >= 1) {
        ei += (size_t)begin_offset - 1;
        node_in_path = main_edges[ei]->sink;
        ++ei;
    } else if (begin_offset <= -1) {
        const size_t n_ins = (size_t)(-begin_offset);
        size_t i = 0;
        cur = create_node(s[i++]);
        initial = cur;
        for (; i < n_ins; ++i) {
            Node *nx = create_node(s[i]);
            create_edge(cur, nx, id);
            cur = nx;
        }
    }
    auto insert_node = [&](char base) {
        if (cur_main >= 0) diverged_.push_back((size_t)cur_main), cur_main = -1;
        if (!cur) {
            cur = create_node(base);
            initial = cur;
        } else {
            Edge *e = cur->edge_to_side(base);
            if (e) e->add_read(arena_, id);
            else e = create_edge(cur, create_node(base), id);
            cur = e->sink;
        }
    };
    // Between two edits the walk streams along the main path (hardware prefetch copes); every edit lands on a main-path
    // node somewhere else: its line, its out-edges and their sinks are dependent misses.  The script says where the next
    // edits fall, so their lines are requested a few edits ahead.
    std::vector<uint32_t> &op_at = op_at_;           // main-path edge index at which op k starts (the reference's edgeInPath)
    op_at.resize(script.size() + 1);
    {
        size_t e2 = begin_offset >= 1 ? (size_t)begin_offset : 0;
        for (size_t k = 0; k < script.size(); ++k) {
            op_at[k] = (uint32_t)(e2 < n_path_edges ? e2 : n_path_edges);
            if (script[k].type == 0) e2 += script[k].num; else if (script[k].type == 2) ++e2;
        }
        op_at[script.size()] = (uint32_t)(e2 < n_path_edges ? e2 : n_path_edges);
    }
    size_t op_k = 0;
    for (const EditOp &op : script) {
        {
            const size_t k9 = op_k + 9, k6 = op_k + 6, k3 = op_k + 3;
            if (k9 < script.size() && op_at[k9] >= 1) __builtin_prefetch(main_edges[op_at[k9] - 1], 0, 1);
            // the edit leaves the path at the edge's source (its out list is searched and extended) and comes back at or behind its sink
            if (k6 < script.size() && op_at[k6] >= 1) { const Edge *pe = main_edges[op_at[k6] - 1]; __builtin_prefetch(pe->source, 1, 1); __builtin_prefetch(pe->sink, 1, 1); }
            if (k3 < script.size() && op_at[k3] >= 1) { const Node *n3 = main_edges[op_at[k3] - 1]->source; for (const Edge *pe : n3->out) __builtin_prefetch(pe, 1, 1); }
            ++op_k;
        }
        if (op.type == 0) {                          // SAME
            if (!cur) initial = cur = node_in_path;
            else {
                if (cur_main >= 0 && (size_t)cur_main + 1 != ei) diverged_.push_back((size_t)cur_main);   // a jump over deleted main-path nodes
                Edge *e = cur->edge_to(node_in_path);
                if (e) e->add_read(arena_, id);
                else e = create_edge(cur, node_in_path, id);
                cur = node_in_path;
            }
            cur_main = (ssize_t)ei;
            if (touch_lo_ == (size_t)-1) touch_lo_ = ei;
            advance();
            // The rest of the run follows the main-path edges main_edges[ei-1 .. ei+num-3] one after the other (the loop below,
            // which re-derives that per base, stays for the path's end, where the reference's walk stops advancing): add the read
            // to each of them and set the walk's state once.
            static const bool no_run_loop = getenv("NSGPU_NO_RUN_FASTPATH") != nullptr;      // debugging aid: per-base loop only
            if (!no_run_loop && op.num > 1 && ei >= 1 && ei + op.num - 2 <= n_path_edges && main_edges[ei - 1]->source == cur && main_edges[ei - 1]->sink == node_in_path) {
                const size_t last_ei = ei + op.num - 2;                 // ei at the start of the run's last base
                Edge *const *pe = main_edges.begin();
                constexpr size_t pf_edge = 40, pf_tail = 20;           // prefetch distances (edges ahead), measured at cfg2
                for (size_t x = ei - 1; x < last_ei; ++x) {
                    if (x + pf_edge < n_path_edges) __builtin_prefetch(pe[x + pf_edge], 1, 1);
                    // second miss of an edge with more than eight reads: the tail of its read list, one line behind the edge's own
                    if (x + pf_tail < n_path_edges) { const Edge *e6 = pe[x + pf_tail]; if (e6->reads.cap != kEdgeInlineReads) __builtin_prefetch(e6->reads.heap + e6->reads.n, 1, 1); }
                    pe[x]->add_read(arena_, id);
                }
                cur = pe[last_ei - 1]->sink;
                cur_main = (ssize_t)last_ei;
                if (last_ei == n_path_edges) ei = n_path_edges, node_in_path = cur;
                else node_in_path = pe[last_ei]->sink, ei = last_ei + 1;
                continue;
            }
            for (size_t i = 1; i < op.num; ++i) {
                // the walk touches one edge (one cache line) per base, in main-path order: fetch ahead
                if (ei + 40 < n_path_edges) __builtin_prefetch(main_edges[ei + 40], 1, 1);
                // cur and node_in_path are consecutive main-path nodes here, and edges are unique per (source, sink)
                // (update_graph looks before it creates; split_path only adds edges out of fresh nodes), so the edge
                // getEdgeTo() would find is the main-path edge itself
                Edge *e = ei >= 1 && ei <= n_path_edges && main_edges[ei - 1]->source == cur && main_edges[ei - 1]->sink == node_in_path
                              ? main_edges[ei - 1] : cur->edge_to(node_in_path);
                e->add_read(arena_, id);
                cur = node_in_path;
                cur_main = (ssize_t)ei;
                advance();
            }
        } else if (op.type == 2) { advance(); }         // DELETE
        else if (op.type == 1) { insert_node((char)op.base); }
    }
    if (end_offset > 0)
        for (size_t i = s.size() - (size_t)end_offset; i < s.size(); ++i) insert_node(s[i]);
    reads.insert(std::make_pair(id, GraphRead{pos, initial, s.length(), rc}));
    touch_idx_ = ei, have_touch_ = true;         // main-path nodes beyond index ei were not modified
    if (dbg) for (size_t i = ei; i < n_path_edges; ++i) if (dbg_before[i] != out_sig(main_edges[i]->sink)) { fprintf(stderr, "UPDATE touched node %zu beyond ei=%zu (begin %zd end %zd)\n", i + 1, ei, begin_offset, end_offset); break; }
}

void ContigGraph::clear_main_path()
{
    const size_t l = main_edges.size();
    for (size_t i = right_off_; i < l; ++i) set_on_main(main_edges[i]->sink, false);
    if (right_off_ < main_edges.size()) main_edges.erase(main_edges.begin() + right_off_, main_edges.end());
    if (main_path.size() > right_off_ + 1) main_path.erase(main_path.begin() + right_off_ + 1, main_path.end());
    for (size_t i = 0; i < left_off_; ++i) set_on_main(main_edges[i]->source, false);
    if (left_off_ > 0) {
        main_edges.erase(main_edges.begin(), main_edges.begin() + left_off_);
        main_path.erase(main_path.begin(), main_path.begin() + left_off_);
        right_off_ -= left_off_;
    }
    left_off_ = 0;
}

void ContigGraph::calculate_main_path_greedy()
{
    static const bool no_splice = getenv("NSGPU_NO_TAIL_SPLICE") != nullptr;     // debugging aid: always re-walk the tail
    const size_t m = main_edges.size();
    const size_t dbg_R = right_off_, dbg_L = left_off_, dbg_cf = consistent_from_;
    // Main-path nodes the update touched but that lie in the part the reference keeps as it is ([left_off_, right_off_))
    // are not re-walked: their choice may now differ from best_out, and they stop counting as consistent.
    if (have_touch_ && touch_lo_ != (size_t)-1 && right_off_ >= 1 && consistent_from_ != (size_t)-1) {
        const size_t hi = touch_idx_ < right_off_ - 1 ? touch_idx_ : right_off_ - 1, lo = touch_lo_ > left_off_ ? touch_lo_ : left_off_;
        if (lo <= hi && consistent_from_ < hi + 1) consistent_from_ = hi + 1;
    }
    const bool try_splice = !no_splice && have_touch_ && left_off_ == 0 && m > 0 && right_off_ <= m && consistent_from_ != (size_t)-1;
    have_touch_ = false;
    if (try_splice) {
        // ---- exact shortcut: re-walk only where the greedy choice can have changed ----
        // The old path from index consistent_from_ on was chosen by best_out on the counts of its time.  The update
        // since then added one read: along main-path edges it raises the count of the edge that was already the
        // (first) maximum, so the choice stands; only where the read left the path over a side edge (diverged_) can
        // another edge win.  Those nodes, the nodes of [right_off_, consistent_from_) and the old end are checked;
        // a changed choice is followed until it re-joins the old path (or ends), and everything between is re-used.
        const size_t R = right_off_;
        std::vector<size_t> &cand = cand_;
        cand.clear();
        const size_t chk_end = consistent_from_ < m + 1 ? consistent_from_ : m + 1;
        for (size_t i = R; i < chk_end; ++i) cand.push_back(i);
        for (size_t d : diverged_) if (d >= R && d >= chk_end && d <= m && (cand.empty() || cand.back() < d)) cand.push_back(d);
        if (cand.empty() || cand.back() != m) cand.push_back(m);
        auto old_node = [&](size_t i) { return i == 0 ? main_edges[0]->source : main_edges[i - 1]->sink; };
        // The candidates are scattered over the path: a node, its out-edges and their counts are three dependent cache
        // misses each.  The list is known up front, so the misses of the next candidates are started while this one is
        // examined: edge c-1 (holds the node pointer) 9 ahead, the node 6 ahead, its out-edges 3 ahead.
        Edge *const *old_edges = main_edges.begin();                     // re-pointed at `saved` once the path is cut
        size_t old_base = 0;                                             // old edge i = old_edges[i - old_base]
        auto fetch_ahead = [&](size_t ci_now) {
            if (ci_now + 9 < cand.size()) { const size_t c9 = cand[ci_now + 9]; if (c9 >= 1 && c9 - 1 >= old_base) __builtin_prefetch(old_edges[c9 - 1 - old_base], 0, 1); }
            if (ci_now + 6 < cand.size()) { const size_t c6 = cand[ci_now + 6]; if (c6 >= 1 && c6 - 1 >= old_base) __builtin_prefetch(old_edges[c6 - 1 - old_base]->sink, 0, 1); }
            if (ci_now + 3 < cand.size()) {
                const size_t c3 = cand[ci_now + 3];
                if (c3 >= 1 && c3 - 1 >= old_base) { const Node *n3 = old_edges[c3 - 1 - old_base]->sink; for (const Edge *pe : n3->out) __builtin_prefetch(pe, 0, 1); }
            }
        };
        // first candidate whose choice changed (nothing to do before it)
        size_t ci = 0;
        for (; ci < cand.size(); ++ci) {
            const size_t c = cand[ci];
            fetch_ahead(ci);
            if (old_node(c)->best_out() != (c < m ? main_edges[c] : nullptr)) break;
        }
        if (ci < cand.size()) {
            const size_t c0 = cand[ci];
            if (c0 + 1 < path_changed_from) path_changed_from = c0 + 1;       // main_path[0 .. c0] stays
            std::vector<Edge *> &saved = saved_;                         // saved[t] = old edge c0 + t, its sink = old node c0 + t + 1
            saved.assign(main_edges.begin() + c0, main_edges.end());
            g_mp_cnt[0] += saved.size(), g_mp_cnt[1] += 1, g_mp_cnt[2] += main_edges.size();
            const std::string saved_str = main_path.substr(c0 + 1);
            Node *at = old_node(c0);
            main_edges.erase(main_edges.begin() + c0, main_edges.end());
            main_path.erase(main_path.begin() + c0 + 1, main_path.end());
            size_t pos = c0;                                             // the new path so far ends at old node `pos` (== at)
            bool ended = false;
            old_edges = saved.data(), old_base = c0;
            for (; ci < cand.size() && !ended; ++ci) {
                const size_t c = cand[ci];
                fetch_ahead(ci);
                if (c < pos) continue;                                   // by-passed by an earlier detour
                // old nodes pos .. c keep their edges
                main_edges.append(saved.begin() + (pos - c0), saved.begin() + (c - c0));
                main_path.append(saved_str, pos - c0, c - pos);
                if (c > pos) at = saved[c - c0 - 1]->sink;
                pos = c;
                Edge *e = at->best_out();
                if (e == (c < m ? saved[c - c0] : nullptr)) continue;
                // detour: follow the greedy walk until it meets the old path again
                ++dbg_spliced;
                for (;;) {
                    if (!e) {                                            // the path ends here: the rest of the old path is off
                        for (size_t t = pos - c0; t < saved.size(); ++t) set_on_main(saved[t]->sink, false);
                        ended = true;
                        break;
                    }
                    Node *nx = e->sink;
                    main_edges.push_back(e);
                    main_path.push_back(nx->base);
                    ++dbg_walked_nodes;
                    if (nx->on_main) {                                   // an old-path node further down (everything else is off the path)
                        size_t j = pos + 1;
                        while (saved[j - c0 - 1]->sink != nx) ++j;
                        for (size_t t = pos + 1; t < j; ++t) set_on_main(saved[t - c0 - 1]->sink, false);
                        pos = j, at = nx;
                        break;
                    }
                    set_on_main(nx, true);
                    at = nx;
                    e = at->best_out();
                }
            }
            if (!ended) {
                main_edges.append(saved.begin() + (pos - c0), saved.end());
                main_path.append(saved_str, pos - c0, std::string::npos);
                dbg_spliced_nodes += saved.size() - (pos - c0);
            }
        }
        static const bool dbg_chk = getenv("NSGPU_SPLICE_CHECK") != nullptr;
        if (dbg_chk) {                                                   // brute force: the plain walk from R must give the same edges
            Node *c2 = old_node(R < main_edges.size() + 1 ? R : 0);
            for (size_t i = R; ; ++i) {
                Edge *bo = c2->best_out();
                Edge *have = i < main_edges.size() ? main_edges[i] : nullptr;
                if (bo != have) {
                    fprintf(stderr, "SPLICE MISMATCH at index %zu of %zu (R=%zu m=%zu cf=%zu chk_end=%zu touch=%zu)\n", i, main_edges.size(), R, m, consistent_from_, chk_end, touch_idx_);
                    fprintf(stderr, "  cand tail:"); for (size_t q = cand.size() > 8 ? cand.size() - 8 : 0; q < cand.size(); ++q) fprintf(stderr, " %zu", cand[q]);
                    fprintf(stderr, "\n  diverged tail:"); for (size_t q = diverged_.size() > 8 ? diverged_.size() - 8 : 0; q < diverged_.size(); ++q) fprintf(stderr, " %zu", diverged_[q]);
                    fprintf(stderr, "\n  out-edges of node:"); for (Edge *oe : c2->out) fprintf(stderr, " [cnt %u sink_main %d %s%s]", oe->count, (int)oe->sink->on_main, oe == bo ? "best" : "", oe == have ? "have" : "");
                    fprintf(stderr, "\n");
                    break;
                }
                if (!bo) break;
                c2 = bo->sink;
            }
        }
        main_edges.back()->sort_reads();
        const read_t ending = *main_edges.back()->reads.begin();
        const GraphRead &er = reads.at(ending);
        end_pos = er.pos + (long)er.len;
        if (R < consistent_from_) consistent_from_ = R;
    } else {
        if (consistent_from_ != (size_t)-1) consistent_from_ = consistent_from_ > left_off_ ? consistent_from_ - left_off_ : 0;
        path_changed_from = left_off_ > 0 ? 0 : std::min(path_changed_from, right_off_ + 1);     // what clear_main_path keeps on the left
        clear_main_path();                            // right_off_ is now the index of right_unchanged_
        if (right_off_ < consistent_from_) consistent_from_ = right_off_;
        Node *cur = right_unchanged_;
        Edge *e;
        while ((e = cur->best_out())) {
            main_edges.push_back(e);
            cur = e->sink;
            set_on_main(cur, true);
            main_path.push_back(cur->base);
            ++dbg_walked_nodes;
        }
        main_edges.back()->sort_reads();
        const read_t ending = *main_edges.back()->reads.begin();
        const GraphRead &er = reads.at(ending);
        end_pos = er.pos + (long)er.len;
    }
    {
        Node *cur = left_unchanged_;
        Edge *e;
        std::string prefix;                          // collected back to front, prepended once
        while ((e = cur->best_in())) {
            main_edges.push_front(e);
            cur = e->source;
            set_on_main(cur, true);
            prefix.push_back(cur->base);
            ++left_off_;
            ++right_off_;
            if (consistent_from_ != (size_t)-1) ++consistent_from_;      // nodes chosen by best_in are not trusted
        }
        if (!prefix.empty()) {
            std::reverse(prefix.begin(), prefix.end());
            main_path.insert(0, prefix);
            path_changed_from = 0;
        }
        main_edges.front()->sort_reads();
        const read_t starting = *main_edges.front()->reads.begin();
        start_pos = reads.at(starting).pos;
    }
    const uint64_t splits_before = n_splits_;
    const auto tc0 = std::chrono::steady_clock::now();
    const uint64_t skipped_before = dbg_cycles_skipped;
    remove_cycles();
    if (dbg_cycles_skipped == skipped_before && n_splits_ == splits_before) ++dbg_cycles_idle;
    dbg_cycles_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
    // A re-routing (split_path) leaves best_out of every consistent main-path node as it was: at a main-path source it
    // replaces a side edge by a new edge with the same reads at the END of the out list (the main edge, being the first
    // maximum, stays the first maximum); deeper levels only touch edges out of side nodes and edges INTO main nodes.
    right_unchanged_ = main_edges.back()->sink;
    right_off_ = main_edges.size();
    left_unchanged_ = main_edges.front()->source;
    left_off_ = 0;
    static const bool dbg_inv = getenv("NSGPU_SPLICE_CHECK") != nullptr;
    if (dbg_inv && consistent_from_ != (size_t)-1) {
        for (size_t i = consistent_from_; i <= main_edges.size(); ++i) {
            Node *nd = i == 0 ? main_edges[0]->source : main_edges[i - 1]->sink;
            Edge *bo = nd->best_out(), *have = i < main_edges.size() ? main_edges[i] : nullptr;
            if (bo != have) { fprintf(stderr, "INVARIANT broken at index %zu of %zu (cf=%zu, spliced=%d, splits %d, call %llu; before: m=%zu R=%zu L=%zu cf=%zu touch=%zu)\n", i, main_edges.size(), consistent_from_, (int)try_splice, (int)(n_splits_ != splits_before), (unsigned long long)dbg_cycles_calls, m, dbg_R, dbg_L, dbg_cf, touch_idx_); break; }
        }
    }
}

void ContigGraph::remove_cycles()
{
    ++dbg_cycles_calls;
    static const bool no_skip = getenv("NSGPU_NO_CYCLE_SKIP") != nullptr;       // debugging aid: always walk, as the reference does
    if (n_multi_in_side_ == 0 && !no_skip) { ++dbg_cycles_skipped; multi_in_list_.clear(); return; }   // exact: walk_and_prune only ever acts on side nodes with in-degree > 1
    static const bool full_scan = getenv("NSGPU_CYCLE_FULLSCAN") != nullptr;   // debugging aid: always the reference's walk over every side branch
    if (!no_skip && !full_scan && remove_cycles_from_list()) { ++dbg_cycles_listed; return; }
    std::vector<Edge *> stack;
    {
        size_t ei = right_off_;
        const size_t end = main_edges.size();
        Node *n = ei < end ? main_edges[ei]->source : main_edges[ei - 1]->sink;
        std::vector<Edge *> copy;
        for (;;) {
            // walk_and_prune returns at once for edges into main-path nodes; only side branches need the
            // (copied, because the walk edits n->out) edge list
            bool side = false;
            // a node whose only out-edge is the path's own edge has no side branch: decided without touching the edge
            if (!(ei < end && n->out.size() == 1 && n->out[0] == main_edges[ei]))
                for (Edge *e : n->out) if (!e->sink->on_main) { side = true; break; }
            if (side) {
                copy.assign(n->out.begin(), n->out.end());
                for (Edge *e : copy) walk_and_prune(e, stack);
            }
            if (ei == end) break;
            n = main_edges[ei]->sink;
            ++ei;
        }
    }
    {
        size_t ei = left_off_ < main_edges.size() ? left_off_ : main_edges.size() - 1;
        std::vector<Edge *> copy;
        for (;;) {
            Node *n = main_edges[ei]->source;
            bool side = false;
            for (Edge *e : n->out) if (!e->sink->on_main) { side = true; break; }
            if (side) {
                copy.assign(n->out.begin(), n->out.end());
                for (Edge *e : copy) walk_and_prune(e, stack);
            }
            if (ei == 0) break;
            --ei;
        }
    }
}

// removeCycles without the search.  The reference walks every side branch of the scanned main-path ranges depth-first and
// splits at each side edge whose sink (still) has in-degree > 1.  The nodes with that property are known (multi_in_list_);
// every edge at which a split can happen lies on a path main node -> ... -> such a node, i.e. its sink is an ancestor-or-self
// of one of them.  So: mark those (walking in-edges backwards; side nodes other than the listed ones have exactly one),
// collect the main-path nodes the marked branches hang off, and run the reference's walk from those nodes only, in the
// reference's order (first loop: path order from right_off_; second loop: backwards from left_off_), descending only into
// marked nodes.  The visited edges are a subsequence of the reference's walk that contains every edge it can split at, in
// the same order; splits create no new node of that kind (split_path gives private copies), so the marks stay a superset.
bool ContigGraph::remove_cycles_from_list()
{
    // the listed nodes that still qualify, each once
    std::vector<Node *> &L = multi_in_list_;
    ++scan_epoch_;
    if (scan_epoch_ == 0) return false;                      // epoch wrapped: one full walk, marks of 2^32 calls ago may linger
    size_t k = 0;
    for (Node *n : L) if (multi_in_side(n) && n->reserved_ != scan_epoch_) { n->reserved_ = scan_epoch_; L[k++] = n; }
    L.resize(k);
    if (k != n_multi_in_side_) return false;                 // something the list does not know about: the full walk decides
    // ancestors
    std::vector<Node *> todo(L.begin(), L.end()), roots;
    while (!todo.empty()) {
        Node *n = todo.back();
        todo.pop_back();
        for (Edge *e : n->in) {
            Node *s = e->source;
            if (s->reserved_ == scan_epoch_) continue;
            s->reserved_ = scan_epoch_;
            if (s->on_main) roots.push_back(s); else todo.push_back(s);
        }
    }
    if (roots.empty()) return true;                          // not reachable from the path at all: the full walk would find nothing either
    // positions of the roots on the path: the path's edge array is searched for the roots' out-edges (no node is touched)
    const size_t m = main_edges.size();
    struct Hit { size_t idx; Node *n; };
    std::vector<Hit> hits;
    {
        std::vector<const Edge *> outs;
        uint64_t bloom = 0;
        for (Node *r : roots) for (const OutRef &o : r->out) { outs.push_back(o.get()); bloom |= 1ull << ((reinterpret_cast<uintptr_t>(o.get()) >> 6) & 63); }
        Edge *const *pe = main_edges.begin();
        auto scan = [&](size_t lo, size_t hi) {                // path edges lo .. hi-1: their sources are path nodes lo .. hi-1
            for (size_t i = lo; i < hi; ++i) {
                const Edge *e = pe[i];
                if (!((bloom >> ((reinterpret_cast<uintptr_t>(e) >> 6) & 63)) & 1)) continue;
                if (std::find(outs.begin(), outs.end(), e) == outs.end()) continue;
                hits.push_back(Hit{i, e->source});
            }
        };
        const size_t left_end = left_off_ < m ? left_off_ + 1 : m;      // second loop: edges left_off_ (or m - 1) down to 0
        if (right_off_ < left_end) scan(0, m);
        else { scan(0, left_end); scan(right_off_, m); }
        Node *last = main_edges[m - 1]->sink;                  // the path's last node has no path edge of its own
        if (last->reserved_ == scan_epoch_) hits.push_back(Hit{m, last});
    }
    std::vector<Edge *> stack, copy;
    auto run = [&](Node *n) {
        copy.assign(n->out.begin(), n->out.end());
        for (Edge *e : copy) walk_and_prune_marked(e, stack);
    };
    // first loop of the reference: nodes right_off_ .. m in path order
    std::sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.idx < b.idx; });
    for (const Hit &h : hits) if (h.idx >= right_off_) run(h.n);
    // second loop: sources of edges min(left_off_, m - 1) .. 0, backwards
    const size_t l0 = left_off_ < m ? left_off_ : m - 1;
    for (size_t i = hits.size(); i-- > 0;) if (hits[i].idx <= l0 && hits[i].idx < m) run(hits[i].n);
    return true;
}

void ContigGraph::walk_and_prune_marked(Edge *e, std::vector<Edge *> &stack)
{
    stack.push_back(e);
    while (!stack.empty()) {
        Edge *curr = stack.back();
        stack.pop_back();
        Node *sink = curr->sink, *source = curr->source;
        if (sink->on_main || sink->reserved_ != scan_epoch_) continue;      // nothing below an unmarked node can be split
        if (sink->in.size() > 1) { curr->sort_reads(); split_path(source, curr, std::vector<read_t>(curr->reads.begin(), curr->reads.end())); }
        for (Edge *o : sink->out) stack.push_back(o);
    }
}

void ContigGraph::walk_and_prune(Edge *e, std::vector<Edge *> &stack)
{
    stack.push_back(e);
    while (!stack.empty()) {
        Edge *curr = stack.back();
        stack.pop_back();
        Node *sink = curr->sink, *source = curr->source;
        if (sink->on_main) continue;
        if (sink->in.size() > 1) { curr->sort_reads(); split_path(source, curr, std::vector<read_t>(curr->reads.begin(), curr->reads.end())); }
        for (Edge *o : sink->out) stack.push_back(o);
    }
}

// Give the reads of edge e (a side branch entering a node that has other ways in) a private copy
// of everything downstream until the main path is reached again; every side node then has
// in-degree one.  Iterative form of the reference's two-visit context stack.
void ContigGraph::split_path(Node *new_pre0, Edge *e0, const std::vector<read_t> &reads0)
{
    ++n_splits_;
    struct Ctx {
        Node *new_pre; Edge *e;
        const std::vector<read_t> *in_reads;       // borrowed from the parent context (or the caller)
        std::vector<read_t> own;                   // intersection computed at the first visit
        bool visited = false;
        Node *old_cur = nullptr;
    };
    std::deque<Ctx> st;                               // deque: stable addresses while children are pushed
    const std::vector<read_t> &first_copy = reads0;   // a copy owned by the caller: e0->reads dies with e0 during the first visit
    st.push_back(Ctx{new_pre0, e0, &first_copy, {}, false, nullptr});
    while (!st.empty()) {
        Ctx &c = st.back();
        if (c.visited) {
            Node *oc = c.old_cur;
            st.pop_back();
            if (oc && oc->in.empty() && oc->out.empty()) remove_node(oc);
            continue;
        }
        c.e->sort_reads();
        std::set_intersection(c.in_reads->begin(), c.in_reads->end(), c.e->reads.begin(), c.e->reads.end(), std::back_inserter(c.own));
        c.visited = true;
        if (c.own.empty()) continue;
        Node *old_cur = c.e->sink;
        c.old_cur = old_cur;
        remove_reads_from_edge(c.e, c.own);
        if (old_cur->on_main) { create_edge(c.new_pre, old_cur, c.own); continue; }
        Node *new_cur = create_node(old_cur->base);
        create_edge(c.new_pre, new_cur, c.own);
        const std::vector<read_t> *mine = &c.own;
        const std::vector<Edge *> outs(old_cur->out.begin(), old_cur->out.end());
        for (Edge *o : outs) st.push_back(Ctx{new_cur, o, mine, {}, false, nullptr});
    }
}

// ---------------------------------------------------------------------------
// emission
// ---------------------------------------------------------------------------
void write_var_uint32(uint32_t v, std::string &out)
{
    while (v > 127) { out.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
    out.push_back((char)(v & 0x7f));
}

size_t optimize_edit_script(const std::vector<EditOp> &in, std::vector<EditOp> &out)
{
    size_t dis = 0;
    out.clear();
    size_t i = 0;
    const size_t n = in.size();
    while (i < n) {
        while (i < n && in[i].type == 0) out.push_back(in[i++]);
        std::string ins;
        size_t n_del = 0;
        while (i < n && in[i].type != 0) {
            if (in[i].type == 1) ins.push_back((char)in[i].base); else ++n_del;
            ++i;
        }
        const size_t n_ins = ins.size(), n_sub = std::min(n_ins, n_del);
        dis += std::max(n_del, n_ins);
        size_t k;
        for (k = 0; k < n_sub; ++k) out.push_back(EditOp{3, (uint8_t)ins[k], 0});
        if (n_ins > n_del) for (; k < n_ins; ++k) out.push_back(EditOp{1, (uint8_t)ins[k], 0});
        else for (; k < n_del; ++k) out.push_back(EditOp{2, (uint8_t)'-', 0});
    }
    return dis;
}

// The nodes a read threads, in order.  Without a base source this is the reference's walk (the out-edge whose read
// list holds the id, Node::getNextNodeInRead).  With the read's own bases at hand the walk is guided by them: a node
// with one out-edge needs no test at all, and among several out-edges the one whose sink carries the read's next base
// is the read's edge whenever it is the only such edge -- the list lookup is only needed to break ties.
// visit(node) for every node of the read in order; visit_run(k) stands for k consecutive main-path nodes that follow the
// node visited last on the main path (only used when the read's bases are at hand).
static inline uint8_t base_bit(char b) { return b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 4 : b == 'T' ? 8 : 16; }

// The out-edge of node n that read `id` takes when its next base is nb: the only out-edge, else the only one whose sink carries
// the base.  When several do (8 % of the positions of a read at cfg2: a side branch that starts with the consensus's next base --
// an inserted copy of it, or a deletion edge onto an equal base), the read is on exactly one of them: the short lists are looked
// through (a side branch holds a read or two, inside the edge's own cache line) and the read is on the longest one if it is on
// none of those -- the long list, sorted on demand and spilled to the arena, is never touched.
static inline bool edge_lists_read(const Edge *e, read_t id)
{
    const uint32_t n = e->reads.n;
    const read_t *b = e->reads.data();
    if (n <= 16) { for (uint32_t x = 0; x < n; ++x) if (b[x] == id) return true; return false; }
    const_cast<Edge *>(e)->sort_reads();
    b = e->reads.data();
    size_t m = n;
    if (id < b[0] || id > b[m - 1]) return false;
    while (m > 1) { const size_t h = m >> 1; b = b[h] <= id ? b + h : b; m -= h; }
    return *b == id;
}
static inline const Edge *way_out_of(const Node *n, char nb, read_t id)
{
    const auto &out = n->out;
    if (out.size() == 1) return out[0].get();
    const Edge *cand[8];
    int cnt = 0;
    for (const OutRef &o : out) if (o.sink_base_is(nb)) { if (cnt < 8) cand[cnt] = o.get(); ++cnt; }
    if (cnt == 1) return cand[0];
    if (cnt == 0 || cnt > 8) return n->edge_in_read(id);
    int big = 0;
    for (int c = 1; c < cnt; ++c) if (cand[c]->reads.n > cand[big]->reads.n) big = c;
    for (int c = 0; c < cnt; ++c) if (c != big && edge_lists_read(cand[c], id)) return cand[c];
    return cand[big];
}

template <class Visit, class VisitRun>
void ContigGraph::walk_read(const GraphRead &r, read_t id, const ReadBases *src, Visit visit, VisitRun visit_run) const
{
    const Node *cur = r.start;
    if (!src) {
        while (cur) { visit(cur); const Edge *e = cur->edge_in_read(id); cur = e ? e->sink : nullptr; }
        return;
    }
    const size_t L = src->len;
    // the read as it lies on the graph (reverse-complemented reads once, into a per-thread buffer)
    static thread_local std::string oriented;
    const char *rb = src->bases;
    if (r.rc) { reverse_complement(src->bases, L, oriented); rb = oriented.data(); }
    const size_t n_main = main_edges.size();
    const char *const cons = main_path.data();           // cons[j] = base of main-path node j
    for (size_t i = 0; i < L;) {
        // here `cur` carries base i and has not been visited
        visit(cur);
        if (cur->on_main) {
            // A read mostly follows the consensus, and while it does no node is looked at: j = index of the current main-path
            // node (cum_weight), next_fork_[j] = first index >= j whose node has more than one way out or ends the path,
            // side_mask_[j] = bases of the side sinks of node j, cons = the path's bases (all set by write_reads).
            size_t j = cur->cum_weight;
            for (;;) {
                if (++i == L) return;
                if (j < n_main) {
                    // the longest stretch on which the read's bases are the consensus's and nothing else could carry them: 8 bases per
                    // compare (read ^ consensus, and the follow_ok_ bytes all 1)
                    const size_t lim = L - i < n_main - j ? L - i : n_main - j;
                    const char *a = rb + i, *b = cons + j + 1;
                    const uint8_t *ok = follow_ok_.data() + j;
                    size_t t = 0;
                    while (t + 8 <= lim) {
                        uint64_t x, y, z;
                        memcpy(&x, a + t, 8), memcpy(&y, b + t, 8), memcpy(&z, ok + t, 8);
                        const uint64_t d = (x ^ y) | (z ^ 0x0101010101010101ull);
                        if (d) { t += (size_t)(__builtin_ctzll(d) >> 3); goto run_done; }
                        t += 8;
                    }
                    while (t < lim && a[t] == b[t] && ok[t]) ++t;
                run_done:
                    if (t) {
                        visit_run(t);
                        j += t, i += t - 1;
                        continue;
                    }
                }
                if (j < n_main && next_fork_[j] != j) {
                    // single way out: it leads to the next main-path node, and so on up to the next fork
                    const size_t j1 = j + 1, stop = next_fork_[j1];
                    const size_t k = stop - j1 < L - i ? stop - j1 : L - i;
                    if (k) { visit_run(k); i += k; if (i == L) return; }
                    j = j1 + k;
                    visit_run(1);                        // node j, right behind the run
                    continue;
                }
                // a fork (or the path's end): when the consensus goes on with the read's next base and no side branch starts
                // with that base, the read's edge is the path's edge (the only out-edge whose sink carries the base)
                const char nb = rb[i];
                if (j < n_main && cons[j + 1] == nb) {
                    if (!(side_mask_[j] & base_bit(nb))) { ++j; visit_run(1); continue; }
                    // a side branch starts with the same base: the read stays on the path unless that branch lists it
                    const read_t *a = amb_ids_.data() + amb_off_[j], *b = amb_ids_.data() + amb_off_[j + 1];
                    bool side = false;
                    for (; a != b; ++a) if (*a == id || *a == kAmbComplex) { side = true; break; }
                    if (!side) { ++j; visit_run(1); continue; }
                }
                cur = way_out_of(main_nodes_[j], nb, id)->sink;
                break;
            }
            continue;
        }
        if (++i == L) break;                             // i = bases consumed
        cur = way_out_of(cur, rb[i], id)->sink;
    }
}

// read2EditScript in one pass over the read's nodes: everything before the first main-path node is an insert, `pos` is that
// node's consensus position (0 for a read that never touches the consensus, whose script is all inserts)
size_t ContigGraph::read_to_edits(const GraphRead &r, read_t id, const ReadBases *src, std::vector<EditOp> &script, uint32_t &pos) const
{
    script.clear();
    script.reserve(r.len / 8 + 16);
    bool seen_main = false;
    size_t dis = 0, at = 0, same = 0;
    pos = 0;
    auto flush = [&]() { if (same > 0) { script.push_back(EditOp{0, 0, (uint32_t)same}); same = 0; } };
    walk_read(r, id, src, [&](const Node *cur) {
        if (cur->on_main) {
            const size_t p = cur->cum_weight;
            if (!seen_main) seen_main = true, pos = (uint32_t)p, at = p;
            if (p > at) flush();
            for (; at < p; ++at) { script.push_back(EditOp{2, (uint8_t)'-', 0}); ++dis; }
            ++same;
            ++at;
        } else {
            flush();
            script.push_back(EditOp{1, (uint8_t)cur->base, 0});
            ++dis;
        }
    }, [&](size_t k) { same += k, at += k; });       // k main-path nodes in a row right behind the last one: k more SAMEs
    flush();
    return dis;
}

std::atomic<uint64_t> g_emit_ns[4];
static inline uint64_t emit_now() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

size_t ContigGraph::write_read(StreamSet &o, const GraphRead &r, read_t id, const ReadBases *src) const
{
    uint32_t offset;
    static thread_local std::vector<EditOp> raw, es;
    const uint64_t w0 = emit_now();
    read_to_edits(r, id, src, raw, offset);
    const uint64_t w1 = emit_now();
    g_emit_ns[1] += w1 - w0;
    write_var_uint32(offset, o.pos);
    const size_t dis = optimize_edit_script(raw, es);
    uint32_t ins_start = 0, ins_end = 0;
    for (size_t i = 0; i != es.size(); ++i) {
        if (es[i].type != 1) break;
        ++ins_start;
        o.base.push_back((char)es[i].base);
    }
    if (ins_start != es.size())
        for (int64_t i = (int64_t)es.size() - 1; i >= 0; --i) {
            if (es[i].type != 1) break;
            ++ins_end;
        }
    write_var_uint32(ins_start, o.pos);
    uint32_t same = 0;
    for (size_t i = ins_start; i < es.size() - ins_end; ++i) {
        switch (es[i].type) {
        case 0: same += es[i].num; break;
        case 1: write_var_uint32(same, o.pos); same = 0; o.type.push_back('i'); o.base.push_back((char)es[i].base); break;
        case 2: write_var_uint32(same, o.pos); same = 0; o.type.push_back('d'); break;
        case 3: write_var_uint32(same, o.pos); same = 0; o.type.push_back('s'); o.base.push_back((char)es[i].base); break;
        }
    }
    write_var_uint32(same, o.pos);
    write_var_uint32(ins_end, o.pos);
    for (size_t i = es.size() - ins_end; i != es.size(); ++i) o.base.push_back((char)es[i].base);
    o.type.push_back('\n');
    g_emit_ns[2] += emit_now() - w1;
    return dis;
}

void ContigGraph::write_main_path(StreamSet &o) const { o.genome += main_path; o.genome.push_back('\n'); }
void ContigGraph::write_read_lone(StreamSet &o) const { o.lone += main_path; o.lone.push_back('\n'); }

void ContigGraph::write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source)
{
    const uint64_t e0 = emit_now();
    main_edges.front()->source->cum_weight = 0;
    const size_t n_main = main_edges.size();
    main_nodes_.resize(n_main + 1);
    next_fork_.resize(n_main + 1);
    side_mask_.assign(n_main + 1, 0);
    amb_off_.assign(n_main + 2, 0);
    amb_ids_.clear();
    // one pass over the path's nodes while their lines are at hand: index, fork flag, bases of the side branches (from the
    // out-edge references; no edge or sink is touched)
    auto note = [&](size_t j, const Node *n) {
        const bool single = n->out.size() == 1 && j < n_main;
        next_fork_[j] = single ? UINT32_MAX : (uint32_t)j;
        amb_off_[j + 1] = (uint32_t)amb_ids_.size();
        if (single) return;
        const Edge *path_edge = j < n_main ? main_edges[j] : nullptr;
        uint8_t m = 0;
        for (const OutRef &o : n->out) if (o.get() != path_edge) m |= base_bit(o.sink_base());
        side_mask_[j] = m;
        // side branches that start with the consensus's own next base (an inserted copy of it, a deletion edge onto an equal
        // base): the reads on them, so that the walk of every other read passes this node without looking at the graph
        if (j < n_main && (m & base_bit(main_path[j + 1]))) {
            const size_t at = amb_ids_.size();
            bool simple = true;
            for (const OutRef &o : n->out)
                if (o.get() != path_edge && o.sink_base_is(main_path[j + 1])) {
                    const Edge *e = o.get();
                    if (e->reads.n > 16) { simple = false; break; }
                    amb_ids_.insert(amb_ids_.end(), e->reads.data(), e->reads.data() + e->reads.n);
                }
            if (!simple) { amb_ids_.resize(at); amb_ids_.push_back(kAmbComplex); }
        }
        amb_off_[j + 1] = (uint32_t)amb_ids_.size();
    };
    main_nodes_[0] = main_edges.front()->source;
    note(0, main_nodes_[0]);
    for (size_t t = 0; t < n_main; ++t) {
        // two dependent misses per position (edge, then its sink): keep both in flight ahead of the loop
        constexpr size_t pf_a = 16;
        if (t + pf_a < n_main) __builtin_prefetch(main_edges[t + pf_a], 0, 1);
        if (t + pf_a / 2 < n_main) __builtin_prefetch(main_edges[t + pf_a / 2]->sink, 1, 1);
        Node *n = main_edges[t]->sink;
        n->cum_weight = t + 1;
        main_nodes_[t + 1] = n;
        note(t + 1, n);
    }
    // follow_ok_[j]: node j has one way out, or none of its side branches starts with the consensus's next base -- then a read
    // whose next base equals cons[j + 1] continues on the path, and the walk can compare whole words of read and consensus
    follow_ok_.assign(n_main + 8, 0);
    for (size_t j = 0; j < n_main; ++j) follow_ok_[j] = next_fork_[j] == UINT32_MAX || !(side_mask_[j] & base_bit(main_path[j + 1]));
    for (size_t j = n_main; j-- > 0;) if (next_fork_[j] == UINT32_MAX) next_fork_[j] = next_fork_[j + 1];
    g_emit_ns[0] += emit_now() - e0;
    read_t prev = 0;
    for (auto &it : reads) {
        const read_t diff = it.first - prev;
        o.id_contigs.append(reinterpret_cast<const char *>(&diff), 4);     // 4 bytes: std::ios::binary == 4 is passed as the count (:998)
        o.complement.push_back(it.second.rc ? 'c' : 'n');
        prev = it.first;
        if (source) { const ReadBases rb = (*source)(it.first); write_read(o, it.second, it.first, &rb); }
        else write_read(o, it.second, it.first, nullptr);
    }
    o.complement.push_back('\n');
}

bool ContigGraph::read_string(read_t id, std::string &out) const
{
    out.clear();
    auto it = reads.find(id);
    if (it == reads.end()) return false;
    const Node *cur = it->second.start;
    while (cur) {
        out.push_back(cur->base);
        Edge *e = cur->edge_in_read(id);
        cur = e ? e->sink : nullptr;
        if (out.size() > it->second.len + 8) return false;
    }
    return out.size() == it->second.len;
}

bool ContigGraph::has_cycle() const
{
    // every read must thread a simple path: a cycle shows up as a read walk longer than the read
    std::string tmp;
    for (auto &it : reads) if (!read_string(it.first, tmp)) return true;
    return false;
}

// ---------------------------------------------------------------------------
// stream sets
// ---------------------------------------------------------------------------
void StreamSet::append(const StreamSet &o)
{
    genome += o.genome; lone += o.lone; pos += o.pos; type += o.type; base += o.base; complement += o.complement; id_contigs += o.id_contigs;
    lone_ids.insert(lone_ids.end(), o.lone_ids.begin(), o.lone_ids.end());
    reads_in_contig.insert(reads_in_contig.end(), o.reads_in_contig.begin(), o.reads_in_contig.end());
}

std::string StreamSet::id_bytes() const
{
    std::string s = id_contigs;
    read_t prev = 0;
    for (read_t r : lone_ids) {
        const read_t diff = r - prev;
        s.append(reinterpret_cast<const char *>(&diff), 4);
        prev = r;
    }
    return s;
}

std::string meta_data(uint64_t n_reads, const std::vector<StreamSet> &threads)
{
    size_t n_contigs = 0;
    for (auto &t : threads) n_contigs += t.reads_in_contig.size();
    std::string s = "numReads=" + std::to_string(n_reads) + "\nnumContigs=" + std::to_string(n_contigs) + "\nnumThr=" +
                    std::to_string(threads.size()) + "\nnumReadsInContig=";
    for (auto &t : threads) for (read_t c : t.reads_in_contig) s += std::to_string(c) + ":";
    s.push_back('\n');
    return s;
}

namespace {
struct CompTable {
    char t[256];
    CompTable() { for (int i = 0; i < 256; ++i) t[i] = (char)i; t['A'] = 'T', t['T'] = 'A', t['C'] = 'G', t['G'] = 'C'; }
};
const CompTable kComp;
}  // namespace

void reverse_complement(const char *s, size_t n, std::string &out)
{
    out.resize(n);
    char *o = &out[0];
    for (size_t i = 0; i < n; ++i) o[i] = kComp.t[(uint8_t)s[n - 1 - i]];
}
void reverse_complement(const std::string &s, std::string &out) { reverse_complement(s.data(), s.size(), out); }

// ---------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------
namespace {
struct Cursor {
    const std::string &s; size_t p = 0;
    explicit Cursor(const std::string &x) : s(x) {}
    bool eof() const { return p >= s.size(); }
    bool get(char &c) { if (p >= s.size()) return false; c = s[p++]; return true; }
    bool var(uint32_t &v) {
        v = 0; uint8_t b, shift = 0;
        do { if (p >= s.size()) return false; b = (uint8_t)s[p++]; v |= (uint32_t)(b & 0x7f) << shift; shift += 7; } while (b & 0x80);
        return true;
    }
    bool u32(uint32_t &v) { if (p + 4 > s.size()) return false; memcpy(&v, s.data() + p, 4); p += 4; return true; }
    bool line(std::string &o) { if (p >= s.size()) return false; size_t e = s.find('\n', p); if (e == std::string::npos) e = s.size(); o.assign(s, p, e - p); p = e + 1; return true; }
};
}  // namespace

bool decode_streams(const StreamSet &ss, std::vector<std::pair<read_t, std::string>> &out, std::string &err)
{
    const std::string idb = ss.id_bytes();
    Cursor genome(ss.genome), id(idb), pos(ss.pos), type(ss.type), base(ss.base), comp(ss.complement), lone(ss.lone);
    std::string g, read;
    auto fail = [&](const char *m) { err = m; return false; };
    while (genome.line(g)) {
        read_t rid = 0;
        for (;;) {
            char c;
            if (!comp.get(c)) return fail(".complement ended early");
            if (c == '\n') break;
            uint32_t inc;
            if (!id.u32(inc)) return fail(".id ended early");
            rid += inc;
            // generateRead
            read.clear();
            uint32_t cur, n_start, n_end, same;
            if (!pos.var(cur) || !pos.var(n_start)) return fail(".pos ended early");
            for (uint32_t i = 0; i < n_start; ++i) { char b; if (!base.get(b)) return fail(".base ended early"); read.push_back(b); }
            for (;;) {
                if (!pos.var(same)) return fail(".pos ended early");
                if ((size_t)cur + same > g.size()) return fail("run past the consensus end");
                read.append(g, cur, same);
                cur += same;
                char t;
                if (!type.get(t)) return fail(".type ended early");
                if (t == '\n') break;
                if (t == 'd') ++cur;
                else if (t == 'i') { char b; if (!base.get(b)) return fail(".base ended early"); read.push_back(b); }
                else if (t == 's') { ++cur; char b; if (!base.get(b)) return fail(".base ended early"); read.push_back(b); }
                else return fail("bad edit type");
            }
            if (!pos.var(n_end)) return fail(".pos ended early");
            for (uint32_t i = 0; i < n_end; ++i) { char b; if (!base.get(b)) return fail(".base ended early"); read.push_back(b); }
            if (c == 'c') { std::string t2; reverse_complement(read, t2); read.swap(t2); }
            out.emplace_back(rid, read);
        }
    }
    read_t rid = 0;
    std::string l;
    while (lone.line(l)) {
        uint32_t inc;
        if (!id.u32(inc)) return fail(".id ended early (lone)");
        rid += inc;
        out.emplace_back(rid, l);
    }
    if (!id.eof() || !pos.eof() || !type.eof() || !base.eof() || !comp.eof()) return fail("trailing bytes in a stream");
    return true;
}

}  // namespace cons
}  // namespace nsgpu
