// consensus.hpp -- host side of the contig builder: the per-base consensus DAG
// (ConsensusGraph, src/ConsensusGraph.cpp:135-159, 400-897), consensus-edit emission and the
// seven output streams (src/ConsensusGraph.cpp:979-1178, src/Edits.cpp:23-60,
// src/DirectoryUtils.cpp:18-28), and the decoder that is the executable spec of the stream
// layout (Decompressor::generateRead, src/Decompressor.cpp:252-314).
//
// Pure host C++ (no HIP): the graph is pointer-chasing, per-contig sequential work; it runs on
// host threads, one contig builder per task, while sketching, candidate lookup and all
// alignment DP run on the GPU (consensus_driver.hip).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "mm2.hpp"

namespace nsgpu {
namespace cons {

typedef uint32_t read_t;

unsigned char *slab_acquire(size_t bytes);
void slab_release(unsigned char *p, size_t bytes);
constexpr size_t kSlabBytes = 512 * 1024;

// Per-graph bump allocator for the overflow storage of the small vectors below: power-of-two blocks (32 B << cls)
// carved out of recycled slabs, with one free list per class.  Nothing is freed one by one when a graph dies -- the
// slabs go back to the thread's slab cache.
class Arena {
public:
    Arena() = default;
    Arena(const Arena &) = delete;
    ~Arena();
    void *alloc(unsigned cls);
    void release(void *p, unsigned cls);
private:
    static constexpr unsigned kClasses = 13;   // 32 B .. 128 KiB from slabs; larger blocks come from malloc
    void *free_[kClasses] = {};
    std::vector<unsigned char *> slabs_;
    size_t used_ = kSlabBytes;
    std::vector<void *> big_;
};

// Vector with N inline slots (a graph node has one or two edges each way, a side edge carries one read): no heap
// block, no extra cache line in the common case.  sizeof == 8 + N * sizeof(T); growth doubles into the graph's Arena.
template <class T, unsigned N>
struct SmallVec {
    static_assert(sizeof(T) * N >= sizeof(T *) && (sizeof(T) * N * 2) % 32 == 0 && ((sizeof(T) * N * 2 / 32) & (sizeof(T) * N * 2 / 32 - 1)) == 0,
                  "overflow blocks must be 32 B << cls");
    uint32_t n = 0, cap = N;
    union { T inl[N]; T *heap; };
    SmallVec() {}
    SmallVec(const SmallVec &) = delete;
    SmallVec &operator=(const SmallVec &) = delete;
    T *data() { return cap == N ? inl : heap; }
    const T *data() const { return cap == N ? inl : heap; }
    T *begin() { return data(); }
    T *end() { return data() + n; }
    const T *begin() const { return data(); }
    const T *end() const { return data() + n; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return data()[i]; }
    const T &operator[](size_t i) const { return data()[i]; }
    static unsigned cls_of(uint32_t c) { unsigned k = 0; for (size_t b = sizeof(T) * c / 32; b > 1; b >>= 1) ++k; return k; }
    void reserve(Arena &a, size_t want)
    {
        if (want <= cap) return;
        uint32_t nc = cap;
        while (nc < want) nc *= 2;
        T *nb = static_cast<T *>(a.alloc(cls_of(nc)));
        memcpy(nb, data(), n * sizeof(T));
        if (cap != N) a.release(heap, cls_of(cap));
        heap = nb, cap = nc;
    }
    void push_back(Arena &a, const T &v) { if (n == cap) reserve(a, n + 1); data()[n++] = v; }
    void insert_at(Arena &a, size_t i, const T &v)
    {
        if (n == cap) reserve(a, n + 1);
        T *d = data();
        if (n - i <= 16) { for (size_t j = n; j > i; --j) d[j] = d[j - 1]; }       // short tails: cheaper than a libc call
        else memmove(d + i + 1, d + i, (n - i) * sizeof(T));
        d[i] = v;
        ++n;
    }
    void erase_at(size_t i) { T *d = data(); memmove(d + i, d + i + 1, (n - i - 1) * sizeof(T)); --n; }
    void assign(Arena &a, const T *p, size_t cnt) { n = 0; reserve(a, cnt); memcpy(data(), p, cnt * sizeof(T)); n = (uint32_t)cnt; }
    void release(Arena &a) { if (cap != N) { a.release(heap, cls_of(cap)); cap = N; } n = 0; }
};

struct Edge;
// An out-edge together with the base of the node it leads to, kept in the low bits of the pointer (edges are 64-byte
// aligned): choosing among a node's ways out by base -- the common question of update_graph and of the emission walk --
// then needs neither the edge's nor the sink's cache line.  Code 0-3 = A C G T, 4 = some other byte (look at the node).
struct OutRef {
    uintptr_t v;
    OutRef() = default;
    OutRef(Edge *e, char sink_base) : v(reinterpret_cast<uintptr_t>(e) | code_of(sink_base)) {}
    static unsigned code_of(char b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; }
    Edge *get() const { return reinterpret_cast<Edge *>(v & ~static_cast<uintptr_t>(63)); }
    operator Edge *() const { return get(); }
    Edge *operator->() const { return get(); }
    unsigned code() const { return (unsigned)(v & 7); }
    inline bool sink_base_is(char b) const;       // == (get()->sink->base == b)
    inline char sink_base() const;
};
struct Node {                                 // 64 bytes, slab-aligned: one cache line
    char base;
    bool on_main = false;
    uint32_t reserved_ = 0;
    SmallVec<OutRef, 2> out;
    SmallVec<Edge *, 2> in;
    size_t cum_weight = 0;
    explicit Node(char b) : base(b) {}
    Edge *edge_to(Node *n) const;             // Node::getEdgeTo          (:33-43)
    Edge *edge_to_side(char b) const;         // Node::getEdgeToSide      (:45-53)
    Edge *best_out() const;                   // Node::getBestEdgeOut     (:55-67) first max wins
    Edge *best_in() const;                    // Node::getBestEdgeIn      (:69-81)
    Edge *edge_in_read(read_t r) const;       // Node::getEdgeInRead      (:83-91)
};
// read ids held inside an edge before the list moves to the arena.  16 (128-byte edges) was measured: fewer second-line
// misses in update_graph do not pay for walking twice the bytes everywhere else (whole path -13 %).
#ifndef NSGPU_EDGE_INLINE_READS
#define NSGPU_EDGE_INLINE_READS 8
#endif
constexpr uint32_t kEdgeInlineReads = NSGPU_EDGE_INLINE_READS;
struct alignas(64) Edge {                     // 64 bytes (8 inline read ids)
    Node *source, *sink;
    read_t count;
    // The reference keeps the ids ascending (a std::set).  Here the first `sorted_n` ids are ascending and the rest is in
    // arrival order: a read follows ~8000 edges and arrives in no particular id order, so inserting in place cost a search
    // and a shift per edge and base, while most lists are never looked at in order.  Whoever needs the order (edge_in_read,
    // the set operations of split_path / remove_reads, the smallest id) calls sort_reads() first.
    uint32_t sorted_n = 0;
    SmallVec<read_t, kEdgeInlineReads> reads;
    void add_read(Arena &a, read_t r);        // Edge::addRead            (:24-28)
    void sort_reads();
};
static_assert(sizeof(Node) == 64 && sizeof(Edge) % 64 == 0, "graph objects are whole cache lines");
static_assert(kEdgeInlineReads != 8 || sizeof(Edge) == 64, "an edge with 8 inline read ids is one cache line");
inline bool OutRef::sink_base_is(char b) const { const unsigned c = code(); return c < 4 ? "ACGT"[c] == b : get()->sink->base == b; }
inline char OutRef::sink_base() const { const unsigned c = code(); return c < 4 ? "ACGT"[c] : get()->sink->base; }

template <class T>
class Pool {                                   // slab allocator with a free list; everything dies with the graph
public:
    ~Pool();
    template <class... A> T *make(A &&...a);
    void free(T *p);
    size_t live = 0;
private:
    std::vector<unsigned char *> slabs_;      // fixed-size slabs, recycled through a per-thread cache (SlabCache)
    std::vector<T *> free_;
    size_t used_in_last_ = 0;
    static constexpr size_t kPerSlab = kSlabBytes / sizeof(T);
};

// Slabs of finished graphs are kept per host thread and handed to the next graph built on that thread (contig
// builders are pinned to threads): no mmap/munmap and no first-touch page faults per contig.

// The seven per-"thread" streams (ConsensusGraphWriter, src/ConsensusGraph.cpp:118-133).  .id keeps
// the contig part (4-byte LE deltas, restarting at 0 per contig) apart from the lone-read ids, which
// the reference appends at the very end of the thread's file (src/Consensus.cpp:129).
struct StreamSet {
    std::string genome, lone, pos, type, base, complement, id_contigs;
    std::vector<read_t> lone_ids;
    std::vector<read_t> reads_in_contig;      // metaData numReadsInContig
    void append(const StreamSet &o);
    std::string id_bytes() const;             // contig deltas + lone deltas (src/ConsensusGraph.cpp:1018-1025)
};

void write_var_uint32(uint32_t v, std::string &out);   // src/DirectoryUtils.cpp:18-28

struct GraphRead { long pos; Node *start; size_t len; bool rc; };
struct ReadBases { const char *bases; size_t len; };   // a read as stored (forward orientation)

// The main path's edges: contiguous (index = position on the path, plain pointer arithmetic in the per-base loops),
// cheap at both ends (the path grows to the right read by read, and now and then to the left).
class EdgePath {
public:
    size_t size() const { return v_.size() - off_; }
    bool empty() const { return size() == 0; }
    Edge **begin() { return v_.data() + off_; }
    Edge **end() { return v_.data() + v_.size(); }
    Edge *const *begin() const { return v_.data() + off_; }
    Edge *const *end() const { return v_.data() + v_.size(); }
    Edge *&operator[](size_t i) { return v_[off_ + i]; }
    Edge *operator[](size_t i) const { return v_[off_ + i]; }
    Edge *front() const { return v_[off_]; }
    Edge *back() const { return v_.back(); }
    void push_back(Edge *e) { v_.push_back(e); }
    void push_front(Edge *e)
    {
        if (off_ == 0) { const size_t slack = size() / 2 + 64; v_.insert(v_.begin(), slack, nullptr); off_ = slack; }
        v_[--off_] = e;
    }
    void erase(Edge **first, Edge **last)
    {
        if (first == last) return;
        if (first == begin()) { off_ += (size_t)(last - first); if (off_ == v_.size()) v_.clear(), off_ = 0; return; }
        v_.erase(v_.begin() + (first - v_.data()), v_.begin() + (last - v_.data()));
    }
    template <class It> void append(It first, It last) { v_.insert(v_.end(), first, last); }
private:
    std::vector<Edge *> v_;
    size_t off_ = 0;
};

class ContigGraph {
public:
    ContigGraph() = default;
    ContigGraph(const ContigGraph &) = delete;
    ssize_t start_pos = 0, end_pos = 0;
    std::string main_path;                     // mainPath.path
    EdgePath main_edges;                       // mainPath.edges
    read_t first_read = 0;
    std::map<read_t, GraphRead> reads;         // readsInGraph (ascending id = output order)

    void initialize(const std::string &seed, read_t id, long pos);                         // :135-159
    void update_graph(const std::string &s, const std::vector<mm2::EditOp> &script, ssize_t begin_offset, ssize_t end_offset,
                      read_t id, long pos, bool rc);                                        // :400-557
    void calculate_main_path_greedy();                                                      // :559-615
    size_t num_reads() const { return reads.size(); }
    size_t num_edges() const { return n_edges_; }
    size_t num_nodes() const { return n_nodes_; }
    void write_main_path(StreamSet &o) const;                                               // :979-982
    // :984-1012; `source` (optional) hands out the stored bases of a read id: emission then walks each read guided by
    // its own bases instead of searching edge read lists (same result, see collect_path)
    void write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source = nullptr);
    void write_read_lone(StreamSet &o) const;                                               // :1014-1016
    // checker used by the tests (Consensus::checkRead, src/Consensus.cpp:342-368)
    bool read_string(read_t id, std::string &out) const;
    bool has_cycle() const;

private:
    Node *right_unchanged_ = nullptr, *left_unchanged_ = nullptr;
    size_t right_off_ = 0, left_off_ = 0;
    size_t n_nodes_ = 0, n_edges_ = 0;
    // number of side nodes (not on the main path) with more than one edge in: remove_cycles has work to do
    // only while this is non-zero (walk_and_prune acts on nothing else)
    size_t n_multi_in_side_ = 0;
    // tail re-use (exact shortcut of calculate_main_path_greedy): node indices > touch_idx_ were not modified by the
    // last update_graph; if in addition the previous remove_cycles re-routed nothing, the old path beyond the point
    // where the greedy walk re-joins it is what the walk would produce again
    size_t touch_idx_ = 0, touch_lo_ = (size_t)-1;
    std::vector<size_t> diverged_, cand_;     // main-path indices where the last update left the path / to re-check
    std::vector<Edge *> saved_;
    std::vector<uint32_t> op_at_;
    std::vector<const Node *> main_nodes_;    // the main path's nodes in order (filled by write_reads for the emission walks)
    std::vector<uint8_t> side_mask_;          // per main-path index: bases of the sinks of the out-edges other than the path's own
    std::vector<uint32_t> next_fork_;         // per main-path index: next node with more than one way out (or the path's end)
    std::vector<uint32_t> amb_off_;           // per main-path index j: amb_ids_[amb_off_[j] .. amb_off_[j + 1]) = the reads on side branches of node j that start with the
    std::vector<read_t> amb_ids_;             //   consensus's next base (kAmbComplex: too many to list -- look at the graph)
    static constexpr read_t kAmbComplex = ~(read_t)0;
    std::vector<uint8_t> follow_ok_;          // per main-path index j < n_main: 1 when a read whose next base is the consensus's next base can only go to main-path node j + 1
    bool have_touch_ = false;
    size_t consistent_from_ = (size_t)-1;     // main-path nodes with index >= this were chosen by best_out on the current counts
    uint64_t n_splits_ = 0;
    static bool multi_in_side(const Node *n) { return !n->on_main && n->in.size() > 1; }
    // every node that becomes a side node with in-degree > 1 is noted: remove_cycles then starts from these instead of
    // walking every side branch to find them (entries may be stale; they are re-checked, and the counter is the referee)
    std::vector<Node *> multi_in_list_;
    uint32_t scan_epoch_ = 0;                  // Node::reserved_ == scan_epoch_: marked by the current remove_cycles call
    void note_multi(Node *n, bool was) { if (!was && multi_in_side(n)) multi_in_list_.push_back(n); }
    void set_on_main(Node *n, bool v)
    {
        const bool was = multi_in_side(n);
        n_multi_in_side_ -= was; n->on_main = v; n_multi_in_side_ += multi_in_side(n);
        note_multi(n, was);
    }
    bool remove_cycles_from_list();            // false: the list does not account for every such node, do the full walk
    void walk_and_prune_marked(Edge *e, std::vector<Edge *> &stack);
public:
    uint64_t dbg_cycles_calls = 0, dbg_cycles_skipped = 0, dbg_spliced = 0, dbg_spliced_nodes = 0, dbg_walked_nodes = 0, dbg_cycles_idle = 0;
    // main_path[0 .. path_changed_from) has not changed since the caller last set this to SIZE_MAX (a lower bound of the common prefix:
    // calculate_main_path_greedy only ever lowers it; 0 = anything may have changed).  Lets the contig engine compare / re-code / re-sketch
    // the tail of a multi-megabase consensus instead of the whole string after every accepted read.
    size_t path_changed_from = 0;
    uint64_t dbg_cycles_listed = 0;            // remove_cycles calls served from multi_in_list_ (no walk over the side branches)
    double dbg_cycles_ms = 0;
private:
    Arena arena_;
    Pool<Node> nodes_;
    Pool<Edge> edges_;
    Node *create_node(char b);
    Edge *create_edge(Node *s, Node *t, read_t r);
    Edge *create_edge(Node *s, Node *t, const std::vector<read_t> &rs);
    void remove_reads_from_edge(Edge *e, const std::vector<read_t> &rs);
    void remove_edge(Edge *e, bool keep_in_source = false, bool keep_in_sink = false);
    void remove_node(Node *n);
    void clear_main_path();                                                                 // :617-651
    void remove_cycles();                                                                   // :653-691
    void walk_and_prune(Edge *e, std::vector<Edge *> &stack);                               // :693-714
    void split_path(Node *new_pre, Edge *e, const std::vector<read_t> &reads2split);        // :716-807
    template <class Visit, class VisitRun> void walk_read(const GraphRead &r, read_t id, const ReadBases *src, Visit visit, VisitRun visit_run) const;
    size_t read_to_edits(const GraphRead &r, read_t id, const ReadBases *src, std::vector<mm2::EditOp> &script, uint32_t &pos) const;   // :1031-1096
    size_t write_read(StreamSet &o, const GraphRead &r, read_t id, const ReadBases *src) const;   // :1098-1178
};

// Edit::optimizeEditScript (src/Edits.cpp:23-60): types 0 SAME 1 INSERT 2 DELETE 3 SUBSTITUTION
size_t optimize_edit_script(const std::vector<mm2::EditOp> &in, std::vector<mm2::EditOp> &out);

// Decoder of one stream set (Decompressor::decompress inner loop + generateRead,
// src/Decompressor.cpp:105-172, 252-314).  Returns (id, read) pairs in file order.
bool decode_streams(const StreamSet &s, std::vector<std::pair<read_t, std::string>> &out, std::string &err);
std::string meta_data(uint64_t n_reads, const std::vector<StreamSet> &threads);            // finishWriteConsensus, src/Consensus.cpp:370-386

void reverse_complement(const std::string &s, std::string &out);                            // include/ReadData.h:163-172
void reverse_complement(const char *s, size_t n, std::string &out);

}  // namespace cons
}  // namespace nsgpu
