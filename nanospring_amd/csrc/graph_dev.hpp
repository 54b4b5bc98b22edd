// graph_dev.hpp -- a contig's consensus DAG as the contig engine sees it.
//   GraphBase    what the engine needs of a graph: the consensus string, the contig's span, an accepted read handed in (submit), polled
//                (ready) and taken over (complete), and the emission of the finished contig.
//   HostGraph    the pointer graph on the host (consensus.cpp), everything done inside submit: the A/B partner (NSGPU_GRAPH=host).
//   DevGraph     the structure-of-arrays graph in HBM (dgraph.hpp), one workgroup per update on a stream of its own; the host keeps the
//                consensus string, patched from what the kernel reports (P bases kept in front, S at the end, the new middle), and the
//                read table.  A finished contig's arrays are copied back once and emitted from there (consensus_soa.cpp).
#pragma once
#include "common.hpp"
#include "consensus_soa.hpp"
#include <memory>
#include <atomic>

namespace nsgpu {

using cons::read_t;

// Power-of-two blocks out of large slabs (HBM: hipMalloc; host: pinned), handed back to per-class free lists: no hipMalloc / hipFree
// -- which synchronises the device -- while the engine runs.
class SlabPool {
public:
    explicit SlabPool(bool pinned_host) : host_(pinned_host) {}
    ~SlabPool();
    SlabPool(const SlabPool &) = delete;
    void *alloc(size_t bytes, size_t *granted);       // nullptr: out of memory (error set)
    void free(void *p, size_t granted);
    void set_slab_bytes(size_t b) { slab_bytes_ = b; }
    size_t in_use() const { return in_use_; }
    size_t peak() const { return peak_; }
    size_t mapped() const { return mapped_; }
private:
    bool host_;
    std::mutex m_;
    std::vector<void *> slabs_;
    std::vector<std::pair<char *, size_t>> tails_;    // unused rests of slabs: (pointer, bytes)
    std::vector<std::pair<size_t, std::vector<void *>>> free_sz_;     // free blocks by exact size
    size_t slab_bytes_ = (size_t)1 << 30, in_use_ = 0, peak_ = 0, mapped_ = 0;
};

struct DgResult;     // what an update kernel writes into pinned memory (dgraph_dev.hip)
class DevGraph;

struct GraphBase {
    read_t first_read = 0;
    size_t path_changed_from = 0;
    virtual ~GraphBase() {}
    virtual const std::string &path() const = 0;
    virtual std::string &path_mut() = 0;              // (a graph without reads: the seed read is its consensus)
    virtual ssize_t start_pos() const = 0;
    virtual ssize_t end_pos() const = 0;
    virtual void set_span(ssize_t s, ssize_t e) = 0;
    virtual size_t num_reads() const = 0;
    virtual size_t num_edges() = 0;
    // the accepted read (src/Consensus.cpp:319-331: initialize on the contig's first, updateGraph, calculateMainPathGreedy)
    virtual int submit(const std::string &query, const mm2::AlnOut &aln, read_t id, bool rc) = 0;
    virtual bool ready() = 0;
    virtual int complete() = 0;
    virtual DevGraph *dev() { return nullptr; }       // the graph in HBM, if it is one
    // the finished contig
    virtual int emit_begin() { return NSGPU_OK; }     // (the device graph: its arrays on the way back to the host)
    virtual void write_main_path(cons::StreamSet &o) const { o.genome += path(); o.genome.push_back('\n'); }
    virtual void write_read_lone(cons::StreamSet &o) const { o.lone += path(); o.lone.push_back('\n'); }
    virtual void write_reads(cons::StreamSet &o, const std::function<cons::ReadBases(cons::read_t)> *source) = 0;
    uint64_t dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // cycles calls / skipped / detours / idle / walked / listed / splits / one-at-a-time excursions
    double dbg_cycles_ms = 0;
};

struct HostGraph final : GraphBase {
    cons::ContigGraph g;
    const std::string &path() const override { return g.main_path; }
    std::string &path_mut() override { return g.main_path; }
    ssize_t start_pos() const override { return g.start_pos; }
    ssize_t end_pos() const override { return g.end_pos; }
    void set_span(ssize_t s, ssize_t e) override { g.start_pos = s, g.end_pos = e; }
    size_t num_reads() const override { return g.num_reads(); }
    size_t num_edges() override { return g.num_edges(); }
    int submit(const std::string &query, const mm2::AlnOut &aln, read_t id, bool rc) override;
    bool ready() override { return true; }
    int complete() override { return NSGPU_OK; }
    void write_reads(cons::StreamSet &o, const std::function<cons::ReadBases(cons::read_t)> *source) override;
};

// what all device graphs of an engine share
struct DevGraphShared {
    SlabPool dev{false}, pin{true};
    hipStream_t serve_stream = nullptr;               // the serve launches and the array copies in front of them, in order
    hipStream_t copy_stream = nullptr;                // finished contigs on their way back
    struct Launch { void *ptrs = nullptr; size_t granted = 0; hipEvent_t done = nullptr; };
    std::mutex serve_m;
    std::vector<Launch> launches;                     // serve launches whose kernel may still be running (their slot lists stay until it is gone)
    std::vector<hipEvent_t> free_events;
    std::atomic<uint64_t> n_launches{0};
    std::atomic<bool> aborted{false};                 // a wait for a report has run out: nobody waits any longer
    std::atomic<uint32_t> next_ticket{1};             // of prepare(): what binds a workgroup to the order it was launched for
    uint32_t max_ops = 0;                             // longest script the staging buffers take: 2 * longest read + slack
    bool check = false;                               // NSGPU_GRAPH_CHECK: every update also on the host, arrays compared
    uint32_t dbg_flags = 0;
    std::atomic<uint64_t> n_updates{0}, n_grow{0}, n_mid_copies{0}, kernel_wait_ns{0}, bytes_back{0}, update_ns{0}, final_wait_ns{0};
    std::atomic<uint64_t> n_seq_updates{0}, n_full_walks{0}, n_splits{0}, n_regrow{0};
    uint64_t edge_thr = ~0ull;                        // --edge-thr: num_edges() must be exact near it
    std::atomic<uint64_t> phase_ticks[8], hist[8], slow_phase[8], cnt[6], cyc[6], rt[4];             // the kernels' own clock (100 MHz) by phase, summed (debug report)
    ~DevGraphShared();
};

class DevGraph final : public GraphBase {
public:
    DevGraph(DevGraphShared *sh, uint32_t builder);
    ~DevGraph() override;
    const std::string &path() const override { return path_; }
    std::string &path_mut() override { return path_; }
    ssize_t start_pos() const override { return start_; }
    ssize_t end_pos() const override { return end_; }
    void set_span(ssize_t s, ssize_t e) override { start_ = s, end_ = e; }
    size_t num_reads() const override { return reads_.size(); }
    size_t num_edges() override;
    int submit(const std::string &query, const mm2::AlnOut &aln, read_t id, bool rc) override;
    bool ready() override;
    int complete() override;
    DevGraph *dev() override { return this; }
    int emit_begin() override;
    // Serve launches: prepare() the graphs that may get a read in a slot (room for any script of a read that long), graph_serve_launch() them in
    // ONE kernel whose workgroups wait for their orders; submit() then only hands the script over; cancel() releases a workgroup that gets none.
    int prepare(size_t read_len);
    void cancel();
    bool armed() const { return armed_; }
    friend int graph_serve_launch(DevGraphShared *sh, DevGraph *const *graphs, size_t n);
    void write_reads(cons::StreamSet &o, const std::function<cons::ReadBases(cons::read_t)> *source) override;
    struct Block { void *p = nullptr; size_t granted = 0, want = 0; };
private:
    DevGraphShared *sh_;
    std::string path_;
    ssize_t start_ = 0, end_ = 0;
    std::map<read_t, cons::SoaRead> reads_;
    dg::Hdr hdr_;                                     // as of the last completed update
    // arrays in HBM: capacities in entries
    Block b_nodes_, b_mark_, b_pidx_, b_edges_, b_chunks_, b_pe_, b_pn_, b_ps_, b_sve_, b_svn_, b_svs_, b_multi_, b_wk_, b_hdr_;
    uint32_t cap_nodes_ = 0, cap_edges_ = 0, cap_chunks_ = 0, cap_path_ = 0, cap_wk_ = 0, cap_multi_ = 0;
    uint32_t path_off_ = 0;                           // where the host believes the path lies (re-centring moves it)
    std::vector<Block> retired_;                      // replaced arrays: back to the pool when the update that followed the copy is done
    Block pin_;                                       // staging: script, seed, result
    uint32_t epoch_ = 0;
    bool inited_ = false, pending_ = false, prepared_ = false, armed_ = false, moved_path_ = false;
    bool finalizing_ = false;                         // the update's first report is in (the consensus), its second (after removeCycles) not yet
    int fail_rc_ = NSGPU_OK;                          // a failure found where no error could be returned: returned by the next call that can
    uint32_t edges_seen_ = 0, seen3_[3] = {0, 0, 0};
    static constexpr uint32_t kEdgeMargin = 1u << 20;
    int finalize(bool wait);
    int kernel_error();
    uint32_t ops_cap_ = 0;
    uint32_t tm_seen_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt_seen_[6] = {0, 0, 0, 0, 0, 0}, cyc_seen_[6] = {0, 0, 0, 0, 0, 0}, rt_seen_[4] = {0, 0, 0, 0}, dbg_seen_[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long p_begin_ = 0, p_end_ = 0;
    // pending update
    read_t p_id_ = 0; long p_pos_ = 0; size_t p_len_ = 0; bool p_rc_ = false; double p_t0_ = 0;
    // the finished contig on the host
    Block e_nodes_, e_edges_, e_chunks_, e_pe_, e_pn_, e_ps_;
    hipEvent_t e_ev_ = nullptr;
    bool e_begun_ = false;
    // NSGPU_GRAPH_CHECK
    std::unique_ptr<cons::SoaGraph> shadow_;
    std::unique_ptr<cons::ContigGraph> ptr_shadow_;
    dg::G view() const;
    int grow(const cons::SoaNeed &need, uint32_t seed_len, hipStream_t st = nullptr);
    int take(Block &b, size_t bytes, bool pinned = false);
    void give(Block &b, bool pinned = false);
    void retire(Block &b) { if (b.p) retired_.push_back(b); b = Block(); }
    int check_against_shadow(const char *where);
};

int graph_serve_launch(DevGraphShared *sh, DevGraph *const *graphs, size_t n);


}  // namespace nsgpu
