// consensus_soa.hpp -- host side of the structure-of-arrays consensus DAG (dgraph.hpp): the arrays of one graph in host memory
// (SoaStore), the reference's ConsensusGraph interface over them with a team of one (SoaGraph: what the CPU test harness drives, and
// what the engine falls back to when it is told to keep the graphs on the host), and the edit emission of a finished contig
// (ConsensusGraph::writeReads / read2EditScript, src/ConsensusGraph.cpp:984-1178) from the arrays -- on the GPU path the arrays of a
// finished contig are copied back from HBM once and emitted from here.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <functional>
#include "consensus.hpp"
#include "dgraph.hpp"

namespace nsgpu {
namespace cons {

struct SoaRead { long pos; uint32_t start; size_t len; bool rc; };

// worst-case growth of one update with a script of n_ops ops of which n_run bases follow main-path edges
struct SoaNeed { uint32_t nodes, edges, chunks, path_side, wk; };
SoaNeed soa_need(uint32_t n_ops, uint32_t n_run, uint32_t n_ins, uint32_t path_len);

struct SoaStore {
    dg::Hdr hdr;
    std::vector<dg::Node> nodes;
    std::vector<dg::Edge> edges;
    std::vector<dg::Chunk> chunks;
    std::vector<uint32_t> mark, pidx, pe, pn, sv_e, sv_n, multi, wk;
    std::vector<uint8_t> ps, sv_s;
    SoaStore() { memset(&hdr, 0, sizeof(hdr)); }
    dg::G view();
    // room for a graph of these sizes; the path arrays are re-centred when a side runs short
    void reserve(uint32_t n_nodes, uint32_t n_edges, uint32_t n_chunks, uint32_t path_side, uint32_t wk_words);
    void ensure(const SoaNeed &need);
    size_t bytes() const;
};

// the aligner's script with the read's overhangs written out (dgraph.hpp update): returns the op words; n_run / n_ins for soa_need
void soa_script(const std::string &s, const std::vector<mm2::EditOp> &script, ssize_t begin_offset, ssize_t end_offset, std::vector<uint32_t> &ops, uint32_t &n_run, uint32_t &n_ins);

// edit emission from the arrays of a finished graph (read-only)
class SoaEmitter {
public:
    SoaEmitter(const dg::G &g, const std::map<read_t, SoaRead> &reads) : g_(g), reads_(reads) {}
    void write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source);
    bool read_string(read_t id, std::string &out) const;
private:
    dg::G g_;
    const std::map<read_t, SoaRead> &reads_;
    std::vector<uint32_t> main_idx_, next_fork_, amb_off_;
    std::vector<read_t> amb_ids_;
    std::vector<uint8_t> side_mask_, follow_ok_;
    static constexpr read_t kAmbComplex = ~(read_t)0;
    uint32_t edge_in_read(uint32_t n, read_t id) const;
    uint32_t way_out_of(uint32_t n, char nb, read_t id) const;
    template <class Visit, class VisitRun> void walk_read(const SoaRead &r, read_t id, const ReadBases *src, Visit visit, VisitRun visit_run) const;
    size_t read_to_edits(const SoaRead &r, read_t id, const ReadBases *src, std::vector<mm2::EditOp> &script, uint32_t &pos) const;
    size_t write_read(StreamSet &o, const SoaRead &r, read_t id, const ReadBases *src) const;
};

class SoaGraph {
public:
    SoaGraph() = default;
    SoaGraph(const SoaGraph &) = delete;
    ssize_t start_pos = 0, end_pos = 0;
    std::string main_path;
    read_t first_read = 0;
    std::map<read_t, SoaRead> reads;
    size_t path_changed_from = 0;
    int dbg_flags_override = -1;                      // tests: this graph's dgraph.hpp debug flags (else NSGPU_SOA_DEBUG_FLAGS)

    void initialize(const std::string &seed, read_t id, long pos);
    void update_graph(const std::string &s, const std::vector<mm2::EditOp> &script, ssize_t begin_offset, ssize_t end_offset, read_t id, long pos, bool rc);
    void calculate_main_path_greedy();
    size_t num_reads() const { return reads.size(); }
    size_t num_edges() const { return st_.hdr.live_edges; }
    size_t num_nodes() const { return st_.hdr.live_nodes; }
    void write_main_path(StreamSet &o) const { o.genome += main_path; o.genome.push_back('\n'); }
    void write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source = nullptr);
    void write_read_lone(StreamSet &o) const { o.lone += main_path; o.lone.push_back('\n'); }
    bool read_string(read_t id, std::string &out);
    bool has_cycle();
    SoaStore &store() { return st_; }
    uint32_t last_error() const { return st_.hdr.err; }
private:
    SoaStore st_;
    bool fresh_ = false;          // initialize() has run and nothing since (its path is whole: the first recompute has nothing to do)
    bool pending_ = false;        // an update whose recompute has not run yet
};

// the consensus after an update, from the string before and what the graph reports (P bases kept in front, S at the end, the new middle)
void soa_patch_path(std::string &path, uint32_t P, uint32_t S, uint32_t new_len, const uint8_t *new_bases /* the new path's bases from index P on */);

}  // namespace cons
}  // namespace nsgpu
