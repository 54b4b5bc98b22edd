// consensus_oracle.cpp -- CPU restatement of NanoSpring's contig stage: Consensus (the greedy -t N loop),
// ConsensusGraph (alignRead's CIGAR -> Edit conversion, the per-base DAG, greedy main path, cycle pruning,
// edit emission, the seven streams + metaData) and the Decompressor's read generator.
//
// TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the GPU contig engine
// (nanospring_amd/csrc/consensus_driver.hip + consensus.cpp); nothing under nanospring_amd/ may include,
// link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.  It shares
// NO code with the product: it is deliberately naive (std::map of reads, std::vector edge lists with sorted
// read-id vectors, new/delete per node and edge, std::stack walks, no shortcut of any kind) and follows the
// reference statement by statement; every function cites the lines it restates (paths relative to the
// NanoSpring tree).
//
// PARITY UNPINNED against reference bytes: src/ConsensusGraph.cpp and src/Consensus.cpp need Boost, which
// this image lacks, so the reference's own objects cannot be compiled here (no stand-in headers are written).
// What IS pinned: the aligner answering alignRead is the reference's own minimap2 (oracle/_ref/libmm2ref.so,
// passed in as a function pointer), the candidate lists come from ns_oracle.c (pinned against the
// reference's ReadFilter/BBHashMap objects), and optimizeEditScript below is checked against vectors
// emitted by the reference's src/Edits.cpp object (tests/golden/edit_cases.npz, tests/test_edits.py).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cassert>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iterator>
#include <map>
#include <stack>
#include <stdexcept>
#include <string>
#include <vector>
#include <sys/types.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint32_t read_t;      // include/Types.h:6-10

extern "C" {
// ns_oracle.c (the MinHash half of the checker)
void oracle_sketch_reads(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, const uint64_t *salts, uint64_t *sketches);
void oracle_index_build(const uint64_t *sketches, uint32_t N, uint32_t n, uint64_t *keys, uint32_t *start, uint32_t *ids, uint32_t *nkeys);
uint64_t oracle_filter_string(const char *s, uint64_t len, uint32_t k, uint32_t N, uint32_t n, uint32_t thr, const uint64_t *salts,
                              const uint64_t *keys, const uint32_t *start, const uint32_t *ids, const uint32_t *nkeys, uint32_t *out,
                              uint64_t cap, uint64_t *n_matches);
}

// reg[0] of mm_map as oracle/ref_mm2_driver.c flattens it
typedef struct { int32_t hits, rs, re, qs, qe, blen, mlen, n_ambi, dp_max, dp_score, score, cnt, rev, mid_occ, n_cigar; } co_hit_t;
typedef int (*co_align_fn)(const char *ref, int rl, const char *qry, int ql, int k, int w, int max_chain_iter, co_hit_t *out, uint32_t *cigar, int cigar_cap);
// the anchors of a pair before chaining (oracle/ref_mm2_driver.c ref_mm_count_seeds): the lock-step schedule's rule for deferred alignments
typedef int64_t (*co_count_fn)(const char *ref, int rl, const char *qry, int ql, int k, int w);
static co_count_fn g_count_fn = nullptr;
static uint32_t g_defer_anchors = 0, g_defer_slots = 0;
extern "C" void cons_oracle_set_defer(void *count_fn, uint32_t anchors, uint32_t slots)
{
    g_count_fn = (co_count_fn)count_fn;
    g_defer_anchors = anchors, g_defer_slots = count_fn ? slots : 0;
}

namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------------------------------
// Edit (include/Edits.h:8-39; constructor src/Edits.cpp:6-21: one payload, a count for SAME, a char else)
// ---------------------------------------------------------------------------------------------------
enum EditType { SAME = 0, INSERT = 1, DELETE = 2, SUBSTITUTION = 3 };
struct Edit {
    EditType type;
    size_t num;     // SAME
    char ch;        // INSERT / DELETE / SUBSTITUTION
    Edit(EditType t, size_t v) : type(t), num(0), ch(0) { if (t == SAME) num = v; else ch = (char)v; }
};

// src/Edits.cpp:23-60 -- every maximal run of non-SAME edits becomes min(#ins,#del) substitutions carrying the first
// inserted chars, then the surplus inserts (remaining chars) or the surplus deletes ('-').  Returns the edit distance.
size_t optimizeEditScript(const std::vector<Edit> &in, std::vector<Edit> &out)
{
    size_t dis = 0;
    out.clear();
    size_t p = 0;
    const size_t end = in.size();
    while (p < end) {
        while (p < end && in[p].type == SAME) out.push_back(in[p++]);
        std::string ins;
        size_t ndel = 0;
        while (p < end && in[p].type != SAME) {
            if (in[p].type == INSERT) ins.push_back(in[p].ch); else ++ndel;
            ++p;
        }
        const size_t nins = ins.size();
        dis += std::max(ndel, nins);
        const size_t nsub = std::min(nins, ndel);
        size_t i;
        for (i = 0; i < nsub; ++i) out.push_back(Edit(SUBSTITUTION, ins[i]));
        if (nins > ndel) { for (; i < nins; ++i) out.push_back(Edit(INSERT, ins[i])); }
        else { for (; i < ndel; ++i) out.push_back(Edit(DELETE, '-')); }
    }
    return dis;
}

// include/Edits.h:73-94
void applyEdits(const char *orig, const std::vector<Edit> &es, std::string &res)
{
    for (const Edit &e : es) {
        switch (e.type) {
        case SAME: for (size_t i = 0; i < e.num; ++i) res.push_back(*orig++); break;
        case INSERT: res.push_back(e.ch); break;
        case DELETE: ++orig; break;
        case SUBSTITUTION: res.push_back(e.ch); ++orig; break;
        }
    }
}

// src/DirectoryUtils.cpp:18-28 -- 7 bits per byte, low group first, 0x80 = "more follows"
void write_var_uint32(uint32_t v, std::string &f)
{
    while (v > 127) { f.push_back((char)(uint8_t)((v & 0x7f) | 0x80)); v >>= 7; }
    f.push_back((char)(uint8_t)(v & 0x7f));
}
// src/DirectoryUtils.cpp:6-16
bool read_var_uint32(const std::string &f, size_t &p, uint32_t &val)
{
    val = 0;
    uint8_t byte, shift = 0;
    do {
        if (p >= f.size()) return false;
        byte = (uint8_t)f[p++];
        val |= (uint32_t)(byte & 0x7f) << shift;
        shift += 7;
    } while (byte & 0x80);
    return true;
}

// src/ReadData.cpp:247-260 + include/ReadData.h:163-172
char toComplement(char b) { switch (b) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return b; } }
std::string toReverseComplement(const std::string &s)
{
    std::string r;
    for (size_t i = s.size(); i-- > 0;) r.push_back(toComplement(s[i]));
    return r;
}

// ---------------------------------------------------------------------------------------------------
// Node / Edge / Path (include/ConsensusGraph.h:20-85, src/ConsensusGraph.cpp:13-116)
// ---------------------------------------------------------------------------------------------------
struct Edge;
struct Node {
    const char base;
    bool onMainPath = false;
    std::vector<Edge *> edgesOut, edgesIn;
    size_t cumulativeWeight = 0;
    Node *allPrev = nullptr, *allNext = nullptr;         // registry for the destructor only (the reference walks the graph instead, :809-823)
    explicit Node(char b) : base(b) {}
    Edge *getEdgeTo(Node *n);
    Edge *getEdgeToSide(char b);
    Edge *getBestEdgeOut();
    Edge *getBestEdgeIn();
    Edge *getEdgeInRead(read_t r) const;
    Node *getNextNodeInRead(read_t r) const;
};
struct Edge {
    Node *source, *sink;
    read_t count;
    std::vector<read_t> reads;          // ascending
    Edge *allPrev = nullptr, *allNext = nullptr;
    Edge(Node *a, Node *b, read_t r) : source(a), sink(b), count(1) { reads.push_back(r); }                       // :13-17
    Edge(Node *a, Node *b, const std::vector<read_t> &rs) : source(a), sink(b), reads(rs) { count = (read_t)reads.size(); }   // :19-22
    void addRead(read_t r) { ++count; reads.insert(std::lower_bound(reads.begin(), reads.end(), r), r); }        // :24-28
};
// :33-42 first out-edge whose sink is n
Edge *Node::getEdgeTo(Node *n) { for (Edge *e : edgesOut) if (e->sink == n) return e; return nullptr; }
// :44-52 first out-edge to a side node with that base
Edge *Node::getEdgeToSide(char b) { for (Edge *e : edgesOut) if (!e->sink->onMainPath && e->sink->base == b) return e; return nullptr; }
// :54-66 heaviest, the first one on ties
Edge *Node::getBestEdgeOut() { Edge *best = nullptr; read_t bc = 0; for (Edge *e : edgesOut) if (e->count > bc) { bc = e->count; best = e; } return best; }
// :68-80
Edge *Node::getBestEdgeIn() { Edge *best = nullptr; read_t bc = 0; for (Edge *e : edgesIn) if (e->count > bc) { bc = e->count; best = e; } return best; }
// :82-90
Edge *Node::getEdgeInRead(read_t r) const { for (Edge *e : edgesOut) if (std::binary_search(e->reads.begin(), e->reads.end(), r)) return e; return nullptr; }
// :92-95
Node *Node::getNextNodeInRead(read_t r) const { Edge *e = getEdgeInRead(r); return e ? e->sink : nullptr; }

struct Path { std::deque<Edge *> edges; std::string path; };

// the seven per-thread files of ConsensusGraphWriter (:118-133) as byte strings
struct Writer { std::string pos, type, base, id, complement, genome, lone; };

// cgw.idFile.write((char*)&diffId, std::ios::binary): the "size" is the enum's value, 4 in libstdc++ (:998, :1022)
void write_id(std::string &f, read_t diff) { f.append((const char *)&diff, 4); }

struct GraphRead { long pos; Node *start; size_t len; bool reverseComplement; };

class ConsensusGraph {
public:
    ssize_t startPos = 0, endPos = 0;
    Path mainPath;
    read_t firstReadId = 0;
    std::map<read_t, GraphRead> readsInGraph;

    ~ConsensusGraph() {
        while (allEdges) { Edge *e = allEdges; allEdges = e->allNext; delete e; }
        while (allNodes) { Node *n = allNodes; allNodes = n->allNext; delete n; }
    }

    // :135-159
    void initialize(const std::string &seed, read_t readId, long pos) {
        const size_t len = seed.length();
        Node *cur = createNode(seed[0]);
        readsInGraph.insert(std::make_pair(readId, GraphRead{pos, cur, seed.length(), false}));
        rightMostUnchangedNode = cur; rightMostUnchangedNodeOffset = 0;
        leftMostUnchangedNode = cur; leftMostUnchangedNodeOffset = 0;
        mainPath.path.push_back(cur->base);
        cur->onMainPath = true;
        cur->cumulativeWeight = 0;
        for (size_t i = 1; i < len; ++i) { Node *nx = createNode(seed[i]); createEdge(cur, nx, readId); cur = nx; }
        startPos = pos;
        endPos = pos + 1;
    }

    // :161-398 -- the minimap2 call is the callback (mm_tbuf_init .. mm_map, :195-217); the rest is the conversion
    bool alignRead(const std::string &s, std::vector<Edit> &editScript, ssize_t &relPos, ssize_t &beginOffset, ssize_t &endOffset,
                   size_t m_k, size_t m_w, size_t max_chain_iter, co_align_fn aligner) {
        const std::string &originalString = mainPath.path;
        co_hit_t r;
        std::vector<uint32_t> cigar(originalString.size() + s.size() + 8);
        aligner(originalString.c_str(), (int)originalString.size(), s.c_str(), (int)s.length(), (int)m_k, (int)m_w, (int)max_chain_iter, &r, cigar.data(),
                (int)cigar.size());
        return convertHit(r, cigar.data(), originalString, s, editScript, relPos, beginOffset, endOffset);
    }

    // :218-397 given reg[0]
    static bool convertHit(const co_hit_t &r, const uint32_t *cigar, const std::string &originalString, const std::string &s, std::vector<Edit> &editScript,
                           ssize_t &relPos, ssize_t &beginOffset, ssize_t &endOffset) {
        const char *Abegin = originalString.c_str();
        bool success = true;
        editScript.clear();
        if (r.hits > 0) {
            if (r.n_cigar < 0) throw std::runtime_error("reg[0] without CIGAR (assert(r->p), :224)");
            int qpos = r.qs, rpos = r.rs;
            const size_t editDis = (size_t)(r.blen - r.mlen + r.n_ambi);                                         // :236
            const int alignedLen = r.qe - r.qs;                                                                  // :238
            if (r.rs > 0 && r.re < (ssize_t)originalString.size()) {                                             // :240
                if (editDis / (double)alignedLen >= 1.0 || (double)alignedLen / s.length() <= 0.0) return false; // :245-257
            }
            relPos = (ssize_t)r.rs - (ssize_t)r.qs;                                                              // :284
            if (r.rs > 0) {                                                                                      // :287-299
                beginOffset = r.rs;
                for (int i = 0; i < r.qs; ++i) editScript.push_back(Edit(INSERT, s[i]));
            } else if (r.rs == 0) beginOffset = -r.qs;
            else throw std::runtime_error("Encountered invalid reference start");
            for (unsigned j = 0; j < (unsigned)r.n_cigar; ++j) {                                                 // :301-342
                const unsigned oplen = cigar[j] >> 4;
                switch ("MIDNSH"[cigar[j] & 0xf]) {
                case 'M': {
                    unsigned count_same = 0;
                    for (unsigned k = 0; k < oplen; ++k) {
                        if (s[qpos] == Abegin[rpos]) ++count_same;
                        else {
                            if (count_same > 0) editScript.push_back(Edit(SAME, count_same));
                            count_same = 0;
                            editScript.push_back(Edit(DELETE, Abegin[rpos]));
                            editScript.push_back(Edit(INSERT, s[qpos]));
                        }
                        ++qpos; ++rpos;
                    }
                    if (count_same != 0) editScript.push_back(Edit(SAME, count_same));
                    break;
                }
                case 'I': for (unsigned k = 0; k < oplen; ++k) { editScript.push_back(Edit(INSERT, s[qpos])); ++qpos; } break;
                case 'D': for (unsigned k = 0; k < oplen; ++k) { editScript.push_back(Edit(DELETE, Abegin[rpos])); ++rpos; } break;
                default: throw std::runtime_error("Encountered invalid CIGAR symbol!");
                }
            }
            if (r.re < (ssize_t)originalString.size()) {                                                         // :345-357
                endOffset = (ssize_t)r.re - (ssize_t)originalString.size();
                for (int i = r.qe; i < (ssize_t)s.length(); ++i) editScript.push_back(Edit(INSERT, s[i]));
            } else if (r.re == (ssize_t)originalString.size()) endOffset = (ssize_t)s.length() - r.qe;
            else throw std::runtime_error("Encountered invalid reference end");
        } else success = false;                                                                                  // :369-372
        if (!success) return false;
        size_t numUnchanged = 0;                                                                                 // :391-397
        for (const Edit &e : editScript) if (e.type == SAME) numUnchanged += e.num;
        return numUnchanged != 0;
    }

    // :400-557
    void updateGraph(const std::string &s, std::vector<Edit> &editScript, ssize_t beginOffset, ssize_t endOffset, read_t readId, long pos, bool reverseComplement) {
        size_t edgeInPath = 0;                               // an index into mainPath.edges stands for the deque iterator
        const size_t edgeInPathEnd = mainPath.edges.size();
        Node *nodeInPath = mainPath.edges[edgeInPath]->source;
        Node *currentNode = nullptr, *initialNode = nullptr;
        size_t numUnchanged = 0;

        if (beginOffset >= 0 || endOffset >= 0) {                                                                // :430-439
            rightMostUnchangedNodeOffset = static_cast<size_t>(std::max(static_cast<ssize_t>(leftMostUnchangedNodeOffset),
                                                                        std::min(static_cast<ssize_t>(rightMostUnchangedNodeOffset), beginOffset)));
            if (rightMostUnchangedNodeOffset > 0) rightMostUnchangedNode = mainPath.edges[rightMostUnchangedNodeOffset - 1]->sink;
            else rightMostUnchangedNode = mainPath.edges[0]->source;
        } else {                                                                                                 // :440-450 (size_t arithmetic, as there)
            leftMostUnchangedNodeOffset = std::min(rightMostUnchangedNodeOffset, std::max(leftMostUnchangedNodeOffset, mainPath.path.size() - 1 + endOffset));
            if (leftMostUnchangedNodeOffset > 0) leftMostUnchangedNode = mainPath.edges[leftMostUnchangedNodeOffset - 1]->sink;
            else leftMostUnchangedNode = mainPath.edges[0]->source;
        }

        auto advanceNodeInPath = [&]() {                                                                         // :452-457
            if (edgeInPath == edgeInPathEnd) return;
            nodeInPath = mainPath.edges[edgeInPath]->sink;
            ++edgeInPath;
        };

        if (beginOffset >= 1) {                                                                                  // :460-486
            edgeInPath += beginOffset - 1;
            nodeInPath = mainPath.edges[edgeInPath]->sink;
            ++edgeInPath;
        } else if (beginOffset <= -1) {
            const size_t numOfNodes2Insert = -beginOffset;
            size_t i = 0;
            currentNode = createNode(s[i++]);
            initialNode = currentNode;
            for (; i < numOfNodes2Insert; ++i) { Node *nx = createNode(s[i]); createEdge(currentNode, nx, readId); currentNode = nx; }
        }

        auto insertNode = [&](char base) {                                                                       // :490-505
            if (!currentNode) { currentNode = createNode(base); initialNode = currentNode; }
            else {
                Edge *edge = currentNode->getEdgeToSide(base);
                if (edge) edge->addRead(readId);
                else { Node *n = createNode(base); edge = createEdge(currentNode, n, readId); }
                currentNode = edge->sink;
            }
        };

        for (const Edit &e : editScript) {                                                                       // :508-544
            if (e.type == SAME) {
                const size_t num = e.num;
                numUnchanged += num;
                if (!currentNode) { initialNode = nodeInPath; currentNode = nodeInPath; }
                else {
                    Edge *edge = currentNode->getEdgeTo(nodeInPath);
                    if (edge) edge->addRead(readId);
                    else edge = createEdge(currentNode, nodeInPath, readId);
                    currentNode = nodeInPath;
                }
                advanceNodeInPath();
                for (size_t i = 1; i < num; ++i) {
                    currentNode->getEdgeTo(nodeInPath)->addRead(readId);
                    currentNode = nodeInPath;
                    advanceNodeInPath();
                }
            } else if (e.type == DELETE) advanceNodeInPath();
            else if (e.type == INSERT) insertNode(e.ch);
        }

        if (endOffset > 0)                                                                                       // :547-552
            for (size_t i = s.size() - (size_t)endOffset; i < s.size(); ++i) insertNode(s[i]);
        assert(numUnchanged > 0);
        (void)numUnchanged;
        readsInGraph.insert(std::make_pair(readId, GraphRead{pos, initialNode, s.length(), reverseComplement}));
    }

    // :559-615
    Path &calculateMainPathGreedy() {
        clearMainPath();
        std::deque<Edge *> &edgesInPath = mainPath.edges;
        std::string &stringPath = mainPath.path;
        {
            Node *cur = rightMostUnchangedNode;
            assert(cur->onMainPath);
            Edge *add;
            while ((add = cur->getBestEdgeOut())) { edgesInPath.push_back(add); cur = add->sink; cur->onMainPath = true; stringPath.push_back(cur->base); }
            const read_t endingReadId = *edgesInPath.back()->reads.begin();
            const GraphRead &endingRead = readsInGraph.at(endingReadId);
            endPos = endingRead.pos + (ssize_t)endingRead.len;
        }
        {
            Node *cur = leftMostUnchangedNode;
            assert(cur->onMainPath);
            Edge *add;
            while ((add = cur->getBestEdgeIn())) {
                edgesInPath.insert(edgesInPath.begin(), add);
                cur = add->source;
                cur->onMainPath = true;
                stringPath.insert(stringPath.begin(), cur->base);
                ++leftMostUnchangedNodeOffset;
                ++rightMostUnchangedNodeOffset;
            }
            const read_t startingReadId = *edgesInPath.front()->reads.begin();
            startPos = readsInGraph.at(startingReadId).pos;
        }
        removeCycles();
        rightMostUnchangedNode = edgesInPath.back()->sink;
        rightMostUnchangedNodeOffset = edgesInPath.size();
        leftMostUnchangedNode = edgesInPath.front()->source;
        leftMostUnchangedNodeOffset = 0;
        return mainPath;
    }

    void writeMainPath(Writer &w) { w.genome += mainPath.path; w.genome.push_back('\n'); }                       // :979-982
    void writeReadLone(Writer &w) { w.lone += mainPath.path; w.lone.push_back('\n'); }                           // :1014-1016
    static void writeIdsLone(Writer &w, const std::vector<read_t> &loneReads) {                                  // :1018-1025
        read_t pasId = 0;
        for (read_t it : loneReads) { write_id(w.id, it - pasId); pasId = it; }
    }

    // :984-1012
    void writeReads(Writer &w) {
        mainPath.edges.front()->source->cumulativeWeight = 0;
        size_t i = 0;
        for (Edge *e : mainPath.edges) e->sink->cumulativeWeight = ++i;
        read_t pasId = 0;
        for (auto &it : readsInGraph) {
            write_id(w.id, it.first - pasId);
            w.complement.push_back(it.second.reverseComplement ? 'c' : 'n');
            pasId = it.first;
            writeRead(w.pos, w.type, w.base, it.second, it.first);
        }
        w.complement.push_back('\n');
    }

    read_t getNumReads() const { return (read_t)readsInGraph.size(); }                                           // :1027
    size_t getNumEdges() const { return numEdges; }                                                              // :1029

    // :1031-1096
    size_t read2EditScript(GraphRead &r, read_t id, std::vector<Edit> &editScript, uint32_t &pos) {
        editScript.clear();
        Node *cur = r.start;
        assert(cur);
        bool intersectWithMainPath = true;
        while (!cur->onMainPath) {
            cur = cur->getNextNodeInRead(id);
            if (!cur) { intersectWithMainPath = false; break; }
        }
        if (!intersectWithMainPath) {
            pos = 0;
            size_t editDis = 0;
            Node *c = r.start;
            do { editScript.push_back(Edit(INSERT, c->base)); ++editDis; } while ((c = c->getNextNodeInRead(id)));
            return editDis;
        }
        pos = (uint32_t)cur->cumulativeWeight;
        size_t editDis = 0, posInMainPath = cur->cumulativeWeight, unchangedCount = 0;
        cur = r.start;
        auto dealWithUnchanged = [&]() { if (unchangedCount > 0) { editScript.push_back(Edit(SAME, unchangedCount)); unchangedCount = 0; } };
        do {
            if (cur->onMainPath) {
                const size_t curPos = cur->cumulativeWeight;
                if (curPos > posInMainPath) dealWithUnchanged();
                for (; posInMainPath < curPos; ++posInMainPath) { editScript.push_back(Edit(DELETE, '-')); ++editDis; }
                ++unchangedCount;
                ++posInMainPath;
            } else {
                dealWithUnchanged();
                editScript.push_back(Edit(INSERT, cur->base));
                ++editDis;
            }
        } while ((cur = cur->getNextNodeInRead(id)));
        dealWithUnchanged();
        return editDis;
    }

    // :1098-1178
    size_t writeRead(std::string &posFile, std::string &editTypeFile, std::string &editBaseFile, GraphRead &r, read_t id) {
        uint32_t offset;
        std::vector<Edit> editScript, newEditScript;
        size_t editDis = read2EditScript(r, id, editScript, offset);
        write_var_uint32(offset, posFile);
        editDis = optimizeEditScript(editScript, newEditScript);
        uint32_t numInsStart = 0, numInsEnd = 0;
        for (size_t i = 0; i != newEditScript.size(); ++i) {
            if (newEditScript[i].type != INSERT) break;
            ++numInsStart;
            editBaseFile.push_back(newEditScript[i].ch);
        }
        if (numInsStart != newEditScript.size())
            for (int64_t i = (int64_t)newEditScript.size() - 1; i >= 0; --i) { if (newEditScript[i].type != INSERT) break; ++numInsEnd; }
        write_var_uint32(numInsStart, posFile);
        uint32_t unchangedCount = 0;
        for (size_t i = numInsStart; i < newEditScript.size() - numInsEnd; ++i) {
            switch (newEditScript[i].type) {
            case SAME: unchangedCount += (uint32_t)newEditScript[i].num; break;
            case INSERT: write_var_uint32(unchangedCount, posFile); unchangedCount = 0; editTypeFile.push_back('i'); editBaseFile.push_back(newEditScript[i].ch); break;
            case DELETE: write_var_uint32(unchangedCount, posFile); unchangedCount = 0; editTypeFile.push_back('d'); break;
            case SUBSTITUTION: write_var_uint32(unchangedCount, posFile); unchangedCount = 0; editTypeFile.push_back('s'); editBaseFile.push_back(newEditScript[i].ch); break;
            }
        }
        write_var_uint32(unchangedCount, posFile);
        write_var_uint32(numInsEnd, posFile);
        for (size_t i = newEditScript.size() - numInsEnd; i != newEditScript.size(); ++i) editBaseFile.push_back(newEditScript[i].ch);
        editTypeFile.push_back('\n');
        return editDis;
    }

    // the template at include/ConsensusGraph.h:509-525 (used by Consensus::checkRead under -DCHECKS)
    bool getRead(read_t read, std::string &out) {
        auto it = readsInGraph.find(read);
        if (it == readsInGraph.end()) return false;
        Node *cur = it->second.start;
        while (cur) { out.push_back(cur->base); cur = cur->getNextNodeInRead(read); }
        return true;
    }

    // the property checkNoCycle (:1187-1250) asserts, tested by colouring (white/grey/black) from every registered node
    bool checkNoCycle() {
        for (Node *n = allNodes; n; n = n->allNext) n->cumulativeWeight = 0;
        for (Node *root = allNodes; root; root = root->allNext) {
            if (root->cumulativeWeight) continue;
            std::vector<std::pair<Node *, size_t>> st;
            st.push_back(std::make_pair(root, (size_t)0));
            root->cumulativeWeight = 1;
            while (!st.empty()) {
                Node *n = st.back().first;
                if (st.back().second < n->edgesOut.size()) {
                    Node *c = n->edgesOut[st.back().second++]->sink;
                    if (c->cumulativeWeight == 1) return false;
                    if (c->cumulativeWeight == 0) { c->cumulativeWeight = 1; st.push_back(std::make_pair(c, (size_t)0)); }
                } else { n->cumulativeWeight = 2; st.pop_back(); }
            }
        }
        return true;
    }

private:
    Node *rightMostUnchangedNode = nullptr;
    size_t rightMostUnchangedNodeOffset = 0;
    Node *leftMostUnchangedNode = nullptr;
    size_t leftMostUnchangedNodeOffset = 0;
    size_t numNodes = 0, numEdges = 0;
    Node *allNodes = nullptr;
    Edge *allEdges = nullptr;

    // :825-846
    Node *createNode(char base) {
        Node *n = new Node(base);
        n->allNext = allNodes; if (allNodes) allNodes->allPrev = n; allNodes = n;
        ++numNodes;
        return n;
    }
    Edge *link(Edge *e) {
        e->allNext = allEdges; if (allEdges) allEdges->allPrev = e; allEdges = e;
        e->source->edgesOut.push_back(e);
        e->sink->edgesIn.push_back(e);
        ++numEdges;
        return e;
    }
    Edge *createEdge(Node *source, Node *sink, read_t read) { return link(new Edge(source, sink, read)); }
    Edge *createEdge(Node *source, Node *sink, const std::vector<read_t> &reads) { return link(new Edge(source, sink, reads)); }

    // :848-859
    void removeReadsFromEdge(Edge *e, const std::vector<read_t> &reads) {
        std::vector<read_t> kept;
        std::set_difference(e->reads.begin(), e->reads.end(), reads.begin(), reads.end(), std::inserter(kept, kept.begin()));
        e->reads.swap(kept);
        e->count = (read_t)e->reads.size();
        if (e->count == 0) removeEdge(e);
    }
    // :861-876 -- out of the source's list goes the FIRST edge with the same sink (not necessarily e itself)
    void removeEdge(Edge *e, bool dontRemoveFromSource = false, bool dontRemoveFromSink = false) {
        if (!dontRemoveFromSource) {
            std::vector<Edge *> &v = e->source->edgesOut;
            v.erase(std::find_if(v.begin(), v.end(), [&](const Edge *p) { return p->sink == e->sink; }));
        }
        if (!dontRemoveFromSink) {
            std::vector<Edge *> &v = e->sink->edgesIn;
            v.erase(std::find(v.begin(), v.end(), e));
        }
        if (e->allPrev) e->allPrev->allNext = e->allNext; else allEdges = e->allNext;
        if (e->allNext) e->allNext->allPrev = e->allPrev;
        delete e;
        --numEdges;
    }
    // :878-897
    void removeNode(Node *n) {
        for (size_t i = 0; i < n->edgesIn.size(); ++i) removeEdge(n->edgesIn[i], false, true);
        for (size_t i = 0; i < n->edgesOut.size(); ++i) removeEdge(n->edgesOut[i], true, false);
        if (n->allPrev) n->allPrev->allNext = n->allNext; else allNodes = n->allNext;
        if (n->allNext) n->allNext->allPrev = n->allPrev;
        delete n;
        --numNodes;
    }

    // :617-651
    void clearMainPath() {
        const size_t l = mainPath.edges.size();
        for (size_t i = rightMostUnchangedNodeOffset; i < l; ++i) mainPath.edges[i]->sink->onMainPath = false;
        if (rightMostUnchangedNodeOffset < mainPath.edges.size()) mainPath.edges.erase(mainPath.edges.begin() + rightMostUnchangedNodeOffset, mainPath.edges.end());
        if (mainPath.path.size() > rightMostUnchangedNodeOffset + 1) mainPath.path.erase(mainPath.path.begin() + rightMostUnchangedNodeOffset + 1, mainPath.path.end());
        for (size_t i = 0; i < leftMostUnchangedNodeOffset; ++i) mainPath.edges[i]->source->onMainPath = false;
        if (leftMostUnchangedNodeOffset > 0) {
            mainPath.edges.erase(mainPath.edges.begin(), mainPath.edges.begin() + leftMostUnchangedNodeOffset);
            mainPath.path.erase(mainPath.path.begin(), mainPath.path.begin() + leftMostUnchangedNodeOffset);
            rightMostUnchangedNodeOffset -= leftMostUnchangedNodeOffset;
        }
        leftMostUnchangedNodeOffset = 0;
    }

    // :653-691
    void removeCycles() {
        size_t edgeOnPath = rightMostUnchangedNodeOffset;
        const size_t edgeOnPathEnd = mainPath.edges.size();
        Node *nodeOnPath = edgeOnPath < edgeOnPathEnd ? mainPath.edges[edgeOnPath]->source : mainPath.edges[edgeOnPath - 1]->sink;
        std::stack<Edge *> callStack;
        while (true) {
            const std::vector<Edge *> edgesOutCopy = nodeOnPath->edgesOut;
            for (Edge *e : edgesOutCopy) walkAndPrune(e, callStack);
            if (edgeOnPath == edgeOnPathEnd) break;
            nodeOnPath = mainPath.edges[edgeOnPath]->sink;
            ++edgeOnPath;
        }
        edgeOnPath = leftMostUnchangedNodeOffset;
        while (true) {
            nodeOnPath = mainPath.edges[edgeOnPath]->source;
            const std::vector<Edge *> edgesOutCopy = nodeOnPath->edgesOut;
            for (Edge *e : edgesOutCopy) walkAndPrune(e, callStack);
            if (edgeOnPath == 0) break;
            --edgeOnPath;
        }
    }

    // :693-714
    void walkAndPrune(Edge *e, std::stack<Edge *> &callStack) {
        callStack.push(e);
        while (!callStack.empty()) {
            Edge *curr = callStack.top();
            callStack.pop();
            Node *sink = curr->sink, *source = curr->source;
            if (sink->onMainPath) continue;
            if (sink->edgesIn.size() > 1) splitPath(source, curr, &curr->reads);
            for (auto it = sink->edgesOut.begin(); it != sink->edgesOut.end(); ++it) callStack.push(*it);
        }
    }

    // :716-807 -- the recursion as an explicit stack of contexts, each visited exactly twice
    struct SplitCtx {
        Node *newPre;
        Edge *e;
        std::vector<read_t> *reads2Split;       // borrowed until the first visit, owned afterwards
        bool hasVisited, owns;
        Node *oldCur;
    };
    void splitPath(Node *newPre, Edge *e, std::vector<read_t> *reads2Split) {
        std::deque<SplitCtx> callStack;          // std::stack<..> over a deque there: references stay valid across pushes
        callStack.push_back(SplitCtx{newPre, e, reads2Split, false, false, nullptr});
        while (!callStack.empty()) {
            SplitCtx &ctx = callStack.back();
            if (ctx.hasVisited) {                // second visit = the context's destructor (:749-753)
                std::vector<read_t> *mine = ctx.owns ? ctx.reads2Split : nullptr;
                Node *oc = ctx.oldCur;
                callStack.pop_back();
                delete mine;
                if (oc && oc->edgesIn.empty() && oc->edgesOut.empty()) removeNode(oc);
                continue;
            }
            std::vector<read_t> *readsInPath2Split = new std::vector<read_t>;
            std::set_intersection(ctx.reads2Split->begin(), ctx.reads2Split->end(), ctx.e->reads.begin(), ctx.e->reads.end(),
                                  std::inserter(*readsInPath2Split, readsInPath2Split->begin()));
            ctx.reads2Split = readsInPath2Split;
            ctx.owns = true;
            ctx.hasVisited = true;
            if (readsInPath2Split->empty()) continue;
            Node *oldCur = ctx.e->sink;
            ctx.oldCur = oldCur;
            removeReadsFromEdge(ctx.e, *readsInPath2Split);
            if (oldCur->onMainPath) { createEdge(ctx.newPre, oldCur, *readsInPath2Split); continue; }
            Node *newCur = createNode(oldCur->base);
            createEdge(ctx.newPre, newCur, *readsInPath2Split);
            for (Edge *it : oldCur->edgesOut) callStack.push_back(SplitCtx{newCur, it, readsInPath2Split, false, false, nullptr});
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// Consensus (src/Consensus.cpp)
// ---------------------------------------------------------------------------------------------------
struct CountStats { uint64_t countMinHash = 0, countMinHashNotInGraph = 0, countMergeSort = 0, countAligner = 0, alignCalls = 0, checkFail = 0; };

// src/Consensus.cpp:405-424
bool checkRepetitive(const std::string &readStr)
{
    const size_t readLen = readStr.length();
    for (size_t i = 1; i <= 6; ++i) {
        size_t countSameBase = 0;
        for (size_t j = 0; j < readLen; ++j) if (readStr[j] == readStr[(j + i) % readLen]) ++countSameBase;
        if (countSameBase > 0.7 * (double)readLen) return true;
    }
    return false;
}


// ---------------------------------------------------------------------------------------------------
// Lock-step virtual threads: the product's deterministic -t N schedule ("Contig stage"), restated as a scheduler AROUND
// the literal thread body below.  The reference's threads interact in exactly three places, all through timing: getRead (which thread
// finds which unclaimed read, src/Consensus.cpp:444-468), the claim after a successful alignRead (:256-277), and the moment a thread
// looks at inGraph[] (:204-208).  Here a logical clock ("slots") arbitrates them instead, so that the result is a function of the data:
//   * thread v belongs to group (v >> 3) % G, G = 4 (or 2; or all to group 0 with G = 1); its body runs only in slots s with s % G == its group
//     ("phase A" of the slot); the description below is for G = 4;
//   * a window query (the forward getFilteredReads of addRelatedReads) and an alignRead each take one period of 4 slots: the thread
//     continues in phase A of the slot 4 after the one in which it asked (with ONE group a window takes no slot: the thread goes on at once,
//     the slot's order is A, C, B, and the threads granted a seed in C go on together after the last grant);
//   * phase B of slot s resolves the claims of group (s + 1) & 3 -- the threads that asked for an alignment in slot s - 3 -- strictly in
//     thread order: the lowest thread wins a read (no try_lock ever fails);
//   * phase C of slot s grants seed reads to the threads of group s & 3 that are waiting in getRead, strictly in thread order, one
//     request per thread and slot; a granted thread runs on (createGraph, its first window) until it has to wait again.
// With one thread this is the reference's -t 1 run; with more it is one of the interleavings the reference's -t N can produce as long
// as seedHops == 0.  seedHops >= 1 replaces getRead's "first unclaimed read at or after the thread's cursor" by the conflict-aware rule
// of LockStep::pickSeed (SURVEY 8e "assign seed reads by MinHash bucket locality").
// ---------------------------------------------------------------------------------------------------
struct LockStep {
    enum Kind { RUN, SEED, SEEDED, WINDOW, ALIGN, DONE };
    struct VT {
        Kind kind = RUN;
        uint32_t at = 0;            // slot in which the thread started to wait
        uint32_t extra = 0;         // an alignment with a long anchor list takes so many MORE slots (cons_oracle_set_defer; include/nsgpu.h nsgpu_set_defer):
                                    // the thread goes on, and its claim is made, that much later -- the alignment itself is the same
        bool ok = false, won = false, go = false;
        read_t r = 0;
        std::condition_variable cv;
    };
    std::mutex m;
    std::condition_variable schedCv;
    std::vector<std::unique_ptr<VT>> vt;
    uint32_t slot = 0;
    int running = 0;
    // conflict-aware seeds: MinHash-locality buckets.  Reads are grouped once, before the first contig, into buckets of the
    // whole-read filter graph (read x -- read y when y is a filter result of x or of x's reverse complement): in id order every read
    // that has no bucket yet opens one and takes every bucketless read within bucketDepth hops of it (breadth first).  Two buckets are
    // adjacent when an edge of the graph joins them.  A contig in flight OCCUPIES the buckets of its seed and of every read it claimed.
    // A seed must lie in a bucket that is neither occupied nor within `rings` adjacency steps of an occupied one; the lowest unclaimed
    // read that qualifies is taken, by the waiting threads in thread order.  seedHops > 0 switches the rule on.
    int seedHops = 0;                            // bucketDepth
    int rings = 1, tailRings = 1, ringsNow = 1;      // tailRings: while more than half of ALL threads wait for a seed in one round
    std::vector<std::vector<read_t>> nbr;        // whole-read filter results of every read (forward and reverse-complement query), incl. itself
    std::vector<uint32_t> bucketOf;
    std::vector<std::vector<uint32_t>> adj;      // per bucket: adjacent buckets, ascending, without itself
    std::vector<uint32_t> occ;                   // per bucket: members of contigs in flight
    std::vector<std::vector<read_t>> members;    // per thread: seed + claimed reads of its contig in flight
    uint64_t nIdleSeedRounds = 0;
    uint32_t G = 4;                              // groups = slots per period (4: the pipelined engine; 1: every thread steps in every slot)
    int group(uint32_t v) const { return G == 1 ? 0 : (int)((v >> 3) % G); }

    // ---- thread side ----
    void park(uint32_t v, Kind k) {
        std::unique_lock<std::mutex> lk(m);
        VT &t = *vt[v];
        t.kind = k; t.at = slot;
        --running;
        schedCv.notify_one();
        t.cv.wait(lk, [&] { return t.go; });
        t.go = false;
    }
    void finish(uint32_t v) { std::unique_lock<std::mutex> lk(m); vt[v]->kind = DONE; --running; schedCv.notify_one(); }
    // ---- scheduler side (called with m held) ----
    void release(std::unique_lock<std::mutex> &, uint32_t v) { VT &t = *vt[v]; t.kind = RUN; t.go = true; ++running; t.cv.notify_one(); }
    void waitIdle(std::unique_lock<std::mutex> &lk) { schedCv.wait(lk, [&] { return running == 0; }); }

    void buildBuckets() {
        const uint32_t N = (uint32_t)nbr.size();
        bucketOf.assign(N, ~0u);
        uint32_t nb = 0;
        std::vector<read_t> cur, nxt;
        for (read_t r = 0; r < N; ++r) {
            if (bucketOf[r] != ~0u) continue;
            const uint32_t b = nb++;
            bucketOf[r] = b;
            cur.assign(1, r);
            for (int d = 0; d < seedHops && !cur.empty(); ++d) {
                nxt.clear();
                for (read_t x : cur) for (read_t y : nbr[x]) if (bucketOf[y] == ~0u) { bucketOf[y] = b; nxt.push_back(y); }
                cur.swap(nxt);
            }
        }
        adj.assign(nb, std::vector<uint32_t>());
        for (read_t x = 0; x < N; ++x) for (read_t y : nbr[x]) if (bucketOf[x] != bucketOf[y]) { adj[bucketOf[x]].push_back(bucketOf[y]); adj[bucketOf[y]].push_back(bucketOf[x]); }
        for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
        occ.assign(nb, 0);
    }
    void addMember(uint32_t v, read_t x) { members[v].push_back(x); ++occ[bucketOf[x]]; }
    void releaseContig(uint32_t v) { for (read_t x : members[v]) --occ[bucketOf[x]]; members[v].clear(); }
    // no occupied bucket within r adjacency steps of b (breadth first over the bucket graph, b itself is step 0)
    bool freeWithin(uint32_t b, int r) const {
        std::vector<uint32_t> cur(1, b), nxt;
        std::vector<uint8_t> seen(occ.size(), 0);
        seen[b] = 1;
        for (int d = 0;; ++d) {
            for (uint32_t x : cur) if (occ[x]) return false;
            if (d == r) return true;
            nxt.clear();
            for (uint32_t x : cur) for (uint32_t a : adj[x]) if (!seen[a]) { seen[a] = 1; nxt.push_back(a); }
            cur.swap(nxt);
        }
    }
    // The seed of thread v (phase C, threads in order; the contigs that the waiting threads finished have left the occupancy before the
    // first of them is served): the lowest unclaimed read whose bucket qualifies is taken and becomes the first member of the thread's new contig.  No such read although unclaimed reads exist:
    // the thread asks again at its group's next slot (returns 0).  No unclaimed read: the thread is done (-1).
    int pickSeed(uint32_t, const std::vector<uint8_t> &inGraph, read_t &out) {
        bool any = false;
        std::vector<int8_t> verdict(occ.size(), -1);                 // per bucket, for this request only
        for (read_t r = 0; r < (read_t)inGraph.size(); ++r) {
            if (inGraph[r]) continue;
            any = true;
            int8_t &v8 = verdict[bucketOf[r]];
            if (v8 < 0) v8 = freeWithin(bucketOf[r], ringsNow) ? 1 : 0;
            if (v8) { out = r; return 1; }
        }
        if (any) ++nIdleSeedRounds;
        return any ? 0 : -1;
    }
};

class Consensus {
public:
    const std::vector<std::string> &reads;     // what rD->getRead(r) returns: DnaBitset round trip = letters folded to ATCG
    size_t avgReadLen;
    uint32_t k, n, thr;
    const uint64_t *salts;
    size_t m_k, m_w, max_chain_iter, edge_threshold;
    int numThr;
    bool runChecks;
    co_align_fn aligner;
    read_t idBase;                            // ids written = local id + idBase (lets a shard of a larger read set be checked)
    read_t numReads;
    std::vector<uint8_t> inGraph, isRepetitive;
    static const uint32_t numLocks = 1u << 24;                                                                   // include/Consensus.h:102
    std::vector<std::atomic<uint8_t>> readStatusLock;
    // MinHashReadFilter state (ns_oracle.c)
    std::vector<uint64_t> keys;
    std::vector<uint32_t> start, ids, nkeys;
    std::vector<Writer> writers;
    std::vector<std::vector<read_t>> numReadsInContig, loneReads;
    std::vector<CountStats> countStats;
    LockStep *ls = nullptr;                   // lock-step virtual threads instead of OpenMP threads (see LockStep)

    Consensus(const std::vector<std::string> &rd) : reads(rd), readStatusLock(numLocks) {}

    static uint32_t &lsTid() { static thread_local uint32_t t = 0; return t; }
    bool try_lock(read_t r) { return readStatusLock[r % numLocks].exchange(1, std::memory_order_acquire) == 0; }
    void unlock(read_t r) { readStatusLock[r % numLocks].store(0, std::memory_order_release); }

    // :426-442
    void initialize() {
        numReads = (read_t)reads.size();
        inGraph.assign(numReads, 0);
        for (auto &l : readStatusLock) l.store(0);
        isRepetitive.assign(numReads, 0);
#pragma omp parallel for num_threads(numThr)
        for (read_t i = 0; i < numReads; ++i) isRepetitive[i] = checkRepetitive(reads[i]);
    }

    // :444-468
    bool getRead(read_t &read) {
        if (ls) {                                                     // wait for the thread's turn in phase C of one of its group's slots
            const uint32_t v = lsTid();
            for (;;) {
                ls->park(v, LockStep::SEED);
                if (ls->seedHops == 0) break;                         // the literal loop below, alone: no try_lock can fail
                const int got = ls->pickSeed(v, inGraph, read);
                if (got < 0) return false;
                if (got > 0) { inGraph[read] = 1; ls->addMember(v, read); if (ls->G == 1) ls->park(v, LockStep::SEEDED); return true; }
            }
            if (ls->G == 1) {                                         // (one group: the granted threads go on together, after the last grant)
                const bool got = getReadLiteral(read);
                if (got) ls->park(v, LockStep::SEEDED);
                return got;
            }
        }
        return getReadLiteral(read);
    }
    bool getReadLiteral(read_t &read) {
        if (read >= numReads) return false;
        while (read < numReads) {
            if (!inGraph[read]) {
                if (!try_lock(read)) ++read;
                else {
                    if (!inGraph[read]) { inGraph[read] = 1; unlock(read); return true; }
                    unlock(read);
                    ++read;
                }
            } else ++read;
        }
        return false;
    }

    // :388-403
    ConsensusGraph *createGraph(read_t &firstUnaddedRead) {
        read_t read = firstUnaddedRead;
        if (!getRead(read)) return nullptr;
        ConsensusGraph *cG = new ConsensusGraph;
        cG->mainPath.path = reads[read];
        cG->startPos = 0;
        cG->endPos = (ssize_t)cG->mainPath.path.size();
        cG->firstReadId = read;
        firstUnaddedRead = read + 1;
        return cG;
    }

    // ReadFilter::getFilteredReads(const std::string&, ..) (src/ReadFilter.cpp:85-97) through ns_oracle.c
    void getFilteredReads(const std::string &s, std::vector<read_t> &results) {
        results.resize(numReads + 1);
        uint64_t m;
        const uint64_t nc = oracle_filter_string(s.data(), s.size(), k, numReads, n, thr, salts, keys.data(), start.data(), ids.data(), nkeys.data(),
                                                 results.data(), numReads, &m);
        results.resize(nc);
    }

    // :342-368
    bool checkRead(ConsensusGraph *cG, read_t read) {
        std::string result;
        if (!cG->getRead(read + idBase, result)) return false;
        if (!cG->readsInGraph.at(read + idBase).reverseComplement) return result == reads[read];
        return toReverseComplement(result) == reads[read];
    }

    // :168-340
    void addRelatedReads(ConsensusGraph *cG, ssize_t curPos, int len, CountStats &cs) {
        const ssize_t offsetInMainPath = curPos - cG->startPos;
        if (len == 0 || offsetInMainPath < 0 || offsetInMainPath >= (ssize_t)cG->mainPath.path.size()) return;
        const size_t b = (size_t)offsetInMainPath;
        const size_t e = (ssize_t)cG->mainPath.path.size() >= offsetInMainPath + (ssize_t)len ? b + (size_t)len : cG->mainPath.path.size();
        const std::string originalString(cG->mainPath.path.begin() + b, cG->mainPath.path.begin() + e);
        const std::string reverseComplementString = toReverseComplement(originalString);
        const bool all[] = {false, true};
        for (bool reverseComplement : all) {
            std::vector<read_t> results;
            if (ls && !reverseComplement && ls->G != 1) ls->park(lsTid(), LockStep::WINDOW);      // one period per window (both strands are asked for at once); none with one group
            getFilteredReads(reverseComplement ? reverseComplementString : originalString, results);
            cs.countMinHash += results.size();
            for (const read_t r : results) {
                if (cG->getNumEdges() >= edge_threshold) return;                                                 // :200
                if (isRepetitive[r]) continue;
                if (inGraph[r]) continue;
                ++cs.countMinHashNotInGraph;
                const std::string &readStr1 = reads[r];
                if (readStr1.size() < 32) continue;
                const std::string readStr = reverseComplement ? toReverseComplement(readStr1) : readStr1;
                std::vector<Edit> editScript;
                ssize_t beginOffset = 0, endOffset = 0, pos = 0;
                ++cs.alignCalls;
                const bool alignStatus = cG->alignRead(readStr, editScript, pos, beginOffset, endOffset, m_k, m_w, max_chain_iter, aligner);
                if (ls) {                                                      // one period per alignment; the claim is phase B's, in thread order
                    LockStep::VT &t = *ls->vt[lsTid()];
                    t.ok = alignStatus, t.r = r, t.won = false;
                    t.extra = 0;
                    if (g_defer_slots && ls->G == 1) {
                        const std::string &ref = cG->mainPath.path;
                        if (g_count_fn(ref.c_str(), (int)ref.size(), readStr.c_str(), (int)readStr.size(), (int)m_k, (int)m_w) > (int64_t)g_defer_anchors) t.extra = g_defer_slots;
                    }
                    ls->park(lsTid(), LockStep::ALIGN);
                    if (!t.won) continue;
                } else {
                if (!alignStatus) continue;
                if (!try_lock(r)) continue;                                                                      // :256-277
                if (inGraph[r]) { unlock(r); continue; }
                inGraph[r] = 1;
                unlock(r);
                }
                ++cs.countAligner;
                if (runChecks) {                                                                                 // -DCHECKS :280-317
                    const std::string &mp = cG->mainPath.path;
                    const std::string origString(mp.begin() + (beginOffset > 0 ? beginOffset : 0), mp.end() + (endOffset > 0 ? 0 : endOffset));
                    const std::string targetString = readStr.substr(beginOffset > 0 ? 0 : -beginOffset,
                                                                    readStr.length() - (beginOffset > 0 ? 0 : -beginOffset) - (endOffset > 0 ? endOffset : 0));
                    std::string resultAfterEdit;
                    applyEdits(origString.c_str(), editScript, resultAfterEdit);
                    if (resultAfterEdit != targetString) ++cs.checkFail;
                }
                if (cG->getNumReads() == 0) {                                                                    // :319-324
                    const std::string mainPathString(cG->mainPath.path);
                    cG->mainPath.path.clear();
                    cG->initialize(mainPathString, cG->firstReadId + idBase, 0);
                    cG->calculateMainPathGreedy();
                }
                cG->updateGraph(readStr, editScript, beginOffset, endOffset, r + idBase, pos, reverseComplement);
                if (runChecks && !(checkRead(cG, r) && cG->checkNoCycle())) ++cs.checkFail;
                cG->calculateMainPathGreedy();
                if (runChecks && !(checkRead(cG, r) && cG->checkNoCycle())) ++cs.checkFail;
            }
        }
    }

    // :21-138 (the parallel region; finishWriteConsensus is meta_data() below)
    void generateAndWriteConsensus() {
        initialize();
        numReadsInContig.assign(numThr, std::vector<read_t>());
        loneReads.assign(numThr, std::vector<read_t>());
        countStats.assign(numThr, CountStats());
        writers.assign(numThr, Writer());
        if (ls) { runLockStep(); return; }
#pragma omp parallel num_threads(numThr)
        {
#ifdef _OPENMP
            const int tid = omp_get_thread_num();
#else
            const int tid = 0;
#endif
            threadBody(tid);
        }
    }

    // the body of the parallel region (:29-137)
    void threadBody(const int tid) {
        {
            Writer &cgw = writers[tid];
            ConsensusGraph *cG = nullptr;
            read_t firstUnaddedRead = 0;
            while ((cG = createGraph(firstUnaddedRead))) {
                const ssize_t initialStartPos = cG->startPos, initialEndPos = cG->endPos;
                const ssize_t len = initialEndPos - initialStartPos;
                const size_t offset = avgReadLen / 4;                                                            // :54
                ssize_t curPos = cG->startPos;
                bool edgesTooMany = false;
                while (len >= 32 && !isRepetitive[cG->firstReadId]) {                                            // :59-77
                    addRelatedReads(cG, curPos, (int)len, countStats[tid]);
                    curPos += offset;
                    if (curPos + len > cG->endPos) break;
                    else if (cG->getNumEdges() >= edge_threshold) { edgesTooMany = true; break; }
                }
                curPos = initialStartPos - (ssize_t)offset;
                while (len >= 32 && !edgesTooMany && !isRepetitive[cG->firstReadId]) {                           // :80-95
                    if (curPos < cG->startPos) break;
                    else if (cG->getNumEdges() >= edge_threshold) { edgesTooMany = true; break; }
                    addRelatedReads(cG, curPos, (int)len, countStats[tid]);
                    curPos -= offset;
                }
                if (cG->getNumReads() == 0) {                                                                    // :98-106
                    cG->writeReadLone(cgw);
                    loneReads[tid].push_back(cG->firstReadId + idBase);
                    numReadsInContig[tid].push_back(1);
                } else {
                    cG->writeMainPath(cgw);
                    cG->writeReads(cgw);
                    numReadsInContig[tid].push_back(cG->getNumReads());
                }
                delete cG;
            }
            ConsensusGraph::writeIdsLone(cgw, loneReads[tid]);                                                   // :129
        }
    }

    // the scheduler of the lock-step virtual threads (see LockStep)
    void runLockStep() {
        LockStep &L = *ls;
        const uint32_t T = (uint32_t)numThr;
        L.vt.clear();
        for (uint32_t v = 0; v < T; ++v) L.vt.emplace_back(new LockStep::VT());
        L.members.assign(T, std::vector<read_t>());
        if (L.seedHops) L.buildBuckets();
        L.slot = 0;
        std::vector<std::thread> th;
        std::unique_lock<std::mutex> lk(L.m);
        L.running = (int)T;
        for (uint32_t v = 0; v < T; ++v) th.emplace_back([this, v] { lsTid() = v; threadBody((int)v); ls->finish(v); });
        L.waitIdle(lk);                                                                 // everybody waits in getRead
        for (;; ++L.slot) {
            const int g = (int)(L.slot % L.G), b = (int)((L.slot + 1) % L.G);
            // phase A: the group's threads whose window or alignment has had its period run on, concurrently (they only READ inGraph)
            for (uint32_t v = 0; v < T; ++v) {
                LockStep::VT &t = *L.vt[v];
                if (L.group(v) == g && (t.kind == LockStep::WINDOW || t.kind == LockStep::ALIGN) && t.at + (t.kind == LockStep::ALIGN ? t.extra : 0u) < L.slot) L.release(lk, v);
            }
            L.waitIdle(lk);
            auto phaseB = [&] {     // claims of group b (asked for G - 1 slots ago), in thread order
                for (uint32_t v = 0; v < T; ++v) {
                    LockStep::VT &t = *L.vt[v];
                    if (L.group(v) != b || t.kind != LockStep::ALIGN || t.at + t.extra + L.G - 1 != L.slot || !t.ok) continue;
                    if (inGraph[t.r]) continue;
                    inGraph[t.r] = 1;
                    t.won = true;
                    if (L.seedHops) L.addMember(v, t.r);
                }
            };
            auto phaseC = [&] {     // seeds of group g, in thread order, one request per thread and slot
                std::vector<uint32_t> req;
                for (uint32_t v = 0; v < T; ++v) if (L.group(v) == g && L.vt[v]->kind == LockStep::SEED) req.push_back(v);
                if (L.seedHops) for (uint32_t v : req) L.releaseContig(v);
                L.ringsNow = 2 * req.size() > (size_t)T ? L.tailRings : L.rings;
                for (uint32_t v : req) { L.release(lk, v); L.waitIdle(lk); }
                // one group: the threads that got a seed take their first steps (graph of the seed read, first window, candidates up
                // to the first alignment) together, after the last grant
                bool any = false;
                for (uint32_t v : req) if (L.vt[v]->kind == LockStep::SEEDED) { L.release(lk, v); any = true; }
                if (any) L.waitIdle(lk);
            };
            // with one group the seeds come first: the engine grants them between its host phase and its GPU batches, so that a
            // fresh contig's first window is asked for in the same slot (a read granted as a seed cannot be claimed in that slot)
            if (L.G == 1) { phaseC(); phaseB(); } else { phaseB(); phaseC(); }
            bool all = true;
            for (uint32_t v = 0; v < T; ++v) all = all && L.vt[v]->kind == LockStep::DONE;
            if (all) break;
        }
        lk.unlock();
        for (std::thread &t : th) t.join();
    }

    // :370-386
    std::string meta_data() const {
        size_t size = 0;
        for (int i = 0; i < numThr; ++i) size += numReadsInContig[i].size();
        std::string m = "numReads=" + std::to_string(numReads) + "\n";
        m += "numContigs=" + std::to_string(size) + "\n";
        m += "numThr=" + std::to_string(numThr) + "\n";
        m += "numReadsInContig=";
        for (int i = 0; i < numThr; ++i) for (size_t j = 0; j < numReadsInContig[i].size(); ++j) m += std::to_string(numReadsInContig[i][j]) + ":";
        m += "\n";
        return m;
    }
};

// ---------------------------------------------------------------------------------------------------
// Decompressor: generateRead (src/Decompressor.cpp:252-314) and the per-thread loop around it (:105-172)
// ---------------------------------------------------------------------------------------------------
struct Cursor { const std::string &f; size_t p; bool get(char &c) { if (p >= f.size()) return false; c = f[p++]; return true; } };

bool generateRead(const std::string &genome, std::string &read, Cursor &posFile, Cursor &editTypeFile, Cursor &editBaseFile, bool reverseComplement)
{
    read.clear();
    uint32_t curPos, numInsStart, numInsEnd;
    if (!read_var_uint32(posFile.f, posFile.p, curPos)) return false;
    if (!read_var_uint32(posFile.f, posFile.p, numInsStart)) return false;
    for (size_t i = 0; i < numInsStart; ++i) { char b; if (!editBaseFile.get(b)) return false; read.push_back(b); }
    while (true) {
        uint32_t numUnchanged;
        if (!read_var_uint32(posFile.f, posFile.p, numUnchanged)) return false;
        for (size_t i = 0; i < numUnchanged; ++i) { if (curPos >= genome.size()) return false; read.push_back(genome[curPos++]); }
        char editType;
        if (!editTypeFile.get(editType)) return false;
        if (editType == '\n') break;
        if (editType == 'd') ++curPos;
        else if (editType == 'i') { char b; if (!editBaseFile.get(b)) return false; read.push_back(b); }
        else if (editType == 's') { ++curPos; char b; if (!editBaseFile.get(b)) return false; read.push_back(b); }
    }
    if (!read_var_uint32(posFile.f, posFile.p, numInsEnd)) return false;
    for (size_t i = 0; i < numInsEnd; ++i) { char b; if (!editBaseFile.get(b)) return false; read.push_back(b); }
    if (reverseComplement) read = toReverseComplement(read);
    return true;
}

bool getline_str(const std::string &f, size_t &p, std::string &line)
{
    if (p >= f.size()) return false;
    const size_t e = f.find('\n', p);
    if (e == std::string::npos) { line = f.substr(p); p = f.size(); }
    else { line = f.substr(p, e - p); p = e + 1; }
    return true;
}

// one thread's file set -> (id, read) in file order; false on a malformed set
bool decodeThread(const Writer &w, std::vector<std::pair<read_t, std::string>> &out)
{
    Cursor posFile{w.pos, 0}, typeFile{w.type, 0}, baseFile{w.base, 0}, complementFile{w.complement, 0};
    size_t gp = 0, ip = 0, lp = 0;
    std::string genome, cur;
    while (getline_str(w.genome, gp, genome)) {
        read_t id = 0;
        while (true) {
            char c;
            if (!complementFile.get(c)) return false;
            if (c == '\n') break;
            if (ip + 4 > w.id.size()) return false;
            read_t inc;
            memcpy(&inc, w.id.data() + ip, 4); ip += 4;
            id += inc;
            if (!generateRead(genome, cur, posFile, typeFile, baseFile, c == 'c')) return false;
            out.push_back(std::make_pair(id, cur));
        }
    }
    std::string lone;
    read_t id = 0;
    while (getline_str(w.lone, lp, lone)) {
        if (ip + 4 > w.id.size()) return false;
        read_t inc;
        memcpy(&inc, w.id.data() + ip, 4); ip += 4;
        id += inc;
        out.push_back(std::make_pair(id, lone));
    }
    return ip == w.id.size() && posFile.p == w.pos.size() && typeFile.p == w.type.size() && baseFile.p == w.base.size() && complementFile.p == w.complement.size();
}

uint8_t *dup_bytes(const std::string &s) { uint8_t *p = (uint8_t *)malloc(s.size() + 1); memcpy(p, s.data(), s.size()); p[s.size()] = 0; return p; }

void fold_reads(const char *bases, const uint64_t *off, uint32_t N, std::vector<std::string> &reads, uint64_t &total)
{
    static const char dna[4] = {'A', 'T', 'C', 'G'};      // DnaBitset round trip (src/dnaToBits.cpp:6-8, 81-98)
    reads.resize(N);
    total = 0;
    for (uint32_t r = 0; r < N; ++r) {
        reads[r].assign(bases + off[r], bases + off[r + 1]);
        for (char &c : reads[r]) c = dna[(c & 2) | ((c & 4) >> 2)];
        total += reads[r].size();
    }
}

}  // namespace

extern "C" {

typedef struct {
    uint64_t n_contigs, n_lone, count_minhash, count_minhash_not_in_graph, count_aligner, n_align_calls, n_bad_roundtrip, n_check_fail;
    double sketch_ms, consensus_ms;
} cons_oracle_stats;

// The reference's hot path (MinHashReadFilter::initialize + Consensus::generateAndWriteConsensus, src/Compressor.cpp:57-104) with
// num_thr OpenMP threads.  num_thr = 1 is deterministic; more threads race for reads exactly as the reference's do.
// streams_out / lens_out hold 7 * num_thr + 1 entries: per thread genome, lone, id, pos, type, base, complement; then metaData.
// Buffers are malloc'ed (free with cons_oracle_free).  Returns 0, or -1 with *err (static text) on an exception.
static int cons_oracle_run_impl(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, uint32_t thr, const uint64_t *salts, int m_k, int m_w,
                                int max_chain_iter, uint64_t edge_threshold, int num_thr, int run_checks, void *align_fn, uint32_t id_base, uint8_t **streams_out,
                                uint64_t *lens_out, cons_oracle_stats *st, int lock_step, int seed_hops, uint64_t *ls_out);

int cons_oracle_run(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, uint32_t thr, const uint64_t *salts, int m_k, int m_w,
                    int max_chain_iter, uint64_t edge_threshold, int num_thr, int run_checks, void *align_fn, uint32_t id_base, uint8_t **streams_out,
                    uint64_t *lens_out, cons_oracle_stats *st)
{
    return cons_oracle_run_impl(bases, off, N, k, n, thr, salts, m_k, m_w, max_chain_iter, edge_threshold, num_thr, run_checks, align_fn, id_base, streams_out, lens_out, st, 0, 0,
                                nullptr);
}

// The same with num_thr LOCK-STEP virtual threads (struct LockStep above: the product's deterministic schedule) and, for seed_hops >= 1,
// conflict-aware seeds.  ls_out[0] = slots of the run, ls_out[1] = seed requests that found every unclaimed read too close to a contig in flight.
int cons_oracle_run_lockstep(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, uint32_t thr, const uint64_t *salts, int m_k, int m_w,
                             int max_chain_iter, uint64_t edge_threshold, int num_thr, int run_checks, void *align_fn, uint32_t id_base, uint8_t **streams_out,
                             uint64_t *lens_out, cons_oracle_stats *st, int seed_hops, int groups, uint64_t *ls_out)
{
    if (groups != 1 && groups != 2 && groups != 4) return -3;
    return cons_oracle_run_impl(bases, off, N, k, n, thr, salts, m_k, m_w, max_chain_iter, edge_threshold, num_thr, run_checks, align_fn, id_base, streams_out, lens_out, st,
                                groups, seed_hops, ls_out);
}

static int cons_oracle_run_impl(const char *bases, const uint64_t *off, uint32_t N, uint32_t k, uint32_t n, uint32_t thr, const uint64_t *salts, int m_k, int m_w,
                                int max_chain_iter, uint64_t edge_threshold, int num_thr, int run_checks, void *align_fn, uint32_t id_base, uint8_t **streams_out,
                                uint64_t *lens_out, cons_oracle_stats *st, int lock_step, int seed_hops, uint64_t *ls_out)
{
    memset(st, 0, sizeof(*st));
    if (!align_fn || num_thr < 1) return -1;
    std::vector<std::string> reads;
    uint64_t total;
    fold_reads(bases, off, N, reads, total);
    Consensus c(reads);
    c.avgReadLen = N ? total / N : 0;                     // src/ReadData.cpp:207
    if (N && c.avgReadLen / 4 == 0) return -2;            // offset = avgReadLen / 4 = 0 (src/Consensus.cpp:54): the reference's window loop never ends
    c.k = k, c.n = n, c.thr = thr, c.salts = salts;
    c.m_k = (size_t)m_k, c.m_w = (size_t)m_w, c.max_chain_iter = (size_t)max_chain_iter, c.edge_threshold = (size_t)edge_threshold;
    c.numThr = num_thr, c.runChecks = run_checks != 0, c.aligner = (co_align_fn)align_fn, c.idBase = id_base;
    double t0 = now_ms();
    {   // rF.initialize(rD)
        std::string folded;
        std::vector<uint64_t> foff(N + 1);
        for (uint32_t r = 0; r < N; ++r) { foff[r] = folded.size(); folded += reads[r]; }
        foff[N] = folded.size();
        std::vector<uint64_t> sk((size_t)N * n);
        c.keys.resize((size_t)N * n); c.start.resize((size_t)(N + 1) * n); c.ids.resize((size_t)N * n); c.nkeys.resize(n);
#ifdef _OPENMP
        omp_set_num_threads(num_thr);
#endif
        oracle_sketch_reads(folded.data(), foff.data(), N, k, n, salts, sk.data());
        oracle_index_build(sk.data(), N, n, c.keys.data(), c.start.data(), c.ids.data(), c.nkeys.data());
    }
    st->sketch_ms = now_ms() - t0;
    LockStep L;
    if (lock_step) {
        c.ls = &L;
        L.G = (uint32_t)lock_step;                        // 1 or 4 groups
        L.seedHops = seed_hops & 255;                     // bucket depth; rings in the next byte (default 1)
        if ((seed_hops >> 8) & 255) L.rings = ((seed_hops >> 8) & 255) - 1;
        L.tailRings = (seed_hops >> 16) & 255 ? ((seed_hops >> 16) & 255) - 1 : L.rings;
        if (L.tailRings > L.rings) L.tailRings = L.rings;
        L.ringsNow = L.rings;
        if (seed_hops > 0) {                              // whole-read filter results of every read, both strands (what nsgpu_filter_all_reads holds)
            L.nbr.assign(N, std::vector<read_t>());
#pragma omp parallel for schedule(dynamic, 16)
            for (uint32_t r = 0; r < N; ++r) {
                std::vector<read_t> a, b;
                c.numReads = N;
                c.getFilteredReads(reads[r], a);
                c.getFilteredReads(toReverseComplement(reads[r]), b);
                a.insert(a.end(), b.begin(), b.end());
                L.nbr[r].swap(a);
            }
        }
    }
    t0 = now_ms();
    try { c.generateAndWriteConsensus(); } catch (const std::exception &) { return -1; }
    if (ls_out) { ls_out[0] = L.slot + 1; ls_out[1] = L.nIdleSeedRounds; }
    st->consensus_ms = now_ms() - t0;
    for (int t = 0; t < num_thr; ++t) {
        st->n_contigs += c.numReadsInContig[t].size();
        st->n_lone += c.loneReads[t].size();
        st->count_minhash += c.countStats[t].countMinHash;
        st->count_minhash_not_in_graph += c.countStats[t].countMinHashNotInGraph;
        st->count_aligner += c.countStats[t].countAligner;
        st->n_align_calls += c.countStats[t].alignCalls;
        st->n_check_fail += c.countStats[t].checkFail;
    }
    {   // the only property the reference itself tests (util/test_script.sh:7-9): decompress(compress(x)) == x
        std::vector<uint8_t> seen(N, 0);
        for (int t = 0; t < num_thr; ++t) {
            std::vector<std::pair<read_t, std::string>> rd;
            if (!decodeThread(c.writers[t], rd)) { st->n_bad_roundtrip += N + 1; continue; }
            for (auto &p : rd) {
                const read_t r = p.first - id_base;
                if (p.first < id_base || r >= N || seen[r] || p.second != reads[r]) ++st->n_bad_roundtrip; else seen[r] = 1;
            }
        }
        for (uint32_t r = 0; r < N; ++r) st->n_bad_roundtrip += !seen[r];
    }
    for (int t = 0; t < num_thr; ++t) {
        const Writer &w = c.writers[t];
        const std::string *parts[7] = {&w.genome, &w.lone, &w.id, &w.pos, &w.type, &w.base, &w.complement};
        for (int i = 0; i < 7; ++i) { streams_out[7 * t + i] = dup_bytes(*parts[i]); lens_out[7 * t + i] = parts[i]->size(); }
    }
    const std::string md = c.meta_data();
    streams_out[7 * num_thr] = dup_bytes(md);
    lens_out[7 * num_thr] = md.size();
    return 0;
}

void cons_oracle_free(void *p) { free(p); }

// alignRead's conversion alone (src/ConsensusGraph.cpp:218-397): reg[0] + CIGAR -> (ok, relPos, beginOffset, endOffset, edits).
// edits: type | char << 8 | num << 16.  Returns the number of edits, -1 when the reference would throw, -2 when edit_cap is too small.
int64_t cons_oracle_convert_hit(const co_hit_t *hit, const uint32_t *cigar, const char *ref, uint64_t rl, const char *qry, uint64_t ql, int32_t *ok,
                                int64_t *rel_pos, int64_t *begin_offset, int64_t *end_offset, uint64_t *edits, uint64_t edit_cap)
{
    const std::string R(ref, ref + rl), S(qry, qry + ql);
    std::vector<Edit> es;
    ssize_t rp = 0, bo = 0, eo = 0;
    try { *ok = ConsensusGraph::convertHit(*hit, cigar, R, S, es, rp, bo, eo) ? 1 : 0; } catch (const std::exception &) { return -1; }
    *rel_pos = rp, *begin_offset = bo, *end_offset = eo;
    if (es.size() > edit_cap) return -2;
    for (size_t i = 0; i < es.size(); ++i) edits[i] = (uint64_t)es[i].type | (uint64_t)(uint8_t)es[i].ch << 8 | (uint64_t)es[i].num << 16;
    return (int64_t)es.size();
}

// Edit::optimizeEditScript on a raw script (types 0 SAME 1 INSERT 2 DELETE): -> optimised script, *dis_out = edit distance
int64_t cons_oracle_optimize_edits(const uint8_t *types, const uint8_t *chars, const uint32_t *nums, uint32_t n, uint8_t *otypes, uint8_t *ochars,
                                   uint32_t *onums, uint32_t cap, uint64_t *dis_out)
{
    std::vector<Edit> in, out;
    for (uint32_t i = 0; i < n; ++i) in.push_back(types[i] == SAME ? Edit(SAME, nums[i]) : Edit((EditType)types[i], chars[i]));
    *dis_out = optimizeEditScript(in, out);
    if (out.size() > cap) return -1;
    for (size_t i = 0; i < out.size(); ++i) { otypes[i] = (uint8_t)out[i].type; ochars[i] = (uint8_t)out[i].ch; onums[i] = (uint32_t)out[i].num; }
    return (int64_t)out.size();
}

int cons_oracle_check_repetitive(const char *s, uint64_t len) { return checkRepetitive(std::string(s, s + len)) ? 1 : 0; }

// Decompressor's loop over ONE thread's file set (genome, lone, id, pos, type, base, complement).  ids_out[i] / reads
// concatenated into bases_out with off_out[i..i+1]; returns the number of reads, -1 malformed, -2 capacity.
int64_t cons_oracle_decode(const uint8_t *const *streams, const uint64_t *lens, uint32_t *ids_out, uint64_t *off_out, uint64_t read_cap, char *bases_out,
                           uint64_t base_cap)
{
    Writer w;
    std::string *parts[7] = {&w.genome, &w.lone, &w.id, &w.pos, &w.type, &w.base, &w.complement};
    for (int i = 0; i < 7; ++i) parts[i]->assign((const char *)streams[i], (const char *)streams[i] + lens[i]);
    std::vector<std::pair<read_t, std::string>> rd;
    if (!decodeThread(w, rd)) return -1;
    if (rd.size() > read_cap) return -2;
    uint64_t p = 0;
    for (size_t i = 0; i < rd.size(); ++i) {
        if (p + rd[i].second.size() > base_cap) return -2;
        ids_out[i] = rd[i].first;
        off_out[i] = p;
        memcpy(bases_out + p, rd[i].second.data(), rd[i].second.size());
        p += rd[i].second.size();
    }
    off_out[rd.size()] = p;
    return (int64_t)rd.size();
}

}
