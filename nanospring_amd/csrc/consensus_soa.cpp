// consensus_soa.cpp -- see consensus_soa.hpp.
#include "consensus_soa.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace nsgpu {
namespace cons {

using mm2::EditOp;
typedef dg::Ops<dg::HostTeam> HostOps;

// ---------------------------------------------------------------------------
// arrays
// ---------------------------------------------------------------------------
SoaNeed soa_need(uint32_t n_ops, uint32_t n_run, uint32_t n_ins, uint32_t path_len)
{
    SoaNeed n;
    n.nodes = n_ins + 64;
    n.edges = n_ins + n_ops + 64;
    n.chunks = n_run + 3 * n_ops + 256;
    n.path_side = n_ins + 64;
    (void)path_len;
    n.wk = HostOps::wk_update_words(n_ops) + (1u << 22);
    return n;
}

dg::G SoaStore::view()
{
    dg::G g;
    g.h = &hdr;
    g.nodes = nodes.data(), g.edges = edges.data(), g.chunks = chunks.data(), g.mark = mark.data(), g.pidx = pidx.data();
    g.pe = pe.data(), g.pn = pn.data(), g.ps = ps.data();
    g.sv_e = sv_e.data(), g.sv_n = sv_n.data(), g.sv_s = sv_s.data();
    g.multi_list = multi.data(), g.wk = wk.data();
    return g;
}

size_t SoaStore::bytes() const
{
    return nodes.capacity() * sizeof(dg::Node) + edges.capacity() * sizeof(dg::Edge) + chunks.capacity() * sizeof(dg::Chunk) +
           (mark.capacity() + pidx.capacity() + pe.capacity() + pn.capacity() + sv_e.capacity() + sv_n.capacity() + multi.capacity() + wk.capacity()) * 4 + ps.capacity() + sv_s.capacity();
}

void SoaStore::reserve(uint32_t n_nodes, uint32_t n_edges, uint32_t n_chunks, uint32_t path_side, uint32_t wk_words)
{
    static const bool exact = getenv("NSGPU_SOA_SLACK") != nullptr;      // (tests of ERR_ROOM: the arrays as large as asked, no more)
    auto grow = [](size_t have, size_t want) { if (exact) return want > have ? want : have; size_t c = have ? have : 1024; while (c < want) c += c / 2 + 1024; return c; };
    // (tests: new entries full of a pattern instead of zeros -- NSGPU_SOA_POISON = the 32-bit word, a bit set per array in NSGPU_SOA_POISON_ARRAYS:
    // 1 nodes, 2 edges, 4 chunks, 8 mark, 16 pidx, 32 the path arrays, 64 their staging, 128 the multi list, 256 the work area.  The arrays in
    // HBM come out of a pool that does not clear them)
    static const char *pz = getenv("NSGPU_SOA_POISON");
    static const uint32_t pw = pz ? (uint32_t)strtoul(pz, nullptr, 0) : 0;
    static const uint32_t pa = getenv("NSGPU_SOA_POISON_ARRAYS") ? (uint32_t)strtoul(getenv("NSGPU_SOA_POISON_ARRAYS"), nullptr, 0) : 0xffffffffu;
    auto fill = [&](void *p, size_t from_bytes, size_t to_bytes, uint32_t bit) { if (!pz || !(pa & bit)) return; uint32_t *q = static_cast<uint32_t *>(p); for (size_t i = from_bytes / 4; i < to_bytes / 4; ++i) q[i] = pw; };
    if (nodes.size() < n_nodes) {
        const size_t c0 = nodes.size(), c = grow(nodes.size(), n_nodes);
        nodes.resize(c); mark.resize(c); pidx.resize(c);
        fill(nodes.data(), c0 * sizeof(dg::Node), c * sizeof(dg::Node), 1), fill(mark.data(), c0 * 4, c * 4, 8), fill(pidx.data(), c0 * 4, c * 4, 16);
    }
    if (edges.size() < n_edges) { const size_t c0 = edges.size(); edges.resize(grow(edges.size(), n_edges)); fill(edges.data(), c0 * sizeof(dg::Edge), edges.size() * sizeof(dg::Edge), 2); }
    if (chunks.size() < n_chunks) { const size_t c0 = chunks.size(); chunks.resize(grow(chunks.size(), n_chunks)); fill(chunks.data(), c0 * sizeof(dg::Chunk), chunks.size() * sizeof(dg::Chunk), 4); }
    if (wk.size() < wk_words) { const size_t c0 = wk.size(); wk.resize(grow(wk.size(), wk_words)); fill(wk.data(), c0 * 4, wk.size() * 4, 256); }
    if (multi.size() < 4096) { multi.resize(4096); fill(multi.data(), 0, 4096 * 4, 128); }
    // the path: room of path_side entries on both sides of [path_off, path_off + m]
    const uint32_t len = hdr.cap_path ? hdr.m + 1 : 0;
    const bool short_left = hdr.path_off < path_side, short_right = (uint64_t)hdr.path_off + len + path_side > pe.size();
    if (short_left || short_right) {
        const size_t cap = grow(0, (size_t)len + 4 * (size_t)path_side + 1024 + len / 2);
        std::vector<uint32_t> e2(cap), n2(cap);
        std::vector<uint8_t> s2(cap);
        const uint32_t off2 = (uint32_t)((cap - len) / 2);
        if (len) {
            memcpy(e2.data() + off2, pe.data() + hdr.path_off, (size_t)(len - 1) * 4);
            memcpy(n2.data() + off2, pn.data() + hdr.path_off, (size_t)len * 4);
            memcpy(s2.data() + off2, ps.data() + hdr.path_off, len);
        }
        if (pz && (pa & 32u)) {                             // (everything outside the copied path)
            for (size_t i = 0; i < cap; ++i) if (i < off2 || i >= (size_t)off2 + len) { e2[i] = pw, n2[i] = pw, s2[i] = (uint8_t)pw; }
            if (len) e2[off2 + len - 1] = pw;
        }
        pe.swap(e2), pn.swap(n2), ps.swap(s2);
        sv_e.assign(cap, pz && (pa & 64u) ? pw : 0), sv_n.assign(cap, pz && (pa & 64u) ? pw : 0), sv_s.assign(cap, pz && (pa & 64u) ? (uint8_t)pw : 0);
        hdr.pos_bias += off2 - hdr.path_off;           // (what the nodes remember of their places moves with the arrays)
        hdr.path_off = off2;
    }
    hdr.cap_nodes = (uint32_t)nodes.size(), hdr.cap_edges = (uint32_t)edges.size(), hdr.cap_chunks = (uint32_t)chunks.size();
    hdr.cap_path = (uint32_t)pe.size(), hdr.cap_wk = (uint32_t)wk.size(), hdr.cap_multi = (uint32_t)multi.size();
}

void SoaStore::ensure(const SoaNeed &need)
{
    // beyond the update's own worst case: room for the private copies a removeCycles behind it may make (a quarter more, at least 64 K entries)
    static const char *sl = getenv("NSGPU_SOA_SLACK");          // tests: next to no spare room (removeCycles' ERR_ROOM on every split of some size)
    const uint32_t slack_n = sl ? (uint32_t)atoi(sl) : std::max<uint32_t>(1u << 16, hdr.n_nodes / 4), slack_e = sl ? (uint32_t)atoi(sl) : std::max<uint32_t>(1u << 16, hdr.n_edges / 4), slack_c = sl ? (uint32_t)atoi(sl) : std::max<uint32_t>(1u << 16, hdr.n_chunks / 4);
    reserve(hdr.n_nodes + need.nodes + slack_n, hdr.n_edges + need.edges + slack_e, hdr.n_chunks + need.chunks + slack_c, need.path_side, need.wk);
}

void soa_script(const std::string &s, const std::vector<EditOp> &script, ssize_t begin_offset, ssize_t end_offset, std::vector<uint32_t> &ops, uint32_t &n_run, uint32_t &n_ins)
{
    ops.clear();
    n_run = n_ins = 0;
    if (begin_offset <= -1) {
        const size_t k = std::min((size_t)(-begin_offset), s.size());
        for (size_t i = 0; i < k; ++i) ops.push_back(dg::op_make(1, (uint8_t)s[i], 0));
        n_ins += (uint32_t)k;
    }
    for (const EditOp &op : script) {
        if (op.type == 0) {
            uint32_t left = op.num;
            while (left) { const uint32_t t = std::min(left, dg::kOpMaxNum); ops.push_back(dg::op_make(0, 0, t)); n_run += t - 1; left -= t; }
        } else if (op.type == 1) { ops.push_back(dg::op_make(1, op.base, 0)); ++n_ins; }
        else if (op.type == 2) ops.push_back(dg::op_make(2, 0, 0));
    }
    if (end_offset > 0) {
        const size_t k = std::min((size_t)end_offset, s.size());
        for (size_t i = s.size() - k; i < s.size(); ++i) ops.push_back(dg::op_make(1, (uint8_t)s[i], 0));
        n_ins += (uint32_t)k;
    }
}

void soa_patch_path(std::string &path, uint32_t P, uint32_t S, uint32_t new_len, const uint8_t *new_bases)
{
    const size_t Lo = path.size();
    if ((size_t)P + S > Lo || (size_t)P + S > new_len) { fprintf(stderr, "nsgpu: consensus patch out of range (internal error): P %u S %u old %zu new %u\n", P, S, Lo, new_len); abort(); }
    const size_t mid = (size_t)new_len - P - S;
    if (Lo == new_len) memcpy(&path[P], new_bases, mid);
    else path.replace(P, Lo - P - S, reinterpret_cast<const char *>(new_bases), mid);
}

// ---------------------------------------------------------------------------
// the reference's interface, on the host
// ---------------------------------------------------------------------------
static void die_on(const dg::Hdr &h, const char *where)
{
    if (!h.err) return;
    fprintf(stderr, "nsgpu: consensus graph (host arrays): error %u in %s, raised at dgraph.hpp:%u (capacity %u / scripts %u / degree %u / walk %u / work area %u)\n", h.err, where, h.err_line, h.err & dg::ERR_CAP,
            h.err & dg::ERR_SCRIPT, h.err & dg::ERR_DEGREE, h.err & dg::ERR_WALK, h.err & dg::ERR_SCRATCH);
    abort();
}

void SoaGraph::initialize(const std::string &seed, read_t id, long pos)
{
    st_.reserve((uint32_t)seed.size() + 1024, (uint32_t)seed.size() + 1024, 1024, (uint32_t)seed.size() + 1024, 1u << 18);
    // (reserve centres an empty path: initialize puts the seed in the middle of the arrays itself)
    dg::HostTeam t;
    HostOps ops(st_.view(), t);
    { static const char *e = getenv("NSGPU_SOA_DEBUG_FLAGS"); st_.hdr.dbg_flags = dbg_flags_override >= 0 ? (uint32_t)dbg_flags_override : e ? (uint32_t)atoi(e) : 0; }       // (tests: the rare branches on every update)
    ops.initialize(reinterpret_cast<const uint8_t *>(seed.data()), (uint32_t)seed.size(), id);
    die_on(st_.hdr, "initialize");
    reads.insert(std::make_pair(id, SoaRead{pos, 0u, seed.size(), false}));
    main_path = seed;
    start_pos = pos, end_pos = pos + (long)seed.size();
    path_changed_from = 0;
    fresh_ = true, pending_ = false;
}

void SoaGraph::update_graph(const std::string &s, const std::vector<EditOp> &script, ssize_t begin_offset, ssize_t end_offset, read_t id, long pos, bool rc)
{
    static thread_local std::vector<uint32_t> ops;
    uint32_t n_run, n_ins;
    soa_script(s, script, begin_offset, end_offset, ops, n_run, n_ins);
    st_.ensure(soa_need((uint32_t)ops.size(), n_run, n_ins, st_.hdr.m + 1));
    dg::HostTeam t;
    HostOps o(st_.view(), t);
    o.update(ops.data(), (uint32_t)ops.size(), (int64_t)begin_offset, (int64_t)end_offset, id);
    die_on(st_.hdr, "update_graph");
    reads.insert(std::make_pair(id, SoaRead{pos, st_.hdr.initial, s.size(), rc}));
    fresh_ = false, pending_ = true;
}

void SoaGraph::calculate_main_path_greedy()
{
    if (fresh_) { fresh_ = false; return; }          // initialize laid the whole seed out as the path
    dg::HostTeam t;
    bool have_path = false;
    for (;;) {
        HostOps o(st_.view(), t);
        if (!have_path) o.main_path();
        if (!st_.hdr.err) { have_path = true; o.finish_path(); }
        if (st_.hdr.err == dg::ERR_SCRATCH && st_.hdr.stage < 3 && !have_path) {      // the walks' lists did not fit and nothing was changed yet: a larger work area
            st_.hdr.err = 0;
            st_.wk.resize(st_.wk.size() * 2);
            st_.hdr.cap_wk = (uint32_t)st_.wk.size();
            continue;
        }
        if (st_.hdr.err == dg::ERR_ROOM) {                    // removeCycles stopped in front of a split that does not fit: room for it, then again
            st_.hdr.err = 0;
            SoaNeed need{st_.hdr.need_nodes, st_.hdr.need_edges, st_.hdr.need_chunks, 0, (uint32_t)st_.wk.size()};
            st_.ensure(need);
            continue;
        }
        break;
    }
    die_on(st_.hdr, "calculate_main_path_greedy");
    const dg::Hdr &h = st_.hdr;
    {   // tests (NSGPU_SOA_POISON): every id handed out so far must have been written -- a record that still is the fill pattern was only counted
        static const char *pz = getenv("NSGPU_SOA_POISON");
        if (pz) {
            const uint32_t pw = (uint32_t)strtoul(pz, nullptr, 0);
            auto untouched = [&](const void *rec, size_t words) { const uint32_t *q = static_cast<const uint32_t *>(rec); for (size_t i = 0; i < words; ++i) if (q[i] != pw) return false; return true; };
            for (uint32_t i = 0; i < h.n_nodes; ++i) if (untouched(&st_.nodes[i], 8)) { fprintf(stderr, "SOA POISON: node %u of %u was never written (edges %u)\n", i, h.n_nodes, h.n_edges); break; }
            for (uint32_t i = 0; i < h.n_edges; ++i) if (untouched(&st_.edges[i], 16)) { fprintf(stderr, "SOA POISON: edge %u of %u was never written (nodes %u)\n", i, h.n_edges, h.n_nodes); break; }
        }
    }
    if (h.old_len != main_path.size()) { fprintf(stderr, "nsgpu: consensus length out of step with the graph (internal error)\n"); abort(); }
    soa_patch_path(main_path, h.P, h.S, h.new_len, st_.ps.data() + h.path_off + h.P);
    if (h.P < path_changed_from) path_changed_from = h.P;
    const SoaRead &er = reads.at(h.ending_id);
    end_pos = er.pos + (long)er.len;
    start_pos = reads.at(h.starting_id).pos;
    pending_ = false;
}

void SoaGraph::write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source)
{
    SoaEmitter em(st_.view(), reads);
    em.write_reads(o, source);
}
bool SoaGraph::read_string(read_t id, std::string &out)
{
    SoaEmitter em(st_.view(), reads);
    return em.read_string(id, out);
}
bool SoaGraph::has_cycle()
{
    SoaEmitter em(st_.view(), reads);
    std::string tmp;
    for (auto &it : reads) if (!em.read_string(it.first, tmp)) return true;
    return false;
}

// ---------------------------------------------------------------------------
// emission (ConsensusGraph::writeReads, read2EditScript; the walk guided by the read's own bases: consensus.cpp)
// ---------------------------------------------------------------------------
static inline uint8_t base_bit(char b) { return b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 4 : b == 'T' ? 8 : 16; }

namespace {
struct View {
    const dg::G &g;
    uint32_t out_ref(const dg::Node &x, uint32_t i) const
    {
        if (i < dg::kOutInl) return x.out[i];
        i -= dg::kOutInl;
        uint32_t c = x.out_ext;
        while (i >= dg::kChunkIds) c = g.chunks[c].next, i -= dg::kChunkIds;
        return g.chunks[c].v[i];
    }
    char sink_base(uint32_t ref) const { const uint32_t c = ref >> 29; return c < 4 ? "ACGT"[c] : (char)g.nodes[g.edges[ref & dg::kRefMask].sink].base; }
    bool edge_has(const dg::Edge &e, uint32_t id) const
    {
        const uint32_t n = e.count, ni = n < dg::kEdgeInl ? n : dg::kEdgeInl;
        for (uint32_t p = 0; p < ni; ++p) if (e.ids[p] == id) return true;
        uint32_t left = n - ni, c = e.head;
        while (left) { const dg::Chunk &k = g.chunks[c]; const uint32_t t = left < dg::kChunkIds ? left : dg::kChunkIds; for (uint32_t p = 0; p < t; ++p) if (k.v[p] == id) return true; left -= t; c = k.next; }
        return false;
    }
};
}  // namespace

// Node::getEdgeInRead (:83-91): the first out-edge whose list holds the read
uint32_t SoaEmitter::edge_in_read(uint32_t n, read_t id) const
{
    const View v{g_};
    const dg::Node &x = g_.nodes[n];
    for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t e = v.out_ref(x, i) & dg::kRefMask; if (v.edge_has(g_.edges[e], id)) return e; }
    return dg::NIL;
}

// The out-edge of node n that read `id` takes when its next base is nb: the only out-edge, else the only one whose sink carries the
// base; when several do, the read is on exactly one of them: the short lists are looked through and the read is on the longest one
// if it is on none of those.
uint32_t SoaEmitter::way_out_of(uint32_t n, char nb, read_t id) const
{
    const View v{g_};
    const dg::Node &x = g_.nodes[n];
    if (x.n_out == 1) return x.out[0] & dg::kRefMask;
    uint32_t cand[8];
    int cnt = 0;
    for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t r = v.out_ref(x, i); if (v.sink_base(r) == nb) { if (cnt < 8) cand[cnt] = r & dg::kRefMask; ++cnt; } }
    if (cnt == 1) return cand[0];
    if (cnt == 0 || cnt > 8) return edge_in_read(n, id);
    int big = 0;
    for (int c = 1; c < cnt; ++c) if (g_.edges[cand[c]].count > g_.edges[cand[big]].count) big = c;
    for (int c = 0; c < cnt; ++c) if (c != big && v.edge_has(g_.edges[cand[c]], id)) return cand[c];
    return cand[big];
}

template <class Visit, class VisitRun>
void SoaEmitter::walk_read(const SoaRead &r, read_t id, const ReadBases *src, Visit visit, VisitRun visit_run) const
{
    uint32_t cur = r.start;
    if (!src) {
        while (cur != dg::NIL) { visit(cur); const uint32_t e = edge_in_read(cur, id); cur = e != dg::NIL ? g_.edges[e].sink : dg::NIL; }
        return;
    }
    const size_t L = src->len;
    static thread_local std::string oriented;
    const char *rb = src->bases;
    if (r.rc) { reverse_complement(src->bases, L, oriented); rb = oriented.data(); }
    const size_t n_main = g_.h->m;
    const char *const cons = reinterpret_cast<const char *>(g_.ps + g_.h->path_off);       // cons[j] = base of main-path node j
    const uint32_t *const main_nodes = g_.pn + g_.h->path_off;
    for (size_t i = 0; i < L;) {
        visit(cur);
        if (g_.nodes[cur].on_main) {
            size_t j = main_idx_[cur];
            for (;;) {
                if (++i == L) return;
                if (j < n_main) {
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
                    if (t) { visit_run(t); j += t, i += t - 1; continue; }
                }
                if (j < n_main && next_fork_[j] != j) {
                    const size_t j1 = j + 1, stop = next_fork_[j1];
                    const size_t k = stop - j1 < L - i ? stop - j1 : L - i;
                    if (k) { visit_run(k); i += k; if (i == L) return; }
                    j = j1 + k;
                    visit_run(1);
                    continue;
                }
                const char nb = rb[i];
                if (j < n_main && cons[j + 1] == nb) {
                    if (!(side_mask_[j] & base_bit(nb))) { ++j; visit_run(1); continue; }
                    const read_t *a = amb_ids_.data() + amb_off_[j], *b = amb_ids_.data() + amb_off_[j + 1];
                    bool side = false;
                    for (; a != b; ++a) if (*a == id || *a == kAmbComplex) { side = true; break; }
                    if (!side) { ++j; visit_run(1); continue; }
                }
                cur = g_.edges[way_out_of(main_nodes[j], nb, id)].sink;
                break;
            }
            continue;
        }
        if (++i == L) break;
        cur = g_.edges[way_out_of(cur, rb[i], id)].sink;
    }
}

size_t SoaEmitter::read_to_edits(const SoaRead &r, read_t id, const ReadBases *src, std::vector<EditOp> &script, uint32_t &pos) const
{
    script.clear();
    script.reserve(r.len / 8 + 16);
    bool seen_main = false;
    size_t dis = 0, at = 0, same = 0;
    pos = 0;
    auto flush = [&]() { if (same > 0) { script.push_back(EditOp{0, 0, (uint32_t)same}); same = 0; } };
    walk_read(r, id, src, [&](uint32_t cur) {
        if (g_.nodes[cur].on_main) {
            const size_t p = main_idx_[cur];
            if (!seen_main) seen_main = true, pos = (uint32_t)p, at = p;
            if (p > at) flush();
            for (; at < p; ++at) { script.push_back(EditOp{2, (uint8_t)'-', 0}); ++dis; }
            ++same;
            ++at;
        } else {
            flush();
            script.push_back(EditOp{1, g_.nodes[cur].base, 0});
            ++dis;
        }
    }, [&](size_t k) { same += k, at += k; });
    flush();
    return dis;
}

size_t SoaEmitter::write_read(StreamSet &o, const SoaRead &r, read_t id, const ReadBases *src) const
{
    uint32_t offset;
    static thread_local std::vector<EditOp> raw, es;
    read_to_edits(r, id, src, raw, offset);
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
    return dis;
}

void SoaEmitter::write_reads(StreamSet &o, const std::function<ReadBases(read_t)> *source)
{
    const View v{g_};
    const dg::Hdr &h = *g_.h;
    const size_t n_main = h.m;
    const uint32_t *pn = g_.pn + h.path_off, *pe = g_.pe + h.path_off;
    const char *cons = reinterpret_cast<const char *>(g_.ps + h.path_off);
    main_idx_.assign(h.n_nodes, 0);
    next_fork_.resize(n_main + 1);
    side_mask_.assign(n_main + 1, 0);
    amb_off_.assign(n_main + 2, 0);
    amb_ids_.clear();
    for (size_t j = 0; j <= n_main; ++j) {
        const uint32_t n = pn[j];
        const dg::Node &x = g_.nodes[n];
        main_idx_[n] = (uint32_t)j;
        const bool single = x.n_out == 1 && j < n_main;
        next_fork_[j] = single ? UINT32_MAX : (uint32_t)j;
        amb_off_[j + 1] = (uint32_t)amb_ids_.size();
        if (single) continue;
        const uint32_t path_edge = j < n_main ? pe[j] : dg::NIL;
        uint8_t m = 0;
        for (uint32_t i = 0; i < x.n_out; ++i) { const uint32_t r = v.out_ref(x, i); if ((r & dg::kRefMask) != path_edge) m |= base_bit(v.sink_base(r)); }
        side_mask_[j] = m;
        if (j < n_main && (m & base_bit(cons[j + 1]))) {
            const size_t at = amb_ids_.size();
            bool simple = true;
            for (uint32_t i = 0; i < x.n_out && simple; ++i) {
                const uint32_t r = v.out_ref(x, i), e = r & dg::kRefMask;
                if (e == path_edge || v.sink_base(r) != cons[j + 1]) continue;
                const dg::Edge &ed = g_.edges[e];
                if (ed.count > 16) { simple = false; break; }
                const uint32_t ni = ed.count < dg::kEdgeInl ? ed.count : dg::kEdgeInl;
                for (uint32_t p = 0; p < ni; ++p) amb_ids_.push_back(ed.ids[p]);
                for (uint32_t p = ni; p < ed.count; ++p) amb_ids_.push_back(g_.chunks[ed.head].v[p - ni]);       // (at most 16 ids: one chunk)
            }
            if (!simple) { amb_ids_.resize(at); amb_ids_.push_back(kAmbComplex); }
        }
        amb_off_[j + 1] = (uint32_t)amb_ids_.size();
    }
    follow_ok_.assign(n_main + 8, 0);
    for (size_t j = 0; j < n_main; ++j) follow_ok_[j] = next_fork_[j] == UINT32_MAX || !(side_mask_[j] & base_bit(cons[j + 1]));
    for (size_t j = n_main; j-- > 0;) if (next_fork_[j] == UINT32_MAX) next_fork_[j] = next_fork_[j + 1];
    read_t prev = 0;
    for (auto &it : reads_) {
        const read_t diff = it.first - prev;
        o.id_contigs.append(reinterpret_cast<const char *>(&diff), 4);
        o.complement.push_back(it.second.rc ? 'c' : 'n');
        prev = it.first;
        if (source) { const ReadBases rb = (*source)(it.first); write_read(o, it.second, it.first, &rb); }
        else write_read(o, it.second, it.first, nullptr);
    }
    o.complement.push_back('\n');
}

bool SoaEmitter::read_string(read_t id, std::string &out) const
{
    out.clear();
    auto it = reads_.find(id);
    if (it == reads_.end()) return false;
    uint32_t cur = it->second.start;
    while (cur != dg::NIL) {
        out.push_back((char)g_.nodes[cur].base);
        const uint32_t e = edge_in_read(cur, id);
        cur = e != dg::NIL ? g_.edges[e].sink : dg::NIL;
        if (out.size() > it->second.len + 8) return false;
    }
    return out.size() == it->second.len;
}

}  // namespace cons
}  // namespace nsgpu
