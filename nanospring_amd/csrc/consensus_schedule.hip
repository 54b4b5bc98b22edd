// consensus_schedule.hip -- who grows which contig when: the seed rule (Consensus::getRead, src/Consensus.cpp:444-468, and the conflict-aware
// rule over MinHash-locality buckets), the claims after alignRead (src/Consensus.cpp:256-277) resolved in builder order, and
// the schedule the library derives from the input itself (nsgpu_set_schedule_auto).  State: engine.hpp.
#include "engine.hpp"

namespace nsgpu {

// ---- the schedule derived from the input (nsgpu_set_schedule_auto; nsgpu_consensus_run with 0 builders) ---------------------------------
// The reference has one knob, -t (src/main.cpp:46-78), and its streams grow with it: contigs that grow at the same time cut each other short.
// This library's knobs -- builders, groups, the seed rule's bucket depth and radii -- trade the same thing, and the right values differ per
// input (cfg2's on cfg3: 19 Mbases/s instead of 80).  What the library knows after nsgpu_build_index decides them:
//   * coverage, from the whole-read filter results per read r (both strands, the read itself included; r = 12 at 20x, 120 at 217x): a deep
//     read set over a small genome has few places for contigs to grow apart, so the buckets must be small (depth 1) for the exclusion radius
//     not to block the whole genome, while a shallow one over a large genome wants depth 3 / 5 rings;
//   * the input size: every builder beyond the first costs ~80 kB of streams (one more contig boundary now and then); 1 builder per 10 Mbases
//     keeps that within 5 % of the streams the reference's own -t 8 writes, but never fewer than the seed rule can keep busy.
// One group: with this few builders a slot is as long as its GPU round trips.  A function of replicated values only: every rank of a
// multi-GPU job derives the same schedule.  tests/oracle_lib.py auto_schedule restates the rule for the lock-step oracle.
AutoSchedule auto_schedule(uint64_t n_reads, uint64_t n_bases, uint64_t n_filter_results)
{
    const double r = n_reads ? (double)n_filter_results / (double)n_reads : 0.0;
    AutoSchedule a;
    uint64_t b_min;
    if (r < 30.0) a.depth = 3, a.rings = 5, a.tail = 3, b_min = 32;
    else if (r < 70.0) a.depth = 2, a.rings = 4, a.tail = 3, b_min = 96;
    else a.depth = 1, a.rings = 4, a.tail = 3, b_min = 128;
    const uint64_t b = std::min<uint64_t>(1024, std::max<uint64_t>(b_min, n_bases / 10000000ull));
    a.builders = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, n_reads ? n_reads : 1));
    return a;
}

// phase 2: (gid, cursor) of every local builder that needs a new contig
void engine_seed_requests(nsgpu_ctx *c, std::vector<uint32_t> &gids, std::vector<uint32_t> &cursors, int group)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    gids.clear(); cursors.clear();
    for (Builder &b : E->D.B) if (in_group(b, group) && b.st == Builder::NEED_CONTIG) { gids.push_back(b.gid); cursors.push_back(b.cursor); }
}

// resolve the seed requests of ALL ranks on the replicated in_graph[], in global builder order
// (Consensus::getRead + createGraph, src/Consensus.cpp:388-403, 444-468); returns how many builders started a contig
uint32_t engine_seed_resolve(nsgpu_ctx *c, const uint32_t *gids, const uint32_t *cursors, uint32_t n)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    std::vector<uint32_t> ord(n);
    for (uint32_t i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return gids[a] < gids[b]; });
    uint32_t started = 0;
    Engine::SeedPolicy &P = E->sp;
    if (P.depth) {
        // conflict-aware seeds (see Engine::SeedPolicy): the contigs the waiting builders finished leave the occupancy, then the free
        // buckets' lowest unclaimed reads are handed out in ascending read order to the builders in global builder order
        for (uint32_t k = 0; k < n; ++k) {
            std::vector<uint32_t> &m = P.members[gids[ord[k]]];
            for (uint32_t x : m) --P.occ[P.bucket_of[x]];
            m.clear();
        }
        // towards the end of a run most builders wait while a few contigs close the last gaps: when more than half of ALL builders ask for a
        // seed in one round, the exclusion radius drops to tail_rings -- a seed in such a gap is one more contig, and halves what is left of it
        P.rings_now = 2ull * n > E->n_total ? P.tail_rings : P.rings;
        std::vector<std::pair<uint32_t, uint32_t>> cand;          // (lowest unclaimed read, bucket) of every bucket a seed may lie in
        const uint32_t nb = (uint32_t)P.occ.size();
        if (P.n_unclaimed) {
            P.blocked.assign(nb, 0);
            P.q_cur.clear();
            for (uint32_t b = 0; b < nb; ++b) if (P.occ[b]) P.q_cur.push_back(b);
            P.spread();
            for (uint32_t b = 0; b < nb; ++b) {
                if (P.blocked[b]) continue;
                uint32_t &i = P.bk_next[b];
                const uint32_t cnt = (uint32_t)(P.bk_off[b + 1] - P.bk_off[b]);
                while (i < cnt && D.in_graph[P.bk_reads[P.bk_off[b] + i]]) ++i;
                if (i < cnt) cand.emplace_back(P.bk_reads[P.bk_off[b] + i], b);
            }
        }
        std::sort(cand.begin(), cand.end());
        size_t ci = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t gid = gids[ord[k]];
            Builder *b = E->local(gid);
            if (P.n_unclaimed == 0) { ++E->n_done_global; if (b) b->st = Builder::DONE; continue; }
            while (ci < cand.size() && P.blocked[cand[ci].second]) ++ci;                    // an earlier grant of this round came too close
            if (ci == cand.size()) { ++P.n_idle; continue; }                                  // asks again at its group's next slot
            const read_t r = cand[ci].first;
            const uint32_t bk = cand[ci].second;
            ++ci;
            D.in_graph[r] = 1;
            --P.n_unclaimed;
            P.members[gid].push_back(r);
            ++P.occ[bk];
            P.q_cur.assign(1, bk);
            P.spread();
            ++started;
            if (b) D.start_contig(*b, r);
        }
        return started;
    }
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t gid = gids[ord[k]];
        read_t r = cursors[ord[k]];
        while (r < D.N && D.in_graph[r]) ++r;
        Builder *b = E->local(gid);
        if (r >= D.N) { ++E->n_done_global; if (b) b->st = Builder::DONE; continue; }
        D.in_graph[r] = 1;
        ++started;
        if (b) D.start_contig(*b, r);
    }
    return started;
}

// the buckets of the conflict-aware seed rule (Engine::SeedPolicy), from the whole-read filter results of every read
// the filter's answer to every whole read, both strands (the edges of the graph the seed buckets are built on; their number per read is what
// the automatic schedule reads the coverage from): into P.wr_off / P.wr_ids
int whole_read_filter(nsgpu_ctx *c, Engine *E)
{
    Engine::SeedPolicy &P = E->sp;
    const uint32_t N = E->D.N;
    if (P.have_wr) return NSGPU_OK;
    uint64_t n_cand = 0;
    if (!c->have_sketch) {
        // a rank of a multi-GPU job whose tables came from the all-to-all holds the sketch rows of its own id range only: the whole-read
        // queries need every row, and sketching all reads here (milliseconds) is cheaper than another exchange -- same rows on every rank
        NS_CHECK(c->have_salts, NSGPU_ERR_ARG, "consensus: no salts");
        NS_TRY(c->sketch.reserve(((size_t)N * c->prm.n + 1) * 8));
        NS_TRY(launch_sketch(c, c->reads, c->sketch.as<uint64_t>(), nullptr));
        NS_HIP(stream_wait(c->stream));
        c->have_sketch = true;
    }
    NS_TRY(nsgpu_filter_all_reads(c, &n_cand));
    P.wr_off.assign(2 * (size_t)N + 1, 0);
    P.wr_ids.assign(n_cand + 1, 0);
    NS_TRY(nsgpu_filter_all_fetch(c, P.wr_off.data(), P.wr_ids.data()));
    c->have_filter_all = false;                                     // the engine's window queries reuse the device buffers
    P.n_filter_results = n_cand, P.have_wr = true;
    return NSGPU_OK;
}

int seed_policy_init(nsgpu_ctx *c, Engine *E)
{
    Engine::SeedPolicy &P = E->sp;
    Driver &D = E->D;
    P.depth = c->seed_bucket_depth, P.rings = c->seed_rings, P.tail_rings = std::min(c->seed_tail_rings, c->seed_rings), P.rings_now = P.rings;
    if (!P.depth) { P.wr_off.clear(), P.wr_ids.clear(); return NSGPU_OK; }
    const uint32_t N = D.N;
    NS_TRY(whole_read_filter(c, E));
    const uint64_t n_cand = P.n_filter_results;
    const std::vector<uint64_t> &off = P.wr_off;
    const std::vector<uint32_t> &ids = P.wr_ids;
    P.bucket_of.assign(N, ~0u);
    uint32_t nb = 0;
    std::vector<uint32_t> cur, nxt;
    for (uint32_t r = 0; r < N; ++r) {
        if (P.bucket_of[r] != ~0u) continue;
        const uint32_t b = nb++;
        P.bucket_of[r] = b;
        cur.assign(1, r);
        for (uint32_t d = 0; d < P.depth && !cur.empty(); ++d) {
            nxt.clear();
            for (uint32_t x : cur)
                for (uint64_t i = off[2 * (size_t)x]; i < off[2 * (size_t)x + 2]; ++i) { const uint32_t y = ids[i]; if (P.bucket_of[y] == ~0u) { P.bucket_of[y] = b; nxt.push_back(y); } }
            cur.swap(nxt);
        }
    }
    std::vector<std::vector<uint32_t>> adj(nb);
    for (uint32_t x = 0; x < N; ++x)
        for (uint64_t i = off[2 * (size_t)x]; i < off[2 * (size_t)x + 2]; ++i) {
            const uint32_t bx = P.bucket_of[x], by = P.bucket_of[ids[i]];
            if (bx != by) { adj[bx].push_back(by); adj[by].push_back(bx); }
        }
    P.adj_off.assign(nb + 1, 0);
    P.adj.clear();
    for (uint32_t b = 0; b < nb; ++b) {
        std::sort(adj[b].begin(), adj[b].end());
        adj[b].erase(std::unique(adj[b].begin(), adj[b].end()), adj[b].end());
        P.adj.insert(P.adj.end(), adj[b].begin(), adj[b].end());
        P.adj_off[b + 1] = P.adj.size();
    }
    P.bk_off.assign(nb + 1, 0);
    for (uint32_t r = 0; r < N; ++r) ++P.bk_off[P.bucket_of[r] + 1];
    for (uint32_t b = 0; b < nb; ++b) P.bk_off[b + 1] += P.bk_off[b];
    P.bk_reads.resize(N);
    { std::vector<uint64_t> fill(P.bk_off.begin(), P.bk_off.end() - 1); for (uint32_t r = 0; r < N; ++r) P.bk_reads[fill[P.bucket_of[r]]++] = r; }
    P.bk_next.assign(nb, 0);
    P.occ.assign(nb, 0);
    P.members.assign(E->n_total, std::vector<uint32_t>());
    P.n_unclaimed = N;
    if (getenv("NSGPU_CONS_DEBUG")) fprintf(stderr, "[cons] seed policy: %u buckets of depth %u over %u reads (%llu filter results), rings %u\n", nb, P.depth, N, (unsigned long long)n_cand, P.rings);
    return NSGPU_OK;
}

// phase 6a: (gid, read) of every local builder whose alignment succeeded
void engine_claim_requests(nsgpu_ctx *c, std::vector<uint32_t> &gids, std::vector<uint32_t> &reads, int group)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    gids.clear(); reads.clear();
    for (Builder &b : E->D.B) if (in_group(b, group) && b.st == Builder::ALIGNED && b.aln.ok) { gids.push_back(b.gid); reads.push_back(b.pend); }
}

// phase 6b: claims of ALL ranks, strictly in global builder order (src/Consensus.cpp:256-277 without lock contention)
void engine_claim_resolve(nsgpu_ctx *c, const uint32_t *gids, const uint32_t *reads, uint32_t n)
{
    Engine *E = static_cast<Engine *>(c->cons_engine);
    Driver &D = E->D;
    std::vector<uint32_t> ord(n);
    for (uint32_t i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return gids[a] < gids[b]; });
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t gid = gids[ord[k]], r = reads[ord[k]];
        if (r >= D.N || D.in_graph[r]) continue;
        D.in_graph[r] = 1;
        if (E->sp.depth) { E->sp.members[gid].push_back(r); ++E->sp.occ[E->sp.bucket_of[r]]; --E->sp.n_unclaimed; }
        if (Builder *b = E->local(gid)) { b->accepted = true; ++b->n_aligner; }
    }
    for (Builder &b : D.B) if (b.st == Builder::ALIGNED) b.st = Builder::GOT_ALIGN;
    ++c->cons_stats.n_rounds;
}

}  // namespace nsgpu
