// consensus_dist.hip -- the contig stage over the ranks of a communicator (one process per GPU; dist.hip has the communicator and the load /
// sketch / table phases): every rank runs the engine's phases for its own builders, the seed requests, the slot's candidate reads and the
// claims travel in small all-gathers and are resolved on replicated tables in global builder order.  State: engine.hpp.
#include "engine.hpp"

using namespace nsgpu;

// The same loop over the ranks of a communicator: after each slot ONE all-gather carries every rank's claim requests (group b)
// and seed requests (group h); both lists are then resolved on every rank in global builder order.  A rank never has more
// requests than local builders, so the exchange buffer has a fixed size and needs no size negotiation.
static int run_consensus_dist(nsgpu_ctx *c, Comm &C, uint32_t n_builders_total, uint32_t n_threads_out, uint64_t *n_coll_out, uint64_t *bytes_out)
{
    NS_CHECK(n_threads_out >= 1, NSGPU_ERR_ARG, "n_threads_out must be >= 1");
    // engine_begin does rank-local GPU work (the seed policy sketches and queries all reads, allocates device and pinned memory): one rank may
    // fail where the others do not, so its return code travels through a status all-gather of its own before anybody enters the slot loop.
    const int rc_begin = engine_begin(c, n_builders_total, C.rank, C.world);
    Engine *E = static_cast<Engine *>(c->cons_engine);
    const uint32_t W = C.world;
    // one exchange = [status | claim count, gids, reads | seed count, gids, cursors]; a rank never has more requests than local builders
    // (0 builders = the automatic schedule's own count, at most 1024: the bound must not depend on what a rank derives -- a rank whose
    // engine_begin failed still has to enter the exchanges with buffers of the size the others use)
    const uint32_t nb_bound = n_builders_total ? n_builders_total : 1024u;
    const size_t cap = (nb_bound + W - 1) / W + 1, blk = 1 + 2 * cap, words = 1 + 2 * blk;
    std::vector<uint32_t> mine(words), all(words * W), ca, cb, sa, sb, ga, gb;
    uint64_t n_coll = 0;
    std::string local_err;
    // Every rank ALWAYS enters the collective and sends its status with its lists, so that one failing rank ends the stage on all ranks
    // instead of leaving the others blocked in an all-gather it never joins.  rc_local: what this rank's part of the slot returned.
    auto exchange = [&](int rc_local, const std::vector<uint32_t> *cl_g, const std::vector<uint32_t> *cl_r, const std::vector<uint32_t> *sd_g, const std::vector<uint32_t> *sd_c) -> int {
        if (rc_local == NSGPU_OK && ((cl_g && cl_g->size() > cap) || (sd_g && sd_g->size() > cap))) { set_error("more requests than local builders"); rc_local = NSGPU_ERR_RANGE; }
        if (rc_local != NSGPU_OK) local_err = nsgpu_last_error();
        std::fill(mine.begin(), mine.end(), 0u);
        mine[0] = (uint32_t)(rc_local != NSGPU_OK);
        if (rc_local == NSGPU_OK && cl_g) { mine[1] = (uint32_t)cl_g->size(); std::copy(cl_g->begin(), cl_g->end(), mine.begin() + 2); std::copy(cl_r->begin(), cl_r->end(), mine.begin() + 2 + cap); }
        if (rc_local == NSGPU_OK && sd_g) { mine[1 + blk] = (uint32_t)sd_g->size(); std::copy(sd_g->begin(), sd_g->end(), mine.begin() + 2 + blk); std::copy(sd_c->begin(), sd_c->end(), mine.begin() + 2 + blk + cap); }
        const int rc_coll = C.all_gather(mine.data(), all.data(), words * 4, false, c->stream);
        ++n_coll;
        if (rc_local != NSGPU_OK) { set_error("%s", local_err.c_str()); return rc_local; }
        NS_TRY(rc_coll);
        for (uint32_t r = 0; r < W; ++r)
            if (all[(size_t)r * words]) { set_error("contig stage: rank %u reported an error; stopping on every rank", r); return NSGPU_ERR_HIP; }
        return NSGPU_OK;
    };
    auto gathered = [&](size_t base) {        // request lists of all ranks: (ga, gb)
        ga.clear(); gb.clear();
        for (uint32_t r = 0; r < W; ++r) {
            const uint32_t *v = all.data() + (size_t)r * words + base;
            ga.insert(ga.end(), v + 1, v + 1 + v[0]);
            gb.insert(gb.end(), v + 1 + cap, v + 1 + cap + v[0]);
        }
    };
    NS_TRY(exchange(rc_begin, nullptr, nullptr, nullptr, nullptr));
    for (uint32_t slot = 0;; ++slot) {
        const int G = n_groups(c);
        const int h = (int)(slot % G), b = (int)((slot + 1) % G);
        if (G == 1) {
            // ONE group: two small all-gathers per slot -- seed requests after the host phase, claim requests after the batches
            int rc = engine_slot(c, slot, 1);
            if (rc == NSGPU_OK) rc = engine_window_loop(c, h);
            if (rc == NSGPU_OK) engine_seed_requests(c, sa, sb, h);
            NS_TRY(exchange(rc, nullptr, nullptr, &sa, &sb));
            gathered(1 + blk);
            rc = NSGPU_OK;
            if (!ga.empty() && engine_seed_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size()) != 0) { engine_advance(c, true, h); rc = engine_window_loop(c, h); }
            // which reads the builders of ALL ranks are about to align: a builder whose read nobody else aligns (and nobody has claimed) cannot
            // lose its claim, so its graph update may ride on the DP phase as on one GPU (engine_early_updates) instead of waiting for the
            // claim exchange -- one more small all-gather per slot
            if (rc == NSGPU_OK) {
                ca.clear(), cb.clear();
                for (const Builder &bb : E->D.B) if (bb.st == Builder::WAIT_ALIGN || (bb.st == Builder::DEFERRED && bb.defer_due == slot)) { ca.push_back(bb.gid); cb.push_back(bb.pend); }
            }
            NS_TRY(exchange(rc, &ca, &cb, nullptr, nullptr));
            gathered(1);
            E->global_pends.assign(gb.begin(), gb.end());
            std::sort(E->global_pends.begin(), E->global_pends.end());
            E->have_global_pends = true;
            rc = engine_slot(c, slot, 2);
            if (rc == NSGPU_OK) engine_claim_requests(c, ca, cb, b);
            NS_TRY(exchange(rc, &ca, &cb, nullptr, nullptr));
            gathered(1);
            engine_claim_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size());
            if (E->n_done_global >= E->n_total) break;
            continue;
        }
        const int rc = engine_slot(c, slot);
        if (rc == NSGPU_OK) { engine_claim_requests(c, ca, cb, b); engine_seed_requests(c, sa, sb, h); }
        NS_TRY(exchange(rc, &ca, &cb, &sa, &sb));
        gathered(1);
        engine_claim_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size());
        gathered(1 + blk);
        if (!ga.empty() && engine_seed_resolve(c, ga.data(), gb.data(), (uint32_t)ga.size()) != 0) {
            if (G < 4) engine_advance(c, true, h);       // (see run_consensus)
            else E->deferred_fresh = h;
        }
        if (E->n_done_global >= E->n_total) break;
    }
    if (n_coll_out) *n_coll_out = n_coll;
    if (bytes_out) *bytes_out = n_coll * (uint64_t)words * 4 * W;
    return engine_finish(c, n_threads_out);
}

extern "C" {

int nsgpu_dist_consensus_run(nsgpu_ctx *c, nsgpu_comm *comm, uint32_t n_builders_total, uint32_t n_threads_out, nsgpu_consensus_stats *stats_out)
{
    NS_CHECK(c && comm && nsgpu_comm_impl(comm), NSGPU_ERR_ARG, "nsgpu_dist_consensus_run: null argument");
    NS_CHECK(c->have_index, NSGPU_ERR_ARG, "nsgpu_dist_consensus_run: build the bucket tables first (nsgpu_dist_sketch_index)");
    NS_HIP(hipSetDevice(c->prm.device));
    uint64_t n_coll = 0, bytes = 0;
    const int rc = run_consensus_dist(c, *nsgpu_comm_impl(comm), n_builders_total, n_threads_out, &n_coll, &bytes);
    if (rc != NSGPU_OK) { if (c->cons_engine) { c->cons_engine_free(c->cons_engine); c->cons_engine = nullptr; } return rc; }
    nsgpu_comm_count(comm, bytes, 0);
    c->cons_stats.reserved = (uint32_t)n_coll;           // collectives of the stage
    if (stats_out) *stats_out = c->cons_stats;
    return NSGPU_OK;
}

}  // extern "C"
