// consensus_debug.hip -- NSGPU_CONS_DEBUG: where the wall time, the CPU time and the memory of a contig stage went (printed once per stage).
#include "engine.hpp"

namespace nsgpu {
namespace mm2 { extern std::atomic<uint64_t> g_step_ns[6]; }
extern double g_finish_ms[5];
extern uint64_t g_dp_shape[5][4][8];
extern double g_sketch_ms[6];
namespace cons { extern std::atomic<uint64_t> g_emit_ns[4]; extern std::atomic<uint64_t> g_mp_cnt[3]; extern std::atomic<int64_t> g_slabs_in_use, g_slabs_peak, g_slabs_mapped; }


// NSGPU_CONS_DEBUG: where the wall time of the slots went (printed once per stage, before the edit emission is waited for)
void debug_report_slots(nsgpu_ctx *c, Engine *E)
{
    fprintf(stderr, "[cons] batches wall-ms: window queries %.0f, sketch+index %.0f (gpu sketch %.0f), align begin %.0f (host loops %.0f, DP launch %.0f)\n", c->cons_stats.filter_ms,
            c->cons_stats.index_ms, c->sketch_mm_ms, E->p1_align_ms, E->p1_host_ms, E->p1_launch_ms);
    double chain_ms = 0;
    for (AlignBatch &ab : E->ab) chain_ms += ab.chain_ms, ab.chain_ms = 0;
    double sw = 0; uint64_t sn = 0, sp = 0, sf = 0;
    for (nsgpu_ctx::SeedWs &w : c->seed_ws) sw += w.ms_wait, sn += w.calls, sp += w.pairs, sf += w.fallbacks, w.ms_wait = 0, w.calls = w.pairs = w.fallbacks = 0;
    fprintf(stderr, "[cons] window-query batches redone the exact multi-step way (a buffer sized in advance did not fit): %llu\n", (unsigned long long)E->n_wq_exact);
    fprintf(stderr, "[cons] DP launches by register class (cumulative; ms per launch x launches; wall of the DP phases %.0f ms):", c->ksw_kernel_ms);
    for (int k = 0; k < 16; ++k) if (c->ksw_class_n[k]) fprintf(stderr, " [%d] %.3f x %llu", k, c->ksw_class_ms[k] / (double)c->ksw_class_n[k], (unsigned long long)c->ksw_class_n[k]);
    fprintf(stderr, "\n");
    fprintf(stderr, "[cons] alignments left to the host's plan, by reason (cumulative): no anchors / flagged pair %llu, several chains %llu, seed filtering %llu, outside the staged span %llu, capacity %llu, DP class %llu\n",
            (unsigned long long)c->plan_why[0], (unsigned long long)c->plan_why[1], (unsigned long long)c->plan_why[2], (unsigned long long)c->plan_why[3], (unsigned long long)c->plan_why[4], (unsigned long long)c->plan_why[5]);
    {
        static const char *kind[5] = {"gap fill", "left ext (query >= target)", "right ext (query >= target)", "left ext (target longer)", "right ext (target longer)"};
        static const char *wd[4] = {"<=256", "<=512", "<=1536", ">1536"};
        fprintf(stderr, "[cons] alignments by their longest DP problem (anti-diagonals < 256 / 512 / 768 / 1024 / 1536 / 2048 / 3072 / more):\n");
        for (int k = 0; k < 5; ++k) for (int w = 0; w < 4; ++w) {
            uint64_t tot = 0; for (int b = 0; b < 8; ++b) tot += g_dp_shape[k][w][b];
            if (!tot) continue;
            fprintf(stderr, "[cons]   %-28s target %-6s:", kind[k], wd[w]);
            for (int b = 0; b < 8; ++b) fprintf(stderr, " %llu", (unsigned long long)g_dp_shape[k][w][b]);
            fprintf(stderr, "\n");
        }
    }
    fprintf(stderr, "[cons] early tasks: loops of part 0 / part 1 %.0f / %.0f ms wall; tasks %.0f ms in sum (delivery + skeleton + conversion %.0f), the longest of each slot %.0f ms in sum\n",
            E->early_part_ms[0], E->early_part_ms[1], E->early_task_ms, E->early_conv_ms, E->early_task_max_ms);
    fprintf(stderr, "[cons] graph updates run ahead of the slot's end: %llu, %.0f ms wall (DP results + updates); status words seen ahead of their data: %llu; builders' tasks taken over by an idle thread: %llu; the slots' last alignments were taken up %.0f ms into their watches (in sum, of %.0f ms)\n", (unsigned long long)E->n_early, E->early_ms, (unsigned long long)E->n_early_retry, (unsigned long long)E->n_early_stolen, E->early_last_claim_ms, E->early_part_ms[0]);
    fprintf(stderr, "[cons] device plan (cumulative): %llu alignments planned on the device, %llu left to the host; DP problems found %llu, not found %llu, unasked %llu\n",
            (unsigned long long)c->plan_pairs_dev, (unsigned long long)c->plan_pairs_host, (unsigned long long)c->plan_hits, (unsigned long long)c->plan_misses, (unsigned long long)c->plan_extra);
    fprintf(stderr, "[cons] index + seeds on the GPU: %llu launches, %llu pairs, wait %.0f ms wall; %llu pairs handed back to the host code\n", (unsigned long long)sn,
            (unsigned long long)sp, sw, (unsigned long long)sf);
    {
        uint64_t lf[5] = {}, la = 0, ll = 0, lc = 0; double ls = 0, lw = 0;
        for (nsgpu_ctx::SeedWs &w : c->seed_ws) {
            for (int b = 0; b < 5; ++b) lf[b] += w.late_flag[b], w.late_flag[b] = 0;
            la += w.late_anchors, ll = std::max(ll, w.late_longest), lc += w.late_calls, ls += w.late_seed_ms, lw += w.late_chain_ms;
            w.late_anchors = w.late_longest = w.late_calls = 0, w.late_seed_ms = w.late_chain_ms = 0;
        }
        if (lc) fprintf(stderr, "[cons] pairs handed back: ties %llu, many %llu, capacity %llu, occ %llu, wide %llu; %llu anchors in sum, longest list %llu; %llu second launches: host index + seeds %.0f ms, chaining wait %.0f ms\n",
                        (unsigned long long)lf[0], (unsigned long long)lf[1], (unsigned long long)lf[2], (unsigned long long)lf[3], (unsigned long long)lf[4], (unsigned long long)la, (unsigned long long)ll,
                        (unsigned long long)lc, ls, lw);
    }
    fprintf(stderr, "[cons] count tables: %llu rebuilt (%llu slots cleared in sum, the longest list %llu), %llu updated in place (%llu hashes removed + added)\n", (unsigned long long)E->dbg_cnt_rebuilds,
            (unsigned long long)E->dbg_cnt_rebuild_slots, (unsigned long long)E->dbg_cnt_rebuild_max, (unsigned long long)E->dbg_cnt_updates, (unsigned long long)E->dbg_cnt_keys);
    if (c->defer_slots) fprintf(stderr, "[cons] deferred alignments (more than %u anchors: %u more slots): %llu; their batches ran %.0f ms in sum beside the slots, the slots waited %.0f ms for them\n",
                                c->defer_anchors, c->defer_slots, (unsigned long long)E->n_deferred, E->defer_run_ms, E->defer_join_ms);
    double cs = 0, ce = 0, cw = 0; uint64_t cn = 0;
    for (nsgpu_ctx::ChainWs &w : c->cws) cs += w.ms_stage, ce += w.ms_enqueue, cw += w.ms_wait, cn += w.calls, w.ms_stage = w.ms_enqueue = w.ms_wait = 0, w.calls = 0;
    fprintf(stderr, "[cons] chaining scores on the GPU (inside sketch+index): %.0f ms wall in %llu calls: staging %.0f, enqueue %.0f, wait (copies + kernel) %.0f\n", chain_ms,
            (unsigned long long)cn, cs, ce, cw);
    if (!E->dbg_batch_sizes.empty()) {        // alignments per batch over the run, in tenths of the run
        const size_t nb = E->dbg_batch_sizes.size();
        fprintf(stderr, "[cons] alignments per batch over the run (%zu batches, mean of each tenth):", nb);
        for (int d = 0; d < 10; ++d) { uint64_t sum = 0; const size_t a = nb * d / 10, b = nb * (d + 1) / 10; for (size_t i = a; i < b; ++i) sum += E->dbg_batch_sizes[i]; fprintf(stderr, " %.1f", b > a ? (double)sum / (double)(b - a) : 0.0); }
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "[cons] one-group slot, wall-ms of its steps: host phase %.0f, windows %.0f, seeds + fresh contigs %.0f, sketch..chain %.0f, DP launch %.0f, results + early updates %.0f, last results %.0f\n",
            E->g1_ms[0], E->g1_ms[1], E->g1_ms[2], E->g1_ms[3], E->g1_ms[4], E->g1_ms[6], E->g1_ms[7]);
    for (int l = 0; l < kMaxGroups; ++l) {
        const double *m = E->lane[l].sk_ms;
        if (m[0] + m[1] + m[2] + m[3] + m[4] + m[5] > 0)
            fprintf(stderr, "[cons] sketch..chain of lane / group %d, wall-ms of its steps: splice plan %.0f, requests %.0f, sketch call %.0f, splice + index loop %.0f, enqueue of tails / seeds / chain / plan / DP %.0f, wait for seeds + chains and the first step %.0f\n",
                    l, m[0], m[1], m[2], m[3], m[4], m[5]);
    }
    fprintf(stderr, "[cons] slots set by: host phase %llu (%.0f ms), batches part 1 %llu (%.0f ms), part 2 %llu (%.0f ms)\n", (unsigned long long)E->slot_long_n[0],
            E->slot_long_ms[0], (unsigned long long)E->slot_long_n[1], E->slot_long_ms[1], (unsigned long long)E->slot_long_n[2], E->slot_long_ms[2]);
}

// NSGPU_CONS_DEBUG: CPU time, memory and the per-step counters of the whole stage
void debug_report_stage(nsgpu_ctx *c, Engine *E, const struct rusage &ru0, double w_begin, double w_slot, double w_seed, double w_claim, double tf)
{
    fprintf(stderr, "[cons] gpu mm_sketch wall-ms %.0f\n", c->sketch_mm_ms);
    if (E->D.gsh) {
        DevGraphShared &G = *E->D.gsh;
        uint64_t splits = 0, seq = 0;
        for (const Builder &b : E->D.B) splits += b.dbg_g[0], seq += b.dbg_g[1];
        fprintf(stderr, "[cons] consensus graphs in HBM: %llu updates in %llu launches%s (removeCycles run again behind a split that did not fit: %llu), %llu array growths, %llu new stretches longer than a report (copied), host waited %.0f ms for first and %.0f ms for second reports in sum; "
                        "%.2f GB of finished contigs copied back; splitPath calls %llu, excursions taken one at a time %llu; pool: HBM peak %.2f GB of %.2f GB mapped, pinned peak %.2f GB of %.2f GB\n",
                (unsigned long long)G.n_updates.load(), (unsigned long long)G.n_launches.load(), G.check ? ", every one checked against the host's arrays" : "", (unsigned long long)G.n_regrow.load(), (unsigned long long)G.n_grow.load(), (unsigned long long)G.n_mid_copies.load(),
                G.kernel_wait_ns.load() / 1e6, G.final_wait_ns.load() / 1e6, G.bytes_back.load() / 1e9, (unsigned long long)splits, (unsigned long long)seq, G.dev.peak() / 1e9, G.dev.mapped() / 1e9, G.pin.peak() / 1e9, G.pin.mapped() / 1e9);
        fprintf(stderr, "[cons] graph kernels, ms in sum by phase (their own clock): tables %.0f, runs %.0f, excursions %.0f, choices %.0f, stitching %.0f, writing %.0f, flags + kept ends %.0f, removeCycles %.0f; launch to report %.0f ms in sum\n",
                G.phase_ticks[0].load() / 1e5, G.phase_ticks[1].load() / 1e5, G.phase_ticks[2].load() / 1e5, G.phase_ticks[3].load() / 1e5, G.phase_ticks[4].load() / 1e5, G.phase_ticks[5].load() / 1e5,
                G.phase_ticks[6].load() / 1e5, G.phase_ticks[7].load() / 1e5, G.update_ns.load() / 1e6);
        fprintf(stderr, "[cons] graph kernels, thread 0's loops: %llu path entries looked at in rejoin searches, %llu detour steps over old side nodes, %llu read ids compared in %llu splitPath contexts, %llu of them made by %llu splits by routes\n",
                (unsigned long long)G.cnt[0].load(), (unsigned long long)G.cnt[1].load(), (unsigned long long)G.cnt[2].load(), (unsigned long long)G.cnt[3].load(), (unsigned long long)G.cnt[5].load(), (unsigned long long)G.cnt[4].load());
        fprintf(stderr, "[cons] graph kernels, removeCycles in parts (ms in sum): marking %.0f, finding the roots %.0f, walks + splits %.0f of which splitPath %.0f (looking for stretches %.0f, stretches and routes by the team %.0f; the splits by routes: walking %.0f, comparing %.0f, copies %.0f, the rest %.0f)\n",
                G.cyc[0].load() / 1e5, G.cyc[1].load() / 1e5, G.cyc[5].load() / 1e5, G.cyc[4].load() / 1e5, G.cyc[2].load() / 1e5, G.cyc[3].load() / 1e5, G.rt[0].load() / 1e5, G.rt[1].load() / 1e5, G.rt[2].load() / 1e5, G.rt[3].load() / 1e5);
        fprintf(stderr, "[cons] graph kernels by duration (< 0.25 / 0.5 / 1 / 2 / 4 / 8 / 16 ms / more): %llu %llu %llu %llu %llu %llu %llu %llu; those of 2 ms and more by their longest phase (tables / runs / excursions / choices / stitching / writing / flags / removeCycles): %llu %llu %llu %llu %llu %llu %llu %llu\n",
                (unsigned long long)G.hist[0].load(), (unsigned long long)G.hist[1].load(), (unsigned long long)G.hist[2].load(), (unsigned long long)G.hist[3].load(), (unsigned long long)G.hist[4].load(), (unsigned long long)G.hist[5].load(),
                (unsigned long long)G.hist[6].load(), (unsigned long long)G.hist[7].load(), (unsigned long long)G.slow_phase[0].load(), (unsigned long long)G.slow_phase[1].load(), (unsigned long long)G.slow_phase[2].load(),
                (unsigned long long)G.slow_phase[3].load(), (unsigned long long)G.slow_phase[4].load(), (unsigned long long)G.slow_phase[5].load(), (unsigned long long)G.slow_phase[6].load(), (unsigned long long)G.slow_phase[7].load());
    }
    if (cons::g_mp_cnt[1].load())
        fprintf(stderr, "[cons] main path (cumulative): %llu recomputes cut the path, on average at %.0f edges before its end of %.0f\n", (unsigned long long)cons::g_mp_cnt[1].load(),
                (double)cons::g_mp_cnt[0].load() / cons::g_mp_cnt[1].load(), (double)cons::g_mp_cnt[2].load() / cons::g_mp_cnt[1].load());
    fprintf(stderr, "[cons] emission cpu-ms: path tables %.0f, read walks %.0f, script folding + stream bytes %.0f\n", cons::g_emit_ns[0] / 1e6, cons::g_emit_ns[1] / 1e6, cons::g_emit_ns[2] / 1e6);
    fprintf(stderr, "[cons] align host cpu-ms: seeds %.0f chain %.0f regs %.0f plan %.0f execute %.0f\n", mm2::g_step_ns[0] / 1e6, mm2::g_step_ns[1] / 1e6,
            mm2::g_step_ns[2] / 1e6, mm2::g_step_ns[3] / 1e6, mm2::g_step_ns[4] / 1e6);
    struct rusage ru1;
    getrusage(RUSAGE_SELF, &ru1);
    const double cpu_s = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) * 1e-6 +
                         (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-6;
    fprintf(stderr, "[cons] serial CPU ms of the role threads (outside pool loops): host %.0f, batches part 1 %.0f, part 2 %.0f\n", E->role_serial_ns[0] / 1e6,
            E->role_serial_ns[1] / 1e6, E->role_serial_ns[2] / 1e6);
    if (FILE *f = fopen("/proc/self/smaps_rollup", "r")) {       // are the graph slabs really on huge pages?
        char line[256];
        long rss = 0, thp = 0;
        while (fgets(line, sizeof(line), f)) { sscanf(line, "Rss: %ld kB", &rss); sscanf(line, "AnonHugePages: %ld kB", &thp); }
        fclose(f);
        fprintf(stderr, "[cons] resident %.1f GB, of it on transparent huge pages %.1f GB; graph slabs of %zu KB: %lld in use, peak %lld, carved %lld (%.1f GB)\n", rss / 1048576.0, thp / 1048576.0,
                cons::kSlabBytes >> 10, (long long)cons::g_slabs_in_use.load(), (long long)cons::g_slabs_peak.load(), (long long)cons::g_slabs_mapped.load(),
                cons::g_slabs_mapped.load() * (double)cons::kSlabBytes / (1u << 30));
    }
    pool_prof_print();
    fprintf(stderr, "[cons] part 2 wall-ms: wait for the DP in flight %.0f, later rounds %.0f (their DP %.0f, %d rounds), results %.0f\n", g_finish_ms[0], g_finish_ms[1],
            g_finish_ms[2], (int)g_finish_ms[4], g_finish_ms[3]);
    for (double &x : g_finish_ms) x = 0;
    fprintf(stderr, "[cons] gpu mm_sketch wall-ms: host staging %.0f, flags..k-mers (1st read-back) %.0f, pushes..offsets (2nd) %.0f, write + read-back %.0f; %.0f MB in, %.1f M minimizers out\n",
            g_sketch_ms[0], g_sketch_ms[1], g_sketch_ms[2], g_sketch_ms[3], g_sketch_ms[4] / 1e6, g_sketch_ms[5] / 1e6);
    for (double &x : g_sketch_ms) x = 0;
    const double sys_s = (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-6;
    fprintf(stderr, "[cons] process CPU time over the stage: %.1f s (%.1f s of it in the kernel; %ld minor faults) = %.1f cores busy on average\n", cpu_s, sys_s,
            ru1.ru_minflt - ru0.ru_minflt, cpu_s / ((now_ms() - E->t0) * 1e-3));
    fprintf(stderr, "[cons] wall-ms: begin %.0f slots %.0f (host phases %.0f, batches %.0f, overlapped) seed %.0f claim %.0f finish %.0f\n", w_begin, w_slot,
            c->cons_stats.graph_ms, c->cons_stats.filter_ms + c->cons_stats.index_ms + c->cons_stats.align_ms, w_seed, w_claim, now_ms() - tf);
}

}  // namespace nsgpu
