// align_batch.hip -- a13: batched ConsensusGraph::alignRead (src/ConsensusGraph.cpp:161-398).
//
// The reference aligns ONE candidate at a time and rebuilds the minimizer index of the whole
// consensus for each of them.  Here a batch of (reference, query) pairs is processed together:
//   * one index per distinct reference string of the batch (host threads),
//   * seeds / chaining / region bookkeeping per pair on host threads (mm2.cpp),
//   * every banded DP of every pair gathered into ONE launch of the ksw_extd2 wavefront
//     kernel per round (ksw2.hip); a pair needs 1-3 rounds (first pass, Z-drop second pass,
//     split regions),
//   * CIGAR -> edit script conversion per pair.
#include "common.hpp"
#include "ksw2.hpp"
#include "mm2.hpp"
#include <atomic>
#include <thread>
#include <chrono>
#include <functional>
#include <mutex>
#include <condition_variable>
#include "host_util.hpp"
#include "ksw_collect.hpp"
#include <sched.h>
#include <pthread.h>
#include <cctype>
#include <memory>
#include <deque>

namespace nsgpu {

// The host side of the contig stage chases pointers through graphs it allocated itself: on a two-socket box half of those
// accesses are remote unless the threads stay on the NUMA node the GPU hangs off (first-touch then puts the graphs there
// too).  Measured on the 2 x 64-core host of an MI355X box: +5 % and a steadier step time; binding tighter (2 or 4 CCDs)
// loses 10-25 % because the threads then share cores.  NSGPU_NO_NUMA_BIND=1 leaves the threads alone.
namespace {
std::mutex g_bind_m;
cpu_set_t g_bind_set;
bool g_bind_on = false;

bool parse_cpulist(const char *s, cpu_set_t &set)
{
    CPU_ZERO(&set);
    bool any = false;
    while (*s) {
        char *e;
        long a = strtol(s, &e, 10), b = a;
        if (e == s) break;
        if (*e == '-') { s = e + 1; b = strtol(s, &e, 10); if (e == s) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &set); any = true; }
        s = *e == ',' ? e + 1 : e;
        if (*s == '\n') break;
    }
    return any;
}
}  // namespace

void pool_bind_this_thread()
{
    std::lock_guard<std::mutex> lk(g_bind_m);
    if (g_bind_on) (void)pthread_setaffinity_np(pthread_self(), sizeof(g_bind_set), &g_bind_set);
}

void pool_rebind_workers();

// called once per context with the GPU's PCI address ("0000:xx:yy.z")
void pool_bind_to_gpu_node(const char *pci_bus_id)
{
    static const bool off = getenv("NSGPU_NO_NUMA_BIND") != nullptr;
    if (off || !pci_bus_id) return;
    char path[256], buf[4096];
    std::string id(pci_bus_id);
    for (char &c : id) c = (char)tolower((unsigned char)c);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id.c_str());
    int node = -1;
    if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return;
    const bool got = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    cpu_set_t node_set, cur, both;
    if (!got || !parse_cpulist(buf, node_set)) return;
    if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return;
    CPU_AND(&both, &node_set, &cur);                      // never leave the CPUs the process was given
    if (CPU_COUNT(&both) < (int)host_threads()) return;   // too few to be worth it (a tight cpuset is somebody's decision already)
    {
        std::lock_guard<std::mutex> lk(g_bind_m);
        if (g_bind_on) return;                            // one binding per process: the first context's GPU
        g_bind_set = both, g_bind_on = true;
    }
    if (getenv("NSGPU_CONS_DEBUG")) fprintf(stderr, "[pool] host threads bound to NUMA node %d of GPU %s: %d CPUs\n", node, id.c_str(), CPU_COUNT(&both));
    pool_rebind_workers();
}

unsigned host_threads()
{
    static unsigned n = [] {
        unsigned v = std::thread::hardware_concurrency();
        // honour a cgroup CPU quota (containers): more runnable threads than quota only adds throttling
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64]; long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long long quota = atoll(q);
                const unsigned lim = (unsigned)((quota + period - 1) / period);
                if (lim >= 1 && lim < v) v = lim;
            }
            fclose(f);
        }
        // one process per GPU: the ranks of a node share the cores (torchrun exports LOCAL_WORLD_SIZE)
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) { const int l = atoi(e); if (l > 1) v = v / (unsigned)l ? v / (unsigned)l : 1; }
        if (const char *e = getenv("NSGPU_THREADS")) { int x = atoi(e); if (x > 0) v = (unsigned)x; }
        if (v == 0) v = 8;
        if (v > 256) v = 256;
        return v;
    }();
    return n;
}

void parallel_for_impl(size_t n, const std::function<void(size_t)> &fn);
template <class F>
static void parallel_for(size_t n, F fn) { parallel_for_impl(n, std::function<void(size_t)>(fn)); }
template <class F>
static void parallel_for(const char *tag, size_t n, F fn) { PoolTag t(tag); parallel_for_impl(n, std::function<void(size_t)>(fn)); }

// Persistent host thread pool.  Several jobs may be in flight at once (the contig engine's host phase and the host
// parts of the other group's GPU batches are submitted from two threads): a job is an index range handed out in chunks
// through an atomic cursor (or, "pinned", one queue per thread with stealing), workers serve the short jobs of the
// batch thread first so that the GPU is not kept waiting behind a long host phase, and every submitter works on its
// own job while it waits.  (Spawning ~256 std::threads per parallel loop cost more than the loops themselves.)
// NSGPU_CONS_DEBUG: thread-CPU time of the pool's loops by the submitter's tag (pool_tag), printed by the contig engine
static const bool g_pool_prof = getenv("NSGPU_CONS_DEBUG") != nullptr;
static const char *g_tag_name[32];
static std::atomic<uint64_t> g_tag_ns[32];
static std::atomic<int> g_n_tags{1};
static thread_local int tl_tag = 0;
static thread_local uint64_t tl_work_ns = 0;
static inline uint64_t thread_cpu_ns() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
int pool_tag(const char *name)
{
    const int prev = tl_tag;
    if (!name) { tl_tag = 0; return prev; }
    static std::mutex m;
    std::lock_guard<std::mutex> lk(m);
    int n = g_n_tags.load();
    for (int i = 1; i < n; ++i) if (!strcmp(g_tag_name[i], name)) { tl_tag = i; return prev; }
    if (n < 32) { g_tag_name[n] = name; g_n_tags.store(n + 1); tl_tag = n; }
    return prev;
}
void pool_tag_restore(int t) { tl_tag = t; }
uint64_t pool_thread_cpu_ns() { return thread_cpu_ns(); }
uint64_t pool_thread_work_ns() { return tl_work_ns; }
size_t pool_bg_max();
void pool_prof_print()
{
    if (!g_pool_prof) return;
    g_tag_name[0] = "untagged";
    fprintf(stderr, "[pool] thread-CPU ms by loop:");
    for (int i = 0; i < g_n_tags.load(); ++i) { fprintf(stderr, " %s %.0f", g_tag_name[i], g_tag_ns[i].exchange(0) / 1e6); }
    fprintf(stderr, " background %.0f", g_tag_ns[31].exchange(0) / 1e6);
    fprintf(stderr, "; longest background queue %zu tasks\n", pool_bg_max());
}

namespace {
class HostPool {
    struct Job {
        int tag = 0;
        const std::function<void(size_t)> *fn = nullptr;
        size_t total = 0, chunk = 1;
        bool pinned = false;
        unsigned n_threads = 1;
        std::atomic<size_t> next{0}, done{0};
        std::vector<std::atomic<size_t>> pos;          // pinned: items taken from thread v's queue (item = v + j * n_threads)
        std::mutex m;
        std::condition_variable cv;
        // runs one chunk; false when the job has nothing left to hand out
        bool work_once(unsigned me)
        {
            if (!g_pool_prof) return work_once_(me);
            const uint64_t t0 = thread_cpu_ns();
            const bool r = work_once_(me);
            const uint64_t d = thread_cpu_ns() - t0;
            g_tag_ns[tag].fetch_add(d, std::memory_order_relaxed);
            tl_work_ns += d;
            return r;
        }
        bool work_once_(unsigned me)
        {
            size_t n_done = 0;
            if (pinned) {
                bool got = false;
                for (unsigned k = 0; k < n_threads && !got; ++k) {
                    const unsigned v = (me + k) % n_threads;
                    if ((size_t)v + pos[v].load(std::memory_order_relaxed) * n_threads >= total) continue;     // cheap pre-check
                    const size_t j = pos[v].fetch_add(1, std::memory_order_relaxed);
                    const size_t i = (size_t)v + j * n_threads;
                    if (i >= total) continue;
                    (*fn)(i);
                    n_done = 1, got = true;
                }
                if (!got) return false;
            } else {
                if (next.load(std::memory_order_relaxed) >= total) return false;
                const size_t b = next.fetch_add(chunk);
                if (b >= total) return false;
                const size_t e = b + chunk < total ? b + chunk : total;
                for (size_t i = b; i < e; ++i) (*fn)(i);
                n_done = e - b;
            }
            if (done.fetch_add(n_done) + n_done == total) { std::lock_guard<std::mutex> lk(m); cv.notify_all(); }
            return true;
        }
    };
public:
    explicit HostPool(unsigned n) : n_(n) { for (unsigned i = 1; i < n_; ++i) th_.emplace_back([this, i] { worker(i); }); }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_a_.store(++gen_, std::memory_order_release); }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // pinned = true: index i is queued for thread i % n_threads (keeps a builder's graph in one core's caches and its
    // slabs in one thread's cache); a thread that runs out of its own items steals from the others
    void run(size_t n, const std::function<void(size_t)> &fn, bool pinned = false)
    {
        auto job = std::make_shared<Job>();
        job->fn = &fn, job->total = n, job->pinned = pinned, job->n_threads = n_, job->tag = tl_tag;
        job->chunk = n / ((size_t)n_ * 8) ? n / ((size_t)n_ * 8) : 1;
        if (pinned) { job->pos = std::vector<std::atomic<size_t>>(n_); for (auto &x : job->pos) x.store(0); }
        {
            std::lock_guard<std::mutex> lk(m_);
            // short (unpinned) jobs go to the front: they are the ones a GPU batch is waiting for
            if (pinned) active_.push_back(job); else active_.insert(active_.begin(), job);
            gen_a_.store(++gen_, std::memory_order_release);
        }
        cv_.notify_all();
        while (job->work_once(0)) {}
        spin_until([&] { return job->done.load(std::memory_order_acquire) == job->total; });      // the last chunks are a few microseconds away
        {
            std::unique_lock<std::mutex> lk(job->m);
            job->cv.wait(lk, [&] { return job->done.load() == job->total; });
        }
        std::lock_guard<std::mutex> lk(m_);
        active_.erase(std::find(active_.begin(), active_.end(), job));
    }
    void rebind()
    {
        std::lock_guard<std::mutex> lk(g_bind_m);
        if (g_bind_on) for (auto &t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(g_bind_set), &g_bind_set);
    }
    // Background work of the lowest priority: run by workers that find nothing to do in any parallel loop, never waited for
    // except by drain().
    void post(std::function<void()> fn)
    {
        if (th_.empty()) { fn(); return; }
        { std::lock_guard<std::mutex> lk(m_); bg_.push_back(std::move(fn)); if (bg_.size() > bg_max_) bg_max_ = bg_.size(); }
        cv_.notify_one();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(m_);
        bg_cv_.wait(lk, [this] { return bg_.empty() && bg_running_ == 0; });
    }
private:
    void worker(unsigned me)
    {
        pool_bind_this_thread();
        uint64_t seen = 0;
        std::vector<std::shared_ptr<Job>> snap;
        for (;;) {
            std::function<void()> bg;
            // The loops of a slot of the contig engine follow each other within microseconds, and a sleeping worker needs tens of them to
            // wake up (a futex round trip per loop and worker, on the slot's critical path): a worker that ran out of work keeps looking
            // for the next loop for a short while before it blocks.  NSGPU_POOL_SPIN_US (default 60; 0 = block at once).
            spin_until([&] { return gen_a_.load(std::memory_order_acquire) != seen; });
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen || stop_ || !bg_.empty(); });
                if (stop_) return;
                if (gen_ == seen) {                       // no new loop: take a background task
                    bg = std::move(bg_.front());
                    bg_.pop_front();
                    ++bg_running_;
                } else {
                    seen = gen_;
                    snap = active_;
                }
            }
            if (bg) {
                const uint64_t t0 = g_pool_prof ? thread_cpu_ns() : 0;
                bg();
                if (g_pool_prof) g_tag_ns[31].fetch_add(thread_cpu_ns() - t0, std::memory_order_relaxed);
                std::lock_guard<std::mutex> lk(m_);
                if (--bg_running_ == 0 && bg_.empty()) bg_cv_.notify_all();
                continue;
            }
            // serve until no job in the snapshot has work left; a job submitted meanwhile changes gen_ and is seen next
            for (bool any = true; any;) {
                any = false;
                for (auto &j : snap) {
                    if (j->work_once(me)) { any = true; break; }      // re-scan from the front: short jobs first
                }
                if (any) {
                    std::lock_guard<std::mutex> lk(m_);
                    if (gen_ != seen) { seen = gen_; snap = active_; }
                }
            }
            snap.clear();
        }
    }
    template <class P> static void spin_until(P ready)
    {
        static const long spin_us = getenv("NSGPU_POOL_SPIN_US") ? atol(getenv("NSGPU_POOL_SPIN_US")) : 60;
        if (spin_us <= 0 || ready()) return;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 1;; ++it) {
            __builtin_ia32_pause();
            if (ready()) return;
            if ((it & 31) == 0 && std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us) return;
        }
    }
    unsigned n_;
    std::atomic<uint64_t> gen_a_{0};                  // gen_, readable without the lock (spin_until)
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, bg_cv_;
    std::deque<std::function<void()>> bg_;
    unsigned bg_running_ = 0;
public:
    size_t bg_max_ = 0;                               // longest the background queue has been (debug print)
private:
    std::vector<std::shared_ptr<Job>> active_;
    uint64_t gen_ = 0;
    bool stop_ = false;
};
}  // namespace

static HostPool &the_pool() { static HostPool pool(host_threads()); return pool; }
size_t pool_bg_max() { if (host_threads() <= 1) return 0; const size_t m = the_pool().bg_max_; the_pool().bg_max_ = 0; return m; }
void pool_rebind_workers() { if (host_threads() > 1) the_pool().rebind(); }

void parallel_for_impl(size_t n, const std::function<void(size_t)> &fn)
{
    if (n == 0) return;
    if (n == 1 || host_threads() <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    the_pool().run(n, fn);
}

void pool_post(std::function<void()> fn)
{
    if (host_threads() <= 1) { fn(); return; }
    the_pool().post(std::move(fn));
}
void pool_drain() { if (host_threads() > 1) the_pool().drain(); }

void parallel_for_pinned_impl(size_t n, const std::function<void(size_t)> &fn)
{
    if (n == 0) return;
    if (host_threads() <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    the_pool().run(n, fn, true);
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One alignment request: query against a reference whose index the caller owns.
// The batch as a two-part job (align_begin / align_finish), so that the contig engine can leave the DP kernels of one
// builder group in flight while it works on another group; align_requests is both parts back to back.
namespace {

// the aligner never reads the score of a gap fill (mm_align1 adds it into dp_score, which nothing on NanoSpring's path looks at): the DP kernels
// may skip the books that only produce it (KSW_EZ_NS_NO_SCORE, ksw2_reg.hip).  NSGPU_KSW_KEEP_SCORE=1: computed as before (A/B switch).
// ... nor the best score on the target's last column of an extension (mm_align1 reads max / max_t / max_q, mqe_t, reach_end and the CIGAR): the
// register kernels may end the sweep of an extension whose target is longer than its query once nothing but ez.mte can still change
// (KSW_EZ_NS_NO_MTE, ksw2_reg.hip).  NSGPU_KSW_KEEP_SCORE=1 keeps both (A/B switch, tests).
int approx_task_flag(int flag)
{
    static const bool keep = getenv("NSGPU_KSW_KEEP_SCORE") != nullptr;
    if (keep) return 0;
    if ((flag & 0x08) && !(flag & 0x10)) return 0x80000;
    if ((flag & 0x40) && !(flag & 0x08)) return 0x100000;
    return 0;
}

mm2::Opt batch_opt(const nsgpu_ctx *c)
{
    mm2::Opt opt;
    opt.k = (int)c->prm.m_k, opt.w = (int)c->prm.m_w, opt.max_chain_iter = (int)c->prm.max_chain_iter;
    return opt;
}
KswParams batch_ksw_params(const mm2::Opt &opt)
{
    KswParams kp;
    kp.sc_mch = opt.a; kp.sc_mis = -opt.b; kp.sc_ambi = -opt.sc_ambi; kp.q = opt.q; kp.e = opt.e; kp.q2 = opt.q2; kp.e2 = opt.e2;
    return kp;
}

// ---- the alignment plan on the device (plan.hip) -----------------------------------------------------------------------------------
// Job j's plan pass has listed the DP problems it needs (cache.missing); those the plan kernel launched for it are on their way already.
void plan_match(nsgpu_ctx *c, AlignBatch &B, uint32_t j)
{
    using namespace mm2;
    if (j >= B.plan_pair.size() || B.plan_pair[j] == ~0u) return;
    const PlanOut po = B.plan_out.as<PlanOut>()[B.plan_pair[j]];
    AlignJob &J = B.jobs[j];
    if (po.flags || po.n_tasks == 0) return;
    const PlanKey *keys = B.plan_keys.as<PlanKey>() + B.plan_base[j];
    static_assert(sizeof(PlanKey) == sizeof(DpKey), "the device plan's keys are DpKeys");
    uint64_t hits = 0, used = 0;
    std::vector<DpKey> &m = J.cache.missing;
    size_t keep = 0;
    for (size_t i = 0; i < m.size(); ++i) {
        bool found = false;
        for (uint32_t t = 0; t < po.n_tasks && !found; ++t) found = memcmp(&keys[t], &m[i], sizeof(DpKey)) == 0;
        if (found) ++hits; else m[keep++] = m[i];
    }
    const uint64_t misses = keep;
    m.resize(keep);
    used = hits;
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->plan_hits += hits, c->plan_misses += misses, c->plan_extra += po.n_tasks - used;
}

// PlanPair of every seeding pair, room for the tasks, the plan kernel behind the chaining kernel on the seeding stream, the DP classes behind it
int batch_plan_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int seed_ws, int dp_ws)
{
    using namespace mm2;
    static const bool off = getenv("NSGPU_NO_DEVICE_PLAN") != nullptr;          // A/B switch: every DP problem planned by the host, as before
    B.plan_ws = -1;
    const size_t n = hi - lo;
    if (off || n == 0 || lo != 0 || hi != B.reqs.size() || ksw_class_config().off) return NSGPU_OK;
    nsgpu_ctx::SeedWs &S = c->seed_ws[seed_ws];
    const size_t n_pairs = B.seed_pairs.size();
    if (n_pairs == 0) return NSGPU_OK;
    const Opt opt = batch_opt(c);
    const KswParams kp = batch_ksw_params(opt);
    NS_TRY(B.plan_pairs.reserve(n_pairs * sizeof(PlanPair)));
    NS_TRY(B.plan_out.reserve(n_pairs * sizeof(PlanOut)));
    PlanPair *pp = B.plan_pairs.as<PlanPair>();
    memset(pp, 0, n_pairs * sizeof(PlanPair));
    memset(B.plan_out.p, 0, n_pairs * sizeof(PlanOut));
    B.plan_base.assign(n, 0), B.plan_pair.assign(n, ~0u);
    uint64_t slots = 0, seq_bound = 0;
    int max_q = 0;
    bool any = false;
    for (size_t i = 0; i < n; ++i) {
        const AlignReq &r = B.reqs[lo + i];
        const uint32_t q = S.pair_of[i];
        if (q == ~0u || !r.qry_dev || !r.ref_dev || r.qry_len >= (1ull << 31) || r.ref_len >= (1ull << 31)) continue;
        const uint32_t cap = (uint32_t)std::min<uint64_t>(256, r.qry_len / (uint64_t)opt.min_ksw_len + 4);
        pp[q] = PlanPair{r.ref_dev, r.qry_dev, r.ref_dev_lo, r.ref_dev_n, (uint32_t)r.ref_len, (uint32_t)r.qry_len, (uint32_t)slots, cap};
        B.plan_base[i] = (uint32_t)slots, B.plan_pair[i] = q;
        slots += cap;
        seq_bound += 3 * (uint64_t)r.qry_len + 4 * (uint64_t)opt.max_gap + 1024;
        max_q = std::max(max_q, (int)std::min<uint64_t>(r.qry_len, (uint64_t)opt.max_gap + 1024));
        any = true;
    }
    if (!any || seq_bound >= (1ull << 32) - (1u << 20) || slots >= (1u << 24)) return NSGPU_OK;
    NS_TRY(B.plan_keys.reserve(slots * sizeof(PlanKey)));
    const SeedChainDev D = gpu_seeds_chain_dev(c, seed_ws, 2 * seed_ws);
    PlanDp dp;
    NS_TRY(ksw_dev_prepare(c, dp_ws, (uint32_t)slots, (uint32_t)n_pairs, seq_bound, D.stream, dp));
    PlanCfg cfg;
    cfg.k = opt.k, cfg.min_cnt = opt.min_cnt, cfg.min_sc = opt.min_chain_score, cfg.bw = opt.bw, cfg.max_gap = opt.max_gap, cfg.min_ksw_len = opt.min_ksw_len;
    cfg.zdrop = opt.zdrop, cfg.end_bonus = opt.end_bonus, cfg.a = opt.a, cfg.q = opt.q, cfg.e = opt.e, cfg.q_max = max_q;
    cfg.kp = kp, cfg.kc = ksw_class_config();
    cfg.approx_flag_or = approx_task_flag(0x08), cfg.ext_flag_or = approx_task_flag(0x40);
    const bool two_part = B.plan_two_part && cfg.kc.long_rows > 0;
    cfg.two_phase = two_part;
    if (B.plan_wait_ev) NS_HIP(hipStreamWaitEvent(D.stream, B.plan_wait_ev, 0));
    NS_TRY(plan_launch(D.stream, (uint32_t)n_pairs, D.lds_anchors, D.res, D.anchors, D.f, D.p, pp, B.plan_out.as<PlanOut>(), B.plan_keys.as<PlanKey>(), dp, cfg));
    if (!B.plan_ev) NS_HIP(hipEventCreateWithFlags(&B.plan_ev, hipEventDisableTiming));
    NS_HIP(hipEventRecord(B.plan_ev, D.stream));
    NS_TRY(ksw_dev_launch(c, dp_ws, max_q, kp, B.plan_ev, pp, B.plan_out.as<PlanOut>(), (uint32_t)n_pairs, two_part));
    B.plan_ws = dp_ws;
    B.plan_delivered.assign(n, 0);
    return NSGPU_OK;
}

// the device-planned results into the jobs' caches (every task of every alignment the kernel planned: the jobs look them up by key); part 0:
// what a two-part batch has ready early, part 1: everything that has not been delivered yet
}  // namespace

// The device-planned results in two steps: wait for a part of them (0: behind the bulk classes; 1: behind everything -- the workspace is
// handed back), then per request the copy of its problems' results into the job's cache.  A caller may do the second step for disjoint
// requests on several threads (the contig engine does it inside each builder's own task).
int batch_plan_wait(nsgpu_ctx *c, AlignBatch &B, int part, KswDevResults &R)
{
    R = KswDevResults{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    if (B.plan_ws < 0) return NSGPU_OK;
    const int ws = B.plan_ws;
    if (part == 1) B.plan_ws = -1;
    NS_TRY(ksw_dev_collect(c, ws, part, R));
    if (!R.res || part != 1) return NSGPU_OK;
    const PlanOut *po = B.plan_out.as<PlanOut>();
    uint64_t n_dev = 0, n_host = 0;
    for (size_t i = 0; i < B.plan_pair.size(); ++i) {
        if (B.plan_pair[i] == ~0u) { ++n_host; continue; }
        const PlanOut o = po[B.plan_pair[i]];
        if (o.flags) ++n_host; else ++n_dev;
        for (int bit = 0; bit < 8; ++bit) if (o.flags >> bit & 1) __atomic_fetch_add(&c->plan_why[bit], 1, __ATOMIC_RELAXED);
    }
    n_host += B.reqs.size() - B.plan_pair.size();
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->plan_pairs_dev += n_dev, c->plan_pairs_host += n_host;
    return NSGPU_OK;
}
// request i's problems from the results of `part` into its job's cache; returns how many (0: not this part's, not planned, lost, or done before).
// Part 1 takes whatever has not been delivered yet unless own_part_only.  ~0u: the status word is up but what it announces does not add up yet
// (a word raised by a kernel that is still running overtook its data): ask again.
// NSGPU_CONS_DEBUG: alignments by their longest DP problem -- [kind: gap fill / left ext / right ext / left ext, target longer / right ext, target
// longer][width class of its target][anti-diagonals, bucketed]
uint64_t g_dp_shape[5][4][8];
const bool g_dp_shape_on = getenv("NSGPU_CONS_DEBUG") != nullptr;
uint32_t batch_plan_deliver_one(AlignBatch &B, const KswDevResults &R, size_t i, int part, bool own_part_only)
{
    using namespace mm2;
    if (!R.res || i >= B.plan_pair.size() || B.plan_pair[i] == ~0u || B.plan_delivered[i]) return 0;
    const uint32_t q = B.plan_pair[i];
    const PlanOut o = B.plan_out.as<PlanOut>()[q];
    if (o.flags || o.n_tasks == 0) return 0;
    if (part == 0 && o.slow) return 0;
    if (own_part_only && part == 1 && !o.slow) return 0;         // (two threads, one per part: the first part's requests are the other thread's)
    const uint32_t st = __atomic_load_n(&R.status[q], __ATOMIC_ACQUIRE);
    if (st != 1u) return 0;                            // (2: the CIGAR arena overflowed: the job asks for its problems again)
    {   // the word may have been raised by a kernel that is still running (ksw_collect.hpp): believe it once what it announces adds up -- the
        // check word is seeded with the batch's epoch and the alignment, and every word counts with its position in the hand-over.  Nothing
        // outside the landing zones is read: an offset that has not arrived yet is all ones (ksw_dev_launch), or anything at all
        if ((uint64_t)B.plan_base[i] + o.n_tasks > R.n_slots) return ~0u;
        uint32_t sum = dv_check_seed(R.epoch, q);
        const uint32_t n_rw = o.n_tasks * (uint32_t)(sizeof(KswResult) / 4);
        const uint32_t *rw = reinterpret_cast<const uint32_t *>(R.res + B.plan_base[i]);
        for (uint32_t k = 0; k < n_rw; ++k) sum += dv_check_term(k, rw[k]);
        const uint64_t base = R.coff[B.plan_base[i]];
        uint64_t total = 0;
        for (uint32_t t = 0; t < o.n_tasks; ++t) {
            const uint32_t slot = B.plan_base[i] + t;
            const uint64_t at = R.coff[slot];
            const uint32_t nc = (uint32_t)R.res[slot].n_cigar;
            if (nc > (1u << 24) || at < base || at - base != total || at + nc > R.cig_cap) return ~0u;       // (the CIGARs of an alignment lie back to back in task order)
            for (uint32_t k = 0; k < nc; ++k) sum += dv_check_term(n_rw + (uint32_t)total + k, R.cig[at + k]);
            total += nc;
        }
        for (uint32_t t = 0; t < o.n_tasks; ++t) sum += dv_check_term(n_rw + (uint32_t)total + t, (uint32_t)R.coff[B.plan_base[i] + t]);
        if (sum != __atomic_load_n(&R.check[q], __ATOMIC_ACQUIRE)) return ~0u;       // not all of it is here yet: ask again
    }
    B.plan_delivered[i] = 1;
    AlignJob &J = B.jobs[i];
    if (J.finished) return 0;
    const PlanKey *keys = B.plan_keys.as<PlanKey>();
    if (g_dp_shape_on) {
        // NSGPU_CONS_DEBUG: what an alignment's longest problem looks like (a slot waits for the slowest of its alignments)
        long long best = -1; int kind = 0, cell_w = 0;
        for (uint32_t t = 0; t < o.n_tasks; ++t) {
            const PlanKey &k = keys[B.plan_base[i] + t];
            const int ql = k.qe - k.qs, tl = k.re - k.rs;
            if (ql <= 0 || tl <= 0) continue;
            const long long rows = ksw_rows_bound(ql, tl, k.w);
            if (rows > best) best = rows, kind = (k.flag & 0x08) ? 0 : (k.flag & 0x02) ? 1 : 2, cell_w = tl <= 256 ? 0 : tl <= 512 ? 1 : tl <= 1536 ? 2 : 3, kind += (!(k.flag & 0x08) && tl > ql) ? 2 : 0;
        }
        if (best >= 0) {
            const int bk = best < 256 ? 0 : best < 512 ? 1 : best < 768 ? 2 : best < 1024 ? 3 : best < 1536 ? 4 : best < 2048 ? 5 : best < 3072 ? 6 : 7;
            __atomic_fetch_add(&g_dp_shape[kind][cell_w][bk], 1, __ATOMIC_RELAXED);
        }
    }
    for (uint32_t t = 0; t < o.n_tasks; ++t) {
        const uint32_t slot = B.plan_base[i] + t;
        const KswResult &r = R.res[slot];
        DpResult d;
        d.max = r.max; d.zdropped = r.zdropped; d.max_q = r.max_q; d.max_t = r.max_t; d.mqe = r.mqe; d.mqe_t = r.mqe_t; d.mte = r.mte;
        d.mte_q = r.mte_q; d.score = r.score; d.reach_end = r.reach_end;
        DpKey k;
        memcpy(&k, &keys[slot], sizeof(DpKey));
        J.cache.put(k, d, R.cig + R.coff[slot], (uint32_t)r.n_cigar);
    }
    return o.n_tasks;
}
// the job of request i to its end if everything it asked for has arrived: its AlnOut is final (B.early_done[i]) -- a job that asks for more (a
// Z-drop's second pass) waits for align_finish's rounds; nothing of a job with a problem in the host-planned batch in flight is touched
bool align_early_one(AlignBatch &B, size_t i, mm2::AlnOut &out)
{
    using namespace mm2;
    if (!B.plan_delivered[i] || B.early_done[i]) return false;
    AlignJob &J = B.jobs[i];
    if (!J.finished) { if (!J.cache.missing.empty()) return false; J.step(); }        // (missing: what the plan pass asked for beyond the device's problems)
    if (!J.finished) return false;
    out.reset();
    align_read_result(J, B.reqs[i].ref, B.reqs[i].ref_len, out);
    B.early_done[i] = 1;
    return true;
}

namespace {
int batch_plan_deliver(nsgpu_ctx *c, AlignBatch &B, int part)
{
    KswDevResults R;
    NS_TRY(batch_plan_wait(c, B, part, R));
    if (!R.res) return NSGPU_OK;
    std::atomic<uint64_t> n_tasks{0};
    parallel_for("align.dp_deliver", B.plan_pair.size(), [&](size_t i) {
        // (behind a waited-for stream everything has arrived: a sum that does not add up twice is a lost alignment -- the job asks again)
        uint32_t got = batch_plan_deliver_one(B, R, i, part, false);
        if (got == ~0u) got = batch_plan_deliver_one(B, R, i, part, false);
        if (got != ~0u) n_tasks += got;
    });
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->aln_dp_tasks += n_tasks.load();
    return NSGPU_OK;
}

// host step of all live jobs, then the DP tasks they are waiting for: descriptors + sequence pool (pinned)
int batch_prepare_round(nsgpu_ctx *c, AlignBatch &B, bool stepped = false)
{
    using namespace mm2;
    const double a0 = now_ms();
    if (!stepped) parallel_for("align.step", B.live.size(), [&](size_t i) { B.jobs[B.live[i]].step(); });
    std::vector<uint32_t> still;
    B.t_off.clear(), B.b_off.clear();
    size_t nt = 0, nb = 0;
    for (uint32_t j : B.live) {
        AlignJob &J = B.jobs[j];
        if (J.finished) continue;
        still.push_back(j);
        if (B.plan_ws >= 0) plan_match(c, B, j);        // what the device planned and launched is not asked for again
        B.t_off.push_back(nt), B.b_off.push_back(nb);
        nt += J.cache.missing.size();
        for (const DpKey &k : J.cache.missing) nb += (size_t)(k.qe - k.qs) + (size_t)(k.re - k.rs);
    }
    B.live.swap(still);
    B.nb = nb;
    if (B.live.empty()) { B.host_ms += now_ms() - a0; return NSGPU_OK; }
    NS_CHECK(nb < (1ull << 32), NSGPU_ERR_RANGE, "align: DP sequence pool exceeds 4 GiB; use smaller batches");
    B.tasks.resize(nt);
    B.task_keys.resize(nt);                          // (the jobs' request lists may be rebuilt before the results are delivered: align_finish_early)
    nsgpu_ctx::KswWs &KW = c->kws[B.ws_index];      // the DP sequence pool is staged in pinned memory: one DMA, no pageable bounce
    if (KW.h_pool_cap < nb + 16) {
        if (KW.h_pool) NS_HIP(hipHostFree(KW.h_pool));
        KW.h_pool = nullptr, KW.h_pool_cap = 0;
        const size_t want = (nb + 16) * 3 / 2 + 4096;
        NS_HIP(hipHostMalloc(reinterpret_cast<void **>(&KW.h_pool), want, hipHostMallocDefault));
        KW.h_pool_cap = want;
    }
    uint8_t *const pool = KW.h_pool;
    parallel_for("align.dp_pack", B.live.size(), [&](size_t li) {
        AlignJob &J = B.jobs[B.live[li]];
        size_t ti = B.t_off[li], bo = B.b_off[li];
        for (const DpKey &k : J.cache.missing) {
            B.task_keys[ti] = k;
            KswTask &t = B.tasks[ti++];
            const int ql = k.qe - k.qs, tl = k.re - k.rs;
            t.qoff = (uint32_t)bo; t.toff = (uint32_t)(bo + ql); t.qlen = ql; t.tlen = tl;
            t.w = k.w; t.zdrop = k.zdrop; t.end_bonus = k.end_bonus; t.flag = k.flag | approx_task_flag(k.flag);
            uint8_t *q = pool + bo, *tt = q + ql;
            const uint8_t *qs = J.qseq.data() + k.qs, *ts = J.ref->seq.data() + k.rs;
            if (k.flag & 0x02) {            // left extension: both sequences reversed (align.c:693-696)
                for (int x = 0; x < ql; ++x) q[x] = qs[ql - 1 - x];
                for (int x = 0; x < tl; ++x) tt[x] = ts[tl - 1 - x];
            } else { memcpy(q, qs, ql); memcpy(tt, ts, tl); }
            bo += (size_t)ql + tl;
        }
    });
    B.host_ms += now_ms() - a0;
    return NSGPU_OK;
}

// DP results back into the jobs' caches
void batch_deliver(AlignBatch &B)
{
    using namespace mm2;
    const double a0 = now_ms();
    parallel_for("align.dp_deliver", B.live.size(), [&](size_t li) {
        AlignJob &J = B.jobs[B.live[li]];
        const size_t t_end = li + 1 < B.t_off.size() ? B.t_off[li + 1] : B.tasks.size();
        for (size_t ti = B.t_off[li]; ti < t_end;) {
            const DpKey &k = B.task_keys[ti];
            const KswResult &r = B.res[ti];
            DpResult d;
            d.max = r.max; d.zdropped = r.zdropped; d.max_q = r.max_q; d.max_t = r.max_t; d.mqe = r.mqe; d.mqe_t = r.mqe_t; d.mte = r.mte;
            d.mte_q = r.mte_q; d.score = r.score; d.reach_end = r.reach_end;
            J.cache.put(k, d, B.cig.data() + B.coff[ti], (uint32_t)r.n_cigar);
            ++ti;
        }
    });
    B.dp_tasks += B.tasks.size();
    ++B.rounds;
    B.host_ms += now_ms() - a0;
}

}  // namespace

// Part 1: seeds, chains, the plan of every region's DP problems, and their launch.  B.reqs (and what its pointers refer to)
// must stay alive until align_finish.
static void batch_start_jobs(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi)
{
    using namespace mm2;
    const Opt opt = batch_opt(c);
    for (size_t i = lo; i < hi; ++i) {
        B.jobs[i].start(B.reqs[i].idx, B.reqs[i].qry, (int)B.reqs[i].qry_len, opt);
        // a query of length 0 has no sketch either way; pre_mz must be non-null to count as "given"
        if (B.reqs[i].qry_mz) B.jobs[i].pre_mz = B.reqs[i].qry_mz, B.jobs[i].n_pre_mz = B.reqs[i].n_qry_mz;
    }
}

// the host code's seeds for the jobs `which` of the batch: lookup table of the references concerned, each once -- pairs may share one --, then collect_seeds
static void host_seed(nsgpu_ctx *c, AlignBatch &B, size_t lo, const std::vector<uint32_t> &which)
{
    using namespace mm2;
    const Opt opt = batch_opt(c);
    std::vector<uint32_t> todo;                       // one request per index without its table
    for (uint32_t i : which) {
        const AlignReq &r = B.reqs[lo + i];
        if (r.idx->has_table) continue;
        bool seen = false;
        for (uint32_t j : todo) seen = seen || B.reqs[lo + j].idx == r.idx;
        if (!seen) todo.push_back(i);
    }
    parallel_for("align.index", todo.size(), [&](size_t k) {
        const AlignReq &r = B.reqs[lo + todo[k]];
        if (r.ref_mz) r.idx->build_from_sketch(r.ref, (uint32_t)r.ref_len, opt.w, opt.k, opt.mid_occ_frac, r.ref_mz, r.n_ref_mz);
        else r.idx->build(r.ref, (uint32_t)r.ref_len, opt.w, opt.k, opt.mid_occ_frac);
    });
    parallel_for("align.seed", which.size(), [&](size_t k) { B.jobs[lo + which[k]].seed(); });
}
// ... and a chaining launch for their lists on chaining workspace cw
static int host_chain_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, const std::vector<uint32_t> &which, int cw)
{
    using namespace mm2;
    nsgpu_ctx::ChainWs &W = c->cws[cw];
    W.lists.clear(), W.off.assign(1, 0), W.avg.clear();
    for (uint32_t i : which) {
        const AlignJob &J = B.jobs[lo + i];
        W.lists.push_back(J.a.data()), W.off.push_back(W.off.back() + J.a.size()), W.avg.push_back(J.avg_qspan);
    }
    return gpu_chain_launch(c, cw, batch_opt(c), W.lists, W.off, W.avg);
}
static int host_seed_and_chain_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, const std::vector<uint32_t> &which, int cw)
{
    host_seed(c, B, lo, which);
    return host_chain_launch(c, B, lo, which, cw);
}

// Index + seeds on the GPU (seeds.hip), the chaining kernel on their lists (chain.hip) / the results and every job's first step
// (chains, regions, DP plan).  Pairs the seeding kernel hands back (anchors sharing a reference position, oversize lists) and
// requests without pinned minimizer lists are seeded by the host code and chained by a second launch.
static int batch_seed_and_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int ws, bool prepared = false)
{
    using namespace mm2;
    const size_t n = hi - lo;
    const Opt opt = batch_opt(c);
    if (!prepared) parallel_for("align.seed", n, [&](size_t i) { B.jobs[lo + i].seed_prepare(); });
    nsgpu_ctx::SeedWs &S = c->seed_ws[ws];
    std::vector<SeedPair> &pairs = B.seed_pairs;
    pairs.clear();
    S.pair_of.assign(n, ~0u);
    S.fb.clear();
    for (size_t i = 0; i < n; ++i) {
        const AlignReq &r = B.reqs[lo + i];
        if (r.qry_mz && r.ref_mz && r.n_ref_mz < (1ull << 30) && r.n_qry_mz < (1ull << 31)) {
            S.pair_of[i] = (uint32_t)pairs.size();
            SeedPair p{};
            p.ref = r.ref_mz_dev ? r.ref_mz_dev : r.ref_mz, p.qry = r.qry_mz, p.n_ref = (uint32_t)r.n_ref_mz, p.n_qry = (uint32_t)r.n_qry_mz;
            p.cnt_tab = r.ref_cnt, p.cnt_meta = r.ref_cnt_meta, p.cnt_bits = r.ref_cnt_bits;
            pairs.push_back(p);
        } else S.fb.push_back((uint32_t)i);
    }
    // seeds and, right behind them on the same stream, the chaining of their lists; the host looks at the results once
    NS_TRY(gpu_seeds_chain_launch(c, ws, 2 * ws, opt, pairs, false));       // (the anchor total: behind the plan kernel, gpu_seeds_total in the callers)
    S.calls += 1, S.pairs += pairs.size();
    // requests without pinned minimizer lists: the host code now, pairs the kernels hand back: in batch_wait_and_step
    nsgpu_ctx::ChainWs &W = c->cws[2 * ws + 1];
    W.lists.clear(), W.off.assign(1, 0), W.avg.clear();
    NS_TRY(host_seed_and_chain_launch(c, B, lo, S.fb, 2 * ws + 1));
    return NSGPU_OK;
}
static int batch_wait_and_step(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int ws)
{
    using namespace mm2;
    const size_t n = hi - lo;
    nsgpu_ctx::SeedWs &S = c->seed_ws[ws];
    const SeedResult *res = nullptr;
    const Anchor *a = nullptr;
    const int32_t *f = nullptr, *p = nullptr, *f2 = nullptr, *p2 = nullptr;
    const double g0 = now_ms();
    NS_TRY(gpu_seeds_chain_wait(c, ws, 2 * ws, res, a, f, p));
    S.ms_wait += now_ms() - g0;
    NS_TRY(gpu_chain_wait(c, 2 * ws + 1, f2, p2));
    const std::vector<uint64_t> &off2 = c->cws[2 * ws + 1].off;
    // (host-seeded jobs keep their own copy of the scores: the workspace's buffer may be used once more below)
    auto take = [&](const std::vector<uint32_t> &which) {
        for (size_t k = 0; k < which.size() && f2; ++k) {
            AlignJob &J = B.jobs[lo + which[k]];
            J.own_f.assign(f2 + off2[k], f2 + off2[k + 1]), J.own_p.assign(p2 + off2[k], p2 + off2[k + 1]);
            J.cf = J.own_f.data(), J.cp = J.own_p.data();
        }
    };
    take(S.fb);
    // pairs handed back by the kernels: host seeds + one more chaining launch (rare: anchors sharing a reference position, oversize lists)
    std::vector<uint32_t> &late = S.late;
    late.clear();
    for (size_t i = 0; i < n; ++i) if (S.pair_of[i] != ~0u && res[S.pair_of[i]].flags) late.push_back((uint32_t)i);
    if (B.skip.size() < B.reqs.size()) B.skip.resize(B.reqs.size(), 0);
    if (!late.empty()) {
        S.fallbacks += late.size();
        const double l0 = now_ms();
        host_seed(c, B, lo, late);
        for (uint32_t i : late) {
            const uint32_t fl = res[S.pair_of[i]].flags;
            for (int b = 0; b < 5; ++b) S.late_flag[b] += fl >> b & 1;
            const uint64_t na = B.jobs[lo + i].a.size();
            S.late_anchors += na, S.late_longest = std::max(S.late_longest, na);
        }
        // the deferred ones (AlignBatch::defer_anchors): a list of 10^4 - 10^5 anchors takes milliseconds to chain and its DP problems as long again
        if (B.defer_anchors) {
            size_t keep = 0;
            for (uint32_t i : late) {
                if (B.jobs[lo + i].a.size() > B.defer_anchors) B.deferred.push_back((uint32_t)(lo + i)), B.skip[lo + i] = 1;
                else late[keep++] = i;
            }
            late.resize(keep);
        }
        const double l1 = now_ms();
        NS_TRY(host_chain_launch(c, B, lo, late, 2 * ws + 1));
        NS_TRY(gpu_chain_wait(c, 2 * ws + 1, f2, p2));
        S.late_seed_ms += l1 - l0, S.late_chain_ms += now_ms() - l1, ++S.late_calls;
        take(late);
    }
    B.chain_ms += now_ms() - g0;
    { std::lock_guard<std::mutex> lk(c->stat_m); c->aln_seed_host += S.fb.size() + late.size(), c->aln_seed_gpu += n - S.fb.size() - late.size(); }
    parallel_for("align.step", n, [&](size_t i) {
        if (B.skip[lo + i]) return;
        AlignJob &J = B.jobs[lo + i];
        const uint32_t q = S.pair_of[i];
        if (q != ~0u && !res[q].flags) {
            J.set_anchors(a ? a + res[q].base : nullptr, res[q].n, res[q].avg);
            if (res[q].n) J.cf = f + res[q].base, J.cp = p + res[q].base;
        }
        J.step();
    });
    return NSGPU_OK;
}

static int prestep_check(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws)
{
    NS_CHECK(lo <= hi && hi <= B.reqs.size(), NSGPU_ERR_ARG, "align_prestep: bad range");
    NS_CHECK(chain_ws >= 0 && chain_ws < (int)(sizeof(c->seed_ws) / sizeof(c->seed_ws[0])), NSGPU_ERR_ARG, "align_prestep: bad workspace");
    // (B.jobs must not be resized here: another range of the same batch may be in its step on other threads -- the caller sizes it)
    NS_CHECK(B.jobs.size() >= B.reqs.size(), NSGPU_ERR_ARG, "align_prestep: size B.jobs first");
    return NSGPU_OK;
}
int align_prestep_start(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi)
{
    NS_TRY(prestep_check(c, B, lo, hi, 0));
    batch_start_jobs(c, B, lo, hi);
    return NSGPU_OK;
}
int align_prestep_launch(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws, bool started_and_seeded, int dp_ws)
{
    NS_TRY(prestep_check(c, B, lo, hi, chain_ws));
    const double a0 = now_ms();
    if (!started_and_seeded) batch_start_jobs(c, B, lo, hi);
    B.plan_ws = -1;
    NS_TRY(batch_seed_and_launch(c, B, lo, hi, chain_ws, started_and_seeded));
    // the plan kernel and the DP launch behind the chaining kernel, without waiting for either (plan.hip)
    if (dp_ws >= 0) NS_TRY(batch_plan_launch(c, B, lo, hi, chain_ws, dp_ws));
    NS_TRY(gpu_seeds_total(c, chain_ws));
    B.host_ms += now_ms() - a0;
    return NSGPU_OK;
}
int align_prestep_finish(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws)
{
    NS_TRY(prestep_check(c, B, lo, hi, chain_ws));
    const double a0 = now_ms();
    NS_TRY(batch_wait_and_step(c, B, lo, hi, chain_ws));
    B.host_ms += now_ms() - a0;
    B.prestepped = true;
    return NSGPU_OK;
}
int align_prestep(nsgpu_ctx *c, AlignBatch &B, size_t lo, size_t hi, int chain_ws)
{
    NS_TRY(align_prestep_launch(c, B, lo, hi, chain_ws, false));
    return align_prestep_finish(c, B, lo, hi, chain_ws);
}

int align_begin(nsgpu_ctx *c, AlignBatch &B, int ws_index)
{
    using namespace mm2;
    const size_t n_pairs = B.reqs.size();
    const bool pre = B.prestepped;
    B.prestepped = false;
    B.ws_index = ws_index, B.in_flight = false;
    if (!pre) B.host_ms = 0;
    B.dp_ms = 0, B.dp_tasks = B.rounds = 0;
    B.live.clear();
    if (n_pairs == 0) return NSGPU_OK;
    // the job objects (and the capacity of their vectors) are kept from batch to batch: freeing and re-allocating the
    // ~100 small blocks of every job cost more CPU than the alignment bookkeeping itself
    if (B.jobs.size() < n_pairs) B.jobs.resize(n_pairs);
    if (!pre) {
        const double a0 = now_ms();
        batch_start_jobs(c, B, 0, n_pairs);
        B.plan_ws = -1;
        NS_TRY(batch_seed_and_launch(c, B, 0, n_pairs, 0));      // direct API calls: chain workspace 0
        NS_TRY(batch_plan_launch(c, B, 0, n_pairs, 0, ws_index));
        NS_TRY(gpu_seeds_total(c, 0));
        NS_TRY(batch_wait_and_step(c, B, 0, n_pairs, 0));
        B.host_ms += now_ms() - a0;
    }
    B.live.clear();
    for (size_t i = 0; i < n_pairs; ++i) if (!(i < B.skip.size() && B.skip[i])) B.live.push_back((uint32_t)i);
    NS_TRY(batch_prepare_round(c, B, true));
    if (B.live.empty()) return NSGPU_OK;
    const double a0 = now_ms();
    // (with the device plan this launch is normally empty: the problems are in flight already)
    NS_TRY(ksw_batch_launch(c, B.tasks, c->kws[ws_index].h_pool, B.nb, batch_ksw_params(batch_opt(c)), B.res, B.cig, B.coff, ws_index));
    B.dp_ms += now_ms() - a0;
    B.in_flight = true;
    return NSGPU_OK;
}

// debug breakdown of part 2 (single batch thread at a time per group; plain doubles are good enough for a debug print):
// [0] wait for the launch in flight, [1] later rounds in total, [2] their DP launches, [3] result conversion, [4] #later rounds
double g_finish_ms[5];

// Part 2: the DP results, the execution of the alignment skeleton on them (further DP rounds, should a plan have missed a
// problem, run synchronously), and ConsensusGraph::alignRead's conversion of every pair.
int align_finish(nsgpu_ctx *c, AlignBatch &B, std::vector<mm2::AlnOut> &outs)
{
    using namespace mm2;
    const size_t n_pairs = B.reqs.size();
    // the result objects keep their vectors (the caller swaps them with the builders' previous results): no 64 KB edit list is
    // allocated on one thread and freed on another per alignment
    if (outs.size() < n_pairs) outs.resize(n_pairs);
    const bool had_early = B.early_done.size() == n_pairs;
    auto skipped = [&](size_t i) { return i < B.skip.size() && B.skip[i]; };         // (deferred: the job has left the batch)
    for (size_t i = 0; i < n_pairs; ++i) if (!(had_early && B.early_done[i]) && !skipped(i)) outs[i].reset();
    if (n_pairs == 0) return NSGPU_OK;
    const KswParams kp = batch_ksw_params(batch_opt(c));
    const double f0 = now_ms();
    if (B.in_flight) {
        const double a0 = now_ms();
        NS_TRY(ksw_batch_collect(c, B.tasks, B.res, B.cig, B.coff, B.ws_index));
        B.dp_ms += now_ms() - a0;
        g_finish_ms[0] += now_ms() - a0;
        B.in_flight = false;
        batch_deliver(B);
    }
    {   // what the device planned and launched (plan.hip): everything align_finish_early has not delivered yet
        const double a0 = now_ms();
        NS_TRY(batch_plan_deliver(c, B, 1));
        B.dp_ms += now_ms() - a0;
        g_finish_ms[0] += now_ms() - a0;
    }
    const double f1 = now_ms();
    for (int round = 0; !B.live.empty(); ++round) {
        NS_CHECK(round < 64, NSGPU_ERR_ARG, "align: no convergence after 64 DP rounds (internal error)");
        NS_TRY(batch_prepare_round(c, B));
        if (B.live.empty()) break;
        const double a0 = now_ms();
        NS_TRY(ksw_run_batch(c, B.tasks, c->kws[B.ws_index].h_pool, B.nb, kp, B.res, B.cig, B.coff, B.ws_index));
        B.dp_ms += now_ms() - a0;
        g_finish_ms[2] += now_ms() - a0;
        g_finish_ms[4] += 1;
        batch_deliver(B);
    }
    const double b0 = now_ms();
    g_finish_ms[1] += b0 - f1;
    parallel_for("align.result", n_pairs, [&](size_t i) { if (!(had_early && B.early_done[i]) && !skipped(i)) align_read_result(B.jobs[i], B.reqs[i].ref, B.reqs[i].ref_len, outs[i]); });
    B.early_done.clear();
    B.skip.clear(), B.deferred.clear();
    B.host_ms += now_ms() - b0;
    g_finish_ms[3] += now_ms() - b0;
    (void)f0;
    {
        std::lock_guard<std::mutex> lk(c->stat_m);
        c->aln_host_ms += B.host_ms, c->aln_dp_ms += B.dp_ms, c->aln_dp_tasks += B.dp_tasks, c->aln_rounds += B.rounds, c->aln_pairs += n_pairs;
    }
    return NSGPU_OK;
}

// The device-planned results of part `part` (0: the alignments without a problem in a late class, behind the bulk classes; 1: the others,
// behind everything): the jobs whose every problem has arrived run their skeleton to the end and are converted; ready[i] = 1 for those.
// The caller sizes outs and B.early_done (one flag per request, zeroed) before the first part.
int align_finish_early(nsgpu_ctx *c, AlignBatch &B, std::vector<mm2::AlnOut> &outs, std::vector<uint8_t> &ready, int part)
{
    using namespace mm2;
    const size_t n_pairs = B.reqs.size();
    ready.assign(n_pairs, 0);
    if (n_pairs == 0 || B.plan_ws < 0 || !B.plan_two_part) return NSGPU_OK;
    NS_CHECK(outs.size() >= n_pairs && B.early_done.size() == n_pairs, NSGPU_ERR_ARG, "align_finish_early: result vectors not sized by the caller");
    const double a0 = now_ms();
    NS_TRY(batch_plan_deliver(c, B, part));
    if (part == 0) g_finish_ms[0] += now_ms() - a0;
    parallel_for("align.early", n_pairs, [&](size_t i) { if (align_early_one(B, i, outs[i])) ready[i] = 1; });
    return NSGPU_OK;
}

int align_seeded_jobs(nsgpu_ctx *c, AlignBatch &B, int chain_ws, int dp_ws, std::vector<mm2::AlnOut> &outs)
{
    using namespace mm2;
    const size_t n = B.reqs.size();
    NS_CHECK(B.jobs.size() >= n, NSGPU_ERR_ARG, "align_seeded_jobs: jobs missing");
    std::vector<uint32_t> all(n);
    for (size_t i = 0; i < n; ++i) all[i] = (uint32_t)i;
    B.skip.clear(), B.deferred.clear(), B.defer_anchors = 0, B.plan_ws = -1, B.host_ms = 0;
    NS_TRY(host_chain_launch(c, B, 0, all, chain_ws));
    const int32_t *f = nullptr, *p = nullptr;
    NS_TRY(gpu_chain_wait(c, chain_ws, f, p));
    const std::vector<uint64_t> &off = c->cws[chain_ws].off;
    for (size_t i = 0; i < n && f; ++i) {
        AlignJob &J = B.jobs[i];
        J.own_f.assign(f + off[i], f + off[i + 1]), J.own_p.assign(p + off[i], p + off[i + 1]);
        J.cf = J.own_f.data(), J.cp = J.own_p.data();
    }
    parallel_for("align.step", n, [&](size_t i) { B.jobs[i].step(); });
    B.prestepped = true;
    NS_TRY(align_begin(c, B, dp_ws));
    return align_finish(c, B, outs);
}

int align_requests(nsgpu_ctx *c, std::vector<AlignReq> &reqs, std::vector<mm2::AlnOut> &outs, int ws_index)
{
    AlignBatch B;
    B.reqs.swap(reqs);
    int rc = align_begin(c, B, ws_index);
    if (rc == NSGPU_OK) rc = align_finish(c, B, outs);
    B.reqs.swap(reqs);
    return rc;
}

int align_batch(nsgpu_ctx *c, const char *refs, const uint64_t *roff, uint32_t n_refs, const char *qrys, const uint64_t *qoff,
                const uint32_t *pair_ref, uint32_t n_pairs, std::vector<mm2::AlnOut> &outs)
{
    using namespace mm2;
    outs.assign(n_pairs, AlnOut());
    if (n_pairs == 0) return NSGPU_OK;
    for (uint32_t i = 0; i < n_pairs; ++i) NS_CHECK(pair_ref[i] < n_refs, NSGPU_ERR_ARG, "pair %u refers to reference %u of %u", i, pair_ref[i], n_refs);
    for (uint32_t i = 0; i < n_refs; ++i) NS_CHECK(roff[i + 1] - roff[i] < (1ull << 31), NSGPU_ERR_RANGE, "reference %u longer than 2^31", i);
    NS_CHECK(c->prm.m_k > 0 && c->prm.m_k <= 28 && c->prm.m_w > 0 && c->prm.m_w < 256, NSGPU_ERR_ARG, "minimap k must be in 1..28 and w in 1..255 (sketch.c:84)");
    const double t0 = now_ms();
    // minimizers of every reference and every query in one GPU batch (mm_sketch.hip)
    std::vector<SketchReq> sk(n_refs + (size_t)n_pairs);
    for (uint32_t i = 0; i < n_refs; ++i) sk[i] = SketchReq{refs + roff[i], (size_t)(roff[i + 1] - roff[i])};
    for (uint32_t i = 0; i < n_pairs; ++i) sk[n_refs + i] = SketchReq{qrys + qoff[i], (size_t)(qoff[i + 1] - qoff[i])};
    const Anchor *mz = nullptr;
    std::vector<uint64_t> mz_off;
    NS_TRY(gpu_mm_sketch(c, sk, (int)c->prm.m_w, (int)c->prm.m_k, mz, mz_off));
    std::vector<RefIndex> idx(n_refs);
    parallel_for("index.build", n_refs, [&](size_t i) {
        idx[i].set_sequence(refs + roff[i], (uint32_t)(roff[i + 1] - roff[i]), (int)c->prm.m_w, (int)c->prm.m_k);     // the lookup table is the GPU's (seeds.hip)
    });
    c->aln_index_ms += now_ms() - t0;
    std::vector<AlignReq> reqs(n_pairs);
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const uint32_t rf = pair_ref[i];
        reqs[i] = AlignReq{&idx[rf], refs + roff[rf], (size_t)(roff[rf + 1] - roff[rf]), qrys + qoff[i], (size_t)(qoff[i + 1] - qoff[i]),
                           mz + mz_off[n_refs + i], (size_t)(mz_off[n_refs + i + 1] - mz_off[n_refs + i]), mz + mz_off[rf], (size_t)(mz_off[rf + 1] - mz_off[rf])};
        // every reference and every query lies whole in the sketch batch's staging buffer in HBM: the plan kernel reads them there
        reqs[i].qry_dev = sketch_dev_seq(c, 0, n_refs + i), reqs[i].ref_dev = sketch_dev_seq(c, 0, rf);
        reqs[i].ref_dev_lo = 0, reqs[i].ref_dev_n = (uint32_t)reqs[i].ref_len;
    }
    return align_requests(c, reqs, outs);
}

}  // namespace nsgpu

using namespace nsgpu;

// mm_sketch (minimap2/sketch.c:77-143, rid 0) of a batch of sequences on the GPU: minimizers of sequence i are the
// (x, y) pairs xy_out[2 * off_out[i]] .. xy_out[2 * off_out[i + 1]) in the reference's output order.
extern "C" int nsgpu_mm_sketch_batch(nsgpu_ctx *c, const char *seqs, const uint64_t *seq_off, uint32_t n, uint32_t w, uint32_t k, uint64_t **xy_out,
                                     uint64_t **off_out)
{
    NS_CHECK(c && seq_off && (n == 0 || seqs) && xy_out && off_out, NSGPU_ERR_ARG, "nsgpu_mm_sketch_batch: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    std::vector<SketchReq> sk(n);
    for (uint32_t i = 0; i < n; ++i) {
        NS_CHECK(seq_off[i + 1] >= seq_off[i], NSGPU_ERR_ARG, "offsets must be non-decreasing");
        sk[i] = SketchReq{seqs + seq_off[i], (size_t)(seq_off[i + 1] - seq_off[i])};
    }
    const mm2::Anchor *mz = nullptr;
    std::vector<uint64_t> off;
    NS_TRY(gpu_mm_sketch(c, sk, (int)w, (int)k, mz, off));
    uint64_t *xy = (uint64_t *)malloc((off[n] * 2 + 1) * 8), *of = (uint64_t *)malloc(((size_t)n + 1) * 8);
    NS_CHECK(xy && of, NSGPU_ERR_NOMEM, "malloc failed");
    if (off[n]) memcpy(xy, mz, off[n] * 16);
    memcpy(of, off.data(), ((size_t)n + 1) * 8);
    *xy_out = xy, *off_out = of;
    return NSGPU_OK;
}

// mm_chain_dp's forward pass (minimap2/chain.c:43-92) for a batch of sorted anchor lists on the GPU: score f and predecessor p
// of every anchor, with minimap2's default max_gap / bw / max_chain_skip and params.max_chain_iter.
extern "C" int nsgpu_chain_scores(nsgpu_ctx *c, const uint64_t *xy, const uint64_t *off, uint32_t n, int32_t *f_out, int32_t *p_out)
{
    NS_CHECK(c && off && (off[n] == 0 || (xy && f_out && p_out)), NSGPU_ERR_ARG, "nsgpu_chain_scores: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    std::vector<const mm2::Anchor *> lists(n);
    std::vector<uint64_t> o(off, off + n + 1);
    std::vector<float> avg(n);
    std::vector<mm2::Anchor> tmp;
    for (uint32_t i = 0; i < n; ++i) {
        NS_CHECK(off[i + 1] >= off[i] && off[i + 1] - off[i] < (1ull << 31), NSGPU_ERR_ARG, "offsets must be non-decreasing, lists shorter than 2^31");
        lists[i] = reinterpret_cast<const mm2::Anchor *>(xy) + off[i];
        tmp.assign(lists[i], lists[i] + (off[i + 1] - off[i]));
        avg[i] = mm2::chain_avg_qspan(tmp);
    }
    const int32_t *f = nullptr, *p = nullptr;
    NS_TRY(gpu_chain_launch(c, 0, batch_opt(c), lists, o, avg));
    NS_TRY(gpu_chain_wait(c, 0, f, p));
    if (f) memcpy(f_out, f, off[n] * sizeof(int32_t)), memcpy(p_out, p, off[n] * sizeof(int32_t));
    return NSGPU_OK;
}

extern "C" int nsgpu_align_batch(nsgpu_ctx *c, const char *refs, const uint64_t *ref_off, uint32_t n_refs, const char *qrys,
                                 const uint64_t *qry_off, const uint32_t *pair_ref, uint32_t n_pairs, nsgpu_aln *out,
                                 uint32_t **cigars_out, nsgpu_edit **edits_out)
{
    NS_CHECK(c && ref_off && qry_off && (n_pairs == 0 || (refs && qrys && pair_ref && out)) && cigars_out && edits_out, NSGPU_ERR_ARG,
             "nsgpu_align_batch: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    std::vector<mm2::AlnOut> outs;
    NS_TRY(align_batch(c, refs, ref_off, n_refs, qrys, qry_off, pair_ref, n_pairs, outs));
    uint64_t nc = 0, ne = 0;
    for (auto &o : outs) nc += o.cigar.size(), ne += o.edits.size();
    uint32_t *cg = (uint32_t *)malloc((nc + 1) * 4);
    nsgpu_edit *ed = (nsgpu_edit *)malloc((ne + 1) * sizeof(nsgpu_edit));
    NS_CHECK(cg && ed, NSGPU_ERR_NOMEM, "malloc failed");
    nc = ne = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const mm2::AlnOut &o = outs[i];
        nsgpu_aln &a = out[i];
        a.ok = o.ok; a.hits = o.hits; a.rel_pos = o.rel_pos; a.begin_offset = o.begin_offset; a.end_offset = o.end_offset;
        a.rs = o.rs; a.re = o.re; a.qs = o.qs; a.qe = o.qe; a.blen = o.blen; a.mlen = o.mlen; a.n_ambi = o.n_ambi; a.dp_max = o.dp_max;
        a.n_cigar = (uint32_t)o.cigar.size(); a.n_edits = (uint32_t)o.edits.size(); a.cigar_off = nc; a.edit_off = ne;
        memcpy(cg + nc, o.cigar.data(), o.cigar.size() * 4);
        for (const mm2::EditOp &e : o.edits) { nsgpu_edit &x = ed[ne++]; x.type = e.type; x.base = e.base; x.reserved = 0; x.num = e.num; }
        nc += o.cigar.size();
    }
    *cigars_out = cg;
    *edits_out = ed;
    return NSGPU_OK;
}

extern "C" int nsgpu_get_align_stats(const nsgpu_ctx *c, nsgpu_align_stats *s)
{
    NS_CHECK(c && s, NSGPU_ERR_ARG, "null argument");
    s->pairs = c->aln_pairs; s->dp_tasks = c->aln_dp_tasks; s->dp_rounds = c->aln_rounds; s->dp_cells = c->ksw_cells;
    s->index_ms = c->aln_index_ms; s->host_ms = c->aln_host_ms; s->dp_ms = c->aln_dp_ms; s->dp_kernel_ms = c->ksw_kernel_ms; s->dp_kernel_sum_ms = c->ksw_kernel_sum_ms;
    s->dp_alg_bytes = c->ksw_alg_bytes;
    s->dp_launches = c->ksw_launches;
    s->host_threads = host_threads();
    s->seed_pairs_gpu = c->aln_seed_gpu, s->seed_pairs_host = c->aln_seed_host;
    s->plan_pairs_dev = c->plan_pairs_dev, s->plan_pairs_host = c->plan_pairs_host, s->plan_hits = c->plan_hits, s->plan_misses = c->plan_misses, s->plan_extra = c->plan_extra;
    return NSGPU_OK;
}

extern "C" int nsgpu_reset_align_stats(nsgpu_ctx *c)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null argument");
    c->aln_pairs = c->aln_dp_tasks = c->aln_rounds = c->aln_seed_gpu = c->aln_seed_host = 0;
    c->plan_pairs_dev = c->plan_pairs_host = c->plan_hits = c->plan_misses = c->plan_extra = 0;
    c->aln_index_ms = c->aln_host_ms = c->aln_dp_ms = 0;
    c->ksw_kernel_ms = c->ksw_cells = c->ksw_alg_bytes = c->ksw_kernel_sum_ms = 0;
    c->ksw_launches = 0;
    return NSGPU_OK;
}
