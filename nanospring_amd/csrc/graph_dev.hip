// graph_dev.hip -- the consensus DAG in HBM: kernels over dgraph.hpp (one workgroup of 256 threads per update), the per-contig host
// handle (DevGraph), the memory pools.  See graph_dev.hpp.
#include "graph_dev.hpp"
#include "host_util.hpp"
#include <time.h>

namespace nsgpu {

// ---------------------------------------------------------------------------------------------------------------------------
// the team of a workgroup (dgraph.hpp's contract: every thread runs the same team calls)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kDgSharedWords = 6144;            // 24 KB of LDS for the tables a team searches (dgraph.hpp append_runs)
struct DevTeam {
    uint32_t *lds;                                   // >= 8 words
    uint32_t *sh;
    __device__ uint32_t *shared() { return sh; }
    __device__ uint32_t shared_words() const { return kDgSharedWords; }
    __device__ uint32_t clock() const { return (uint32_t)wall_clock64(); }       // 100 MHz
    __device__ void add_to(uint32_t *p, uint32_t v) { atomicAdd(p, v); }
    __device__ void min_to(uint32_t *p, uint32_t v) { atomicMin(p, v); }
    __device__ uint32_t peek(const uint32_t *p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ bool cas(uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v) == expect; }
    __device__ bool helper() const { return threadIdx.x >= 64u; }
    __device__ uint32_t crew_rank() const { return threadIdx.x == 0 ? 0u : threadIdx.x - 63u; }
    __device__ uint32_t crew_size() const { return blockDim.x > 64u ? blockDim.x - 63u : 1u; }
    __device__ uint32_t tid() const { return threadIdx.x; }
    __device__ uint32_t size() const { return blockDim.x; }
    __device__ void sync() { __syncthreads(); }
    __device__ uint32_t bcast(uint32_t v)
    {
        __syncthreads();
        if (threadIdx.x == 0) lds[0] = v;
        __syncthreads();
        return lds[0];
    }
    __device__ uint32_t scan(uint32_t v, uint32_t &total)
    {
        const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
        uint32_t x = v;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        __syncthreads();
        if (lane == 63u || threadIdx.x == blockDim.x - 1) lds[1 + w] = x;
        __syncthreads();
        uint32_t base = 0, tot = 0;
        for (uint32_t i = 0; i < nw; ++i) { const uint32_t t = lds[1 + i]; if (i < w) base += t; tot += t; }
        total = tot;
        return base + x - v;
    }
    __device__ uint32_t min_all(uint32_t v)
    {
        const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
#pragma unroll
        for (uint32_t d = 32; d; d >>= 1) { const uint32_t y = __shfl_xor(v, d, 64); v = y < v ? y : v; }
        __syncthreads();
        if (lane == 0) lds[1 + w] = v;
        __syncthreads();
        uint32_t r = lds[1];
        for (uint32_t i = 1; i < nw; ++i) { const uint32_t t = lds[1 + i]; r = t < r ? t : r; }
        return r;
    }
    __device__ uint32_t max_all(uint32_t v)
    {
        const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
#pragma unroll
        for (uint32_t d = 32; d; d >>= 1) { const uint32_t y = __shfl_xor(v, d, 64); v = y > v ? y : v; }
        __syncthreads();
        if (lane == 0) lds[1 + w] = v;
        __syncthreads();
        uint32_t r = lds[1];
        for (uint32_t i = 1; i < nw; ++i) { const uint32_t t = lds[1 + i]; r = t > r ? t : r; }
        return r;
    }
};

struct DgSetup { uint32_t cap_nodes, cap_edges, cap_chunks, cap_path, cap_wk, cap_multi, path_off, dbg_flags; };      // path_off != NIL: the host has moved the path arrays
constexpr uint32_t kMidCap = 48u << 10;              // bytes of new consensus a result carries itself (longer: the host copies them)
// Two reports per update: the first when the consensus is final (header as of then + the new stretch of the consensus), the second when
// removeCycles behind it is done (the header again: the graph's final size).  A status word == the launch's epoch once its part is in place; the
// check words = sum of that part's words + epoch.
struct DgResult {
    uint32_t status, check;                           // the first report
    uint32_t status2, check2;                         // the second
    dg::Hdr hdr;
    dg::Hdr hdr2;
    uint32_t mid_len, pad[3];
    uint8_t mid[kMidCap];
};
constexpr int kDgThreads = 512;                      // the most a launch may have (NSGPU_GRAPH_THREADS: 256 or 512)

__device__ static void dg_apply_setup(const dg::G &g, const DgSetup &s)
{
    dg::Hdr &h = *g.h;
    h.cap_nodes = s.cap_nodes, h.cap_edges = s.cap_edges, h.cap_chunks = s.cap_chunks, h.cap_path = s.cap_path, h.cap_wk = s.cap_wk, h.cap_multi = s.cap_multi;
    if (s.path_off != dg::NIL) { h.pos_bias += s.path_off - h.path_off; h.path_off = s.path_off; }
    h.dbg_flags = s.dbg_flags;
}

// What a graph's workgroup is told (pinned host memory, written by the host while the kernel waits).  cmd is written last.
// The order word carries the ticket of the prepare() it belongs to (code | ticket << 3): a workgroup is launched with its graph's ticket and takes
// nothing else.  A launch whose workgroups START late -- its kernel queued behind others on a shared hardware queue: seen with a second process on
// the GPU -- would otherwise find the slot record already re-armed for the graph's NEXT update and run that update beside the workgroup it was meant
// for (the same script applied twice).  Tickets are unique over all graphs of the engine (a staging block may change hands).
enum : uint32_t { DG_CMD_NONE = 0, DG_CMD_UPDATE = 1, DG_CMD_INIT_UPDATE = 2, DG_CMD_CANCEL = 3 };
constexpr uint32_t kDgTicketShift = 3, kDgTicketMask = (1u << 29) - 1u;
struct DgSlot;
struct DgArm { DgSlot *slot; uint32_t ticket, pad; };
struct DgSlot {
    uint32_t cmd, epoch, n_ops, id, seed_len, first_id;
    long long begin_offset, end_offset;
    dg::G g;
    DgSetup setup;
    const uint32_t *ops;
    const uint8_t *seed;
    DgResult *res;
};
static_assert(sizeof(DgSlot) % 4 == 0, "slot records are read as words");

__device__ static void dg_report(const dg::G &g, DevTeam &t, DgResult *res, uint32_t epoch, bool second)
{
    const dg::Hdr &h = *g.h;
    const uint32_t *hw = reinterpret_cast<const uint32_t *>(&h);
    uint32_t *rw = reinterpret_cast<uint32_t *>(second ? &res->hdr2 : &res->hdr);
    uint32_t sum = 0;
    for (uint32_t i = threadIdx.x; i < sizeof(dg::Hdr) / 4; i += blockDim.x) { const uint32_t v = hw[i]; rw[i] = v; sum += v * (i + 1); }
    if (!second) {
        uint32_t mid = 0;
        if (!h.err && h.new_len >= h.P + h.S) mid = h.new_len - h.P - h.S;
        const uint32_t mid_here = mid <= kMidCap ? mid : 0;
        const uint8_t *src = g.ps + h.path_off + h.P;
        for (uint32_t i = threadIdx.x; i < mid_here; i += blockDim.x) { const uint8_t b = src[i]; res->mid[i] = b; sum += (uint32_t)b * (i + 7u); }
        if (threadIdx.x == 0) { res->mid_len = mid; sum += mid * 3u; }
    }
    __threadfence_system();
    uint32_t tot;
    (void)t.scan(sum, tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        *(second ? &res->check2 : &res->check) = tot + epoch;
        __threadfence_system();
        __hip_atomic_store(second ? &res->status2 : &res->status, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
}

// One workgroup per graph.  It waits for its graph's order -- the accepted read's script is known only when the alignment's last DP problem
// is done and the host has finished the alignment, long after this launch -- then runs updateGraph + calculateMainPathGreedy + removeCycles and
// reports into pinned memory.  ONE launch for the graphs of a whole slot: kernels launched one by one from eighty streams queue up behind each
// other and behind the DP kernels on the runtime's eight hardware queues (measured: 3.3 ms from launch to report for 0.7 ms of kernel).
// A workgroup that is never told anything leaves after `patience` ticks of the 100 MHz clock (the host always cancels what it does not use).
__global__ __launch_bounds__(kDgThreads) void dg_serve_kernel(const DgArm *arms, unsigned long long patience, unsigned long long late_start)
{
    __shared__ uint32_t lds[16];
    __shared__ uint32_t shm[kDgSharedWords];
    __shared__ uint32_t slot_w[sizeof(DgSlot) / 4];
    DgSlot *s = arms[blockIdx.x].slot;
    const uint32_t ticket = arms[blockIdx.x].ticket;
    if (threadIdx.x == 0) {
        // (tests, NSGPU_GRAPH_LATE_START_US: the workgroups come to life this late -- what a launch queued behind other kernels looks like)
        if (late_start) { const unsigned long long w0 = wall_clock64(); while (wall_clock64() - w0 < late_start) __builtin_amdgcn_s_sleep(64); }
        const unsigned long long t0 = wall_clock64();
        uint32_t cmd;
        for (;;) {
            const uint32_t w = __hip_atomic_load(&s->cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((w >> kDgTicketShift) != ticket) { cmd = DG_CMD_CANCEL; break; }       // not this launch's order any more: the host has moved on
            cmd = w & ((1u << kDgTicketShift) - 1u);
            if (cmd != DG_CMD_NONE) break;
            if (wall_clock64() - t0 > patience) { cmd = DG_CMD_CANCEL; break; }
            __builtin_amdgcn_s_sleep(64);
        }
        lds[8] = cmd;
    }
    __syncthreads();
    const uint32_t cmd = lds[8];
    if (cmd != DG_CMD_UPDATE && cmd != DG_CMD_INIT_UPDATE) return;
    for (uint32_t i = threadIdx.x; i < sizeof(DgSlot) / 4; i += blockDim.x) slot_w[i] = reinterpret_cast<const volatile uint32_t *>(s)[i];
    __syncthreads();
    const DgSlot &L = *reinterpret_cast<const DgSlot *>(slot_w);
    const dg::G g = L.g;
    uint32_t *lp = lds, *sp = shm;
    asm volatile("" : "+s"(lp), "+s"(sp));            // (keeps the compiler from folding the team into a constant it cannot initialise)
    DevTeam t{lp, sp};
    if (cmd == DG_CMD_INIT_UPDATE) {
        if (threadIdx.x == 0) { uint32_t *w = reinterpret_cast<uint32_t *>(g.h); for (uint32_t i = 0; i < sizeof(dg::Hdr) / 4; ++i) w[i] = 0; }
        __syncthreads();
    }
    if (threadIdx.x == 0) dg_apply_setup(g, L.setup);
    __syncthreads();
    dg::Ops<DevTeam> o(g, t);
    if (cmd == DG_CMD_INIT_UPDATE) o.initialize(L.seed, L.seed_len, L.first_id);
    __syncthreads();
    if (!o.failed()) o.update(L.ops, L.n_ops, L.begin_offset, L.end_offset, L.id);
    __syncthreads();
    if (!o.failed()) o.main_path();
    __syncthreads();
    dg_report(g, t, L.res, L.epoch, false);           // the consensus is final: the builder goes on
    if (!o.failed()) o.finish_path();
    __syncthreads();
    dg_report(g, t, L.res, L.epoch, true);
}

// removeCycles again for an update whose workgroup stopped in front of a split that did not fit (ERR_ROOM): the host has made room (the slot record
// holds the arrays' new places and sizes; it is the update's own otherwise)
__global__ __launch_bounds__(kDgThreads) void dg_finish_kernel(const DgSlot *s)
{
    __shared__ uint32_t lds[16];
    __shared__ uint32_t shm[kDgSharedWords];
    __shared__ uint32_t slot_w[sizeof(DgSlot) / 4];
    for (uint32_t i = threadIdx.x; i < sizeof(DgSlot) / 4; i += blockDim.x) slot_w[i] = reinterpret_cast<const volatile uint32_t *>(s)[i];
    __syncthreads();
    const DgSlot &L = *reinterpret_cast<const DgSlot *>(slot_w);
    const dg::G g = L.g;
    uint32_t *lp = lds, *sp = shm;
    asm volatile("" : "+s"(lp), "+s"(sp));
    DevTeam t{lp, sp};
    if (threadIdx.x == 0) { dg_apply_setup(g, L.setup); if (g.h->err == dg::ERR_ROOM) g.h->err = 0; }
    __syncthreads();
    dg::Ops<DevTeam> o(g, t);
    if (!o.failed()) o.finish_path();
    __syncthreads();
    dg_report(g, t, L.res, L.epoch, true);
}

// ---------------------------------------------------------------------------------------------------------------------------
// pools
// ---------------------------------------------------------------------------------------------------------------------------
static inline int class_of(size_t bytes, size_t *granted)
{
    // quarter-octave classes from 4 KiB: (4 + q) << (k - 2)
    size_t b = bytes < 4096 ? 4096 : bytes;
    int k = 63 - __builtin_clzll(b);
    for (int q = 0; q < 4; ++q) { const size_t s = (size_t)(4 + q) << (k - 2); if (s >= b) { *granted = s; return (k - 12) * 4 + q; } }
    *granted = (size_t)1 << (k + 1);
    return (k + 1 - 12) * 4;
}
SlabPool::~SlabPool()
{
    for (void *p : slabs_) { if (host_) (void)hipHostFree(p); else (void)hipFree(p); }
}
void *SlabPool::alloc(size_t bytes, size_t *granted)
{
    size_t g = 0;
    const int cls = class_of(bytes, &g);
    *granted = g;
    if (cls < 0 || cls >= 48 * 4) { set_error("graph memory pool: a block of %zu bytes is beyond the pool's classes", bytes); return nullptr; }
    std::lock_guard<std::mutex> lk(m_);
    // a free block of this very size (the sizes are few: quarter-octave classes)
    for (size_t i = 0; i < free_sz_.size(); ++i)
        if (free_sz_[i].first == g && !free_sz_[i].second.empty()) { void *p = free_sz_[i].second.back(); free_sz_[i].second.pop_back(); in_use_ += g; if (in_use_ > peak_) peak_ = in_use_; return p; }
    // no block of this size: the smallest free one of a larger class up to twice the size rather than new memory (graphs outgrow their blocks at
    // different times: the sizes the pool holds free are rarely the very size asked for; the block keeps its own size when it comes back)
    {
        size_t best = 0, bi = 0;
        for (size_t i = 0; i < free_sz_.size(); ++i)
            if (free_sz_[i].first > g && free_sz_[i].first <= 2 * g && !free_sz_[i].second.empty() && (!best || free_sz_[i].first < best)) best = free_sz_[i].first, bi = i;
        if (best) { void *p = free_sz_[bi].second.back(); free_sz_[bi].second.pop_back(); *granted = best; in_use_ += best; if (in_use_ > peak_) peak_ = in_use_; return p; }
    }
    const size_t ga = (g + 255) & ~(size_t)255;
    for (auto &t : tails_)
        if (t.second >= ga) { void *p = t.first; t.first += ga, t.second -= ga; in_use_ += g; if (in_use_ > peak_) peak_ = in_use_; return p; }
    const size_t sb = ga > slab_bytes_ ? ga : slab_bytes_;
    void *s = nullptr;
    const hipError_t e = host_ ? hipHostMalloc(&s, sb, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) : hipMalloc(&s, sb);
    if (e != hipSuccess || !s) { set_error("graph memory pool: %s(%zu) failed: %s", host_ ? "hipHostMalloc" : "hipMalloc", sb, hipGetErrorString(e)); return nullptr; }
    // (tests: fresh slabs full of a pattern -- HBM straight from the driver is often zero, and code that leans on that works until the process gets
    // memory another one has used)
    static const char *poison = getenv("NSGPU_POOL_POISON");         // the 32-bit word every fresh slab is filled with (decimal)
    if (poison) { const uint32_t w = (uint32_t)strtoul(poison, nullptr, 0); if (host_) { uint32_t *q = static_cast<uint32_t *>(s); for (size_t i = 0; i < sb / 4; ++i) q[i] = w; } else { (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(s), (int)w, sb / 4); (void)hipDeviceSynchronize(); } }     // (the fill is in place before anything is written there)
    slabs_.push_back(s);
    mapped_ += sb;
    tails_.push_back(std::make_pair(static_cast<char *>(s) + ga, sb - ga));
    in_use_ += g;
    if (in_use_ > peak_) peak_ = in_use_;
    return s;
}
void SlabPool::free(void *p, size_t granted)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(m_);
    in_use_ -= granted;
    for (auto &f : free_sz_) if (f.first == granted) { f.second.push_back(p); return; }
    free_sz_.push_back(std::make_pair(granted, std::vector<void *>(1, p)));
}

DevGraphShared::~DevGraphShared()
{
    if (serve_stream) { (void)hipStreamSynchronize(serve_stream); (void)hipStreamDestroy(serve_stream); }
    for (Launch &L : launches) { pin.free(L.ptrs, L.granted); (void)hipEventDestroy(L.done); }
    for (hipEvent_t e : free_events) (void)hipEventDestroy(e);
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the pointer graph on the host
// ---------------------------------------------------------------------------------------------------------------------------
int HostGraph::submit(const std::string &query, const mm2::AlnOut &aln, read_t id, bool rc)
{
    if (g.num_reads() == 0) {                       // src/Consensus.cpp:319-324
        const std::string seed = g.main_path;
        g.main_path.clear();
        g.first_read = first_read;
        g.initialize(seed, first_read, 0);
        g.calculate_main_path_greedy();
        path_changed_from = 0;
    }
    g.path_changed_from = path_changed_from;
    g.update_graph(query, aln.edits, (ssize_t)aln.begin_offset, (ssize_t)aln.end_offset, id, (long)aln.rel_pos, rc);
    g.calculate_main_path_greedy();
    path_changed_from = g.path_changed_from;
    dbg[0] = g.dbg_cycles_calls, dbg[1] = g.dbg_cycles_skipped, dbg[2] = g.dbg_spliced, dbg[3] = g.dbg_cycles_idle, dbg[4] = g.dbg_walked_nodes, dbg[5] = g.dbg_cycles_listed;
    dbg_cycles_ms = g.dbg_cycles_ms;
    return NSGPU_OK;
}
void HostGraph::write_reads(cons::StreamSet &o, const std::function<cons::ReadBases(cons::read_t)> *source) { g.first_read = first_read; g.write_reads(o, source); }

// ---------------------------------------------------------------------------------------------------------------------------
// the graph in HBM
// ---------------------------------------------------------------------------------------------------------------------------
static inline double now_ms_() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

DevGraph::DevGraph(DevGraphShared *sh, uint32_t builder) : sh_(sh) { (void)builder; memset(&hdr_, 0, sizeof(hdr_)); }
DevGraph::~DevGraph()
{
    if (armed_ && !pending_) cancel();
    if (pending_) { const double w0 = now_ms_(); while (!ready() && now_ms_() - w0 < 5000.0) { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); } if (ready()) (void)complete(); }
    if (finalizing_) (void)finalize(true);
    if (e_begun_ && e_ev_) (void)hipEventSynchronize(e_ev_);
    if (e_ev_) (void)hipEventDestroy(e_ev_);
    for (Block &b : retired_) give(b);
    Block *dev[] = {&b_nodes_, &b_mark_, &b_pidx_, &b_edges_, &b_chunks_, &b_pe_, &b_pn_, &b_ps_, &b_sve_, &b_svn_, &b_svs_, &b_multi_, &b_wk_, &b_hdr_};
    for (Block *b : dev) give(*b);
    Block *pin[] = {&pin_, &e_nodes_, &e_edges_, &e_chunks_, &e_pe_, &e_pn_, &e_ps_};
    for (Block *b : pin) give(*b, true);
}
int DevGraph::take(Block &b, size_t bytes, bool pinned)
{
    b.p = (pinned ? sh_->pin : sh_->dev).alloc(bytes, &b.granted);
    b.want = bytes;
    return b.p ? NSGPU_OK : NSGPU_ERR_NOMEM;
}
void DevGraph::give(Block &b, bool pinned)
{
    if (b.p) (pinned ? sh_->pin : sh_->dev).free(b.p, b.granted);
    b = Block();
}
dg::G DevGraph::view() const
{
    dg::G g;
    g.h = static_cast<dg::Hdr *>(b_hdr_.p);
    g.nodes = static_cast<dg::Node *>(b_nodes_.p), g.edges = static_cast<dg::Edge *>(b_edges_.p), g.chunks = static_cast<dg::Chunk *>(b_chunks_.p), g.mark = static_cast<uint32_t *>(b_mark_.p), g.pidx = static_cast<uint32_t *>(b_pidx_.p);
    g.pe = static_cast<uint32_t *>(b_pe_.p), g.pn = static_cast<uint32_t *>(b_pn_.p), g.ps = static_cast<uint8_t *>(b_ps_.p);
    g.sv_e = static_cast<uint32_t *>(b_sve_.p), g.sv_n = static_cast<uint32_t *>(b_svn_.p), g.sv_s = static_cast<uint8_t *>(b_svs_.p);
    g.multi_list = static_cast<uint32_t *>(b_multi_.p), g.wk = static_cast<uint32_t *>(b_wk_.p);
    return g;
}

// room for the next update (worst case of its script + what a removeCycles behind it may copy); arrays that grow are copied on the
// serve stream, in front of the kernel that will use them, and the old ones go back to the pool when that update has reported
int DevGraph::grow(const cons::SoaNeed &need, uint32_t seed_len, hipStream_t st)
{
    if (!st) st = sh_->serve_stream;
    std::function<size_t(size_t, size_t)> bigger = [](size_t have, size_t want) { size_t c = have ? have : 4096; while (c < want) c += c / 2 + 4096; return c; };
    const uint32_t n_nodes = inited_ ? hdr_.n_nodes : seed_len, n_edges = inited_ ? hdr_.n_edges : seed_len, n_chunks = inited_ ? hdr_.n_chunks : 0;
    // (spare room for the private copies a removeCycles behind the update may make; when a split needs more the kernel says so and is run again: ERR_ROOM.
    // NSGPU_GRAPH_SLACK, tests: next to none, arrays as large as asked)
    static const char *sl = getenv("NSGPU_GRAPH_SLACK");
    const uint32_t slack_n = sl ? (uint32_t)atoi(sl) : std::max<uint32_t>(1u << 15, n_nodes / 4), slack_e = sl ? (uint32_t)atoi(sl) : std::max<uint32_t>(1u << 15, n_edges / 4), slack_c = sl ? (uint32_t)atoi(sl) : std::max<uint32_t>(1u << 15, n_chunks / 4);
    if (sl) bigger = [](size_t have, size_t want) { return want > have ? want : have; };
    bool grew = false;
    if (!b_hdr_.p) { NS_TRY(take(b_hdr_, sizeof(dg::Hdr))); cap_multi_ = 16384; NS_TRY(take(b_multi_, (size_t)cap_multi_ * 4)); }
    const uint64_t want_n = (uint64_t)n_nodes + need.nodes + slack_n, want_e = (uint64_t)n_edges + need.edges + slack_e, want_c = (uint64_t)n_chunks + need.chunks + slack_c;
    NS_CHECK(want_n < dg::kRefMask && want_e < dg::kRefMask && want_c < 0xfffffff0ull, NSGPU_ERR_RANGE, "consensus graph: more than 2^29 nodes or edges in one contig");
    if (want_n > cap_nodes_) {
        const size_t c = bigger(cap_nodes_, want_n);
        Block nb, mb, xb;
        NS_TRY(take(nb, c * sizeof(dg::Node))); NS_TRY(take(mb, c * 4)); NS_TRY(take(xb, c * 4));
        if (inited_ && n_nodes) {
            NS_HIP(hipMemcpyAsync(nb.p, b_nodes_.p, (size_t)n_nodes * sizeof(dg::Node), hipMemcpyDeviceToDevice, st));
            NS_HIP(hipMemcpyAsync(mb.p, b_mark_.p, (size_t)n_nodes * 4, hipMemcpyDeviceToDevice, st));
            NS_HIP(hipMemcpyAsync(xb.p, b_pidx_.p, (size_t)n_nodes * 4, hipMemcpyDeviceToDevice, st));
        }
        retire(b_nodes_), retire(b_mark_), retire(b_pidx_);
        b_nodes_ = nb, b_mark_ = mb, b_pidx_ = xb, cap_nodes_ = (uint32_t)c, grew = true;
    }
    if (want_e > cap_edges_) {
        const size_t c = bigger(cap_edges_, want_e);
        Block nb;
        NS_TRY(take(nb, c * sizeof(dg::Edge)));
        if (inited_ && n_edges) NS_HIP(hipMemcpyAsync(nb.p, b_edges_.p, (size_t)n_edges * sizeof(dg::Edge), hipMemcpyDeviceToDevice, st));
        retire(b_edges_);
        b_edges_ = nb, cap_edges_ = (uint32_t)c, grew = true;
    }
    if (want_c > cap_chunks_) {
        const size_t c = bigger(cap_chunks_, want_c);
        Block nb;
        NS_TRY(take(nb, c * sizeof(dg::Chunk)));
        if (inited_ && n_chunks) NS_HIP(hipMemcpyAsync(nb.p, b_chunks_.p, (size_t)n_chunks * sizeof(dg::Chunk), hipMemcpyDeviceToDevice, st));
        retire(b_chunks_);
        b_chunks_ = nb, cap_chunks_ = (uint32_t)c, grew = true;
    }
    if (need.wk > cap_wk_) {
        const size_t c = bigger(cap_wk_, need.wk);
        retire(b_wk_);
        NS_TRY(take(b_wk_, c * 4));
        cap_wk_ = (uint32_t)c, grew = true;
    }
    // the path: need.path_side entries free on both sides
    const uint32_t len = inited_ ? hdr_.m + 1 : 0;
    const uint32_t side = inited_ ? need.path_side : seed_len + need.path_side;
    const bool short_left = path_off_ < side, short_right = (uint64_t)path_off_ + len + side > cap_path_;
    if (!inited_ || short_left || short_right) {
        const size_t c = bigger(0, (size_t)len + (inited_ ? 0 : seed_len) + 4 * (size_t)need.path_side + 4096 + len / 2);
        Block e2, n2, s2;
        NS_TRY(take(e2, c * 4)); NS_TRY(take(n2, c * 4)); NS_TRY(take(s2, c));
        const uint32_t off2 = (uint32_t)((c - len) / 2);
        if (len) {
            if (len > 1) NS_HIP(hipMemcpyAsync(static_cast<uint32_t *>(e2.p) + off2, static_cast<uint32_t *>(b_pe_.p) + path_off_, (size_t)(len - 1) * 4, hipMemcpyDeviceToDevice, st));
            NS_HIP(hipMemcpyAsync(static_cast<uint32_t *>(n2.p) + off2, static_cast<uint32_t *>(b_pn_.p) + path_off_, (size_t)len * 4, hipMemcpyDeviceToDevice, st));
            NS_HIP(hipMemcpyAsync(static_cast<uint8_t *>(s2.p) + off2, static_cast<uint8_t *>(b_ps_.p) + path_off_, len, hipMemcpyDeviceToDevice, st));
        }
        retire(b_pe_), retire(b_pn_), retire(b_ps_), retire(b_sve_), retire(b_svn_), retire(b_svs_);
        b_pe_ = e2, b_pn_ = n2, b_ps_ = s2;
        NS_TRY(take(b_sve_, c * 4)); NS_TRY(take(b_svn_, c * 4)); NS_TRY(take(b_svs_, c));
        cap_path_ = (uint32_t)c, path_off_ = off2, grew = true, moved_path_ = inited_;
    }
    if (grew) sh_->n_grow += 1;
    return NSGPU_OK;
}

// Room for ANY script of a read of `read_len` bases (the script itself is known only when the alignment is: the arrays must be in place before
// the graph's workgroup is launched), and the slot record a workgroup of a serve launch reads.
int DevGraph::prepare(size_t read_len)
{
    NS_CHECK(!pending_ && !armed_, NSGPU_ERR_ARG, "consensus graph: prepared with an update in flight (internal error)");
    NS_TRY(finalize(true));
    const uint32_t seed_len = inited_ ? 0 : (uint32_t)path_.size();
    NS_CHECK(inited_ || seed_len >= 1, NSGPU_ERR_ARG, "consensus graph: an empty seed read");
    const uint32_t L = (uint32_t)std::min<size_t>(read_len, 0x3fffffffu);
    const uint32_t max_ops = 2 * L + 8;                                   // every base an INSERT and a DELETE between each two at the very worst
    const cons::SoaNeed need = cons::soa_need(max_ops, L, L, inited_ ? hdr_.m + 1 : seed_len);
    NS_TRY(grow(need, seed_len));
    const size_t stage_bytes = sizeof(DgSlot) + sizeof(DgResult) + ((size_t)max_ops + 16) * 4 + seed_len + 64;
    if (stage_bytes > pin_.granted) { give(pin_, true); NS_TRY(take(pin_, std::max(stage_bytes + stage_bytes / 4, sizeof(DgSlot) + sizeof(DgResult) + ((size_t)sh_->max_ops + 16) * 4), true)); }
    DgSlot *slot = static_cast<DgSlot *>(pin_.p);
    DgResult *res = reinterpret_cast<DgResult *>(slot + 1);
    uint32_t *ops_pin = reinterpret_cast<uint32_t *>(res + 1);
    uint8_t *seed_pin = reinterpret_cast<uint8_t *>(ops_pin + max_ops + 16);
    ops_cap_ = max_ops;
    epoch_ = (sh_->next_ticket.fetch_add(1) & kDgTicketMask);
    if (epoch_ == 0) epoch_ = (sh_->next_ticket.fetch_add(1) & kDgTicketMask);
    slot->cmd = DG_CMD_NONE | (epoch_ << kDgTicketShift), slot->epoch = epoch_, slot->n_ops = 0, slot->id = 0, slot->seed_len = seed_len, slot->first_id = (uint32_t)first_read;
    slot->begin_offset = slot->end_offset = 0;
    slot->g = view();
    slot->setup = DgSetup{cap_nodes_, cap_edges_, cap_chunks_, cap_path_, cap_wk_, cap_multi_, moved_path_ ? path_off_ : dg::NIL, sh_->dbg_flags};
    slot->ops = ops_pin, slot->seed = seed_pin, slot->res = res;
    if (!inited_) memcpy(seed_pin, path_.data(), seed_len);
    res->status = 0, res->status2 = 0;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    prepared_ = true;
    return NSGPU_OK;
}

// ONE launch for the graphs given (all prepared): their workgroups wait for their orders
int graph_serve_launch(DevGraphShared *sh, DevGraph *const *graphs, size_t n)
{
    if (!n) return NSGPU_OK;
    std::lock_guard<std::mutex> lk(sh->serve_m);
    // the slot pointers of this launch: a pinned array that lives until a later launch finds this one's kernel gone
    for (size_t i = 0; i < sh->launches.size();) {
        DevGraphShared::Launch &L = sh->launches[i];
        if (hipEventQuery(L.done) == hipSuccess) { sh->pin.free(L.ptrs, L.granted); sh->free_events.push_back(L.done); L = sh->launches.back(); sh->launches.pop_back(); }
        else ++i;
    }
    DevGraphShared::Launch L;
    L.ptrs = sh->pin.alloc(n * sizeof(DgArm), &L.granted);
    NS_CHECK(L.ptrs, NSGPU_ERR_NOMEM, "consensus graph: no pinned memory for a launch's slot list");
    if (!sh->free_events.empty()) { L.done = sh->free_events.back(); sh->free_events.pop_back(); }
    else NS_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    DgArm *ptrs = static_cast<DgArm *>(L.ptrs);
    for (size_t i = 0; i < n; ++i) {
        DevGraph *g = graphs[i];
        NS_CHECK(g->prepared_ && !g->armed_, NSGPU_ERR_ARG, "consensus graph: a serve launch over a graph that was not prepared (internal error)");
        ptrs[i].slot = static_cast<DgSlot *>(g->pin_.p), ptrs[i].ticket = g->epoch_, ptrs[i].pad = 0;
        g->armed_ = true, g->prepared_ = false;
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);
    static const unsigned long long patience = [] { const char *e = getenv("NSGPU_WAIT_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return (unsigned long long)((v > 0 ? v : 120.0) * 1e8); }();
    static const int n_thr = [] { const char *e = getenv("NSGPU_GRAPH_THREADS"); const int v = e ? atoi(e) : 0; return v == 512 || v == 256 || v == 128 ? v : 512; }();
    static const unsigned long long late = [] { const char *e = getenv("NSGPU_GRAPH_LATE_START_US"); return e ? (unsigned long long)(atof(e) * 100.0) : 0ull; }();
    hipLaunchKernelGGL(dg_serve_kernel, dim3((uint32_t)n), dim3(n_thr), 0, sh->serve_stream, ptrs, patience, late);
    NS_HIP(hipGetLastError());
    NS_HIP(hipEventRecord(L.done, sh->serve_stream));
    sh->launches.push_back(L);
    sh->n_launches += 1;
    return NSGPU_OK;
}

void DevGraph::cancel()
{
    if (!armed_ || pending_) return;
    DgSlot *slot = static_cast<DgSlot *>(pin_.p);
    __atomic_store_n(&slot->cmd, (uint32_t)DG_CMD_CANCEL | (epoch_ << kDgTicketShift), __ATOMIC_RELEASE);
    armed_ = false;
    // (the arrays prepare() replaced stay retired until the next update reports: nothing has read the new ones yet, but nothing is lost either)
}

// the accepted read: its script into the staging area, then the order to the graph's workgroup (one of a serve launch if the graph is armed,
// else a launch of its own)
int DevGraph::submit(const std::string &query, const mm2::AlnOut &aln, read_t id, bool rc)
{
    NS_CHECK(!pending_, NSGPU_ERR_ARG, "consensus graph: an update is still in flight (internal error)");
    if (fail_rc_ != NSGPU_OK) return fail_rc_;
    if (!armed_) {
        if (!prepared_) NS_TRY(prepare(query.size()));
        DevGraph *self = this;
        NS_TRY(graph_serve_launch(sh_, &self, 1));
    }
    static thread_local std::vector<uint32_t> ops;
    uint32_t n_run, n_ins;
    cons::soa_script(query, aln.edits, (ssize_t)aln.begin_offset, (ssize_t)aln.end_offset, ops, n_run, n_ins);
    const uint32_t n_ops = (uint32_t)ops.size();
    DgSlot *slot = static_cast<DgSlot *>(pin_.p);
    if (n_ops > ops_cap_) { cancel(); set_error("consensus graph: a script of %u ops for a read of %zu bases (internal error)", n_ops, query.size()); return NSGPU_ERR_RANGE; }
    memcpy(const_cast<uint32_t *>(slot->ops), ops.data(), (size_t)n_ops * 4);
    slot->n_ops = n_ops, slot->id = (uint32_t)id, slot->begin_offset = (long long)aln.begin_offset, slot->end_offset = (long long)aln.end_offset;
    const bool first = !inited_;
    if (first) {
        reads_.insert(std::make_pair(first_read, cons::SoaRead{0, 0u, path_.size(), false}));
        start_ = 0, end_ = (ssize_t)path_.size();
        path_changed_from = 0;
        path_off_ = (cap_path_ - (uint32_t)path_.size()) / 2;            // (where initialize puts the seed)
        inited_ = true;
        if (sh_->check) {
            shadow_.reset(new cons::SoaGraph()); shadow_->first_read = first_read; shadow_->initialize(path_, first_read, 0); shadow_->calculate_main_path_greedy();
            ptr_shadow_.reset(new cons::ContigGraph()); ptr_shadow_->first_read = first_read; ptr_shadow_->initialize(path_, first_read, 0); ptr_shadow_->calculate_main_path_greedy();
        }
    }
    p_begin_ = (long long)aln.begin_offset, p_end_ = (long long)aln.end_offset;
    p_id_ = id, p_pos_ = (long)aln.rel_pos, p_len_ = query.size(), p_rc_ = rc, p_t0_ = now_ms_();
    __atomic_store_n(&slot->cmd, (uint32_t)(first ? DG_CMD_INIT_UPDATE : DG_CMD_UPDATE) | (epoch_ << kDgTicketShift), __ATOMIC_RELEASE);
    pending_ = true, armed_ = false;
    sh_->n_updates += 1;
    if (shadow_) {
        shadow_->update_graph(query, aln.edits, (ssize_t)aln.begin_offset, (ssize_t)aln.end_offset, id, (long)aln.rel_pos, rc);
        shadow_->calculate_main_path_greedy();
        // (and the pointer graph: the two host graphs must agree on the consensus, the span and the size after every read)
        ptr_shadow_->update_graph(query, aln.edits, (ssize_t)aln.begin_offset, (ssize_t)aln.end_offset, id, (long)aln.rel_pos, rc);
        ptr_shadow_->calculate_main_path_greedy();
        if (ptr_shadow_->main_path != shadow_->main_path || ptr_shadow_->start_pos != shadow_->start_pos || ptr_shadow_->end_pos != shadow_->end_pos || ptr_shadow_->num_edges() != shadow_->num_edges()) {
            size_t d = 0;
            while (d < ptr_shadow_->main_path.size() && d < shadow_->main_path.size() && ptr_shadow_->main_path[d] == shadow_->main_path[d]) ++d;
            set_error("consensus graph check: the structure-of-arrays graph and the pointer graph differ after read %u (the contig's read number %zu): consensus %zu / %zu bases, first difference at %zu; start %zd / %zd, end %zd / %zd, edges %zu / %zu; begin %lld end %lld, %zu ops, read of %zu",
                      (unsigned)id, reads_.size(), shadow_->main_path.size(), ptr_shadow_->main_path.size(), d, shadow_->start_pos, ptr_shadow_->start_pos, shadow_->end_pos, ptr_shadow_->end_pos, shadow_->num_edges(), ptr_shadow_->num_edges(),
                      (long long)aln.begin_offset, (long long)aln.end_offset, aln.edits.size(), query.size());
            return fail_rc_ = NSGPU_ERR_RANGE;
        }
    }
    return NSGPU_OK;
}

bool DevGraph::ready()
{
    if (!pending_) return true;
    const DgResult *res = reinterpret_cast<const DgResult *>(static_cast<const DgSlot *>(pin_.p) + 1);
    if (__atomic_load_n(&res->status, __ATOMIC_ACQUIRE) != epoch_) return false;
    // is everything the word announces here?
    const uint32_t *hw = reinterpret_cast<const uint32_t *>(&res->hdr);
    uint32_t sum = 0;
    for (uint32_t i = 0; i < sizeof(dg::Hdr) / 4; ++i) sum += hw[i] * (i + 1);
    const uint32_t mid = res->mid_len, mid_here = mid <= kMidCap ? mid : 0;
    for (uint32_t i = 0; i < mid_here; ++i) sum += (uint32_t)res->mid[i] * (i + 7u);
    sum += mid * 3u;
    return sum + epoch_ == res->check;
}

int DevGraph::complete()
{
    if (!pending_) return NSGPU_OK;
    if (!ready()) {
        const double w0 = now_ms_();
        static const double give_up_ms = [] { const char *e = getenv("NSGPU_WAIT_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return (v > 0 ? v : 120.0) * 1e3; }();
        int spins = 0;
        while (!ready()) {
            if (++spins > 200) { timespec ts = {0, 5000}; nanosleep(&ts, nullptr); }
            // (one wait that has run out ends the others too: the stage is lost, and a slot's worth of updates must not sit out the limit one after the other)
            if (sh_->aborted.load()) { set_error("consensus graph: given up (another update of this stage was not reported in time)"); return NSGPU_ERR_HIP; }
            if (now_ms_() - w0 > give_up_ms) {
                sh_->aborted.store(true);
                const hipError_t e = hipStreamQuery(sh_->serve_stream);
                set_error("consensus graph: an update has not been reported after %.0f s (stream: %s)", give_up_ms / 1e3, hipGetErrorString(e));
                return NSGPU_ERR_HIP;
            }
        }
        sh_->kernel_wait_ns += (uint64_t)((now_ms_() - w0) * 1e6);
    }
    const DgResult *res = reinterpret_cast<const DgResult *>(static_cast<const DgSlot *>(pin_.p) + 1);
    hdr_ = res->hdr;
    pending_ = false;
    moved_path_ = false;                                  // (the kernel has taken the arrays' new place over into the graph's header)
    finalizing_ = true;                                   // (removeCycles behind the recompute may still be running: finalize())
    sh_->update_ns += (uint64_t)((now_ms_() - p_t0_) * 1e6);
    if (hdr_.err) return kernel_error();
    NS_CHECK(hdr_.old_len == path_.size() && hdr_.path_off + hdr_.m + 1 <= cap_path_, NSGPU_ERR_RANGE, "consensus graph: the kernel's path is out of step with the host's (internal error)");
    path_off_ = hdr_.path_off;
    const uint32_t mid = res->mid_len;
    if (mid <= kMidCap) cons::soa_patch_path(path_, hdr_.P, hdr_.S, hdr_.new_len, res->mid);
    else {
        std::vector<uint8_t> tmp(mid);
        NS_HIP(hipMemcpy(tmp.data(), static_cast<uint8_t *>(b_ps_.p) + hdr_.path_off + hdr_.P, mid, hipMemcpyDeviceToHost));
        cons::soa_patch_path(path_, hdr_.P, hdr_.S, hdr_.new_len, tmp.data());
        sh_->n_mid_copies += 1;
    }
    if (hdr_.P < path_changed_from) path_changed_from = hdr_.P;
    reads_.insert(std::make_pair(p_id_, cons::SoaRead{p_pos_, hdr_.initial, p_len_, p_rc_}));
    auto er = reads_.find(hdr_.ending_id), sr = reads_.find(hdr_.starting_id);
    NS_CHECK(er != reads_.end() && sr != reads_.end(), NSGPU_ERR_RANGE, "consensus graph: the path's end reads are not in the read table (internal error)");
    end_ = er->second.pos + (long)er->second.len;
    start_ = sr->second.pos;
    edges_seen_ = hdr_.live_edges;
    if (shadow_) NS_TRY(finalize(true));                  // (the check compares whole arrays: both reports first)
    return NSGPU_OK;
}

int DevGraph::kernel_error()
{
    set_error("consensus graph kernel: error %u raised at dgraph.hpp:%u (capacity %u / script %u / degree %u / walk %u / work area %u) at read %u, %u nodes, %u edges, path %u; info %u %u %u %u; R %u Lf %u touch %u..%u cons_from %u begin %lld end %lld", hdr_.err, hdr_.err_line, hdr_.err & dg::ERR_CAP,
              hdr_.err & dg::ERR_SCRIPT, hdr_.err & dg::ERR_DEGREE, hdr_.err & dg::ERR_WALK, hdr_.err & dg::ERR_SCRATCH, (unsigned)p_id_, hdr_.n_nodes, hdr_.n_edges, hdr_.m, hdr_.err_info[0], hdr_.err_info[1], hdr_.err_info[2], hdr_.err_info[3], hdr_.right_off, hdr_.left_off, hdr_.upd_touch_lo, hdr_.upd_touch_hi, hdr_.cons_from, p_begin_, p_end_);
    return NSGPU_ERR_RANGE;
}

// The update's second report (removeCycles is done: the graph's final size and counters).  wait = false: only if it is there.
int DevGraph::finalize(bool wait)
{
    if (!finalizing_) return fail_rc_;
    const DgResult *res = reinterpret_cast<const DgResult *>(static_cast<const DgSlot *>(pin_.p) + 1);
    auto there = [&]() {
        if (__atomic_load_n(&res->status2, __ATOMIC_ACQUIRE) != epoch_) return false;
        const uint32_t *hw = reinterpret_cast<const uint32_t *>(&res->hdr2);
        uint32_t sum = 0;
        for (uint32_t i = 0; i < sizeof(dg::Hdr) / 4; ++i) sum += hw[i] * (i + 1);
        return sum + epoch_ == res->check2;
    };
    if (!there()) {
        if (!wait) return NSGPU_OK;
        const double w0 = now_ms_();
        static const double give_up_ms = [] { const char *e = getenv("NSGPU_WAIT_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return (v > 0 ? v : 120.0) * 1e3; }();
        int spins = 0;
        while (!there()) {
            if (++spins > 200) { timespec ts = {0, 5000}; nanosleep(&ts, nullptr); }
            if (sh_->aborted.load()) { set_error("consensus graph: given up (another update of this stage was not reported in time)"); return fail_rc_ = NSGPU_ERR_HIP; }
            if (now_ms_() - w0 > give_up_ms) { sh_->aborted.store(true); set_error("consensus graph: an update's second report has not come after %.0f s", give_up_ms / 1e3); return fail_rc_ = NSGPU_ERR_HIP; }
        }
        sh_->final_wait_ns += (uint64_t)((now_ms_() - w0) * 1e6);
    }
    if (res->hdr2.err == dg::ERR_ROOM) {
        // removeCycles stopped in front of a split whose copies do not fit (nothing of it done, the graph whole): room for it, and the same call again
        // in a launch of its own behind the array copies.
        // (On the stream of the copies back, not the serve stream: a slot's launch may be waiting there for its orders, and whoever waits for this
        // graph -- a contig that has just ended -- must not wait for that slot's end.)
        hdr_ = res->hdr2;
        hdr_.err = 0;
        cons::SoaNeed need{hdr_.need_nodes, hdr_.need_edges, hdr_.need_chunks, 0u, cap_wk_};
        { const int rc = grow(need, 0, sh_->copy_stream); if (rc != NSGPU_OK) return fail_rc_ = rc; }
        DgSlot *slot = static_cast<DgSlot *>(pin_.p);
        slot->g = view();
        slot->setup = DgSetup{cap_nodes_, cap_edges_, cap_chunks_, cap_path_, cap_wk_, cap_multi_, moved_path_ ? path_off_ : dg::NIL, sh_->dbg_flags};
        DgResult *rw = reinterpret_cast<DgResult *>(slot + 1);
        rw->status2 = 0;
        __atomic_thread_fence(__ATOMIC_RELEASE);
        static const int n_thr = [] { const char *e = getenv("NSGPU_GRAPH_THREADS"); const int v = e ? atoi(e) : 0; return v == 512 || v == 256 || v == 128 ? v : 512; }();
        {
            std::lock_guard<std::mutex> lk(sh_->serve_m);
            hipLaunchKernelGGL(dg_finish_kernel, dim3(1), dim3(n_thr), 0, sh_->copy_stream, static_cast<const DgSlot *>(slot));
            NS_HIP(hipGetLastError());
        }
        sh_->n_regrow += 1;
        return finalize(wait);
    }
    const uint32_t edges1 = hdr_.live_edges;
    hdr_ = res->hdr2;
    finalizing_ = false;
    for (Block &b : retired_) give(b);
    retired_.clear();
    if (hdr_.err) return fail_rc_ = kernel_error();
    // (num_edges() answered with the first report's count while this one was outstanding, which is exact as long as removeCycles cannot carry
    // the graph over the edge threshold from kEdgeMargin below it)
    if (hdr_.live_edges - edges1 >= kEdgeMargin) { set_error("consensus graph: one removeCycles added %u edges (internal limit %u)", hdr_.live_edges - edges1, kEdgeMargin); return fail_rc_ = NSGPU_ERR_RANGE; }
    {
        uint32_t tot = 0, worst = 0, worst_i = 0;
        for (int i = 0; i < 8; ++i) { const uint32_t d = (uint32_t)(hdr_.st_tm[i] - tm_seen_[i]); sh_->phase_ticks[i] += d; tm_seen_[i] = hdr_.st_tm[i]; tot += d; if (d > worst) worst = d, worst_i = (uint32_t)i; }
        int bk = 0;
        for (uint32_t lim = 25000; bk < 7 && tot >= lim; lim *= 2) ++bk;          // < 0.25 / 0.5 / 1 / 2 / 4 / 8 / 16 ms / more
        sh_->hist[bk] += 1;
        if (tot >= 200000) sh_->slow_phase[worst_i] += 1;                            // updates of 2 ms and more: by their longest phase
        static const bool slow_dbg = getenv("NSGPU_GRAPH_SLOW") != nullptr;
        if (slow_dbg && tot >= 800000)
            fprintf(stderr, "[graph] slow update %.1f ms (phase %u: %.1f): path %u, read len %zu, splits %u (by routes %u: %u ctx) ctx %u pops %u probes %u covering %u anc %u detours %u steps %u walked %u multi %u gap %u ended %u R %u Lf %u m %u la %u lenF %u touch_hi %u begin %lld end %lld; removeCycles in parts: marking %.1f roots %.1f walks+splits %.1f (splitPath %.1f: looking %.1f, team %.1f); routes: walk %.1f compare %.1f copies %.1f rest %.1f\n", tot / 1e5, worst_i, worst / 1e5, hdr_.m, p_len_,
                    hdr_.st_splits - dbg_seen_[0], hdr_.st_routes - dbg_seen_[11], hdr_.st_route_ctx - dbg_seen_[12], hdr_.st_ctx - dbg_seen_[1], hdr_.st_pops - dbg_seen_[2], hdr_.st_probes - dbg_seen_[3], hdr_.st_probed - dbg_seen_[4], hdr_.st_anc - dbg_seen_[5], hdr_.st_detours - dbg_seen_[6], hdr_.st_steps - dbg_seen_[7], hdr_.st_walked - dbg_seen_[8], hdr_.n_multi, hdr_.st_gap - dbg_seen_[9], hdr_.st_ended - dbg_seen_[10], hdr_.st_last[0], hdr_.st_last[1], hdr_.st_last[2], hdr_.st_last[3], hdr_.st_last[4], hdr_.st_last[5], p_begin_, p_end_, (hdr_.st_cyc[0] - cyc_seen_[0]) / 1e5, (hdr_.st_cyc[1] - cyc_seen_[1]) / 1e5, (hdr_.st_cyc[5] - cyc_seen_[5]) / 1e5, (hdr_.st_cyc[4] - cyc_seen_[4]) / 1e5, (hdr_.st_cyc[2] - cyc_seen_[2]) / 1e5, (hdr_.st_cyc[3] - cyc_seen_[3]) / 1e5, (hdr_.st_rt[0] - rt_seen_[0]) / 1e5, (hdr_.st_rt[1] - rt_seen_[1]) / 1e5, (hdr_.st_rt[2] - rt_seen_[2]) / 1e5, (hdr_.st_rt[3] - rt_seen_[3]) / 1e5);
        dbg_seen_[0] = hdr_.st_splits, dbg_seen_[1] = hdr_.st_ctx, dbg_seen_[2] = hdr_.st_pops, dbg_seen_[3] = hdr_.st_probes, dbg_seen_[4] = hdr_.st_probed, dbg_seen_[5] = hdr_.st_anc, dbg_seen_[6] = hdr_.st_detours, dbg_seen_[7] = hdr_.st_steps, dbg_seen_[8] = hdr_.st_walked, dbg_seen_[9] = hdr_.st_gap, dbg_seen_[10] = hdr_.st_ended, dbg_seen_[11] = hdr_.st_routes, dbg_seen_[12] = hdr_.st_route_ctx;
        for (int i = 0; i < 4; ++i) { sh_->rt[i] += (uint32_t)(hdr_.st_rt[i] - rt_seen_[i]); rt_seen_[i] = hdr_.st_rt[i]; }
    }
    for (int i = 0; i < 6; ++i) { sh_->cyc[i] += (uint32_t)(hdr_.st_cyc[i] - cyc_seen_[i]); cyc_seen_[i] = hdr_.st_cyc[i]; }
    sh_->cnt[0] += hdr_.st_search - cnt_seen_[0], sh_->cnt[1] += hdr_.st_steps - cnt_seen_[1], sh_->cnt[2] += hdr_.st_idscan - cnt_seen_[2], sh_->cnt[3] += hdr_.st_ctx - cnt_seen_[3], sh_->cnt[4] += hdr_.st_routes - cnt_seen_[4], sh_->cnt[5] += hdr_.st_route_ctx - cnt_seen_[5];
    cnt_seen_[0] = hdr_.st_search, cnt_seen_[1] = hdr_.st_steps, cnt_seen_[2] = hdr_.st_idscan, cnt_seen_[3] = hdr_.st_ctx, cnt_seen_[4] = hdr_.st_routes, cnt_seen_[5] = hdr_.st_route_ctx;
    sh_->n_seq_updates += hdr_.st_seq_exc - seen3_[0], sh_->n_full_walks += hdr_.st_full_walk - seen3_[1], sh_->n_splits += hdr_.st_splits - seen3_[2];
    seen3_[0] = hdr_.st_seq_exc, seen3_[1] = hdr_.st_full_walk, seen3_[2] = hdr_.st_splits;
    dbg[0] += 1, dbg[1] = dbg[0] - hdr_.st_cycles_run, dbg[2] = hdr_.st_detours, dbg[4] = hdr_.st_walked, dbg[5] = hdr_.st_cycles_run - hdr_.st_full_walk, dbg[6] = hdr_.st_splits, dbg[7] = hdr_.st_seq_exc;
    if (shadow_) { const int rc = check_against_shadow("update"); if (rc != NSGPU_OK) return fail_rc_ = rc; }
    return NSGPU_OK;
}

// numEdges for the edge-threshold tests (src/Consensus.cpp:69, 92, 196): final once the update's second report is in; until then the first
// report's count stands in where that cannot change the test's outcome
size_t DevGraph::num_edges()
{
    if (finalizing_) {
        (void)finalize(false);
        if (finalizing_ && (uint64_t)hdr_.live_edges + kEdgeMargin >= sh_->edge_thr) (void)finalize(true);
    }
    return hdr_.live_edges;
}

// NSGPU_GRAPH_CHECK: the arrays in HBM against the same update run on the host by the team of one -- ids are handed out in a fixed order, so
// every node, edge and read list must be the same
int DevGraph::check_against_shadow(const char *where)
{
    cons::SoaStore &S = shadow_->store();
    const dg::Hdr &a = hdr_, &b = S.hdr;
    { static const bool solo = getenv("NSGPU_GRAPH_SOLO") != nullptr; if (solo) NS_HIP(hipStreamSynchronize(sh_->serve_stream)); }      // (tests: the update's kernel has ended, not only reported)
    auto bad = [&](const char *what, uint64_t x, uint64_t y) { set_error("consensus graph check (%s, read %u): %s differs: device %llu, host %llu", where, (unsigned)p_id_, what, (unsigned long long)x, (unsigned long long)y); return NSGPU_ERR_RANGE; };
    if (path_ != shadow_->main_path) {
        size_t d = 0;
        while (d < path_.size() && d < shadow_->main_path.size() && path_[d] == shadow_->main_path[d]) ++d;
        set_error("consensus graph check (%s, read %u): the consensus differs: device %zu bases, host %zu, first difference at %zu; device / host: P %u / %u, S %u / %u, old_len %u / %u, new_len %u / %u, m %u / %u, path_off %u / %u, R Lf m la lenF touch_hi %u %u %u %u %u %u / %u %u %u %u %u %u, detours %u / %u, walked %u / %u, disagreements %u / %u, ended %u / %u, gap %u / %u",
                  where, (unsigned)p_id_, path_.size(), shadow_->main_path.size(), d, a.P, b.P, a.S, b.S, a.old_len, b.old_len, a.new_len, b.new_len, a.m, b.m, a.path_off, b.path_off,
                  a.st_last[0], a.st_last[1], a.st_last[2], a.st_last[3], a.st_last[4], a.st_last[5], b.st_last[0], b.st_last[1], b.st_last[2], b.st_last[3], b.st_last[4], b.st_last[5],
                  a.st_detours, b.st_detours, a.st_walked, b.st_walked, a.st_dis, b.st_dis, a.st_ended, b.st_ended, a.st_gap, b.st_gap);
        {   // where the two paths part, and what the node in front of that offers on both sides
            std::vector<dg::Node> nodes(a.n_nodes); std::vector<dg::Edge> edges(a.n_edges); std::vector<dg::Chunk> chunks(a.n_chunks); std::vector<uint32_t> pn(a.m + 1);
            (void)hipMemcpy(nodes.data(), b_nodes_.p, nodes.size() * sizeof(dg::Node), hipMemcpyDeviceToHost);
            if (!edges.empty()) (void)hipMemcpy(edges.data(), b_edges_.p, edges.size() * sizeof(dg::Edge), hipMemcpyDeviceToHost);
            if (!chunks.empty()) (void)hipMemcpy(chunks.data(), b_chunks_.p, chunks.size() * sizeof(dg::Chunk), hipMemcpyDeviceToHost);
            (void)hipMemcpy(pn.data(), static_cast<uint32_t *>(b_pn_.p) + a.path_off, ((size_t)a.m + 1) * 4, hipMemcpyDeviceToHost);
            uint32_t i = 0;
            while (i <= a.m && i <= b.m && pn[i] == S.pn[b.path_off + i]) ++i;
            std::string more = "; the paths part at node index " + std::to_string(i);
            if (i > 0 && i <= a.m) {
                const uint32_t n = pn[i - 1];
                auto list_at = [](const std::vector<dg::Chunk> &ch, const uint32_t *inl, uint32_t n_inl, uint32_t ext, uint32_t k) { if (k < n_inl) return inl[k]; k -= n_inl; uint32_t c = ext; while (k >= dg::kChunkIds) c = ch[c].next, k -= dg::kChunkIds; return ch[c].v[k]; };
                const dg::Node &x = nodes[n], &y = S.nodes[n];
                more += " behind node " + std::to_string(n) + " (n_out " + std::to_string(x.n_out) + " / " + std::to_string(y.n_out) + ", out_ext " + std::to_string(x.out_ext) + " / " + std::to_string(y.out_ext) + "): device outs";
                for (uint32_t k = 0; k < x.n_out && k < 8; ++k) { const uint32_t r = list_at(chunks, x.out, dg::kOutInl, x.out_ext, k) & dg::kRefMask; more += " " + std::to_string(r) + "(count " + std::to_string(r < edges.size() ? edges[r].count : 0u) + " sink " + std::to_string(r < edges.size() ? edges[r].sink : 0u) + ")"; }
                more += "; host outs";
                for (uint32_t k = 0; k < y.n_out && k < 8; ++k) { const uint32_t r = list_at(S.chunks, y.out, dg::kOutInl, y.out_ext, k) & dg::kRefMask; more += " " + std::to_string(r) + "(count " + std::to_string(S.edges[r].count) + " sink " + std::to_string(S.edges[r].sink) + ")"; }
                more += "; next node device " + std::to_string(pn[i]) + " host " + std::to_string(i <= b.m ? S.pn[b.path_off + i] : 0u);
            }
            std::string msg = nsgpu_last_error();
            set_error("%s%s", msg.c_str(), more.c_str());
        }
        return NSGPU_ERR_RANGE;
    }
    if (a.n_nodes != b.n_nodes) return bad("n_nodes", a.n_nodes, b.n_nodes);
    if (a.n_edges != b.n_edges) return bad("n_edges", a.n_edges, b.n_edges);
    if (a.n_chunks != b.n_chunks) return bad("n_chunks", a.n_chunks, b.n_chunks);
    if (a.live_nodes != b.live_nodes || a.live_edges != b.live_edges) return bad("live counts", a.live_edges, b.live_edges);
    if (a.m != b.m) return bad("path length", a.m, b.m);
    if (a.n_multi != b.n_multi) return bad("n_multi", a.n_multi, b.n_multi);
    if (a.P != b.P || a.S != b.S) return bad("P / S", ((uint64_t)a.P << 32) | a.S, ((uint64_t)b.P << 32) | b.S);
    if (a.initial != b.initial || a.ending_id != b.ending_id || a.starting_id != b.starting_id) return bad("initial / ending / starting", a.initial, b.initial);
    std::vector<dg::Node> nodes(a.n_nodes);
    std::vector<dg::Edge> edges(a.n_edges);
    std::vector<dg::Chunk> chunks(a.n_chunks);
    std::vector<uint32_t> pe(a.m), pn(a.m + 1);
    NS_HIP(hipMemcpy(nodes.data(), b_nodes_.p, nodes.size() * sizeof(dg::Node), hipMemcpyDeviceToHost));
    if (!edges.empty()) NS_HIP(hipMemcpy(edges.data(), b_edges_.p, edges.size() * sizeof(dg::Edge), hipMemcpyDeviceToHost));
    if (!chunks.empty()) NS_HIP(hipMemcpy(chunks.data(), b_chunks_.p, chunks.size() * sizeof(dg::Chunk), hipMemcpyDeviceToHost));
    if (a.m) NS_HIP(hipMemcpy(pe.data(), static_cast<uint32_t *>(b_pe_.p) + a.path_off, (size_t)a.m * 4, hipMemcpyDeviceToHost));
    NS_HIP(hipMemcpy(pn.data(), static_cast<uint32_t *>(b_pn_.p) + a.path_off, ((size_t)a.m + 1) * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < a.m; ++i) if (pe[i] != S.pe[b.path_off + i]) return bad("a path edge", pe[i], S.pe[b.path_off + i]);
    for (uint32_t i = 0; i <= a.m; ++i) if (pn[i] != S.pn[b.path_off + i]) return bad("a path node", pn[i], S.pn[b.path_off + i]);
    auto list_at = [](const std::vector<dg::Chunk> &ch, const uint32_t *inl, uint32_t n_inl, uint32_t ext, uint32_t i) { if (i < n_inl) return inl[i]; i -= n_inl; uint32_t c = ext; while (i >= dg::kChunkIds) c = ch[c].next, i -= dg::kChunkIds; return ch[c].v[i]; };
    for (uint32_t n = 0; n < a.n_nodes; ++n) {
        const dg::Node &x = nodes[n], &y = S.nodes[n];
        if (x.n_out != y.n_out || x.n_in != y.n_in || x.base != y.base || x.on_main != y.on_main) { set_error("consensus graph check (%s, read %u): the header of node %u differs: device n_out %u n_in %u base %u on_main %u, host n_out %u n_in %u base %u on_main %u (of %u nodes, %u before this update; splits by routes so far %u / %u, their contexts %u / %u, all contexts %u / %u)", where, (unsigned)p_id_, n, x.n_out, x.n_in, x.base, x.on_main, y.n_out, y.n_in, y.base, y.on_main, a.n_nodes, a.upd_nodes0, a.st_routes, b.st_routes, a.st_route_ctx, b.st_route_ctx, a.st_ctx, b.st_ctx); return NSGPU_ERR_RANGE; }
        for (uint32_t i = 0; i < x.n_out; ++i) if (list_at(chunks, x.out, dg::kOutInl, x.out_ext, i) != list_at(S.chunks, y.out, dg::kOutInl, y.out_ext, i)) return bad("an out reference of node", n, i);
        for (uint32_t i = 0; i < x.n_in; ++i) if (list_at(chunks, x.in, dg::kInInl, x.in_ext, i) != list_at(S.chunks, y.in, dg::kInInl, y.in_ext, i)) {
            std::string da, ho;
            for (uint32_t q = 0; q < x.n_in; ++q) { da += " " + std::to_string(list_at(chunks, x.in, dg::kInInl, x.in_ext, q)); ho += " " + std::to_string(list_at(S.chunks, y.in, dg::kInInl, y.in_ext, q)); }
            set_error("consensus graph check (%s, read %u): in reference %u of node %u differs (on_main %u, n_in %u, in_ext %u / %u, n_chunks %u, nodes %u, edges %u): device%s; host%s; splits by routes so far %u / %u, chain runs+routes ctx %u / %u", where, (unsigned)p_id_, i, n, x.on_main, x.n_in, x.in_ext, y.in_ext, a.n_chunks, a.n_nodes, a.n_edges, da.c_str(), ho.c_str(), a.st_routes, b.st_routes, a.st_ctx, b.st_ctx);
            return NSGPU_ERR_RANGE;
        }
    }
    for (uint32_t e = 0; e < a.n_edges; ++e) {
        const dg::Edge &x = edges[e], &y = S.edges[e];
        if (x.src != y.src || x.sink != y.sink || x.count != y.count) return bad("an edge's header", e, x.count);
        for (uint32_t p = 0; p < x.count; ++p) if (list_at(chunks, x.ids, dg::kEdgeInl, x.head, p) != list_at(S.chunks, y.ids, dg::kEdgeInl, y.head, p)) return bad("a read id of edge", e, p);
    }
    return NSGPU_OK;
}

// the finished contig: its arrays on the way back to the host (pinned blocks of the pool), emitted from there by whoever runs the emission
int DevGraph::emit_begin()
{
    if (e_begun_ || !inited_) return NSGPU_OK;
    NS_CHECK(!pending_, NSGPU_ERR_ARG, "consensus graph: emission with an update in flight (internal error)");
    NS_TRY(finalize(true));
    const dg::Hdr &h = hdr_;       // (the path lies at path_off_: a prepare() whose update never came may have moved the arrays since the header was reported)
    NS_TRY(take(e_nodes_, std::max<size_t>(64, (size_t)h.n_nodes * sizeof(dg::Node)), true));
    NS_TRY(take(e_edges_, std::max<size_t>(64, (size_t)h.n_edges * sizeof(dg::Edge)), true));
    NS_TRY(take(e_chunks_, std::max<size_t>(64, (size_t)h.n_chunks * sizeof(dg::Chunk)), true));
    NS_TRY(take(e_pe_, ((size_t)h.m + 1) * 4, true)); NS_TRY(take(e_pn_, ((size_t)h.m + 1) * 4, true)); NS_TRY(take(e_ps_, (size_t)h.m + 1, true));
    hipStream_t cs = sh_->copy_stream;
    // (a prepare() whose update never came may have left array copies queued on the serve stream: the copies back come behind them)
    if (!e_ev_) NS_HIP(hipEventCreateWithFlags(&e_ev_, hipEventDisableTiming));
    NS_HIP(hipEventRecord(e_ev_, sh_->serve_stream));
    NS_HIP(hipStreamWaitEvent(cs, e_ev_, 0));
    NS_HIP(hipMemcpyAsync(e_nodes_.p, b_nodes_.p, (size_t)h.n_nodes * sizeof(dg::Node), hipMemcpyDeviceToHost, cs));
    if (h.n_edges) NS_HIP(hipMemcpyAsync(e_edges_.p, b_edges_.p, (size_t)h.n_edges * sizeof(dg::Edge), hipMemcpyDeviceToHost, cs));
    if (h.n_chunks) NS_HIP(hipMemcpyAsync(e_chunks_.p, b_chunks_.p, (size_t)h.n_chunks * sizeof(dg::Chunk), hipMemcpyDeviceToHost, cs));
    if (h.m) NS_HIP(hipMemcpyAsync(e_pe_.p, static_cast<uint32_t *>(b_pe_.p) + path_off_, (size_t)h.m * 4, hipMemcpyDeviceToHost, cs));
    NS_HIP(hipMemcpyAsync(e_pn_.p, static_cast<uint32_t *>(b_pn_.p) + path_off_, ((size_t)h.m + 1) * 4, hipMemcpyDeviceToHost, cs));
    NS_HIP(hipMemcpyAsync(e_ps_.p, static_cast<uint8_t *>(b_ps_.p) + path_off_, (size_t)h.m + 1, hipMemcpyDeviceToHost, cs));
    if (!e_ev_) NS_HIP(hipEventCreateWithFlags(&e_ev_, hipEventDisableTiming));
    NS_HIP(hipEventRecord(e_ev_, cs));
    sh_->bytes_back += (uint64_t)h.n_nodes * sizeof(dg::Node) + (uint64_t)h.n_edges * sizeof(dg::Edge) + (uint64_t)h.n_chunks * sizeof(dg::Chunk) + (uint64_t)h.m * 9;
    e_begun_ = true;
    return NSGPU_OK;
}

void DevGraph::write_reads(cons::StreamSet &o, const std::function<cons::ReadBases(cons::read_t)> *source)
{
    if (!e_begun_ && emit_begin() != NSGPU_OK) { fprintf(stderr, "nsgpu: %s\n", nsgpu_last_error()); abort(); }
    if (event_wait(e_ev_) != hipSuccess) { fprintf(stderr, "nsgpu: a finished contig's arrays did not arrive on the host\n"); abort(); }
    // the arrays in HBM are no longer needed
    Block *dev[] = {&b_nodes_, &b_mark_, &b_pidx_, &b_edges_, &b_chunks_, &b_pe_, &b_pn_, &b_ps_, &b_sve_, &b_svn_, &b_svs_, &b_multi_, &b_wk_, &b_hdr_};
    for (Block *b : dev) give(*b);
    dg::Hdr h = hdr_;
    h.path_off = 0;
    dg::G g;
    memset(&g, 0, sizeof(g));
    g.h = &h;
    g.nodes = static_cast<dg::Node *>(e_nodes_.p), g.edges = static_cast<dg::Edge *>(e_edges_.p), g.chunks = static_cast<dg::Chunk *>(e_chunks_.p);
    g.pe = static_cast<uint32_t *>(e_pe_.p), g.pn = static_cast<uint32_t *>(e_pn_.p), g.ps = static_cast<uint8_t *>(e_ps_.p);
    cons::SoaEmitter em(g, reads_);
    em.write_reads(o, source);
    Block *pin[] = {&e_nodes_, &e_edges_, &e_chunks_, &e_pe_, &e_pn_, &e_ps_};
    for (Block *b : pin) give(*b, true);
}

}  // namespace nsgpu
