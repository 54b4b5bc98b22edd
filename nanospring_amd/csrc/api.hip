// api.hip -- the C-ABI of libnsgpu.so (include/nsgpu.h): context life cycle, the
// HBM read store and the MinHash stages.  No CPU fallback exists: without a
// gfx950 device nsgpu_create fails with NSGPU_ERR_NODEV.
#include "common.hpp"
#include <atomic>
#include "host_util.hpp"
#include <cstdarg>
#include <chrono>
#include <time.h>
#include <sys/prctl.h>
#include <algorithm>

namespace nsgpu {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    static const bool loud = getenv("NSGPU_CONS_DEBUG") != nullptr;      // (an error raised on a worker thread reaches the caller as a code only)
    if (loud) fprintf(stderr, "[nsgpu] error: %s\n", g_err);
}

// NSGPU_WAIT_TIMEOUT_S=n: a wait for the GPU that lasts longer than n seconds fails (hipErrorNotReady, with a message saying so)
// instead of blocking for ever -- the tests run with it, so that a lost kernel shows up as an error, not as a hung process.
static double wait_timeout_s()
{
    static const double t = [] { const char *e = getenv("NSGPU_WAIT_TIMEOUT_S"); return e ? atof(e) : 0.0; }();
    return t;
}
// NSGPU_SEGV_TRACE=1 (debugging aid): a backtrace of the faulting thread on stderr before the process dies of SIGSEGV / SIGBUS / SIGABRT
#include <execinfo.h>
#include <signal.h>
static void segv_trace(int sig)
{
    void *bt[64];
    const int n = backtrace(bt, 64);
    const char msg[] = "nsgpu: fatal signal, backtrace of the faulting thread:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
static const bool g_segv_trace = [] {
    if (!getenv("NSGPU_SEGV_TRACE")) return false;
    signal(SIGSEGV, segv_trace), signal(SIGBUS, segv_trace), signal(SIGABRT, segv_trace);
    return true;
}();
static std::atomic<uint64_t> g_host_waits{0};       // host waits for GPU work (streams and events), process-wide: nsgpu_host_wait_count

static hipError_t stream_wait_impl(hipStream_t s, int spin_us, bool spin_only = false)
{
    g_host_waits.fetch_add(1, std::memory_order_relaxed);
    const double limit = wait_timeout_s();
    if (spin_only && limit <= 0) return hipStreamSynchronize(s);
    // the default timer slack (50 us) would stretch every 20 us sleep to ~75 us: 1 us of slack for threads that wait here
    static thread_local const int slack_set = prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    (void)slack_set;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (limit > 0 && dt > std::chrono::duration<double>(limit)) {
            fprintf(stderr, "nsgpu: a wait for stream %p exceeded NSGPU_WAIT_TIMEOUT_S = %g s\n", (void *)s, limit);
            return hipErrorNotReady;
        }
        if (spin_only || dt < std::chrono::microseconds(spin_us)) continue;
        timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
}
hipError_t stream_wait(hipStream_t s) { return stream_wait_impl(s, 20); }
// the same for an event (a point inside a stream's work): poll, sleep between polls
hipError_t event_wait(hipEvent_t ev)
{
    g_host_waits.fetch_add(1, std::memory_order_relaxed);
    const double limit = wait_timeout_s();
    static thread_local const int slack_set = prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    (void)slack_set;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (limit > 0 && dt > std::chrono::duration<double>(limit)) { fprintf(stderr, "nsgpu: a wait for an event exceeded NSGPU_WAIT_TIMEOUT_S = %g s\n", limit); return hipErrorNotReady; }
        if (dt < std::chrono::microseconds(20)) continue;
        timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
}
// For the waits between the short kernels of one batch step (a few hundred microseconds of GPU work, several times per
// pipeline slot, on the slot's critical path): the runtime's own busy-wait.  Measured at cfg2: sleeping between polls costs
// 100-200 us per wait until the thread is back on a core of the CPU-saturated cgroup (window queries 0.53 -> 1.08 s per
// step), polling hipStreamQuery for 0.5 ms first still 0.67 s; the spin costs ~1 CPU-second per batch thread and step.
hipError_t stream_wait_short(hipStream_t s)
{
    return stream_wait_impl(s, 0, true);
}

static uint64_t row_bytes_h(uint32_t len) { return ((((uint64_t)len + 3) / 4 + 15) & ~(uint64_t)15) + 16; }

// decide the HBM layout of a set of sequences and upload offsets/lengths
static int store_prepare(nsgpu_ctx *c, SeqStore &st, const uint32_t *len, uint32_t n)
{
    st.n = n;
    st.h_len.assign(len, len + n);
    st.h_poff.resize((size_t)n + 1);
    uint64_t off = 0, nb = 0;
    uint32_t mx = 0;
    for (uint32_t r = 0; r < n; ++r) {
        st.h_poff[r] = off;
        off += row_bytes_h(len[r]);
        nb += len[r];
        mx = std::max(mx, len[r]);
    }
    st.h_poff[n] = off;
    st.packed_bytes = off;
    st.n_bases = nb;
    st.max_len = mx;
    NS_TRY(st.packed.reserve(off + 64));
    NS_TRY(st.poff.reserve(((size_t)n + 1) * 8));
    NS_TRY(st.len.reserve(((size_t)n + 1) * 4));
    NS_HIP(hipMemcpyAsync(st.poff.p, st.h_poff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    if (n) NS_HIP(hipMemcpyAsync(st.len.p, st.h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    return NSGPU_OK;
}

int store_prepare_lens(nsgpu_ctx *c, SeqStore &st, const uint32_t *len, uint32_t n) { return store_prepare(c, st, len, n); }
static int store_from_ascii(nsgpu_ctx *c, SeqStore &st, const char *bases, const uint64_t *off, uint32_t n);
int store_from_ascii_shard(nsgpu_ctx *c, SeqStore &st, const char *bases, const uint64_t *off, uint32_t n) { return store_from_ascii(c, st, bases, off, n); }   // dist.hip

static int store_from_ascii(nsgpu_ctx *c, SeqStore &st, const char *bases, const uint64_t *off, uint32_t n)
{
    std::vector<uint32_t> len(n);
    for (uint32_t r = 0; r < n; ++r) {
        NS_CHECK(off[r + 1] >= off[r], NSGPU_ERR_ARG, "offsets must be non-decreasing (read %u)", r);
        NS_CHECK(off[r + 1] - off[r] <= 0xFFFFFFF0ull, NSGPU_ERR_RANGE, "read %u longer than 2^32-16 bases", r);
        len[r] = (uint32_t)(off[r + 1] - off[r]);
    }
    NS_TRY(store_prepare(c, st, len.data(), n));
    const uint64_t total = n ? off[n] - off[0] : 0;
    NS_TRY(c->ascii.reserve(total + 64));
    NS_TRY(c->aoff.reserve(((size_t)n + 1) * 8));
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t r = 0; r <= n; ++r) rel[r] = off[r] - off[0];
    if (total) NS_HIP(hipMemcpyAsync(c->ascii.p, bases + off[0], total, hipMemcpyHostToDevice, c->stream));
    NS_HIP(hipMemcpyAsync(c->aoff.p, rel.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    NS_HIP(hipEventRecord(c->t_kernel.a, c->stream));
    NS_TRY(launch_pack_ascii(c, c->ascii.as<char>(), c->aoff.as<uint64_t>(), st));
    NS_HIP(hipEventRecord(c->t_kernel.b, c->stream));
    NS_HIP(stream_wait_short(c->stream));   // rel / len vectors go out of scope
    NS_HIP(hipEventElapsedTime(&c->timing.pack_ms, c->t_kernel.a, c->t_kernel.b));
    return NSGPU_OK;
}

}  // namespace nsgpu

using namespace nsgpu;

namespace nsgpu {
// Stream priorities.  Measured on MI355X / ROCm 7.2 (profiles/r02_stream_priority_ab.txt): the role defaults below.
int role_stream_create(hipStream_t *st, const char *role)
{
    int least = 0, greatest = 0;
    NS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    std::string var = std::string("NSGPU_PRIO_") + role;
    for (char &ch : var) ch = (char)toupper((unsigned char)ch);
    const char *e = getenv(var.c_str());
    // (the runtime pools its hardware queues by priority: streams of different priorities never share one.  The graph streams are kept out of the
    // pool of the streams the DP results come by: a serve launch holds its queue until the slot's last order, and whatever is enqueued behind it on
    // the same hardware queue waits that long -- the kernels its orders depend on must never be there.)
    std::string want = e ? e : (!strcmp(role, "sketch") ? "lo" : !strcmp(role, "seeds") ? "hi" : !strcmp(role, "dp_side") ? "hi" : !strcmp(role, "graph") || !strcmp(role, "graph_copy") ? "lo" : "mid");
    const int prio = want == "lo" ? least : want == "hi" ? greatest : (least + greatest) / 2;
    NS_HIP(hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio));
    return NSGPU_OK;
}
}  // namespace nsgpu

namespace nsgpu {
// The host copy of the reads that the contig engine works from is the folded ASCII text (1 B/base) by default.  For inputs where that
// matters -- every rank of a multi-GPU run holds ALL reads: 50 GB per rank at BASELINE cfg4 -- it can be the packed rows instead,
// copied back from HBM once: NSGPU_PACKED_MIRROR=1 (0: never; unset: from 16 Gbases on).
int mirror_finalize(nsgpu_ctx *c, bool force_packed)
{
    static const char *e = getenv("NSGPU_PACKED_MIRROR");
    const bool want = force_packed || (e ? atoi(e) != 0 : c->reads.n_bases >= (16ull << 30));
    c->packed_mirror = false;
    c->h_packed.clear(), c->h_packed.shrink_to_fit();
    if (!want || c->reads.n == 0) return NSGPU_OK;
    c->h_packed.resize(c->reads.packed_bytes + 64);
    NS_HIP(hipMemcpyAsync(c->h_packed.data(), c->reads.packed.p, c->reads.packed_bytes, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait(c->stream));
    c->h_bases.clear(), c->h_bases.shrink_to_fit();
    c->packed_mirror = true;
    return NSGPU_OK;
}
// read r as the engine sees it (folded A/T/C/G): a pointer into the ASCII mirror, or the packed row decoded into `buf`
const char *mirror_read(const nsgpu_ctx *c, uint32_t r, std::string &buf)
{
    if (!c->packed_mirror) return c->h_bases.data() + c->h_off[r];
    static const struct Lut { uint32_t w[256]; Lut() { static const char dna[4] = {'A', 'T', 'C', 'G'};
        for (int b = 0; b < 256; ++b) { char t[4] = {dna[b >> 6 & 3], dna[b >> 4 & 3], dna[b >> 2 & 3], dna[b & 3]}; memcpy(&w[b], t, 4); } } } lut;
    const size_t L = (size_t)(c->h_off[r + 1] - c->h_off[r]);
    buf.resize((L + 3) & ~(size_t)3);
    const uint8_t *row = c->h_packed.data() + c->reads.h_poff[r];
    char *d = &buf[0];
    for (size_t k = 0; k < (L + 3) / 4; ++k) memcpy(d + 4 * k, &lut.w[row[k]], 4);
    buf.resize(L);
    return buf.data();
}
}  // namespace nsgpu

extern "C" {

const char *nsgpu_last_error(void) { return g_err; }
const char *nsgpu_version(void) { return "nsgpu 0.1 (gfx950, hipcc " __VERSION__ ")"; }
void nsgpu_free(void *p) { free(p); }

void nsgpu_default_params(nsgpu_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->k = 23; p->n = 60; p->overlap_sketch_thr = 6;          // src/main.cpp:52-60
    p->m_k = 20; p->m_w = 50; p->max_chain_iter = 400;        // src/main.cpp:62-70
    p->edge_threshold = 4000000;                              // src/main.cpp:72
    p->device = 0;
}

int nsgpu_create(const nsgpu_params *p, nsgpu_ctx **ctx_out)
{
    NS_CHECK(p && ctx_out, NSGPU_ERR_ARG, "nsgpu_create: null argument");
    NS_CHECK(p->k >= 1 && p->k <= 31, NSGPU_ERR_ARG, "k must be in 1..31 (2k-bit k-mers in a u64; k=32 is UB in the reference too, src/ReadFilter.cpp:145)");
    NS_CHECK(p->n >= 1 && p->n <= 256, NSGPU_ERR_ARG, "n (sketch size) must be in 1..256");
    // The contig stage drives ~30 streams (sketch, seeds + chaining, DP main and side streams per builder group, window queries).
    // The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), in order within a queue: a 0.3 ms
    // sketch then waits behind another group's 3 ms DP kernel.  8 queues: +12 % whole path (profiles/r02_stream_priority_ab.txt).
    // Read by the runtime when it initialises, i.e. effective here only if this is the process's first HIP call; set it in the
    // process environment otherwise.
    if (!getenv("NSGPU_NO_SETENV")) setenv("GPU_MAX_HW_QUEUES", "8", 0);       // (a host that manages its own environment opts out)
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device visible (%s); libnsgpu has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return NSGPU_ERR_NODEV;
    }
    NS_CHECK(p->device >= 0 && p->device < ndev, NSGPU_ERR_ARG, "device %d out of range (have %d)", p->device, ndev);
    NS_HIP(hipSetDevice(p->device));
    hipDeviceProp_t prop;
    NS_HIP(hipGetDeviceProperties(&prop, p->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libnsgpu is built for gfx950 only", p->device, prop.gcnArchName);
        return NSGPU_ERR_NODEV;
    }
    {
        char bus[64] = "";
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), p->device) == hipSuccess) pool_bind_to_gpu_node(bus);
    }
    nsgpu_ctx *c = new nsgpu_ctx();
    c->prm = *p;
    c->n_cu = prop.multiProcessorCount;
    memset(&c->timing, 0, sizeof(c->timing));
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; set_error("hipStreamCreate failed"); return NSGPU_ERR_HIP; }
    c->stream = c->own_stream;
    if (c->t_stage.init() != NSGPU_OK || c->t_kernel.init() != NSGPU_OK) { delete c; return NSGPU_ERR_HIP; }
    *ctx_out = c;
    return NSGPU_OK;
}

void nsgpu_destroy(nsgpu_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->prm.device);
    (void)stream_wait(c->stream);
    c->reads.release(); c->queries.release();
    DevBuf *bufs[] = {&c->ascii, &c->aoff, &c->salts, &c->sketch, &c->sketch_rc, &c->qsketch, &c->idx_keys, &c->idx_ids, &c->idx_tmp_k,
                      &c->idx_tmp_v, &c->idx_tmp_e, &c->idx_tmp_e2, &c->idx_sort_ws, &c->f_pool, &c->f_qstart, &c->f_qcnt, &c->f_qm, &c->f_off,
                      &c->f_ids, &c->f_ctrl, &c->f_ovf_list, &c->f_ovf_cnt, &c->f_scan_ws, &c->rep_flags, &c->fq_cnt, &c->fq_base, &c->fq_nlpos, &c->fq_len};
    for (DevBuf *b : bufs) b->release();
    c->t_stage.destroy(); c->t_kernel.destroy();
    if (c->cons_engine) c->cons_engine_free(c->cons_engine);
    if (c->graph_shared) c->graph_shared_free(c->graph_shared);
    for (nsgpu_ctx::SketchWs &w : c->sws) {
        DevBuf *sb[] = {&w.seqs, &w.soff, &w.len, &w.sob, &w.vf, &w.mk, &w.vr, &w.linv, &w.npf, &w.pushf, &w.npr, &w.pr, &w.V, &w.hk, &w.PX, &w.PY, &w.PRUN,
                        &w.PSEQ, &w.rm, &w.nout, &w.oscan, &w.off, &w.out, &w.scan_ws};
        for (DevBuf *b : sb) b->release();
        if (w.h_seqs) (void)hipHostFree(w.h_seqs);
        if (w.h_out) (void)hipHostFree(w.h_out);
        w.h_meta.release();
        w.h_concat.release();
        w.h_compact.release();
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    for (nsgpu_ctx::KswWs &w : c->kws) {
        if (w.stream) (void)stream_wait(w.stream);
        DevBuf *kb[] = {&w.k_tasks, &w.k_order, &w.k_seqs, &w.k_p, &w.k_cig, &w.k_res, &w.k_slab, &w.k_ncig, &w.k_coff, &w.k_cig2, &w.scan_ws};
        for (DevBuf *b : kb) b->release();
        for (int i = 0; i < 3; ++i) { if (w.side_stream[i]) (void)hipStreamDestroy(w.side_stream[i]); if (w.side_done[i]) (void)hipEventDestroy(w.side_done[i]); }
        if (w.side_fork) (void)hipEventDestroy(w.side_fork);
        if (w.t_a) (void)hipEventDestroy(w.t_a);
        if (w.t_b) (void)hipEventDestroy(w.t_b);
        for (hipEvent_t e : w.ev) if (e) (void)hipEventDestroy(e);
        if (w.stream) (void)hipStreamDestroy(w.stream);
        if (w.h_pool) (void)hipHostFree(w.h_pool);
        w.h_res.release(); w.h_coff.release(); w.h_cig.release();
    }
    for (nsgpu_ctx::ChainWs &w : c->cws) {
        w.d_in.release(); w.d_out.release(); w.d_marks.release(); w.h_in.release(); w.h_out.release();
        if (w.stream) (void)hipStreamDestroy(w.stream);
        if (w.stream2) (void)hipStreamDestroy(w.stream2);
    }
    for (nsgpu_ctx::SeedWs &w : c->seed_ws) {
        w.d_tab.release(); w.d_next.release(); w.d_ys.release(); w.d_tmp.release(); w.d_out.release(); w.d_counter.release();
        w.h_pairs.release(); w.h_res.release(); w.h_ref.release(); w.h_jobs.release();
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    c->pin_small.release(); c->pin_foff.release(); c->pin_fids.release(); c->pin_wq.release(); c->pin_wq_out.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int nsgpu_set_stream(nsgpu_ctx *c, void *s)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_HIP(stream_wait(c->stream));
    c->stream = s ? reinterpret_cast<hipStream_t>(s) : c->own_stream;
    return NSGPU_OK;
}

int nsgpu_sync(nsgpu_ctx *c)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_HIP(stream_wait(c->stream));
    return NSGPU_OK;
}

int nsgpu_load_reads_ascii(nsgpu_ctx *c, const char *bases, const uint64_t *off, uint32_t n)
{
    NS_CHECK(c && off && (bases || n == 0 || off[n] == off[0]), NSGPU_ERR_ARG, "nsgpu_load_reads_ascii: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    c->have_sketch = c->have_index = c->have_filter_all = c->have_cons = false;
    NS_TRY(store_from_ascii(c, c->reads, bases, off, n));
    // host mirror, folded exactly like DnaBitset (src/dnaToBits.cpp:6-8, 81-98)
    static const char dna[4] = {'A', 'T', 'C', 'G'};
    const uint64_t total = n ? off[n] - off[0] : 0;
    c->h_bases.resize(total + 1);
    c->h_off.resize((size_t)n + 1);
    for (uint32_t r = 0; r <= n; ++r) c->h_off[r] = off[r] - off[0];
    const char *src = bases + (n ? off[0] : 0);
    for (uint64_t i = 0; i < total; ++i) c->h_bases[i] = dna[(src[i] & 2) | ((src[i] & 4) >> 2)];
    return mirror_finalize(c);
}

int nsgpu_load_reads_packed(nsgpu_ctx *c, const uint8_t *packed, const uint64_t *byte_off, const uint32_t *len, uint32_t n)
{
    NS_CHECK(c && (n == 0 || (packed && byte_off && len)), NSGPU_ERR_ARG, "nsgpu_load_reads_packed: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    c->have_sketch = c->have_index = c->have_filter_all = false;
    NS_TRY(store_prepare(c, c->reads, len, n));
    SeqStore &st = c->reads;
    // DnaBitset bytes are already in the row format; rows only need re-basing to
    // 16-byte aligned, zero padded slots.
    std::vector<uint8_t> stage(st.packed_bytes + 64, 0);
    for (uint32_t r = 0; r < n; ++r) memcpy(stage.data() + st.h_poff[r], packed + byte_off[r], ((size_t)len[r] + 3) / 4);
    NS_HIP(hipMemcpyAsync(st.packed.p, stage.data(), st.packed_bytes, hipMemcpyHostToDevice, c->stream));
    NS_HIP(stream_wait(c->stream));
    {
        static const char dna[4] = {'A', 'T', 'C', 'G'};
        c->h_off.resize((size_t)n + 1);
        uint64_t tot = 0;
        for (uint32_t r = 0; r < n; ++r) { c->h_off[r] = tot; tot += len[r]; }
        c->h_off[n] = tot;
        c->h_bases.resize(tot + 1);
        for (uint32_t r = 0; r < n; ++r) {
            const uint8_t *pk = packed + byte_off[r];
            char *dst = c->h_bases.data() + c->h_off[r];
            for (uint32_t i = 0; i < len[r]; ++i) dst[i] = dna[(pk[i >> 2] >> (6 - 2 * (i & 3))) & 3];
        }
        c->have_cons = false;
    }
    return mirror_finalize(c);
}

uint32_t nsgpu_num_reads(const nsgpu_ctx *c) { return c ? c->reads.n : 0; }
uint64_t nsgpu_num_bases(const nsgpu_ctx *c) { return c ? c->reads.n_bases : 0; }

int nsgpu_get_read_packed(nsgpu_ctx *c, uint32_t r, uint8_t *out, uint32_t *len_out)
{
    NS_CHECK(c && out, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(r < c->reads.n, NSGPU_ERR_ARG, "read id %u out of range (%u reads)", r, c->reads.n);
    const uint32_t L = c->reads.h_len[r];
    if (len_out) *len_out = L;
    const size_t nb = ((size_t)L + 3) / 4;
    if (nb) NS_HIP(hipMemcpyAsync(out, c->reads.packed.as<uint8_t>() + c->reads.h_poff[r], nb, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait(c->stream));
    return NSGPU_OK;
}

int nsgpu_get_read(nsgpu_ctx *c, uint32_t r, char *out, uint32_t *len_out)
{
    NS_CHECK(c && out, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(r < c->reads.n, NSGPU_ERR_ARG, "read id %u out of range (%u reads)", r, c->reads.n);
    const uint32_t L = c->reads.h_len[r];
    std::vector<uint8_t> pk(((size_t)L + 3) / 4 + 1);
    NS_TRY(nsgpu_get_read_packed(c, r, pk.data(), len_out));
    static const char dna[4] = {'A', 'T', 'C', 'G'};          // DnaBitset::to_string, src/dnaToBits.cpp:81-98
    for (uint32_t i = 0; i < L; ++i) out[i] = dna[(pk[i >> 2] >> (6 - 2 * (i & 3))) & 3];
    return NSGPU_OK;
}

int nsgpu_sketch(nsgpu_ctx *c, const uint64_t *salts, uint64_t *sketches_out)
{
    NS_CHECK(c && salts, NSGPU_ERR_ARG, "nsgpu_sketch: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    const uint32_t N = c->reads.n, n = c->prm.n;
    NS_TRY(c->salts.reserve((size_t)n * 8));
    NS_HIP(hipMemcpyAsync(c->salts.p, salts, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    c->have_salts = true;
    NS_TRY(c->sketch.reserve(((size_t)N * n + 1) * 8));
    NS_HIP(hipEventRecord(c->t_stage.a, c->stream));
    NS_TRY(launch_sketch(c, c->reads, c->sketch.as<uint64_t>(), nullptr));
    NS_HIP(hipEventRecord(c->t_stage.b, c->stream));
    if (sketches_out && N) NS_HIP(hipMemcpyAsync(sketches_out, c->sketch.p, (size_t)N * n * 8, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait(c->stream));
    NS_HIP(hipEventElapsedTime(&c->timing.sketch_kernel_ms, c->t_stage.a, c->t_stage.b));
    c->timing.sketch_ms = c->timing.sketch_kernel_ms;
    c->have_sketch = true;
    c->have_index = c->have_filter_all = false;
    return NSGPU_OK;
}

int nsgpu_sketch_range(nsgpu_ctx *c, const uint64_t *salts, uint32_t lo, uint32_t hi)
{
    NS_CHECK(c && salts, NSGPU_ERR_ARG, "nsgpu_sketch_range: null argument");
    NS_CHECK(lo <= hi && hi <= c->reads.n, NSGPU_ERR_ARG, "nsgpu_sketch_range: bad range");
    NS_HIP(hipSetDevice(c->prm.device));
    const uint32_t N = c->reads.n, n = c->prm.n;
    NS_TRY(c->salts.reserve((size_t)n * 8));
    NS_HIP(hipMemcpyAsync(c->salts.p, salts, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    c->have_salts = true;
    NS_TRY(c->sketch.reserve(((size_t)N * n + 1) * 8));
    NS_HIP(hipEventRecord(c->t_stage.a, c->stream));
    NS_TRY(launch_sketch_range(c, c->reads, lo, hi, c->sketch.as<uint64_t>()));
    NS_HIP(hipEventRecord(c->t_stage.b, c->stream));
    NS_HIP(stream_wait(c->stream));
    NS_HIP(hipEventElapsedTime(&c->timing.sketch_kernel_ms, c->t_stage.a, c->t_stage.b));
    c->timing.sketch_ms = c->timing.sketch_kernel_ms;
    c->have_index = c->have_filter_all = false;
    return NSGPU_OK;
}

static int sketch_rows_copy(nsgpu_ctx *c, uint32_t lo, uint32_t hi, void *buf, int buf_on_device, bool to_buf)
{
    NS_CHECK(c && buf, NSGPU_ERR_ARG, "sketch rows: null argument");
    NS_CHECK(lo <= hi && hi <= c->reads.n, NSGPU_ERR_ARG, "sketch rows: bad range");
    NS_HIP(hipSetDevice(c->prm.device));
    const size_t n = c->prm.n, bytes = (size_t)(hi - lo) * n * 8;
    NS_TRY(c->sketch.reserve(((size_t)c->reads.n * n + 1) * 8));
    uint8_t *rows = c->sketch.as<uint8_t>() + (size_t)lo * n * 8;
    const hipMemcpyKind kind = buf_on_device ? hipMemcpyDeviceToDevice : (to_buf ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
    if (bytes) NS_HIP(hipMemcpyAsync(to_buf ? buf : (void *)rows, to_buf ? (const void *)rows : buf, bytes, kind, c->stream));
    NS_HIP(stream_wait(c->stream));
    return NSGPU_OK;
}

int nsgpu_sketch_rows_get(nsgpu_ctx *c, uint32_t lo, uint32_t hi, void *dst, int dst_on_device) { return sketch_rows_copy(c, lo, hi, dst, dst_on_device, true); }

int nsgpu_sketch_rows_set(nsgpu_ctx *c, uint32_t lo, uint32_t hi, const void *src, int src_on_device)
{
    NS_TRY(sketch_rows_copy(c, lo, hi, const_cast<void *>(src), src_on_device, false));
    if (lo == 0 && hi == c->reads.n) c->have_sketch = true;
    return NSGPU_OK;
}

int nsgpu_sketch_mark_complete(nsgpu_ctx *c)
{
    NS_CHECK(c && c->have_salts, NSGPU_ERR_ARG, "nsgpu_sketch_mark_complete: no salts / sketch rows yet");
    c->have_sketch = true;
    c->have_index = c->have_filter_all = false;
    return NSGPU_OK;
}

int nsgpu_build_index(nsgpu_ctx *c)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(c->have_sketch, NSGPU_ERR_ARG, "nsgpu_build_index: call nsgpu_sketch first");
    NS_HIP(hipSetDevice(c->prm.device));
    NS_HIP(hipEventRecord(c->t_stage.a, c->stream));
    NS_TRY(build_index(c));
    NS_HIP(hipEventRecord(c->t_stage.b, c->stream));
    NS_HIP(stream_wait(c->stream));
    NS_HIP(hipEventElapsedTime(&c->timing.index_ms, c->t_stage.a, c->t_stage.b));
    c->have_index = true;
    c->have_filter_all = false;
    return NSGPU_OK;
}

int nsgpu_index_export(nsgpu_ctx *c, uint32_t j, uint64_t *keys_out, uint32_t *start_out, uint32_t *ids_out, uint32_t *nkeys_out)
{
    NS_CHECK(c && keys_out && start_out && ids_out && nkeys_out, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(c->have_index, NSGPU_ERR_ARG, "nsgpu_index_export: no index");
    NS_CHECK(j < c->prm.n, NSGPU_ERR_ARG, "table %u out of range", j);
    const uint32_t N = c->reads.n;
    std::vector<uint64_t> k(N);
    if (N) {
        NS_HIP(hipMemcpyAsync(k.data(), c->idx_keys.as<uint64_t>() + (size_t)j * N, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
        NS_HIP(hipMemcpyAsync(ids_out, c->idx_ids.as<uint32_t>() + (size_t)j * N, (size_t)N * 4, hipMemcpyDeviceToHost, c->stream));
    }
    NS_HIP(stream_wait(c->stream));
    uint32_t u = 0;
    for (uint32_t i = 0; i < N; ++i)
        if (i == 0 || k[i] != k[i - 1]) { keys_out[u] = k[i]; start_out[u] = i; ++u; }
    start_out[u] = N;
    *nkeys_out = u;
    return NSGPU_OK;
}

int nsgpu_filter_strings_impl(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq)
{
    NS_CHECK(c->have_index && c->have_salts, NSGPU_ERR_ARG, "filter: build the index first");
    NS_HIP(hipSetDevice(c->prm.device));
    NS_HIP(hipEventRecord(c->t_stage.a, c->stream));
    NS_TRY(store_from_ascii(c, c->queries, strs, qoff, nq));
    NS_TRY(c->qsketch.reserve(((size_t)nq * c->prm.n + 1) * 8));
    NS_TRY(launch_sketch(c, c->queries, c->qsketch.as<uint64_t>(), nullptr));
    NS_TRY(run_filter(c, c->qsketch.as<uint64_t>(), nullptr, nq, false));
    NS_HIP(hipEventRecord(c->t_stage.b, c->stream));
    NS_HIP(stream_wait(c->stream));
    NS_HIP(hipEventElapsedTime(&c->timing.filter_ms, c->t_stage.a, c->t_stage.b));
    c->have_filter_all = false;
    return NSGPU_OK;
}

int nsgpu_filter_batch(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq, uint64_t **out_off, uint32_t **out_ids)
{
    NS_CHECK(c && qoff && out_off && out_ids, NSGPU_ERR_ARG, "nsgpu_filter_batch: null argument");
    if (nq > 0 && nq <= 4096 && c->have_index && c->have_salts) {
        // a batch of window queries: one kernel from the strings to the candidate lists (kernels_minhash.hip window_query_kernel); what it
        // declines -- a query with more than 2048 table matches -- goes through the multi-pass kernels below
        NS_HIP(hipSetDevice(c->prm.device));
        const uint64_t *fo = nullptr;
        const uint32_t *fi = nullptr;
        bool redo = false;
        NS_TRY(run_window_queries_fast(c, strs, qoff, nq, fo, fi, &redo));
        if (!redo) {
            uint64_t *off = (uint64_t *)malloc(((size_t)nq + 1) * 8);
            uint32_t *ids = (uint32_t *)malloc((fo[nq] + 1) * 4);
            NS_CHECK(off && ids, NSGPU_ERR_NOMEM, "malloc failed");
            memcpy(off, fo, ((size_t)nq + 1) * 8);
            if (fo[nq]) memcpy(ids, fi, fo[nq] * 4);
            *out_off = off, *out_ids = ids;
            return NSGPU_OK;
        }
    }
    NS_TRY(nsgpu_filter_strings_impl(c, strs, qoff, nq));
    uint64_t *off = (uint64_t *)malloc(((size_t)nq + 1) * 8);
    uint32_t *ids = (uint32_t *)malloc((c->f_total + 1) * 4);
    NS_CHECK(off && ids, NSGPU_ERR_NOMEM, "malloc failed");
    NS_HIP(hipMemcpyAsync(off, c->f_off.p, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->f_total) NS_HIP(hipMemcpyAsync(ids, c->f_ids.p, c->f_total * 4, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait(c->stream));
    *out_off = off;
    *out_ids = ids;
    return NSGPU_OK;
}

int nsgpu_filter(nsgpu_ctx *c, const char *s, size_t len, uint32_t **ids, size_t *n_ids)
{
    NS_CHECK(c && ids && n_ids && (s || len == 0), NSGPU_ERR_ARG, "nsgpu_filter: null argument");
    uint64_t qoff[2] = {0, (uint64_t)len};
    uint64_t *off = nullptr;
    NS_TRY(nsgpu_filter_batch(c, s, qoff, 1, &off, ids));
    *n_ids = (size_t)off[1];
    free(off);
    return NSGPU_OK;
}

int nsgpu_filter_all_reads(nsgpu_ctx *c, uint64_t *n_candidates_out)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "null ctx");
    NS_CHECK(c->have_index && c->have_sketch, NSGPU_ERR_ARG, "nsgpu_filter_all_reads: build the index first");
    NS_HIP(hipSetDevice(c->prm.device));
    const uint32_t N = c->reads.n, n = c->prm.n;
    NS_CHECK((uint64_t)N * 2 < (1ull << 32), NSGPU_ERR_RANGE, "too many reads for 2N queries");
    NS_TRY(c->sketch_rc.reserve(((size_t)N * n + 1) * 8));
    NS_HIP(hipEventRecord(c->t_stage.a, c->stream));
    NS_TRY(launch_sketch(c, c->reads, nullptr, c->sketch_rc.as<uint64_t>()));
    NS_TRY(run_filter(c, c->sketch.as<uint64_t>(), c->sketch_rc.as<uint64_t>(), 2 * N, true));
    NS_HIP(hipEventRecord(c->t_stage.b, c->stream));
    NS_HIP(stream_wait(c->stream));
    NS_HIP(hipEventElapsedTime(&c->timing.filter_ms, c->t_stage.a, c->t_stage.b));
    c->have_filter_all = true;
    if (n_candidates_out) *n_candidates_out = c->f_total;
    return NSGPU_OK;
}

int nsgpu_filter_all_fetch(nsgpu_ctx *c, uint64_t *off_out, uint32_t *ids_out)
{
    NS_CHECK(c && off_out, NSGPU_ERR_ARG, "null argument");
    NS_CHECK(c->have_filter_all, NSGPU_ERR_ARG, "nsgpu_filter_all_fetch: call nsgpu_filter_all_reads first");
    NS_HIP(hipMemcpyAsync(off_out, c->f_off.p, ((size_t)c->f_nq + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->f_total && ids_out) NS_HIP(hipMemcpyAsync(ids_out, c->f_ids.p, c->f_total * 4, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait(c->stream));
    return NSGPU_OK;
}

int nsgpu_check_repetitive(nsgpu_ctx *c, uint8_t *flags_out)
{
    NS_CHECK(c && flags_out, NSGPU_ERR_ARG, "null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    const uint32_t N = c->reads.n;
    NS_TRY(c->rep_flags.reserve((size_t)N + 16));
    NS_HIP(hipEventRecord(c->t_stage.a, c->stream));
    NS_TRY(launch_repetitive(c, c->reads, c->rep_flags.as<uint8_t>()));
    NS_HIP(hipEventRecord(c->t_stage.b, c->stream));
    if (N) NS_HIP(hipMemcpyAsync(flags_out, c->rep_flags.p, N, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait(c->stream));
    NS_HIP(hipEventElapsedTime(&c->timing.repetitive_ms, c->t_stage.a, c->t_stage.b));
    return NSGPU_OK;
}

int nsgpu_get_timing(const nsgpu_ctx *c, nsgpu_timing *t)
{
    NS_CHECK(c && t, NSGPU_ERR_ARG, "null argument");
    *t = c->timing;
    return NSGPU_OK;
}

}  // extern "C"

namespace nsgpu {
int filter_strings_device(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq) { return nsgpu_filter_strings_impl(c, strs, qoff, nq); }
}

// how many times this process's host threads have waited for GPU work through the library (stream and event waits): with the slot count of a
// contig stage it says how many hand-overs a slot costs (bench.py: config.host_waits_per_slot)
extern "C" uint64_t nsgpu_host_wait_count(void) { return g_host_waits.load(std::memory_order_relaxed); }
