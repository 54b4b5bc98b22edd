// dist.hip -- SURVEY 8(e): the hot path over the GPUs of one node, one process per GPU, in C++ behind the C-ABI (so that a
// C++11 host such as Compressor::compress() can drive N GPUs without Python).
//
//   reads      shard by id (rank r loads ids [lo_r, hi_r)); nsgpu_dist_load_reads replicates the shards (all-gather) so that
//              every GPU holds all 2-bit reads -- alignment candidates come from anywhere (12.5 GB at 50 Gbases: fits 288 GB)
//   sketch     every rank sketches ITS id range only (a1-a6 need no collective)
//   bucket tables, two selectable ways (nsgpu_dist_sketch_index):
//     NSGPU_DIST_REPLICATE   all-gather of the sketch rows (N * n * 8 B), every rank sorts all n tables
//     NSGPU_DIST_ALLTOALL    the north-star partitioning: table j is OWNED by rank j % world; an all-to-all(v) of the
//                            (slot, key, id) tuples -- rank p sends column j of its rows to the owner of j (ids are implicit:
//                            rows travel in id order) -- the owner sorts its n / world tables over all N reads, and the sorted
//                            tables are all-gathered so that window queries stay local (no per-query traffic; the tables are
//                            72 MB per 100 k reads).  Same index, bit for bit, as one process builds.
//   contig stage (nsgpu_dist_consensus_run, consensus_driver.hip): lock-step slots, ONE small all-gather of the claim /
//              seed request lists per slot, resolved in global builder order on a replicated claim table.
//
// Transport: RCCL over xGMI, bound at run time (dlopen of librccl.so.1, the library torch ships or /opt/rocm's) so that
// libnsgpu itself links nothing but the HIP runtime; or host callbacks (tests drive the same C++ code over gloo / TCP,
// a C++ host may plug MPI).  All collectives of a communicator are issued from ONE thread at a time.
#include "common.hpp"
#include "dist.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <rocprim/rocprim.hpp>

namespace nsgpu {

namespace {

// ---- RCCL through dlopen -------------------------------------------------------------------------------------------
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

RcclApi &rccl()
{
    static RcclApi A = [] {
        RcclApi a;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) { a.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (a.h) break; }
        if (!a.h) return a;
#define NS_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.h, name))
        NS_SYM(GetUniqueId, "ncclGetUniqueId"); NS_SYM(CommInitRank, "ncclCommInitRank"); NS_SYM(CommDestroy, "ncclCommDestroy");
        NS_SYM(GetErrorString, "ncclGetErrorString"); NS_SYM(AllGather, "ncclAllGather"); NS_SYM(Send, "ncclSend"); NS_SYM(Recv, "ncclRecv");
        NS_SYM(GroupStart, "ncclGroupStart"); NS_SYM(GroupEnd, "ncclGroupEnd");
#undef NS_SYM
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GetErrorString && a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
        return a;
    }();
    return A;
}

#define NS_NCCL(expr)                                                                                                   \
    do {                                                                                                                \
        ncclResult_t r_ = (expr);                                                                                       \
        if (r_ != ncclSuccess) { set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); return NSGPU_ERR_HIP; } \
    } while (0)

struct RcclComm : Comm {
    ncclComm_t comm = nullptr;
    DevBuf d_send, d_recv;          // staging for host-buffer collectives
    ~RcclComm() override { if (comm) (void)rccl().CommDestroy(comm); }
    int all_gather(const void *send, void *recv, size_t bytes, bool device, hipStream_t st) override
    {
        if (bytes == 0) return NSGPU_OK;
        if (device) { NS_NCCL(rccl().AllGather(send, recv, bytes, ncclChar, comm, st)); return NSGPU_OK; }
        NS_TRY(d_send.reserve(bytes + 16));
        NS_TRY(d_recv.reserve(bytes * world + 16));
        NS_HIP(hipMemcpyAsync(d_send.p, send, bytes, hipMemcpyHostToDevice, st));
        NS_NCCL(rccl().AllGather(d_send.p, d_recv.p, bytes, ncclChar, comm, st));
        NS_HIP(hipMemcpyAsync(recv, d_recv.p, bytes * world, hipMemcpyDeviceToHost, st));
        NS_HIP(hipStreamSynchronize(st));
        return NSGPU_OK;
    }
    int all_to_all_v(const void *send, const size_t *sb, void *recv, const size_t *rb, bool device, hipStream_t st) override
    {
        NS_CHECK(device, NSGPU_ERR_ARG, "RCCL all-to-all: device buffers only");
        // xGMI is point to point: all peers at once (one grouped send/recv per peer) drives every link concurrently
        size_t so = 0, ro = 0;
        NS_NCCL(rccl().GroupStart());
        for (uint32_t p = 0; p < world; ++p) {
            if (sb[p]) NS_NCCL(rccl().Send(static_cast<const char *>(send) + so, sb[p], ncclChar, (int)p, comm, st));
            if (rb[p]) NS_NCCL(rccl().Recv(static_cast<char *>(recv) + ro, rb[p], ncclChar, (int)p, comm, st));
            so += sb[p], ro += rb[p];
        }
        NS_NCCL(rccl().GroupEnd());
        return NSGPU_OK;
    }
};

// ---- host callbacks (gloo / MPI / TCP of the caller); device buffers are staged through host memory ----------------
struct CallbackComm : Comm {
    nsgpu_comm_callbacks cb;
    std::vector<uint8_t> hs, hr;
    int all_gather(const void *send, void *recv, size_t bytes, bool device, hipStream_t st) override
    {
        if (bytes == 0) return NSGPU_OK;
        const void *s = send;
        void *r = recv;
        if (device) {
            hs.resize(bytes), hr.resize(bytes * world);
            NS_HIP(hipMemcpyAsync(hs.data(), send, bytes, hipMemcpyDeviceToHost, st));
            NS_HIP(hipStreamSynchronize(st));
            s = hs.data(), r = hr.data();
        }
        NS_CHECK(cb.all_gather(cb.user, s, r, (uint64_t)bytes) == 0, NSGPU_ERR_HIP, "communicator callback all_gather failed");
        if (device) { NS_HIP(hipMemcpyAsync(recv, hr.data(), bytes * world, hipMemcpyHostToDevice, st)); NS_HIP(hipStreamSynchronize(st)); }
        return NSGPU_OK;
    }
    int all_to_all_v(const void *send, const size_t *sb, void *recv, const size_t *rb, bool device, hipStream_t st) override
    {
        size_t st_bytes = 0, rt_bytes = 0;
        std::vector<uint64_t> s64(world), r64(world);
        for (uint32_t p = 0; p < world; ++p) st_bytes += sb[p], rt_bytes += rb[p], s64[p] = sb[p], r64[p] = rb[p];
        const void *s = send;
        void *r = recv;
        if (device) {
            hs.resize(st_bytes + 1), hr.resize(rt_bytes + 1);
            if (st_bytes) NS_HIP(hipMemcpyAsync(hs.data(), send, st_bytes, hipMemcpyDeviceToHost, st));
            NS_HIP(hipStreamSynchronize(st));
            s = hs.data(), r = hr.data();
        }
        NS_CHECK(cb.all_to_all(cb.user, s, s64.data(), r, r64.data()) == 0, NSGPU_ERR_HIP, "communicator callback all_to_all failed");
        if (device && rt_bytes) { NS_HIP(hipMemcpyAsync(recv, hr.data(), rt_bytes, hipMemcpyHostToDevice, st)); NS_HIP(hipStreamSynchronize(st)); }
        return NSGPU_OK;
    }
};

// ---- kernels of the all-to-all bucket build ---------------------------------------------------------------------------
// send block for owner q: [slots of q, ascending][my rows] keys
__global__ __launch_bounds__(256) void dist_pack_columns_kernel(const uint64_t *__restrict__ sketch, uint32_t lo, uint32_t rows, uint32_t n, uint32_t world,
                                                                const uint64_t *__restrict__ blk_off /* [world] in keys */, uint64_t *__restrict__ out)
{
    // one thread per (row, slot): coalesced reads of the row-major sketch; the writes of one slot are contiguous over rows
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rows * n) return;
    const uint32_t r = (uint32_t)(i / n), j = (uint32_t)(i % n);
    const uint32_t q = j % world, s = j / world;
    out[blk_off[q] + (uint64_t)s * rows + r] = sketch[((uint64_t)lo + r) * n + j];
}

// received: from source p a block [my slots][rows_p]; column s of the full matrix = the blocks' rows in id order
__global__ __launch_bounds__(256) void dist_assemble_kernel(const uint64_t *__restrict__ recv, const uint64_t *__restrict__ src_off /* [world] in keys */,
                                                            const uint32_t *__restrict__ src_lo /* [world + 1] */, uint32_t world, uint32_t n_own, uint32_t N,
                                                            uint64_t *__restrict__ keys, uint64_t *__restrict__ ents)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_own * N) return;
    const uint32_t s = (uint32_t)(i / N), id = (uint32_t)(i % N);
    uint32_t p = 0;
    while (p + 1 < world && id >= src_lo[p + 1]) ++p;
    const uint32_t rows_p = src_lo[p + 1] - src_lo[p];
    keys[i] = recv[src_off[p] + (uint64_t)s * rows_p + (id - src_lo[p])];
    ents[i] = (uint64_t)s << 32 | id;
}

__global__ __launch_bounds__(256) void dist_ids_kernel(const uint64_t *__restrict__ ents, uint64_t total, uint32_t *__restrict__ ids)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) ids[i] = (uint32_t)ents[i];
}

// gathered [world][max_own][N] -> idx[j * N ..] with j = q + s * world
__global__ __launch_bounds__(256) void dist_place_kernel(const uint64_t *__restrict__ gk, const uint32_t *__restrict__ gi, uint32_t world, uint32_t max_own, uint32_t n,
                                                         uint32_t N, uint64_t *__restrict__ idx_keys, uint32_t *__restrict__ idx_ids)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n * N) return;
    const uint32_t j = (uint32_t)(i / N), t = (uint32_t)(i % N);
    const uint32_t q = j % world, s = j / world;
    const uint64_t src = ((uint64_t)q * max_own + s) * N + t;
    idx_keys[i] = gk[src];
    idx_ids[i] = gi[src];
}

}  // namespace

}  // namespace nsgpu

using namespace nsgpu;

struct nsgpu_comm {
    std::unique_ptr<Comm> impl;
    nsgpu_ctx *ctx = nullptr;
    // the rank's id range in the replicated read set and everybody's (set by nsgpu_dist_load_reads)
    std::vector<uint32_t> lo;      // [world + 1]
    DevBuf d_a, d_b, d_c, d_d, d_e, d_f, d_meta;
    uint64_t bytes_all_gather = 0, bytes_all_to_all = 0;      // received by this rank, all calls so far
    ~nsgpu_comm() { for (DevBuf *b : {&d_a, &d_b, &d_c, &d_d, &d_e, &d_f, &d_meta}) b->release(); }
};

Comm *nsgpu_comm_impl(nsgpu_comm *c) { return c ? c->impl.get() : nullptr; }
void nsgpu_comm_count(nsgpu_comm *c, uint64_t all_gather_bytes, uint64_t all_to_all_bytes) { if (c) c->bytes_all_gather += all_gather_bytes, c->bytes_all_to_all += all_to_all_bytes; }
namespace nsgpu {
int store_prepare_lens(nsgpu_ctx *c, SeqStore &st, const uint32_t *len, uint32_t n);       // api.hip
int store_from_ascii_shard(nsgpu_ctx *c, SeqStore &st, const char *bases, const uint64_t *off, uint32_t n);
}

extern "C" {

int nsgpu_comm_unique_id(uint8_t *id_out)
{
    NS_CHECK(id_out, NSGPU_ERR_ARG, "nsgpu_comm_unique_id: null argument");
    NS_CHECK(rccl().ok, NSGPU_ERR_HIP, "RCCL (librccl.so.1) could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
    ncclUniqueId id;
    NS_NCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == NSGPU_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return NSGPU_OK;
}

int nsgpu_comm_init_rccl(nsgpu_ctx *c, const uint8_t *id, uint32_t rank, uint32_t world, nsgpu_comm **out)
{
    NS_CHECK(c && id && out && world >= 1 && rank < world, NSGPU_ERR_ARG, "nsgpu_comm_init_rccl: bad argument");
    NS_CHECK(rccl().ok, NSGPU_ERR_HIP, "RCCL (librccl.so.1) could not be loaded");
    NS_HIP(hipSetDevice(c->prm.device));
    std::unique_ptr<RcclComm> rc(new RcclComm);
    rc->rank = rank, rc->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    NS_NCCL(rccl().CommInitRank(&rc->comm, (int)world, uid, (int)rank));
    nsgpu_comm *nc = new nsgpu_comm;
    nc->impl = std::move(rc);
    nc->ctx = c;
    *out = nc;
    return NSGPU_OK;
}

int nsgpu_comm_init_callbacks(nsgpu_ctx *c, const nsgpu_comm_callbacks *cb, uint32_t rank, uint32_t world, nsgpu_comm **out)
{
    NS_CHECK(c && cb && cb->all_gather && cb->all_to_all && out && world >= 1 && rank < world, NSGPU_ERR_ARG, "nsgpu_comm_init_callbacks: bad argument");
    std::unique_ptr<CallbackComm> cc(new CallbackComm);
    cc->rank = rank, cc->world = world, cc->cb = *cb;
    nsgpu_comm *nc = new nsgpu_comm;
    nc->impl = std::move(cc);
    nc->ctx = c;
    *out = nc;
    return NSGPU_OK;
}

void nsgpu_comm_destroy(nsgpu_comm *c) { delete c; }

// Shards in rank order (rank r passes reads [lo_r, hi_r) of one read set) -> every rank holds all reads: 2-bit rows in HBM and, as the
// contig engine's host copy, the same rows (0.25 B/base per rank; the ASCII text of the other ranks' reads never exists anywhere).
// Each rank packs its own shard on its GPU; the packed rows are all-gathered device to device in pieces of at most 256 MiB per rank
// (the row layout is a function of the lengths alone, so a shard's rows are one contiguous byte range of the replicated store).
int nsgpu_dist_load_reads(nsgpu_ctx *c, nsgpu_comm *cm, const char *bases, const uint64_t *off, uint32_t n_local, uint32_t *lo_out, uint32_t *hi_out)
{
    NS_CHECK(c && cm && off && cm->ctx == c, NSGPU_ERR_ARG, "nsgpu_dist_load_reads: bad argument");
    NS_HIP(hipSetDevice(c->prm.device));
    Comm &C = *cm->impl;
    const uint32_t W = C.world;
    const hipStream_t st = c->stream;
    c->have_sketch = c->have_index = c->have_filter_all = c->have_cons = false;
    // the own shard: ASCII -> 2-bit rows on this GPU (a store of its own)
    SeqStore mine_st;
    NS_TRY(store_from_ascii_shard(c, mine_st, bases, off, n_local));
    c->ascii.release();                                              // the shard's text does not stay in HBM
    std::vector<uint64_t> cnt(2 * (size_t)W);
    const uint64_t mine[2] = {n_local, mine_st.packed_bytes};
    NS_TRY(C.all_gather(mine, cnt.data(), sizeof(mine), false, st));
    cm->bytes_all_gather += sizeof(mine) * W;
    uint64_t max_n = 1, max_b = 1, tot_n = 0;
    for (uint32_t p = 0; p < W; ++p) { max_n = std::max(max_n, cnt[2 * p]); max_b = std::max(max_b, cnt[2 * p + 1]); tot_n += cnt[2 * p]; }
    NS_CHECK(tot_n < (1ull << 32), NSGPU_ERR_RANGE, "Too many reads for read_t (src/ReadData.cpp:194-196)");
    // lengths (host) -> the layout of the replicated store
    std::vector<uint32_t> len_mine(max_n, 0), len_all(max_n * W), lens(tot_n);
    for (uint32_t r = 0; r < n_local; ++r) len_mine[r] = (uint32_t)(off[r + 1] - off[r]);
    NS_TRY(C.all_gather(len_mine.data(), len_all.data(), max_n * 4, false, st));
    cm->bytes_all_gather += max_n * 4 * W;
    cm->lo.assign(W + 1, 0);
    uint64_t k = 0;
    for (uint32_t p = 0; p < W; ++p) {
        cm->lo[p] = (uint32_t)k;
        for (uint64_t r = 0; r < cnt[2 * p]; ++r) lens[k++] = len_all[(uint64_t)p * max_n + r];
    }
    cm->lo[W] = (uint32_t)k;
    SeqStore &S = c->reads;
    NS_TRY(store_prepare_lens(c, S, lens.data(), (uint32_t)tot_n));
    for (uint32_t p = 0; p < W; ++p)
        NS_CHECK(S.h_poff[cm->lo[p + 1]] - S.h_poff[cm->lo[p]] == cnt[2 * p + 1], NSGPU_ERR_ARG, "nsgpu_dist_load_reads: the shards' rows do not add up");
    // packed rows, device to device, bounded staging
    const uint64_t piece = std::min<uint64_t>(256ull << 20, max_b);
    NS_TRY(cm->d_a.reserve(piece + 16));
    NS_TRY(cm->d_b.reserve(piece * W + 16));
    for (uint64_t o = 0; o < max_b; o += piece) {
        const uint64_t m = std::min(piece, max_b - o);
        const uint64_t have = o < mine_st.packed_bytes ? std::min(m, mine_st.packed_bytes - o) : 0;
        if (have) NS_HIP(hipMemcpyAsync(cm->d_a.p, mine_st.packed.as<uint8_t>() + o, have, hipMemcpyDeviceToDevice, st));
        NS_TRY(C.all_gather(cm->d_a.p, cm->d_b.p, m, true, st));
        cm->bytes_all_gather += m * W;
        for (uint32_t p = 0; p < W; ++p) {
            const uint64_t pb = cnt[2 * p + 1];
            if (o < pb) NS_HIP(hipMemcpyAsync(S.packed.as<uint8_t>() + S.h_poff[cm->lo[p]] + o, cm->d_b.as<uint8_t>() + (uint64_t)p * m, std::min(m, pb - o), hipMemcpyDeviceToDevice, st));
        }
        NS_HIP(stream_wait(st));
    }
    mine_st.release();
    cm->d_a.release(); cm->d_b.release();                            // up to W x 256 MiB: not kept for the small exchanges that follow
    // host side: offsets in bases, and the packed rows as the engine's copy of the reads (never the ASCII text of all reads)
    c->h_off.assign((size_t)tot_n + 1, 0);
    for (uint64_t r = 0; r < tot_n; ++r) c->h_off[r + 1] = c->h_off[r] + lens[r];
    c->h_bases.clear(), c->h_bases.shrink_to_fit();
    NS_TRY(mirror_finalize(c, true));
    if (lo_out) *lo_out = cm->lo[C.rank];
    if (hi_out) *hi_out = cm->lo[C.rank + 1];
    return NSGPU_OK;
}

// bytes this rank received in all-gathers / all-to-alls so far (load + every sketch_index + every consensus run), for bench.py's record
int nsgpu_comm_stats(const nsgpu_comm *cm, uint64_t *bytes_all_gather, uint64_t *bytes_all_to_all, uint64_t *host_bytes_reads)
{
    NS_CHECK(cm, NSGPU_ERR_ARG, "nsgpu_comm_stats: null communicator");
    if (bytes_all_gather) *bytes_all_gather = cm->bytes_all_gather;
    if (bytes_all_to_all) *bytes_all_to_all = cm->bytes_all_to_all;
    // what the contig engine's host copy of ALL reads occupies on this rank: packed rows (or the ASCII text) + the offset tables
    if (host_bytes_reads) {
        const nsgpu_ctx *c = cm->ctx;
        *host_bytes_reads = c->h_packed.capacity() + c->h_bases.capacity() + c->h_off.capacity() * 8 + c->reads.h_poff.capacity() * 8 + c->reads.h_len.capacity() * 4;
    }
    return NSGPU_OK;
}

// Sketch the own id range, exchange, build the n bucket tables (mode: NSGPU_DIST_REPLICATE / NSGPU_DIST_ALLTOALL).
int nsgpu_dist_sketch_index(nsgpu_ctx *c, nsgpu_comm *cm, const uint64_t *salts, int mode)
{
    NS_CHECK(c && cm && salts && cm->ctx == c && cm->lo.size() == (size_t)cm->impl->world + 1, NSGPU_ERR_ARG, "nsgpu_dist_sketch_index: load the reads with nsgpu_dist_load_reads first");
    NS_CHECK(mode == NSGPU_DIST_REPLICATE || mode == NSGPU_DIST_ALLTOALL, NSGPU_ERR_ARG, "nsgpu_dist_sketch_index: unknown mode");
    NS_HIP(hipSetDevice(c->prm.device));
    Comm &C = *cm->impl;
    const uint32_t W = C.world, me = C.rank, N = c->reads.n, n = c->prm.n;
    NS_CHECK(N == cm->lo[W], NSGPU_ERR_ARG, "nsgpu_dist_sketch_index: the context's reads are not the replicated set");
    const uint32_t lo = cm->lo[me], hi = cm->lo[me + 1], rows = hi - lo;
    NS_TRY(nsgpu_sketch_range(c, salts, lo, hi));
    const hipStream_t st = c->stream;
    uint32_t max_rows = 1;
    for (uint32_t p = 0; p < W; ++p) max_rows = std::max(max_rows, cm->lo[p + 1] - cm->lo[p]);
    if (mode == NSGPU_DIST_REPLICATE || W == 1) {
        if (W > 1) {
            const size_t blk = (size_t)max_rows * n * 8;
            NS_TRY(cm->d_a.reserve(blk + 16));
            NS_TRY(cm->d_b.reserve(blk * W + 16));
            if (rows) NS_HIP(hipMemcpyAsync(cm->d_a.p, c->sketch.as<uint8_t>() + (size_t)lo * n * 8, (size_t)rows * n * 8, hipMemcpyDeviceToDevice, st));
            NS_TRY(C.all_gather(cm->d_a.p, cm->d_b.p, blk, true, st));
            cm->bytes_all_gather += blk * W;
            for (uint32_t p = 0; p < W; ++p) {
                const uint32_t rp = cm->lo[p + 1] - cm->lo[p];
                if (p != me && rp)
                    NS_HIP(hipMemcpyAsync(c->sketch.as<uint8_t>() + (size_t)cm->lo[p] * n * 8, cm->d_b.as<uint8_t>() + (size_t)p * blk, (size_t)rp * n * 8, hipMemcpyDeviceToDevice, st));
            }
            NS_HIP(stream_wait(st));
        }
        NS_TRY(nsgpu_sketch_mark_complete(c));
        return nsgpu_build_index(c);
    }
    // ---- all-to-all: column j goes to rank j % W ----
    NS_CHECK((uint64_t)N * n < (1ull << 32), NSGPU_ERR_RANGE, "index: n * N exceeds 2^32 entries per device; shard the reads");
    auto n_own = [&](uint32_t q) { return (n > q ? (n - q + W - 1) / W : 0u); };
    const uint32_t own = n_own(me), max_own = n_own(0);
    std::vector<size_t> sb(W), rb(W);
    std::vector<uint64_t> blk_off(W), src_off(W);
    uint64_t so = 0, ro = 0;
    for (uint32_t p = 0; p < W; ++p) {
        blk_off[p] = so, src_off[p] = ro;
        sb[p] = (size_t)n_own(p) * rows * 8;
        rb[p] = (size_t)own * (cm->lo[p + 1] - cm->lo[p]) * 8;
        so += (uint64_t)n_own(p) * rows, ro += (uint64_t)own * (cm->lo[p + 1] - cm->lo[p]);
    }
    NS_TRY(cm->d_a.reserve(so * 8 + 16));
    NS_TRY(cm->d_b.reserve(ro * 8 + 16));
    NS_TRY(cm->d_meta.reserve((size_t)W * 24 + 64));
    uint64_t *d_blk = cm->d_meta.as<uint64_t>(), *d_src = d_blk + W;
    uint32_t *d_lo = reinterpret_cast<uint32_t *>(d_src + W);
    NS_HIP(hipMemcpyAsync(d_blk, blk_off.data(), (size_t)W * 8, hipMemcpyHostToDevice, st));
    NS_HIP(hipMemcpyAsync(d_src, src_off.data(), (size_t)W * 8, hipMemcpyHostToDevice, st));
    NS_HIP(hipMemcpyAsync(d_lo, cm->lo.data(), ((size_t)W + 1) * 4, hipMemcpyHostToDevice, st));
    if (rows) {
        const uint64_t tot = (uint64_t)rows * n;
        hipLaunchKernelGGL(dist_pack_columns_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, st, c->sketch.as<uint64_t>(), lo, rows, n, W, d_blk, cm->d_a.as<uint64_t>());
        NS_HIP(hipGetLastError());
    }
    NS_HIP(stream_wait(st));                       // blk_off etc. are stack/host vectors: uploaded before they go out of scope
    NS_TRY(C.all_to_all_v(cm->d_a.p, sb.data(), cm->d_b.p, rb.data(), true, st));
    cm->bytes_all_to_all += ro * 8;
    // the owner's tables: sort (key) then (slot), as build_index does for all n
    const uint64_t total = (uint64_t)own * N;
    const size_t gblk = (size_t)max_own * N;
    NS_TRY(cm->d_c.reserve((total + 1) * 8)); NS_TRY(cm->d_d.reserve((total + 1) * 8)); NS_TRY(cm->d_e.reserve((total + 1) * 8)); NS_TRY(cm->d_f.reserve((total + 1) * 8));
    NS_TRY(c->idx_tmp_k.reserve(gblk * 8 + 16));   // this rank's sorted keys, padded to max_own tables
    NS_TRY(c->idx_tmp_v.reserve(gblk * 4 + 16));   //                    ids
    if (total) {
        uint64_t *k0 = cm->d_c.as<uint64_t>(), *e0 = cm->d_d.as<uint64_t>(), *k1 = cm->d_e.as<uint64_t>(), *e1 = cm->d_f.as<uint64_t>();
        hipLaunchKernelGGL(dist_assemble_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, cm->d_b.as<uint64_t>(), d_src, d_lo, W, own, N, k0, e0);
        NS_HIP(hipGetLastError());
        size_t ws_a = 0, ws_b = 0;
        NS_HIP(rocprim::radix_sort_pairs(nullptr, ws_a, k0, k1, e0, e1, (size_t)total, 0u, 64u, st));
        NS_HIP(rocprim::radix_sort_pairs(nullptr, ws_b, e1, e0, k1, c->idx_tmp_k.as<uint64_t>(), (size_t)total, 32u, 40u, st));
        NS_TRY(c->idx_sort_ws.reserve(std::max(ws_a, ws_b) + 16));
        NS_HIP(rocprim::radix_sort_pairs(c->idx_sort_ws.p, ws_a, k0, k1, e0, e1, (size_t)total, 0u, 64u, st));
        NS_HIP(rocprim::radix_sort_pairs(c->idx_sort_ws.p, ws_b, e1, e0, k1, c->idx_tmp_k.as<uint64_t>(), (size_t)total, 32u, 40u, st));
        hipLaunchKernelGGL(dist_ids_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, e0, total, c->idx_tmp_v.as<uint32_t>());
        NS_HIP(hipGetLastError());
    }
    // everybody gets every table: queries stay local
    NS_TRY(c->idx_tmp_e.reserve(gblk * 8 * W + 16));
    NS_TRY(c->idx_tmp_e2.reserve(gblk * 4 * W + 16));
    NS_TRY(C.all_gather(c->idx_tmp_k.p, c->idx_tmp_e.p, gblk * 8, true, st));
    NS_TRY(C.all_gather(c->idx_tmp_v.p, c->idx_tmp_e2.p, gblk * 4, true, st));
    cm->bytes_all_gather += (uint64_t)gblk * 12 * W;
    const uint64_t all = (uint64_t)N * n;
    NS_TRY(c->idx_keys.reserve((all + 1) * 8));
    NS_TRY(c->idx_ids.reserve((all + 1) * 4));
    if (all) {
        hipLaunchKernelGGL(dist_place_kernel, dim3((uint32_t)((all + 255) / 256)), dim3(256), 0, st, c->idx_tmp_e.as<uint64_t>(), c->idx_tmp_e2.as<uint32_t>(), W, max_own, n, N,
                           c->idx_keys.as<uint64_t>(), c->idx_ids.as<uint32_t>());
        NS_HIP(hipGetLastError());
    }
    NS_HIP(stream_wait(st));
    c->have_index = true;
    c->have_filter_all = false;
    return NSGPU_OK;
}

}  // extern "C"
