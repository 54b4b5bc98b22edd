// bwt.hip -- SURVEY 8 row f4, the part with a unique answer: the Burrows-Wheeler transform of one block of a stream file.
//
// Compressor::compress() (src/Compressor.cpp:111-143) hands every stream file to bsc::BSC_compress (src/bsc.cpp:1045-1057), which cuts it
// into blocks of 48 MB and, per block, runs libbsc's block sorter -- bsc_bwt_encode (libbsc/bwt/bwt.cpp:46-79): the BWT of the block with
// its primary index and the auxiliary indexes the decoder's parallel inverse uses -- before the QLFC entropy coder.  The sorter is 4.6 of
// the 13.6 CPU-seconds the back end takes per cfg2 step (profiles/r02_backend_coders.json), and unlike the coders behind it its output is
// a function of the input alone: B = T[n-1] followed by T[SA[k] - 1] for the suffixes in ascending order, the row of suffix 0 left out
// (the end of the block sorts below every byte), primary index = rank of suffix 0 + 1, auxiliary index j = rank of suffix j * r + 1.
//
// Here: a suffix array by prefix doubling, every round one rocPRIM radix sort over all suffixes (keys = the ranks of the two halves),
// O(log(longest repeat)) rounds of ~10 ms for 48 MB.  HBM-bound integer work; nothing of the coders (QLFC / LZMA2) is restated
// (DESIGN.md section 8).
#include "common.hpp"
#include <rocprim/rocprim.hpp>

namespace nsgpu {
namespace {

// round 0: the first 7 bytes of every suffix, zero-padded, then how many of them exist: equal padded bytes with different lengths means
// the shorter suffix is a prefix of the longer one, and the end of the block sorts first
__global__ __launch_bounds__(256) void bwt_first_keys_kernel(const uint8_t *__restrict__ T, uint32_t n, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t have = n - i < 7u ? n - i : 7u;
    uint64_t k = 0;
#pragma unroll
    for (uint32_t b = 0; b < 7; ++b) k = (k << 8) | (b < have ? T[i + b] : 0u);
    keys[i] = (k << 8) | have;
    vals[i] = i;
}

// position k of the sorted order starts a new group when its key differs from its predecessor's: heads carry k + 1, the others 0 -- an
// inclusive max scan then gives every suffix the 1-based position of its group's head as its rank
__global__ __launch_bounds__(256) void bwt_heads_kernel(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ head)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    head[k] = (k == 0 || keys[k] != keys[k - 1]) ? k + 1 : 0u;
}
__global__ __launch_bounds__(256) void bwt_scatter_ranks_kernel(const uint32_t *__restrict__ vals, const uint32_t *__restrict__ head_scanned, uint32_t n, uint32_t *__restrict__ rank)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    rank[vals[k]] = head_scanned[k];
}
// keys of the next round: (rank of the first h characters, rank of the h after them; 0 = past the end, below every rank)
__global__ __launch_bounds__(256) void bwt_pair_keys_kernel(const uint32_t *__restrict__ rank, uint32_t n, uint32_t h, uint32_t bits, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t r2 = (uint64_t)i + h < n ? rank[i + h] : 0u;
    keys[i] = ((uint64_t)rank[i] << bits) | r2;
    vals[i] = i;
}
struct IsHead { __device__ uint32_t operator()(uint32_t x) const { return x ? 1u : 0u; } };

// B[0] = T[n-1]; the sorted suffixes follow, the row of suffix 0 left out; aux[j] = rank of suffix j * r, 1-based (aux[0] = the primary index)
__global__ __launch_bounds__(256) void bwt_emit_kernel(const uint8_t *__restrict__ T, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ rank, uint32_t n,
                                                       uint8_t *__restrict__ out)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    if (k == 0) out[0] = T[n - 1];
    const uint32_t s = sa[k], r0 = rank[0] - 1;
    if (s == 0) return;
    out[k < r0 ? k + 1 : k] = T[s - 1];
}
__global__ __launch_bounds__(256) void bwt_aux_kernel(const uint32_t *__restrict__ rank, uint32_t n, uint32_t rate, uint32_t n_aux, int32_t *__restrict__ aux)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n_aux) return;
    aux[j] = (int32_t)rank[(uint64_t)j * rate];
}

}  // namespace
}  // namespace nsgpu

using namespace nsgpu;

extern "C" int nsgpu_bwt_block(nsgpu_ctx *c, const uint8_t *in, uint64_t n64, uint8_t *out, int32_t *primary_index, uint32_t aux_rate, int32_t *aux, uint32_t *n_aux_out,
                               double *gpu_ms_out, uint32_t *rounds_out)
{
    NS_CHECK(c != nullptr && primary_index != nullptr, NSGPU_ERR_ARG, "bwt: null argument");
    NS_CHECK(n64 == 0 || (in != nullptr && out != nullptr), NSGPU_ERR_ARG, "bwt: null buffer");
    NS_CHECK(n64 < (1ull << 30), NSGPU_ERR_RANGE, "bwt: a block of %llu bytes (the back end cuts stream files into 48 MB blocks, src/bsc.cpp:1045-1057)", (unsigned long long)n64);
    NS_CHECK(aux_rate == 0 || (aux_rate & (aux_rate - 1)) == 0, NSGPU_ERR_ARG, "bwt: the auxiliary index rate must be a power of two");
    NS_HIP(hipSetDevice(c->prm.device));
    const uint32_t n = (uint32_t)n64;
    const uint32_t n_aux = aux_rate && n ? (n - 1) / aux_rate + 1 : 0;
    if (n_aux_out) *n_aux_out = n_aux;
    if (gpu_ms_out) *gpu_ms_out = 0;
    if (rounds_out) *rounds_out = 0;
    *primary_index = 0;
    if (n == 0) return NSGPU_OK;
    NS_CHECK(n_aux == 0 || aux != nullptr, NSGPU_ERR_ARG, "bwt: auxiliary indexes asked for without a buffer");
    hipStream_t st = c->stream;
    DevBuf d_t, d_out, d_k0, d_k1, d_v0, d_v1, d_rank, d_head, d_ws, d_cnt, d_aux;
    struct Free { DevBuf *b[11]; ~Free() { for (DevBuf *x : b) x->release(); } } fr{{&d_t, &d_out, &d_k0, &d_k1, &d_v0, &d_v1, &d_rank, &d_head, &d_ws, &d_cnt, &d_aux}};
    NS_TRY(d_t.reserve(n + 16));
    NS_TRY(d_out.reserve(n + 16));
    NS_TRY(d_k0.reserve((size_t)n * 8)); NS_TRY(d_k1.reserve((size_t)n * 8));
    NS_TRY(d_v0.reserve((size_t)n * 4)); NS_TRY(d_v1.reserve((size_t)n * 4));
    NS_TRY(d_rank.reserve((size_t)n * 4)); NS_TRY(d_head.reserve((size_t)n * 4));
    NS_TRY(d_cnt.reserve(16));
    if (n_aux) NS_TRY(d_aux.reserve((size_t)n_aux * 4));
    uint64_t *k0 = d_k0.as<uint64_t>(), *k1 = d_k1.as<uint64_t>();
    uint32_t *v0 = d_v0.as<uint32_t>(), *v1 = d_v1.as<uint32_t>(), *rank = d_rank.as<uint32_t>(), *head = d_head.as<uint32_t>();
    uint32_t bits = 1;
    while ((1ull << bits) <= (uint64_t)n) ++bits;                 // ranks are 1 .. n
    size_t ws_sort = 0, ws_scan = 0, ws_red = 0;
    NS_HIP(rocprim::radix_sort_pairs(nullptr, ws_sort, k0, k1, v0, v1, (size_t)n, 0u, 64u, st));
    NS_HIP(rocprim::inclusive_scan(nullptr, ws_scan, head, head, (size_t)n, rocprim::maximum<uint32_t>(), st));
    NS_HIP(rocprim::reduce(nullptr, ws_red, rocprim::make_transform_iterator(head, IsHead()), d_cnt.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    NS_TRY(d_ws.reserve(std::max(ws_sort, std::max(ws_scan, ws_red)) + 16));
    NS_HIP(hipMemcpyAsync(d_t.p, in, n, hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    NS_HIP(hipEventCreate(&e0));
    NS_HIP(hipEventCreate(&e1));
    struct Ev { hipEvent_t a, b; ~Ev() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } } evs{e0, e1};
    NS_HIP(hipEventRecord(e0, st));
    const dim3 grid((n + 255) / 256), block(256);
    const uint8_t *T = d_t.as<uint8_t>();
    hipLaunchKernelGGL(bwt_first_keys_kernel, grid, block, 0, st, T, n, k0, v0);
    uint32_t rounds = 0;
    uint32_t h = 7;
    uint32_t end_bit = 64;
    for (;;) {
        ++rounds;
        size_t w = d_ws.cap;
        NS_HIP(rocprim::radix_sort_pairs(d_ws.p, w, k0, k1, v0, v1, (size_t)n, 0u, end_bit, st));
        hipLaunchKernelGGL(bwt_heads_kernel, grid, block, 0, st, k1, n, head);
        w = d_ws.cap;
        NS_HIP(rocprim::reduce(d_ws.p, w, rocprim::make_transform_iterator(head, IsHead()), d_cnt.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        w = d_ws.cap;
        NS_HIP(rocprim::inclusive_scan(d_ws.p, w, head, head, (size_t)n, rocprim::maximum<uint32_t>(), st));
        hipLaunchKernelGGL(bwt_scatter_ranks_kernel, grid, block, 0, st, v1, head, n, rank);
        NS_HIP(hipGetLastError());
        uint32_t groups = 0;
        NS_HIP(hipMemcpyAsync(&groups, d_cnt.p, 4, hipMemcpyDeviceToHost, st));
        NS_HIP(stream_wait(st));
        if (groups == n) break;
        NS_CHECK(h < n, NSGPU_ERR_ARG, "bwt: suffixes still tied after %u characters of a %u-byte block (internal error)", h, n);
        hipLaunchKernelGGL(bwt_pair_keys_kernel, grid, block, 0, st, rank, n, h, bits, k0, v0);
        end_bit = 2 * bits;
        h = h > (1u << 30) ? h : h * 2;
    }
    hipLaunchKernelGGL(bwt_emit_kernel, grid, block, 0, st, T, v1, rank, n, d_out.as<uint8_t>());
    if (n_aux) hipLaunchKernelGGL(bwt_aux_kernel, dim3((n_aux + 255) / 256), block, 0, st, rank, n, aux_rate, n_aux, d_aux.as<int32_t>());
    NS_HIP(hipGetLastError());
    NS_HIP(hipEventRecord(e1, st));
    NS_HIP(hipMemcpyAsync(out, d_out.p, n, hipMemcpyDeviceToHost, st));
    if (n_aux) NS_HIP(hipMemcpyAsync(aux, d_aux.p, (size_t)n_aux * 4, hipMemcpyDeviceToHost, st));
    NS_HIP(stream_wait(st));
    *primary_index = n_aux ? aux[0] : 0;
    if (!n_aux) {
        int32_t r0 = 0;
        NS_HIP(hipMemcpy(&r0, rank, 4, hipMemcpyDeviceToHost));
        *primary_index = r0;
    }
    float ms = 0;
    NS_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (gpu_ms_out) *gpu_ms_out = ms;
    if (rounds_out) *rounds_out = rounds;
    return NSGPU_OK;
}
