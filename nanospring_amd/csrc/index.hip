// index.hip -- a7: the n bucket tables (MinHashReadFilter::populateHashTables,
// src/ReadFilter.cpp:159-172; BBHashMap::initialize, src/BBHashMap.cpp:10-99).
//
// The reference builds, per table j, a minimal perfect hash over the distinct
// sketch values and a CSR of ascending read ids per value.  Only the mapping
// "value -> ascending id list" is observable (pushMatchesInVector,
// src/BBHashMap.cpp:101-120), so on the GPU each table is simply the column
// sketch[.][j] sorted by (value, read id): a value's id list is the contiguous
// run of that value and lookups are binary searches (kernels_minhash.hip).
//
// Sort = two stable LSD radix passes over all n*N entries at once
//   pass A: by the 64-bit value, carrying e = (j << 32 | r)
//   pass B: by bits 32..40 of e (the table index), carrying the value
// so the final order is (j, value, r).  The radix sort primitive is rocPRIM's
// device-wide onesweep sort (ROCm's own library; plain library call, as
// hipBLASLt would be for a plain GEMM).
#include "common.hpp"
#include <rocprim/rocprim.hpp>

namespace nsgpu {

__global__ __launch_bounds__(256) void index_transpose_kernel(const uint64_t *__restrict__ sketch, uint32_t N, uint32_t n,
                                                              uint64_t *__restrict__ keys, uint64_t *__restrict__ ents)
{
    // sketch is [N][n]; emit column-major entry (j, r) at j*N + r.  A 64x64 tile
    // through LDS keeps both sides coalesced.
    __shared__ uint64_t tile[64][65];
    const uint32_t r0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
    for (uint32_t rr = ty; rr < 64; rr += 4) {
        const uint32_t r = r0 + rr, j = j0 + tx;
        tile[rr][tx] = (r < N && j < n) ? sketch[(size_t)r * n + j] : 0ull;
    }
    __syncthreads();
    for (uint32_t jj = ty; jj < 64; jj += 4) {
        const uint32_t j = j0 + jj, r = r0 + tx;
        if (r < N && j < n) {
            keys[(size_t)j * N + r] = tile[tx][jj];
            ents[(size_t)j * N + r] = ((uint64_t)j << 32) | r;
        }
    }
}

__global__ __launch_bounds__(256) void index_split_kernel(const uint64_t *__restrict__ ents, uint64_t total, uint32_t *__restrict__ ids)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) ids[i] = (uint32_t)ents[i];
}

int build_index(nsgpu_ctx *c)
{
    const uint32_t N = c->reads.n, n = c->prm.n;
    const uint64_t total = (uint64_t)N * n;
    NS_TRY(c->idx_keys.reserve((total + 1) * 8));
    NS_TRY(c->idx_ids.reserve((total + 1) * 4));
    if (total == 0) return NSGPU_OK;
    NS_CHECK(total < (1ull << 32), NSGPU_ERR_RANGE, "index: n * N = %llu exceeds 2^32 entries per device; shard the reads", (unsigned long long)total);
    NS_TRY(c->idx_tmp_k.reserve(total * 8));
    NS_TRY(c->idx_tmp_v.reserve(total * 8));
    NS_TRY(c->idx_tmp_e.reserve(total * 8));
    NS_TRY(c->idx_tmp_e2.reserve(total * 8));
    uint64_t *k0 = c->idx_tmp_k.as<uint64_t>(), *e0 = c->idx_tmp_e.as<uint64_t>();
    uint64_t *k1 = c->idx_tmp_v.as<uint64_t>(), *e1 = c->idx_tmp_e2.as<uint64_t>();
    uint64_t *kf = c->idx_keys.as<uint64_t>();

    dim3 grid((N + 63) / 64, (n + 63) / 64);
    hipLaunchKernelGGL(index_transpose_kernel, grid, dim3(256), 0, c->stream, c->sketch.as<uint64_t>(), N, n, k0, e0);
    NS_HIP(hipGetLastError());

    size_t ws_a = 0, ws_b = 0;
    NS_HIP(rocprim::radix_sort_pairs(nullptr, ws_a, k0, k1, e0, e1, (size_t)total, 0u, 64u, c->stream));
    NS_HIP(rocprim::radix_sort_pairs(nullptr, ws_b, e1, e0, k1, kf, (size_t)total, 32u, 40u, c->stream));
    NS_TRY(c->idx_sort_ws.reserve((ws_a > ws_b ? ws_a : ws_b) + 16));
    // pass A: (value) -> k1, e1
    NS_HIP(rocprim::radix_sort_pairs(c->idx_sort_ws.p, ws_a, k0, k1, e0, e1, (size_t)total, 0u, 64u, c->stream));
    // pass B: keys = e (bits 32..39 = table), values = sketch value -> e0, kf
    NS_HIP(rocprim::radix_sort_pairs(c->idx_sort_ws.p, ws_b, e1, e0, k1, kf, (size_t)total, 32u, 40u, c->stream));
    hipLaunchKernelGGL(index_split_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, e0, total, c->idx_ids.as<uint32_t>());
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

// exclusive scan of n u32 values into n+1 u64 offsets (d_in must have a readable,
// zero element at index n).
int scan_u32_to_u64(nsgpu_ctx *c, const uint32_t *d_in, uint64_t *d_out, uint32_t n) { return scan_u32_to_u64(c->f_scan_ws, c->stream, d_in, d_out, n); }

int scan_u32_to_u64(DevBuf &scratch, hipStream_t stream, const uint32_t *d_in, uint64_t *d_out, uint32_t n)
{
    size_t ws = 0;
    NS_HIP(rocprim::exclusive_scan(nullptr, ws, d_in, d_out, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), stream));
    NS_TRY(scratch.reserve(ws + 16));
    NS_HIP(rocprim::exclusive_scan(scratch.p, ws, d_in, d_out, (uint64_t)0, (size_t)n + 1, rocprim::plus<uint64_t>(), stream));
    return NSGPU_OK;
}

}  // namespace nsgpu
