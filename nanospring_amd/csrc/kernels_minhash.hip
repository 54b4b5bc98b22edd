// kernels_minhash.hip -- gfx950 kernels for the MinHash half of the hot path:
//   a1  ASCII -> 2-bit rows            (src/dnaToBits.cpp:10-36)
//   a5  salted k-mer min-sketch        (src/ReadFilter.cpp:117-152)
//   a8/a9 bucket lookup + multiplicity (src/BBHashMap.cpp:101-120, src/ReadFilter.cpp:65-83)
//   a10 repetitive-read flag           (src/Consensus.cpp:405-424)
// All integer work; wave = 64 lanes; no MFMA anywhere.
#include "common.hpp"

namespace nsgpu {

// ----------------------------------------------------------------------------
// a1: pack.  One workgroup per read (grid-stride), lanes over output dwords.
// A dword holds 16 bases MSB-first in MEMORY byte order (byte 0 = bases 0..3),
// which is exactly the DnaBitset byte layout.
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t base_code(uint32_t c) { return (c & 2u) | ((c & 4u) >> 2); }

__host__ __device__ __forceinline__ uint64_t row_bytes(uint32_t len)
{
    return ((((uint64_t)len + 3) / 4 + 15) & ~(uint64_t)15) + 16;
}

__global__ __launch_bounds__(256) void pack_ascii_kernel(const char *__restrict__ ascii, const uint64_t *__restrict__ aoff,
                                                         const uint64_t *__restrict__ poff, const uint32_t *__restrict__ len,
                                                         uint8_t *__restrict__ packed, uint32_t n)
{
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint8_t *s = reinterpret_cast<const uint8_t *>(ascii) + aoff[r];
        const uint32_t L = len[r];
        const uint32_t ndw = (uint32_t)(row_bytes(L) / 4);
        uint32_t *out = reinterpret_cast<uint32_t *>(packed + poff[r]);
        for (uint32_t w = threadIdx.x; w < ndw; w += blockDim.x) {
            uint32_t v = 0;
            const uint32_t b0 = w * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t p = b0 + i;
                const uint32_t code = p < L ? base_code(s[p]) : 0u;
                v = (v << 2) | code;
            }
            out[w] = __builtin_bswap32(v);
        }
    }
}

int launch_pack_ascii(nsgpu_ctx *c, const char *d_ascii, const uint64_t *d_aoff, SeqStore &st)
{
    if (st.n == 0) return NSGPU_OK;
    uint32_t grid = st.n < 65536u ? st.n : 65536u;
    hipLaunchKernelGGL(pack_ascii_kernel, dim3(grid), dim3(256), 0, c->stream, d_ascii, d_aoff,
                       st.poff.as<uint64_t>(), st.len.as<uint32_t>(), st.packed.as<uint8_t>(), st.n);
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

// ----------------------------------------------------------------------------
// a5: sketch.  One 256-thread workgroup per sequence.  Per chunk of SK_CHUNK
// k-mer positions:
//   stage  - the chunk's packed dwords go to LDS (byte-swapped, so that bit 31 is
//            the first base);
//   expand - every thread rebuilds 4 k-mers from three LDS dwords and stores them
//            (and, when asked, their reverse complements) as u64 in LDS;
//   reduce - lanes own SALTS (lane l <-> salt g*64+l), waves own k-mer subsets;
//            each wave streams its subset out of LDS with broadcast 16-byte reads
//            and keeps a running min of kmer ^ salt per lane.
// The 64-bit "hash" is the identity (libstdc++ std::hash<uint64_t>), see
// SURVEY 0, determinism trap 2.
// MODE bit 0: forward sketch, bit 1: reverse-complement sketch.
// ----------------------------------------------------------------------------
constexpr int SK_CHUNK = 1024;

__device__ __forceinline__ uint64_t revcomp_kmer(uint64_t x, uint32_t k)
{
    uint64_t y = __brevll(x);
    y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
    y >>= (64 - 2 * k);
    const uint64_t mask = (~0ull) >> (64 - 2 * k);
    return (y ^ 0x5555555555555555ull) & mask;   // complement = code ^ 1 in the A0 T1 C2 G3 alphabet
}

template <int MODE>
__global__ __launch_bounds__(256) void sketch_kernel(const uint8_t *__restrict__ packed, const uint64_t *__restrict__ poff,
                                                     const uint32_t *__restrict__ len, uint32_t nseq, uint32_t k, uint32_t n,
                                                     const uint64_t *__restrict__ salts, uint64_t *__restrict__ out_f,
                                                     uint64_t *__restrict__ out_r)
{
    constexpr bool FWD = (MODE & 1) != 0, REV = (MODE & 2) != 0;
    __shared__ uint32_t s_dw[SK_CHUNK / 16 + 4];
    __shared__ __attribute__((aligned(16))) uint64_t s_kf[FWD ? SK_CHUNK + 2 : 2];
    __shared__ __attribute__((aligned(16))) uint64_t s_kr[REV ? SK_CHUNK + 2 : 2];
    __shared__ uint64_t s_red[2][4][64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_groups = (n + 63) >> 6;          // 1..4 groups of 64 salts
    const uint32_t n_subsets = 4 / n_groups;           // k-mer subsets per chunk
    const uint32_t grp = wave % n_groups, subset = wave / n_groups;
    const bool wave_on = subset < n_subsets;
    const uint32_t salt_idx = grp * 64 + lane;
    const bool lane_on = wave_on && salt_idx < n;
    const uint64_t salt = lane_on ? salts[salt_idx] : 0ull;
    const uint32_t per = SK_CHUNK / n_subsets;

    for (uint32_t seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
        const uint32_t L = len[seq];
        if (L + 1 < k) {            // len < k-1: the reference leaves the zero-initialised row untouched
            if (tid < n) {
                if (FWD) out_f[(size_t)seq * n + tid] = 0ull;
                if (REV) out_r[(size_t)seq * n + tid] = 0ull;
            }
            continue;
        }
        const uint32_t nk = L + 1 - k;                  // may be 0 (len == k-1): row of all ones
        const uint32_t *row = reinterpret_cast<const uint32_t *>(packed + poff[seq]);
        const uint32_t row_dw = (uint32_t)(row_bytes(L) / 4);
        uint64_t mf = ~0ull, mr = ~0ull;

        for (uint32_t cb = 0; cb < nk; cb += SK_CHUNK) {
            const uint32_t cnt = (nk - cb) < (uint32_t)SK_CHUNK ? (nk - cb) : (uint32_t)SK_CHUNK;
            if (tid < SK_CHUNK / 16 + 4) {
                const uint32_t gd = cb / 16 + tid;
                s_dw[tid] = gd < row_dw ? __builtin_bswap32(row[gd]) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lp = tid * 4 + j;
                if (lp < cnt) {
                    const uint32_t di = lp >> 4, o = (lp & 15) * 2;
                    const uint64_t hi = ((uint64_t)s_dw[di] << 32) | s_dw[di + 1];
                    const uint64_t v = (hi << o) | ((uint64_t)s_dw[di + 2] >> (32 - o));
                    const uint64_t km = v >> (64 - 2 * k);
                    if (FWD) { s_kf[lp] = km; if (lp + 1 == cnt) s_kf[lp + 1] = km; }
                    if (REV) { const uint64_t kr = revcomp_kmer(km, k); s_kr[lp] = kr; if (lp + 1 == cnt) s_kr[lp + 1] = kr; }
                }
            }
            __syncthreads();
            if (wave_on) {
                const uint32_t lo = subset * per;
                uint32_t hi = lo + per < cnt ? lo + per : cnt;
                hi = (hi + 1) & ~1u;                    // odd tail: s_k?[cnt] duplicates the last k-mer
                if (FWD) {
#pragma unroll 4
                    for (uint32_t i = lo; i < hi; i += 2) {
                        const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(&s_kf[i]);
                        const uint64_t a = kk.x ^ salt, b = kk.y ^ salt;
                        mf = a < mf ? a : mf;
                        mf = b < mf ? b : mf;
                    }
                }
                if (REV) {
#pragma unroll 4
                    for (uint32_t i = lo; i < hi; i += 2) {
                        const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(&s_kr[i]);
                        const uint64_t a = kk.x ^ salt, b = kk.y ^ salt;
                        mr = a < mr ? a : mr;
                        mr = b < mr ? b : mr;
                    }
                }
            }
            __syncthreads();
        }
        // combine the k-mer subsets of each salt group
        if (FWD) s_red[0][wave][lane] = mf;
        if (REV) s_red[1][wave][lane] = mr;
        __syncthreads();
        if (tid < n) {
            const uint32_t g = tid >> 6, l = tid & 63;
            if (FWD) {
                uint64_t m = ~0ull;
                for (uint32_t s = 0; s < n_subsets; ++s) { const uint64_t x = s_red[0][s * n_groups + g][l]; m = x < m ? x : m; }
                out_f[(size_t)seq * n + tid] = m;
            }
            if (REV) {
                uint64_t m = ~0ull;
                for (uint32_t s = 0; s < n_subsets; ++s) { const uint64_t x = s_red[1][s * n_groups + g][l]; m = x < m ? x : m; }
                out_r[(size_t)seq * n + tid] = m;
            }
        }
        __syncthreads();
    }
}

int launch_sketch(nsgpu_ctx *c, const SeqStore &st, uint64_t *d_out_fwd, uint64_t *d_out_rc)
{
    if (st.n == 0) return NSGPU_OK;
    const uint32_t grid = st.n < 262144u ? st.n : 262144u;
    const uint8_t *pk = st.packed.as<uint8_t>();
    const uint64_t *po = st.poff.as<uint64_t>();
    const uint32_t *ln = st.len.as<uint32_t>();
    const uint64_t *sa = c->salts.as<uint64_t>();
    if (d_out_fwd && d_out_rc)
        hipLaunchKernelGGL(sketch_kernel<3>, dim3(grid), dim3(256), 0, c->stream, pk, po, ln, st.n, c->prm.k, c->prm.n, sa, d_out_fwd, d_out_rc);
    else if (d_out_fwd)
        hipLaunchKernelGGL(sketch_kernel<1>, dim3(grid), dim3(256), 0, c->stream, pk, po, ln, st.n, c->prm.k, c->prm.n, sa, d_out_fwd, d_out_rc);
    else
        hipLaunchKernelGGL(sketch_kernel<2>, dim3(grid), dim3(256), 0, c->stream, pk, po, ln, st.n, c->prm.k, c->prm.n, sa, d_out_fwd, d_out_rc);
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

// forward sketches of reads lo..hi only (multi-GPU: every rank sketches its own id range, the rows of the
// other ranks arrive by all-gather)
int launch_sketch_range(nsgpu_ctx *c, const SeqStore &st, uint32_t lo, uint32_t hi, uint64_t *d_out_fwd)
{
    if (hi <= lo) return NSGPU_OK;
    const uint32_t cnt = hi - lo, n = c->prm.n;
    const uint32_t grid = cnt < 262144u ? cnt : 262144u;
    hipLaunchKernelGGL(sketch_kernel<1>, dim3(grid), dim3(256), 0, c->stream, st.packed.as<uint8_t>(), st.poff.as<uint64_t>() + lo, st.len.as<uint32_t>() + lo, cnt,
                       c->prm.k, n, c->salts.as<uint64_t>(), d_out_fwd + (size_t)lo * n, (uint64_t *)nullptr);
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

// ----------------------------------------------------------------------------
// a10: checkRepetitive.  One workgroup per read; for shift 1..6 count the
// positions j with read[j] == read[(j+shift) % L]; flag if any count exceeds
// 0.7 * L (double compare, as the reference).  Works on the 2-bit codes, which
// are injective on the A/T/C/G strings ReadData::getRead returns.
// ----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t code_at(const uint8_t *row, uint32_t p) { return (row[p >> 2] >> (6 - 2 * (p & 3))) & 3u; }

__global__ __launch_bounds__(256) void repetitive_kernel(const uint8_t *__restrict__ packed, const uint64_t *__restrict__ poff,
                                                         const uint32_t *__restrict__ len, uint32_t nseq, uint8_t *__restrict__ flags)
{
    __shared__ uint32_t s_cnt[6];
    for (uint32_t seq = blockIdx.x; seq < nseq; seq += gridDim.x) {
        const uint32_t L = len[seq];
        const uint8_t *row = packed + poff[seq];
        if (threadIdx.x < 6) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        uint32_t c[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t j = threadIdx.x; j < L; j += blockDim.x) {
            const uint32_t a = code_at(row, j);
#pragma unroll
            for (uint32_t sh = 1; sh <= 6; ++sh) {
                uint32_t p = j + sh;
                p = p >= L ? p % L : p;
                c[sh - 1] += (code_at(row, p) == a);
            }
        }
#pragma unroll
        for (int sh = 0; sh < 6; ++sh) {
            uint32_t v = c[sh];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt[sh], v);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint8_t f = 0;
            for (int sh = 0; sh < 6; ++sh)
                if ((double)s_cnt[sh] > 0.7 * (double)L) f = 1;
            flags[seq] = f;
        }
        __syncthreads();
    }
}

int launch_repetitive(nsgpu_ctx *c, const SeqStore &st, uint8_t *d_flags)
{
    if (st.n == 0) return NSGPU_OK;
    const uint32_t grid = st.n < 65536u ? st.n : 65536u;
    hipLaunchKernelGGL(repetitive_kernel, dim3(grid), dim3(256), 0, c->stream, st.packed.as<uint8_t>(), st.poff.as<uint64_t>(),
                       st.len.as<uint32_t>(), st.n, d_flags);
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

// ----------------------------------------------------------------------------
// a8/a9: candidate filter.
//   search : one wave per query, lanes over the n tables.  Each lane finds the
//            [lb, lb+cnt) run of its sketch value in its table's sorted key column
//            (the run IS the reference's per-key id list, ascending by read id).
//            M = sum(cnt); at most floor(M / thr) ids can pass, which sizes the
//            query's slice of the staging pool exactly -> no atomics, no retry.
//   count  : one wave per query: gather the M ids into LDS, bitonic sort, emit
//            the ids whose run length is >= thr, ascending.
//   heavy  : queries with M > F_CAP (repeat-rich data) use per-workgroup counter
//            arrays in HBM instead of an LDS sort.
//   compact: staging slices -> final CSR (after an exclusive scan of the counts).
// ----------------------------------------------------------------------------
constexpr uint32_t F_CAP = 2048;
constexpr uint32_t F_HEAVY_WGS = 16;

__device__ __forceinline__ const uint64_t *query_row(const uint64_t *qa, const uint64_t *qb, bool interleave, uint32_t q, uint32_t n)
{
    if (!interleave) return qa + (size_t)q * n;
    return ((q & 1) ? qb : qa) + (size_t)(q >> 1) * n;
}

__global__ __launch_bounds__(256) void filter_search_kernel(const uint64_t *__restrict__ qa, const uint64_t *__restrict__ qb, int interleave,
                                                            uint32_t nq, uint32_t n, uint32_t thr1, uint32_t N,
                                                            const uint64_t *__restrict__ idx_keys, uint32_t *__restrict__ lb_out,
                                                            uint32_t *__restrict__ cnt_out, uint32_t *__restrict__ qm, uint32_t *__restrict__ qcap)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = wave_global; q < nq; q += n_waves) {
        const uint64_t *qs = query_row(qa, qb, interleave != 0, q, n);
        uint32_t msum = 0;
        for (uint32_t l = lane; l < ((n + 63) & ~63u); l += 64) {
            uint32_t lb = 0, c = 0;
            if (l < n) {
                const uint64_t key = qs[l];
                const uint64_t *K = idx_keys + (size_t)l * N;
                uint32_t lo = 0, hi = N;
                while (lo < hi) {
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    if (K[mid] < key) lo = mid + 1; else hi = mid;
                }
                lb = lo;
                if (lb < N && K[lb] == key) {
                    // gallop for the end of the run, then bisect
                    uint32_t step = 1, a = lb, b;
                    while (a + step < N && K[a + step] == key) { a += step; step <<= 1; }
                    b = a + step < N ? a + step : N;      // K[a]==key, K[b]!=key or b==N
                    while (a + 1 < b) {
                        const uint32_t mid = a + ((b - a) >> 1);
                        if (K[mid] == key) a = mid; else b = mid;
                    }
                    c = a + 1 - lb;
                }
                lb_out[(size_t)q * n + l] = lb;
                cnt_out[(size_t)q * n + l] = c;
            }
            msum += c;
        }
        for (int o = 32; o > 0; o >>= 1) msum += __shfl_down(msum, o, 64);
        if (lane == 0) { qm[q] = msum; qcap[q] = msum / thr1; }
    }
}

__global__ __launch_bounds__(64) void filter_count_kernel(uint32_t nq, uint32_t n, uint32_t thr1, uint32_t N,
                                                          const uint32_t *__restrict__ idx_ids, const uint32_t *__restrict__ lb_in,
                                                          const uint32_t *__restrict__ cnt_in, const uint32_t *__restrict__ qm,
                                                          const uint64_t *__restrict__ soff, uint32_t *__restrict__ staging,
                                                          uint32_t *__restrict__ qcnt, uint32_t *__restrict__ ovf_list,
                                                          uint32_t *__restrict__ ctrl, uint64_t staging_cap)
{
    __shared__ uint32_t s_ids[F_CAP];
    __shared__ uint32_t s_off[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint32_t M = qm[q];
        if (M == 0) { if (lane == 0) qcnt[q] = 0; continue; }
        // (the single-wait path sizes the staging area before it knows the total: a query whose slice does not fit is dropped and
        // the batch flagged -- the caller redoes it the exact way)
        if (soff[q + 1] > staging_cap) { if (lane == 0) { qcnt[q] = 0; ctrl[1] = 1; } continue; }
        if (M > F_CAP) {
            if (lane == 0) { const uint32_t slot = atomicAdd(&ctrl[0], 1u); ovf_list[slot] = q; qcnt[q] = 0; }
            continue;
        }
        // exclusive prefix of the per-table counts (n <= 256)
        uint32_t carry = 0;
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t l = base + lane;
            const uint32_t c = l < n ? cnt_in[(size_t)q * n + l] : 0;
            uint32_t inc = c;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)lane >= o) inc += t; }
            if (l < n) s_off[l] = carry + inc - c;
            carry += __shfl(inc, 63, 64);
        }
        __syncthreads();
        for (uint32_t l = lane; l < n; l += 64) {
            const uint32_t c = cnt_in[(size_t)q * n + l];
            if (c) {
                const uint32_t *src = idx_ids + (size_t)l * N + lb_in[(size_t)q * n + l];
                const uint32_t o = s_off[l];
                for (uint32_t t = 0; t < c; ++t) s_ids[o + t] = src[t];
            }
        }
        uint32_t P = 64;
        while (P < M) P <<= 1;
        for (uint32_t i = M + lane; i < P; i += 64) s_ids[i] = 0xFFFFFFFFu;
        __syncthreads();
        // bitonic sort of P (power of two) keys by one wave
        for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < (P >> 1); i += 64) {
                    const uint32_t pos = 2 * j * (i / j) + (i % j);
                    const uint32_t a = s_ids[pos], b = s_ids[pos + j];
                    const bool up = (pos & k2) == 0;
                    if ((a > b) == up) { s_ids[pos] = b; s_ids[pos + j] = a; }
                }
                __syncthreads();
            }
        }
        // ids whose multiplicity reaches thr1, in ascending order; compacted in place
        uint32_t total = 0;
        for (uint32_t base = 0; base < M; base += 64) {
            const uint32_t i = base + lane;
            uint32_t v = 0;
            bool ok = false;
            if (i < M) {
                v = s_ids[i];
                const bool start = (i == 0) || (s_ids[i - 1] != v);
                ok = start && (i + thr1 - 1 < M) && (s_ids[i + thr1 - 1] == v);
            }
            const unsigned long long mask = __ballot(ok);
            const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
            __syncthreads();
            if (ok) s_ids[total + rank] = v;
            total += __popcll(mask);
            __syncthreads();
        }
        uint32_t *dst = staging + soff[q];
        for (uint32_t i = lane; i < total; i += 64) dst[i] = s_ids[i];
        if (lane == 0) qcnt[q] = total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void filter_heavy_kernel(uint32_t n, uint32_t thr1, uint32_t N, const uint32_t *__restrict__ idx_ids,
                                                           const uint32_t *__restrict__ lb_in, const uint32_t *__restrict__ cnt_in,
                                                           const uint64_t *__restrict__ soff, uint32_t *__restrict__ staging,
                                                           uint32_t *__restrict__ qcnt, const uint32_t *__restrict__ ovf_list,
                                                           uint32_t *__restrict__ ctrl, uint32_t *__restrict__ counters, uint64_t staging_cap)
{
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_total;
    uint32_t *cn = counters + (size_t)blockIdx.x * N;     // zero on entry, zero on exit
    const uint32_t n_ovf = ctrl[0];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t oi = blockIdx.x; oi < n_ovf; oi += gridDim.x) {
        const uint32_t q = ovf_list[oi];
        if (soff[q + 1] > staging_cap) { if (threadIdx.x == 0) ctrl[1] = 1; continue; }
        for (uint32_t l = 0; l < n; ++l) {
            const uint32_t c = cnt_in[(size_t)q * n + l];
            const uint32_t *src = idx_ids + (size_t)l * N + lb_in[(size_t)q * n + l];
            for (uint32_t t = threadIdx.x; t < c; t += 256) atomicAdd(&cn[src[t]], 1u);
        }
        __threadfence();
        if (threadIdx.x == 0) s_total = 0;
        __syncthreads();
        uint32_t *dst = staging + soff[q];
        for (uint32_t base = 0; base < N; base += 256) {
            const uint32_t r = base + threadIdx.x;
            uint32_t v = 0;
            if (r < N) {
                v = __hip_atomic_load(&cn[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v) __hip_atomic_store(&cn[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const bool ok = v >= thr1 && v > 0;
            const unsigned long long mask = __ballot(ok);
            if (lane == 0) s_wsum[wave] = __popcll(mask);
            __syncthreads();
            uint32_t before = s_total;
            for (uint32_t w = 0; w < wave; ++w) before += s_wsum[w];
            if (ok) dst[before + __popcll(mask & ((1ull << lane) - 1ull))] = r;
            __syncthreads();
            if (threadIdx.x == 0) s_total += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
            __syncthreads();
        }
        if (threadIdx.x == 0) qcnt[q] = s_total;
        __threadfence();
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void filter_compact_kernel(uint32_t nq, const uint64_t *__restrict__ soff, const uint32_t *__restrict__ staging,
                                                             const uint32_t *__restrict__ qcnt, const uint64_t *__restrict__ off,
                                                             uint32_t *__restrict__ ids, uint64_t ids_cap, uint64_t *__restrict__ off_copy, uint32_t *__restrict__ ctrl)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    // (off_copy: the offsets next to the ids, e.g. both in pinned host memory; ids_cap: a destination sized before the total was known)
    if (off_copy) for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q <= nq; q += gridDim.x * blockDim.x) off_copy[q] = off[q];
    if (off[nq] > ids_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[2] = 1; return; }
    for (uint32_t q = wave_global; q < nq; q += n_waves) {
        const uint32_t c = qcnt[q];
        const uint32_t *src = staging + soff[q];
        uint32_t *dst = ids + off[q];
        for (uint32_t i = lane; i < c; i += 64) dst[i] = src[i];
    }
}

// Runs the whole filter for nq query sketches already on the device.  Results:
// c->f_off (nq+1 u64), c->f_ids, c->f_total.
int run_filter(nsgpu_ctx *c, const uint64_t *d_q_even, const uint64_t *d_q_odd, uint32_t nq, bool interleave)
{
    const uint32_t n = c->prm.n, N = c->reads.n;
    const uint32_t thr1 = c->prm.overlap_sketch_thr ? c->prm.overlap_sketch_thr : 1u;
    c->f_nq = nq;
    c->f_total = 0;
    NS_TRY(c->f_off.reserve(((size_t)nq + 1) * 8));
    if (nq == 0) { NS_HIP(hipMemsetAsync(c->f_off.p, 0, 8, c->stream)); return NSGPU_OK; }
    NS_TRY(c->f_qstart.reserve(((size_t)nq + 1) * 8));            // staging offsets (u64)
    NS_TRY(c->f_qcnt.reserve(((size_t)nq + 1) * 4 * 3));           // qcnt | qm | qcap, each nq+1 u32
    NS_TRY(c->f_qm.reserve((size_t)nq * n * 8));                   // lb | cnt, each nq*n u32
    NS_TRY(c->f_ctrl.reserve(64));
    NS_TRY(c->f_ovf_list.reserve((size_t)nq * 4));
    uint32_t *qcnt = c->f_qcnt.as<uint32_t>(), *qm = qcnt + (nq + 1), *qcap = qm + (nq + 1);
    uint32_t *lb = c->f_qm.as<uint32_t>(), *cnt = lb + (size_t)nq * n;
    uint64_t *soff = c->f_qstart.as<uint64_t>();
    NS_HIP(hipMemsetAsync(c->f_ctrl.p, 0, 64, c->stream));
    NS_HIP(hipMemsetAsync(qcnt, 0, ((size_t)nq + 1) * 4 * 3, c->stream));

    NS_HIP(hipEventRecord(c->t_kernel.a, c->stream));
    {
        const uint32_t waves = nq;
        uint32_t grid = (waves + 3) / 4;
        if (grid > 65536u) grid = 65536u;
        hipLaunchKernelGGL(filter_search_kernel, dim3(grid), dim3(256), 0, c->stream, d_q_even, d_q_odd, interleave ? 1 : 0, nq, n, thr1, N,
                           c->idx_keys.as<uint64_t>(), lb, cnt, qm, qcap);
        NS_HIP(hipGetLastError());
    }
    NS_TRY(scan_u32_to_u64(c, qcap, soff, nq));
    // scalars come back through pinned memory: a copy into a pageable variable blocks inside the runtime, busy-waiting
    NS_TRY(c->pin_small.reserve(64));
    volatile uint64_t *const ps = c->pin_small.as<volatile uint64_t>();       // [0] staging total [1] overflow count [2] f_total [3] matches
    uint64_t staging_total = 0, m_total = 0;
    NS_HIP(hipMemcpyAsync(c->pin_small.p, soff + nq, 8, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait_short(c->stream));
    staging_total = ps[0];
    NS_TRY(c->f_pool.reserve((staging_total + 1) * 4));
    {
        uint32_t grid = nq < 262144u ? nq : 262144u;
        hipLaunchKernelGGL(filter_count_kernel, dim3(grid), dim3(64), 0, c->stream, nq, n, thr1, N, c->idx_ids.as<uint32_t>(), lb, cnt, qm, soff,
                           c->f_pool.as<uint32_t>(), qcnt, c->f_ovf_list.as<uint32_t>(), c->f_ctrl.as<uint32_t>(), ~0ull);
        NS_HIP(hipGetLastError());
    }
    ps[1] = 0;
    NS_HIP(hipMemcpyAsync(c->pin_small.as<uint64_t>() + 1, c->f_ctrl.p, 4, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait_short(c->stream));
    const uint32_t n_ovf = (uint32_t)ps[1];
    if (n_ovf) {
        const uint32_t wgs = n_ovf < F_HEAVY_WGS ? n_ovf : F_HEAVY_WGS;
        const size_t need = (size_t)F_HEAVY_WGS * N * 4;
        if (c->f_ovf_cnt.cap < need) {
            NS_TRY(c->f_ovf_cnt.reserve(need));
            NS_HIP(hipMemsetAsync(c->f_ovf_cnt.p, 0, c->f_ovf_cnt.cap, c->stream));
        }
        hipLaunchKernelGGL(filter_heavy_kernel, dim3(wgs), dim3(256), 0, c->stream, n, thr1, N, c->idx_ids.as<uint32_t>(), lb, cnt, soff,
                           c->f_pool.as<uint32_t>(), qcnt, c->f_ovf_list.as<uint32_t>(), c->f_ctrl.as<uint32_t>(), c->f_ovf_cnt.as<uint32_t>(), ~0ull);
        NS_HIP(hipGetLastError());
    }
    NS_TRY(scan_u32_to_u64(c, qcnt, c->f_off.as<uint64_t>(), nq));
    NS_HIP(hipMemcpyAsync(c->pin_small.as<uint64_t>() + 2, c->f_off.as<uint64_t>() + nq, 8, hipMemcpyDeviceToHost, c->stream));
    NS_HIP(stream_wait_short(c->stream));
    c->f_total = ps[2];
    NS_TRY(c->f_ids.reserve((c->f_total + 1) * 4));
    {
        uint32_t grid = (nq + 3) / 4;
        if (grid > 65536u) grid = 65536u;
        hipLaunchKernelGGL(filter_compact_kernel, dim3(grid), dim3(256), 0, c->stream, nq, soff, c->f_pool.as<uint32_t>(), qcnt,
                           c->f_off.as<uint64_t>(), c->f_ids.as<uint32_t>(), ~0ull, (uint64_t *)nullptr, c->f_ctrl.as<uint32_t>());
        NS_HIP(hipGetLastError());
    }
    NS_HIP(hipEventRecord(c->t_kernel.b, c->stream));
    // total M (for the algorithmic-bytes figure): scan qm into the (now free) lb area
    {
        uint64_t *tmp = reinterpret_cast<uint64_t *>(lb);
        if (c->filter_stats && (size_t)nq * n * 8 >= ((size_t)nq + 1) * 8) {
            NS_TRY(scan_u32_to_u64(c, qm, tmp, nq));
            ps[3] = 0;
            NS_HIP(hipMemcpyAsync(c->pin_small.as<uint64_t>() + 3, tmp + nq, 8, hipMemcpyDeviceToHost, c->stream));
            m_total = 1;                                   // marker: read ps[3] after the wait
        }
    }
    NS_HIP(stream_wait_short(c->stream));
    if (m_total) m_total = ps[3];
    NS_HIP(hipEventElapsedTime(&c->timing.filter_kernel_ms, c->t_kernel.a, c->t_kernel.b));
    c->timing.filter_matches = m_total;
    return NSGPU_OK;
}


// ----------------------------------------------------------------------------
// a5 + a8 + a9 fused: ONE kernel answers a batch of window queries (ReadFilter::getFilteredReads of a string, src/ReadFilter.cpp:49-83).
// A slot of the contig engine asks for a few dozen windows and waits for the answer: what that costs is the chain of dependent GPU
// operations, not their volume (pack -> sketch -> search -> scan -> count -> heavy -> scan -> compact + six copies: 0.37 ms per slot for
// 0.1 ms of kernels).  Here one 256-thread workgroup takes a query from its ASCII string to its candidate list: the sketch (the same
// stage / expand / reduce as sketch_kernel, the 2-bit dwords made from the text in LDS), one binary search per table (threads over the n
// tables), the M matching ids gathered into LDS, a bitonic sort by the whole workgroup, the ids whose multiplicity reaches the threshold
// written in ascending order into the query's slot of a pinned host buffer.  Queries the LDS sort cannot take (M > F_CAP: repeat-rich
// data) or with more results than a slot holds raise a flag, and the caller redoes the batch with the multi-pass kernels.
// ----------------------------------------------------------------------------
constexpr uint32_t WQ_SLOT = 512;            // result ids per query slot (a query can have floor(M / thr) <= F_CAP / thr results)

__global__ __launch_bounds__(256) void window_query_kernel(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ aoff, uint32_t nq, uint32_t k, uint32_t n,
                                                           uint32_t thr1, uint32_t N, const uint64_t *__restrict__ salts, const uint64_t *__restrict__ idx_keys,
                                                           const uint32_t *__restrict__ idx_ids, uint32_t *__restrict__ out_cnt, uint32_t *__restrict__ out_ids,
                                                           uint32_t *__restrict__ flags)
{
    __shared__ uint32_t s_dw[SK_CHUNK / 16 + 4];
    __shared__ __attribute__((aligned(16))) uint64_t s_kf[SK_CHUNK + 2];
    __shared__ uint64_t s_red[4][64];
    __shared__ uint64_t s_sk[256];
    __shared__ uint32_t s_lb[256], s_cn[256], s_off[257];
    __shared__ uint32_t s_ids[F_CAP];
    __shared__ uint32_t s_w[4], s_tot;
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *str = ascii + aoff[q];
    const uint32_t L = (uint32_t)(aoff[q + 1] - aoff[q]);
    // ---- the sketch (string2Sketch, src/ReadFilter.cpp:117-131): see sketch_kernel ----
    {
        const uint32_t n_groups = (n + 63) >> 6, n_subsets = 4 / n_groups;
        const uint32_t grp = wave % n_groups, subset = wave / n_groups;
        const bool wave_on = subset < n_subsets;
        const uint32_t salt_idx = grp * 64 + lane;
        const uint64_t salt = wave_on && salt_idx < n ? salts[salt_idx] : 0ull;
        const uint32_t per = SK_CHUNK / n_subsets;
        if (L + 1 < k) { if (tid < n) s_sk[tid] = 0ull; }            // len < k-1: the zero-initialised row
        else {
            const uint32_t nk = L + 1 - k;                             // 0 (len == k-1): all ones
            uint64_t mf = ~0ull;
            for (uint32_t cb = 0; cb < nk; cb += SK_CHUNK) {
                const uint32_t cnt = (nk - cb) < (uint32_t)SK_CHUNK ? (nk - cb) : (uint32_t)SK_CHUNK;
                if (tid < SK_CHUNK / 16 + 4) {
                    // dword gd of the 2-bit row, bit 31 = its first base (pack_ascii_kernel + the byte swap of sketch_kernel's staging)
                    const uint32_t b0 = (cb / 16 + tid) * 16;
                    uint32_t v = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const uint32_t p = b0 + i; v = (v << 2) | (p < L ? base_code(str[p]) : 0u); }
                    s_dw[tid] = v;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lp = tid * 4 + j;
                    if (lp < cnt) {
                        const uint32_t di = lp >> 4, o = (lp & 15) * 2;
                        const uint64_t hi = ((uint64_t)s_dw[di] << 32) | s_dw[di + 1];
                        const uint64_t v = (hi << o) | ((uint64_t)s_dw[di + 2] >> (32 - o));
                        const uint64_t km = v >> (64 - 2 * k);
                        s_kf[lp] = km;
                        if (lp + 1 == cnt) s_kf[lp + 1] = km;
                    }
                }
                __syncthreads();
                if (wave_on) {
                    const uint32_t lo = subset * per;
                    uint32_t hi = lo + per < cnt ? lo + per : cnt;
                    hi = (hi + 1) & ~1u;
#pragma unroll 4
                    for (uint32_t i = lo; i < hi; i += 2) {
                        const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(&s_kf[i]);
                        const uint64_t a = kk.x ^ salt, b = kk.y ^ salt;
                        mf = a < mf ? a : mf;
                        mf = b < mf ? b : mf;
                    }
                }
                __syncthreads();
            }
            s_red[wave][lane] = mf;
            __syncthreads();
            if (tid < n) {
                const uint32_t g = tid >> 6, l = tid & 63;
                uint64_t m = ~0ull;
                for (uint32_t sb = 0; sb < n_subsets; ++sb) { const uint64_t x = s_red[sb * n_groups + g][l]; m = x < m ? x : m; }
                s_sk[tid] = m;
            }
        }
    }
    __syncthreads();
    // ---- one table per thread: the run [lb, lb + c) of the query's sketch value (filter_search_kernel) ----
    {
        uint32_t lb = 0, c = 0;
        if (tid < n && N) {
            const uint64_t key = s_sk[tid];
            const uint64_t *K = idx_keys + (size_t)tid * N;
            uint32_t lo = 0, hi = N;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (K[mid] < key) lo = mid + 1; else hi = mid;
            }
            lb = lo;
            if (lb < N && K[lb] == key) {
                uint32_t step = 1, a = lb, b;
                while (a + step < N && K[a + step] == key) { a += step; step <<= 1; }
                b = a + step < N ? a + step : N;
                while (a + 1 < b) {
                    const uint32_t mid = a + ((b - a) >> 1);
                    if (K[mid] == key) a = mid; else b = mid;
                }
                c = a + 1 - lb;
            }
        }
        s_lb[tid] = lb, s_cn[tid] = c;
    }
    __syncthreads();
    if (tid == 0) { uint32_t t = 0; for (uint32_t l = 0; l < n; ++l) { s_off[l] = t; t += s_cn[l]; } s_off[n] = t; }
    __syncthreads();
    const uint32_t M = s_off[n];
    if (M == 0) { if (tid == 0) out_cnt[q] = 0; return; }
    if (M > F_CAP || M / thr1 > WQ_SLOT) { if (tid == 0) { out_cnt[q] = 0; flags[0] = 1; } return; }
    // ---- the M ids into LDS, sorted (filter_count_kernel, with four waves) ----
    if (tid < n) {
        const uint32_t c = s_cn[tid];
        const uint32_t *src = idx_ids + (size_t)tid * N + s_lb[tid];
        const uint32_t o = s_off[tid];
        for (uint32_t t = 0; t < c; ++t) s_ids[o + t] = src[t];
    }
    uint32_t P = 64;
    while (P < M) P <<= 1;
    __syncthreads();
    for (uint32_t i = M + tid; i < P; i += 256) s_ids[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= P; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < (P >> 1); i += 256) {
                const uint32_t pos = 2 * j * (i / j) + (i % j);
                const uint32_t a = s_ids[pos], b = s_ids[pos + j];
                const bool up = (pos & k2) == 0;
                if ((a > b) == up) { s_ids[pos] = b; s_ids[pos + j] = a; }
            }
            __syncthreads();
        }
    }
    // ---- ids whose multiplicity reaches thr1, ascending, into the query's slot ----
    uint32_t *dst = out_ids + (size_t)q * WQ_SLOT;
    if (tid == 0) s_tot = 0;
    __syncthreads();
    for (uint32_t base = 0; base < M; base += 256) {
        const uint32_t i = base + tid;
        uint32_t v = 0;
        bool ok = false;
        if (i < M) {
            v = s_ids[i];
            const bool start = (i == 0) || (s_ids[i - 1] != v);
            ok = start && (i + thr1 - 1 < M) && (s_ids[i + thr1 - 1] == v);
        }
        const unsigned long long mask = __ballot(ok);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = s_tot;
        for (uint32_t w = 0; w < wave; ++w) before += s_w[w];
        if (ok) dst[before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = v;
        __syncthreads();
        if (tid == 0) s_tot += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (tid == 0) out_cnt[q] = s_tot;
}

// The window queries of a batch with ONE copy, ONE kernel and ONE host wait (window_query_kernel): the strings go to HBM through one pinned
// staging block, the candidate lists come back as the kernel's own stores into pinned memory and are laid out as a CSR here.  `*redo`: a query
// the kernel does not take (see there) -- the caller runs the batch through the multi-pass kernels (filter_strings_device).
int run_window_queries_fast(nsgpu_ctx *c, const char *strs, const uint64_t *qoff, uint32_t nq, const uint64_t *&off_out, const uint32_t *&ids_out, bool *redo)
{
    *redo = false;
    off_out = nullptr, ids_out = nullptr;
    const uint32_t n = c->prm.n, N = c->reads.n;
    const uint32_t thr1 = c->prm.overlap_sketch_thr ? c->prm.overlap_sketch_thr : 1u;
    NS_CHECK(nq > 0, NSGPU_ERR_ARG, "run_window_queries_fast: no queries");
    if (n > 256 || c->prm.k < 1 || c->prm.k > 31 || nq > (1u << 20)) { *redo = true; return NSGPU_OK; }
    const uint64_t total = qoff[nq] - qoff[0];
    // one pinned staging block: ascii | aoff
    const size_t o_aoff = (total + 15) & ~(size_t)15, stage_bytes = o_aoff + ((size_t)nq + 1) * 8 + 16;
    NS_TRY(c->pin_wq.reserve(stage_bytes));
    uint8_t *h = c->pin_wq.as<uint8_t>();
    uint64_t *h_aoff = reinterpret_cast<uint64_t *>(h + o_aoff);
    memcpy(h, strs + qoff[0], total);
    for (uint32_t q = 0; q <= nq; ++q) {
        h_aoff[q] = qoff[q] - qoff[0];
        if (q) NS_CHECK(qoff[q] >= qoff[q - 1] && qoff[q] - qoff[q - 1] <= 0xFFFFFFF0ull, NSGPU_ERR_RANGE, "window %u longer than 2^32-16 bases", q - 1);
    }
    // (the kernel reads the windows where they lie in pinned memory -- every byte once, coalesced: 20 KB per launch -- instead of behind a copy
    // kernel of their own: 6-8 us and a launch gap in front of every window launch, the kernel trace)
    // the answers: cnt[nq] | flags[4] | ids[nq][WQ_SLOT], written by the kernel
    const size_t o_flags = (size_t)nq * 4, o_ids = (o_flags + 16 + 15) & ~(size_t)15;
    NS_TRY(c->pin_wq_out.reserve(o_ids + (size_t)nq * WQ_SLOT * 4 + 64));
    uint32_t *h_cnt = c->pin_wq_out.as<uint32_t>();
    uint32_t *h_flags = reinterpret_cast<uint32_t *>(c->pin_wq_out.as<uint8_t>() + o_flags);
    uint32_t *h_ids = reinterpret_cast<uint32_t *>(c->pin_wq_out.as<uint8_t>() + o_ids);
    h_flags[0] = 0;
    hipLaunchKernelGGL(window_query_kernel, dim3(nq), dim3(256), 0, c->stream, h, h_aoff, nq,
                       c->prm.k, n, thr1, N, c->salts.as<uint64_t>(), c->idx_keys.as<uint64_t>(), c->idx_ids.as<uint32_t>(), h_cnt, h_ids, h_flags);
    NS_HIP(hipGetLastError());
    NS_HIP(stream_wait_short(c->stream));
    c->have_filter_all = false;
    if (h_flags[0]) { *redo = true; return NSGPU_OK; }
    c->wq_off.resize((size_t)nq + 1);
    uint64_t tot = 0;
    for (uint32_t q = 0; q < nq; ++q) { c->wq_off[q] = tot; tot += h_cnt[q]; }
    c->wq_off[nq] = tot;
    c->wq_ids.resize(tot + 1);
    for (uint32_t q = 0; q < nq; ++q) if (h_cnt[q]) memcpy(c->wq_ids.data() + c->wq_off[q], h_ids + (size_t)q * WQ_SLOT, (size_t)h_cnt[q] * 4);
    c->f_total = tot;
    c->f_nq = nq;
    off_out = c->wq_off.data(), ids_out = c->wq_ids.data();
    return NSGPU_OK;
}

}  // namespace nsgpu
