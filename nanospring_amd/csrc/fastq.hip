// fastq.hip -- SURVEY 8(f3): FASTQ text -> 2-bit rows in HBM, the step in front of the hot path
// (ReadData::loadFromFastqFile, src/ReadData.cpp:78-221; DnaBitset, src/dnaToBits.cpp:10-36).
//
// The reference reads the file with std::getline, serially: line 4r is the name, line 4r+1 the bases (the WHOLE line:
// a '\r' of a CRLF file is a base like any other byte and folds through baseToInt), lines 4r+2 and 4r+3 are skipped.
// A record whose base line is missing (file ends after a name line) is a read of length 0; an unterminated last line
// counts when it is not empty.  Here the text goes to HBM once and is parsed there: newline counts per 1 KiB
// segment (one 16-byte load per lane), a prefix sum, newline positions, then (start, length) of every base line; the existing pack kernel reads the
// bases straight out of the text (no intermediate copy).  The host keeps its folded mirror of the reads (the contig
// stage walks reads on the host) from the same (start, length) table.
#include "common.hpp"
#include "host_util.hpp"
#include <rocprim/rocprim.hpp>
#include <cstring>

namespace nsgpu {

int store_prepare_lens(nsgpu_ctx *c, SeqStore &st, const uint32_t *len, uint32_t n);     // api.hip

namespace {

constexpr uint32_t kSeg = 1024;    // bytes of text per wave in the counting / position passes: one 16-byte load per lane

// flags (bit 7 of each byte) of the bytes of w that equal '\n'
__device__ __forceinline__ uint32_t nl_flags(uint32_t w)
{
    const uint32_t x = w ^ 0x0A0A0A0Au;                                   // '\n' bytes become zero
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;       // exact zero-byte test (no borrow between bytes)
}

__device__ __forceinline__ uint4 load16(const uint8_t *__restrict__ text, uint64_t n, uint64_t b)
{
    if (b + 16 <= n) return *reinterpret_cast<const uint4 *>(text + b);
    uint32_t w[4] = {0, 0, 0, 0};                                         // the ragged tail: missing bytes read as 0 (not a newline)
    for (uint64_t i = b; i < n; ++i) w[(i - b) >> 2] |= (uint32_t)text[i] << (8 * ((i - b) & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(256) void fq_count_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t n_seg, uint32_t *__restrict__ cnt)
{
    const uint32_t sg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sg > n_seg) return;
    if (sg == n_seg) { if (lane == 0) cnt[sg] = 0; return; }
    const uint64_t b = (uint64_t)sg * kSeg + lane * 16;
    uint32_t k = 0;
    if (b < n) {
        const uint4 v = load16(text, n, b);
        k = __popc(nl_flags(v.x)) + __popc(nl_flags(v.y)) + __popc(nl_flags(v.z)) + __popc(nl_flags(v.w));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o, 64);
    if (lane == 0) cnt[sg] = k;
}

__global__ __launch_bounds__(256) void fq_positions_kernel(const uint8_t *__restrict__ text, uint64_t n, uint32_t n_seg, const uint32_t *__restrict__ base,
                                                           uint32_t *__restrict__ nlpos)
{
    const uint32_t sg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sg >= n_seg) return;
    const uint64_t b = (uint64_t)sg * kSeg + lane * 16;
    uint32_t f[4] = {0, 0, 0, 0}, k = 0;
    if (b < n) {
        const uint4 v = load16(text, n, b);
        f[0] = nl_flags(v.x), f[1] = nl_flags(v.y), f[2] = nl_flags(v.z), f[3] = nl_flags(v.w);
        k = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
    }
    uint32_t incl = k;                                                    // inclusive prefix over the lanes of the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += t; }
    if (!k) return;
    uint32_t at = base[sg] + incl - k;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        for (uint32_t m = f[w]; m; m &= m - 1) nlpos[at++] = (uint32_t)(b + w * 4 + ((__ffs(m) - 1) >> 3));
}

// read r = line 4r + 1; line k starts behind newline k - 1 and ends at newline k (or at the end of the text)
__global__ __launch_bounds__(256) void fq_records_kernel(const uint32_t *__restrict__ nlpos, uint32_t n_nl, uint64_t n_lines, uint64_t n_bytes, uint32_t n_reads,
                                                         uint64_t *__restrict__ start, uint32_t *__restrict__ len)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t li = 4ull * r + 1;
    uint64_t s = 0, e = 0;
    if (li < n_lines) {
        s = (uint64_t)nlpos[li - 1] + 1;
        e = li < n_nl ? nlpos[li] : n_bytes;
    }
    start[r] = s;
    len[r] = (uint32_t)(e - s);
}

}  // namespace

int load_fastq(nsgpu_ctx *c, const char *text, size_t n_bytes, uint32_t *n_reads_out)
{
    NS_CHECK(n_bytes > 0, NSGPU_ERR_ARG, "nsgpu_load_fastq: empty input (the reference asserts numReads != 0, src/ReadData.cpp:141)");
    NS_CHECK(n_bytes < 0xFFFFFF00ull, NSGPU_ERR_RANGE, "nsgpu_load_fastq: at most 4 GiB of text per call");
    const double t0 = now_ms();
    const hipStream_t st = c->stream;
    const uint32_t n_seg = (uint32_t)((n_bytes + kSeg - 1) / kSeg);
    DevBuf &dtext = c->ascii;                     // the text stays resident until the rows are packed
    NS_TRY(dtext.reserve(n_bytes + 64));
    NS_TRY(c->fq_cnt.reserve(((size_t)n_seg + 2) * 4));
    NS_TRY(c->fq_base.reserve(((size_t)n_seg + 2) * 4));
    NS_HIP(hipMemcpyAsync(dtext.p, text, n_bytes, hipMemcpyHostToDevice, st));
    NS_HIP(hipEventRecord(c->t_kernel.a, st));
    hipLaunchKernelGGL(fq_count_kernel, dim3((n_seg + 1 + 3) / 4), dim3(256), 0, st, dtext.as<uint8_t>(), (uint64_t)n_bytes, n_seg, c->fq_cnt.as<uint32_t>());
    NS_HIP(hipGetLastError());
    {
        size_t ws = 0;
        NS_HIP(rocprim::exclusive_scan(nullptr, ws, c->fq_cnt.as<uint32_t>(), c->fq_base.as<uint32_t>(), 0u, (size_t)n_seg + 1, rocprim::plus<uint32_t>(), st));
        NS_TRY(c->f_scan_ws.reserve(ws + 16));
        NS_HIP(rocprim::exclusive_scan(c->f_scan_ws.p, ws, c->fq_cnt.as<uint32_t>(), c->fq_base.as<uint32_t>(), 0u, (size_t)n_seg + 1, rocprim::plus<uint32_t>(), st));
    }
    uint32_t n_nl = 0;
    NS_HIP(hipMemcpyAsync(&n_nl, c->fq_base.as<uint32_t>() + n_seg, 4, hipMemcpyDeviceToHost, st));
    NS_HIP(stream_wait(st));
    const uint64_t n_lines = (uint64_t)n_nl + (text[n_bytes - 1] != '\n' ? 1 : 0);      // getline: an unterminated last line counts when non-empty
    const uint64_t n_reads64 = (n_lines + 3) / 4;
    NS_CHECK(n_reads64 > 0, NSGPU_ERR_ARG, "nsgpu_load_fastq: no reads");
    NS_CHECK(n_reads64 < 0xFFFFFFFFull, NSGPU_ERR_RANGE, "Too many reads for read_t type to handle.");          // src/ReadData.cpp:122-124
    const uint32_t n_reads = (uint32_t)n_reads64;
    NS_TRY(c->fq_nlpos.reserve(((size_t)n_nl + 2) * 4));
    NS_TRY(c->aoff.reserve(((size_t)n_reads + 1) * 8));
    NS_TRY(c->fq_len.reserve(((size_t)n_reads + 1) * 4));
    hipLaunchKernelGGL(fq_positions_kernel, dim3((n_seg + 3) / 4), dim3(256), 0, st, dtext.as<uint8_t>(), (uint64_t)n_bytes, n_seg, c->fq_base.as<uint32_t>(),
                       c->fq_nlpos.as<uint32_t>());
    hipLaunchKernelGGL(fq_records_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, st, c->fq_nlpos.as<uint32_t>(), n_nl, n_lines, (uint64_t)n_bytes, n_reads,
                       c->aoff.as<uint64_t>(), c->fq_len.as<uint32_t>());
    NS_HIP(hipGetLastError());
    std::vector<uint64_t> start(n_reads);
    std::vector<uint32_t> len(n_reads);
    NS_HIP(hipMemcpyAsync(start.data(), c->aoff.p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, st));
    NS_HIP(hipMemcpyAsync(len.data(), c->fq_len.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, st));
    NS_HIP(stream_wait(st));
    // rows: the pack kernel reads read r at text + start[r], len[r] bases
    c->have_sketch = c->have_index = c->have_filter_all = c->have_cons = false;
    NS_TRY(store_prepare_lens(c, c->reads, len.data(), n_reads));
    NS_TRY(launch_pack_ascii(c, dtext.as<char>(), c->aoff.as<uint64_t>(), c->reads));
    NS_HIP(hipEventRecord(c->t_kernel.b, st));
    // host mirror, folded exactly like DnaBitset (src/dnaToBits.cpp:6-8, 81-98), while the pack kernel runs
    static const char dna[4] = {'A', 'T', 'C', 'G'};
    c->h_off.resize((size_t)n_reads + 1);
    uint64_t tot = 0;
    for (uint32_t r = 0; r < n_reads; ++r) { c->h_off[r] = tot; tot += len[r]; }
    c->h_off[n_reads] = tot;
    c->h_bases.resize(tot + 1);
    par_for(n_reads, [&](size_t r) {
        const char *src = text + start[r];
        char *dst = c->h_bases.data() + c->h_off[r];
        for (uint32_t i = 0; i < len[r]; ++i) dst[i] = dna[(src[i] & 2) | ((src[i] & 4) >> 2)];
    });
    NS_HIP(stream_wait(st));
    NS_HIP(hipEventElapsedTime(&c->timing.pack_ms, c->t_kernel.a, c->t_kernel.b));
    c->fastq_ms = now_ms() - t0;
    if (n_reads_out) *n_reads_out = n_reads;
    return NSGPU_OK;
}

}  // namespace nsgpu

using namespace nsgpu;

extern "C" int nsgpu_load_fastq(nsgpu_ctx *c, const char *text, size_t n_bytes, uint32_t *n_reads_out)
{
    NS_CHECK(c && text, NSGPU_ERR_ARG, "nsgpu_load_fastq: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    return load_fastq(c, text, n_bytes, n_reads_out);
}
