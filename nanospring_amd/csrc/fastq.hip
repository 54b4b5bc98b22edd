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
#include <map>
#include <mutex>
#include <string>

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

// mirror: build the host copy of the reads (false: the chunked ingest, which wants the 2-bit rows only once its input is large)
int load_fastq(nsgpu_ctx *c, const char *text, size_t n_bytes, uint32_t *n_reads_out, bool mirror = true)
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
    c->h_bases.clear();
    if (mirror) {
        c->h_bases.resize(tot + 1);
        par_for(n_reads, [&](size_t r) {
            const char *src = text + start[r];
            char *dst = c->h_bases.data() + c->h_off[r];
            for (uint32_t i = 0; i < len[r]; ++i) dst[i] = dna[(src[i] & 2) | ((src[i] & 4) >> 2)];
        });
    }
    NS_HIP(stream_wait(st));
    NS_HIP(hipEventElapsedTime(&c->timing.pack_ms, c->t_kernel.a, c->t_kernel.b));
    if (mirror) NS_TRY(mirror_finalize(c));
    c->fastq_ms = now_ms() - t0;
    if (n_reads_out) *n_reads_out = n_reads;
    return NSGPU_OK;
}

// ---- chunked ingest: inputs beyond 4 GiB (the reference's logs are 85-130 Gbases), records may straddle chunk borders -------------------
// State of an ingest in progress.  Every chunk is parsed on the GPU like a small file of its own after the unfinished tail of the
// previous one (everything behind the last complete 4-line record) has been put in front of it; its reads are packed into a scratch
// store and appended to the accumulated rows in HBM (row offsets are relative, so appending is one device copy).
struct FastqIngest {
    std::string carry;                  // bytes behind the last complete record
    DevBuf rows;                        // accumulated 2-bit rows
    uint64_t row_bytes = 0;
    std::vector<uint64_t> poff;         // row offsets of the accumulated reads
    std::vector<uint32_t> len;
    std::vector<char> h_bases;          // folded host mirror (until the input is large enough for the packed mirror: keep_ascii)
    std::vector<uint64_t> h_off{0};
    bool keep_ascii = true;
    bool active = false;
};

static FastqIngest &ingest_of(nsgpu_ctx *c)
{
    static std::mutex m;
    static std::map<nsgpu_ctx *, FastqIngest> table;        // a context ingests one file at a time; state dies with ..._end
    std::lock_guard<std::mutex> lk(m);
    return table[c];
}

// reads of `text` (a whole number of records unless `last`): parsed + packed into c->queries, appended to the ingest
static int ingest_text(nsgpu_ctx *c, FastqIngest &I, const char *text, size_t n_bytes)
{
    if (n_bytes == 0) return NSGPU_OK;
    SeqStore keep;
    std::swap(keep, c->reads);                                    // load_fastq fills c->reads: borrow it
    std::vector<char> hb;
    std::vector<uint64_t> ho;
    hb.swap(c->h_bases), ho.swap(c->h_off);
    uint32_t n = 0;
    // the ASCII text of the reads is kept only while the packed mirror is not (going to be) in use: NSGPU_PACKED_MIRROR, or the
    // accumulated input past the automatic threshold (api.hip mirror_finalize)
    static const char *pm = getenv("NSGPU_PACKED_MIRROR");
    if (pm ? atoi(pm) != 0 : I.h_off.back() >= (16ull << 30)) { I.keep_ascii = false; I.h_bases.clear(); I.h_bases.shrink_to_fit(); }
    int rc = load_fastq(c, text, n_bytes, &n, I.keep_ascii);
    if (rc == NSGPU_OK) {
        SeqStore &S = c->reads;
        // grow-with-copy of the accumulated rows
        if (I.row_bytes + S.packed_bytes + 64 > I.rows.cap) {
            DevBuf bigger;
            rc = bigger.reserve((I.row_bytes + S.packed_bytes) * 2 + (64 << 20));
            if (rc == NSGPU_OK && I.row_bytes) { if (hipMemcpyAsync(bigger.p, I.rows.p, I.row_bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = NSGPU_ERR_HIP; }
            if (rc == NSGPU_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = NSGPU_ERR_HIP;
            if (rc == NSGPU_OK) { std::swap(bigger.p, I.rows.p); std::swap(bigger.cap, I.rows.cap); }
            bigger.release();
        }
        if (rc == NSGPU_OK && S.packed_bytes) {
            if (hipMemcpyAsync(static_cast<uint8_t *>(I.rows.p) + I.row_bytes, S.packed.p, S.packed_bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rc = NSGPU_ERR_HIP;
            if (rc == NSGPU_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = NSGPU_ERR_HIP;
        }
        if (rc == NSGPU_OK) {
            for (uint32_t r = 0; r < n; ++r) I.poff.push_back(I.row_bytes + S.h_poff[r]);
            I.len.insert(I.len.end(), S.h_len.begin(), S.h_len.end());
            I.row_bytes += S.packed_bytes;
            const uint64_t base = I.h_off.back();
            if (I.keep_ascii) I.h_bases.insert(I.h_bases.end(), c->h_bases.begin(), c->h_bases.begin() + (ptrdiff_t)c->h_off[n]);
            for (uint32_t r = 1; r <= n; ++r) I.h_off.push_back(base + c->h_off[r]);
        }
    }
    std::swap(keep, c->reads);
    keep.release();
    hb.swap(c->h_bases), ho.swap(c->h_off);
    return rc;
}

}  // namespace nsgpu

using namespace nsgpu;

extern "C" int nsgpu_load_fastq_begin(nsgpu_ctx *c)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "nsgpu_load_fastq_begin: null context");
    FastqIngest &I = ingest_of(c);
    I = FastqIngest();
    I.active = true;
    return NSGPU_OK;
}

extern "C" int nsgpu_load_fastq_chunk(nsgpu_ctx *c, const char *text, size_t n_bytes)
{
    NS_CHECK(c && (text || n_bytes == 0), NSGPU_ERR_ARG, "nsgpu_load_fastq_chunk: null argument");
    FastqIngest &I = ingest_of(c);
    NS_CHECK(I.active, NSGPU_ERR_ARG, "nsgpu_load_fastq_chunk: call nsgpu_load_fastq_begin first");
    NS_CHECK(n_bytes < 0xF0000000ull, NSGPU_ERR_RANGE, "nsgpu_load_fastq_chunk: at most 3.75 GiB per chunk");
    NS_HIP(hipSetDevice(c->prm.device));
    // complete records of carry + chunk: everything up to the newline that ends line 4m
    std::string &buf = I.carry;
    buf.append(text, n_bytes);
    uint64_t nl = 0;
    size_t cut = 0;                                   // one past the newline ending the last complete record
    for (const char *p = buf.data(), *e = p + buf.size(); (p = static_cast<const char *>(memchr(p, '\n', (size_t)(e - p)))) != nullptr; ++p)
        if ((++nl & 3) == 0) cut = (size_t)(p - buf.data()) + 1;
    if (cut == 0) return NSGPU_OK;
    NS_TRY(ingest_text(c, I, buf.data(), cut));
    buf.erase(0, cut);
    return NSGPU_OK;
}

extern "C" int nsgpu_load_fastq_end(nsgpu_ctx *c, uint32_t *n_reads_out)
{
    NS_CHECK(c, NSGPU_ERR_ARG, "nsgpu_load_fastq_end: null context");
    FastqIngest &I = ingest_of(c);
    NS_CHECK(I.active, NSGPU_ERR_ARG, "nsgpu_load_fastq_end: no ingest in progress");
    NS_HIP(hipSetDevice(c->prm.device));
    // the tail: an incomplete last record, parsed with getline's end-of-file rules (a missing base line is an empty read; an
    // unterminated last line counts when it is not empty)
    if (!I.carry.empty()) NS_TRY(ingest_text(c, I, I.carry.data(), I.carry.size()));
    const size_t n = I.len.size();
    NS_CHECK(n > 0, NSGPU_ERR_ARG, "nsgpu_load_fastq: no reads");
    NS_CHECK(n < 0xFFFFFFFFull, NSGPU_ERR_RANGE, "Too many reads for read_t type to handle.");
    SeqStore &S = c->reads;
    S.release();
    S.n = (uint32_t)n;
    S.h_len = I.len;
    S.h_poff = I.poff;
    S.h_poff.push_back(I.row_bytes);
    S.packed_bytes = I.row_bytes;
    S.n_bases = I.h_off.back();
    S.max_len = 0;
    for (uint32_t l : I.len) S.max_len = std::max(S.max_len, l);
    std::swap(S.packed.p, I.rows.p), std::swap(S.packed.cap, I.rows.cap);
    NS_TRY(S.poff.reserve((n + 1) * 8));
    NS_TRY(S.len.reserve((n + 1) * 4));
    NS_HIP(hipMemcpyAsync(S.poff.p, S.h_poff.data(), (n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    NS_HIP(hipMemcpyAsync(S.len.p, S.h_len.data(), n * 4, hipMemcpyHostToDevice, c->stream));
    NS_HIP(hipStreamSynchronize(c->stream));
    c->h_bases.swap(I.h_bases);
    c->h_bases.push_back(0);
    c->h_off.swap(I.h_off);
    c->have_sketch = c->have_index = c->have_filter_all = c->have_cons = false;
    const bool force_packed = !I.keep_ascii;
    I = FastqIngest();
    if (n_reads_out) *n_reads_out = (uint32_t)n;
    return mirror_finalize(c, force_packed);
}

extern "C" int nsgpu_load_fastq(nsgpu_ctx *c, const char *text, size_t n_bytes, uint32_t *n_reads_out)
{
    NS_CHECK(c && text, NSGPU_ERR_ARG, "nsgpu_load_fastq: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    return load_fastq(c, text, n_bytes, n_reads_out);
}

// ---- ReadData::loadFromFile(fileName, FASTQ, gzip_flag) (src/ReadData.cpp:12-26, 78-101, 156-171): the file itself, plain or gzip ----
// The reference puts boost::iostreams::gzip_decompressor in front of the same getline loop; here zlib inflates piece by piece on a
// helper thread while the GPU parses and packs the piece before (nsgpu_load_fastq_chunk).  Concatenated gzip members are read
// through like `zcat` does (what the reference's test script compares with, util/test_script.sh:9).
#include <zlib.h>
#include <thread>

namespace {
struct FileSource {
    FILE *f = nullptr;
    bool gz = false, z_open = false, eof = false;
    z_stream zs;
    std::vector<unsigned char> in;
    std::string err;
    ~FileSource() { if (z_open) inflateEnd(&zs); if (f) fclose(f); }
    // fills out[0 .. cap) as far as the file goes; returns the bytes written (0 = end of input), or (size_t)-1 on error
    size_t read(char *out, size_t cap)
    {
        if (!gz) { const size_t n = fread(out, 1, cap, f); if (n < cap && ferror(f)) { err = "read error"; return (size_t)-1; } return n; }
        size_t done = 0;
        while (done < cap && !eof) {
            if (zs.avail_in == 0) {
                zs.next_in = in.data();
                zs.avail_in = (uInt)fread(in.data(), 1, in.size(), f);
                if (zs.avail_in == 0) {
                    if (ferror(f)) { err = "read error"; return (size_t)-1; }
                    if (z_mid) { err = "unexpected end of the gzip stream"; return (size_t)-1; }
                    eof = true;
                    break;
                }
            }
            const size_t room = std::min<size_t>(cap - done, 1u << 30);
            zs.next_out = reinterpret_cast<Bytef *>(out + done);
            zs.avail_out = (uInt)room;
            const int r = inflate(&zs, Z_NO_FLUSH);
            done += room - zs.avail_out;
            z_mid = true;
            if (r == Z_STREAM_END) {                       // next member, if any
                z_mid = false;
                if (inflateReset(&zs) != Z_OK) { err = "inflateReset failed"; return (size_t)-1; }
            } else if (r != Z_OK && r != Z_BUF_ERROR) { err = std::string("gzip data error: ") + (zs.msg ? zs.msg : "?"); return (size_t)-1; }
        }
        return done;
    }
    bool z_mid = false;
};
}  // namespace

extern "C" int nsgpu_load_fastq_file(nsgpu_ctx *c, const char *path, int gzip_flag, uint32_t *n_reads_out)
{
    NS_CHECK(c && path, NSGPU_ERR_ARG, "nsgpu_load_fastq_file: null argument");
    FileSource src;
    src.f = fopen(path, "rb");
    NS_CHECK(src.f, NSGPU_ERR_ARG, "Can't open input file: %s", path);
    if (gzip_flag < 0) {                                   // by content: the two magic bytes of a gzip member
        unsigned char m[2] = {0, 0};
        const size_t n = fread(m, 1, 2, src.f);
        gzip_flag = n == 2 && m[0] == 0x1f && m[1] == 0x8b;
        rewind(src.f);
    }
    src.gz = gzip_flag != 0;
    if (src.gz) {
        memset(&src.zs, 0, sizeof(src.zs));
        NS_CHECK(inflateInit2(&src.zs, 15 + 16) == Z_OK, NSGPU_ERR_NOMEM, "inflateInit2 failed");
        src.z_open = true;
        src.in.resize(4u << 20);
    }
    static const size_t piece = [] { const char *e = getenv("NSGPU_FASTQ_PIECE_MB"); const size_t mb = e ? strtoull(e, nullptr, 10) : 0; return (mb ? mb : 512) << 20; }();
    std::vector<char> buf[2];
    buf[0].resize(piece), buf[1].resize(piece);
    NS_TRY(nsgpu_load_fastq_begin(c));
    size_t got = src.read(buf[0].data(), piece);
    int rc = NSGPU_OK;
    for (int cur = 0; got != 0 && got != (size_t)-1; cur ^= 1) {
        size_t next = 0;
        std::thread t([&] { next = src.read(buf[cur ^ 1].data(), piece); });       // inflate the next piece while the GPU takes this one
        rc = nsgpu_load_fastq_chunk(c, buf[cur].data(), got);
        t.join();
        if (rc != NSGPU_OK) break;
        got = next;
    }
    if (rc == NSGPU_OK && got == (size_t)-1) { set_error("%s: %s", path, src.err.c_str()); rc = NSGPU_ERR_ARG; }
    if (rc != NSGPU_OK) { ingest_of(c) = FastqIngest(); return rc; }
    return nsgpu_load_fastq_end(c, n_reads_out);
}
