// mm_sketch.hip -- minimizer sketches (mm_sketch, minimap2/sketch.c:77-143) of a batch of sequences on the GPU.
//
// The contig engine needs, per round, the (w,k)-minimizers of every changed consensus (the single-sequence index of
// ConsensusGraph::alignRead, src/ConsensusGraph.cpp:186-198) and of every candidate read (mm_map's query sketch).
// mm_sketch is written as a sequential state machine (ring buffer of the last w k-mer hashes, a running minimum, tie
// and palindrome rules); run lane-per-sequence it is latency-bound (measured: 75 ms per batch).  Here it is restated
// as data-parallel passes over the concatenated batch -- every statement of the state machine is a function of
//   * the k-mer ending at a position (over the last k VALID bases: invalid bytes are skipped, not shifted in),
//   * run = valid non-palindromic bases since the last invalid byte (two prefix sums and a running maximum),
//   * the push sequence (what the reference writes into its ring: every position except palindromic k-mers), and
//   * RM(p) = the right-most minimum of the last w pushes -- which is what the reference's `min` holds after push p:
//     `<=` lets a newer equal hash win, and the ring re-scan picks the right-most smallest entry.
// Outputs are then a per-push event list (old minimum replaced / slid out, ties at the first full window and after a
// re-scan), counted, prefix-summed and written in order.  Pinned bit for bit against the reference's own object code
// (tests/golden/mm_sketch_cases.npz, tests/test_mm_sketch_gpu.py).
#include "common.hpp"
#include "mm2.hpp"
#include "host_util.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cstring>

namespace nsgpu {

namespace {

constexpr uint64_t kU64Max = ~0ull;

__device__ __forceinline__ uint64_t hash64_masked_dev(uint64_t key, uint64_t mask)
{
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// seq_nt4_table restricted to what it distinguishes: ACGTU in either case and the raw codes 0..3
__device__ __forceinline__ int nt4_dev(uint8_t ch)
{
    if (ch < 4) return ch;
    switch (ch | 0x20) {
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': case 'u': return 3;
    }
    return 4;
}

// Layout: sequence s occupies positions soff[s] .. soff[s] + len[s] of the concatenated batch, followed by at least
// one padding position up to the next multiple of 16 (a padding position is "invalid" for the run count, but pushes
// nothing); soff[n] = B.
struct Batch {
    const uint8_t *seqs; const uint32_t *soff; const uint32_t *len; uint32_t n, B;
    int w, k;
};

__global__ __launch_bounds__(256) void sk_block_owner_kernel(Batch b, uint32_t *__restrict__ sob)
{
    const uint32_t blk = blockIdx.x * 256 + threadIdx.x;
    if (blk >= b.B / 16) return;
    const uint32_t pos = blk * 16;
    uint32_t lo = 0, hi = b.n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (b.soff[mid] <= pos) lo = mid; else hi = mid; }
    sob[blk] = lo;
}

// pass 1: valid flags and the "last invalid" markers
__global__ __launch_bounds__(256) void sk_flags_kernel(Batch b, const uint32_t *__restrict__ sob, uint32_t *__restrict__ vf, uint32_t *__restrict__ mk)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > b.B) return;
    if (i == b.B) { vf[i] = 0; mk[i] = 0; return; }
    const uint32_t s = sob[i >> 4];
    const bool inlen = i - b.soff[s] < b.len[s];
    const bool valid = inlen && nt4_dev(b.seqs[i]) < 4;
    vf[i] = valid;
    mk[i] = valid ? 0u : i + 1;
}

// pass 2: the valid bases, compacted
__global__ __launch_bounds__(256) void sk_compact_kernel(Batch b, const uint32_t *__restrict__ vf, const uint32_t *__restrict__ vr, uint8_t *__restrict__ V)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= b.B || !vf[i]) return;
    V[vr[i] - 1] = (uint8_t)nt4_dev(b.seqs[i]);          // vr = INCLUSIVE count of valid positions (one scan with the last-invalid maximum)
}

// pass 3: k-mer over the last k valid bases of the sequence (fewer at its start: the reference starts from fw = rv = 0),
// palindrome flag, strand and hash; flags "non-palindromic valid" and "pushes"
__global__ __launch_bounds__(256) void sk_kmer_kernel(Batch b, const uint32_t *__restrict__ sob, const uint32_t *__restrict__ vf, const uint32_t *__restrict__ vr,
                                                      const uint8_t *__restrict__ V, uint64_t *__restrict__ hk, uint32_t *__restrict__ npf, uint32_t *__restrict__ pushf)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > b.B) return;
    if (i == b.B) { npf[i] = 0; pushf[i] = 0; return; }
    const uint32_t s = sob[i >> 4];
    const bool inlen = i - b.soff[s] < b.len[s];
    if (!vf[i]) { npf[i] = 0; pushf[i] = inlen; return; }
    const uint32_t s0 = b.soff[s];
    const uint32_t r = vr[i] - 1, lr = r - (vr[s0] - vf[s0]);      // exclusive ranks out of the inclusive scan
    const int k = b.k;
    const uint64_t shift1 = 2 * (uint64_t)(k - 1);
    const int m = lr < (uint32_t)(k - 1) ? (int)lr : k - 1;
    uint64_t fw = 0, rv = 0;
    for (int t = 0; t <= m; ++t) {
        const uint64_t c = V[r - t];
        fw |= c << (2 * t);
        rv |= (3ull ^ c) << (shift1 - 2 * t);
    }
    const bool pal = fw == rv;
    const uint64_t strand = fw < rv ? 0 : 1, mask = (1ull << 2 * k) - 1;
    hk[i] = hash64_masked_dev(strand ? rv : fw, mask) << 8 | strand;
    npf[i] = !pal;
    pushf[i] = !pal;
}

// pass 4: the push sequence: (x, y) the reference writes into its ring, and `run` at that moment
__global__ __launch_bounds__(256) void sk_push_kernel(Batch b, const uint32_t *__restrict__ sob, const uint32_t *__restrict__ vf, const uint32_t *__restrict__ pushf,
                                                      const uint32_t *__restrict__ pr, const uint32_t *__restrict__ npr, const uint32_t *__restrict__ linv,
                                                      const uint64_t *__restrict__ hk, uint64_t *__restrict__ PX, uint64_t *__restrict__ PY, uint32_t *__restrict__ PRUN,
                                                      uint32_t *__restrict__ PSEQ)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= b.B || !pushf[i]) return;
    const uint32_t s = sob[i >> 4], p = pr[i];
    uint64_t cx = kU64Max, cy = kU64Max;
    uint32_t run = 0;
    if (vf[i]) {
        const uint32_t lm = linv[i];                        // position + 1 of the last invalid / padding byte before i (0: none)
        run = npr[i] + 1 - (lm ? npr[lm - 1] : 0u);
        if (run >= (uint32_t)b.k) {
            const uint64_t h = hk[i];
            cx = (h & ~0xffull) | (uint64_t)b.k;            // span = min(run before this base + 1, k) = k here
            cy = (uint64_t)((i - b.soff[s]) << 1) | (h & 1);
        }
    }
    PX[p] = cx; PY[p] = cy; PRUN[p] = run; PSEQ[p] = s;
}

// pass 5: rm[p] = push index of the right-most minimum of the window of push p, (p - w, p] clipped to the sequence.
// A block stages its 256 pushes plus the w - 1 before them in LDS; every thread scans its window there.
__global__ __launch_bounds__(256) void sk_window_min_kernel(Batch b, const uint64_t *__restrict__ PX, const uint32_t *__restrict__ PSEQ, const uint32_t *__restrict__ pr,
                                                            uint32_t P, uint32_t *__restrict__ rm)
{
    __shared__ uint64_t tile[512];
    const uint32_t b0 = blockIdx.x * 256, w = (uint32_t)b.w;
    const uint32_t first = b0 >= w - 1 ? b0 - (w - 1) : 0;            // tile[j] = PX[first + j]
    for (uint32_t j = threadIdx.x; j < 512; j += 256) { const uint32_t q = first + j; tile[j] = q < P && q < b0 + 256 ? PX[q] : kU64Max; }
    __syncthreads();
    const uint32_t p = b0 + threadIdx.x;
    if (p >= P) return;
    const uint32_t p0 = pr[b.soff[PSEQ[p]]];
    const uint32_t lo = p - p0 >= w ? p - w + 1 : p0;
    uint64_t bx = tile[p - first];
    uint32_t bi = p;
    for (uint32_t q = p; q > lo;) { --q; const uint64_t x = tile[q - first]; if (x < bx) bx = x, bi = q; }
    rm[p] = bi;
}

// pass 6/7: the events of push p, in the reference's order.  WRITE = false counts them, WRITE = true writes them at o.
template <bool WRITE>
__device__ __forceinline__ uint32_t sk_events(const Batch &b, const uint64_t *__restrict__ PX, const uint64_t *__restrict__ PY, const uint32_t *__restrict__ PRUN,
                                             const uint32_t *__restrict__ rm, uint32_t p, uint32_t p0, uint32_t p1, uint64_t *__restrict__ o)
{
    const uint32_t w = (uint32_t)b.w, k = (uint32_t)b.k;
    const uint64_t cx = PX[p];
    const uint32_t run = PRUN[p];
    uint32_t cnt = 0;
    auto emit = [&](uint32_t q) { if (WRITE) { o[2 * (size_t)cnt] = PX[q]; o[2 * (size_t)cnt + 1] = PY[q]; } ++cnt; };
    // ties of the minimum (x, at push `self`) among pushes [lo, hi), oldest first
    auto ties = [&](uint64_t x, uint32_t self, uint32_t lo, uint32_t hi_excl) { for (uint32_t q = lo; q < hi_excl; ++q) if (PX[q] == x && q != self) emit(q); };
    const uint32_t lo_cur = p - p0 >= w ? p - w + 1 : p0;                  // window of push p: (p - w, p], clipped to the sequence
    // `min` before this push: right-most minimum of the previous window (MAX before the first push)
    uint64_t prev_x = kU64Max;
    uint32_t prev_i = 0;
    bool prev_at_expired = false;                                          // the reference's slot == min_slot: min sits in the slot being overwritten
    if (p > p0) {
        prev_i = rm[p - 1];
        prev_x = PX[prev_i];
        prev_at_expired = p - p0 >= w && prev_i == p - w;
    }
    if (run == w + k - 1 && prev_x != kU64Max) ties(prev_x, prev_i, lo_cur, p);             // first full window: older entries only
    if (cx <= prev_x) {
        if (run >= w + k && prev_x != kU64Max) emit(prev_i);
    } else if (prev_at_expired) {
        if (run >= w + k - 1 && prev_x != kU64Max) emit(prev_i);
        const uint32_t now_i = rm[p];
        const uint64_t now_x = PX[now_i];
        if (run >= w + k - 1 && now_x != kU64Max) ties(now_x, now_i, lo_cur, p + 1);
    }
    if (p + 1 == p1) { const uint32_t now_i = rm[p]; if (PX[now_i] != kU64Max) emit(now_i); }   // the closing `if (min.x != UINT64_MAX) push(min)`
    return cnt;
}

__global__ __launch_bounds__(256) void sk_count_kernel(Batch b, const uint64_t *__restrict__ PX, const uint64_t *__restrict__ PY, const uint32_t *__restrict__ PRUN,
                                                       const uint32_t *__restrict__ PSEQ, const uint32_t *__restrict__ pr, const uint32_t *__restrict__ rm, uint32_t P,
                                                       uint32_t *__restrict__ nout)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p > P) return;
    if (p == P) { nout[p] = 0; return; }
    const uint32_t s = PSEQ[p];
    nout[p] = sk_events<false>(b, PX, PY, PRUN, rm, p, pr[b.soff[s]], pr[b.soff[s + 1]], nullptr);
}

__global__ __launch_bounds__(256) void sk_write_kernel(Batch b, const uint64_t *__restrict__ PX, const uint64_t *__restrict__ PY, const uint32_t *__restrict__ PRUN,
                                                       const uint32_t *__restrict__ PSEQ, const uint32_t *__restrict__ pr, const uint32_t *__restrict__ rm, uint32_t P,
                                                       const uint32_t *__restrict__ oscan, uint64_t *__restrict__ out)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P || oscan[p + 1] == oscan[p]) return;
    const uint32_t s = PSEQ[p];
    sk_events<true>(b, PX, PY, PRUN, rm, p, pr[b.soff[s]], pr[b.soff[s + 1]], out + 2 * (size_t)oscan[p]);
}

// output offset of every sequence = events before its first push
__global__ __launch_bounds__(256) void sk_offsets_kernel(Batch b, const uint32_t *__restrict__ pr, const uint32_t *__restrict__ oscan, uint64_t *__restrict__ off)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s > b.n) return;
    off[s] = oscan[pr[b.soff[s]]];
}


// ---------------------------------------------------------------------------------------------------------------------
// Fused path.  What the contig engine sketches are consensus strings and reads: pure A/C/G/T.  Then every position is a valid
// base, a k-mer is positional, and the state machine is local: a workgroup takes 1024 consecutive positions of one sequence
// plus a halo, builds k-mers and hashes in LDS, the right-most minima of the windows of the last w PUSHES and the events --
// once to count (skf_kernel<false>), once, after ONE scan over the tile counts, to write (skf_kernel<true>): 6 GPU operations
// and one host synchronisation per call instead of ~25 and three.
// Symmetric k-mers (equal to their reverse complement: 4^-(k/2) per position, i.e. a few per batch, and whole runs of them in
// (AT)n or (ACGT)n repeats) push nothing and do not count in `run`: inside a tile they are marked and skipped exactly; a tile
// whose halo would need more than kPalHalo of them, any byte that is not ACGT, or a sequence so full of them that `run`
// could still be short beyond its first tile (checked by the host from the per-tile counts) sends the batch to the general
// passes above instead.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTile = 1024, kPalHalo = 16;
constexpr uint64_t kPal = ~0ull - 1;            // px marker of a symmetric k-mer (a hash << 8 never gets there)

struct Tile { uint32_t seq, t0; };

// tiles == nullptr: ONE sequence (b.soff[0], b.len[0]), tile i = its positions [1024 i, 1024 (i + 1)).
// tile_bad (optional): per tile, 1 where the kernel gave up on the tile (what raises flags[0] / flags[1] for the whole batch).
// tile_stride != 0 (WRITE only): no offsets from a scan -- tile i writes into its own slot of tile_stride entries and reports its count in
// tile_cnt[i] (and tile_pal[i]): one launch per sequence, for callers that cannot wait between a count and a write pass.
template <bool WRITE>
__global__ __launch_bounds__(256) void skf_kernel(Batch b, const Tile *__restrict__ tiles, uint32_t n_tiles, uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ tile_pal,
                                                  const uint32_t *__restrict__ tile_off, uint64_t *__restrict__ out, uint64_t out_cap,
                                                  uint32_t *__restrict__ flags /* [0] bad input, [1] output overflow */, uint32_t *__restrict__ tile_bad, uint32_t tile_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int w = b.w, k = b.k;
    const int back = w + kPalHalo + 1;                                      // positions before t0 held in px
    const int n_px = kTile + back + kPalHalo, n_rm = kTile + kPalHalo + 1;  // px also looks kPalHalo positions ahead (is a push the sequence's last?); rm / lo for positions t0 - kPalHalo - 1 ..
    uint64_t *px = reinterpret_cast<uint64_t *>(lds);                       // px[j] = position t0 - back + j
    uint32_t *rmv = reinterpret_cast<uint32_t *>(px + n_px + 1);            // right-most minimum of the window ending at the position (a push)
    uint32_t *lov = rmv + n_rm + 1;                                         // oldest push of that window | full << 31
    uint32_t *cnt = lov + n_rm + 1;                                         // per-thread event counts
    uint8_t *code = reinterpret_cast<uint8_t *>(cnt + 260) + 2 * (size_t)(n_px + 2) * 2;     // behind the two uint16 doubling arrays; code[j] = base t0 - back - k + 1 + j
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const uint32_t s = tiles ? tiles[tile].seq : 0u, t0 = tiles ? tiles[tile].t0 : tile * (uint32_t)kTile, len = b.len[s];
    auto give_up = [&](int which) { flags[which] = 1; if (tile_bad) tile_bad[tile] = 1; };
    const uint8_t *seq = b.seqs + b.soff[s];
    const int tid = threadIdx.x;
    const long px0 = (long)t0 - back, base0 = px0 - k + 1;
    const int n_code = n_px + k - 1;
    bool bad = false;
    for (int j = tid; j < n_code; j += 256) {
        const long i = base0 + j;
        uint8_t c = 0;
        if (i >= 0 && i < (long)len) { const int v = nt4_dev(seq[i]); bad |= v > 3; c = (uint8_t)(v & 3); }
        code[j] = c;
    }
    __syncthreads();
    // px: hash << 8 | strand of the k-mer ending at the position; MAX before the first whole k-mer; kPal for symmetric k-mers
    const uint64_t mask = (1ull << 2 * k) - 1, shift1 = 2 * (uint64_t)(k - 1);
    const int per = (n_px + 255) / 256;
    uint32_t my_pal = 0, my_pal_region = 0;
    {
        const int j0 = tid * per;
        uint64_t fw = 0, rv = 0;
        bool have = false;
        for (int jj = 0; jj < per; ++jj) {
            const int j = j0 + jj;
            if (j >= n_px) break;
            const long i = px0 + j;
            uint64_t x = kU64Max;
            if (i >= (long)k - 1 && i < (long)len) {
                const int cj = j + k - 1;                              // index of base i in code[]
                if (!have) {
                    fw = rv = 0;
                    for (int t = 0; t < k; ++t) { const uint64_t c = code[cj - t]; fw |= c << (2 * t); rv |= (3ull ^ c) << (shift1 - 2 * t); }
                    have = true;
                } else {
                    const uint64_t c = code[cj];
                    fw = (fw << 2 | c) & mask;
                    rv = rv >> 2 | (3ull ^ c) << shift1;
                }
                if (fw == rv) { x = kPal; ++my_pal_region; my_pal += i >= (long)t0 && i < (long)t0 + kTile; }
                else { const uint64_t strand = fw < rv ? 0 : 1; x = hash64_masked_dev(strand ? rv : fw, mask) << 8 | strand; }
            } else if (i >= 0 && i < (long)len) {
                // a position before the first whole k-mer: symmetric there means the partial words agree (sketch.c:100 compares them as they are)
                const int cj = j + k - 1;
                uint64_t f2 = 0, r2 = 0;
                for (int t = 0; t <= (int)i; ++t) { const uint64_t c = code[cj - t]; f2 |= c << (2 * t); r2 |= (3ull ^ c) << (shift1 - 2 * t); }
                if (f2 == r2) { x = kPal; ++my_pal_region; my_pal += i >= (long)t0 && i < (long)t0 + kTile; }
            }
            px[j] = x;
        }
    }
    if (bad) give_up(0);
    cnt[tid] = my_pal | my_pal_region << 16;
    __syncthreads();
    uint32_t pal_in_tile = 0, pal_region = 0;
    for (int i = 0; i < 256; ++i) pal_in_tile += cnt[i] & 0xffffu, pal_region += cnt[i] >> 16;
    __syncthreads();
    if (t0 == 0) {
        // `info` stays empty until l (= run) reaches k (sketch.c:104): symmetric k-mers among the first bases delay that
        if (tid == 0) {
            int run = 0;
            for (long i = 0; i < (long)len && i < (long)kTile; ++i) {
                const uint64_t v = px[(int)(i - px0)];
                if (v == kPal) continue;
                if (++run >= k) break;
                px[(int)(i - px0)] = kU64Max;
            }
        }
    }
    __syncthreads();
    auto hashOf = [&](long q) -> uint64_t { const uint64_t v = px[(int)(q - px0)]; return v >= kPal ? v : v >> 8; };      // MAX: a push without k-mer; kPal: no push
    // window of the last w pushes ending at every push position in [t0 - kPalHalo - 1, t0 + kTile)
    if (pal_region == 0) {
        // no symmetric k-mer anywhere in the region (all but a handful of tiles): pushes = positions, and the right-most minimum of a
        // window comes from log2(w) doubling steps -- R_j[p] = right-most minimum of the last 2^j positions -- instead of w compares
        uint16_t *ra = reinterpret_cast<uint16_t *>(cnt + 260), *rb = ra + n_px + 2;            // positions relative to px0
        auto better = [&](uint32_t a, uint32_t bq) { const uint64_t xa = px[a], xb = px[bq]; return (xb >= kPal ? kU64Max : xb >> 8) <= (xa >= kPal ? kU64Max : xa >> 8) ? bq : a; };
        for (int j = tid; j < n_px; j += 256) ra[j] = (uint16_t)j;
        __syncthreads();
        int span = 1;
        while (2 * span <= w) {
            for (int j = tid; j < n_px; j += 256) {
                const long pos = px0 + j;
                rb[j] = (j - span >= 0 && pos - span >= 0) ? (uint16_t)better(ra[j - span], ra[j]) : ra[j];
            }
            __syncthreads();
            uint16_t *t_ = ra; ra = rb; rb = t_;
            span *= 2;
        }
        const int rest = w - span;                 // window = [p - w + 1, p - w + span] u [p - span + 1, p]
        for (int j = tid; j < n_rm; j += 256) {
            const long p = (long)t0 - kPalHalo - 1 + j;
            uint32_t bi = 0, lo = 0;
            if (p >= 0 && p < (long)len) {
                const int jp = (int)(p - px0);
                uint32_t r = ra[jp];
                if (rest > 0 && p - rest >= 0) r = better(ra[jp - rest], r);
                bi = (uint32_t)(px0 + r);
                lo = (uint32_t)(p >= w - 1 ? p - w + 1 : 0) | (p >= w - 1 ? 0x80000000u : 0u);
            }
            rmv[j] = bi, lov[j] = lo;
        }
    } else
    for (int j = tid; j < n_rm; j += 256) {
        const long p = (long)t0 - kPalHalo - 1 + j;
        uint32_t bi = 0, lo = 0;
        if (p >= 0 && p < (long)len && hashOf(p) != kPal) {
            uint64_t bx = hashOf(p);
            bi = (uint32_t)p, lo = (uint32_t)p;
            int got = 1;
            long q = p;
            while (got < w && q > 0) {
                --q;
                if (q < px0) { bad = true; break; }                    // more symmetric k-mers in the halo than it was sized for
                const uint64_t x = hashOf(q);
                if (x == kPal) continue;
                ++got, lo = (uint32_t)q;
                if (x < bx) bx = x, bi = (uint32_t)q;
            }
            lo |= got == w ? 0x80000000u : 0u;
        }
        rmv[j] = bi, lov[j] = lo;
    }
    if (bad) give_up(0);
    __syncthreads();
    const long rm0 = (long)t0 - kPalHalo - 1;
    // pushes before a position and `run` are exact in a sequence's first tile; beyond it they are >= w + k unless the sequence is
    // riddled with symmetric k-mers (the host checks the per-tile counts and redoes such a batch by the general passes)
    auto emit_xy = [&](uint32_t q, uint64_t *o) {
        const uint64_t v = px[(int)((long)q - px0)];
        o[0] = (v & ~0xffull) | (uint64_t)k;
        o[1] = (uint64_t)q << 1 | (v & 1);
    };
    const int ppt = kTile / 256;                                  // 4 consecutive positions per thread: outputs stay in push order
    // non-symmetric positions before the thread's first one, inside this tile (first tile: = pushes before it in the sequence)
    uint32_t my_cnt = 0;
    uint64_t *o = nullptr;
    for (int pass = 0; pass < (WRITE ? 2 : 1); ++pass) {
        if (pass == 1) {
            cnt[tid] = my_cnt;
            __syncthreads();
            uint32_t before = 0;
            for (int t = 0; t < tid; ++t) before += cnt[t];
            if (tile_stride && tid == 0) { uint32_t tot = 0; for (int t = 0; t < 256; ++t) tot += cnt[t]; tile_cnt[tile] = tot; tile_pal[tile] = pal_in_tile; }
            const uint64_t at = tile_stride ? (uint64_t)tile * tile_stride + before : (uint64_t)tile_off[tile] + before;
            if (tile_stride ? before + my_cnt > tile_stride : at + my_cnt > out_cap) { if (my_cnt) give_up(1); return; }
            o = out + 2 * at;
        }
        uint32_t c = 0;
        for (int jj = 0; jj < ppt; ++jj) {
            const uint32_t p = t0 + (uint32_t)(tid * ppt + jj);
            if (p >= len) break;
            const uint64_t cx = hashOf(p);
            if (cx == kPal) continue;
            uint32_t run = 0x7fffffffu;                               // "large": beyond the first tile
            if (t0 == 0 && p + 1 < (uint32_t)(w + k) + pal_in_tile) { run = 0; for (uint32_t q = 0; q <= p; ++q) run += hashOf(q) != kPal; }   // only ~w + k deep matters
            auto emit = [&](uint32_t q) { if (pass == 1) emit_xy(q, o + 2 * (size_t)c); ++c; };
            auto ties = [&](uint64_t x, uint32_t self, uint32_t lo, uint32_t hi_excl) { for (uint32_t q = lo; q < hi_excl; ++q) if (hashOf(q) == x && q != self) emit(q); };
            const uint32_t lw = lov[(long)p - rm0], lo_cur = lw & 0x7fffffffu;
            // the previous push
            long pp = (long)p - 1;
            while (pp >= 0 && pp >= rm0 && hashOf(pp) == kPal) --pp;
            uint64_t prev_x = kU64Max;
            uint32_t prev_i = 0;
            bool prev_at_expired = false;
            if (pp >= 0) {
                if (pp < rm0) { give_up(0); return; }
                prev_i = rmv[pp - rm0];
                prev_x = hashOf(prev_i);
                const uint32_t plo = lov[pp - rm0];
                prev_at_expired = (plo & 0x80000000u) && prev_i == (plo & 0x7fffffffu);      // w pushes before p, and the minimum is the one that expires
            }
            if (run == (uint32_t)(w + k - 1) && prev_x != kU64Max) ties(prev_x, prev_i, lo_cur, p);
            if (cx <= prev_x) {
                if (run >= (uint32_t)(w + k) && prev_x != kU64Max) emit(prev_i);
            } else if (prev_at_expired) {
                if (run >= (uint32_t)(w + k - 1) && prev_x != kU64Max) emit(prev_i);
                const uint32_t now_i = rmv[(long)p - rm0];
                const uint64_t now_x = hashOf(now_i);
                if (run >= (uint32_t)(w + k - 1) && now_x != kU64Max) ties(now_x, now_i, lo_cur, p + 1);
            }
            // the closing `if (min.x != UINT64_MAX) push(min)`: p is the sequence's last push
            bool last = true;
            for (uint32_t q = p + 1; q < len && last; ++q) {
                if (q >= t0 + kTile + kPalHalo) { give_up(0); last = false; break; }      // a longer run of symmetric k-mers than the look-ahead: general passes
                last = hashOf(q) == kPal;
            }
            if (last) { const uint32_t now_i = rmv[(long)p - rm0]; if (hashOf(now_i) != kU64Max) emit(now_i); }
        }
        my_cnt = c;
        if (!WRITE) {
            cnt[tid] = c;
            __syncthreads();
            if (tid == 0) { uint32_t t = 0; for (int i = 0; i < 256; ++i) t += cnt[i]; tile_cnt[tile] = t; }
            __syncthreads();
            cnt[tid] = my_pal;
            __syncthreads();
            if (tid == 0) { uint32_t t = 0; for (int i = 0; i < 256; ++i) t += cnt[i]; tile_pal[tile] = t; }
        }
    }
}

}  // namespace

static size_t skf_lds_bytes(int w, int k)
{
    return (size_t)(kTile + w + 2 * kPalHalo + 3) * 8 + 2 * (size_t)(kTile + kPalHalo + 3) * 4 + 260 * 4 + 4 * (size_t)(kTile + w + 2 * kPalHalo + 4) + (size_t)(kTile + w + 2 * kPalHalo + k + 2) + 64;
}

static int pinned_reserve(uint8_t *&p, size_t &cap, size_t want)
{
    if (cap >= want) return NSGPU_OK;
    if (p) NS_HIP(hipHostFree(p));
    p = nullptr, cap = 0;
    const size_t sz = want * 3 / 2 + 4096;
    NS_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), sz, hipHostMallocDefault));
    cap = sz;
    return NSGPU_OK;
}

template <class Op>
static int scan_u32(nsgpu_ctx::SketchWs &W, hipStream_t st, const uint32_t *in, uint32_t *out, size_t n, bool inclusive_max, Op)
{
    size_t ws = 0;
    if (inclusive_max) {
        NS_HIP(rocprim::inclusive_scan(nullptr, ws, in, out, n, rocprim::maximum<uint32_t>(), st));
        NS_TRY(W.scan_ws.reserve(ws + 16));
        NS_HIP(rocprim::inclusive_scan(W.scan_ws.p, ws, in, out, n, rocprim::maximum<uint32_t>(), st));
    } else {
        NS_HIP(rocprim::exclusive_scan(nullptr, ws, in, out, 0u, n, rocprim::plus<uint32_t>(), st));
        NS_TRY(W.scan_ws.reserve(ws + 16));
        NS_HIP(rocprim::exclusive_scan(W.scan_ws.p, ws, in, out, 0u, n, rocprim::plus<uint32_t>(), st));
    }
    return NSGPU_OK;
}


// debug breakdown (one caller at a time): host staging, up to the push count, up to the offsets, write + read-back; bytes in, minimizers out
double g_sketch_ms[6];


// The fused path (skf_kernel).  Returns 1 when the batch has to be redone by the general passes (a byte other than ACGT, a k-mer equal
// to its reverse complement, or more minimizers than the staging buffer was sized for), 0 when done, < 0 on errors.
static int gpu_mm_sketch_fused(nsgpu_ctx *c, const std::vector<SketchReq> &reqs, int w, int k, const mm2::Anchor *&out, std::vector<uint64_t> &out_off, int ws, size_t n_stage_only)
{
    const size_t n = reqs.size(), n_sk = n - n_stage_only;      // the last n_stage_only requests are staged in HBM (sketch_dev_seq) and not sketched
    const double t0 = now_ms();
    nsgpu_ctx::SketchWs &W = c->sws[ws];
    if (!W.stream) NS_TRY(role_stream_create(&W.stream, "sketch"));
    const hipStream_t st = W.stream;
    uint64_t bytes = 0, n_tiles = 0;
    for (size_t i = 0; i < n; ++i) {
        NS_CHECK(reqs[i].len < (1ull << 31), NSGPU_ERR_RANGE, "sequence %zu longer than 2^31", i);
        bytes += (reqs[i].len + 16) & ~(uint64_t)15;
        if (i < n_sk) n_tiles += (reqs[i].len + kTile - 1) / kTile;
    }
    NS_CHECK(bytes < (1ull << 31), NSGPU_ERR_RANGE, "sketch batch exceeds 2 GiB of sequence; use smaller batches");
    // ONE pinned staging area: [sequences | soff | len | tiles], one H2D copy
    const size_t o_soff = (bytes + 15) & ~(size_t)15, o_len = o_soff + (n + 1) * 4, o_tiles = (o_len + n * 4 + 15) & ~(size_t)15, stage_bytes = o_tiles + n_tiles * sizeof(Tile) + 16;
    NS_TRY(pinned_reserve(W.h_seqs, W.h_cap, stage_bytes));
    uint32_t *soff = reinterpret_cast<uint32_t *>(W.h_seqs + o_soff), *len = reinterpret_cast<uint32_t *>(W.h_seqs + o_len);
    Tile *tiles = reinterpret_cast<Tile *>(W.h_seqs + o_tiles);
    std::vector<uint32_t> first_tile(n + 1);
    {
        uint64_t b = 0, t = 0;
        for (size_t i = 0; i < n; ++i) {
            soff[i] = (uint32_t)b, len[i] = (uint32_t)reqs[i].len, first_tile[i] = (uint32_t)t;
            if (i < n_sk) for (uint64_t p = 0; p < reqs[i].len; p += kTile) tiles[t++] = Tile{(uint32_t)i, (uint32_t)p};
            b += (reqs[i].len + 16) & ~(uint64_t)15;
        }
        soff[n] = (uint32_t)b, first_tile[n] = (uint32_t)t;
    }
    par_for("sketch.stage", n, [&](size_t i) { memcpy(W.h_seqs + soff[i], reqs[i].ptr, reqs[i].len); });
    const double t_staged = now_ms();
    NS_TRY(W.seqs.reserve(stage_bytes + 64));
    NS_HIP(hipMemcpyAsync(W.seqs.p, W.h_seqs, stage_bytes, hipMemcpyHostToDevice, st));
    const Batch bt{W.seqs.as<uint8_t>(), reinterpret_cast<const uint32_t *>(W.seqs.as<uint8_t>() + o_soff), reinterpret_cast<const uint32_t *>(W.seqs.as<uint8_t>() + o_len),
                   (uint32_t)n, (uint32_t)bytes, w, k};
    const Tile *d_tiles = reinterpret_cast<const Tile *>(W.seqs.as<uint8_t>() + o_tiles);
    // ONE launch: every tile writes its minimizers into its own slot of `stride` entries in pinned host memory (the GPU's stores travel over
    // PCIe; no copy operation, no size known to the host in advance) and reports its count; the host lays the slots out back to back below.
    // (Until round 5: a count pass, a scan over the tile counts and a write pass -- four dependent GPU operations more, 0.1 ms per call.)
    const uint32_t stride = (uint32_t)std::min<uint64_t>(1100, std::max<uint64_t>(64, 6ull * kTile / (uint64_t)(w + 1) + 32));
    NS_TRY(W.h_meta.reserve((2 * n_tiles + 8) * 4 + 64));
    NS_TRY(pinned_reserve(W.h_out, W.h_out_cap, (size_t)n_tiles * stride * 16 + 16));
    uint32_t *h_flags = W.h_meta.as<uint32_t>(), *h_tcnt = h_flags + 2, *h_tpal = h_tcnt + n_tiles + 2;
    h_flags[0] = h_flags[1] = 0;
    const size_t lds = skf_lds_bytes(w, k);
    if (n_tiles) {
        hipLaunchKernelGGL((skf_kernel<true>), dim3((uint32_t)n_tiles), dim3(256), lds, st, bt, d_tiles, (uint32_t)n_tiles, h_tcnt, h_tpal, (const uint32_t *)nullptr,
                           reinterpret_cast<uint64_t *>(W.h_out), (uint64_t)n_tiles * stride, h_flags, (uint32_t *)nullptr, stride);
        NS_HIP(hipGetLastError());
    }
    NS_HIP(stream_wait_short(st));
    // `run` beyond a sequence's first tile was taken as >= w + k: true unless symmetric k-mers ate the difference
    bool run_ok = true;
    for (size_t i = 0; i < n && run_ok; ++i) {
        uint64_t pal = 0;
        for (uint32_t t = first_tile[i]; t < first_tile[i + 1]; ++t) {
            if (t > first_tile[i] && (uint64_t)tiles[t].t0 < pal + (uint64_t)(w + k)) { run_ok = false; break; }
            pal += h_tpal[t];
        }
    }
    if (h_flags[0] || h_flags[1] || !run_ok) {
        static const bool dbg = getenv("NSGPU_CONS_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[sketch] fused path gives up: bad input %u, a tile with more minimizers than its slot %u, %zu sequences, %llu tiles\n", h_flags[0], h_flags[1],
                         n, (unsigned long long)n_tiles);
        return 1;
    }
    // the slots back to back, in pinned memory as well (the seeding kernel reads the lists where they are left)
    std::vector<uint64_t> &toff = W.h_toff;
    toff.resize(n_tiles + 1);
    uint64_t total = 0;
    for (uint64_t t = 0; t < n_tiles; ++t) { toff[t] = total; total += h_tcnt[t]; }
    toff[n_tiles] = total;
    NS_TRY(W.h_compact.reserve((total + 1) * 16));
    {
        const mm2::Anchor *slots = reinterpret_cast<const mm2::Anchor *>(W.h_out);
        mm2::Anchor *dst = W.h_compact.as<mm2::Anchor>();
        if (n_tiles >= 512) par_for("sketch.stage", (size_t)((n_tiles + 255) / 256), [&](size_t c0) {
            for (uint64_t t = c0 * 256; t < n_tiles && t < (c0 + 1) * 256; ++t) memcpy(dst + toff[t], slots + t * stride, (size_t)h_tcnt[t] * 16);
        });
        else for (uint64_t t = 0; t < n_tiles; ++t) memcpy(dst + toff[t], slots + t * stride, (size_t)h_tcnt[t] * 16);
    }
    const std::vector<uint64_t> &h_toff = toff;
    for (size_t i = 0; i <= n; ++i) out_off[i] = h_toff[first_tile[i]];
    out = W.h_compact.as<mm2::Anchor>();
    W.staged_soff = soff, W.staged_n = n;              // (pinned: valid until the workspace's next call, like the device copy of the sequences)
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->sketch_mm_ms += now_ms() - t0;
    g_sketch_ms[0] += t_staged - t0, g_sketch_ms[3] += now_ms() - t_staged, g_sketch_ms[4] += (double)bytes, g_sketch_ms[5] += (double)out_off[n];
    return 0;
}

static int gpu_mm_sketch_one(nsgpu_ctx *c, const std::vector<SketchReq> &reqs, int w, int k, const mm2::Anchor *&out, std::vector<uint64_t> &out_off, int ws, size_t n_stage_only);

// request i of the workspace's last batch as it lies in HBM (the bytes the caller passed), or nullptr when that batch did not go through the
// fused path's staging buffer; valid until the workspace's next call
const uint8_t *sketch_dev_seq(const nsgpu_ctx *c, int ws, size_t i)
{
    const nsgpu_ctx::SketchWs &W = c->sws[ws];
    return W.staged_soff && i < W.staged_n ? W.seqs.as<uint8_t>() + W.staged_soff[i] : nullptr;
}

// Any number of sequences: batches beyond the 32-bit position space of the kernels (2 GiB of sequence) are sketched piece by piece and
// the results concatenated (a contig engine with many builders on multi-megabase contigs can get there; it cannot act on an error).
int gpu_mm_sketch(nsgpu_ctx *c, const std::vector<SketchReq> &reqs, int w, int k, const mm2::Anchor *&out, std::vector<uint64_t> &out_off, int ws, size_t n_stage_only)
{
    NS_CHECK(n_stage_only <= reqs.size() && (ws == 0 || ws == 1), NSGPU_ERR_ARG, "gpu_mm_sketch: bad arguments");
    c->sws[ws].staged_soff = nullptr, c->sws[ws].staged_n = 0;
    // (NSGPU_SKETCH_PIECE_KB: a small piece size lets a test force the split)
    static const uint64_t kPiece = [] { const char *e = getenv("NSGPU_SKETCH_PIECE_KB"); const uint64_t kb = e ? strtoull(e, nullptr, 10) : 0; return kb ? kb << 10 : 1500ull << 20; }();
    uint64_t bytes = 0;
    for (const SketchReq &r : reqs) bytes += (r.len + 16) & ~(uint64_t)15;
    if (bytes < kPiece) return gpu_mm_sketch_one(c, reqs, w, k, out, out_off, ws, n_stage_only);
    // The pieces' results are concatenated in PINNED memory of this context and workspace: the seeding kernel reads the lists where this
    // function leaves them (AlignReq.qry_mz / ref_mz are device-visible pointers), and two contexts must not share a buffer.  The result
    // stays valid until the workspace's next call.
    NS_CHECK(ws == 0 || ws == 1, NSGPU_ERR_ARG, "gpu_mm_sketch: workspace 0 or 1");
    nsgpu::PinBuf &all = c->sws[ws].h_concat;
    uint64_t n_all = 0;
    out_off.assign(reqs.size() + 1, 0);
    std::vector<SketchReq> part;
    std::vector<uint64_t> poff;
    size_t i = 0;
    while (i < reqs.size()) {
        part.clear();
        uint64_t b = 0;
        const size_t first = i;
        while (i < reqs.size() && (part.empty() || b + ((reqs[i].len + 16) & ~(uint64_t)15) < kPiece)) { b += (reqs[i].len + 16) & ~(uint64_t)15; part.push_back(reqs[i++]); }
        const mm2::Anchor *po = nullptr;
        NS_TRY(gpu_mm_sketch_one(c, part, w, k, po, poff, ws, 0));
        for (size_t j = 0; j < part.size(); ++j) out_off[first + j + 1] = n_all + poff[j + 1];
        const uint64_t n_new = poff[part.size()];
        if ((n_all + n_new + 1) * sizeof(mm2::Anchor) > all.cap) {          // grow, keeping what is there
            nsgpu::PinBuf bigger;
            NS_TRY(bigger.reserve(2 * (n_all + n_new + 1) * sizeof(mm2::Anchor)));
            if (n_all) memcpy(bigger.p, all.p, n_all * sizeof(mm2::Anchor));
            all.release();
            all = bigger;
        }
        if (n_new) memcpy(all.as<mm2::Anchor>() + n_all, po, n_new * sizeof(mm2::Anchor));
        n_all += n_new;
    }
    out = all.as<mm2::Anchor>();
    c->sws[ws].staged_soff = nullptr, c->sws[ws].staged_n = 0;      // (the staging buffer holds the last piece only: sketch_dev_seq has nothing to offer)
    return NSGPU_OK;
}

static int gpu_mm_sketch_one(nsgpu_ctx *c, const std::vector<SketchReq> &reqs_in, int w, int k, const mm2::Anchor *&out, std::vector<uint64_t> &out_off, int ws, size_t n_stage_only)
{
    out_off.assign(reqs_in.size() + 1, 0);
    out = nullptr;
    if (reqs_in.empty()) return NSGPU_OK;
    if (k > 0 && k <= 28 && w > 0 && w < 256 && (ws == 0 || ws == 1)) {
        const int rc = gpu_mm_sketch_fused(c, reqs_in, w, k, out, out_off, ws, n_stage_only);
        if (rc <= 0) return rc;
        out_off.assign(reqs_in.size() + 1, 0);
        out = nullptr;
    }
    // the general passes sketch the requests proper; the staged-only ones have no minimizers and (here) no device copy the caller may use
    c->sws[ws].staged_soff = nullptr, c->sws[ws].staged_n = 0;
    const std::vector<SketchReq> reqs(reqs_in.begin(), reqs_in.end() - (ptrdiff_t)n_stage_only);
    const size_t n = reqs.size();
    struct PadOff { std::vector<uint64_t> &o; size_t total; ~PadOff() { const uint64_t last = o.empty() ? 0 : o.back(); o.resize(total + 1, last); } } pad_off{out_off, reqs_in.size()};
    out_off.assign(n + 1, 0);
    if (n == 0) return NSGPU_OK;
    NS_CHECK(k > 0 && k <= 28 && w > 0 && w < 256, NSGPU_ERR_ARG, "minimap k must be in 1..28 and w in 1..255 (sketch.c:84)");
    const double t0 = now_ms();
    NS_CHECK(ws == 0 || ws == 1, NSGPU_ERR_ARG, "gpu_mm_sketch: workspace 0 or 1");
    nsgpu_ctx::SketchWs &W = c->sws[ws];
    // a stream of its own: in the contig engine the sketches of one builder group run while another group's window queries
    // use the context's stream
    if (!W.stream) NS_TRY(role_stream_create(&W.stream, "sketch"));
    const hipStream_t st = W.stream;
    std::vector<uint32_t> soff(n + 1), len(n);
    uint64_t bytes = 0;
    for (size_t i = 0; i < n; ++i) {
        NS_CHECK(reqs[i].len < (1ull << 31), NSGPU_ERR_RANGE, "sequence %zu longer than 2^31", i);
        soff[i] = (uint32_t)bytes;
        len[i] = (uint32_t)reqs[i].len;
        bytes += (reqs[i].len + 16) & ~(uint64_t)15;                   // at least one padding position after every sequence
        NS_CHECK(bytes < (1ull << 31), NSGPU_ERR_RANGE, "sketch batch exceeds 2 GiB of sequence; use smaller batches");
    }
    soff[n] = (uint32_t)bytes;
    const uint32_t B = (uint32_t)bytes;
    // sequences through pinned memory, one H2D copy (padding bytes are never interpreted)
    NS_TRY(pinned_reserve(W.h_seqs, W.h_cap, bytes + 16));
    par_for("sketch.stage", n, [&](size_t i) { memcpy(W.h_seqs + soff[i], reqs[i].ptr, reqs[i].len); });
    const double t_staged = now_ms();
    DevBuf *u32bufs[] = {&W.vf, &W.mk, &W.vr, &W.linv, &W.npf, &W.pushf, &W.npr, &W.pr};
    for (DevBuf *b : u32bufs) NS_TRY(b->reserve(((size_t)B + 2) * 4));
    NS_TRY(W.seqs.reserve(bytes + 64));
    NS_TRY(W.soff.reserve((n + 1) * 4));
    NS_TRY(W.len.reserve(n * 4 + 4));
    NS_TRY(W.sob.reserve(((size_t)B / 16 + 1) * 4));
    NS_TRY(W.V.reserve((size_t)B + 64));
    NS_TRY(W.hk.reserve(((size_t)B + 1) * 8));
    NS_HIP(hipMemcpyAsync(W.seqs.p, W.h_seqs, bytes, hipMemcpyHostToDevice, st));
    if (n_stage_only == 0) { W.staged_soff_v = soff; W.staged_soff = W.staged_soff_v.data(), W.staged_n = n; }      // (sketch_dev_seq: the requests lie in W.seqs here too)
    NS_HIP(hipMemcpyAsync(W.soff.p, soff.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
    NS_HIP(hipMemcpyAsync(W.len.p, len.data(), n * 4, hipMemcpyHostToDevice, st));
    const Batch bt{W.seqs.as<uint8_t>(), W.soff.as<uint32_t>(), W.len.as<uint32_t>(), (uint32_t)n, B, w, k};
    const uint32_t gB = (B + 1 + 255) / 256;
    hipLaunchKernelGGL(sk_block_owner_kernel, dim3((B / 16 + 255) / 256), dim3(256), 0, st, bt, W.sob.as<uint32_t>());
    hipLaunchKernelGGL(sk_flags_kernel, dim3(gB), dim3(256), 0, st, bt, W.sob.as<uint32_t>(), W.vf.as<uint32_t>(), W.mk.as<uint32_t>());
    NS_HIP(hipGetLastError());
    {   // valid-position count (inclusive sum) and position of the last invalid byte (inclusive maximum) in ONE scan over pairs
        using T2 = rocprim::tuple<uint32_t, uint32_t>;
        auto in2 = rocprim::make_zip_iterator(rocprim::make_tuple(W.vf.as<uint32_t>(), W.mk.as<uint32_t>()));
        auto out2 = rocprim::make_zip_iterator(rocprim::make_tuple(W.vr.as<uint32_t>(), W.linv.as<uint32_t>()));
        auto sum_max = [] __device__(const T2 &a, const T2 &b) {
            const uint32_t ma = rocprim::get<1>(a), mb = rocprim::get<1>(b);
            return T2(rocprim::get<0>(a) + rocprim::get<0>(b), ma > mb ? ma : mb);
        };
        size_t ws_bytes = 0;
        NS_HIP(rocprim::inclusive_scan(nullptr, ws_bytes, in2, out2, (size_t)B + 1, sum_max, st));
        NS_TRY(W.scan_ws.reserve(ws_bytes + 16));
        NS_HIP(rocprim::inclusive_scan(W.scan_ws.p, ws_bytes, in2, out2, (size_t)B + 1, sum_max, st));
    }
    hipLaunchKernelGGL(sk_compact_kernel, dim3(gB), dim3(256), 0, st, bt, W.vf.as<uint32_t>(), W.vr.as<uint32_t>(), W.V.as<uint8_t>());
    hipLaunchKernelGGL(sk_kmer_kernel, dim3(gB), dim3(256), 0, st, bt, W.sob.as<uint32_t>(), W.vf.as<uint32_t>(), W.vr.as<uint32_t>(), W.V.as<uint8_t>(),
                       W.hk.as<uint64_t>(), W.npf.as<uint32_t>(), W.pushf.as<uint32_t>());
    NS_HIP(hipGetLastError());
    {   // the two exclusive sums (non-palindromic k-mers, pushes) in ONE scan over pairs: two launches less per call -- the
        // stage is bound by the number of GPU operations, not by their size
        using T2 = rocprim::tuple<uint32_t, uint32_t>;
        auto in2 = rocprim::make_zip_iterator(rocprim::make_tuple(W.npf.as<uint32_t>(), W.pushf.as<uint32_t>()));
        auto out2 = rocprim::make_zip_iterator(rocprim::make_tuple(W.npr.as<uint32_t>(), W.pr.as<uint32_t>()));
        auto plus2 = [] __device__(const T2 &a, const T2 &b) { return T2(rocprim::get<0>(a) + rocprim::get<0>(b), rocprim::get<1>(a) + rocprim::get<1>(b)); };
        size_t ws_bytes = 0;
        NS_HIP(rocprim::exclusive_scan(nullptr, ws_bytes, in2, out2, T2(0u, 0u), (size_t)B + 1, plus2, st));
        NS_TRY(W.scan_ws.reserve(ws_bytes + 16));
        NS_HIP(rocprim::exclusive_scan(W.scan_ws.p, ws_bytes, in2, out2, T2(0u, 0u), (size_t)B + 1, plus2, st));
    }
    NS_TRY(W.h_meta.reserve((n + 1) * 8 + 64));
    NS_HIP(hipMemcpyAsync(W.h_meta.p, W.pr.as<uint32_t>() + B, 4, hipMemcpyDeviceToHost, st));
    NS_HIP(stream_wait_short(st));
    const uint32_t P = *W.h_meta.as<uint32_t>();                         // pushes of the whole batch
    const double t1 = now_ms();
    NS_TRY(W.PX.reserve(((size_t)P + 1) * 8));
    NS_TRY(W.PY.reserve(((size_t)P + 1) * 8));
    NS_TRY(W.PRUN.reserve(((size_t)P + 2) * 4));
    NS_TRY(W.PSEQ.reserve(((size_t)P + 2) * 4));
    NS_TRY(W.nout.reserve(((size_t)P + 2) * 4));
    NS_TRY(W.rm.reserve(((size_t)P + 2) * 4));
    NS_TRY(W.oscan.reserve(((size_t)P + 2) * 4));
    NS_TRY(W.off.reserve((n + 1) * 8));
    const uint32_t gP = (P + 1 + 255) / 256;
    hipLaunchKernelGGL(sk_push_kernel, dim3(gB), dim3(256), 0, st, bt, W.sob.as<uint32_t>(), W.vf.as<uint32_t>(), W.pushf.as<uint32_t>(), W.pr.as<uint32_t>(),
                       W.npr.as<uint32_t>(), W.linv.as<uint32_t>(), W.hk.as<uint64_t>(), W.PX.as<uint64_t>(), W.PY.as<uint64_t>(), W.PRUN.as<uint32_t>(),
                       W.PSEQ.as<uint32_t>());
    hipLaunchKernelGGL(sk_window_min_kernel, dim3(gP), dim3(256), 0, st, bt, W.PX.as<uint64_t>(), W.PSEQ.as<uint32_t>(), W.pr.as<uint32_t>(), P, W.rm.as<uint32_t>());
    hipLaunchKernelGGL(sk_count_kernel, dim3(gP), dim3(256), 0, st, bt, W.PX.as<uint64_t>(), W.PY.as<uint64_t>(), W.PRUN.as<uint32_t>(), W.PSEQ.as<uint32_t>(),
                       W.pr.as<uint32_t>(), W.rm.as<uint32_t>(), P, W.nout.as<uint32_t>());
    NS_HIP(hipGetLastError());
    NS_TRY(scan_u32(W, st, W.nout.as<uint32_t>(), W.oscan.as<uint32_t>(), (size_t)P + 1, false, 0));
    // the offsets go straight into pinned host memory (device-visible): one copy operation less on the GPU's front end
    hipLaunchKernelGGL(sk_offsets_kernel, dim3(((uint32_t)n + 1 + 255) / 256), dim3(256), 0, st, bt, W.pr.as<uint32_t>(), W.oscan.as<uint32_t>(), W.h_meta.as<uint64_t>());
    NS_HIP(hipGetLastError());
    NS_HIP(stream_wait_short(st));
    memcpy(out_off.data(), W.h_meta.p, (n + 1) * 8);
    const uint64_t total = out_off[n];
    const double t2 = now_ms();
    NS_TRY(W.out.reserve(total * 16 + 16));
    uint8_t *&h_out = W.h_out;
    NS_TRY(pinned_reserve(W.h_out, W.h_out_cap, total * 16 + 16));
    if (total) {
        hipLaunchKernelGGL(sk_write_kernel, dim3(gP), dim3(256), 0, st, bt, W.PX.as<uint64_t>(), W.PY.as<uint64_t>(), W.PRUN.as<uint32_t>(), W.PSEQ.as<uint32_t>(),
                           W.pr.as<uint32_t>(), W.rm.as<uint32_t>(), P, W.oscan.as<uint32_t>(), W.out.as<uint64_t>());
        NS_HIP(hipGetLastError());
        NS_HIP(hipMemcpyAsync(h_out, W.out.p, total * 16, hipMemcpyDeviceToHost, st));
    }
    NS_HIP(stream_wait_short(st));
    out = reinterpret_cast<const mm2::Anchor *>(h_out);
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->sketch_mm_ms += now_ms() - t0;
    g_sketch_ms[0] += t_staged - t0, g_sketch_ms[1] += t1 - t_staged, g_sketch_ms[2] += t2 - t1, g_sketch_ms[3] += now_ms() - t2, g_sketch_ms[4] += (double)bytes, g_sketch_ms[5] += (double)total;
    return NSGPU_OK;
}

}  // namespace nsgpu
