/*
 * ns_oracle.c -- CPU restatement of NanoSpring's MinHash read filter.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path;
 * nothing under nanospring_amd/ may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Pinning: every function here is checked bit-for-bit against the reference's
 * own objects (oracle/_ref/nsref, built from /root/reference/src/ReadFilter.cpp,
 * BBHashMap.cpp, dnaToBits.cpp) by tests/test_oracle_pin.py when oracle/_ref
 * exists, and against tests/golden/minhash_*.npz (vectors emitted by that same
 * reference build; generator tests/golden/make_golden.py) everywhere else.
 *
 * Each function cites the reference lines it restates (paths relative to the
 * NanoSpring tree).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* src/dnaToBits.cpp:6-8 and src/ReadFilter.cpp:113-115 -- A0 T1 C2 G3, any
 * other byte maps through the same two bits (N -> 3). */
static inline unsigned base_code(char c) { return (unsigned)((c & 2) | ((c & 4) >> 2)); }

/* src/dnaToBits.cpp:10-36 -- 4 bases per byte, first base in bits 7:6, the last
 * byte is left-justified (low bits zero). out must hold (len+3)/4 bytes. */
void oracle_pack2bit(const char *s, uint64_t len, uint8_t *out)
{
    uint64_t nb = len / 4 + (len % 4 != 0), i;
    memset(out, 0, nb);
    for (i = 0; i < len; ++i)
        out[i >> 2] |= (uint8_t)(base_code(s[i]) << (6 - 2 * (i & 3)));
}

/* src/dnaToBits.cpp:81-98 */
void oracle_unpack2bit(const uint8_t *in, uint64_t len, char *out)
{
    static const char dna[4] = {'A', 'T', 'C', 'G'};
    uint64_t i;
    for (i = 0; i < len; ++i) out[i] = dna[(in[i >> 2] >> (6 - 2 * (i & 3))) & 3];
}

/* include/ReadData.h:163-172 + src/ReadData.cpp:247-260 -- reverse complement on
 * ASCII; A<->T, C<->G. */
void oracle_revcomp(const char *s, uint64_t len, char *out)
{
    uint64_t i;
    for (i = 0; i < len; ++i) {
        char c = s[len - 1 - i], r;
        switch (c) { case 'A': r = 'T'; break; case 'T': r = 'A'; break;
                     case 'C': r = 'G'; break; case 'G': r = 'C'; break; default: r = c; }
        out[i] = r;
    }
}

/* src/ReadFilter.cpp:117-152 (string2Sketch, hashKMer, string2KMers, kMerToInt).
 * std::hash<uint64_t> is the identity in libstdc++, so the hash is kmer ^ salt.
 * len <  k-1 : sketch untouched (caller zero-fills, src/ReadFilter.cpp:21)
 * len == k-1 : all ones
 * else       : min over the len-k+1 k-mers. */
void oracle_sketch(const char *s, uint64_t len, uint32_t k, uint32_t n,
                   const uint64_t *salts, uint64_t *sketch)
{
    int64_t nk = (int64_t)len - (int64_t)k + 1;
    uint64_t mask, kmer = 0, i;
    uint32_t l;
    if (nk < 0) return;
    for (l = 0; l < n; ++l) sketch[l] = ~(uint64_t)0;
    if (nk == 0) return;
    mask = (k >= 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
    for (i = 0; i < len; ++i) {
        kmer = ((kmer << 2) | base_code(s[i])) & mask;
        if (i + 1 >= k)
            for (l = 0; l < n; ++l) {
                uint64_t h = kmer ^ salts[l];
                if (h < sketch[l]) sketch[l] = h;
            }
    }
}

/* src/ReadFilter.cpp:21-44 -- sketches for all reads, row-major [N][n], zero-
 * initialised.  Reads are given as concatenated ASCII with N+1 offsets. */
void oracle_sketch_reads(const char *bases, const uint64_t *off, uint32_t N, uint32_t k,
                         uint32_t n, const uint64_t *salts, uint64_t *sketches)
{
    int64_t r;
    memset(sketches, 0, (size_t)N * n * sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 16)
    for (r = 0; r < (int64_t)N; ++r)
        oracle_sketch(bases + off[r], off[r + 1] - off[r], k, n, salts, sketches + (size_t)r * n);
}

/* ---- bucket tables: src/BBHashMap.cpp:10-99 -------------------------------
 * Semantics kept: per slot j, the distinct keys and for each key the ascending
 * list of read ids whose sketch[j] equals it.  The MPHF itself is an
 * implementation detail; here the keys are kept sorted
 * and looked up by binary search.
 * Layout: keys[j*N + i] (i < nkeys[j]) ascending; start[j*(N+1) + i] CSR offsets
 * into ids[j*N ...]; start[j*(N+1)+nkeys[j]] == N. */
typedef struct { uint64_t key; uint32_t id; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

void oracle_index_build(const uint64_t *sketches, uint32_t N, uint32_t n,
                        uint64_t *keys, uint32_t *start, uint32_t *ids, uint32_t *nkeys)
{
    int64_t j;
#pragma omp parallel for schedule(dynamic, 1)
    for (j = 0; j < (int64_t)n; ++j) {
        kv_t *a = (kv_t *)malloc((size_t)(N ? N : 1) * sizeof(kv_t));
        uint64_t *kj = keys + (size_t)j * N;
        uint32_t *sj = start + (size_t)j * (N + 1), *ij = ids + (size_t)j * N;
        uint32_t r, u = 0;
        for (r = 0; r < N; ++r) { a[r].key = sketches[(size_t)r * n + j]; a[r].id = r; }
        qsort(a, N, sizeof(kv_t), kv_cmp);
        for (r = 0; r < N; ++r) {
            if (r == 0 || a[r].key != a[r - 1].key) { kj[u] = a[r].key; sj[u] = r; ++u; }
            ij[r] = a[r].id;
        }
        sj[u] = N;
        nkeys[j] = u;
        free(a);
    }
}

static int u32_cmp(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y);
}

/* src/BBHashMap.cpp:101-120 + src/ReadFilter.cpp:65-83 -- for each slot look the
 * sketch value up, concatenate the id lists, sort, and keep every id whose
 * multiplicity is >= thr (ascending).  Returns the number of ids written to
 * out (capacity cap; ids beyond cap are counted but not stored). *n_matches
 * receives the total list length M (before thresholding). */
uint64_t oracle_filter_sketch(const uint64_t *q, uint32_t N, uint32_t n, uint32_t thr,
                              const uint64_t *keys, const uint32_t *start, const uint32_t *ids,
                              const uint32_t *nkeys, uint32_t *out, uint64_t cap,
                              uint64_t *n_matches)
{
    uint64_t M = 0, cnt = 0, i, jn;
    uint32_t j, *m;
    uint64_t mcap = 1024;
    m = (uint32_t *)malloc(mcap * sizeof(uint32_t));
    for (j = 0; j < n; ++j) {
        const uint64_t *kj = keys + (size_t)j * N;
        const uint32_t *sj = start + (size_t)j * (N + 1), *ij = ids + (size_t)j * N;
        uint32_t lo = 0, hi = nkeys[j];
        while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (kj[mid] < q[j]) lo = mid + 1; else hi = mid; }
        if (lo < nkeys[j] && kj[lo] == q[j]) {
            uint32_t a = sj[lo], b = sj[lo + 1], t;
            if (M + (b - a) > mcap) { while (M + (b - a) > mcap) mcap *= 2; m = (uint32_t *)realloc(m, mcap * sizeof(uint32_t)); }
            for (t = a; t < b; ++t) m[M++] = ij[t];
        }
    }
    qsort(m, M, sizeof(uint32_t), u32_cmp);
    for (i = 0; i < M; i = jn) {
        jn = i + 1;
        while (jn < M && m[jn] == m[i]) ++jn;
        if (jn - i >= thr) { if (cnt < cap) out[cnt] = m[i]; ++cnt; }
    }
    free(m);
    if (n_matches) *n_matches = M;
    return cnt;
}

/* src/ReadFilter.cpp:85-97 -- string query. */
uint64_t oracle_filter_string(const char *s, uint64_t len, uint32_t k, uint32_t N, uint32_t n,
                              uint32_t thr, const uint64_t *salts, const uint64_t *keys,
                              const uint32_t *start, const uint32_t *ids, const uint32_t *nkeys,
                              uint32_t *out, uint64_t cap, uint64_t *n_matches)
{
    uint64_t *q = (uint64_t *)calloc(n, sizeof(uint64_t)), r;
    oracle_sketch(s, len, k, n, salts, q);
    r = oracle_filter_sketch(q, N, n, thr, keys, start, ids, nkeys, out, cap, n_matches);
    free(q);
    return r;
}

/* src/Consensus.cpp:405-424 -- checkRepetitive: for shift in 1..6 count the
 * positions where read[i] == read[(i+shift) % L]; repetitive if any count
 * exceeds 0.7*L (size_t vs double compare). */
int oracle_check_repetitive(const char *s, uint64_t len)
{
    uint64_t sh, j;
    for (sh = 1; sh <= 6; ++sh) {
        uint64_t same = 0;
        for (j = 0; j < len; ++j) same += (s[j] == s[(j + sh) % len]);
        if ((double)same > 0.7 * (double)len) return 1;
    }
    return 0;
}

/* f3: ReadData::loadFromFastqFile_highmem / _lowmem (src/ReadData.cpp:86-151, 156-221) on text in memory, as the
 * std::getline loop it is:
 *     while (getline(name)) { getline(bases); ++numReads; getline(plus); getline(quality); }
 * getline extracts up to (not including) '\n'; it fails -- and leaves the string EMPTY -- only when no character at all
 * could be extracted.  So a read is the whole line 4r+1 (a '\r' included), a missing base line is a read of length 0,
 * and an unterminated last line counts when it is not empty.  start[r] / len[r] locate read r's bases in the text.
 * Returns numReads (also when it exceeds cap; then only the first cap entries are written). */
static int fq_getline(const char *text, uint64_t n, uint64_t *pos, uint64_t *b, uint64_t *e)
{
    if (*pos >= n) { *b = *e = n; return 0; }          /* nothing left: failbit, string erased */
    *b = *pos;
    uint64_t i = *pos;
    while (i < n && text[i] != '\n') ++i;
    *e = i;
    *pos = i < n ? i + 1 : n;                          /* the delimiter is extracted and dropped */
    return 1;
}

uint64_t oracle_fastq_index(const char *text, uint64_t n, uint64_t *start, uint32_t *len, uint64_t cap)
{
    uint64_t pos = 0, nr = 0, b, e;
    while (fq_getline(text, n, &pos, &b, &e)) {
        if (!fq_getline(text, n, &pos, &b, &e)) b = e = 0;
        if (nr < cap) { start[nr] = b; len[nr] = (uint32_t)(e - b); }
        ++nr;
        fq_getline(text, n, &pos, &b, &e);
        fq_getline(text, n, &pos, &b, &e);
    }
    return nr;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
