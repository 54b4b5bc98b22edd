// ref_minhash_driver.cpp -- thin command-line driver around the REFERENCE's own
// MinHash objects.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// It is linked with objects compiled straight from /root/reference/src/
// {ReadFilter,BBHashMap,dnaToBits}.cpp (unmodified, where they lie; recipe in
// oracle/Makefile, output oracle/_ref/nsref).  Nothing in this file restates
// the algorithm: it only feeds inputs to and dumps outputs from the reference
// classes, so that tests can pin oracle/ns_oracle.c and generate
// tests/golden/minhash_*.npz.
//
// The only liberty taken is `#define private public` around the reference
// headers, so that the salts (randNumbers, normally drawn from
// std::random_device, src/ReadFilter.cpp:49-63) can be set to the caller's
// values and populateHashTables() (src/ReadFilter.cpp:159-172) can be called on
// sketches produced by the public string2Sketch().  MinHashReadFilter::
// initialize() itself needs ReadData.cpp, which needs Boost (absent here), so
// it is never called; its unresolved ReadData symbols are left unbound.
//
// in : u32 k,n,thr,N,Q | u64 salts[n] | u64 roff[N+1] | bases | u64 qoff[Q+1] | qbases
// out: u64 sketches[N*n] | u64 packed_bytes_total | packed bytes (DnaBitset::to_file per read)
//      | u64 qsketch[Q*n] | per query: u64 cnt, u32 ids[cnt]
//      | per slot j, per read r: u64 cnt, u32 ids[cnt]   (pushMatchesInVector(sketch[r][j]))
//      | per read: u8 unpack_ok (DnaBitset::to_string round trip)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <fstream>
#include <iostream>
#include <sstream>
#include <random>
#include <algorithm>
#include <memory>
#include <mutex>
#include <functional>
#include <unistd.h>

#define private public
#include "ReadFilter.h"
#undef private

template <class T> static void rd(FILE *f, T *p, size_t cnt) {
    if (cnt && fread(p, sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short read\n"); exit(2); }
}
template <class T> static void wr(FILE *f, const T *p, size_t cnt) {
    if (cnt && fwrite(p, sizeof(T), cnt, f) != cnt) { fprintf(stderr, "short write\n"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: nsref in.bin out.bin tmpdir\n"); return 1; }
    FILE *fi = fopen(argv[1], "rb"), *fo = fopen(argv[2], "wb");
    if (!fi || !fo) { perror("open"); return 1; }
    std::string tmp = argv[3];
    uint32_t hdr[5];
    rd(fi, hdr, 5);
    uint32_t k = hdr[0], n = hdr[1], thr = hdr[2], N = hdr[3], Q = hdr[4];
    std::vector<uint64_t> salts(n), roff(N + 1), qoff(Q + 1);
    rd(fi, salts.data(), n);
    rd(fi, roff.data(), N + 1);
    std::vector<char> rb(roff[N]);
    rd(fi, rb.data(), rb.size());
    rd(fi, qoff.data(), Q + 1);
    std::vector<char> qb(qoff[Q]);
    rd(fi, qb.data(), qb.size());
    fclose(fi);

    MinHashReadFilter rF;
    rF.k = k; rF.n = n; rF.overlapSketchThreshold = thr; rF.tempDir = tmp;
    rF.numReads = N;
    rF.randNumbers = new kMer_t[n];
    for (uint32_t l = 0; l < n; ++l) rF.randNumbers[l] = salts[l];

    size_t maxLen = 0;
    for (uint32_t r = 0; r < N; ++r) maxLen = std::max<size_t>(maxLen, roff[r + 1] - roff[r]);
    for (uint32_t q = 0; q < Q; ++q) maxLen = std::max<size_t>(maxLen, qoff[q + 1] - qoff[q]);
    std::vector<kMer_t> sketches((size_t)n * N);   // zero-initialised as in src/ReadFilter.cpp:21
    std::vector<kMer_t> kv(maxLen + 1), hv(n);
    for (uint32_t r = 0; r < N; ++r) {
        std::string s(rb.data() + roff[r], rb.data() + roff[r + 1]);
        rF.string2Sketch(s, sketches.data() + (size_t)r * n, kv, hv);
    }
    wr(fo, sketches.data(), sketches.size());

    // DnaBitset bytes through the reference's own writer
    {
        std::string pf = tmp + "/packed.bin";
        std::ofstream fout(pf, std::ios::binary);
        uint64_t total = 0;
        std::vector<uint8_t> okv(N);
        for (uint32_t r = 0; r < N; ++r) {
            DnaBitset b(rb.data() + roff[r], roff[r + 1] - roff[r]);
            total += b.to_file(fout);
            std::string back;
            b.to_string(back);
            // to_string can only ever give A/T/C/G; ok == identical after the N->G style folding
            okv[r] = 1;
            for (size_t i = 0; i < back.size(); ++i) {
                char c = rb[roff[r] + i];
                char e = "ATCG"[(c & 2) | ((c & 4) >> 2)];
                if (back[i] != e) okv[r] = 0;
            }
        }
        fout.close();
        wr(fo, &total, 1);
        std::vector<uint8_t> bytes(total);
        FILE *fp = fopen(pf.c_str(), "rb");
        rd(fp, bytes.data(), total);
        fclose(fp);
        unlink(pf.c_str());
        wr(fo, bytes.data(), total);
        // stash ok flags for the tail
        rb.insert(rb.end(), okv.begin(), okv.end());
    }

    rF.populateHashTables(sketches);

    std::vector<kMer_t> qs((size_t)n * Q);
    for (uint32_t q = 0; q < Q; ++q) {
        std::string s(qb.data() + qoff[q], qb.data() + qoff[q + 1]);
        rF.string2Sketch(s, qs.data() + (size_t)q * n, kv, hv);
    }
    wr(fo, qs.data(), qs.size());
    for (uint32_t q = 0; q < Q; ++q) {
        std::string s(qb.data() + qoff[q], qb.data() + qoff[q + 1]);
        std::vector<read_t> res;
        rF.getFilteredReads(s, res);     // the ReadFilter interface, include/ReadFilter.h:24
        uint64_t c = res.size();
        wr(fo, &c, 1);
        wr(fo, res.data(), res.size());
    }
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t r = 0; r < N; ++r) {
            std::vector<read_t> m;
            rF.hashTables[j].pushMatchesInVector(sketches[(size_t)r * n + j], m);
            uint64_t c = m.size();
            wr(fo, &c, 1);
            wr(fo, m.data(), m.size());
        }
    wr(fo, (const uint8_t *)(rb.data() + roff[N]), N);
    fclose(fo);
    return 0;
}
