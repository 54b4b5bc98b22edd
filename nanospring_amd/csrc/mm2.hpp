// ---------------------------------------------------------------------------------------------------------------------
// Third-party notice.  The functions in this file reproduce, decision for decision, the behaviour of minimap2 v2.17
// (https://github.com/lh3/minimap2; files cited per function) -- bit-exact tie orders and thresholds are part of the
// contract of NanoSpring's on-disk format, so the order of the decisions is minimap2's by necessity.  minimap2 is
//   Copyright (c) 2018- Dana-Farber Cancer Institute, 2017-2018 Broad Institute, Inc.
// and distributed under the MIT License; its full text is in THIRD_PARTY_NOTICES.md at the root of this repository.
// ---------------------------------------------------------------------------------------------------------------------
// mm2.hpp -- host side of the "ReadAligner" half of the hot path: the decision
// chain of minimap2 v2.17 as NanoSpring drives it (ConsensusGraph::alignRead,
// src/ConsensusGraph.cpp:161-398): minimizer sketch, single-sequence index,
// seeds, chaining, region bookkeeping and the alignment skeleton.  Every banded
// DP (ksw_extd2) is NOT computed here: the skeleton asks for it through DpCache
// and the batch driver runs all outstanding requests of all pairs in one launch
// of the HIP wavefront kernel (ksw2.hip).
//
// Written from scratch against the behaviour of the reference (cited per
// function, paths relative to the NanoSpring tree); bit-exactness is tested
// against the reference's own minimap2 build (tests/test_align_gpu.py,
// tests/golden/align_*.npz).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <map>
#include <string>
#include <memory>

namespace nsgpu {
namespace mm2 {

struct Anchor { uint64_t x, y; };            // mm128_t (minimap2/minimap.h:53)

// seed flags (minimap2/mmpriv.h:17-23)
constexpr uint64_t SEED_LONG_JOIN = 1ull << 40, SEED_IGNORE = 1ull << 41, SEED_TANDEM = 1ull << 42;
constexpr int PARENT_UNSET = -1, PARENT_TMP_PRI = -2;

// mm_mapopt_t after mm_set_opt(0,...) + NanoSpring's overrides (minimap2/options.c:14-54,
// src/ConsensusGraph.cpp:200-203); SURVEY appendix A5.
struct Opt {
    int k = 20, w = 50, bucket_bits = 14;
    int seed = 11;
    float mid_occ_frac = 2e-4f;
    int min_cnt = 3, min_chain_score = 40, bw = 500, max_gap = 5000, max_chain_skip = 25, max_chain_iter = 400;
    float chain_gap_scale = 1.0f, mask_level = 0.5f;
    int mask_len = 0x7fffffff;
    float pri_ratio = 0.8f;
    int best_n = 5, max_join_long = 20000, max_join_short = 2000, min_join_flank_sc = 1000;
    float min_join_flank_ratio = 0.5f, alt_drop = 0.15f;
    int a = 2, b = 4, q = 4, e = 2, q2 = 24, e2 = 1, sc_ambi = 1, zdrop = 400, zdrop_inv = 200, end_bonus = -1;
    int min_dp_max = 80, min_ksw_len = 200;
    float max_clip_ratio = 1.0f;
};

void nt4_codes(const char *s, size_t n, uint8_t *out);                                            // seq_nt4_table applied to a string
void mm_sketch(const char *str, int len, int w, int k, uint32_t rid, std::vector<Anchor> &out);   // minimap2/sketch.c:77-143
void radix_sort_128x(Anchor *beg, Anchor *end);                                                  // minimap2/ksort.h:98-151 via misc.c:153-156
void radix_sort_64(uint64_t *beg, uint64_t *end);                                                // misc.c:158-159

// mm_idx_str for ONE sequence + mm_mapopt_update (index.c:386-434, 164-185; options.c:56-66)
struct RefIndex {
    int k = 0, w = 0;
    uint32_t len = 0;
    std::vector<uint8_t> seq;          // nt4 codes (mm_idx_getseq source; mmpriv.h:29-30)
    std::vector<uint64_t> keys;        // distinct minimizer hashes, ascending
    std::vector<uint32_t> start;       // CSR into pos
    std::vector<uint64_t> pos;         // per key: y values ascending (index.c:230)
    std::vector<uint32_t> slot;        // open-addressing table over keys (index into keys + 1, 0 = empty); get() probes it
    uint32_t slot_shift = 64;
    std::vector<uint64_t> sort_keys_, sort_tmp_;  // scratch of build_from_sketch
    int32_t mid_occ = 0;
    bool has_table = false;            // keys / pos / slot / mid_occ are built (the batch driver seeds on the GPU and needs only seq)
    void set_sequence(const char *s, uint32_t n, int w_, int k_);     // k, w, len and the nt4 codes only
    void set_sequence_from(const char *s, uint32_t n, int w_, int k_, size_t from);     // the same when s[0 .. from) is what the codes already hold
    // the same when s shares a prefix of P and a suffix of S bases with the string the codes belong to: the suffix's codes move, the middle is coded
    void set_sequence_spliced(const char *s, uint32_t n, int w_, int k_, size_t P, size_t S);
    void build(const char *s, uint32_t n, int w_, int k_, float mid_occ_frac);
    // same, with the sequence's minimizers (mm_sketch order, rid 0) supplied by the caller
    void build_from_sketch(const char *s, uint32_t n, int w_, int k_, float mid_occ_frac, const Anchor *mz, size_t n_mz);
    const uint64_t *get(uint64_t minier, int *n) const;   // mm_idx_get (index.c:81-98)
};

struct Extra {                         // mm_extra_t (minimap.h:77-84)
    int32_t dp_score = 0, dp_max = 0, dp_max2 = 0;
    uint32_t n_ambi = 0;
    std::vector<uint32_t> cigar;
};

struct Reg {                           // mm_reg1_t (minimap.h:86-103)
    int32_t id = 0, cnt = 0, rid = 0, score = 0, qs = 0, qe = 0, rs = 0, re = 0, parent = 0, subsc = 0, as = 0, mlen = 0, blen = 0,
            n_sub = 0, score0 = 0;
    uint8_t split = 0, rev = 0, inv = 0, sam_pri = 0, split_inv = 0;
    uint32_t hash = 0;
    bool has_p = false;
    Extra p;
};

// One banded-DP request of the skeleton (mm_align_pair, align.c:313-339).  Coordinates
// are on the forward query / reference; `left` = the left extension, whose two
// sequences are reversed before the DP (align.c:693-696).
struct DpKey {
    int32_t qs, qe, rs, re, w, zdrop, end_bonus, flag;
    bool operator<(const DpKey &o) const;
};
struct CigSpan {                       // a CIGAR held by somebody else (the job's DpCache pool)
    const uint32_t *p = nullptr; uint32_t n = 0;
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    const uint32_t *begin() const { return p; }
    const uint32_t *end() const { return p + n; }
    uint32_t operator[](size_t i) const { return p[i]; }
};
struct DpResult {                      // ksw_extz_t; the CIGAR lives in the cache's pool
    uint32_t max = 0; int32_t zdropped = 0, max_q = -1, max_t = -1, mqe = 0, mqe_t = -1, mte = 0, mte_q = -1, score = 0, reach_end = 0;
    uint32_t cig_off = 0, n_cigar = 0;
};
// The DP results a job has received so far.  Flat arrays that keep their capacity from one alignment to the next: a
// job asks for a few dozen DPs, in (nearly) the order it later reads them, so a scan that starts behind the last hit
// finds most keys at once -- and nothing is allocated or freed per result.
struct DpCache {
    std::vector<DpKey> keys;
    std::vector<DpResult> vals;
    std::vector<uint32_t> pool;
    std::vector<DpKey> missing;        // requests discovered by the last step()
    size_t cursor = 0;
    const DpResult *find(const DpKey &k);
    const DpResult *get(const DpKey &k);           // find, or note the key as missing
    const DpResult &at(const DpKey &k);            // must be present
    void put(const DpKey &k, const DpResult &scalars, const uint32_t *cigar, uint32_t n_cigar);
    CigSpan cigar(const DpResult &r) const { return CigSpan{pool.data() + r.cig_off, r.n_cigar}; }
    void clear() { keys.clear(); vals.clear(); pool.clear(); missing.clear(); cursor = 0; }
    void swap(DpCache &o) { keys.swap(o.keys); vals.swap(o.vals); pool.swap(o.pool); missing.swap(o.missing); std::swap(cursor, o.cursor); }
};

float chain_avg_qspan(const std::vector<Anchor> &a);
// the sequential remainder of mm_chain_dp behind the forward pass (chain.c:94-164): chains u[] and the anchors reordered chain by chain
void chain_finish_scores(const Opt &o, std::vector<Anchor> &a, const int32_t *f, const int32_t *p, std::vector<uint64_t> &u);
#ifdef NSGPU_HOST_CHAIN
// the chaining recurrence as a plain loop: compiled into the CPU test harness only, the product runs chain.hip
void chain_forward_host(const Opt &o, const std::vector<Anchor> &a, float avg_qspan, int32_t *f, int32_t *p);
#endif

// The alignment of one (reference, query) pair as a resumable job.
struct AlignJob {
    const RefIndex *ref = nullptr;
    const char *qstr = nullptr;
    int qlen = 0;
    Opt opt;
    // results
    bool finished = false;
    std::vector<Reg> regs;             // final hits (mm_map output order)
    // internals
    std::vector<uint8_t> qseq;         // nt4 codes, forward strand only (MM_F_FOR_ONLY)
    std::vector<Anchor> a;
    int32_t n_a = 0;
    int cur = 0;                       // region being aligned in the skeleton loop
    bool seeded = false, prepared = false, chained = false;
    float avg_qspan = 0.f;             // mean query span of the anchors (chain.c:36-37), an input of the chaining scores
    const int32_t *cf = nullptr, *cp = nullptr;   // chaining score / predecessor of every anchor from the GPU pass (chain.hip); consumed by step()
    std::vector<int32_t> own_f, own_p; // a job's own copy of the scores (host-seeded jobs of a batch; NSGPU_HOST_CHAIN builds)
    const Anchor *pre_mz = nullptr;    // the query's minimizers, when the caller sketched it (set after start())
    size_t n_pre_mz = 0;
    DpCache cache;
    void start(const RefIndex *r, const char *q, int ql, const Opt &o);
    void seed_prepare();               // the query's nt4 codes
    void seed();                       // + anchors (a) from the host index, sorted as the chaining expects them
    void set_anchors(const Anchor *p, size_t n, float avg);          // ... or the anchors as seeds.hip computed them
    bool step();                       // true when finished; otherwise cache.missing is non-empty
    void swap_storage(AlignJob &o) { regs.swap(o.regs); qseq.swap(o.qseq); a.swap(o.a); cache.swap(o.cache); }
};

// ConsensusGraph::alignRead's conversion of reg[0] (src/ConsensusGraph.cpp:219-397)
struct EditOp { uint8_t type; uint8_t base; uint32_t num; };   // type: 0 SAME(num) 1 INSERT(base) 2 DELETE(base)
struct AlnOut {
    int32_t ok = 0, hits = 0;
    int64_t rel_pos = 0, begin_offset = 0, end_offset = 0;
    int32_t rs = 0, re = 0, qs = 0, qe = 0, blen = 0, mlen = 0, n_ambi = 0, dp_max = 0, n_cigar = 0;
    std::vector<uint32_t> cigar;
    std::vector<EditOp> edits;
    // back to the default state, keeping the vectors' storage (results are recycled between builders and batches)
    void reset() { ok = hits = 0; rel_pos = begin_offset = end_offset = 0; rs = re = qs = qe = blen = mlen = n_ambi = dp_max = n_cigar = 0; cigar.clear(); edits.clear(); }
};
void align_read_result(const AlignJob &job, const char *ref, size_t ref_len, AlnOut &out);
void edits_from_hit(int hits, int rs, int re, int qs, int qe, int blen, int mlen, int n_ambi, int dp_max, bool has_p,
                    const std::vector<uint32_t> &cigar, const char *ref, size_t ref_len, const char *qry, size_t qry_len, AlnOut &out);

}  // namespace mm2
}  // namespace nsgpu
