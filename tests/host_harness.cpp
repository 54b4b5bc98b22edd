// host_harness.cpp -- TEST-ONLY driver for the host-side aligner logic (nanospring_amd/csrc/mm2.cpp).
// It runs AlignJob with the DP requests answered by the CPU oracle (oracle/ksw2_oracle.c), so that
// the decision chain (sketch, index, seeds, chaining, regions, skeleton, CIGAR fixing, edit
// conversion) can be checked against the reference's minimap2 without a GPU.  Built by
// tests/host_lib.py into tests/_build/; never part of libnsgpu.so.
#include <cstring>
#include <vector>
#include <algorithm>
#include "../nanospring_amd/csrc/mm2.hpp"

extern "C" {
typedef struct {
    uint32_t max; int32_t zdropped;
    int32_t max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} oracle_ez_t;
int oracle_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t sc_mch, int8_t sc_mis, int8_t sc_ambi_mat,
                     int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                     oracle_ez_t *ez, uint32_t *cigar_out, int cigar_cap);
}

using namespace nsgpu::mm2;

static void answer(AlignJob &J)
{
    std::vector<DpKey> miss = J.cache.missing;
    for (const DpKey &k : miss) {
        const int ql = k.qe - k.qs, tl = k.re - k.rs;
        std::vector<uint8_t> q(J.qseq.begin() + k.qs, J.qseq.begin() + k.qe), t(J.ref->seq.begin() + k.rs, J.ref->seq.begin() + k.re);
        if (k.flag & 0x02) { std::reverse(q.begin(), q.end()); std::reverse(t.begin(), t.end()); }   // left extension
        std::vector<uint32_t> cig(ql + tl + 4);
        oracle_ez_t ez;
        const Opt &o = J.opt;
        int n = oracle_ksw_extd2(ql, q.data(), tl, t.data(), (int8_t)o.a, (int8_t)-o.b, (int8_t)-o.sc_ambi, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2,
                                 (int8_t)o.e2, k.w, k.zdrop, k.end_bonus, k.flag, &ez, cig.data(), (int)cig.size());
        DpResult r;
        r.max = ez.max; r.zdropped = ez.zdropped; r.max_q = ez.max_q; r.max_t = ez.max_t; r.mqe = ez.mqe; r.mqe_t = ez.mqe_t;
        r.mte = ez.mte; r.mte_q = ez.mte_q; r.score = ez.score; r.reach_end = ez.reach_end;
        r.cigar.assign(cig.begin(), cig.begin() + (n > 0 ? n : 0));
        J.cache.done[k] = r;
    }
}

extern "C" {

typedef struct {
    int32_t ok, hits;
    int64_t rel_pos, begin_offset, end_offset;
    int32_t rs, re, qs, qe, blen, mlen, n_ambi, dp_max, n_cigar, mid_occ, n_rounds, n_dp;
} harness_aln_t;

// returns the number of edits; cigar/edits receive at most *_cap entries. edits: type | base<<8 | num<<16 (u64)
int harness_align(const char *ref, int rl, const char *qry, int ql, int k, int w, int max_chain_iter, harness_aln_t *out,
                  uint32_t *cigar, int cigar_cap, uint64_t *edits, int edit_cap)
{
    RefIndex ri;
    ri.build(ref, (uint32_t)rl, w, k, 2e-4f);
    Opt o;
    o.k = k, o.w = w, o.max_chain_iter = max_chain_iter;
    AlignJob J;
    J.start(&ri, qry, ql, o);
    int rounds = 0, n_dp = 0;
    while (!J.step()) { n_dp += (int)J.cache.missing.size(); answer(J); ++rounds; }
    AlnOut ao;
    align_read_result(J, ref, (size_t)rl, ao);
    out->ok = ao.ok; out->hits = ao.hits; out->rel_pos = ao.rel_pos; out->begin_offset = ao.begin_offset; out->end_offset = ao.end_offset;
    out->rs = ao.rs; out->re = ao.re; out->qs = ao.qs; out->qe = ao.qe; out->blen = ao.blen; out->mlen = ao.mlen; out->n_ambi = ao.n_ambi;
    out->dp_max = ao.dp_max; out->n_cigar = ao.n_cigar; out->mid_occ = ri.mid_occ; out->n_rounds = rounds; out->n_dp = n_dp;
    for (int i = 0; i < (int)ao.cigar.size() && i < cigar_cap; ++i) cigar[i] = ao.cigar[i];
    for (int i = 0; i < (int)ao.edits.size() && i < edit_cap; ++i)
        edits[i] = (uint64_t)ao.edits[i].type | (uint64_t)ao.edits[i].base << 8 | (uint64_t)ao.edits[i].num << 16;
    return (int)ao.edits.size();
}

int harness_sketch(const char *s, int len, int w, int k, uint64_t *xy, int cap)
{
    std::vector<Anchor> v;
    mm_sketch(s, len, w, k, 0, v);
    for (int i = 0; i < (int)v.size() && i < cap; ++i) { xy[2 * i] = v[i].x; xy[2 * i + 1] = v[i].y; }
    return (int)v.size();
}

void harness_radix_sort_128x(uint64_t *xy, int64_t n) { radix_sort_128x((Anchor *)xy, (Anchor *)xy + n); }
void harness_radix_sort_64(uint64_t *x, int64_t n) { radix_sort_64(x, x + n); }

}
