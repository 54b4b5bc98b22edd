/* ref_mm2_driver.c -- flat C entry points over the REFERENCE's vendored minimap2
 * (v2.17-r974-dirty), compiled from /root/reference/minimap2/ where it lies
 * (oracle/Makefile -> oracle/_ref/libmm2ref.so).  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing here restates minimap2: each function sets up the exact call sequence
 * NanoSpring uses (src/ConsensusGraph.cpp:195-217) and flattens the result
 * structs so that ctypes can read them.
 */
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include <stdint.h>
#include "minimap.h"
#include "mmpriv.h"
#include "ksw2.h"
#include "kalloc.h"

typedef struct {
    int32_t hits;
    int32_t rs, re, qs, qe;
    int32_t blen, mlen, n_ambi, dp_max, dp_score, score, cnt, rev;
    int32_t mid_occ;
    int32_t n_cigar;
} ref_aln_t;

/* The call sequence of ConsensusGraph::alignRead (src/ConsensusGraph.cpp:195-217);
 * reg[0] only (:223).  cigar receives at most cigar_cap entries. */
int ref_mm2_align(const char *ref, int rl, const char *qry, int ql, int k, int w, int max_chain_iter,
                  ref_aln_t *out, uint32_t *cigar, int cigar_cap)
{
    mm_tbuf_t *b = mm_tbuf_init();
    mm_idxopt_t iopt;
    mm_mapopt_t mopt;
    mm_idx_t *idx;
    mm_reg1_t *reg;
    int hits = 0, i;
    (void)rl;
    mm_set_opt(0, &iopt, &mopt);
    mopt.flag |= MM_F_CIGAR;
    mopt.flag |= MM_F_FOR_ONLY;
    mopt.max_chain_iter = max_chain_iter;
    idx = mm_idx_str(w, k, 0, 14, 1, &ref, NULL);
    mm_mapopt_update(&mopt, idx);
    reg = mm_map(idx, ql, qry, &hits, b, &mopt, NULL);
    memset(out, 0, sizeof(*out));
    out->hits = hits;
    out->mid_occ = mopt.mid_occ;
    if (hits > 0) {
        mm_reg1_t *r = &reg[0];
        out->rs = r->rs; out->re = r->re; out->qs = r->qs; out->qe = r->qe;
        out->blen = r->blen; out->mlen = r->mlen; out->score = r->score; out->cnt = r->cnt; out->rev = r->rev;
        if (r->p) {
            out->n_ambi = r->p->n_ambi; out->dp_max = r->p->dp_max; out->dp_score = r->p->dp_score;
            out->n_cigar = r->p->n_cigar;
            for (i = 0; i < (int)r->p->n_cigar && i < cigar_cap; ++i) cigar[i] = r->p->cigar[i];
        } else out->n_cigar = -1;
        for (i = 0; i < hits; ++i) free(reg[i].p);
    }
    free(reg);
    mm_tbuf_destroy(b);
    mm_idx_destroy(idx);
    return hits;
}

/* The number of anchors collect_seed_hits (minimap2/map.c:215-247) gives mm_map_frag's first chaining pass for this pair, counted with the
 * library's own index and sketch through its public calls: every query minimizer whose hash the reference holds fewer than mid_occ times
 * (collect_matches, map.c:103-118) contributes its same-strand hits (skip_seed with MM_F_FOR_ONLY, map.c:139-145).  The lock-step oracle's
 * rule for deferred alignments is stated on this number (oracle/consensus_oracle.cpp LockStep::VT::extra).  Thread-safe. */
int64_t ref_mm_count_seeds(const char *ref, int rl, const char *qry, int ql, int k, int w)
{
    mm_idxopt_t iopt;
    mm_mapopt_t mopt;
    mm_idx_t *idx;
    mm128_v mv = {0, 0, 0};
    int64_t n = 0;
    size_t i;
    (void)rl;
    mm_set_opt(0, &iopt, &mopt);
    mopt.flag |= MM_F_CIGAR | MM_F_FOR_ONLY;
    idx = mm_idx_str(w, k, 0, 14, 1, &ref, NULL);
    mm_mapopt_update(&mopt, idx);
    mm_sketch(0, qry, ql, idx->w, idx->k, 0, idx->flag & MM_I_HPC, &mv);
    for (i = 0; i < mv.n; ++i) {
        int t, j;
        const uint64_t *cr = mm_idx_get(idx, mv.a[i].x >> 8, &t);
        if (t >= mopt.mid_occ) continue;
        for (j = 0; j < t; ++j) if ((cr[j] & 1) == (mv.a[i].y & 1)) ++n;
    }
    free(mv.a);
    mm_idx_destroy(idx);
    return n;
}

/* mm_sketch (minimap2/sketch.c:77) -> flat (x,y) pairs. returns count. */
int ref_mm_sketch(const char *s, int len, int w, int k, uint32_t rid, int is_hpc, uint64_t *xy, int cap)
{
    mm128_v p = {0, 0, 0};
    int i, n;
    mm_sketch(0, s, len, w, k, rid, is_hpc, &p);
    n = (int)p.n;
    for (i = 0; i < n && i < cap; ++i) { xy[2 * i] = p.a[i].x; xy[2 * i + 1] = p.a[i].y; }
    free(p.a);
    return n;
}

/* mid_occ for a single-sequence index (mm_idx_str + mm_mapopt_update). */
int ref_mm_mid_occ(const char *ref, int k, int w)
{
    mm_idxopt_t iopt;
    mm_mapopt_t mopt;
    mm_idx_t *idx;
    int r;
    mm_set_opt(0, &iopt, &mopt);
    idx = mm_idx_str(w, k, 0, 14, 1, &ref, NULL);
    mm_mapopt_update(&mopt, idx);
    r = mopt.mid_occ;
    mm_idx_destroy(idx);
    return r;
}

/* mm_chain_dp (minimap2/chain.c:22) on caller-provided anchors (sorted as
 * collect_seed_hits leaves them).  a is modified/reallocated by the callee, so
 * it is copied in with kmalloc(km=0 -> malloc).  Outputs: n_u chains in u[],
 * reordered anchors in a_out. */
int ref_mm_chain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                    float gap_scale, int is_cdna, int n_segs, int64_t n, const uint64_t *axy, uint64_t *u_out, uint64_t *a_out, int64_t *n_a_out)
{
    mm128_t *a = (mm128_t *)malloc((n ? n : 1) * sizeof(mm128_t));
    uint64_t *u = 0;
    int n_u = 0, i, j, na = 0;
    for (i = 0; i < n; ++i) { a[i].x = axy[2 * i]; a[i].y = axy[2 * i + 1]; }
    a = mm_chain_dp(max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc, gap_scale, is_cdna, n_segs, n, a, &n_u, &u, 0);
    for (i = 0; i < n_u; ++i) { u_out[i] = u[i]; na += (int32_t)u[i]; }
    for (j = 0; j < na; ++j) { a_out[2 * j] = a[j].x; a_out[2 * j + 1] = a[j].y; }
    *n_a_out = na;
    free(u);
    free(a);
    return n_u;
}

typedef struct {
    uint32_t max; int32_t zdropped;
    int32_t max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} ref_ez_t;

/* ksw_extd2_sse (minimap2/ksw2_extd2_sse.c:34) through the library's own
 * dispatcher, with the scoring matrix of ksw_gen_simple_mat (align.c:9-22). */
void ref_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int a, int b, int sc_ambi,
                   int q, int e, int q2, int e2, int w, int zdrop, int end_bonus, int flag,
                   ref_ez_t *out, uint32_t *cigar, int cigar_cap)
{
    int8_t mat[25];
    int i, j;
    ksw_extz_t ez;
    a = a < 0 ? -a : a; b = b > 0 ? -b : b; sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
    for (i = 0; i < 4; ++i) { for (j = 0; j < 4; ++j) mat[i * 5 + j] = i == j ? a : b; mat[i * 5 + 4] = sc_ambi; }
    for (j = 0; j < 5; ++j) mat[4 * 5 + j] = sc_ambi;
    memset(&ez, 0, sizeof(ez));
    ksw_extd2_sse(0, qlen, query, tlen, target, 5, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, &ez);
    out->max = ez.max; out->zdropped = ez.zdropped; out->max_q = ez.max_q; out->max_t = ez.max_t;
    out->mqe = ez.mqe; out->mqe_t = ez.mqe_t; out->mte = ez.mte; out->mte_q = ez.mte_q; out->score = ez.score;
    out->n_cigar = ez.n_cigar; out->reach_end = ez.reach_end;
    for (i = 0; i < ez.n_cigar && i < cigar_cap; ++i) cigar[i] = ez.cigar[i];
    kfree(0, ez.cigar);
}

/* radix_sort_128x / radix_sort_64 (misc.c:153-159) -- tie order matters. */
void ref_radix_sort_128x(uint64_t *xy, int64_t n) { radix_sort_128x((mm128_t *)xy, (mm128_t *)xy + n); }
void ref_radix_sort_64(uint64_t *x, int64_t n) { radix_sort_64(x, x + n); }


/* The anchors mm_map_frag hands to mm_chain_dp (minimap2/map.c:293-303: collect_minimizers + collect_seed_hits, sorted by
 * radix_sort_128x), for the call sequence of ConsensusGraph::alignRead.  collect_seed_hits is static, so the anchors are taken from the
 * library's own debug print (mm_dbg_flag & MM_DBG_PRINT_SEED, map.c:298-303: "SD" lines carry the low 32 bits of x, the strand, the low
 * 32 bits of y and the span byte, in array order; "RS" the repetitive length): stderr is pointed at a temporary file for the duration
 * of the call.  xy receives x = strand << 63 | reference position, y = span << 32 | query position.  Returns the number of anchors
 * (-1: no temporary file). */
extern int mm_dbg_flag;
int64_t ref_mm_seeds(const char *ref, int rl, const char *qry, int ql, int k, int w, int max_chain_iter, uint64_t *xy, int64_t cap, int32_t *mid_occ, int32_t *rep_len)
{
    ref_aln_t aln;
    char line[512];
    int64_t n = 0;
    int saved, tmpfd, old_flag = mm_dbg_flag;
    FILE *tmp = tmpfile();
    if (!tmp) return -1;
    tmpfd = fileno(tmp);
    fflush(stderr);
    saved = dup(2);
    dup2(tmpfd, 2);
    mm_dbg_flag |= 4;                       /* MM_DBG_PRINT_SEED */
    ref_mm2_align(ref, rl, qry, ql, k, w, max_chain_iter, &aln, 0, 0);
    mm_dbg_flag = old_flag;
    fflush(stderr);
    dup2(saved, 2);
    close(saved);
    *mid_occ = aln.mid_occ;
    *rep_len = -1;
    rewind(tmp);
    while (fgets(line, sizeof(line), tmp)) {
        if (line[0] == 'R' && line[1] == 'S') { *rep_len = atoi(line + 3); continue; }
        if (line[0] == 'S' && line[1] == 'D') {
            char name[64], strand;
            int x, y, span, d;
            if (sscanf(line, "SD\t%63s\t%d\t%c\t%d\t%d\t%d", name, &x, &strand, &y, &span, &d) != 6) continue;
            if (n < cap) { xy[2 * n] = (uint64_t)(uint32_t)x | (strand == '-' ? 1ull << 63 : 0); xy[2 * n + 1] = (uint64_t)span << 32 | (uint32_t)y; }
            ++n;
        }
    }
    fclose(tmp);
    return n;
}
