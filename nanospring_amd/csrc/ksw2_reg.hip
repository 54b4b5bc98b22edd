// ksw2_reg.hip -- a14h, second generation: ksw_extd2 (minimap2/ksw2_extd2_sse.c:34-401, ksw2.h:103-176) with the DP state
// in REGISTERS.
//
// minimap2's formulation keeps u, v, x, y, x2, y2 per TARGET position t; anti-diagonal r updates t in [st, en] from
// (u, y, y2)[t] and (x, v, x2)[t - 1] of the previous anti-diagonal.  So a lane that owns a FIXED set of t's never has to
// move its state: it lives in VGPRs for the whole problem, the (t - 1) neighbour arrives by one DPP lane shift, and cells
// outside [st, en] simply keep their registers -- which is exactly the reference's "stale array content" that the
// lane-exact semantics need (the SSE kernel sweeps 16-aligned blocks; out-of-band cells of the first / last block are
// computed from stale values and feed in-band cells once the band binds).
//
//   * 2 cells per lane, packed 2 x int16 per VGPR, all arithmetic in v_pk_*_i16 (one instruction = 128 cells per wave);
//   * lane l of wave w owns t = (c * NW + w) * 128 + 2 l + {0, 1} for c = 0 .. NCH-1 (statically indexed register sets);
//   * NW = 1: one wave per problem, no LDS traffic for state at all (LDS holds only the reversed query);
//     NW > 1 (long, latency-bound problems): the waves of a workgroup split the anti-diagonal; only the seam cell between
//     two waves goes through LDS, ONE barrier per anti-diagonal, the per-row score bookkeeping (exact max / Z-drop) lags
//     one row behind so that it never adds a second barrier;
//   * int8 wrap-around: with |scores| and gap costs as small as minimap2's, the only values that can leave int8 are
//     a, b, a2, b2 after the "- (z - q)" adjustment, in out-of-band garbage cells whose x2 / y2 grow row after row
//     (checked over thousands of problems with an instrumented oracle); those four are wrapped
//     ((v << 8) >> 8 per half), everything else provably stays inside int8 and needs no emulation.  The host routes a
//     problem here only if its parameters keep that proof valid (ksw_reg_eligible) -- anything else runs on ksw2.hip.
//   * the score row s[] lives in registers too (stale outside the reference's 16-byte score stores of the row).
//
// The traceback matrix p (1 B per computed cell, the reference's layout so that the backtrack is index compatible) is
// scratch in HBM; wave 0 walks it at the end (backtrack_and_store_wave).
#include "common.hpp"
#include "ksw2.hpp"
#include "ksw_class.hpp"
#include "ksw_collect.hpp"
#include <mutex>
#include "host_util.hpp"

namespace nsgpu {

#define KSW_NEG_INF (-0x40000000)
#define KSW_EZ_SCORE_ONLY 0x01
#define KSW_EZ_RIGHT 0x02
#define KSW_EZ_APPROX_MAX 0x08
#define KSW_EZ_APPROX_DROP 0x10
#define KSW_EZ_EXTZ_ONLY 0x40
#define KSW_EZ_REV_CIGAR 0x80
#define KSW_EZ_NS_SERIAL_BACKTRACK 0x20000   // debugging aid / A-B switch (NSGPU_KSW_SERIAL_BACKTRACK=1): one lane walks the traceback
#define KSW_EZ_NS_ALL_BOOKS 0x40000          // A-B switch (NSGPU_KSW_ALL_BOOKS=1): approx mode, several waves: every wave keeps the books
#define KSW_EZ_NS_EARLY_EXIT 0x10000      // not minimap2's: set by the host code of this library (ksw2.hip) unless NSGPU_KSW_NO_EARLY_EXIT
// not minimap2's either: the caller does not read ez.score of this approximate-mode problem (KSW_EZ_APPROX_MAX without KSW_EZ_APPROX_DROP).  The
// score is all that the per-row books of that mode produce -- the greedy H0 walk (ksw2_extd2_sse.c:367-383) decides nothing: no Z-drop, no
// maximum, and the backtrack starts at the matrix's corner -- so the kernels skip the books and return score 0.  The aligner's gap fills carry it
// (align_batch.hip / plan.hip): mm_align1 only ever adds a gap fill's score into dp_score, which NanoSpring never looks at (src/ConsensusGraph.cpp:
// 219-397 reads rs, re, qs, qe, blen, mlen, n_ambi and the CIGAR); the public nsgpu_ksw_extd2_batch passes callers' flags through untouched.
#define KSW_EZ_NS_NO_SCORE 0x80000
// not minimap2's either: the caller does not read ez.mte / ez.mte_q / ez.score of this exact-mode extension (KSW_EZ_EXTZ_ONLY).  mm_align1 reads
// max / max_t / max_q, mqe_t, reach_end, zdropped and the CIGAR of an extension (align.c:700-780), never the best score on the target's last
// column -- and that column is all the sweep still works for once the query has ended on every diagonal that can matter: see the second early
// exit in ksw_reg_run.  The aligner's extensions carry the flag (align_batch.hip / plan.hip); nsgpu_ksw_extd2_batch passes callers' flags through.
#define KSW_EZ_NS_NO_MTE 0x100000

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ s2 S2(int x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ s2 S2u(u2 x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ u2 U2(s2 x) { return __builtin_bit_cast(u2, x); }
__device__ __forceinline__ int I32(s2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ s2 splat(int v) { const short s = (short)v; return (s2){s, s}; }
__device__ __forceinline__ s2 pmax(s2 a, s2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2 pmin(s2 a, s2 b) { return __builtin_elementwise_min(a, b); }
// min(x, 1) per unsigned half: 0 stays 0, anything else becomes 1.  Through inline asm: the compiler otherwise recognises the
// pattern as a compare and scalarises every use into v_cmp (SDWA) + v_cndmask pairs per half (measured: 126 VALU instead of ~75 per
// 128 cells).
__device__ __forceinline__ u2 to01(u2 x) { u2 r; asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(x)); return r; }
// 0 where the halves are equal, 1 where m > c (m >= c required)
__device__ __forceinline__ u2 ne01(s2 m, s2 c) { return to01(U2((s2)(m - c))); }
// 0xffff in the halves where lo <= t <= hi (all values < 2^15), else 0
__device__ __forceinline__ uint32_t range_mask(s2 t, int lo, int hi)
{
    const s2 a = t - splat(lo), b = splat(hi) - t;                 // both >= 0 inside
    const s2 o = S2(I32(a) | I32(b));
    return ~(uint32_t)I32((s2)(o >> (s2){15, 15}));
}
__device__ __forceinline__ s2 wrap8(s2 v) { return (s2)((s2)(v << (s2){8, 8}) >> (s2){8, 8}); }
// (mask & a) | (~mask & b)
__device__ __forceinline__ s2 sel(uint32_t mask, s2 a, s2 b) { return S2((int)((mask & (uint32_t)I32(a)) | (~mask & (uint32_t)I32(b)))); }
// lane l receives lane l-1's value; lane 0 receives `first` (wave_shr:1, bound_ctrl off keeps `old` in lane 0)
__device__ __forceinline__ int shr1(int first, int v) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); }
// {lo: the upper half of prev, hi: the lower half of own}
__device__ __forceinline__ s2 left_nb(s2 own, int prev) { return S2((int)__builtin_amdgcn_alignbit((uint32_t)I32(own), (uint32_t)prev, 16)); }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    // DPP reduction: quad, row, then across rows; every lane of row 3 / lane 63 ends with the maximum of the wave
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  v = v > t ? v : t;     // quad_perm [1,0,3,2]
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  v = v > t ? v : t;     // quad_perm [2,3,0,1]
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); v = v > t ? v : t;     // row_half_mirror
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); v = v > t ? v : t;     // row_mirror
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true); v = v > t ? v : t;     // row_bcast15 -> rows 1, 3
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true); v = v > t ? v : t;     // row_bcast31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct RowRange { int st0, en0, st, en; bool empty; };
__device__ __forceinline__ RowRange row_range(int r, int qlen, int tlen, int w)
{
    RowRange o;
    int st = 0, en = tlen - 1;
    if (st < r - qlen + 1) st = r - qlen + 1;
    if (en > r) en = r;
    if (st < ((r - w + 1) >> 1)) st = (r - w + 1) >> 1;
    if (en > ((r + w) >> 1)) en = (r + w) >> 1;
    o.empty = st > en;
    o.st0 = st, o.en0 = en;
    o.st = st / 16 * 16, o.en = (en + 16) / 16 * 16 - 1;
    return o;
}

// everything a row needs that is the same for all lanes
struct Consts {
    int q, e, q2, e2, qe, qe2, sc_mch, sc_mis, sc_N, long_thres, long_diff;
};

// One anti-diagonal for the two cells of a lane.  Inputs: score z, own (u, y, y2), left neighbour's (x, v, x2) -- all of
// the previous anti-diagonal.  ksw2_extd2_sse.c:228-322 (left-aligned gaps) / the KSW_EZ_RIGHT twin.
template <bool RIGHT>
__device__ __forceinline__ void cell_pair(const Consts &K, s2 z, s2 ut, s2 yo, s2 y2o, s2 xt1, s2 vt1, s2 x2t1, s2 &un, s2 &vn, s2 &xn, s2 &yn,
                                          s2 &x2n, s2 &y2n, uint32_t &dbytes)
{
    s2 a = xt1 + vt1, b = yo + ut, a2 = x2t1 + vt1, b2 = y2o + ut;
    const s2 m = pmax(pmax(pmax(pmax(z, a), b), a2), b2);
    u2 d;
    const u2 one = (u2){1, 1};
    if (!RIGHT) {      // strict '>' at every step: the FIRST of (z, a, b, a2, b2) that reaches the maximum
        const u2 n0 = ne01(m, z), n1 = ne01(m, a), n2 = ne01(m, b), n3 = ne01(m, a2);
        d = n0 * (one + n1 * (one + n2 * (one + n3)));
    } else {           // '>=' at every step: the LAST one that reaches it
        const u2 n1 = ne01(m, a), n2 = ne01(m, b), n3 = ne01(m, a2), n4 = ne01(m, b2);
        d = (u2){4, 4} - n4 * (one + n3 * (one + n2 * (one + n1)));
    }
    const s2 zc = pmin(m, splat(K.sc_mch));
    un = zc - vt1, vn = zc - ut;
    const s2 t1 = zc - splat(K.q), t2 = zc - splat(K.q2);
    a = wrap8(a - t1), b = wrap8(b - t1), a2 = wrap8(a2 - t2), b2 = wrap8(b2 - t2);
    const s2 zero = splat(0);
    const s2 pa = pmax(a, zero), pb = pmax(b, zero), pa2 = pmax(a2, zero), pb2 = pmax(b2, zero);
    xn = pa - splat(K.qe), yn = pb - splat(K.qe), x2n = pa2 - splat(K.qe2), y2n = pb2 - splat(K.qe2);
    u2 fa, fb, fa2, fb2;
    if (!RIGHT) {      // a > 0
        fa = to01(U2(pa)), fb = to01(U2(pb)), fa2 = to01(U2(pa2)), fb2 = to01(U2(pb2));
    } else {           // a >= 0
        const s2 o = splat(1);
        fa = to01(U2(pmax(a + o, zero))), fb = to01(U2(pmax(b + o, zero)));
        fa2 = to01(U2(pmax(a2 + o, zero))), fb2 = to01(U2(pmax(b2 + o, zero)));
    }
    d = d + fa * (u2){8, 8} + fb * (u2){16, 16} + fa2 * (u2){32, 32} + fb2 * (u2){64, 64};
    dbytes = __builtin_amdgcn_perm(0u, __builtin_bit_cast(uint32_t, d), 0x0c0c0200u);        // the two low bytes of the halves
}

// score of the two cells: match / mismatch, either base = 4 -> sc_N (ksw2_extd2_sse.c:165-184)
__device__ __forceinline__ s2 score_pair(const Consts &K, s2 tq, s2 tt)
{
    const s2 x = S2(I32(tq) ^ I32(tt));
    const u2 ne = to01(U2(x));
    s2 z = S2u(ne * U2(splat(K.sc_mis - K.sc_mch))) + splat(K.sc_mch);
    const u2 isn = U2(S2((I32(tq) | I32(tt)) >> 2 & 0x00010001));
    z = z + S2u(isn * U2((s2)(splat(K.sc_N) - z)));
    return z;
}

__device__ __forceinline__ int key_rank(int t, int st0, int en0, int en1)
{
    // tie order of the reference's 4-lane scan (ksw2_extd2_sse.c:323-358): en0 first, then lane (t - st0) & 3 by ascending t, then the tail
    const int cls = t == en0 ? 0 : t < en1 ? 1 + ((t - st0) & 3) : 5;
    return cls << 13 | t;
}

// The value of cell t held as half (t & 1) of lane (t >> 1) & 63 of register set (t >> 7) % ... : NW == 1 only.
template <int NCH>
__device__ __forceinline__ int fetch16(const s2 (&R)[NCH], int t)
{
    const int c = t >> 7, l = (t >> 1) & 63;        // a cell beyond the register sets reads as 0 (never on in-band shapes)
    int v = 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) { const int x = __builtin_amdgcn_readlane(I32(R[k]), l); v = k == c ? x : v; }
    return (int)(short)(t & 1 ? v >> 16 : v & 0xffff);
}
template <int NCH>
__device__ __forceinline__ int fetch32(const int (&Rlo)[NCH], const int (&Rhi)[NCH], int t)
{
    const int c = t >> 7, l = (t >> 1) & 63;
    int v = 0;
#pragma unroll
    for (int k = 0; k < NCH; ++k) { const int x = __builtin_amdgcn_readlane(t & 1 ? Rhi[k] : Rlo[k], l); v = k == c ? x : v; }
    return v;
}

__device__ void backtrack_and_store(const KswTask &tk, int w, int ncol16, const uint8_t *p, uint32_t *cig_pool, KswResult *res_out, int ez_max, int ez_zdropped,
                                    int ez_max_q, int ez_max_t, int ez_mqe, int ez_mqe_t, int ez_mte, int ez_mte_q, int ez_score)
{
    // ksw2_extd2_sse.c:389-398 + ksw_backtrack (ksw2.h:119-151, is_rot = 1); one lane
    const int qlen = tk.qlen, tlen = tk.tlen, flag = tk.flag;
    int ez_reach_end = 0;
    uint32_t n_cigar = 0;
    if (!(flag & KSW_EZ_SCORE_ONLY)) {
        int i0 = -1, j0 = -1;
        if (!ez_zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
        else if (!ez_zdropped && (flag & KSW_EZ_EXTZ_ONLY) && ez_mqe + tk.end_bonus > ez_max) ez_reach_end = 1, i0 = ez_mqe_t, j0 = qlen - 1;
        else if (ez_max_t >= 0 && ez_max_q >= 0) i0 = ez_max_t, j0 = ez_max_q;
        if (i0 >= 0 && j0 >= 0) {
            uint32_t *cig = cig_pool + tk.cig_off;
            int i = i0, j = j0, state = 0;
            uint32_t cur_op = 0xffffffffu, cur_len = 0;
            while (i >= 0 && j >= 0) {
                const int r = i + j;
                const RowRange rr = row_range(r, qlen, tlen, w);
                int force_state = -1;
                if (i < rr.st) force_state = 2;
                if (i > rr.en) force_state = 1;
                const uint32_t tmp = force_state < 0 ? p[(size_t)r * ncol16 + i - rr.st] : 0u;
                if (state == 0) state = tmp & 7;
                else if (!(tmp >> (state + 2) & 1)) state = 0;
                if (state == 0) state = tmp & 7;
                if (force_state >= 0) state = force_state;
                uint32_t op;
                if (state == 0) op = 0, --i, --j;
                else if (state == 1 || state == 3) op = 2, --i;
                else op = 1, --j;
                if (op == cur_op) ++cur_len;
                else { if (cur_len) cig[n_cigar++] = cur_len << 4 | cur_op; cur_op = op, cur_len = 1; }
            }
            if (i >= 0) { if (cur_op == 2) cur_len += i + 1; else { if (cur_len) cig[n_cigar++] = cur_len << 4 | cur_op; cur_op = 2, cur_len = i + 1; } }
            if (j >= 0) { if (cur_op == 1) cur_len += j + 1; else { if (cur_len) cig[n_cigar++] = cur_len << 4 | cur_op; cur_op = 1, cur_len = j + 1; } }
            if (cur_len) cig[n_cigar++] = cur_len << 4 | cur_op;
            if (!(flag & KSW_EZ_REV_CIGAR))
                for (uint32_t a = 0; a < n_cigar >> 1; ++a) { const uint32_t t_ = cig[a]; cig[a] = cig[n_cigar - 1 - a]; cig[n_cigar - 1 - a] = t_; }
        }
    }
    KswResult o;
    o.max = (uint32_t)ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t; o.mqe = ez_mqe; o.mqe_t = ez_mqe_t;
    o.mte = ez_mte; o.mte_q = ez_mte_q; o.score = ez_score; o.n_cigar = (int)n_cigar; o.reach_end = ez_reach_end;
    res_out[tk.out_idx] = o;
}

// The same walk by one WAVE.  The serial walk is a chain of dependent one-byte loads (one L2 round trip per step, 500-1500 steps for a
// problem of the multi-wave classes).  Here every round the 64 lanes load the traceback bytes of the next 64 cells ALONG THE DIRECTION THE
// WALK HAS (diagonal in state 0, up in a deletion, left in an insertion) in one go; the walk then consumes them, in order and with the
// reference's logic step by step, for as long as its moves follow that direction -- a run of matches, a long gap -- and starts the next
// round where it left it.  One round trip per run instead of per step; the steps, hence the CIGAR, are the serial walk's.
__device__ void backtrack_and_store_wave(const KswTask &tk, int w, int ncol16, const uint8_t *p, uint32_t *cig_pool, KswResult *res_out, int ez_max, int ez_zdropped,
                                         int ez_max_q, int ez_max_t, int ez_mqe, int ez_mqe_t, int ez_mte, int ez_mte_q, int ez_score, int lane)
{
    const int qlen = tk.qlen, tlen = tk.tlen, flag = tk.flag;
    int ez_reach_end = 0;
    uint32_t n_cigar = 0;
    if (!(flag & KSW_EZ_SCORE_ONLY)) {
        int i0 = -1, j0 = -1;
        if (!ez_zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
        else if (!ez_zdropped && (flag & KSW_EZ_EXTZ_ONLY) && ez_mqe + tk.end_bonus > ez_max) ez_reach_end = 1, i0 = ez_mqe_t, j0 = qlen - 1;
        else if (ez_max_t >= 0 && ez_max_q >= 0) i0 = ez_max_t, j0 = ez_max_q;
        if (i0 >= 0 && j0 >= 0) {
            uint32_t *cig = cig_pool + tk.cig_off;
            int i = i0, j = j0, state = 0;
            uint32_t cur_op = 0xffffffffu, cur_len = 0;
            while (i >= 0 && j >= 0) {
                const int di = state == 2 || state == 4 ? 0 : 1, dj = state == 1 || state == 3 ? 0 : 1;      // the direction of this round
                uint32_t mine = 0;
                {
                    const int li = i - lane * di, lj = j - lane * dj;
                    if (li >= 0 && lj >= 0) {
                        const int r = li + lj;
                        const RowRange rr = row_range(r, qlen, tlen, w);
                        int force_state = -1;
                        if (li < rr.st) force_state = 2;
                        if (li > rr.en) force_state = 1;
                        mine = (force_state < 0 ? (uint32_t)p[(size_t)r * ncol16 + li - rr.st] : 0u) | (uint32_t)(force_state + 1) << 8;
                    }
                }
                for (int l = 0; l < 64 && i >= 0 && j >= 0; ++l) {
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)mine, l);
                    const uint32_t tmp = x & 0xffu;
                    const int force_state = (int)(x >> 8) - 1;
                    if (state == 0) state = tmp & 7;
                    else if (!(tmp >> (state + 2) & 1)) state = 0;
                    if (state == 0) state = tmp & 7;
                    if (force_state >= 0) state = force_state;
                    uint32_t op;
                    int mi = 0, mj = 0;
                    if (state == 0) op = 0, mi = mj = 1;
                    else if (state == 1 || state == 3) op = 2, mi = 1;
                    else op = 1, mj = 1;
                    i -= mi, j -= mj;
                    if (op == cur_op) ++cur_len;
                    else { if (cur_len) { if (lane == 0) cig[n_cigar] = cur_len << 4 | cur_op; ++n_cigar; } cur_op = op, cur_len = 1; }
                    if (mi != di || mj != dj) break;                      // the walk turned: the other lanes' bytes are not on its way
                }
            }
            if (i >= 0) { if (cur_op == 2) cur_len += i + 1; else { if (cur_len) { if (lane == 0) cig[n_cigar] = cur_len << 4 | cur_op; ++n_cigar; } cur_op = 2, cur_len = i + 1; } }
            if (j >= 0) { if (cur_op == 1) cur_len += j + 1; else { if (cur_len) { if (lane == 0) cig[n_cigar] = cur_len << 4 | cur_op; ++n_cigar; } cur_op = 1, cur_len = j + 1; } }
            if (cur_len) { if (lane == 0) cig[n_cigar] = cur_len << 4 | cur_op; ++n_cigar; }
            if (!(flag & KSW_EZ_REV_CIGAR) && lane == 0)
                for (uint32_t a = 0; a < n_cigar >> 1; ++a) { const uint32_t t_ = cig[a]; cig[a] = cig[n_cigar - 1 - a]; cig[n_cigar - 1 - a] = t_; }
        }
    }
    if (lane == 0) {
        KswResult o;
        o.max = (uint32_t)ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t; o.mqe = ez_mqe; o.mqe_t = ez_mqe_t;
        o.mte = ez_mte; o.mte_q = ez_mte_q; o.score = ez_score; o.n_cigar = (int)n_cigar; o.reach_end = ez_reach_end;
        res_out[tk.out_idx] = o;
    }
}

// LDS layout of a problem (bytes):
//   [0, QB)        qA: T zero bytes, the reversed query, zero bytes up to QB   (QB = 2 T + round16(qlen) + 16)
//   [QB, 2 QB)     qB: qA shifted by one byte (qB[j] = qA[j + 1]), so that the two query bases of a lane are always ONE aligned 16-bit read
//   NW > 1 only:   seam[2][NB] {x|v|x2 as 3 x int16 in 8 bytes, H} ; pub[2] {key[NW], h_en0, h_st0} ; uv[2][T] (approx: u16 | v16 per cell)
// T = NW * NCH * 128 cells.
__host__ __device__ inline int reg_qb(int T, int qlen) { return 2 * T + (qlen + 15) / 16 * 16 + 16; }

// -------------------------------------------------------------------------------------------------------------------
// The row loop.  NW waves, NCH register sets per lane.
// -------------------------------------------------------------------------------------------------------------------
// Per-row bookkeeping on the VECTOR unit.  The values are the same in every lane, but on this chip a scalar instruction
// costs a wave twice the issue time of a vector one (one scalar unit per CU against four SIMD-32s), and the first version
// of this kernel, with ~300 scalar instructions per anti-diagonal, was bound by exactly that.  vreg() hides a value from
// the compiler's uniformity analysis so that what is derived from it stays in VGPRs; decisions come back to the scalar
// unit through one v_readfirstlane per row.
// -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int vreg(int x) { asm volatile("" : "+v"(x)); return x; }

struct Ez { int max, max_t, max_q, mte, mte_q, mqe, mqe_t; };

// ksw_apply_zdrop (ksw2.h:160-176, is_rot = 1) in select form; returns 1 when the row Z-drops
__device__ __forceinline__ int zdrop_row(Ez &z, int r, int max_H, int max_t, int zdrop, int e2)
{
    const bool better = max_H > z.max;
    const int tl = max_t - z.max_t, ql = (r - max_t) - z.max_q;
    const int l = tl > ql ? tl - ql : ql - tl;
    const bool drop = !better && max_t >= z.max_t && r - max_t >= z.max_q && zdrop >= 0 && z.max - max_H > zdrop + l * e2;
    z.max_t = better ? max_t : z.max_t, z.max_q = better ? r - max_t : z.max_q, z.max = better ? max_H : z.max;
    return drop ? 1 : 0;
}

// -------------------------------------------------------------------------------------------------------------------
// The row loop.  NW waves, NCH register sets per lane.
//   APPROX        KSW_EZ_APPROX_MAX (gap fills): no per-row maximum
//   FAST          APPROX, no KSW_EZ_APPROX_DROP and a band that never binds (w >= qlen + tlen): every cell is a true DP
//                 cell, u and v are differences of ONE well-defined score matrix, so the score the reference accumulates
//                 along its greedy H0 path equals the sum along ANY path -- here the matrix's left column (v of cell 0 on
//                 rows < qlen) and then its bottom row (u of cell st0 on the rows after), which each lane adds up for its own
//                 cells: no per-row bookkeeping at all
// -------------------------------------------------------------------------------------------------------------------
template <int NW, int NCH, bool APPROX, bool RIGHT, bool FAST>
__device__ void ksw_reg_run(const KswTask &tk, const KswParams &pr, const uint8_t *__restrict__ seqs, uint8_t *__restrict__ p_pool,
                            uint32_t *__restrict__ cig_pool, KswResult *__restrict__ res_out, uint8_t *lds)
{
    static_assert(!FAST || (APPROX && NW == 1), "the path-independent score is used by the one-wave gap-fill classes");
    // (a variant with a wave that owns no cells and only keeps the books -- an odd wave count -- was measured in round 3 and did not pay:
    // the bookkeeping of row r - 1 already runs behind the barrier of row r; removed.  Every wave computes.)
    constexpr int NWC = NW;                      // computing waves
    constexpr int T = NWC * NCH * 128;
    constexpr int NB = NWC * NCH;                // blocks of 128 cells
    const int lane = threadIdx.x & 63, wv = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int cw = wv;                           // index among the computing waves
    constexpr bool computes = true;
    const int qlen = tk.qlen, tlen = tk.tlen, flag = tk.flag, zdrop = tk.zdrop;
    Consts K;
    K.q = pr.q, K.e = pr.e, K.q2 = pr.q2, K.e2 = pr.e2;
    if (K.q2 + K.e2 < K.q + K.e) { int t_ = K.q; K.q = K.q2; K.q2 = t_; t_ = K.e; K.e = K.e2; K.e2 = t_; }
    K.qe = K.q + K.e, K.qe2 = K.q2 + K.e2;
    K.sc_mch = pr.sc_mch, K.sc_mis = pr.sc_mis;
    K.sc_N = pr.sc_ambi == 0 ? -K.e2 : pr.sc_ambi;
    K.long_thres = K.e != K.e2 ? (K.q2 - K.q) / (K.e - K.e2) - 1 : 0;
    if (K.q2 + K.e2 + K.long_thres * K.e2 > K.q + K.e + K.long_thres * K.e) ++K.long_thres;
    K.long_diff = K.long_thres * (K.e - K.e2) - (K.q2 - K.q) - K.e2;
    int w = tk.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    const int ncol16 = n_col_ * 16;

    const int QB = reg_qb(T, qlen);
    uint8_t *qA = lds, *qB = lds + QB;
    // NW > 1 scratch (see reg_lds_bytes)
    uint2 *seam = reinterpret_cast<uint2 *>(lds + 2 * QB);                 // [2][NB] {x | v << 16, x2}: (x, v, x2) of a block's last cell
    int *seam_h = reinterpret_cast<int *>(lds + 2 * QB + 2 * NB * 8);      // [2][NB] its H
    uint32_t *pub = reinterpret_cast<uint32_t *>(lds + 2 * QB + 2 * NB * 12);   // [3][NW + 2]: per-wave best key, H[en0], H[st0] of a row
    uint32_t *uv4 = pub + 3 * (NW + 2);                                    // [3][4] approx: u | v << 16 of cells L .. L + 3 (L = last_H0_t two rows back)
    uint32_t *stop_flag = uv4 + 12;                                        // exact, NW > 1: wave 0 (the only one that keeps the books) saw a Z-drop
    // approx, NW > 1 (ROT below): the books' state after each row, [2][8] {last_H0_t, stop, H0, max, max_t, max_q, score}, 16-byte aligned
    int *bst = reinterpret_cast<int *>(lds + 2 * QB + ((2 * NB * 12 + 3 * (NW + 2) * 4 + 52 + 15) & ~15));

    for (int i = threadIdx.x; i < (2 * QB) / 4; i += NW * 64) reinterpret_cast<uint32_t *>(lds)[i] = 0;
    for (int i = threadIdx.x; i < 2 * NB * 3 + 3 * (NW + 2) + 13 + 20; i += NW * 64) reinterpret_cast<uint32_t *>(lds + 2 * QB)[i] = 0;
    if (NW > 1) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
        const uint8_t *query = seqs + tk.qoff;
        for (int j = threadIdx.x; j < qlen; j += NW * 64) {
            const uint8_t b = query[qlen - 1 - j];
            qA[T + j] = b;
            qB[T + j - 1] = b;
        }
    }
    // register state
    s2 TT[NCH], TP[NCH], SC[NCH], U[NCH], V[NCH], X[NCH], Y[NCH], X2[NCH], Y2[NCH];      // TP: the two t's of the lane
    s2 ACC[NCH];                                                                         // FAST: the lane's share of the score
    int HL[NCH], HH[NCH];
    {
        const uint8_t *target = seqs + tk.toff;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t0 = computes ? (c * NWC + cw) * 128 + 2 * lane : T;      // (the books wave: beyond every row)
            const int b0 = t0 < tlen ? target[t0] : 0, b1 = t0 + 1 < tlen ? target[t0 + 1] : 0;
            TT[c] = S2(b0 | b1 << 16);
            TP[c] = S2(t0 | (t0 + 1) << 16);
            SC[c] = splat(0), ACC[c] = splat(0);
            U[c] = V[c] = X[c] = Y[c] = splat(-K.q - K.e);
            X2[c] = Y2[c] = splat(-K.q2 - K.e2);
            HL[c] = HH[c] = KSW_NEG_INF;
        }
    }
    if (NW > 1) {
        // seams of the initial state: every block's last cell holds the initial values
        for (int i = threadIdx.x; i < 2 * NB; i += NW * 64) {
            const uint32_t b0 = (uint32_t)(-K.q - K.e) & 0xffffu, b1 = (uint32_t)(-K.q2 - K.e2) & 0xffffu;
            seam[i] = make_uint2(b0 | b0 << 16, b1);
            seam_h[i] = KSW_NEG_INF;
        }
        __syncthreads();
    } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    Ez z;
    z.max = vreg(0), z.max_t = vreg(-1), z.max_q = vreg(-1), z.mte = vreg(KSW_NEG_INF), z.mte_q = vreg(-1), z.mqe = vreg(KSW_NEG_INF), z.mqe_t = vreg(-1);
    int ez_zdropped = 0, ez_score = KSW_NEG_INF;
    uint8_t *p = p_pool + tk.p_off;
    int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
    const int n_rows = qlen + tlen - 1, c1 = qlen - 1, c2 = tlen - 1;
    // NW > 1: the bookkeeping of row r - 1 is done after the barrier of row r (a lag of one row, so that it never costs a
    // second barrier); these describe the row that is still owed
    int lag_r = -1, lag_st0 = 0, lag_en0 = 0, lag_en = 0, lag_L = 0;
    bool brk = false;
    // Approx mode with several waves: the books of a row (the reference's greedy H0 walk: ~100 instructions and two LDS round trips) are
    // kept by ONE wave per row instead of by all of them -- a wave's row time is its instruction count, and a row lasts as long as its
    // busiest wave -- namely by the wave that follows the row's last active block, which computes the fewest blocks.  The state travels
    // through LDS (bst[slot]): the books of row r - 1 are done behind the barrier of row r and written to slot r & 1; behind that
    // barrier every wave reads slot (r - 1) & 1 -- the state after row r - 2, complete since the barrier before -- for the stop flag and
    // for last_H0_t, from which the cells to publish for the books two rows on are known (it moves by at most one cell per row, the
    // four published cells cover that: uv4).  KSW_EZ_NS_ALL_BOOKS (NSGPU_KSW_ALL_BOOKS=1) keeps the books in every wave, as before.
    // KSW_EZ_NS_NO_SCORE: the score is all these books produce (no KSW_EZ_APPROX_DROP): nobody keeps them
    const bool nob = APPROX && (flag & KSW_EZ_NS_NO_SCORE) && !(flag & KSW_EZ_APPROX_DROP);
    const bool ROT = APPROX && NW > 1 && !(flag & KSW_EZ_NS_ALL_BOOKS) && !nob;
    int Lpub = 0, Lpub_prev = 0;                        // last_H0_t the current / the previous row's publication is relative to
    auto books_load = [&](const int *sp) { last_H0_t = sp[0], H0 = sp[2], z.max = sp[3], z.max_t = sp[4], z.max_q = sp[5], ez_score = sp[6]; };
    if (ROT) {
        if (threadIdx.x < 2) {
            int *sp = bst + threadIdx.x * 8;
            sp[0] = 0, sp[1] = 0, sp[2] = 0, sp[3] = 0, sp[4] = -1, sp[5] = -1, sp[6] = KSW_NEG_INF;
        }
        __syncthreads();
    }

    // ---- exact mode, rows r > 0 and r == 0 alike: ksw2_extd2_sse.c:359-366 given the row's maximum and H[en0], H[st0] ----
    // ---- exact early exit for extensions that run off the target's end (a read hanging over the end of its contig: qlen in the
    // thousands against a few hundred target bases; these problems set the length of a DP launch) ----
    // Once the target is exhausted, the reference keeps sweeping anti-diagonals until the band runs out (st > en: zdropped = 1, break) --
    // about 2 tlen + w rows, most of them deep in the query's overhang where nothing can happen any more.  A cell (t, q) of row r = t + q
    // with q >= t holds at most  sc_mch (t + 1) - g(q - t),  g(l) = min(q + e l, q2 + e2 l): the score of t + 1 matches and ONE gap of the
    // length difference (gap costs are sub-additive).  That is largest at t = tlen - 1 and falls by at least e2 per row.  When it is
    // below the best last-column score so far (mte <= max) for the NEXT row, no later row can raise max or mte; if, in addition, the band
    // runs out before any row reaches the query's end (row 2 (tlen - 1) + w + 1 <= qlen - 1: mqe, reach_end and the score stay unset)
    // the reference's final state is exactly: zdropped = 1 (by its Z-drop test or by the band), max / mte as they are now, and a
    // backtrack from the maximum, which never looks at a later row.  So the sweep may stop here with that state.  The margin of 32
    // covers the cells along the band's lower edge, whose inputs from outside the band are a row or two old (they sit w off the
    // diagonal, hundreds of points below the bound's cell).  NSGPU_KSW_NO_EARLY_EXIT=1 in the host code switches it off (A/B, tests).
    const bool early_ok = !APPROX && (flag & KSW_EZ_NS_EARLY_EXIT) && 2 * c2 + w + 1 <= c1;
    // ---- the mirror image: an extension whose TARGET window is longer than the query (minimap2 hands the DP about twice the query's length
    // of target, align.c:617-677: every read that ends inside its contig).  Past row 2 (qlen - 1) every cell (t, q) of a row r has
    // t - q >= d = r - 2 (qlen - 1): the query has ended on the main diagonal and the sweep goes on, cell by cell, along diagonals ever further
    // to the right, until the target's last column -- qlen + tlen - 1 rows in all, the last third of them for nothing but ez.mte.  Such a cell
    // holds at most  U(d) = sc_mch qlen - g(d),  g(l) = min(q + e l, q2 + e2 l): qlen matches and the deletions of the length difference in one
    // run (gap costs are sub-additive), falling with every row.  Once U(d0) + 32 <= min(max, mqe) for the NEXT row's d0 (margin as above, for
    // the cells along the band's edge) no later row can raise max or mqe (both need a strictly greater score).  What is left is the Z-drop
    // test: a later row r + j has the cell (r + j - c1, c1) on the query's last row, which holds at least  H(r - c1, c1) - g(j)  (one more
    // deletion run), so max - max_H' <= max - H[st0] + q2 + e2 j, while its threshold is zdrop + e2 l with l >= (d0 - 1 + j) - (max_t - max_q)
    // for whichever cell is the row's maximum: no row drops when  max - H[st0] + q2 + 32 <= zdrop + e2 (d0 - 1 - (max_t - max_q)).  Then the
    // reference's final state is the present one -- zdropped = 0, or 1 when the band runs out before the target does (c2 > (c1 + c2 + w) / 2:
    // its loop ends with `st > en`, a later Z-drop would say the same) -- except for mte / mte_q / score, which the caller has declared unread
    // (KSW_EZ_NS_NO_MTE).  Validated against the oracle on target-longer problems of every kind (tests/test_ksw2_gpu.py).
    const bool tl_exit = !APPROX && (flag & KSW_EZ_NS_NO_MTE) && (flag & KSW_EZ_EXTZ_ONLY) && (flag & KSW_EZ_NS_EARLY_EXIT) && c2 > c1;
    const bool band_out = c2 > ((c1 + c2 + w) >> 1);
    auto exact_row = [&](int r, int st0, int en0, int en, int max_H, int max_t, int h_en0, int h_st0) {
        if (en0 == c2) { const bool up = h_en0 > z.mte; z.mte_q = up ? r - en : z.mte_q, z.mte = up ? h_en0 : z.mte; }
        if (r - st0 == c1) { const bool up = h_st0 > z.mqe; z.mqe_t = up ? st0 : z.mqe_t, z.mqe = up ? h_st0 : z.mqe; }
        int stop = zdrop_row(z, r, max_H, max_t, zdrop, K.e2);
        if (early_ok) {
            const int L = r + 1 - 2 * c2;                       // q - t of the next row's cell on the target's last column
            const int g1 = K.q + K.e * L, g2 = K.q2 + K.e2 * L;
            const int bound = K.sc_mch * tlen - (g1 < g2 ? g1 : g2) + 32;
            stop |= (L >= 1 && bound < z.mte) ? 1 : 0;
        }
        if (__builtin_amdgcn_readfirstlane(stop)) { ez_zdropped = 1; brk = true; }
        else if (tl_exit && r >= 2 * c1 && r - st0 == c1) {
            const int d0 = r + 1 - 2 * c1;
            const int g1 = K.q + K.e * d0, g2 = K.q2 + K.e2 * d0;
            const int ub = K.sc_mch * qlen - (g1 < g2 ? g1 : g2) + 32;
            const int room = d0 - 1 - (z.max_t - z.max_q);
            int done = (ub <= z.max && ub <= z.mqe) ? 1 : 0;
            if (!band_out) done &= (room >= 0 && (zdrop < 0 || z.max - h_st0 + K.q2 + 32 <= zdrop + K.e2 * room)) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(done)) { ez_zdropped = band_out ? 1 : 0; brk = true; }
        }
        if (!brk && r == n_rows - 1 && en0 == c2) ez_score = __builtin_amdgcn_readfirstlane(h_en0);
    };
    // ---- approx mode with the reference's greedy H0 (ksw2_extd2_sse.c:367-383): banded or KSW_EZ_APPROX_DROP problems only ----
    auto approx_row = [&](int r, int st0, int en0, auto getv, auto getu) {
        if (r > 0) {
            if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                const int d0 = getv(last_H0_t), d1 = getu(last_H0_t + 1);
                if (d0 > d1) H0 += d0;
                else H0 += d1, ++last_H0_t;
            } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                H0 += getv(last_H0_t);
            } else {
                ++last_H0_t, H0 += getu(last_H0_t);
            }
        } else H0 = getv(0) - K.qe, last_H0_t = 0;
        if (flag & KSW_EZ_APPROX_DROP)
            if (__builtin_amdgcn_readfirstlane(zdrop_row(z, r, H0, last_H0_t, zdrop, K.e2))) { ez_zdropped = 1; brk = true; }
        if (!brk && r == n_rows - 1 && en0 == c2) ez_score = H0;
    };
    // NW > 1: bookkeeping of the owed row from what its waves published (triple-buffered: a fast wave may already be
    // publishing row r + 1 while a slow one still reads row r - 1)
    auto lag_row = [&]() {
        const int r = lag_r;
        lag_r = -1;
        const uint32_t *pb = pub + (r % 3) * (NW + 2);
        if (!APPROX) {
            uint32_t best = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) best = pb[i] > best ? pb[i] : best;
            const int bv = vreg((int)best);
            const int max_H = (int)((uint32_t)bv >> 16) - 32768, max_t = (int)((0xffffu - ((uint32_t)bv & 0xffffu)) & 8191u);
            exact_row(r, lag_st0, lag_en0, lag_en, max_H, max_t, vreg((int)pb[NW]), vreg((int)pb[NW + 1]));
        } else {
            const uint32_t *uv = uv4 + (r % 3) * 4;
            const int L = lag_L;      // the cells published for this row are L .. L + 3
            approx_row(r, lag_st0, lag_en0, [&](int t) { return (int)(short)(uv[t - L] >> 16); }, [&](int t) { return (int)(short)(uv[t - L] & 0xffffu); });
        }
    };

    int r = 0;
    for (; r < n_rows; ++r) {
        // row limits (ksw2_extd2_sse.c:138-147); st0, en0 >= 0 here, so the 16-alignment is plain bit arithmetic
        int st0 = r - c1, en0 = r;
        { const int b = (r - w + 1) >> 1; st0 = st0 > b ? st0 : b; st0 = st0 > 0 ? st0 : 0; }
        { const int b = (r + w) >> 1; en0 = en0 < b ? en0 : b; en0 = en0 < c2 ? en0 : c2; }
        if (st0 > en0) {
            if (ROT) {
                // every wave is here (r is uniform): the state after row r - 2 was written behind the last barrier
                __syncthreads();
                const int *sp = bst + ((r + 1) & 1) * 8;
                books_load(sp);
                if (sp[1]) { ez_zdropped = 1; brk = true; break; }
                lag_L = Lpub_prev;                     // (the window the owed row r - 1 was published with)
            }
            if (NW > 1 && lag_r >= 0 && !nob) { lag_row(); if (brk) break; }       // the owed row comes first: it may have Z-dropped (every wave leaves here: no flag needed)
            ez_zdropped = 1;
            break;
        }
        const int st = st0 & ~15, en = en0 | 15;
        const int sc_last = st0 + ((en0 - st0) & ~15) + 15;               // last cell of the row's 16-byte score stores
        const int hi_t = en > sc_last ? en : sc_last;                     // last cell touched by this row (state or score store)
        int bnd = -K.e2;
        if (r <= K.long_thres) bnd = r == 0 ? -K.q - K.e : r < K.long_thres ? -K.e : K.long_diff;
        const bool need_const = st == 0 || st - 1 < last_st || st - 1 > last_en;
        const int nbv_c = st > 0 ? -K.q - K.e : bnd;
        const int qbase = T + c1 - r;
        const uint8_t *qsrc = (qbase & 1) ? qB + qbase - 1 : qA + qbase;   // + t0 (even) is 2-byte aligned either way
        const int en1 = st0 + ((en0 - st0) & ~3);
        const uint32_t prow = (uint32_t)r * (uint32_t)ncol16 - (uint32_t)st;     // the traceback of one problem is < 4 GiB (host check)
        s2 pox = splat(0), pov = splat(0), pox2 = splat(0);
        int pohh = 0;
        uint32_t best = 0;
        int pub_en0 = 0, pub_st0 = 0;
        bool own_en0 = false, own_st0 = false;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int blk = c * NWC + cw, tb = blk * 128;
            const s2 ox = X[c], ov = V[c], ox2 = X2[c];
            const int ohl = HL[c], ohh = HH[c];
            if (computes && tb <= hi_t && tb + 127 >= st) {
                const int t0 = tb + 2 * lane;
                const uint32_t qw = *reinterpret_cast<const uint16_t *>(qsrc + t0);
                const s2 tq = S2((int)__builtin_amdgcn_perm(0u, qw, 0x0c010c00u));       // two bytes -> two halves
                // (x, v, x2) of cell t0 - 1: the upper half of the lane to the left; lane 0: the last cell of the previous block
                int fx, fv, fx2, fh = 0;
                if (NW == 1) {
                    fx = __builtin_amdgcn_readlane(I32(pox), 63), fv = __builtin_amdgcn_readlane(I32(pov), 63), fx2 = __builtin_amdgcn_readlane(I32(pox2), 63);
                    if (!APPROX) fh = __builtin_amdgcn_readlane(pohh, 63);
                } else {
                    const int sb = ((r + 1) & 1) * NB + (blk > 0 ? blk - 1 : 0);      // written at the end of row r - 1
                    const uint2 sv = seam[sb];
                    fx = (int)(sv.x << 16), fv = (int)(sv.x & 0xffff0000u), fx2 = (int)(sv.y << 16);
                    if (!APPROX) fh = seam_h[sb];
                }
                s2 xt1 = left_nb(ox, shr1(fx, I32(ox))), vt1 = left_nb(ov, shr1(fv, I32(ov))), x2t1 = left_nb(ox2, shr1(fx2, I32(ox2)));
                const int hleft_lo = !APPROX ? shr1(fh, ohh) : 0;
                if (need_const && t0 == st) {        // one lane of the row, if any: the sweep's first cell sees the boundary constants
                    xt1 = sel(0xffffu, splat(-K.q - K.e), xt1), vt1 = sel(0xffffu, splat(nbv_c), vt1), x2t1 = sel(0xffffu, splat(-K.q2 - K.e2), x2t1);
                }
                s2 ut = U[c], yo = Y[c], y2o = Y2[c];
                if (en >= r && (uint32_t)(r - t0) < 2u) {     // the lane that owns cell t == r: u[r], y[r], y2[r] boundary of this row
                    const uint32_t m = r == t0 ? 0xffffu : 0xffff0000u;
                    ut = sel(m, splat(bnd), ut), yo = sel(m, splat(-K.q - K.e), yo), y2o = sel(m, splat(-K.q2 - K.e2), y2o);
                }
                SC[c] = sel(range_mask(TP[c], st0, sc_last), score_pair(K, tq, TT[c]), SC[c]);
                if (t0 >= st && t0 <= en) {
                    s2 un, vn, xn, yn, x2n, y2n;
                    uint32_t dbytes;
                    cell_pair<RIGHT>(K, SC[c], ut, yo, y2o, xt1, vt1, x2t1, un, vn, xn, yn, x2n, y2n, dbytes);
                    U[c] = un, V[c] = vn, X[c] = xn, Y[c] = yn, X2[c] = x2n, Y2[c] = y2n;
                    *reinterpret_cast<uint16_t *>(p + (prow + (uint32_t)t0)) = (uint16_t)dbytes;
                    if (FAST) {
                        // the one cell of this row on the score path: cell 0 while r < qlen (its v), cell st0 = r - qlen + 1 afterwards (its u)
                        const s2 e = S2(I32(TP[c]) ^ I32(splat(st0)));
                        const s2 m = S2u(to01(U2(e))) - splat(1);                     // 0xffff in the half whose t == st0
                        ACC[c] = ACC[c] + S2(I32(m) & I32(r < qlen ? vn : un));
                    }
                    if (!APPROX) {
                        const int vlo = (int)vn.x, vhi = (int)vn.y, ulo = (int)un.x, uhi = (int)un.y;
                        int nl = ohl, nh = ohh;
                        if (r > 0) {
                            const bool in_lo = t0 >= st0 && t0 < en0, in_hi = t0 + 1 >= st0 && t0 + 1 < en0;
                            if (in_lo) nl = ohl + vlo;
                            if (in_hi) nh = ohh + vhi;
                            if (t0 == en0) nl = en0 > 0 ? hleft_lo + ulo : ohl + vlo;
                            if (t0 + 1 == en0) nh = ohl + uhi;                        // en0 = t0 + 1 > 0: H[en0 - 1] is the lane's own lower cell
                            if (in_lo || t0 == en0) { const uint32_t k = (uint32_t)(nl + 32768) << 16 | (0xffffu - (uint32_t)key_rank(t0, st0, en0, en1)); best = k > best ? k : best; }
                            if (in_hi || t0 + 1 == en0) { const uint32_t k = (uint32_t)(nh + 32768) << 16 | (0xffffu - (uint32_t)key_rank(t0 + 1, st0, en0, en1)); best = k > best ? k : best; }
                        } else if (t0 == 0) {
                            nl = vlo - K.qe;
                            best = (uint32_t)(nl + 32768) << 16 | (0xffffu - 0u);
                        }
                        HL[c] = nl, HH[c] = nh;
                        if (t0 == en0 || t0 + 1 == en0) own_en0 = true, pub_en0 = t0 == en0 ? nl : nh;
                        if (t0 == st0 || t0 + 1 == st0) own_st0 = true, pub_st0 = t0 == st0 ? nl : nh;
                    }
                }
            }
            pox = ox, pov = ov, pox2 = ox2, pohh = ohh;
        }
        if (NW == 1) {
            // bookkeeping of this row, straight from the registers
            if (!APPROX) {
                const int bv = vreg((int)wave_max_u32(best));
                const int max_H = (int)((uint32_t)bv >> 16) - 32768, max_t = (int)((0xffffu - ((uint32_t)bv & 0xffffu)) & 8191u);
                // H[en0] / H[st0]: the owning lane hands it to everybody through LDS (only the rows that can move mte / mqe / the score)
                int h_en0 = 0, h_st0 = 0;
                const bool want_en0 = en0 == c2, want_st0 = r - st0 == c1;
                if (want_en0 && own_en0) pub[0] = (uint32_t)pub_en0;
                if (want_st0 && own_st0) pub[1] = (uint32_t)pub_st0;
                if (want_en0 || want_st0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (want_en0) h_en0 = (int)pub[0];
                    if (want_st0) h_st0 = (int)pub[1];
                    h_en0 = vreg(h_en0), h_st0 = vreg(h_st0);
                }
                exact_row(r, st0, en0, en, max_H, max_t, h_en0, h_st0);
                if (brk) break;
            } else if (!FAST) {
                approx_row(r, st0, en0, [&](int t) { return fetch16<NCH>(V, t); }, [&](int t) { return fetch16<NCH>(U, t); });
                if (brk) break;
            }
        } else {
            if (APPROX && !nob) {
                // The bookkeeping of this row will need v[L'] and u[L' + 1] with L' = last_H0_t after row r - 1, which is L or L + 1 for
                // the L known now (rows up to r - 2 are booked): publish cells L .. L + 3 -- from whatever the registers hold, updated
                // this row or stale, which is what the reference's arrays would hold
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t0 = computes ? (c * NWC + cw) * 128 + 2 * lane : -(1 << 20);
                    const uint32_t d0 = (uint32_t)(t0 - (ROT ? Lpub : last_H0_t));
                    if (d0 < 4u) uv4[(r % 3) * 4 + d0] = (uint32_t)(uint16_t)U[c].x | (uint32_t)(uint16_t)V[c].x << 16;
                    if (d0 + 1u < 4u) uv4[(r % 3) * 4 + d0 + 1u] = (uint32_t)(uint16_t)U[c].y | (uint32_t)(uint16_t)V[c].y << 16;
                }
            }
            // publish: the seam cells row r + 1 can need (new values = its "previous anti-diagonal"; cell tb + 128 is inside
            // [st, en] of row r + 1 only if it is inside [st, en + 16] of this row), this wave's best key, H[en0] / H[st0]
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int blk = c * NWC + cw, nx = blk * 128 + 128;
                if (computes && nx >= st && nx <= en + 16 && lane == 63) {
                    const uint32_t xx = (uint32_t)I32(X[c]) >> 16, vv = (uint32_t)I32(V[c]) & 0xffff0000u, xx2 = (uint32_t)I32(X2[c]) >> 16;
                    seam[(r & 1) * NB + blk] = make_uint2(xx | vv, xx2);
                    if (!APPROX) seam_h[(r & 1) * NB + blk] = HH[c];
                }
            }
            if (!APPROX) {
                uint32_t *pb = pub + (r % 3) * (NW + 2);
                const uint32_t bw = wave_max_u32(best);
                if (lane == 0) pb[wv] = bw;
                if (own_en0) pb[NW] = (uint32_t)pub_en0;
                if (own_st0) pb[NW + 1] = (uint32_t)pub_st0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int L_used = ROT ? Lpub : last_H0_t;    // what this row's uv4 slots are relative to
            if (ROT) {
                const int *sp = bst + ((r + 1) & 1) * 8;          // the state after row r - 2
                const int sL = sp[0], sstop = sp[1];
                if (sstop) { books_load(sp); ez_zdropped = 1; brk = true; break; }      // every wave reads the same slot: all leave here
                if (lag_r >= 0 && wv == ((hi_t >> 7) + 1) % NW) {
                    books_load(sp);
                    lag_L = Lpub_prev;
                    lag_row();
                    int *so = bst + (r & 1) * 8;
                    if (lane == 0) { so[0] = last_H0_t, so[1] = brk ? 1 : 0, so[2] = H0, so[3] = z.max, so[4] = z.max_t, so[5] = z.max_q, so[6] = ez_score; }
                    brk = false, ez_zdropped = 0;                 // the others learn of a Z-drop behind the next barrier, and so does this wave
                }
                lag_r = r, lag_st0 = st0, lag_en0 = en0, lag_en = en;
                Lpub_prev = Lpub, Lpub = sL;
            } else
            if (nob) {}
            else if (APPROX || wv == 0) {
                // exact mode: only wave 0 keeps the books (the same ~100 instructions in every wave were most of a row's cost); when it
                // sees the Z-drop it raises the flag and still meets the others at their next barrier, where they read it and leave too
                if (lag_r >= 0) {
                    lag_row();
                    if (brk) {
                        if (!APPROX) { if (lane == 0) *stop_flag = 1u; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
                        break;
                    }
                }
                lag_r = r, lag_st0 = st0, lag_en0 = en0, lag_en = en, lag_L = L_used;
            } else if (*stop_flag) break;                 // raised by wave 0 before it arrived at this barrier
        }
        last_st = st, last_en = en;
    }
    if (ROT) {
        // the state after the last but one row was written behind the last barrier: one more, then wave 0 (whose lane 0 reports) does the last row
        __syncthreads();
        if (wv == 0 && !brk && !ez_zdropped) {
            const int *sp = bst + ((r + 1) & 1) * 8;
            books_load(sp);
            if (sp[1]) ez_zdropped = 1;
            else if (lag_r >= 0) { lag_L = Lpub_prev; lag_row(); }
        }
    } else
    if (NW > 1 && lag_r >= 0 && !brk && !ez_zdropped && !nob) lag_row();      // the last row's bookkeeping (its publication is behind a barrier already)
    if (nob && !FAST) ez_score = 0;
    if (FAST) {
        // unbanded approx problems never Z-drop and always reach the last row: score = H(tlen - 1, qlen - 1) = the path sum - (q + e)
        int sum = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) sum += (int)ACC[c].x + (int)ACC[c].y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        ez_score = (flag & KSW_EZ_NS_NO_SCORE) ? 0 : sum - K.qe;          // (a banded problem on this path: nobody reads its score)
    }
    __threadfence_block();
    __syncthreads();        // every traceback byte must have landed before one lane walks it
    if (flag & KSW_EZ_NS_SERIAL_BACKTRACK) {
        if (threadIdx.x == 0)
            backtrack_and_store(tk, w, ncol16, p, cig_pool, res_out, __builtin_amdgcn_readfirstlane(z.max), ez_zdropped, __builtin_amdgcn_readfirstlane(z.max_q),
                                __builtin_amdgcn_readfirstlane(z.max_t), __builtin_amdgcn_readfirstlane(z.mqe), __builtin_amdgcn_readfirstlane(z.mqe_t),
                                __builtin_amdgcn_readfirstlane(z.mte), __builtin_amdgcn_readfirstlane(z.mte_q), ez_score);
    } else if (threadIdx.x < 64)
        backtrack_and_store_wave(tk, w, ncol16, p, cig_pool, res_out, __builtin_amdgcn_readfirstlane(z.max), __builtin_amdgcn_readfirstlane(ez_zdropped),
                                 __builtin_amdgcn_readfirstlane(z.max_q), __builtin_amdgcn_readfirstlane(z.max_t), __builtin_amdgcn_readfirstlane(z.mqe),
                                 __builtin_amdgcn_readfirstlane(z.mqe_t), __builtin_amdgcn_readfirstlane(z.mte), __builtin_amdgcn_readfirstlane(z.mte_q),
                                 __builtin_amdgcn_readfirstlane(ez_score), (int)threadIdx.x);
}

// (Rounds 4-5, measured and removed: a systolic variant without a per-row barrier -- four waves on contiguous stretches of the target, the seam
// cell handed to the right-hand neighbour through tagged records in an LDS ring, a fifth wave keeping the books.  Faster per anti-diagonal
// than <6,2> on 64 equal problems alone on the chip (0.83 against 1.04 us for 300 .. 500 columns), slower in the engine in every arrangement
// tried: all ranges 7.54 s per cfg2 step, only 513 .. 1536 columns 7.10 s, only 257 .. 512 columns 6.96 s, against 6.67 / 6.69 s without it
// (round 5, interleaved on one box) -- five waves and 32 .. 96 KB of LDS per problem beside the 1 700 one-wave problems of the same launch.)

template <int NW, int NCH>
__global__ __launch_bounds__(NW * 64) void ksw_extd2_reg_kernel(const KswTask *__restrict__ tasks, const uint32_t *__restrict__ order, uint32_t n, KswParams pr,
                                                                const uint8_t *__restrict__ seqs, uint8_t *__restrict__ p_pool, uint32_t *__restrict__ cig_pool,
                                                                KswResult *__restrict__ res, const uint32_t *__restrict__ n_dev, DvCollect dc)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // n_dev: the number of problems is in device memory (a list written by the plan kernel, plan.hip; the grid is an upper bound)
    if (n_dev) n = *n_dev;
    if (blockIdx.x >= n) return;
    const uint32_t ti = order[blockIdx.x];
    if (ti == ~0u) return;                       // an entry the plan kernel reserved and gave back
    const KswTask tk = tasks[ti];
    const bool approx = (tk.flag & KSW_EZ_APPROX_MAX) != 0, right = (tk.flag & KSW_EZ_RIGHT) != 0;      // uniform per workgroup
#ifndef NSGPU_NO_SETPRIO
    // A launch is over when its longest problem is, and a SIMD's one 16-lane ALU is shared by the waves resident on it: the problems with the
    // most anti-diagonals win the issue arbitration against the short ones beside them
    {
        const long long rows = ksw_rows_bound(tk.qlen, tk.tlen, tk.w);
        if (rows > 1200) __builtin_amdgcn_s_setprio(3);
        else if (rows > 600) __builtin_amdgcn_s_setprio(2);
        else if (rows > 300) __builtin_amdgcn_s_setprio(1);
    }
#endif
    if (approx) {
        // gap fills whose band never binds and that cannot Z-drop take the path-independent score (see ksw_reg_run)
        const int w = tk.w < 0 ? tk.qlen + tk.tlen : tk.w;
        const bool fast = NW == 1 && !(tk.flag & KSW_EZ_APPROX_DROP) && (w >= tk.qlen + tk.tlen || (tk.flag & KSW_EZ_NS_NO_SCORE));
        if (right) ksw_reg_run<NW, NCH, true, true, false>(tk, pr, seqs, p_pool, cig_pool, res, lds);
        else if (fast) ksw_reg_run<NW, NCH, true, false, NW == 1>(tk, pr, seqs, p_pool, cig_pool, res, lds);
        else ksw_reg_run<NW, NCH, true, false, false>(tk, pr, seqs, p_pool, cig_pool, res, lds);
    } else {
        if (right) ksw_reg_run<NW, NCH, false, true, false>(tk, pr, seqs, p_pool, cig_pool, res, lds);
        else ksw_reg_run<NW, NCH, false, false, false>(tk, pr, seqs, p_pool, cig_pool, res, lds);
    }
    dev_problem_done(dc, ti, lds);             // (device-planned batches: the alignment's last problem hands it to the host)
}

// Classes.  Short problems (the gap fills) are throughput: one wave each.  Problems with many anti-diagonals (extensions of a few
// thousand bases against a few hundred) are latency: the launch is over when its longest problem is, so they get one 128-cell block per
// wave and row.
struct RegClass { int nw, nch; };
// Classes 2 and 4 .. 7 are retired (<4,3>, the latency twins <2,1> / <4,1>, <5,3> / <9,5> with a books wave: measured in rounds 2-3, slower;
// their numbers stay unused so that the others keep theirs).
// Classes 9 .. 11 belonged to the systolic kernel (removed in round 5, see above).
// Class 12 is <1,4> once more: the long problems of the one-wave classes in a launch of their own (device-planned batches, ksw_class.hpp).
// Round 5, interleaved on one box (profiles/r05_wave_shapes_ab.txt): the long problems of the one-wave classes (class 12) on <4,1> -- one 128-cell
// block per wave and row instead of four -- 1.51 -> 1.22 ms per launch; 513 .. 1536 columns (class 8) on <12,1> instead of <6,2>: 1.74 -> 1.53 ms;
// up to 5120 columns (class 3) on <16,3> instead of <8,5>: 3.20 -> 3.63 ms, not adopted.  (Rounds 2-3 had measured <2,1> / <4,1> slower than the
// one-wave classes: that was before the books left the gap fills' rows.)
static RegClass reg_class_of(int cls)
{
    // (the bulk classes on more waves -- class 0 on <2,1>, class 1 on <4,1> or <2,2> -- were measured too, profiles/r05_wave_shapes_bulk_ab.txt: the
    // launches of class 1 1.36 -> 1.29 ms on <4,1>, the wall of the DP phases -1 %, nothing on the step; <2,1> costs class 8 0.15 ms.  Not adopted.)
    constexpr RegClass base[KSW_REG_CLASSES] = {{1, 2}, {1, 4}, {6, 2}, {8, 5}, {1, 2}, {1, 4}, {6, 2}, {8, 5}, {12, 1}, {5, 1}, {5, 2}, {5, 3}, {4, 1}};
    return base[cls];
}
static int reg_compute_waves(int cls) { return reg_class_of(cls).nw; }

}  // namespace

int ksw_reg_cells(int cls) { return reg_compute_waves(cls) * reg_class_of(cls).nch * 128; }
int ksw_reg_threads(int cls) { return reg_class_of(cls).nw * 64; }

size_t ksw_reg_lds_bytes(int cls, int qlen)
{
    const int nw = reg_class_of(cls).nw, nb = reg_compute_waves(cls) * reg_class_of(cls).nch;
    size_t b = 2 * (size_t)reg_qb(ksw_reg_cells(cls), qlen);
    b += (size_t)2 * nb * 12 + (size_t)3 * (nw + 2) * 4 + 52;      // seams / publication slots (a one-wave class uses two of the slots) / stop flag
    b += 16 + 64;                                                  // the approx books' state, two slots (aligned)
    return b + 16;
}

// Which register-resident class serves the problem, or -1: the proofs behind the packed arithmetic (no int8 wrap outside the four
// adjusted gap terms, 16-bit H keys) hold for minimap2-sized scores and gap costs and for problems that fit a class.  The rule itself is
// ksw_class.hpp's (shared with the device-side alignment plan); the switches are read here, once:
//   NSGPU_KSW_NO_REG       debugging aid: first-generation kernels only
//   NSGPU_KSW_PROMOTE_ROWS the long problems of the narrowest class go with the <1,4> launch (ksw2.hip; default 520)
// (all bit-exact either way)
const KswClassCfg &ksw_class_config()
{
    static const KswClassCfg cfg = [] {
        KswClassCfg k;
        k.off = getenv("NSGPU_KSW_NO_REG") != nullptr;
        k.promote_rows = getenv("NSGPU_KSW_PROMOTE_ROWS") ? atoi(getenv("NSGPU_KSW_PROMOTE_ROWS")) : 520;
        k.long_rows = getenv("NSGPU_KSW_LONG_ROWS") ? atoi(getenv("NSGPU_KSW_LONG_ROWS")) : 900;
        k.flag_or = 0;
        if (getenv("NSGPU_KSW_ALL_BOOKS")) k.flag_or |= KSW_EZ_NS_ALL_BOOKS;                  // approx mode, several waves: every wave keeps the books
        if (getenv("NSGPU_KSW_SERIAL_BACKTRACK")) k.flag_or |= KSW_EZ_NS_SERIAL_BACKTRACK;    // one lane walks the traceback
        if (!getenv("NSGPU_KSW_NO_EARLY_EXIT")) k.flag_or |= KSW_EZ_NS_EARLY_EXIT;            // the exact early exit of overhang extensions
        return k;
    }();
    return cfg;
}

int ksw_reg_class(const KswTask &t, const KswParams &pr)
{
    return ksw_reg_class_hd(t.qlen, t.tlen, t.w, t.flag, pr, ksw_class_config());
}

int ksw_reg_launch(int cls, hipStream_t st, uint32_t m, size_t lds_bytes, const KswTask *tasks, const uint32_t *order, const KswParams &pr, const uint8_t *seqs,
                   uint8_t *p_pool, uint32_t *cig_pool, KswResult *res, const uint32_t *n_dev, const DvCollect *dc_in)
{
    DvCollect dc;
    if (dc_in) dc = *dc_in; else memset(&dc, 0, sizeof(dc));
#define NS_REG_LAUNCH(NW_, NCH_)                                                                                                              \
    {                                                                                                                                         \
        static LdsAttr attr;          /* launches come from several DP workspaces / threads, and per device */                              \
        if (lds_bytes > 32768) NS_TRY(attr.raise(lds_bytes, reinterpret_cast<const void *>(ksw_extd2_reg_kernel<NW_, NCH_>)));                \
        hipLaunchKernelGGL((ksw_extd2_reg_kernel<NW_, NCH_>), dim3(m), dim3(NW_ * 64), lds_bytes, st, tasks, order, m, pr, seqs, p_pool, cig_pool, res, n_dev, dc); \
    }
    switch (cls) {
    case 0: NS_REG_LAUNCH(1, 2) break;
    case 1: NS_REG_LAUNCH(1, 4) break;
    case 3: NS_REG_LAUNCH(8, 5) break;
    case 8: NS_REG_LAUNCH(12, 1) break;
    case 12: NS_REG_LAUNCH(4, 1) break;
    default: NS_CHECK(false, NSGPU_ERR_ARG, "ksw: bad register class");
    }
#undef NS_REG_LAUNCH
    NS_HIP(hipGetLastError());
    return NSGPU_OK;
}

}  // namespace nsgpu
