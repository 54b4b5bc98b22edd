// ksw2.hip -- a14h: banded dual-affine-gap extension / global alignment with
// traceback, bit-identical to minimap2's ksw_extd2_sse
// (minimap2/ksw2_extd2_sse.c:34-401, ksw2.h:103-176; v2.17-r974-dirty).
//
// Integer anti-diagonal wavefront kernel, no MFMA.  One 64-lane wave owns one DP
// problem; lanes run along the anti-diagonal (index t = target position), the
// (t-1) neighbour of the previous anti-diagonal arrives by a lane shuffle with a
// carried value across 64-cell chunks -- the wave64 counterpart of the
// reference's _mm_slli_si128 / x1_ carry.
//
// "Lane-exact" like the oracle (oracle/ksw2_oracle.c): the reference computes
// 16-cell blocks, so out-of-band cells inside the first/last block exist, hold
// stale-derived values and feed in-band cells once the band binds.  The kernel
// therefore sweeps the same 16-aligned [st, en] ranges, keeps the score row s,
// the target copy sf and the reversed query qr contiguous (16-byte score stores
// may spill from s into sf, exactly as in the reference's memory picture) and
// does int8 wrap-around arithmetic.
//
// Per-problem working set: 8 B of DP state per target cell + 2 B (s, sf) + the
// reversed query -- held in LDS (classes of 4/16/64 KiB per wave) or, for huge
// problems (LONG_JOIN gap fills up to 20 kb), in an HBM scratch slab.  The
// traceback matrix p (1 B per computed cell) is scratch in HBM; it is written
// once with coalesced 64-byte stores and read sparsely by the backtrack.
#include "common.hpp"
#include "ksw2.hpp"
#include "ksw_class.hpp"
#include "ksw_collect.hpp"
#include "host_util.hpp"
#include <rocprim/rocprim.hpp>

namespace nsgpu {

#define KSW_NEG_INF (-0x40000000)
#define KSW_EZ_SCORE_ONLY 0x01
#define KSW_EZ_RIGHT 0x02
#define KSW_EZ_GENERIC_SC 0x04
#define KSW_EZ_APPROX_MAX 0x08
#define KSW_EZ_APPROX_DROP 0x10
#define KSW_EZ_EXTZ_ONLY 0x40
#define KSW_EZ_REV_CIGAR 0x80

__device__ __forceinline__ int sx8(int v) { return (int)(int8_t)(v & 0xff); }

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

__device__ __forceinline__ unsigned long long shfl_max_u64(unsigned long long k)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o, 64);
        k = other > k ? other : k;
    }
    return k;
}

// Ordering between the lanes of the ONE wave that owns a problem.  LDS operations of a wave execute in issue
// order, so for LDS-resident state it is enough to drain the LDS queue (and stop the compiler from moving
// memory operations across): no s_barrier and, above all, no wait for the traceback stores still in flight
// to HBM (a __syncthreads() implies vmcnt(0), i.e. a full HBM store round trip on every anti-diagonal).
template <bool LDS_STATE> __device__ __forceinline__ void wave_sync()
{
    if (LDS_STATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
}

// BP: byte pointer, SP: uint2 (8-byte state) pointer, HP: int32 pointer -- LDS or HBM.
template <bool LDS_STATE, class BP, class SP, class HP>
__device__ void ksw_extd2_wave(const KswTask &tk, const KswParams &pr, const uint8_t *__restrict__ seqs, uint8_t *__restrict__ p_pool,
                               uint32_t *__restrict__ cig_pool, KswResult *__restrict__ res_out, SP S, BP bytes, HP H)
{
    const int lane = threadIdx.x & 63;
    const int qlen = tk.qlen, tlen = tk.tlen, flag = tk.flag, zdrop = tk.zdrop;
    int q = pr.q, e = pr.e, q2 = pr.q2, e2 = pr.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    const int sc_mch = pr.sc_mch, sc_mis = pr.sc_mis;
    const int sc_N = pr.sc_ambi == 0 ? -e2 : pr.sc_ambi;
    const bool approx_max = (flag & KSW_EZ_APPROX_MAX) != 0, right = (flag & KSW_EZ_RIGHT) != 0;

    // ksw_reset_extz
    int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
    int ez_score = KSW_NEG_INF, ez_reach_end = 0;
    uint32_t n_cigar = 0;

    int w = tk.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int tlen_ = (tlen + 15) / 16, T16 = tlen_ * 16;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    const int ncol16 = n_col_ * 16;
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

    BP s = bytes, sf = bytes + T16, qr = bytes + 2 * T16;
    const int qr_bytes = ((qlen + 15) / 16 + 1) * 16;
    // init: state = {u,v,x,y = -q-e ; x2,y2 = -q2-e2}; s, sf, qr zero then filled
    {
        const uint32_t b0 = (uint32_t)(-q - e) & 0xff, b1 = (uint32_t)(-q2 - e2) & 0xff;
        const uint2 init = make_uint2(b0 | b0 << 8 | b0 << 16 | b0 << 24, b1 | b1 << 8);
        for (int t = lane; t < T16; t += 64) S[t] = init;
        for (int i = lane; i < 2 * T16 + qr_bytes; i += 64) bytes[i] = 0;
        if (!approx_max) for (int t = lane; t < T16; t += 64) H[t] = KSW_NEG_INF;
    }
    wave_sync<LDS_STATE>();
    {
        const uint8_t *query = seqs + tk.qoff, *target = seqs + tk.toff;
        for (int t = lane; t < qlen; t += 64) qr[t] = query[qlen - 1 - t];
        for (int t = lane; t < tlen; t += 64) sf[t] = target[t];
    }
    wave_sync<LDS_STATE>();

    uint8_t *p = p_pool + tk.p_off;
    int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
    const int n_rows = qlen + tlen - 1;

    for (int r = 0; r < n_rows; ++r) {
        const RowRange rr = row_range(r, qlen, tlen, w);
        if (rr.empty) { ez_zdropped = 1; break; }
        const int st0 = rr.st0, en0 = rr.en0, st = rr.st, en = rr.en;
        int x1, x21, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) {
                const uint2 sv = S[st - 1];
                x1 = sx8(sv.x >> 16), x21 = sx8(sv.y), v1 = sx8(sv.x >> 8);
            } else x1 = sx8(-q - e), x21 = sx8(-q2 - e2), v1 = sx8(-q - e);
        } else {
            x1 = sx8(-q - e), x21 = sx8(-q2 - e2);
            v1 = r == 0 ? sx8(-q - e) : r < long_thres ? sx8(-e) : r == long_thres ? sx8(long_diff) : sx8(-e2);
        }
        if (en >= r && lane == 0) {
            uint2 sv = S[r];
            const uint32_t ur = (uint32_t)(r == 0 ? (-q - e) : r < long_thres ? (-e) : r == long_thres ? long_diff : (-e2)) & 0xff;
            sv.x = (sv.x & 0x00ffff00u) | ur | (((uint32_t)(-q - e) & 0xff) << 24);       // u[r], y[r]
            sv.y = (sv.y & 0xffff00ffu) | (((uint32_t)(-q2 - e2) & 0xff) << 8);           // y2[r]
            S[r] = sv;
        }
        // score row: 16-byte blocks from st0; may run past en0 and from s into sf
        {
            const int nsc = ((en0 - st0) / 16 + 1) * 16;
            const int qoff = qlen - 1 - r;      // qrr = qr + qoff, may be negative (reads sf's tail)
            for (int i = lane; i < nsc; i += 64) {
                const int t = st0 + i;
                const int sq = sf[t], sq2 = qr[qoff + t];
                int z = sq == sq2 ? sc_mch : sc_mis;
                if (sq == 4 || sq2 == 4) z = sc_N;
                s[t] = (uint8_t)z;
            }
        }
        wave_sync<LDS_STATE>();
        // core loop over the 16-aligned range
        {
            int cx = x1, cx2 = x21, cv = v1;
            uint8_t *prow = p + (size_t)r * ncol16 - st;
            for (int c = st; c <= en; c += 64) {
                const int t = c + lane;
                const bool on = t <= en;
                uint2 sv = make_uint2(0, 0);
                int z = 0;
                if (on) { sv = S[t]; z = sx8(s[t]); }
                const int ut = sx8(sv.x), vo = sx8(sv.x >> 8), xo = sx8(sv.x >> 16), yo = sx8(sv.x >> 24);
                const int x2o = sx8(sv.y), y2o = sx8(sv.y >> 8);
                int xt1 = __shfl_up(xo, 1, 64), vt1 = __shfl_up(vo, 1, 64), x2t1 = __shfl_up(x2o, 1, 64);
                if (lane == 0) xt1 = cx, vt1 = cv, x2t1 = cx2;
                cx = __shfl(xo, 63, 64), cv = __shfl(vo, 63, 64), cx2 = __shfl(x2o, 63, 64);
                int a = sx8(xt1 + vt1), b = sx8(yo + ut), a2 = sx8(x2t1 + vt1), b2 = sx8(y2o + ut), d;
                if (!right) {
                    d = a > z ? 1 : 0;  z = z > a ? z : a;
                    d = b > z ? 2 : d;  z = z > b ? z : b;
                    d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
                    d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
                } else {
                    d = z > a ? 0 : 1;  z = z > a ? z : a;
                    d = z > b ? d : 2;  z = z > b ? z : b;
                    d = z > a2 ? d : 3; z = z > a2 ? z : a2;
                    d = z > b2 ? d : 4; z = z > b2 ? z : b2;
                }
                z = z < sc_mch ? z : sc_mch;
                const int un = sx8(z - vt1), vn = sx8(z - ut);
                int tmp = sx8(z - q);
                a = sx8(a - tmp), b = sx8(b - tmp);
                tmp = sx8(z - q2);
                a2 = sx8(a2 - tmp), b2 = sx8(b2 - tmp);
                int xn, yn, x2n, y2n;
                if (!right) {
                    xn = sx8((a > 0 ? a : 0) - qe);    d |= a > 0 ? 0x08 : 0;
                    yn = sx8((b > 0 ? b : 0) - qe);    d |= b > 0 ? 0x10 : 0;
                    x2n = sx8((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
                    y2n = sx8((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
                } else {
                    xn = sx8((0 > a ? 0 : a) - qe);    d |= 0 > a ? 0 : 0x08;
                    yn = sx8((0 > b ? 0 : b) - qe);    d |= 0 > b ? 0 : 0x10;
                    x2n = sx8((0 > a2 ? 0 : a2) - qe2); d |= 0 > a2 ? 0 : 0x20;
                    y2n = sx8((0 > b2 ? 0 : b2) - qe2); d |= 0 > b2 ? 0 : 0x40;
                }
                if (on) {
                    S[t] = make_uint2((uint32_t)(un & 0xff) | (uint32_t)(vn & 0xff) << 8 | (uint32_t)(xn & 0xff) << 16 | (uint32_t)(yn & 0xff) << 24,
                                      (uint32_t)(x2n & 0xff) | (uint32_t)(y2n & 0xff) << 8);
                    prow[t] = (uint8_t)d;
                }
            }
        }
        wave_sync<LDS_STATE>();
        bool brk = false;
        if (!approx_max) {
            int max_H, max_t;
            if (r > 0) {
                // H[en0] first (from the OLD H[en0-1]), then H[t] += v[t] for st0 <= t < en0
                int h_en0;
                {
                    const uint2 se = S[en0];
                    h_en0 = en0 > 0 ? H[en0 - 1] + sx8(se.x) : H[en0] + sx8(se.x >> 8);
                }
                wave_sync<LDS_STATE>();
                const int en1 = st0 + (en0 - st0) / 4 * 4;
                unsigned long long best = ((unsigned long long)((long long)h_en0 + 0x80000000ll) << 32) | 0xFFFFFFFFull;  // rank -1: wins ties
                for (int t = st0 + lane; t < en0; t += 64) {
                    const int h = H[t] + sx8(S[t].x >> 8);
                    H[t] = h;
                    const uint32_t rank = t < en1 ? (uint32_t)((t - st0) & 3) * 0x100000u + (uint32_t)t : 4u * 0x100000u + (uint32_t)t;
                    const unsigned long long key = ((unsigned long long)((long long)h + 0x80000000ll) << 32) | (0xFFFFFFFEull - rank);
                    best = key > best ? key : best;
                }
                if (lane == 0) H[en0] = h_en0;
                best = shfl_max_u64(best);
                max_H = (int)((long long)(best >> 32) - 0x80000000ll);
                const uint32_t lo = (uint32_t)best;
                max_t = lo == 0xFFFFFFFFu ? en0 : (int)((0xFFFFFFFEu - lo) & 0xFFFFFu);
                wave_sync<LDS_STATE>();
            } else {
                const int h = sx8(S[0].x >> 8) - qe;
                wave_sync<LDS_STATE>();
                if (lane == 0) H[0] = h;
                max_H = h, max_t = 0;
                wave_sync<LDS_STATE>();
            }
            const int h_en0 = H[en0], h_st0 = H[st0];
            if (en0 == tlen - 1 && h_en0 > ez_mte) ez_mte = h_en0, ez_mte_q = r - en;
            if (r - st0 == qlen - 1 && h_st0 > ez_mqe) ez_mqe = h_st0, ez_mqe_t = st0;
            // ksw_apply_zdrop(ez, 1, max_H, r, max_t, zdrop, e2)
            if (max_H > ez_max) {
                ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
            } else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
                const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
                if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; brk = true; }
            }
            if (!brk && r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int d0 = sx8(S[last_H0_t].x >> 8), d1 = sx8(S[last_H0_t + 1].x);
                    if (d0 > d1) H0 += d0;
                    else H0 += d1, ++last_H0_t;
                } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                    H0 += sx8(S[last_H0_t].x >> 8);
                } else {
                    ++last_H0_t, H0 += sx8(S[last_H0_t].x);
                }
            } else H0 = sx8(S[0].x >> 8) - qe, last_H0_t = 0;
            if (flag & KSW_EZ_APPROX_DROP) {
                if (H0 > ez_max) {
                    ez_max = H0, ez_max_t = last_H0_t, ez_max_q = r - last_H0_t;
                } else if (last_H0_t >= ez_max_t && r - last_H0_t >= ez_max_q) {
                    const int tl = last_H0_t - ez_max_t, ql = (r - last_H0_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
                    if (zdrop >= 0 && ez_max - H0 > zdrop + l * e2) { ez_zdropped = 1; brk = true; }
                }
            }
            if (!brk && r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H0;
        }
        if (brk) break;
        last_st = st, last_en = en;
    }
    __threadfence_block();
    __syncthreads();        // every traceback byte must have landed before lane 0 walks it

    // backtrack (ksw2.h:119-151, is_rot = 1) by lane 0
    if (!(flag & KSW_EZ_SCORE_ONLY)) {
        int i0 = -1, j0 = -1;
        if (!ez_zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
        else if (!ez_zdropped && (flag & KSW_EZ_EXTZ_ONLY) && ez_mqe + tk.end_bonus > ez_max) ez_reach_end = 1, i0 = ez_mqe_t, j0 = qlen - 1;
        else if (ez_max_t >= 0 && ez_max_q >= 0) i0 = ez_max_t, j0 = ez_max_q;
        if (lane == 0 && i0 >= 0 && j0 >= 0) {
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
    if (lane == 0) {
        KswResult o;
        o.max = (uint32_t)ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t; o.mqe = ez_mqe; o.mqe_t = ez_mqe_t;
        o.mte = ez_mte; o.mte_q = ez_mte_q; o.score = ez_score; o.n_cigar = (int)n_cigar; o.reach_end = ez_reach_end;
        res_out[tk.out_idx] = o;
    }
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;

__global__ __launch_bounds__(64) void ksw_extd2_lds_kernel(const KswTask *__restrict__ tasks, const uint32_t *__restrict__ order, uint32_t n,
                                                           KswParams pr, const uint8_t *__restrict__ seqs, uint8_t *__restrict__ p_pool,
                                                           uint32_t *__restrict__ cig_pool, KswResult *__restrict__ res)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const KswTask tk = tasks[order[i]];
    const int T16 = (tk.tlen + 15) / 16 * 16;
    uint2 *S = reinterpret_cast<uint2 *>(lds);
    int *H = reinterpret_cast<int *>(lds + (size_t)T16 * 8);
    const bool exact = !(tk.flag & KSW_EZ_APPROX_MAX);
    uint8_t *bytes = lds + (size_t)T16 * 8 + (exact ? (size_t)T16 * 4 : 0);
    ksw_extd2_wave<true, uint8_t *, uint2 *, int *>(tk, pr, seqs, p_pool, cig_pool, res, S, bytes, H);
}

__global__ __launch_bounds__(64) void ksw_extd2_hbm_kernel(const KswTask *__restrict__ tasks, const uint32_t *__restrict__ order, uint32_t n,
                                                           KswParams pr, const uint8_t *__restrict__ seqs, uint8_t *__restrict__ p_pool,
                                                           uint32_t *__restrict__ cig_pool, KswResult *__restrict__ res,
                                                           uint8_t *__restrict__ slab, size_t slab_stride)
{
    // persistent-style: each workgroup owns one slab and walks the task list
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const KswTask tk = tasks[order[i]];
        const int T16 = (tk.tlen + 15) / 16 * 16;
        uint8_t *base = slab + (size_t)blockIdx.x * slab_stride;
        uint2 *S = reinterpret_cast<uint2 *>(base);
        int *H = reinterpret_cast<int *>(base + (size_t)T16 * 8);
        uint8_t *bytes = base + (size_t)T16 * 12;
        ksw_extd2_wave<false, uint8_t *, uint2 *, int *>(tk, pr, seqs, p_pool, cig_pool, res, S, bytes, H);
        __syncthreads();
    }
}


// ----------------------------------------------------------------------------
// v2: workgroup-per-problem kernel (NT = 64 ... 256 threads).  Every anti-diagonal is swept in two phases:
//   read   : each thread loads the state of its (up to MAXPOS) cells, the (x, v, x2) of the left neighbour
//            straight from LDS (no cross-lane shuffles, no carried values), and its score -- recomputed when the
//            cell lies in the reference's 16-byte score-store range of this row, the stale s[t] otherwise;
//   write  : after a barrier, the recurrences, the new state, s[t] and the traceback byte.
// Same lane-exact semantics as ksw_extd2_wave (16-aligned sweep, stale scores, s|sf|qr contiguity, int8 wrap).
// Long problems get 256 threads, i.e. 4 waves per anti-diagonal instead of 1: they are latency-bound.
// ----------------------------------------------------------------------------
// Workgroup barrier for LDS-only communication: this wave's LDS traffic is drained (lgkmcnt(0)) before it arrives; the
// traceback bytes still in flight to HBM are NOT waited for.  (__syncthreads() implies vmcnt(0): with one traceback store
// per cell and row that was a full HBM store round trip, ~1 us, on every anti-diagonal of the long problems.)  The one
// __syncthreads() behind the row loop makes the traceback visible before lane 0 walks it.
static __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, int MAXPOS>
__global__ __launch_bounds__(NT) void ksw_extd2_wg_kernel(const KswTask *__restrict__ tasks, const uint32_t *__restrict__ order, uint32_t n_tasks,
                                                          KswParams pr, const uint8_t *__restrict__ seqs, uint8_t *__restrict__ p_pool,
                                                          uint32_t *__restrict__ cig_pool, KswResult *__restrict__ res_out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ unsigned long long s_best[NT / 64];
    if (blockIdx.x >= n_tasks) return;
    const KswTask tk = tasks[order[blockIdx.x]];
    const int tid = threadIdx.x;
    const int qlen = tk.qlen, tlen = tk.tlen, flag = tk.flag, zdrop = tk.zdrop;
    int q = pr.q, e = pr.e, q2 = pr.q2, e2 = pr.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    const int sc_mch = pr.sc_mch, sc_mis = pr.sc_mis;
    const int sc_N = pr.sc_ambi == 0 ? -e2 : pr.sc_ambi;
    const bool approx_max = (flag & KSW_EZ_APPROX_MAX) != 0, right = (flag & KSW_EZ_RIGHT) != 0;
    int ez_max = 0, ez_zdropped = 0, ez_max_q = -1, ez_max_t = -1, ez_mqe = KSW_NEG_INF, ez_mqe_t = -1, ez_mte = KSW_NEG_INF, ez_mte_q = -1;
    int ez_score = KSW_NEG_INF, ez_reach_end = 0;
    uint32_t n_cigar = 0;
    int w = tk.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int T16 = (tlen + 15) / 16 * 16;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    const int ncol16 = n_col_ * 16;
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

    uint2 *S = reinterpret_cast<uint2 *>(lds);
    int *H = reinterpret_cast<int *>(lds + (size_t)T16 * 8);
    uint8_t *bytes = lds + (size_t)T16 * 8 + (approx_max ? 0 : (size_t)T16 * 4);
    uint8_t *s = bytes, *sf = bytes + T16, *qr = bytes + 2 * T16;
    const int qr_bytes = ((qlen + 15) / 16 + 1) * 16;
    const uint32_t b0 = (uint32_t)(-q - e) & 0xff, b1 = (uint32_t)(-q2 - e2) & 0xff;
    {
        const uint2 init = make_uint2(b0 | b0 << 8 | b0 << 16 | b0 << 24, b1 | b1 << 8);
        for (int t = tid; t < T16; t += NT) S[t] = init;
        for (int i = tid; i < 2 * T16 + qr_bytes; i += NT) bytes[i] = 0;
        if (!approx_max) for (int t = tid; t < T16; t += NT) H[t] = KSW_NEG_INF;
    }
    __syncthreads();
    {
        const uint8_t *query = seqs + tk.qoff, *target = seqs + tk.toff;
        for (int t = tid; t < qlen; t += NT) qr[t] = query[qlen - 1 - t];
        for (int t = tid; t < tlen; t += NT) sf[t] = target[t];
    }
    __syncthreads();

    uint8_t *p = p_pool + tk.p_off;
    int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
    const int n_rows = qlen + tlen - 1;
    for (int r = 0; r < n_rows; ++r) {
        const RowRange rr = row_range(r, qlen, tlen, w);
        if (rr.empty) { ez_zdropped = 1; break; }
        const int st0 = rr.st0, en0 = rr.en0, st = rr.st, en = rr.en;
        // ---- read phase ----
        uint32_t nb_first;                                         // (x, v, x2) seen by the first cell of the sweep
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { const uint2 sv = S[st - 1]; nb_first = ((sv.x >> 16) & 0xff) << 16 | ((sv.x >> 8) & 0xff) << 8 | (sv.y & 0xff); }
            else nb_first = b0 << 16 | b0 << 8 | b1;
        } else {
            const uint32_t v1 = (uint32_t)(r == 0 ? (-q - e) : r < long_thres ? (-e) : r == long_thres ? long_diff : (-e2)) & 0xff;
            nb_first = b0 << 16 | v1 << 8 | b1;
        }
        const int sc_end = st0 + ((en0 - st0) / 16 + 1) * 16;    // exclusive end of this row's score stores
        const int qoff = qlen - 1 - r;
        uint2 sv[MAXPOS];
        uint32_t nb[MAXPOS];
        int zz[MAXPOS];
        int h_left = 0;
        // exact max (extensions): the row's H update and arg-max ride along with the write phase -- every cell's H moves by
        // its own new v (H[en0] by its new u on top of the OLD H[en0 - 1], read above), so no extra pass over the row
        const int en1 = st0 + (en0 - st0) / 4 * 4;
        unsigned long long best = 0;
#pragma unroll
        for (int k = 0; k < MAXPOS; ++k) {
            const int t = st + tid + k * NT;
            sv[k] = make_uint2(0, 0), nb[k] = 0, zz[k] = 0;
            if (t <= en) {
                uint2 a = S[t];
                if (t == r) {                                      // u[r], y[r], y2[r] boundary of this row (en >= r holds: r is inside [st, en])
                    const uint32_t ur = (uint32_t)(r == 0 ? (-q - e) : r < long_thres ? (-e) : r == long_thres ? long_diff : (-e2)) & 0xff;
                    a.x = (a.x & 0x00ffff00u) | ur | (b0 << 24);
                    a.y = (a.y & 0xffff00ffu) | (b1 << 8);
                }
                sv[k] = a;
                if (!approx_max && r > 0 && t == rr.en0 && t > 0) h_left = H[t - 1];   // H[en0 - 1] BEFORE this row's update
                if (t == st) nb[k] = nb_first;
                else { const uint2 l = S[t - 1]; nb[k] = ((l.x >> 16) & 0xff) << 16 | ((l.x >> 8) & 0xff) << 8 | (l.y & 0xff); }
                if (t >= st0 && t < sc_end) {
                    const int sq = sf[t], sq2 = qr[qoff + t];
                    int z = sq == sq2 ? sc_mch : sc_mis;
                    if (sq == 4 || sq2 == 4) z = sc_N;
                    zz[k] = z;
                } else zz[k] = sx8(s[t]);
            }
        }
        // score stores that run past the aligned end of the sweep (at most 15 cells; they may spill from s into sf)
        int t_extra = en + 1 + tid, z_extra = 0;
        const bool has_extra = tid < 16 && t_extra < sc_end;
        if (has_extra) {
            const int sq = sf[t_extra], sq2 = qr[qoff + t_extra];
            z_extra = sq == sq2 ? sc_mch : sc_mis;
            if (sq == 4 || sq2 == 4) z_extra = sc_N;
        }
        lds_barrier();
        // ---- write phase ----
        uint8_t *prow = p + (size_t)r * ncol16 - st;
#pragma unroll
        for (int k = 0; k < MAXPOS; ++k) {
            const int t = st + tid + k * NT;
            if (t <= en) {
                int z = zz[k];
                const int ut = sx8(sv[k].x), yo = sx8(sv[k].x >> 24), y2o = sx8(sv[k].y >> 8);
                const int xt1 = sx8(nb[k] >> 16), vt1 = sx8(nb[k] >> 8), x2t1 = sx8(nb[k]);
                int a = sx8(xt1 + vt1), b = sx8(yo + ut), a2 = sx8(x2t1 + vt1), b2 = sx8(y2o + ut), d;
                if (!right) {
                    d = a > z ? 1 : 0;  z = z > a ? z : a;
                    d = b > z ? 2 : d;  z = z > b ? z : b;
                    d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
                    d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
                } else {
                    d = z > a ? 0 : 1;  z = z > a ? z : a;
                    d = z > b ? d : 2;  z = z > b ? z : b;
                    d = z > a2 ? d : 3; z = z > a2 ? z : a2;
                    d = z > b2 ? d : 4; z = z > b2 ? z : b2;
                }
                z = z < sc_mch ? z : sc_mch;
                const int un = sx8(z - vt1), vn = sx8(z - ut);
                int tmp = sx8(z - q);
                a = sx8(a - tmp), b = sx8(b - tmp);
                tmp = sx8(z - q2);
                a2 = sx8(a2 - tmp), b2 = sx8(b2 - tmp);
                int xn, yn, x2n, y2n;
                if (!right) {
                    xn = sx8((a > 0 ? a : 0) - qe);    d |= a > 0 ? 0x08 : 0;
                    yn = sx8((b > 0 ? b : 0) - qe);    d |= b > 0 ? 0x10 : 0;
                    x2n = sx8((a2 > 0 ? a2 : 0) - qe2); d |= a2 > 0 ? 0x20 : 0;
                    y2n = sx8((b2 > 0 ? b2 : 0) - qe2); d |= b2 > 0 ? 0x40 : 0;
                } else {
                    xn = sx8((0 > a ? 0 : a) - qe);    d |= 0 > a ? 0 : 0x08;
                    yn = sx8((0 > b ? 0 : b) - qe);    d |= 0 > b ? 0 : 0x10;
                    x2n = sx8((0 > a2 ? 0 : a2) - qe2); d |= 0 > a2 ? 0 : 0x20;
                    y2n = sx8((0 > b2 ? 0 : b2) - qe2); d |= 0 > b2 ? 0 : 0x40;
                }
                S[t] = make_uint2((uint32_t)(un & 0xff) | (uint32_t)(vn & 0xff) << 8 | (uint32_t)(xn & 0xff) << 16 | (uint32_t)(yn & 0xff) << 24,
                                  (uint32_t)(x2n & 0xff) | (uint32_t)(y2n & 0xff) << 8);
                if (t >= st0 && t < sc_end) s[t] = (uint8_t)zz[k];
                prow[t] = (uint8_t)d;
                if (!approx_max && r > 0) {
                    if (t >= st0 && t < en0) {
                        const int h = H[t] + vn;
                        H[t] = h;
                        const uint32_t rank = t < en1 ? (uint32_t)((t - st0) & 3) * 0x100000u + (uint32_t)t : 4u * 0x100000u + (uint32_t)t;
                        const unsigned long long key = ((unsigned long long)((long long)h + 0x80000000ll) << 32) | (0xFFFFFFFEull - rank);
                        best = key > best ? key : best;
                    } else if (t == en0) {
                        const int h = en0 > 0 ? h_left + un : H[en0] + vn;
                        H[en0] = h;
                        const unsigned long long key = ((unsigned long long)((long long)h + 0x80000000ll) << 32) | 0xFFFFFFFFull;
                        best = key > best ? key : best;
                    }
                }
            }
        }
        if (has_extra) s[t_extra] = (uint8_t)z_extra;
        if (!approx_max && r > 0) {
            best = shfl_max_u64(best);
            if (NT > 64 && (tid & 63) == 0) s_best[tid >> 6] = best;
        }
        lds_barrier();
        // ---- score bookkeeping (uniform) ----
        bool brk = false;
        if (!approx_max) {
            int max_H, max_t;
            if (r > 0) {
                if (NT > 64) {
#pragma unroll
                    for (int i = 0; i < NT / 64; ++i) best = s_best[i] > best ? s_best[i] : best;
                }
                max_H = (int)((long long)(best >> 32) - 0x80000000ll);
                const uint32_t lo = (uint32_t)best;
                max_t = lo == 0xFFFFFFFFu ? en0 : (int)((0xFFFFFFFEu - lo) & 0xFFFFFu);
            } else {
                const int h = sx8(S[0].x >> 8) - qe;
                __syncthreads();
                if (tid == 0) H[0] = h;
                max_H = h, max_t = 0;
                __syncthreads();
            }
            const int h_en0 = H[en0], h_st0 = H[st0];
            if (en0 == tlen - 1 && h_en0 > ez_mte) ez_mte = h_en0, ez_mte_q = r - en;
            if (r - st0 == qlen - 1 && h_st0 > ez_mqe) ez_mqe = h_st0, ez_mqe_t = st0;
            if (max_H > ez_max) {
                ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
            } else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
                const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
                if (zdrop >= 0 && ez_max - max_H > zdrop + l * e2) { ez_zdropped = 1; brk = true; }
            }
            if (!brk && r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int d0 = sx8(S[last_H0_t].x >> 8), d1 = sx8(S[last_H0_t + 1].x);
                    if (d0 > d1) H0 += d0;
                    else H0 += d1, ++last_H0_t;
                } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                    H0 += sx8(S[last_H0_t].x >> 8);
                } else {
                    ++last_H0_t, H0 += sx8(S[last_H0_t].x);
                }
            } else H0 = sx8(S[0].x >> 8) - qe, last_H0_t = 0;
            if (flag & KSW_EZ_APPROX_DROP) {
                if (H0 > ez_max) {
                    ez_max = H0, ez_max_t = last_H0_t, ez_max_q = r - last_H0_t;
                } else if (last_H0_t >= ez_max_t && r - last_H0_t >= ez_max_q) {
                    const int tl = last_H0_t - ez_max_t, ql = (r - last_H0_t) - ez_max_q, l = tl > ql ? tl - ql : ql - tl;
                    if (zdrop >= 0 && ez_max - H0 > zdrop + l * e2) { ez_zdropped = 1; brk = true; }
                }
            }
            if (!brk && r == qlen + tlen - 2 && en0 == tlen - 1) ez_score = H0;
        }
        if (brk) break;
        last_st = st, last_en = en;
    }
    __threadfence_block();
    __syncthreads();
    if (!(flag & KSW_EZ_SCORE_ONLY)) {
        int i0 = -1, j0 = -1;
        if (!ez_zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
        else if (!ez_zdropped && (flag & KSW_EZ_EXTZ_ONLY) && ez_mqe + tk.end_bonus > ez_max) ez_reach_end = 1, i0 = ez_mqe_t, j0 = qlen - 1;
        else if (ez_max_t >= 0 && ez_max_q >= 0) i0 = ez_max_t, j0 = ez_max_q;
        if (tid == 0 && i0 >= 0 && j0 >= 0) {
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
    if (tid == 0) {
        KswResult o;
        o.max = (uint32_t)ez_max; o.zdropped = ez_zdropped; o.max_q = ez_max_q; o.max_t = ez_max_t; o.mqe = ez_mqe; o.mqe_t = ez_mqe_t;
        o.mte = ez_mte; o.mte_q = ez_mte_q; o.score = ez_score; o.n_cigar = (int)n_cigar; o.reach_end = ez_reach_end;
        res_out[tk.out_idx] = o;
    }
}

// widest 16-aligned sweep of a problem (cells per anti-diagonal)
static int ksw_max_width(int qlen, int tlen, int w)
{
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int T16 = (tlen + 15) / 16 * 16;
    const long long band = (long long)w + 1 + 32;
    return (int)(band < T16 ? band : T16);
}

// CIGAR compaction: every problem owns a worst-case slice (qlen + tlen + 2 entries) of the CIGAR pool, of which it
// uses a handful; only the used entries travel back over PCIe.
struct NcigarAt {
    const KswResult *res; uint32_t n;
    __host__ __device__ uint64_t operator()(uint32_t i) const { return i < n ? (uint64_t)(uint32_t)res[i].n_cigar : 0ull; }
};

__global__ __launch_bounds__(256) void ksw_cigar_gather_kernel(const KswTask *__restrict__ tasks, const KswResult *__restrict__ res, uint32_t n,
                                                               const uint64_t *__restrict__ off, const uint32_t *__restrict__ pool,
                                                               uint32_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = w; i < n; i += nw) {
        const uint32_t c = (uint32_t)res[i].n_cigar;
        const uint32_t *src = pool + tasks[i].cig_off;
        uint32_t *dst = out + off[i];
        for (uint32_t k = lane; k < c; k += 64) dst[k] = src[k];
    }
}

size_t ksw_lds_bytes(int qlen, int tlen, int flag)
{
    const size_t T16 = (size_t)(tlen + 15) / 16 * 16;
    const size_t qrb = ((size_t)(qlen + 15) / 16 + 1) * 16;
    return T16 * 8 + ((flag & KSW_EZ_APPROX_MAX) ? 0 : T16 * 4) + 2 * T16 + qrb + 16;
}

size_t ksw_p_bytes(int qlen, int tlen, int w)
{
    if (qlen <= 0 || tlen <= 0) return 0;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    return ((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16;
}

// Runs a batch.  tasks/seqs are host arrays; results and CIGARs come back to the host.
// Task fields p_off / cig_off / out_idx are filled here.
int ksw_run_batch(nsgpu_ctx *c, std::vector<KswTask> &tasks, const uint8_t *seqs, size_t seq_bytes, const KswParams &pr,
                  std::vector<KswResult> &results, std::vector<uint32_t> &cigars, std::vector<uint64_t> &cig_off, int ws_index)
{
    NS_TRY(ksw_batch_launch(c, tasks, seqs, seq_bytes, pr, results, cigars, cig_off, ws_index));
    return ksw_batch_collect(c, tasks, results, cigars, cig_off, ws_index);
}

// First half: uploads, all DP launches of the batch and the CIGAR-length scan are enqueued; nothing is waited for.  tasks,
// results and cig_off must stay alive (and seqs until the stream has consumed it: pinned memory) up to ksw_batch_collect
// with the same workspace.
int ksw_batch_launch(nsgpu_ctx *c, std::vector<KswTask> &tasks, const uint8_t *seqs, size_t seq_bytes, const KswParams &pr,
                     std::vector<KswResult> &results, std::vector<uint32_t> &cigars, std::vector<uint64_t> &cig_off, int ws_index)
{
    const size_t n = tasks.size();
    NS_CHECK(ws_index >= 0 && ws_index <= 7, NSGPU_ERR_ARG, "ksw: workspace index must be 0..7");
    nsgpu_ctx::KswWs &W = c->kws[ws_index];
    // workspace 0 works on the context's stream; workspace 1 owns one (the second half batch of the contig engine)
    if (ws_index >= 1 && !W.stream) NS_TRY(role_stream_create(&W.stream, "dp"));
    const hipStream_t S = ws_index == 0 ? c->stream : W.stream;
    if (!W.t_a) { NS_HIP(hipEventCreate(&W.t_a)); NS_HIP(hipEventCreate(&W.t_b)); }
    results.assign(n, KswResult());
    cig_off.assign(n + 1, 0);
    cigars.clear();
    c->kws[ws_index].pend_n = 0;
    if (n == 0) return NSGPU_OK;
    static const size_t kClass[3] = {4096, 16384, 65536};
    std::vector<uint32_t> order[4];          // fallback: one wave per problem (LDS classes 0..2, HBM slab 3)
    // workgroup kernel for the long, latency-bound problems of LDS classes 1 and 2: 256 threads (more waves only add barrier cost), up to 5 / 8 cells per
    // thread and row; the bulk of small gap fills (class 0) is throughput-bound and stays on one wave per problem
    std::vector<uint32_t> wg[3];
    // second generation (ksw2_reg.hip): state in registers; serves every problem whose parameters and shape keep its proofs valid
    std::vector<uint32_t> reg[KSW_REG_CLASSES];
    size_t reg_lds[KSW_REG_CLASSES] = {};
    static const int kWgThreads[3] = {0, 256, 256}, kWgMaxPos[3] = {0, 5, 8};      // widest sweep served: 1280 / 2048 cells (256 x 5 / 8 or 512 x 3 / 4)
    static const bool no_wg = getenv("NSGPU_KSW_NO_WG") != nullptr;      // debugging aid: fallback kernels only
    constexpr bool wg512 = true;          // 512 threads per long problem (8 waves; 4 waves measured slower: DP wall +7 %)
    // (latency twins of the one-wave classes -- one 128-cell block per wave for exact problems with many anti-diagonals -- were measured at the
    // one-group schedule in round 3: 700 rows: wait for the DP 5.4 instead of 4.7 s per step; a barrier per anti-diagonal costs more than the
    // second block of a lane saves.  Removed.)
    size_t p_total = 0, cig_total = 0, hbm_stride = 0;
    // per-problem sizes and classes on all host threads (a batch has ~10^4 problems and this thread is on the slot's critical
    // path), then one short serial pass for the running offsets and the class lists
    std::vector<uint32_t> &pb = W.h_pbytes;            // traceback bytes (64-aligned), 0 for an empty problem
    std::vector<uint8_t> &cl = W.h_class;              // 0..3 one-wave classes, 4 + c workgroup classes, 255 empty, 254 / 253 errors
    pb.resize(n), cl.resize(n);
    const size_t chunk = 1024, n_chunks = (n + chunk - 1) / chunk;
    std::vector<size_t> chunk_stride(n_chunks, 0);
    const KswClassCfg kcfg = ksw_class_config();
    par_for("align.dp_classify", n_chunks, [&](size_t ci) {
        size_t hs = 0;
        for (size_t i = ci * chunk, e = std::min(n, (ci + 1) * chunk); i < e; ++i) {
            const KswTask &t = tasks[i];
            KswResult &o = results[i];                 // reset state (ksw_reset_extz): what an empty problem returns
            o.max = 0; o.zdropped = 0; o.max_q = o.max_t = o.mqe_t = o.mte_q = -1; o.mqe = o.mte = o.score = KSW_NEG_INF; o.n_cigar = 0; o.reach_end = 0;
            pb[i] = 0;
            if (t.qlen < 0 || t.tlen < 0) { cl[i] = 254; continue; }
            if (t.flag & KSW_EZ_GENERIC_SC) { cl[i] = 253; continue; }
            if (t.qlen <= 0 || t.tlen <= 0) { cl[i] = 255; continue; }
            const size_t pbytes = (ksw_p_bytes(t.qlen, t.tlen, t.w) + 63) & ~(size_t)63;
            if (pbytes >= (1ull << 32)) { cl[i] = 252; continue; }      // traceback of one problem beyond 4 GiB (e.g. 50 kb x 50 kb unbanded)
            pb[i] = (uint32_t)pbytes;
            // (the <1,2> launch follows the <1,4> launch on the main stream and each lasts as long as its longest problem: the few LONG
            // problems of the narrow class -- a read's overhang against the last bases of a consensus: a thousand anti-diagonals and
            // more, where a gap fill of this width has at most 511 -- go with the <1,4> launch, which handles any narrower problem:
            // ksw_launch_class_hd, NSGPU_KSW_PROMOTE_ROWS)
            const int rcls = ksw_launch_class_hd(t.qlen, t.tlen, t.w, t.flag, pr, kcfg);
            if (rcls >= 0) { cl[i] = (uint8_t)(16 + rcls); continue; }
            const size_t need = ksw_lds_bytes(t.qlen, t.tlen, t.flag);
            const int cls = need <= kClass[0] ? 0 : need <= kClass[1] ? 1 : need <= kClass[2] ? 2 : 3;
            if (cls == 3) { const size_t hn = ksw_lds_bytes(t.qlen, t.tlen, 0); if (hn > hs) hs = hn; }
            const bool to_wg = cls >= 1 && cls < 3 && !no_wg && ksw_max_width(t.qlen, t.tlen, t.w) <= kWgThreads[cls] * kWgMaxPos[cls];
            cl[i] = (uint8_t)(to_wg ? 4 + cls : cls);
        }
        chunk_stride[ci] = hs;
    });
    for (size_t ci = 0; ci < n_chunks; ++ci) if (chunk_stride[ci] > hbm_stride) hbm_stride = chunk_stride[ci];
    for (size_t i = 0; i < n; ++i) {
        KswTask &t = tasks[i];
        NS_CHECK(cl[i] != 254, NSGPU_ERR_ARG, "ksw: negative length");
        NS_CHECK(cl[i] != 253, NSGPU_ERR_ARG, "ksw: KSW_EZ_GENERIC_SC is not on NanoSpring's path");
        NS_CHECK(cl[i] != 252, NSGPU_ERR_RANGE, "ksw: traceback matrix of one problem exceeds 4 GiB (band the problem or split it)");
        t.flag |= kcfg.flag_or;          // KSW_EZ_NS_* bits (ksw2_reg.hip): early exit unless switched off, the A/B switches
        t.out_idx = (uint32_t)i;
        t.p_off = p_total;
        t.cig_off = (uint32_t)cig_total;
        cig_off[i] = cig_total;
        if (cl[i] == 255) continue;                  // result stays "reset" (ksw_reset_extz), as the reference returns early
        p_total += pb[i];
        cig_total += (size_t)t.qlen + t.tlen + 2;
        NS_CHECK(cig_total < (1ull << 32), NSGPU_ERR_RANGE, "ksw batch too large (cigar pool)");
        if (cl[i] >= 16) {
            const int rc = cl[i] - 16;
            reg[rc].push_back((uint32_t)i);
            const size_t lb = ksw_reg_lds_bytes(rc, t.qlen);
            if (lb > reg_lds[rc]) reg_lds[rc] = lb;
        } else if (cl[i] >= 4) wg[cl[i] - 4].push_back((uint32_t)i);
        else order[cl[i]].push_back((uint32_t)i);
    }
    cig_off[n] = cig_total;
    NS_TRY(W.k_tasks.reserve(n * sizeof(KswTask)));
    NS_TRY(W.k_order.reserve(n * 4 + 16));
    NS_TRY(W.k_seqs.reserve(seq_bytes + 64));
    NS_TRY(W.k_p.reserve(p_total + 256));
    NS_TRY(W.k_cig.reserve((cig_total + 16) * 4));
    NS_TRY(W.k_res.reserve(n * sizeof(KswResult)));
    NS_HIP(hipMemcpyAsync(W.k_tasks.p, tasks.data(), n * sizeof(KswTask), hipMemcpyHostToDevice, S));
    NS_HIP(hipMemcpyAsync(W.k_seqs.p, seqs, seq_bytes, hipMemcpyHostToDevice, S));
    NS_HIP(hipMemcpyAsync(W.k_res.p, results.data(), n * sizeof(KswResult), hipMemcpyHostToDevice, S));
    std::vector<uint32_t> &flat = W.h_flat;          // stays alive while the upload may still be reading it
    flat.clear();
    size_t start[5] = {0, 0, 0, 0, 0}, wg_start[3] = {0, 0, 0}, reg_start[KSW_REG_CLASSES] = {};
    // big problems first inside a class (longest-processing-time-first): 64 buckets by the logarithm of the cell count, taken
    // in descending order -- the schedule only needs the rough order, a comparison sort of every batch does not pay
    // (measured, round 4: neither the order inside a class, nor LDS sized for a 6 kb query, nor thousands of surplus workgroups that leave at
    // once change a launch's duration -- the problems of a launch are all resident at once)
    auto lpt_order = [&](std::vector<uint32_t> &v) {
        if (v.size() < 2) return;
        uint32_t cnt[65] = {0};
        std::vector<uint8_t> &bk = W.h_bucket;
        bk.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const uint64_t cells = (uint64_t)tasks[v[i]].qlen * (uint64_t)tasks[v[i]].tlen;
            // 4 buckets per doubling: floor(log2) * 4 + the next two bits
            const int lg = cells > 1 ? 63 - __builtin_clzll(cells) : 0;
            const int b4 = lg >= 2 ? (int)((cells >> (lg - 2)) & 3) : 0;
            int b = lg * 4 + b4 - 40;                       // 2^10 cells -> bucket 0
            b = b < 0 ? 0 : b > 63 ? 63 : b;
            bk[i] = (uint8_t)(63 - b);                      // descending
            ++cnt[bk[i] + 1];
        }
        for (int i = 0; i < 64; ++i) cnt[i + 1] += cnt[i];
        std::vector<uint32_t> &tmp = W.h_tmp;
        tmp.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) tmp[cnt[bk[i]]++] = v[i];
        v.swap(tmp);
    };
    for (int k = 0; k < 4; ++k) {
        lpt_order(order[k]);
        start[k] = flat.size();
        flat.insert(flat.end(), order[k].begin(), order[k].end());
    }
    start[4] = flat.size();
    for (int k = 0; k < 3; ++k) {
        lpt_order(wg[k]);
        wg_start[k] = flat.size();
        flat.insert(flat.end(), wg[k].begin(), wg[k].end());
    }
    for (int k = 0; k < KSW_REG_CLASSES; ++k) {
        lpt_order(reg[k]);
        reg_start[k] = flat.size();
        flat.insert(flat.end(), reg[k].begin(), reg[k].end());
    }
    if (!flat.empty()) NS_HIP(hipMemcpyAsync(W.k_order.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, S));
    static const bool dbg = getenv("NSGPU_KSW_DEBUG") != nullptr;     // per-launch log (adds a sync per launch)
    NS_HIP(hipEventRecord(W.t_a, S));
    // The few long problems (extensions up to 5000 x 5000) are latency-bound on one wave each and leave the chip
    // idle: they run on side streams, concurrently with the bulk of small gap fills on the main stream.
    if (!W.side_stream[0]) {
        for (int i = 0; i < 3; ++i) { NS_TRY(role_stream_create(&W.side_stream[i], "dp_side")); NS_HIP(hipEventCreateWithFlags(&W.side_done[i], hipEventDisableTiming)); }      // the long problems must be dispatched first
        NS_HIP(hipEventCreateWithFlags(&W.side_fork, hipEventDisableTiming));
    }
    NS_HIP(hipEventRecord(W.side_fork, S));
    bool side_used[3] = {false, false, false};
    size_t n_ev = 0;
    uint64_t n_launch = 0;
    auto ev_at = [&](size_t i) -> hipEvent_t { while (W.ev.size() <= i) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); W.ev.push_back(e); } return W.ev[i]; };
    // register-resident classes: the multi-wave ones (long problems) first and on the high-priority side streams, the one-wave bulk last
    for (int k = KSW_REG_CLASSES - 1; k >= 0; --k) {
        const uint32_t m = (uint32_t)reg[k].size();
        if (!m) continue;
        hipStream_t st = S;
        if (k >= 2 && !dbg) { const int si = k >= 9 ? k - 9 : k == 8 ? 1 : k >= 6 ? k - 5 : k >= 4 ? 0 : k - 1; st = W.side_stream[si]; NS_HIP(hipStreamWaitEvent(st, W.side_fork, 0)); side_used[si] = true; }
        double dbg_t0 = 0;
        if (dbg) { NS_HIP(stream_wait(S)); dbg_t0 = now_ms(); }
        W.ev_class.resize(n_ev / 2 + 1); W.ev_class[n_ev / 2] = k;
        NS_HIP(hipEventRecord(ev_at(n_ev++), st));
        NS_TRY(ksw_reg_launch(k, st, m, reg_lds[k], W.k_tasks.as<KswTask>(), W.k_order.as<uint32_t>() + reg_start[k], pr, W.k_seqs.as<uint8_t>(), W.k_p.as<uint8_t>(),
                              W.k_cig.as<uint32_t>(), W.k_res.as<KswResult>()));
        NS_HIP(hipEventRecord(ev_at(n_ev++), st));
        ++n_launch;
        if (dbg) {
            NS_HIP(stream_wait(S));
            double cells = 0, mx = 0;
            for (uint32_t i : reg[k]) { const double x = (double)tasks[i].qlen * tasks[i].tlen; cells += x; if (x > mx) mx = x; }
            fprintf(stderr, "KSW reg class %d tasks %u cells %.3g max %.3g (q %d t %d) ms %.3f\n", k, m, cells, mx, tasks[reg[k][0]].qlen, tasks[reg[k][0]].tlen, now_ms() - dbg_t0);
        }
    }
    for (int k = 2; k >= 0; --k) {
        const uint32_t m = (uint32_t)wg[k].size();
        if (!m) continue;
        hipStream_t st = S;
        if (k > 0 && !dbg) { st = W.side_stream[k]; NS_HIP(hipStreamWaitEvent(st, W.side_fork, 0)); side_used[k] = true; }
        double dbg_t0 = 0;
        if (dbg) { NS_HIP(stream_wait(S)); dbg_t0 = now_ms(); }
        const uint32_t *ord = W.k_order.as<uint32_t>() + wg_start[k];
        NS_HIP(hipEventRecord(ev_at(n_ev++), st));
        if (k == 1)
        {
            if (wg512) hipLaunchKernelGGL((ksw_extd2_wg_kernel<512, 3>), dim3(m), dim3(512), kClass[1], st, W.k_tasks.as<KswTask>(), ord, m, pr, W.k_seqs.as<uint8_t>(),
                               W.k_p.as<uint8_t>(), W.k_cig.as<uint32_t>(), W.k_res.as<KswResult>());
            else
            hipLaunchKernelGGL((ksw_extd2_wg_kernel<256, 5>), dim3(m), dim3(256), kClass[1], st, W.k_tasks.as<KswTask>(), ord, m, pr, W.k_seqs.as<uint8_t>(),
                               W.k_p.as<uint8_t>(), W.k_cig.as<uint32_t>(), W.k_res.as<KswResult>());
        } else {
            static const hipError_t attr_wg8 = hipFuncSetAttribute(reinterpret_cast<const void *>(ksw_extd2_wg_kernel<256, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kClass[2]);
            NS_HIP(attr_wg8);
            static const hipError_t attr_wg8b = hipFuncSetAttribute(reinterpret_cast<const void *>(ksw_extd2_wg_kernel<512, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kClass[2]);
            NS_HIP(attr_wg8b);
            if (wg512) hipLaunchKernelGGL((ksw_extd2_wg_kernel<512, 4>), dim3(m), dim3(512), kClass[2], st, W.k_tasks.as<KswTask>(), ord, m, pr, W.k_seqs.as<uint8_t>(),
                               W.k_p.as<uint8_t>(), W.k_cig.as<uint32_t>(), W.k_res.as<KswResult>());
            else
            hipLaunchKernelGGL((ksw_extd2_wg_kernel<256, 8>), dim3(m), dim3(256), kClass[2], st, W.k_tasks.as<KswTask>(), ord, m, pr, W.k_seqs.as<uint8_t>(),
                               W.k_p.as<uint8_t>(), W.k_cig.as<uint32_t>(), W.k_res.as<KswResult>());
        }
        NS_HIP(hipGetLastError());
        NS_HIP(hipEventRecord(ev_at(n_ev++), st));
        ++n_launch;
        if (dbg) {
            NS_HIP(stream_wait(S));
            double cells = 0, mx = 0;
            for (uint32_t i : wg[k]) { const double x = (double)tasks[i].qlen * tasks[i].tlen; cells += x; if (x > mx) mx = x; }
            fprintf(stderr, "KSW class %d tasks %u cells %.3g max %.3g (q %d t %d) ms %.3f\n", k + 4, m, cells, mx, tasks[wg[k][0]].qlen, tasks[wg[k][0]].tlen, now_ms() - dbg_t0);
        }
    }
    for (int k = 2; k >= 0; --k) {
        const uint32_t m = (uint32_t)order[k].size();
        if (!m) continue;
        hipStream_t st = S;
        if (k > 0 && !dbg) { st = W.side_stream[k]; NS_HIP(hipStreamWaitEvent(st, W.side_fork, 0)); side_used[k] = true; }
        double dbg_t0 = 0;
        if (dbg) { NS_HIP(stream_wait(S)); dbg_t0 = now_ms(); }
        if (kClass[k] > 49152) {
            static const hipError_t attr_lds = hipFuncSetAttribute(reinterpret_cast<const void *>(ksw_extd2_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kClass[2]);
            NS_HIP(attr_lds);
        }
        NS_HIP(hipEventRecord(ev_at(n_ev++), st));
        hipLaunchKernelGGL(ksw_extd2_lds_kernel, dim3(m), dim3(64), kClass[k], st, W.k_tasks.as<KswTask>(),
                           W.k_order.as<uint32_t>() + start[k], m, pr, W.k_seqs.as<uint8_t>(), W.k_p.as<uint8_t>(), W.k_cig.as<uint32_t>(),
                           W.k_res.as<KswResult>());
        NS_HIP(hipGetLastError());
        NS_HIP(hipEventRecord(ev_at(n_ev++), st));
        ++n_launch;
        if (dbg) {
            NS_HIP(stream_wait(S));
            double cells = 0, mx = 0;
            for (uint32_t i : order[k]) { const double x = (double)tasks[i].qlen * tasks[i].tlen; cells += x; if (x > mx) mx = x; }
            fprintf(stderr, "KSW class %d tasks %u cells %.3g max %.3g (q %d t %d) ms %.3f\n", k, m, cells, mx, tasks[order[k][0]].qlen, tasks[order[k][0]].tlen, now_ms() - dbg_t0);
        }
    }
    if (!order[3].empty()) {
        NS_HIP(hipStreamWaitEvent(W.side_stream[0], W.side_fork, 0));
        side_used[0] = true;
        const uint32_t m = (uint32_t)order[3].size();
        const uint32_t wgs = m < 512u ? m : 512u;
        hbm_stride = (hbm_stride + 255) & ~(size_t)255;
        NS_TRY(W.k_slab.reserve((size_t)wgs * hbm_stride));
        NS_HIP(hipEventRecord(ev_at(n_ev++), W.side_stream[0]));
        hipLaunchKernelGGL(ksw_extd2_hbm_kernel, dim3(wgs), dim3(64), 0, W.side_stream[0], W.k_tasks.as<KswTask>(), W.k_order.as<uint32_t>() + start[3], m, pr,
                           W.k_seqs.as<uint8_t>(), W.k_p.as<uint8_t>(), W.k_cig.as<uint32_t>(), W.k_res.as<KswResult>(), W.k_slab.as<uint8_t>(),
                           hbm_stride);
        NS_HIP(hipGetLastError());
        NS_HIP(hipEventRecord(ev_at(n_ev++), W.side_stream[0]));
        ++n_launch;
    }
    for (int i = 0; i < 3; ++i)
        if (side_used[i]) { NS_HIP(hipEventRecord(W.side_done[i], W.side_stream[i])); NS_HIP(hipStreamWaitEvent(S, W.side_done[i], 0)); }
    NS_HIP(hipEventRecord(W.t_b, S));
    // compact the CIGARs on the device, then fetch results + used CIGAR entries only
    NS_TRY(W.k_coff.reserve((n + 2) * 8));
    {   // exclusive sum of the CIGAR lengths, read out of the result records by the scan itself (no pass to extract them first)
        auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), NcigarAt{W.k_res.as<KswResult>(), (uint32_t)n});
        size_t ws_bytes = 0;
        NS_HIP(rocprim::exclusive_scan(nullptr, ws_bytes, in, W.k_coff.as<uint64_t>(), (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), S));
        NS_TRY(W.scan_ws.reserve(ws_bytes + 16));
        NS_HIP(rocprim::exclusive_scan(W.scan_ws.p, ws_bytes, in, W.k_coff.as<uint64_t>(), (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), S));
    }
    W.pend_n = n, W.pend_n_ev = n_ev, W.pend_n_launch = n_launch;
    return NSGPU_OK;
}

// Second half: waits for the batch, compacts the CIGARs on the device and fetches results + used CIGAR entries.
int ksw_batch_collect(nsgpu_ctx *c, std::vector<KswTask> &tasks, std::vector<KswResult> &results, std::vector<uint32_t> &cigars,
                      std::vector<uint64_t> &cig_off, int ws_index)
{
    nsgpu_ctx::KswWs &W = c->kws[ws_index];
    const size_t n = W.pend_n, n_ev = W.pend_n_ev;
    const uint64_t n_launch = W.pend_n_launch;
    if (n == 0) return NSGPU_OK;
    NS_CHECK(n == tasks.size() && results.size() == n && cig_off.size() == n + 1, NSGPU_ERR_ARG, "ksw: collect does not match the launched batch");
    const hipStream_t S = ws_index == 0 ? c->stream : W.stream;
    W.pend_n = 0;
    // read-backs land in pinned memory (a copy into the pageable vectors would block inside the runtime, busy-waiting for the
    // DP kernels queued before it) and are copied out once the stream has drained
    NS_TRY(W.h_res.reserve(n * sizeof(KswResult) + 64));
    NS_TRY(W.h_coff.reserve((n + 1) * 8 + 64));
    NS_HIP(hipMemcpyAsync(W.h_res.p, W.k_res.p, n * sizeof(KswResult), hipMemcpyDeviceToHost, S));
    NS_HIP(hipMemcpyAsync(W.h_coff.p, W.k_coff.p, (n + 1) * 8, hipMemcpyDeviceToHost, S));
    NS_HIP(stream_wait(S));
    memcpy(results.data(), W.h_res.p, n * sizeof(KswResult));
    memcpy(cig_off.data(), W.h_coff.p, (n + 1) * 8);
    const uint64_t used = cig_off[n];
    NS_TRY(W.k_cig2.reserve((used + 16) * 4));
    if (used) {
        uint32_t grid = (uint32_t)((n + 3) / 4);
        if (grid > 16384u) grid = 16384u;
        hipLaunchKernelGGL(ksw_cigar_gather_kernel, dim3(grid), dim3(256), 0, S, W.k_tasks.as<KswTask>(), W.k_res.as<KswResult>(), (uint32_t)n,
                           W.k_coff.as<uint64_t>(), W.k_cig.as<uint32_t>(), W.k_cig2.as<uint32_t>());
        NS_HIP(hipGetLastError());
    }
    cigars.resize(used + 1);
    NS_TRY(W.h_cig.reserve((used + 1) * 4 + 64));
    if (used) NS_HIP(hipMemcpyAsync(W.h_cig.p, W.k_cig2.p, used * 4, hipMemcpyDeviceToHost, S));
    NS_HIP(stream_wait_short(S));
    if (used) memcpy(cigars.data(), W.h_cig.p, used * 4);
    float ms = 0;
    NS_HIP(hipEventElapsedTime(&ms, W.t_a, W.t_b));
    double sum_ms = 0, cells = 0, alg = 0;
    for (size_t i = 0; i + 1 < n_ev; i += 2) {
        float d = 0;
        if (hipEventElapsedTime(&d, W.ev[i], W.ev[i + 1]) == hipSuccess) { sum_ms += d; if (i / 2 < W.ev_class.size() && W.ev_class[i / 2] >= 0) { c->ksw_class_ms[W.ev_class[i / 2]] += d; ++c->ksw_class_n[W.ev_class[i / 2]]; } }
    }
    W.ev_class.assign(W.ev_class.size(), -1);
    for (auto &t : tasks) cells += (double)t.qlen * t.tlen;
    // algorithmic HBM bytes of a DP problem: both sequences in, CIGAR + result out (the traceback matrix is scratch)
    for (size_t i = 0; i < n; ++i) alg += (double)tasks[i].qlen + tasks[i].tlen + 4.0 * results[i].n_cigar + sizeof(KswResult);
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->ksw_kernel_ms += ms;
    c->ksw_kernel_sum_ms += sum_ms;
    c->ksw_cells += cells;
    c->ksw_alg_bytes += alg;
    c->ksw_launches += n_launch;
    return NSGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A batch whose tasks are written on the DEVICE by the plan kernel (plan.hip) -- the DP launch of the contig engine's slot without a host
// round trip between chaining and DP.  The host sizes buffers for n_slots task slots and upper bounds of the scratch, the plan kernel fills
// task descriptors, sequences and one launch list per register class, every class is launched over its list with the count read from
// device memory (grid = n_slots: the surplus workgroups leave at once), and the results reach the host as the kernels' own stores into
// pinned memory.
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
// One workgroup per alignment of the part (0: none of its problems runs in a late class; 1: the others): ksw_collect.hpp's hand-over behind a whole
// launch -- for batches whose DP kernels do not hand over themselves (NSGPU_KSW_NO_INLINE_COLLECT=1)
__global__ __launch_bounds__(64) void ksw_dev_collect_kernel(DvCollect dc, uint32_t part)
{
    __shared__ uint32_t s_off[256];
    const uint32_t b = blockIdx.x;
    const PlanOut o = dc.outs[b];
    if (o.flags || o.n_tasks == 0 || (o.slow != 0) != (part != 0)) return;
    dev_collect_pair(dc, b, threadIdx.x, s_off);
}

// the batch's last kernel: its counters for the host, then the word that says everything has been handed over
__global__ void ksw_dev_ctrl_kernel(const DvCtrl *__restrict__ ctrl, DvCtrl *__restrict__ h_ctrl)
{
    if (threadIdx.x == 0) {
        DvCtrl v = *ctrl;
        v.done = 0;
        *h_ctrl = v;
        __threadfence_system();
        h_ctrl->done = 1u;
    }
}
}  // namespace

// workgroups of a class's launch over its device-side list: every task slot for the narrow classes (the bulk), a few per alignment for the wide
// ones -- an empty workgroup of those still claims its tens of KB of LDS on a CU for the microseconds it lives.  What a list cannot take is
// left to the host by the plan kernel.
static uint32_t dev_class_grid(int cls, uint32_t n_slots, uint32_t n_pairs)
{
    if (cls == 3) return std::min<uint32_t>(n_slots, n_pairs / 2 + 16);          // (<8,5>: ~100 KB of LDS per workgroup)
    const uint32_t per_pair = cls == 0 || cls == 1 || cls == 4 || cls == 5 || cls == 9 ? 0u : cls == 7 ? 4u : 8u;
    return per_pair ? std::min<uint32_t>(n_slots, per_pair * n_pairs + 32) : n_slots;
}

int ksw_dev_prepare(nsgpu_ctx *c, int ws_index, uint32_t n_slots, uint32_t n_pairs, uint64_t seq_bytes_bound, hipStream_t st, PlanDp &dp)
{
    NS_CHECK(ws_index >= 0 && ws_index <= 3, NSGPU_ERR_ARG, "ksw_dev_prepare: workspace index must be 0..3");
    nsgpu_ctx::KswWs &W = c->kws[ws_index];
    NS_CHECK(!W.dv_pending, NSGPU_ERR_ARG, "ksw_dev_prepare: the workspace's last device batch was not collected");
    NS_CHECK(seq_bytes_bound < (1ull << 32) - 65536, NSGPU_ERR_RANGE, "device-planned DP batch: sequence pool beyond 4 GiB");
    W.dv_slots = n_slots, W.dv_pairs = n_pairs;
    const uint64_t p_cap = std::max<uint64_t>(W.dv_p_hint, 768ull << 20);
    const uint64_t cig_cap = seq_bytes_bound + 2ull * n_slots + 64;
    NS_TRY(W.dv_tasks.reserve((size_t)n_slots * sizeof(KswTask) + 64));
    NS_TRY(W.dv_list.reserve((size_t)KSW_REG_CLASSES * n_slots * 4 + 64));
    NS_TRY(W.dv_ctrl.reserve(sizeof(DvCtrl)));
    NS_TRY(W.dv_seqs.reserve(seq_bytes_bound + 64));
    NS_TRY(W.dv_p.reserve(p_cap + 256));
    NS_TRY(W.dv_cig.reserve(cig_cap * 4));
    NS_TRY(W.dv_res.reserve((size_t)n_slots * sizeof(KswResult) + 64));
    NS_TRY(W.dv_coff.reserve(((size_t)n_slots + 2) * 8));
    NS_TRY(W.dv_tpair.reserve((size_t)n_slots * 4 + 64));
    NS_TRY(W.dv_pdone.reserve((size_t)n_pairs * 4 + 64));
    // the two clears on the workspace's own stream (idle: its last batch was collected slots ago), beside the seeding and chaining kernels that `st`
    // is busy with; `st` -- the plan kernel comes next on it -- waits for them.  (On `st` they were four fill kernels, ~70 us, between the chaining and
    // the plan kernel of every slot: the kernel trace.)
    {
        if (ws_index >= 1 && !W.stream) NS_TRY(role_stream_create(&W.stream, "dp"));
        const hipStream_t cs = ws_index == 0 ? c->stream : W.stream;
        if (cs == st) {
            NS_HIP(hipMemsetAsync(W.dv_pdone.p, 0, (size_t)n_pairs * 4 + 4, st));
            NS_HIP(hipMemsetAsync(W.dv_ctrl.p, 0, sizeof(DvCtrl), st));
        } else {
            if (!W.dv_clear_ev) NS_HIP(hipEventCreateWithFlags(&W.dv_clear_ev, hipEventDisableTiming));
            NS_HIP(hipMemsetAsync(W.dv_pdone.p, 0, (size_t)n_pairs * 4 + 4, cs));
            NS_HIP(hipMemsetAsync(W.dv_ctrl.p, 0, sizeof(DvCtrl), cs));
            NS_HIP(hipEventRecord(W.dv_clear_ev, cs));
            NS_HIP(hipStreamWaitEvent(st, W.dv_clear_ev, 0));
        }
    }
    // (the task slots and their results are cleared by the plan kernel itself: every slot belongs to one alignment's wave)
    dp.tasks = W.dv_tasks.as<KswTask>(), dp.res = W.dv_res.as<KswResult>(), dp.class_list = W.dv_list.as<uint32_t>();
    dp.class_cnt = W.dv_ctrl.as<DvCtrl>()->class_cnt, dp.n_slots = n_slots, dp.seqs = W.dv_seqs.as<uint8_t>();
    dp.cursors = W.dv_ctrl.as<DvCtrl>()->cursors;
    dp.task_pair = W.dv_tpair.as<uint32_t>();
    dp.pair_done = W.dv_pdone.as<uint32_t>();
    for (int k = 0; k < KSW_REG_CLASSES; ++k) dp.class_grid[k] = dev_class_grid(k, n_slots, n_pairs);
    dp.p_cap = p_cap, dp.cig_cap = (uint32_t)std::min<uint64_t>(cig_cap, 0xffffffffull), dp.seq_cap = (uint32_t)seq_bytes_bound;
    return NSGPU_OK;
}

int ksw_dev_launch(nsgpu_ctx *c, int ws_index, int max_qlen, const KswParams &pr, hipEvent_t after, const PlanPair *pairs, const PlanOut *outs, uint32_t n_pairs, bool two_phase)
{
    nsgpu_ctx::KswWs &W = c->kws[ws_index];
    const uint32_t n = W.dv_slots;
    if (n == 0) return NSGPU_OK;
    if (ws_index >= 1 && !W.stream) NS_TRY(role_stream_create(&W.stream, "dp"));
    const hipStream_t S = ws_index == 0 ? c->stream : W.stream;
    if (!W.dv_a) { NS_HIP(hipEventCreate(&W.dv_a)); NS_HIP(hipEventCreate(&W.dv_b)); NS_HIP(hipEventCreateWithFlags(&W.dv_part0, hipEventDisableTiming)); }
    if (!W.side_stream[0]) {
        for (int i = 0; i < 3; ++i) { NS_TRY(role_stream_create(&W.side_stream[i], "dp_side")); NS_HIP(hipEventCreateWithFlags(&W.side_done[i], hipEventDisableTiming)); }
        NS_HIP(hipEventCreateWithFlags(&W.side_fork, hipEventDisableTiming));
    }
    NS_HIP(hipStreamWaitEvent(S, after, 0));
    NS_HIP(hipEventRecord(W.dv_a, S));
    NS_HIP(hipEventRecord(W.side_fork, S));
    const KswClassCfg &kc = ksw_class_config();
    // the classes the plan kernel's rule (ksw_launch_class_hd) can name under the current switches
    uint32_t classes = 1u << 0 | 1u << 1;
    classes |= 1u << 8 | 1u << 3;
    if (two_phase && kc.long_rows > 0) classes |= 1u << 12;
    W.dv_classes = classes;
    bool side_used[3] = {false, false, false};
    while (W.dv_ev.size() < 2 * KSW_REG_CLASSES) { hipEvent_t e = nullptr; NS_HIP(hipEventCreate(&e)); W.dv_ev.push_back(e); }
    DvCtrl *ctrl = W.dv_ctrl.as<DvCtrl>();
    // pinned landing zones: results and CIGAR offsets per task slot, the CIGAR arena, one status word per alignment
    const uint64_t hcap = std::max<uint64_t>(W.dv_hcig_hint, 1u << 20);          // CIGAR entries the arena takes
    NS_TRY(W.hv_cig.reserve(hcap * 4 + 64));
    NS_TRY(W.hv_res.reserve((size_t)n * sizeof(KswResult) + 64));
    NS_TRY(W.hv_coff.reserve(((size_t)n + 1) * 8 + 64));
    NS_TRY(W.hv_ctrl.reserve(sizeof(DvCtrl)));
    NS_TRY(W.hv_status.reserve((size_t)n_pairs * 4 + 64));
    NS_TRY(W.hv_check.reserve((size_t)n_pairs * 4 + 64));
    // nothing of the batch before may look like this one's: status and check words cleared, offsets beyond every arena, and a new epoch in the
    // check word's seed (a status word that overtook its data must find no self-consistent old hand-over under it)
    memset(W.hv_status.p, 0, (size_t)n_pairs * 4);
    memset(W.hv_check.p, 0, (size_t)n_pairs * 4);
    memset(W.hv_coff.p, 0xff, ((size_t)n + 1) * 8);
    ++W.dv_epoch;
    W.hv_ctrl.as<DvCtrl>()->done = 0;
    W.dv_hcig_cap = hcap, W.dv_npairs_launched = n_pairs, W.dv_two_phase = two_phase;
    // The DP kernels hand every alignment over themselves, the moment its last problem is done (ksw_collect.hpp) -- the host finishes and applies
    // it while the launch's longer problems still run.  NSGPU_KSW_NO_INLINE_COLLECT=1: a collecting kernel behind the bulk classes and one
    // behind everything, as in the first version (A/B switch).
    static const bool no_inline = getenv("NSGPU_KSW_NO_INLINE_COLLECT") != nullptr;
    W.dv_inline = !no_inline;
    DvCollect dc{pairs, outs, W.dv_tasks.as<KswTask>(), W.dv_res.as<KswResult>(), W.dv_cig.as<uint32_t>(), W.dv_tpair.as<uint32_t>(), W.dv_inline ? W.dv_pdone.as<uint32_t>() : nullptr,
                 W.hv_res.as<KswResult>(), W.hv_coff.as<uint64_t>(), W.hv_cig.as<uint32_t>(), hcap, W.hv_status.as<uint32_t>(), W.hv_check.as<uint32_t>(), W.dv_epoch, ctrl};
    auto launch_class = [&](int k, hipStream_t st) -> int {
        NS_HIP(hipEventRecord(W.dv_ev[2 * k], st));
        NS_TRY(ksw_reg_launch(k, st, dev_class_grid(k, n, W.dv_pairs), ksw_reg_lds_bytes(k, max_qlen), W.dv_tasks.as<KswTask>(), W.dv_list.as<uint32_t>() + (size_t)k * n, pr, W.dv_seqs.as<uint8_t>(),
                              W.dv_p.as<uint8_t>(), W.dv_cig.as<uint32_t>(), W.dv_res.as<KswResult>(), ctrl->class_cnt + k, &dc));
        NS_HIP(hipEventRecord(W.dv_ev[2 * k + 1], st));
        return NSGPU_OK;
    };
    auto collect = [&](uint32_t part) {
        if (W.dv_inline) return;
        hipLaunchKernelGGL(ksw_dev_collect_kernel, dim3(n_pairs), dim3(64), 0, S, dc, part);
    };
    // the late classes -- the multi-wave ones (wide problems) and, in a two-part batch, the long problems of the one-wave classes -- on the side
    // streams; the bulk on the main stream, and behind it the first part of the results
    for (int k = KSW_REG_CLASSES - 1; k >= 2; --k) {
        if (!(classes >> k & 1)) continue;
        const int si = k == 12 ? 2 : k == 8 ? 1 : k == 3 ? 0 : k >= 6 ? k - 5 : k >= 4 ? 0 : k - 1;        // (each of the three late classes on a stream of its own)
        hipStream_t st = W.side_stream[si];
        // (straight behind the plan kernel's event, not behind the main stream's wait for it: one cross-stream hop less -- ~25 us -- in front of the late classes)
        if (!side_used[si]) NS_HIP(hipStreamWaitEvent(st, after, 0));
        side_used[si] = true;
        NS_TRY(launch_class(k, st));
    }
    // In a two-part batch the two bulk classes run side by side: what counts is when the LAST of the ordinary problems is done (the first part
    // of the results), and back to back they take as long as the late classes do.  (A one-part batch launches them one after the other: the
    // <1,2> launch behind the <1,4> launch measured better there -- they share SIMDs otherwise.)
    if (two_phase && (classes & 3u) == 3u) {
        if (!W.bulk_stream) { NS_TRY(role_stream_create(&W.bulk_stream, "dp")); NS_HIP(hipEventCreateWithFlags(&W.bulk_done, hipEventDisableTiming)); }
        NS_HIP(hipStreamWaitEvent(W.bulk_stream, after, 0));
        NS_TRY(launch_class(1, W.bulk_stream));
        NS_HIP(hipEventRecord(W.bulk_done, W.bulk_stream));
        NS_TRY(launch_class(0, S));
        NS_HIP(hipStreamWaitEvent(S, W.bulk_done, 0));
    } else
    for (int k = 1; k >= 0; --k) if (classes >> k & 1) NS_TRY(launch_class(k, S));
    if (two_phase) {
        collect(0u);
        NS_HIP(hipEventRecord(W.dv_part0, S));
    }
    for (int i = 0; i < 3; ++i)
        if (side_used[i]) { NS_HIP(hipEventRecord(W.side_done[i], W.side_stream[i])); NS_HIP(hipStreamWaitEvent(S, W.side_done[i], 0)); }
    NS_HIP(hipEventRecord(W.dv_b, S));
    if (!two_phase) collect(0u);
    collect(1u);
    hipLaunchKernelGGL(ksw_dev_ctrl_kernel, dim3(1), dim3(64), 0, S, ctrl, W.hv_ctrl.as<DvCtrl>());
    NS_HIP(hipGetLastError());
    W.dv_pending = true;
    return NSGPU_OK;
}

static KswDevResults dev_results_of(const nsgpu_ctx::KswWs &W)
{
    return KswDevResults{W.hv_res.as<KswResult>(), W.hv_coff.as<uint64_t>(), W.hv_cig.as<uint32_t>(), W.hv_status.as<uint32_t>(), W.hv_check.as<uint32_t>(), W.dv_hcig_cap, W.dv_slots, W.dv_epoch};
}

// The landing zones of the workspace's batch in flight, without waiting for anything: when its DP kernels hand the alignments over themselves
// (ksw_collect.hpp) out.status says per alignment whether its results are there, and *done that the whole batch has been handed over.
// false: no such batch (the caller waits for the parts with ksw_dev_collect).
bool ksw_dev_poll(nsgpu_ctx *c, int ws_index, KswDevResults &out, const volatile uint32_t *&done)
{
    nsgpu_ctx::KswWs &W = c->kws[ws_index];
    out = KswDevResults{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    done = nullptr;
    if (!W.dv_pending || !W.dv_inline) return false;
    out = dev_results_of(W);
    done = &W.hv_ctrl.as<DvCtrl>()->done;
    return true;
}

// part 0: what is behind the bulk of the one-wave problems (a two-part batch only: otherwise nothing is there before part 1); part 1: everything
int ksw_dev_collect(nsgpu_ctx *c, int ws_index, int part, KswDevResults &out)
{
    nsgpu_ctx::KswWs &W = c->kws[ws_index];
    out = KswDevResults{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    if (!W.dv_pending) return NSGPU_OK;
    if (part == 0) {
        if (!W.dv_two_phase) return NSGPU_OK;
        NS_HIP(event_wait(W.dv_part0));
        out = dev_results_of(W);
        return NSGPU_OK;
    }
    W.dv_pending = false;
    const DvCtrl *hc = W.hv_ctrl.as<DvCtrl>();
    // (the closing word is up already when the caller watched the hand-overs to their end: the stream is a few microseconds from complete,
    // and a sleeping wait would cost a wake-up on a busy host -- 0.1 ms per slot)
    if (__atomic_load_n(&hc->done, __ATOMIC_ACQUIRE)) NS_HIP(stream_wait_short(ws_index == 0 ? c->stream : W.stream));
    else NS_HIP(stream_wait(ws_index == 0 ? c->stream : W.stream));
    // scratch that did not fit: the plan kernel left those alignments to the host; larger next time
    if (hc->cursors[0] > std::max<uint64_t>(W.dv_p_hint, 768ull << 20)) W.dv_p_hint = hc->cursors[0] + hc->cursors[0] / 2;
    if (hc->cig_out > W.dv_hcig_cap) W.dv_hcig_hint = hc->cig_out + hc->cig_out / 2;
    out = dev_results_of(W);
    float ms = 0;
    NS_HIP(hipEventElapsedTime(&ms, W.dv_a, W.dv_b));
    double sum_ms = 0;
    uint64_t n_launch = 0;
    for (int k = 0; k < KSW_REG_CLASSES; ++k)
        if (W.dv_classes >> k & 1) {
            // (every class that was launched counts, whether the plan kernel gave it problems or not: an empty launch is a launch to rocprofv3
            // too, and `launches` / `avg_launch_ms` of the bench line are compared with its kernel trace)
            float d = 0;
            if (hipEventElapsedTime(&d, W.dv_ev[2 * k], W.dv_ev[2 * k + 1]) == hipSuccess) { sum_ms += d; if (hc->class_cnt[k]) { c->ksw_class_ms[k] += d; ++c->ksw_class_n[k]; } }
            ++n_launch;
        }
    std::lock_guard<std::mutex> lk(c->stat_m);
    c->ksw_kernel_ms += ms;
    c->ksw_kernel_sum_ms += sum_ms;
    c->ksw_cells += (double)hc->cells;
    c->ksw_alg_bytes += (double)hc->alg_bytes;
    c->ksw_launches += n_launch;
    return NSGPU_OK;
}

}  // namespace nsgpu

using namespace nsgpu;

extern "C" int nsgpu_ksw_extd2_batch(nsgpu_ctx *c, uint32_t n, const uint8_t *seqs, const uint64_t *qoff, const int32_t *qlen,
                                     const uint64_t *toff, const int32_t *tlen, const int32_t *w, const int32_t *zdrop,
                                     const int32_t *end_bonus, const int32_t *flag, const nsgpu_ksw_params *prm, nsgpu_ksw_ez *ez_out,
                                     uint64_t **cigar_off_out, uint32_t **cigar_out)
{
    NS_CHECK(c && (n == 0 || (seqs && qoff && qlen && toff && tlen && w && zdrop && end_bonus && flag)) && prm && ez_out && cigar_off_out && cigar_out,
             NSGPU_ERR_ARG, "nsgpu_ksw_extd2_batch: null argument");
    NS_HIP(hipSetDevice(c->prm.device));
    std::vector<KswTask> tasks(n);
    size_t seq_bytes = 0;
    for (uint32_t i = 0; i < n; ++i) {
        KswTask &t = tasks[i];
        NS_CHECK(qoff[i] + (uint64_t)(qlen[i] > 0 ? qlen[i] : 0) < (1ull << 32) && toff[i] + (uint64_t)(tlen[i] > 0 ? tlen[i] : 0) < (1ull << 32),
                 NSGPU_ERR_RANGE, "sequence pool larger than 4 GiB");
        t.qoff = (uint32_t)qoff[i]; t.toff = (uint32_t)toff[i]; t.qlen = qlen[i]; t.tlen = tlen[i]; t.w = w[i]; t.zdrop = zdrop[i];
        t.end_bonus = end_bonus[i]; t.flag = flag[i];
        seq_bytes = std::max<size_t>(seq_bytes, std::max<size_t>(qoff[i] + (qlen[i] > 0 ? qlen[i] : 0), toff[i] + (tlen[i] > 0 ? tlen[i] : 0)));
    }
    KswParams pr;
    pr.sc_mch = prm->a < 0 ? -prm->a : prm->a;               // ksw_gen_simple_mat, align.c:9-22
    pr.sc_mis = prm->b > 0 ? -prm->b : prm->b;
    pr.sc_ambi = prm->sc_ambi > 0 ? -prm->sc_ambi : prm->sc_ambi;
    pr.q = prm->q; pr.e = prm->e; pr.q2 = prm->q2; pr.e2 = prm->e2;
    {
        int mn = pr.sc_mis < pr.sc_ambi ? pr.sc_mis : pr.sc_ambi;
        if (pr.sc_mch < mn) mn = pr.sc_mch;
        int qq = pr.q, ee = pr.e;
        if (pr.q2 + pr.e2 < pr.q + pr.e) qq = pr.q2, ee = pr.e2;
        NS_CHECK(-mn <= 2 * (qq + ee), NSGPU_ERR_ARG, "ksw: -min_sc > 2*(q+e): the reference returns without aligning (ksw2_extd2_sse.c:99)");
    }
    std::vector<KswResult> res;
    std::vector<uint32_t> cig;
    std::vector<uint64_t> coff;
    NS_TRY(ksw_run_batch(c, tasks, seqs, seq_bytes, pr, res, cig, coff));
    uint64_t *oo = (uint64_t *)malloc(((size_t)n + 1) * 8);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; ++i) tot += res[i].n_cigar;
    uint32_t *oc = (uint32_t *)malloc((tot + 1) * 4);
    NS_CHECK(oo && oc, NSGPU_ERR_NOMEM, "malloc failed");
    tot = 0;
    for (uint32_t i = 0; i < n; ++i) {
        oo[i] = tot;
        memcpy(oc + tot, cig.data() + coff[i], (size_t)res[i].n_cigar * 4);
        tot += res[i].n_cigar;
        memcpy(&ez_out[i], &res[i], sizeof(KswResult));
    }
    oo[n] = tot;
    *cigar_off_out = oo;
    *cigar_out = oc;
    return NSGPU_OK;
}
