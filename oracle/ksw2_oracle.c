/*
 * ksw2_oracle.c -- scalar, lane-exact restatement of minimap2's
 * ksw_extd2_sse (minimap2/ksw2_extd2_sse.c:34-401, v2.17-r974-dirty) plus
 * ksw_backtrack / ksw_apply_zdrop / ksw_reset_extz (minimap2/ksw2.h:103-176).
 *
 * TEST INFRASTRUCTURE ONLY (parity checker + cpu_baseline leg of bench.py).
 *
 * Pinning: tests/test_ksw2_oracle.py compares every output field and the CIGAR
 * with the reference's own ksw_extd2_sse (oracle/_ref/libmm2ref.so) on seeded
 * random problems for every flag combination NanoSpring reaches (0x08, 0x00,
 * 0x40, 0xC2; align.c:690-778), including band-limited and Z-dropped cases, and
 * with tests/golden/ksw2_cases.npz elsewhere.
 *
 * "Lane-exact": the SSE kernel works on 16-cell blocks, so cells outside the
 * band but inside the first/last block of an anti-diagonal are computed from
 * stale array contents and can feed in-band cells once the band (w) binds.  To
 * be bit-identical in those cases too, this file keeps the reference's memory
 * picture: int8 arrays u,v,x,y,x2,y2 of tlen_*16 cells, then the score row s,
 * the target copy sf and the reversed query qr laid out CONTIGUOUSLY (the 16-byte
 * stores of the score loop may run from s into sf, and reads of qr/sf may run
 * across the seams), all zero-initialised as kcalloc leaves them, with int8
 * wrap-around arithmetic.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KSW_NEG_INF -0x40000000
#define KSW_EZ_SCORE_ONLY  0x01
#define KSW_EZ_RIGHT       0x02
#define KSW_EZ_GENERIC_SC  0x04
#define KSW_EZ_APPROX_MAX  0x08
#define KSW_EZ_APPROX_DROP 0x10
#define KSW_EZ_EXTZ_ONLY   0x40
#define KSW_EZ_REV_CIGAR   0x80

typedef struct {
    uint32_t max; int32_t zdropped;
    int32_t max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
} oracle_ez_t;

static inline int8_t w8(int v) { return (int8_t)(uint8_t)(v & 0xff); }

/* ksw2.h:103-117 */
static uint32_t *push_cigar(int *n_cigar, int *m_cigar, uint32_t *cigar, uint32_t op, int len)
{
    if (*n_cigar == 0 || op != (cigar[(*n_cigar) - 1] & 0xf)) {
        if (*n_cigar == *m_cigar) {
            *m_cigar = *m_cigar ? (*m_cigar) << 1 : 4;
            cigar = (uint32_t *)realloc(cigar, (size_t)(*m_cigar) << 2);
        }
        cigar[(*n_cigar)++] = (uint32_t)len << 4 | op;
    } else cigar[(*n_cigar) - 1] += (uint32_t)len << 4;
    return cigar;
}

/* ksw2.h:119-151, is_rot = 1, min_intron_len = 0 */
static void backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0,
                      int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{
    int n_cigar = 0, m_cigar = *m_cigar_, i = i0, j = j0, r, state = 0;
    uint32_t *cigar = *cigar_, tmp;
    while (i >= 0 && j >= 0) {
        int force_state = -1;
        r = i + j;
        if (i < off[r]) force_state = 2;
        if (off_end && i > off_end[r]) force_state = 1;
        tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 0, 1), --i, --j;
        else if (state == 1 || state == 3) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, 1), --i;
        else cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, 1), --j;
    }
    if (i >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 2, i + 1);
    if (j >= 0) cigar = push_cigar(&n_cigar, &m_cigar, cigar, 1, j + 1);
    if (!is_rev)
        for (i = 0; i < n_cigar >> 1; ++i)
            tmp = cigar[i], cigar[i] = cigar[n_cigar - 1 - i], cigar[n_cigar - 1 - i] = tmp;
    *m_cigar_ = m_cigar, *n_cigar_ = n_cigar, *cigar_ = cigar;
}

/* ksw2.h:160-176, is_rot = 1 */
static int apply_zdrop(oracle_ez_t *ez, int32_t H, int r, int t, int zdrop, int8_t e)
{
    if (H > (int32_t)ez->max) {
        ez->max = H, ez->max_t = t, ez->max_q = r - t;
    } else if (t >= ez->max_t && r - t >= ez->max_q) {
        int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
        l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {   /* ez->max is uint32:31; promoted as int in C */
            ez->zdropped = 1;
            return 1;
        }
    }
    return 0;
}

/* mat is the 5x5 matrix of ksw_gen_simple_mat (align.c:9-22): mat[0]=a, mat[1]=-b, mat[24]=-sc_ambi.
 * Returns the number of CIGAR ops; cigar_out receives at most cigar_cap of them. */
int oracle_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t sc_mch, int8_t sc_mis, int8_t sc_ambi_mat,
                     int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                     oracle_ez_t *ez, uint32_t *cigar_out, int cigar_cap)
{
    const int m = 5;
    int r, t, qe, qe2, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, wl, wr, long_thres, long_diff, T16;
    int with_cigar = !(flag & KSW_EZ_SCORE_ONLY), approx_max = !!(flag & KSW_EZ_APPROX_MAX);
    int32_t *H = 0, H0 = 0, last_H0_t = 0;
    int8_t *u, *v, *x, *y, *x2, *y2, *s, sc_N;
    uint8_t *sf, *qr, *mem, *p = 0;
    uint32_t *cigar = 0;
    int m_cigar = 0, n_cigar = 0;

    /* ksw_reset_extz */
    ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
    ez->max = 0, ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
    ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
    if (qlen <= 0 || tlen <= 0) return 0;
    if (flag & KSW_EZ_GENERIC_SC) return -1;   /* never set by mm_align_pair */

    if (q2 + e2 < q + e) { int8_t tt; tt = q, q = q2, q2 = tt, tt = e, e = e2, e2 = tt; }
    qe = q + e, qe2 = q2 + e2;
    sc_N = sc_ambi_mat == 0 ? (int8_t)-e2 : sc_ambi_mat;

    if (w < 0) w = tlen > qlen ? tlen : qlen;
    wl = wr = w;
    tlen_ = (tlen + 15) / 16;
    n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    qlen_ = (qlen + 15) / 16;
    {
        int max_sc = sc_mch, min_sc = sc_mis;
        if (sc_ambi_mat > max_sc) max_sc = sc_ambi_mat;
        if (sc_ambi_mat < min_sc) min_sc = sc_ambi_mat;
        if (sc_mis > max_sc) max_sc = sc_mis;
        if (sc_mch < min_sc) min_sc = sc_mch;
        if (-min_sc > 2 * (q + e)) return 0;
    }
    long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    long_diff = long_thres * (e - e2) - (q2 - q) - e2;

    T16 = tlen_ * 16;
    mem = (uint8_t *)calloc((size_t)T16 * 8 + (size_t)(qlen_ + 1) * 16 + 64, 1);
    u = (int8_t *)mem, v = u + T16, x = v + T16, y = x + T16, x2 = y + T16, y2 = x2 + T16;
    s = y2 + T16, sf = (uint8_t *)(s + T16), qr = sf + T16;
    memset(u, w8(-q - e), T16); memset(v, w8(-q - e), T16);
    memset(x, w8(-q - e), T16); memset(y, w8(-q - e), T16);
    memset(x2, w8(-q2 - e2), T16); memset(y2, w8(-q2 - e2), T16);
    if (!approx_max) {
        H = (int32_t *)malloc((size_t)T16 * 4);
        for (t = 0; t < T16; ++t) H[t] = KSW_NEG_INF;
    }
    if (with_cigar) {
        p = (uint8_t *)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
        off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
        off_end = off + qlen + tlen - 1;
    }
    for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
    memcpy(sf, target, tlen);

    for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1, st0, en0;
        int8_t x1, x21, v1;
        const uint8_t *qrr = qr + (qlen - 1 - r);     /* may point below qr (into sf): contiguous on purpose */
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
        if (en > (r + wl) >> 1) en = (r + wl) >> 1;
        if (st > en) { ez->zdropped = 1; break; }
        st0 = st, en0 = en;
        st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
            else x1 = w8(-q - e), x21 = w8(-q2 - e2), v1 = w8(-q - e);
        } else {
            x1 = w8(-q - e), x21 = w8(-q2 - e2);
            v1 = r == 0 ? w8(-q - e) : r < long_thres ? w8(-e) : r == long_thres ? w8(long_diff) : w8(-e2);
        }
        if (en >= r) {
            y[r] = w8(-q - e), y2[r] = w8(-q2 - e2);
            u[r] = r == 0 ? w8(-q - e) : r < long_thres ? w8(-e) : r == long_thres ? w8(long_diff) : w8(-e2);
        }
        /* score row: 16-byte blocks starting at st0 (unaligned), may run past en0 and past s into sf */
        for (t = st0; t <= en0; t += 16) {
            int i;
            int8_t tmp[16];
            for (i = 0; i < 16; ++i) {
                uint8_t sq = sf[t + i], sq2 = qrr[t + i];
                int8_t z = sq == sq2 ? sc_mch : sc_mis;
                if (sq == (uint8_t)(m - 1) || sq2 == (uint8_t)(m - 1)) z = sc_N;
                tmp[i] = z;
            }
            memcpy(s + t, tmp, 16);      /* loads complete before the store, as in SSE */
        }
        if (with_cigar) off[r] = st, off_end[r] = en;
        {
            uint8_t *pr = with_cigar ? p + (size_t)r * n_col_ * 16 - st : 0;
            int8_t xc = x1, x2c = x21, vc = v1;       /* carried (t-1) values of the previous anti-diagonal */
            const int right = !!(flag & KSW_EZ_RIGHT);
            for (t = st; t <= en; ++t) {
                int8_t z = s[t], a, b, a2, b2, xt1 = xc, x2t1 = x2c, vt1 = vc, ut = u[t], d = 0, tmp;
                xc = x[t], x2c = x2[t], vc = v[t];
                a = w8(xt1 + vt1), b = w8(y[t] + ut), a2 = w8(x2t1 + vt1), b2 = w8(y2[t] + ut);
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
                u[t] = w8(z - vt1);
                v[t] = w8(z - ut);
                tmp = w8(z - q);  a = w8(a - tmp);  b = w8(b - tmp);
                tmp = w8(z - q2); a2 = w8(a2 - tmp); b2 = w8(b2 - tmp);
                if (!right) {
                    x[t]  = w8((a  > 0 ? a  : 0) - qe);  if (a  > 0) d |= 0x08;
                    y[t]  = w8((b  > 0 ? b  : 0) - qe);  if (b  > 0) d |= 0x10;
                    x2[t] = w8((a2 > 0 ? a2 : 0) - qe2); if (a2 > 0) d |= 0x20;
                    y2[t] = w8((b2 > 0 ? b2 : 0) - qe2); if (b2 > 0) d |= 0x40;
                } else {
                    x[t]  = w8((0 > a  ? 0 : a ) - qe);  if (!(0 > a )) d |= 0x08;
                    y[t]  = w8((0 > b  ? 0 : b ) - qe);  if (!(0 > b )) d |= 0x10;
                    x2[t] = w8((0 > a2 ? 0 : a2) - qe2); if (!(0 > a2)) d |= 0x20;
                    y2[t] = w8((0 > b2 ? 0 : b2) - qe2); if (!(0 > b2)) d |= 0x40;
                }
                if (pr) pr[t] = (uint8_t)d;
            }
        }
        if (!approx_max) {
            int32_t max_H, max_t;
            if (r > 0) {
                int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
                max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
                max_t = en0;
                for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
                for (t = st0; t < en1; t += 4)
                    for (i = 0; i < 4; ++i) {
                        H[t + i] += (int32_t)v[t + i];
                        if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
                    }
                for (i = 0; i < 4; ++i)
                    if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
                for (; t < en0; ++t) {
                    H[t] += (int32_t)v[t];
                    if (H[t] > max_H) max_H = H[t], max_t = t;
                }
            } else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
            if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
            if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
            if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0;
                    else H0 += d1, ++last_H0_t;
                } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                    H0 += v[last_H0_t];
                } else {
                    ++last_H0_t, H0 += u[last_H0_t];
                }
            } else H0 = v[0] - qe, last_H0_t = 0;
            if ((flag & KSW_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
        }
        last_st = st, last_en = en;
    }
    free(mem);
    if (!approx_max) free(H);
    if (with_cigar) {
        int rev_cigar = !!(flag & KSW_EZ_REV_CIGAR);
        if (!ez->zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) {
            backtrack(rev_cigar, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &m_cigar, &n_cigar, &cigar);
        } else if (!ez->zdropped && (flag & KSW_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
            ez->reach_end = 1;
            backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->mqe_t, qlen - 1, &m_cigar, &n_cigar, &cigar);
        } else if (ez->max_t >= 0 && ez->max_q >= 0) {
            backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &m_cigar, &n_cigar, &cigar);
        }
        free(p); free(off);
    }
    ez->n_cigar = n_cigar;
    for (t = 0; t < n_cigar && t < cigar_cap; ++t) cigar_out[t] = cigar[t];
    free(cigar);
    return n_cigar;
}
